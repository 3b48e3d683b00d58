#!/usr/bin/env python3
"""Wall time of plm.mean_field (mean-field DCA): first call (loads rocSOLVER), a repeat, and the
headline shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
def timed(label, msa, **kw):
    t0 = time.perf_counter(); out = plm.mean_field(msa, 21, **kw); dt = time.perf_counter() - t0
    print("%-28s N=%d L=%d: %.3f s" % (label, msa.shape[0], msa.shape[1], dt), flush=True)
    return out
small, _ = synthetic_msa(500, 40, seed=3)
timed("first call", small)
timed("second call", small)
N = int(os.environ.get("PLM_N", 50000)); L = int(os.environ.get("PLM_L", 300))
big, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
timed("headline, all outputs", big)
out = timed("headline, no dense J", big, want_full=False)
top = np.dstack(np.unravel_index(np.argsort(-np.triu(out["di"], 6), axis=None)[:5], out["di"].shape))[0]
print("top DI pairs (|i-j| >= 6):", top.tolist())
