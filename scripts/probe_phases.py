#!/usr/bin/env python3
"""Phase breakdown of k_fwd / k_bwd from a -DPLM_PROBE=1 build (PLM_HIP_LIB=<probe .so>): per-wave cycles
spent in the pre-barrier vmcnt wait, the barrier, the MFMA section and the epilogue (headline workload)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evcouplings_amd import plm, _lib
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
N = int(os.environ.get("PLM_N", 50000)); L = int(os.environ.get("PLM_L", 300))
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
ctx = plm.PlmContext(msa, q=21, max_iter=2, epsilon=1e-12)
ctx.set_weights(np.full(N, 0.9, np.float32)); ctx.marginals(pairs=False); ctx.set_x(None)
ctx.optimize()
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
lib.plm_probe_read(buf, 1)
for _ in range(3):
    ctx.eval()
lib.plm_probe_read(buf, 0)
for k, name in enumerate(("k_fwd", "k_bwd")):
    vm, bar, mm, tot, epi, waves = [buf[8 * k + i] for i in range(6)]
    waves = max(1, waves)
    print("%s: waves %d  per-wave cycles: total %.0f | vmcnt %.0f (%.1f%%) barrier %.0f (%.1f%%) mfma-section %.0f (%.1f%%) "
          "epilogue %.0f (%.1f%%)" % (name, waves, tot / waves, vm / waves, 100 * vm / tot, bar / waves, 100 * bar / tot,
                                       mm / waves, 100 * mm / tot, epi / waves, 100 * epi / tot))
