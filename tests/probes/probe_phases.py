#!/usr/bin/env python3
"""Phase breakdown of k_fwd / k_bwd from a -DPLM_PROBE=1 build (PLM_HIP_LIB=<probe .so>): per-wave cycles
spent waiting (pre-barrier vmcnt + barrier) and in the epilogue (headline workload)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm, _lib
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
N = int(os.environ.get("PLM_N", 50000)); L = int(os.environ.get("PLM_L", 300))
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
ctx = plm.PlmContext(msa, q=21, max_iter=2, epsilon=1e-3)
ctx.set_weights(np.full(N, 0.9, np.float32)); ctx.marginals(pairs=False); ctx.set_x(None)
ctx.optimize()
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
lib.plm_probe_read(buf, 1)
for _ in range(3):
    ctx.eval()
lib.plm_probe_read(buf, 0)
for k, name in enumerate(("k_fwd", "k_bwd")):
    wait, tot, epi, waves = buf[8 * k + 0], buf[8 * k + 3], buf[8 * k + 4], max(1, buf[8 * k + 5])
    print("%s: waves %d  per-wave cycles: total %.0f | pre-barrier vmcnt wait + barrier %.0f (%.1f%%) epilogue %.0f (%.1f%%)"
          % (name, waves, tot / waves, wait / waves, 100 * wait / max(1, tot), epi / waves, 100 * epi / max(1, tot)))
