#!/usr/bin/env python3
"""GPU-side, under `rocprofv3 --kernel-trace --stats`: ONE shard of the G-shard layout runs plm_ctx_time_kernels, so the
per-kernel table shows where a rank's fixed costs sit.  usage: shard_trace.py G r [N L]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
G, r = int(sys.argv[1]), int(sys.argv[2])
N = int(sys.argv[3]) if len(sys.argv) > 4 else 50000
L = int(sys.argv[4]) if len(sys.argv) > 4 else 300
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
rng = np.random.default_rng(1)
x0 = (0.01 * rng.normal(size=plm.n_params(L, 21))).astype(np.float32)
kw = dict(n_shards=G, shard=r, sharded_state=True) if G > 1 else {}
with plm.PlmContext(msa, q=21, max_iter=20, epsilon=1e-3, **kw) as c:
    c.set_weights(np.full(N, 0.9, np.float32))
    c.set_x(x0)
    km = c.time_kernels(reps=int(os.environ.get("PLM_REPS", 5)))
print({k: round(v, 4) for k, v in km.items()})
