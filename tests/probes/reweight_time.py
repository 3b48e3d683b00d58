import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
for N, L, k in ((50000, 300, 1), (100000, 300, 3), (50000, 500, 4), (30000, 600, 5), (20000, 200, 2)):
    msa, _ = synthetic_msa(N, L, seed=BASE_SEED + k)
    with plm.PlmContext(msa, q=21, max_iter=5, epsilon=1e-3) as ctx:
        ctx.reweight()
        km = ctx.time_kernels(reps=3)
    print("N=%d L=%d reweight %.3f ms" % (N, L, km["reweight"]), flush=True)
