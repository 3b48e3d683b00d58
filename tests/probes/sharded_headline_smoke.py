#!/usr/bin/env python3
"""GPU-side: sharded-state mode at the headline shape with all shards on one GPU (threads): checks the
8-way partition (19 site blocks -> 3,3,3,3,3,3,1,0) against the single-GPU result."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.dist import ThreadedShards
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
N = int(os.environ.get("PLM_N", 20000)); L = int(os.environ.get("PLM_L", 300)); its = int(os.environ.get("PLM_ITERS", 15))
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
t = time.time(); ref = plm.fit(msa, 21, max_iter=its, epsilon=1e-12, want_fij=False); t1 = time.time() - t
for n in (2, 8):
    ts = ThreadedShards(n)
    t = time.time(); outs = ts.fit(msa, q=21, max_iter=its, epsilon=1e-12, want_fij=False); tn = time.time() - t
    d = max(np.abs(o["cn"] - ref["cn"]).max() for o in outs)
    same = all(np.array_equal(o["cn"], outs[0]["cn"]) for o in outs)
    print("N=%d L=%d %d iterations: single %.2fs | %d shards (one GPU, threads) %.2fs  max|dcn| vs single %.2e  ranks identical: %s  fx %.6f vs %.6f  collectives: %s" % (
        N, L, its, t1, n, tn, d, same, outs[0]["fx"], ref["fx"], ts.n_calls))
