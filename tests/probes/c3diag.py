import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
msa, _ = synthetic_msa(100000, 300, seed=BASE_SEED + 3)
with plm.PlmContext(msa, q=21, max_iter=1000, epsilon=1e-3) as ctx:
    ctx.reweight(); ctx.marginals(pairs=False); ctx.set_x(None)
    t = time.time(); r = ctx.optimize(); tf = time.time() - t
print("fit %.2f s, %d it, %d ev, %s" % (tf, r["iters"], r["n_evals"], r["status_msg"]))
tab = r["table"]
prev_t = 0
for row in tab:
    if row[0] % 10 == 0 or row[0] > len(tab) - 5 or abs(row[6] - 1.0) > 1e-6:
        print("it %4d t %.2f dt %.3f cond %.4e fx %.4f step %.4g" % (row[0], row[1], row[1] - prev_t, row[2], row[3], row[6]))
    prev_t = row[1]
