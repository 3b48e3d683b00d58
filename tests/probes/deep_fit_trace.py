import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa
N, L = 500000, 300
msa, planted = synthetic_msa(N, L, seed=N + L)
t = time.time()
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
r = plm.fit(msa, q=21, max_iter=int(sys.argv[1]), epsilon=eps, want_fij=False)
print("fit %.1f s, %d it / %d ev, status: %s" % (time.time() - t, r["iters"], r["n_evals"], r["status_msg"]))
tab = r["table"]
for row in tab[::50] + tab[-3:]:
    print("it %4d  t %7.2f  cond %.4e  fx %.8e" % (row[0], row[1], row[2], row[3]))
