#!/usr/bin/env python3
"""GPU-side probe: iteration counts / cond trajectory of the HIP fit vs the f64 oracle."""
import os, sys, time, json
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa
from oracle.oracle import Oracle

print("affinity cpus:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
Q = 21
N, L = 3000, 64
msa, _ = synthetic_msa(N, L, seed=42)
lj = plm.default_lambda_j(L, Q)
orc = Oracle("f64")
for eps in (1e-3, 1e-4):
    t = time.time(); ref = orc.fit(msa, Q, lambda_j=lj, max_iter=2000, epsilon=eps); to = time.time() - t
    t = time.time(); res = plm.fit(msa, Q, lambda_j=lj, max_iter=2000, epsilon=eps); tg = time.time() - t
    print("eps=%g oracle: iters=%d evals=%d status=%d fx=%.6f (%.1fs) | hip: iters=%d evals=%d status=%d fx=%.6f (%.2fs) | max|dcn|=%.3g" % (
        eps, ref["iters"], ref["nevals"], ref["status"], ref["fx"], to, res["iters"], res["n_evals"], res["status"], res["fx"], tg,
        np.abs(res["cn"] - ref["cn"]).max()))
# gradient accuracy near the optimum
x = ref["x"].astype(np.float32)
w = ref["weights"].astype(np.float32)
fx, nll, g = plm.evaluate(msa, w, Q, 0.01, lj, x)
fxo, nllo, go = orc.eval(msa, w.astype(np.float64), Q, 0.01, lj, x.astype(np.float64))
print("near optimum: |g_oracle|=%.4g |g_hip - g_oracle|=%.4g |x|=%.4g fx diff=%.4g (fx=%.1f)" % (
    np.linalg.norm(go), np.linalg.norm(g - go), np.linalg.norm(x), fx - fxo, fxo))
# headline trajectory
N, L = 50000, 300
msa, _ = synthetic_msa(N, L, seed=20260922)
t = time.time()
res = plm.fit(msa, Q, max_iter=700, epsilon=1e-4, want_fij=False)
print("headline: iters=%d evals=%d status=%d (%s) %.2fs" % (res["iters"], res["n_evals"], res["status"], res["status_msg"], time.time() - t))
for r in res["table"]:
    if r[0] <= 5 or r[0] % 25 == 0 or r[0] == res["iters"]:
        print("  it=%4d t=%7.3f cond=%.3e fx=%.4f nll=%.4f |h|=%.3f |e|=%.3f" % r)
