#!/usr/bin/env python3
"""GPU probe: where does the rounding noise of the f32-class gradient come from at BASELINE scale?

Runs the headline fit for a while, then compares the HIP gradient at the reached point with the f64 oracle and
with HIP evaluations whose roundings were moved without changing the mathematics (PLM_JEXP_BIAS: forward
pass; PLM_KSPLIT: backward accumulation chains).  Test infrastructure (uses oracle/)."""
import os, sys, time, json
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
from oracle.oracle import Oracle

Q = 21
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ITERS = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
msa, _ = synthetic_msa(N, L, seed=BASE_SEED)
nh = L * Q


def split(v):
    return float(np.linalg.norm(v[:nh])), float(np.linalg.norm(v[nh:]))


def gpu_eval(x, w, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = str(v)
    try:
        with plm.PlmContext(msa, Q) as c:
            c.set_weights(w)
            c.set_x(x)
            fx, nll = c.eval()
            g = c.get_g()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return fx, g


out = {"N": N, "L": L}
with plm.PlmContext(msa, Q, max_iter=ITERS, epsilon=1e-5) as ctx:
    w, counts, neff = ctx.reweight()
    ctx.marginals(pairs=False)
    ctx.set_x(None)
    t = time.time()
    r = ctx.optimize()
    out["fit"] = dict(iters=r["iters"], evals=r["n_evals"], status=r["status"], seconds=time.time() - t,
                      final_cond=r["table"][-1][2] if r["table"] else None)
    x = ctx.get_x()
print("fit:", out["fit"], flush=True)
xn = float(np.linalg.norm(x))
fx0, g0 = gpu_eval(x, w)
t = time.time()
orc = Oracle("f64")
fxo, nllo, go = orc.eval(msa, w.astype(np.float64), Q, 0.01, plm.default_lambda_j(L, Q), x.astype(np.float64))
out["oracle_seconds"] = time.time() - t
out["x_norm"] = xn
out["g_oracle"] = split(go)
out["fx_rel_err"] = abs(fx0 - fxo) / abs(fxo)
d = g0 - go
out["err_vs_oracle"] = dict(h=split(d)[0], J=split(d)[1], total=float(np.linalg.norm(d)), cond_units=float(np.linalg.norm(d)) / xn,
                            max_abs=float(np.abs(d).max()), max_abs_over_gmax=float(np.abs(d).max() / np.abs(go).max()))
print(json.dumps(out), flush=True)
for name, env in (("repeat", {}), ("jexp-1", {"PLM_JEXP_BIAS": -1}), ("jexp-2", {"PLM_JEXP_BIAS": -2}),
                  ("ksplit16", {"PLM_KSPLIT": 16}), ("ksplit1", {"PLM_KSPLIT": 1})):
    fx1, g1 = gpu_eval(x, w, env)
    dd, do = g1 - g0, g1 - go
    out[name] = dict(vs_base=split(dd), vs_oracle=split(do), fx_diff=fx1 - fx0)
    print(name, out[name], flush=True)
# random point of the same scale: is the error a property of the point?
rng = np.random.default_rng(5)
xr = (x + rng.normal(0, 0.02, x.size).astype(np.float32) * np.abs(x).mean()).astype(np.float32)
fxr, gr = gpu_eval(xr, w)
fxro, _, gro = orc.eval(msa, w.astype(np.float64), Q, 0.01, plm.default_lambda_j(L, Q), xr.astype(np.float64))
out["random_point"] = dict(err=split(gr - gro), g=split(gro), fx_rel_err=abs(fxr - fxro) / abs(fxro))
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/noise_probe_%d_%d.json" % (N, L), "w"), indent=1)
