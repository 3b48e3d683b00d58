#!/usr/bin/env python3
"""GPU-side probe: error of the forward GEMM's potentials HJ[s,i,a] = sum_{j != i} J_ij(a, x_sj) AT THE POINT A FIT STOPS, plain
and accurate instantiation (PLM_FWD_ACCURATE = 0 | 1), against a float64 one-hot GEMM in numpy on a sample of sequences:
random part (averages out in the gradient sums) and the part that is the same for every sequence of a (site, state)
(adds up N-fold).  Then, in float64 on the host, what that error alone does to the softmax and to the field gradient.
usage: potentials_probe.py [headline|config3] [NS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
CONFIGS = {"config2": (20000, 200, 2), "headline": (50000, 300, 1), "config3": (100000, 300, 3)}
name = sys.argv[1] if len(sys.argv) > 1 else "headline"
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
N, L, k = CONFIGS[name]
q = 21
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + k)
with plm.PlmContext(msa, q, max_iter=3000, epsilon=1e-3) as ctx:
    w, _, neff = ctx.reweight(); ctx.marginals(pairs=False); ctx.set_x(None)
    r = ctx.optimize()
    x = ctx.get_x()
hi = x[:L * q].reshape(L, q)
jij = x[L * q:].reshape(-1, q, q)
sub = np.ascontiguousarray(msa[:NS])
t = time.time()
J = np.zeros((L, q, L, q))
iu, ju = np.triu_indices(L, 1)
J[iu, :, ju, :] = jij.astype(np.float64)
J[ju, :, iu, :] = np.transpose(jij.astype(np.float64), (0, 2, 1))
X = np.zeros((NS, L, q)); X[np.arange(NS)[:, None], np.arange(L)[None, :], sub] = 1.0
ref = (X.reshape(NS, L * q) @ J.reshape(L * q, L * q)).reshape(NS, L, q)
print("%s: fit %d iterations (%s); |x| %.1f; f64 reference GEMM on %d sequences %.1f s; mean |HJ| %.3f, max |HJ| %.2f; mean |HJ + h| %.3f" % (
    name, r["iters"], r["status_msg"][:30], np.linalg.norm(x), NS, time.time() - t, np.abs(ref).mean(), np.abs(ref).max(),
    np.abs(ref + hi[None].astype(np.float64)).mean()), flush=True)
H = ref + hi[None].astype(np.float64)
P = np.exp(H - H.max(2, keepdims=True)); P /= P.sum(2, keepdims=True)
for mode in ("0", "1"):
    os.environ["PLM_FWD_ACCURATE"] = mode
    hj = plm.potentials(sub, q, hi, jij).astype(np.float64)
    err = hj - ref
    col = err.mean(0)                                   # per (site, state): the part common to all sequences
    Hh = hj + hi[None].astype(np.float64)
    Ph = np.exp(Hh - Hh.max(2, keepdims=True)); Ph /= Ph.sum(2, keepdims=True)
    dP = Ph - P
    gh = dP.sum(0) * (N / float(NS))                    # field-gradient error this forward error alone would cause (w = 1)
    print("  forward %s: err rms %.3e  mean %.2e  mean(err*sign) %.2e | per-(i,a) mean over sequences: rms %.3e (random alone: %.3e)"
          " | dP rms %.3e, coherent dP (mean over s) rms %.3e | implied field-gradient error / |x| = %.3e" % (
              "accurate" if mode == "1" else "plain   ", err.std(), err.mean(), (err * np.sign(ref)).mean(),
              np.sqrt((col ** 2).mean()), err.std() / np.sqrt(NS), dP.std(), np.sqrt((dP.mean(0) ** 2).mean()),
              np.linalg.norm(gh) / np.linalg.norm(x)), flush=True)
    # the f32 rounding of the returned array itself, for scale
    r32 = ref.astype(np.float32).astype(np.float64) - ref
    if mode == "1":
        print("  (f32 rounding of the exact potentials: rms %.3e, per-(i,a) mean rms %.3e)" % (r32.std(), np.sqrt((r32.mean(0) ** 2).mean())))
os.environ.pop("PLM_FWD_ACCURATE", None)
