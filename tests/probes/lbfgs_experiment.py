#!/usr/bin/env python3
"""CPU experiment (oracle eval, python L-BFGS): effect of history m and a diagonal H0."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd.synthetic import synthetic_msa
from oracle.oracle import Oracle

Q = 21
N, L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 48
msa, _ = synthetic_msa(N, L, seed=42)
orc = Oracle("f64")
w = 1.0 / orc.reweight(msa, 0.8)
neff = w.sum()
fi, fij = orc.marginals(msa, w, Q)
lh, lj = 0.01, 0.01 * 20 * (L - 1)
nh = L * Q

def f(x):
    fx, nll, g = orc.eval(msa, w, Q, lh, lj, x)
    return fx, g

def start():
    x = np.zeros(nh + fij.size)
    h = np.log(fi + 1.0 / neff); h -= h.mean(axis=1, keepdims=True)
    x[:nh] = h.ravel()
    return x

def diag_hess():
    # independent-site approximation of the Hessian diagonal
    dh = neff * fi * (1 - fi) + 2 * lh
    # d2/dJ_ij(a,b)^2 = sum_s w [x_sj=b] P_si(a)(1-P_si(a)) + sym  ~ neff*( f_j(b) f_i(a)(1-f_i(a)) + f_i(a) f_j(b)(1-f_j(b)) )
    iu, ju = np.triu_indices(L, 1)
    fa = fi[iu][:, :, None]; fb = fi[ju][:, None, :]
    dj = neff * (fb * fa * (1 - fa) + fa * fb * (1 - fb)) + 2 * lj
    return np.concatenate([dh.ravel(), dj.ravel()])

def lbfgs(m, precond, eps=1e-3, maxit=3000):
    x = start(); fx, g = f(x); nev = 1
    D = 1.0 / diag_hess() if precond else None
    S, Y = [], []
    d = -g * (D if precond else 1.0 / np.linalg.norm(g))
    for k in range(1, maxit + 1):
        t, dg0 = 1.0, g @ d
        if dg0 >= 0: d = -g; dg0 = g @ d
        while True:   # backtracking + weak curvature check (enough for an iteration-count comparison)
            xn = x + t * d; fn, gn = f(xn); nev += 1
            if fn <= fx + 1e-4 * t * dg0: break
            t *= 0.5
            if t < 1e-12: return k, nev, fx
        s, y = xn - x, gn - g
        x, fx, g = xn, fn, gn
        if np.linalg.norm(g) / max(1, np.linalg.norm(x)) <= eps: return k, nev, fx
        if y @ s > 1e-12 * (y @ y):
            S.append(s); Y.append(y)
            if len(S) > m: S.pop(0); Y.pop(0)
        q = -g.copy(); al = []
        for s_, y_ in zip(reversed(S), reversed(Y)):
            a = (s_ @ q) / (y_ @ s_); al.append(a); q -= a * y_
        if precond: q *= D
        elif S: q *= (S[-1] @ Y[-1]) / (Y[-1] @ Y[-1])
        for (s_, y_), a in zip(zip(S, Y), reversed(al)):
            b = (y_ @ q) / (y_ @ s_); q += (a - b) * s_
        d = q
    return maxit, nev, fx

for m, pc in ((6, False), (20, False), (6, True), (20, True)):
    t = time.time(); k, nev, fx = lbfgs(m, pc)
    print("N=%d L=%d m=%2d precond=%-5s -> iters=%4d evals=%4d fx=%.6f (%.0fs)" % (N, L, m, pc, k, nev, fx, time.time() - t), flush=True)
