#!/usr/bin/env python3
"""CPU lab (torch float64) for the FIELD SOLVER of the variable-projection fit: how many passes over
the stored potentials does an evaluation need?  Models what plm_host.cpp / k_hsolve do -- sampled Hessian sums, cached
inverses, the step cap, rounds with a host check between them, the L-BFGS extrapolation of the fields as warm start --
and variants of it (per-site BFGS updates of the cached inverse, exact Hessians).  Test infrastructure: not imported by
the product.

usage: field_solver_lab.py N L variant[,variant...] [iterations]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from evcouplings_amd.synthetic import synthetic_msa
from oracle.oracle import Oracle

torch.set_num_threads(int(os.environ.get("LAB_THREADS", "8")))
Q = 21
N, L = int(sys.argv[1]), int(sys.argv[2])
variants = sys.argv[3].split(",")
ITERS = int(sys.argv[4]) if len(sys.argv) > 4 else 40
EPS = 1e-3
CAP = float(os.environ.get("LAB_CAP", "3.0"))
SAMPLE = int(os.environ.get("LAB_SAMPLE", "4"))      # every SAMPLE-th tile of 256 sequences carries Hessian sums
msa, _ = synthetic_msa(N, L, seed=42)
orc = Oracle("f64")
w_np = 1.0 / orc.reweight(msa, 0.8)
neff = w_np.sum()
lh, lj = 0.01, 0.01 * (Q - 1) * (L - 1)
D = L * Q
X = torch.zeros((N, D), dtype=torch.float64)
X[torch.arange(N)[:, None], torch.arange(L)[None, :] * Q + torch.from_numpy(msa.astype(np.int64))] = 1.0
Xi = X.reshape(N, L, Q)
w = torch.from_numpy(w_np)
fi = (X * w[:, None]).sum(0).reshape(L, Q) / neff
site = torch.arange(D) // Q
upper = site[:, None] < site[None, :]
nJ = int(upper.sum())
nh = D
tile = torch.arange(N) // 256
samp = (tile % SAMPLE) == 0
samp_scale = float(tile.max() + 1) / float(len(torch.unique(tile[samp])))
EYE = torch.eye(Q, dtype=torch.float64)[None]


class Solver:
    def __init__(self, variant):
        self.variant = variant
        self.hinv = None
        self.age = -1
        self.refresh_next = False
        self.passes = 0          # passes over HJ
        self.hess_passes = 0
        self.steps = 0
        self.rounds = 0
        self.newton = 2
        self.gprev = None
        self.hprev = None
        self.cost = 0.0          # ms at the headline: plain pass 0.33, with sampled Hessian sums 0.41, with residual planes 0.52
        self.c_prev = 3          # chain: steps the previous evaluation needed
        self.extra_final = 0

    def stats(self, HJ, h, hess, rt=False):
        self.passes += 1
        self.cost += 0.52 if rt else (0.41 if hess else 0.33)
        H = HJ + h[None]
        lse = torch.logsumexp(H, dim=2)
        P = torch.exp(H - lse[:, :, None])
        g = (w[:, None, None] * (P - Xi)).sum(0) + 2 * lh * h
        Hh = None
        if hess:
            self.hess_passes += 1
            if "exact" in self.variant:
                wP = w[:, None, None] * P
                Hh = torch.diag_embed(wP.sum(0)) - torch.einsum("sia,sib->iab", wP, P) + 2 * lh * EYE
            else:
                wP = (w[:, None, None] * P)
                M = torch.einsum("sia,sib->iab", wP[samp], P[samp]) * samp_scale
                if "xdiag" in self.variant:
                    # round 6 experiment: the DIAGONAL second-order sums M_aa = sum_s w P_a^2 from every tile (exact), the
                    # off-diagonal ones sampled and rescaled symmetrically until their row sums are m_a - M_aa
                    m = wP.sum(0)
                    Md = (wP * P).sum(0)
                    off = M - torch.diag_embed(torch.diagonal(M, dim1=1, dim2=2))
                    tgt = (m - Md).clamp_min(0)
                    D = torch.ones_like(m)
                    for _ in range(3):
                        rs = (D[:, :, None] * off * D[:, None, :]).sum(2)
                        D = D * torch.sqrt(tgt / rs.clamp_min(1e-300))
                    M = D[:, :, None] * off * D[:, None, :] + torch.diag_embed(Md)
                elif "sink" in self.variant:
                    # k_hsolve (round 5): the sampled second-order sums rescaled symmetrically until their row sums are
                    # the exact first-order sums m = sum_s w P (three Sinkhorn sweeps)
                    m = wP.sum(0)
                    D = torch.ones_like(m)
                    for _ in range(3):
                        rs = (D[:, :, None] * M * D[:, None, :]).sum(2)
                        D = D * torch.sqrt(m / rs.clamp_min(1e-300))
                    M = D[:, :, None] * M * D[:, None, :]
                Hh = torch.diag_embed(M.sum(2)) - M + 2 * lh * EYE
        return g, Hh, (H, lse, P)

    def step(self, h, g):
        dh = torch.einsum("iab,ib->ia", self.hinv, g)
        mx = dh.abs().max(1).values
        cap = torch.where(mx > CAP, CAP / mx, torch.ones_like(mx))
        return h - cap[:, None] * dh

    def bfgs_update(self, s, y):
        # per-site BFGS update of the inverse: Hinv <- (I - r s y^T) Hinv (I - r y s^T) + r s s^T
        sy = (s * y).sum(1)
        ok = sy > 1e-12 * (y * y).sum(1).clamp_min(1e-300)
        r = torch.where(ok, 1.0 / sy.clamp_min(1e-300), torch.zeros_like(sy))
        Hy = torch.einsum("iab,ib->ia", self.hinv, y)
        yHy = (y * Hy).sum(1)
        upd = (-r[:, None, None] * (s[:, :, None] * Hy[:, None, :] + Hy[:, :, None] * s[:, None, :])
               + (r * r * yHy + r)[:, None, None] * s[:, :, None] * s[:, None, :])
        self.hinv = self.hinv + torch.where(ok[:, None, None], upd, torch.zeros_like(upd))

    def solve_chain(self, HJ, h, tol2):
        """one device-side chain: every pass = gradient sums (+ sampled Hessian sums while the previous evaluation says the
        solver is still far), per-site step for the sites above their share of the tolerance; the pass at which no site
        moves is final.  Residual planes are written by the passes from the predicted last one on; a chain that converges
        earlier pays one residual-only pass."""
        v = self.variant
        tol_site2 = 4.0 * tol2 / L
        hess_upto = self.c_prev if self.c_prev >= 2 else (1 if (self.age < 0 or self.age >= 64) else 0)
        if "allhess" in v: hess_upto = 99
        rt_from = max(0, self.c_prev - (1 if "early" in v else 0))
        hist = []
        for p in range(40):
            rt = p >= rt_from
            if "dev" in v:      # the device predicts from the passes of THIS chain whether the next one will be the last
                if p == 0: rt = self.c_prev == 0
                else:
                    rate = min(1.0, hist[-1] / hist[-2]) if p >= 2 else float(os.environ.get("LAB_RATE0", "1e-3"))
                    rt = hist[-1] * max(1e-6, rate) <= tol2 * float(os.environ.get("LAB_MARGIN", "1"))
            hess = (p < hess_upto or self.age < 0) and not rt
            g, Hh, fin = self.stats(HJ, h, hess, rt)
            if hess:
                self.hinv = torch.linalg.inv(Hh)
                self.age = 0
            else:
                self.age += 1
            g2 = (g * g).sum(1)
            hist.append(float(g2.sum()))
            move = g2 > tol_site2
            if os.environ.get("LAB_TRACE"):
                print("      pass %d hess %d rt %d |g_h| %.3e tol %.3e open sites %d" % (p, hess, rt, float(g2.sum()) ** 0.5, tol2 ** 0.5, int(move.sum())))
            if not bool(move.any()):
                break
            dh = torch.einsum("iab,ib->ia", self.hinv, g)
            mx = dh.abs().max(1).values
            cap = torch.where(mx > CAP, CAP / mx, torch.ones_like(mx))
            h = torch.where(move[:, None], h - cap[:, None] * dh, h)
            self.steps += 1
        if not rt:
            self.cost += 0.50
            self.passes += 1
            self.extra_final += 1
        self.rounds += 1
        self.c_prev = p
        return h, fin, float(g2.sum())

    def solve(self, HJ, h, tol2):
        """-> fields, (H, lse, P) at them, |g_h|^2"""
        v = self.variant
        if v.startswith("chain"):
            return self.solve_chain(HJ, h, tol2)
        bfgs = "bfgs" in v
        prev, gh2 = float("inf"), float("inf")
        newton = self.newton
        have = None
        rounds = 0
        for rnd in range(64):
            if bfgs:
                refresh = self.age < 0 or (rnd == 0 and (self.age >= 64 or self.refresh_next)) or (rnd > 0 and "rr" in v and self.age > 0)
            else:
                refresh = self.age < 0 or ((self.age >= 64 or self.refresh_next) if rnd == 0 else self.age > 0)
            for it in range(newton):
                full = it == 0 and refresh
                if full or have is None:
                    g, Hh, _ = self.stats(HJ, h, full)
                else:
                    g = have
                have = None
                if bfgs and self.gprev is not None and not full:
                    self.bfgs_update(h - self.hprev, g - self.gprev)
                if full:
                    self.hinv = torch.linalg.inv(Hh)
                    self.age = 0
                else:
                    self.age += 1
                self.gprev, self.hprev = g, h
                h = self.step(h, g)
                self.steps += 1
            g, _, fin = self.stats(HJ, h, False, rt=(rnd <= 1 or gh2 * (max(1e-6, gh2 / prev) if (rnd > 1 and prev < float("inf") and gh2 < prev) else 1e-6) <= tol2))
            rt_written = (rnd <= 1 or gh2 * (max(1e-6, gh2 / prev) if (rnd > 1 and prev < float("inf") and gh2 < prev) else 1e-6) <= tol2)
            if bfgs and self.gprev is not None:
                self.bfgs_update(h - self.hprev, g - self.gprev)
                self.gprev, self.hprev = g, h
            have = g
            prev, gh2 = gh2, float((g * g).sum())
            rounds = rnd
            if os.environ.get("LAB_TRACE"):
                print("      round %d newton %d |g_h| %.3e tol %.3e" % (rnd, newton, gh2 ** 0.5, tol2 ** 0.5))
            if not (gh2 > tol2) or rnd >= 8 or (rnd > 1 and gh2 > 0.25 * prev):
                break
            newton = 2
        if not rt_written:
            self.cost += 0.50
            self.passes += 1
        self.rounds += rounds + 1
        self.refresh_next = rounds > 0
        if rounds > 0:
            self.newton = min(3, self.newton + 1)
        elif gh2 < 0.25 * tol2:
            self.newton = max(0, self.newton - 1)
        # the cached pair must not straddle two coupling vectors
        self.gprev = None
        return h, fin, gh2


def run(variant):
    sol = Solver(variant)
    h0 = torch.log(fi + 1.0 / neff)
    h0 = h0 - h0.mean(1, keepdim=True)
    n = nh + nJ

    def evaluate(x, first=False):
        W = torch.zeros((D, D), dtype=torch.float64); W[upper] = x[nh:]; W = W + W.T
        HJ = (X @ W).reshape(N, L, Q)
        xn = max(1.0, float(x.norm()))
        tol = max(0.1 * EPS * xn, 2e-7 * (neff * L * Q) ** 0.5)
        if first:
            sol.newton = 3
        h, (H, lse, P), gh2 = sol.solve(HJ, x[:nh].reshape(L, Q), tol * tol)
        if first:
            sol.newton = 2
        x = x.clone(); x[:nh] = h.reshape(-1)
        Hx = (H * Xi).sum(2)
        nll = -(w[:, None] * (Hx - lse)).sum()
        R = (w[:, None, None] * P).reshape(N, D) - w[:, None] * X
        G = X.T @ R
        g = torch.zeros(n, dtype=torch.float64)
        g[nh:] = (G + G.T)[upper] + 2 * lj * x[nh:]
        fx = nll + lh * (h ** 2).sum() + lj * (x[nh:] ** 2).sum()
        return x, fx.item(), g, gh2

    x = torch.zeros(n, dtype=torch.float64); x[:nh] = h0.reshape(-1)
    x, fx, g, gh2 = evaluate(x, True)
    nev = 1
    S, Y = [], []
    d = -g / g.norm()
    m = 6
    log = []
    for k in range(1, ITERS + 1):
        t, dg0 = 1.0, (g @ d).item()
        p0 = sol.passes
        c0 = sol.cost
        while True:
            xt, fn, gn, gh2 = evaluate(x + t * d); nev += 1
            dgn = (gn @ (xt - x)).item() / t if False else (gn @ d).item()
            if fn <= fx + 1e-4 * t * dg0 and abs(dgn) <= 0.9 * abs(dg0): break
            if fn <= fx + 1e-4 * t * dg0 and dgn < 0:
                t *= 2.0
                if t > 64: break
                continue
            t *= 0.5
            if t < 1e-10: print("ls fail"); return
        s, y = xt - x, gn - g
        x, fx, g = xt, fn, gn
        cond = (g.norm().item() ** 2 + gh2) ** 0.5 / max(1.0, x.norm().item())
        log.append((k, nev, sol.passes - p0, cond, sol.cost - c0))
        if os.environ.get("LAB_TRACE") or k % 10 == 0:
            print("  %s it=%3d ev=%3d passes(this it)=%2d total passes=%4d hess=%3d steps=%4d rounds=%3d cond=%.3e" % (
                variant, k, nev, sol.passes - p0, sol.passes, sol.hess_passes, sol.steps, sol.rounds, cond), flush=True)
        if cond <= EPS: break
        if (y @ s).item() > 1e-12 * (y @ y).item():
            S.append(s); Y.append(y)
            if len(S) > m: S.pop(0); Y.pop(0)
        q = -g.clone(); al = []
        for s_, y_ in zip(reversed(S), reversed(Y)):
            a = ((s_ @ q) / (y_ @ s_)).item(); al.append(a); q -= a * y_
        if S: q *= ((S[-1] @ Y[-1]) / (Y[-1] @ Y[-1])).item()
        for (s_, y_), a in zip(zip(S, Y), reversed(al)):
            b = ((y_ @ q) / (y_ @ s_)).item(); q += (a - b) * s_
        d = q
    print("%-10s N=%d L=%d: iters=%d evals=%d passes=%d (%.2f / eval) hessian passes=%d newton steps=%d rounds=%d cond=%.3e" % (
        variant, N, L, k, nev, sol.passes, sol.passes / nev, sol.hess_passes, sol.steps, sol.rounds, log[-1][3]))
    win = [r for r in log if 6 <= r[0] <= 25]
    if win:
        print("           iterations 6-25: %.2f passes / iteration, %.2f ms of passes / iteration; whole run %.2f ms / evaluation; extra final passes %d" % (
            sum(r[2] for r in win) / len(win), sum(r[4] for r in win) / len(win), sol.cost / nev, sol.extra_final))


for v in variants:
    t0 = time.time()
    run(v)
    print("           (%.0f s)" % (time.time() - t0))
