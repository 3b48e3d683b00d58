#!/usr/bin/env python3
"""CPU lab (torch float64, dense one-hot GEMM): which L-BFGS variant reaches |g|/|x| < eps fastest on the
symmetric PLM objective.  Test infrastructure: not imported by the product.

usage: opt_lab.py N L variant[,variant...] [maxit] [eps]
variants: plain | diag0 (fixed independent-site diagonal H0) | diagx (exact diagonal, refreshed every R its)
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
MAXIT = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
EPS = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-3
msa, _ = synthetic_msa(N, L, seed=42)
orc = Oracle("f64")
w_np = 1.0 / orc.reweight(msa, 0.8)
neff = w_np.sum()
lh, lj = 0.01, 0.01 * (Q - 1) * (L - 1)
D = L * Q
X = torch.zeros((N, D), dtype=torch.float64)
X[torch.arange(N)[:, None], torch.arange(L)[None, :] * Q + torch.from_numpy(msa.astype(np.int64))] = 1.0
w = torch.from_numpy(w_np)
fi = (X * w[:, None]).sum(0).reshape(L, Q) / neff
site = torch.arange(D) // Q
upper = site[:, None] < site[None, :]          # (i,a),(j,b) with i<j
nJ = int(upper.sum())
nh = D


def unpack(x):
    h = x[:nh].reshape(L, Q)
    W = torch.zeros((D, D), dtype=torch.float64)
    W[upper] = x[nh:]
    return h, W + W.T


def evaluate(x, want_diag=False):
    h, W = unpack(x)
    H = (X @ W).reshape(N, L, Q) + h[None]
    lse = torch.logsumexp(H, dim=2)
    P = torch.exp(H - lse[:, :, None])
    Hx = (H.reshape(N, D) * X).reshape(N, L, Q).sum(2)
    nll = -(w[:, None] * (Hx - lse)).sum()
    R = (w[:, None, None] * P).reshape(N, D) - w[:, None] * X
    G = X.T @ R                                  # G[(j,b),(i,a)]
    gJ = (G + G.T)[upper] + 2 * lj * x[nh:]
    gh = R.sum(0) + 2 * lh * x[:nh]
    fx = nll + lh * (x[:nh] ** 2).sum() + lj * (x[nh:] ** 2).sum()
    g = torch.cat([gh, gJ])
    if not want_diag:
        return fx.item(), g
    V = (w[:, None, None] * P * (1 - P)).reshape(N, D)
    dh = V.sum(0) + 2 * lh
    C = X.T @ V                                  # C[(j,b),(i,a)] = sum_s w [x_sj=b] P_si(a)(1-P_si(a))
    dJ = (C + C.T)[upper] + 2 * lj
    return fx.item(), g, torch.cat([dh, dJ])


def start():
    x = torch.zeros(nh + nJ, dtype=torch.float64)
    h = torch.log(fi + 1.0 / neff)
    h = h - h.mean(1, keepdim=True)
    x[:nh] = h.reshape(-1)
    return x


def diag_indep():
    p = fi
    dh = (neff * p * (1 - p) + 2 * lh).reshape(-1)
    v = (neff * p * (1 - p)).reshape(-1)         # per (i,a)
    f = (fi).reshape(-1)
    C = f[:, None] * v[None, :]                  # [(j,b),(i,a)] ~ f_j(b) * N p_i(a)(1-p_i(a))
    dJ = (C + C.T)[upper] + 2 * lj
    return torch.cat([dh, dJ])


class KFac:
    """Block preconditioner: per site i and state a, (v_ia * C_{-i} + reg)^-1 on the feature index, where C is the
    weighted second-moment matrix of [kappa, one-hot(x)] and v_ia = p_i(a)(1-p_i(a)) at the start point.
    mode 'ideal': exact per-site eigendecomposition; 'shifts:k': full-C inverses at k geometric shifts (own block
    not excluded) -- the GPU-feasible approximation."""
    def __init__(self, mode):
        self.mode = mode
        p = torch.softmax(start()[:nh].reshape(L, Q), dim=1)
        self.v = (p * (1 - p)).clamp_min(1e-12)                    # (L,Q)
        self.kappa = (lj / (2 * lh)) ** 0.5
        Xb = torch.cat([torch.full((N, 1), self.kappa, dtype=torch.float64), X], dim=1)
        self.C = Xb.T @ (Xb * w[:, None])                            # (D+1, D+1)
        if mode == "ideal":
            self.eig = []
            for i in range(L):
                keep = torch.cat([torch.tensor([0]), 1 + torch.nonzero(site != i).squeeze(1)])
                c, V = torch.linalg.eigh(self.C[keep][:, keep])
                self.eig.append((keep, c.clamp_min(0), V))
        else:
            k = int(mode.split(":")[1])
            vv = self.v.reshape(-1)
            lo, hi = (lj / vv.max()).item(), (lj / vv.min().clamp_min(1e-6)).item()
            self.shifts = torch.logspace(np.log10(lo), np.log10(hi), k, dtype=torch.float64)
            self.inv = [torch.linalg.inv(self.C + t * torch.eye(D + 1, dtype=torch.float64)) for t in self.shifts]
            tau = lj / vv
            self.assign = torch.argmin((torch.log(tau)[:, None] - torch.log(self.shifts)[None, :]).abs(), dim=1)

    def apply(self, g):
        Gs = torch.zeros((D, D), dtype=torch.float64)
        Gs[upper] = g[nh:] * 0.5
        Gs = Gs + Gs.T                                               # [(j,b),(i,a)]
        B = torch.cat([(g[:nh] * self.kappa)[None, :], Gs], dim=0)   # (D+1, D): column (i,a), row 0 = bias
        out = torch.zeros_like(B)
        if self.mode == "ideal":
            for i in range(L):
                keep, c, V = self.eig[i]
                cols = slice(i * Q, (i + 1) * Q)
                b = V.T @ B[keep][:, cols]                           # (n, Q)
                b = b / (self.v[i][None, :] * c[:, None] + lj)
                out[keep, cols] = V @ b
        else:
            vv = self.v.reshape(-1)
            for k in range(len(self.shifts)):
                cols = torch.nonzero(self.assign == k).squeeze(1)
                if len(cols):
                    out[:, cols] = (self.inv[k] @ B[:, cols]) / vv[cols][None, :]
            for i in range(L):
                out[1 + i * Q:1 + (i + 1) * Q, i * Q:(i + 1) * Q] = 0
        ph = out[0] * self.kappa
        P = out[1:]
        pJ = 0.5 * (P + P.T)[upper]
        return torch.cat([ph, pJ])


class Gauge:
    """Gauge-aware preconditioner.  J_ij = Jhat + alpha 1^T + 1 beta^T + gamma: the row/column-mean parts act on the
    likelihood exactly like fields of site i / j, so per (i,a) the vector (h_i(a), alpha_ij(a) ..., beta_ji(a) ...)
    has Hessian  Lambda + d_ia 1 1^T  (d = data curvature of the field, Lambda = L2 terms): Sherman-Morrison.
    The zero-sum part gets the independent-site diagonal."""
    def __init__(self, mode):
        self.mode = mode
        p = torch.softmax(start()[:nh].reshape(L, Q), dim=1)
        self.d = neff * p * (1 - p)                                  # (L,Q) data curvature of fields
        iu, ju = torch.triu_indices(L, L, 1)
        self.iu, self.ju = iu, ju
        fa, fb = fi[iu][:, :, None], fi[ju][:, None, :]
        va, vb = (p * (1 - p))[iu][:, :, None], (p * (1 - p))[ju][:, None, :]
        self.dz = neff * (fb * va + fa * vb) + 2 * lj                # (pairs,Q,Q)
        self.npair = len(iu)
        # upper-mask order <-> (pair, a, b): x[nh:] is W[upper] in row-major order of the D x D matrix
        idx = torch.zeros((D, D), dtype=torch.long)
        idx[upper] = torch.arange(nJ)
        self.perm = torch.stack([idx[i * Q:(i + 1) * Q, j * Q:(j + 1) * Q] for i, j in zip(iu.tolist(), ju.tolist())])

    def apply(self, g):
        gh = g[:nh].reshape(L, Q)
        gJ = g[nh:][self.perm]                                       # (pairs,Q,Q) [a (site i), b (site j)]
        gam = gJ.mean(dim=(1, 2), keepdim=True)
        r = gJ.mean(dim=2, keepdim=True) - gam                       # (pairs,Q,1)  row part (acts on site i)
        c = gJ.mean(dim=1, keepdim=True) - gam                       # (pairs,1,Q)  col part (acts on site j)
        zs = gJ - r - c - gam
        l0, l1 = 2 * lh, 2 * lj * Q
        # per site/state: v0 = gh, v_k = Q * (row or col part) for the L-1 partners
        vsum = torch.zeros((L, Q), dtype=torch.float64)              # 1^T Lambda^-1 v
        vsum += gh / l0
        vsum.index_add_(0, self.iu, Q * r[:, :, 0] / l1)
        vsum.index_add_(0, self.ju, Q * c[:, 0, :] / l1)
        S = 1.0 / l0 + (L - 1) / l1
        corr = self.d * vsum / (1 + self.d * S)                      # (L,Q)
        dh = gh / l0 - corr / l0
        dalpha = Q * r[:, :, 0] / l1 - corr[self.iu] / l1            # (pairs,Q)
        dbeta = Q * c[:, 0, :] / l1 - corr[self.ju] / l1
        dgam = gam / (2 * lj)
        z = zs / self.dz
        z = z - z.mean(dim=2, keepdim=True) - z.mean(dim=1, keepdim=True) + z.mean(dim=(1, 2), keepdim=True)
        dJ = z + dalpha[:, :, None] + dbeta[:, None, :] + dgam
        out = torch.zeros_like(g)
        out[:nh] = dh.reshape(-1)
        out[nh:][self.perm.reshape(-1)] = dJ.reshape(-1)
        return out


def lbfgs(variant, m=6, refresh=100):
    x = start()
    nev = 1
    op = None
    if variant.startswith("gauge"):
        op = Gauge(variant)
        fx, g = evaluate(x)
        Dinv = None
    elif variant.startswith("kfac"):
        op = KFac(variant.split("-", 1)[1])
        fx, g = evaluate(x)
        Dinv = None
    elif variant.startswith("diagx"):
        fx, g, dd = evaluate(x, True)
        Dinv = 1.0 / dd
    else:
        fx, g = evaluate(x)
        Dinv = 1.0 / diag_indep() if variant.startswith("diag0") else None
    if ":" in variant:
        refresh = int(variant.split(":")[1])
    S, Y = [], []
    d = -g * (Dinv if Dinv is not None else 1.0 / g.norm())
    if op is not None:
        d = -op.apply(g)
    t0 = time.time()
    marks = {}
    for k in range(1, MAXIT + 1):
        t, dg0 = 1.0, (g @ d).item()
        if dg0 >= 0:
            d = -op.apply(g) if op is not None else -g * (Dinv if Dinv is not None else 1.0)
            dg0 = (g @ d).item(); S, Y = [], []
        while True:
            xn = x + t * d
            fn, gn = evaluate(xn); nev += 1
            if fn <= fx + 1e-4 * t * dg0 and abs((gn @ d).item()) <= 0.9 * abs(dg0):
                break
            if fn <= fx + 1e-4 * t * dg0 and (gn @ d).item() < 0:   # too short: extend
                t *= 2.0
                if t > 64: break
                continue
            t *= 0.5
            if t < 1e-10:
                print("  line search failed at", k); return k, nev, fx
        s, y = xn - x, gn - g
        x, fx, g = xn, fn, gn
        cond = (g.norm() / max(1.0, x.norm())).item()
        for thr in (1.0, 0.1, 0.01, 0.003, 0.001):
            if cond < thr and thr not in marks:
                marks[thr] = k
        if k % 100 == 0 or cond <= EPS:
            print("  %s it=%4d ev=%4d cond=%.3e fx=%.6f |gh|=%.3e |gJ|=%.3e |h|=%.2f |J|=%.2f (%.0fs)" % (
                variant, k, nev, cond, fx, g[:nh].norm().item(), g[nh:].norm().item(), x[:nh].norm().item(),
                x[nh:].norm().item(), time.time() - t0), flush=True)
        if cond <= EPS:
            break
        if (y @ s).item() > 1e-12 * (y @ y).item():
            S.append(s); Y.append(y)
            if len(S) > m: S.pop(0); Y.pop(0)
        if variant.startswith("diagx") and k % refresh == 0:
            _, _, dd = evaluate(x, True); nev += 1
            Dinv = 1.0 / dd
        q = -g.clone(); al = []
        for s_, y_ in zip(reversed(S), reversed(Y)):
            a = ((s_ @ q) / (y_ @ s_)).item(); al.append(a); q -= a * y_
        if op is not None:
            q = op.apply(q)
            if S:
                q *= ((S[-1] @ Y[-1]) / (Y[-1] @ op.apply(Y[-1]))).item()
        elif Dinv is not None:
            if S:
                gam = ((S[-1] @ Y[-1]) / (Y[-1] @ (Dinv * Y[-1]))).item()
            else:
                gam = 1.0
            q = q * Dinv * gam
        elif S:
            q *= ((S[-1] @ Y[-1]) / (Y[-1] @ Y[-1])).item()
        for (s_, y_), a in zip(zip(S, Y), reversed(al)):
            b = ((y_ @ q) / (y_ @ s_)).item(); q += (a - b) * s_
        d = q
    print("%s N=%d L=%d: iters=%d evals=%d fx=%.6f marks=%s (%.0fs)" % (variant, N, L, k, nev, fx, marks, time.time() - t0), flush=True)
    return x


if __name__ == "__main__" and not os.environ.get("LAB_PCG") and not os.environ.get("LAB_NCG") and not os.environ.get("LAB_SPEC") and not os.environ.get("LAB_VARPRO"):
    for v in variants:
        lbfgs(v)


def hvp_factory(x):
    """Exact Hessian-vector product at x (same GEMM structure as the gradient)."""
    h, W = unpack(x)
    H = (X @ W).reshape(N, L, Q) + h[None]
    P = torch.softmax(H, dim=2)

    def hvp(v):
        dh, dW = unpack(v)
        dH = (X @ dW).reshape(N, L, Q) + dh[None]
        dP = P * (dH - (P * dH).sum(2, keepdim=True))
        R = (w[:, None, None] * dP).reshape(N, D)
        G = X.T @ R
        return torch.cat([R.sum(0) + 2 * lh * v[:nh], (G + G.T)[upper] + 2 * lj * v[nh:]])
    return hvp


def pcg_study(xstar):
    hvp = hvp_factory(xstar)
    torch.manual_seed(1)
    b = torch.randn(nh + nJ, dtype=torch.float64)
    b = hvp(b)          # rhs in the range, solution = random vector (all modes excited)
    ops = {"none": lambda r: r, "diag0": (lambda dinv: (lambda r: r * dinv))(1.0 / diag_indep()),
           "gauge": Gauge("gauge").apply, "kfac4": KFac("shifts:4").apply}
    _, _, dd = evaluate(xstar, True)
    ops["diagx*"] = (lambda dinv: (lambda r: r * dinv))(1.0 / dd)
    for name, M in ops.items():
        xk = torch.zeros_like(b); r = b.clone(); z = M(r); p = z.clone(); rz = (r @ z).item(); r0 = r.norm().item()
        hist = {}
        for k in range(1, 3001):
            Ap = hvp(p); a = rz / (p @ Ap).item()
            xk += a * p; r -= a * Ap
            rel = r.norm().item() / r0
            for thr in (1e-1, 1e-2, 1e-3, 1e-4):
                if rel < thr and thr not in hist: hist[thr] = k
            if rel < 1e-4: break
            z = M(r); rz2 = (r @ z).item(); p = z + (rz2 / rz) * p; rz = rz2
        print("PCG %-7s iterations to residual reduction: %s" % (name, hist), flush=True)


if os.environ.get("LAB_PCG"):
    xs = lbfgs("gauge")
    pcg_study(xs)


def newton_cg(switch=0.3, eta=0.1, precond=None, maxcg=400):
    """L-BFGS (plain) until cond < switch, then truncated Newton: CG on exact Hessian-vector products."""
    global EPS
    eps_final = EPS
    EPS = switch
    t0 = time.time()
    x = lbfgs("plain")
    EPS = eps_final
    fx, g = evaluate(x)
    work = 0
    Minv = (1.0 / diag_indep()) if precond == "diag0" else None
    for it in range(1, 50):
        cond = (g.norm() / max(1.0, x.norm())).item()
        print("  newton step %d: cond=%.3e fx=%.6f work(evals+hvps)=%d" % (it, cond, fx, work), flush=True)
        if cond <= EPS:
            break
        hvp = hvp_factory(x); work += 1         # base evaluation stores P
        b = -g
        d = torch.zeros_like(b); r = b.clone()
        z = r * Minv if Minv is not None else r
        p = z.clone(); rz = (r @ z).item(); r0 = r.norm().item()
        tol = min(eta, cond ** 0.5) if eta < 0 else eta
        for k in range(1, maxcg + 1):
            Ap = hvp(p); work += 1
            a = rz / (p @ Ap).item()
            d += a * p; r -= a * Ap
            if r.norm().item() <= tol * r0:
                break
            z = r * Minv if Minv is not None else r
            rz2 = (r @ z).item(); p = z + (rz2 / rz) * p; rz = rz2
        t = 1.0
        while True:
            fn, gn = evaluate(x + t * d); work += 1
            if fn <= fx + 1e-4 * t * (g @ d).item(): break
            t *= 0.5
        print("    cg iterations %d, step %.3f" % (k, t), flush=True)
        x, fx, g = x + t * d, fn, gn
    print("newton-cg(switch=%g, eta=%g, precond=%s) N=%d L=%d: extra work after L-BFGS phase = %d (%.0fs total)" % (
        switch, eta, precond, N, L, work, time.time() - t0), flush=True)


if os.environ.get("LAB_NCG"):
    sw, eta = [float(v) for v in os.environ["LAB_NCG"].split(",")[:2]]
    pc = os.environ["LAB_NCG"].split(",")[2] if os.environ["LAB_NCG"].count(",") >= 2 else None
    newton_cg(sw, eta, pc)


def spectrum_study(xstar):
    import scipy.sparse.linalg as sla
    hvp = hvp_factory(xstar)
    n = nh + nJ
    op = sla.LinearOperator((n, n), matvec=lambda v: hvp(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64))).numpy(), dtype=np.float64)
    gz = Gauge("gauge")
    vals, vecs = sla.eigsh(op, k=12, which="LA", tol=1e-3)
    print("top eigenvalues:", np.round(vals[::-1], 1))
    for k in range(11, 5, -1):
        v = torch.from_numpy(vecs[:, k])
        vh = v[:nh]; vJ = v[nh:][gz.perm]
        gam = vJ.mean(dim=(1, 2), keepdim=True); r = vJ.mean(2, keepdim=True) - gam; c = vJ.mean(1, keepdim=True) - gam
        zs = vJ - r - c - gam
        print("  lambda=%.1f: |h|^2=%.3f |rowcol|^2=%.3f |gamma|^2=%.3f |zerosum|^2=%.3f" % (
            vals[k], (vh ** 2).sum(), Q * ((r ** 2).sum() + (c ** 2).sum()), Q * Q * (gam ** 2).sum(), (zs ** 2).sum()))
    vals_s, _ = sla.eigsh(op, k=6, which="SA", tol=1e-2, maxiter=3000)
    print("bottom eigenvalues:", np.round(vals_s, 3))


if os.environ.get("LAB_SPEC"):
    xs = lbfgs("gauge")
    spectrum_study(xs)


def varpro(m=6):
    """Variable projection: fields always optimal for the current couplings (separable per-site Newton on h),
    L-BFGS over the couplings only."""
    Xi = X.reshape(N, L, Q)
    state = {"h": start()[:nh].reshape(L, Q).clone(), "newton": 0}

    def reduced(xJ):
        W = torch.zeros((D, D), dtype=torch.float64); W[upper] = xJ; W = W + W.T
        HJ = (X @ W).reshape(N, L, Q)
        h = state["h"]
        for it in range(50):
            H = HJ + h[None]
            lse = torch.logsumexp(H, dim=2)
            P = torch.exp(H - lse[:, :, None])
            gh = (w[:, None, None] * (P - Xi)).sum(0) + 2 * lh * h                       # (L,Q)
            if gh.norm().item() < 1e-9 * max(1.0, h.norm().item()):
                break
            wP = w[:, None, None] * P
            Hh = torch.diag_embed(wP.sum(0)) - torch.einsum("sia,sib->iab", wP, P) + 2 * lh * torch.eye(Q, dtype=torch.float64)[None]
            h = h - torch.linalg.solve(Hh, gh[:, :, None])[:, :, 0]
            state["newton"] += 1
        state["h"] = h
        Hx = (H * Xi).sum(2)
        nll = -(w[:, None] * (Hx - lse)).sum()
        R = (w[:, None, None] * P).reshape(N, D) - w[:, None] * X
        G = X.T @ R
        gJ = (G + G.T)[upper] + 2 * lj * xJ
        fx = nll + lh * (h ** 2).sum() + lj * (xJ ** 2).sum()
        return fx.item(), gJ

    x = torch.zeros(nJ, dtype=torch.float64)
    fx, g = reduced(x); nev = 1
    S, Y = [], []
    d = -g / g.norm()
    t0 = time.time(); marks = {}
    for k in range(1, MAXIT + 1):
        t, dg0 = 1.0, (g @ d).item()
        hsave = state["h"].clone()
        while True:
            fn, gn = reduced(x + t * d); nev += 1
            if fn <= fx + 1e-4 * t * dg0 and abs((gn @ d).item()) <= 0.9 * abs(dg0): break
            if fn <= fx + 1e-4 * t * dg0 and (gn @ d).item() < 0:
                t *= 2.0
                if t > 64: break
                continue
            t *= 0.5
            state["h"] = hsave.clone()
            if t < 1e-10: print("ls fail"); return
        s, y = t * d, gn - g
        x, fx, g = x + t * d, fn, gn
        xn = (x.norm() ** 2 + state["h"].norm() ** 2).sqrt().item()
        cond = g.norm().item() / max(1.0, xn)
        for thr in (1.0, 0.1, 0.01, 0.003, 0.001):
            if cond < thr and thr not in marks: marks[thr] = k
        if k % 50 == 0 or cond <= EPS:
            print("  varpro it=%4d ev=%4d newton=%d cond=%.3e fx=%.6f |h|=%.2f |J|=%.2f (%.0fs)" % (
                k, nev, state["newton"], cond, fx, state["h"].norm().item(), x.norm().item(), time.time() - t0), flush=True)
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
    print("varpro N=%d L=%d: iters=%d evals=%d newton=%d fx=%.6f marks=%s (%.0fs)" % (N, L, k, nev, state["newton"], fx, marks, time.time() - t0))


if os.environ.get("LAB_VARPRO"):
    varpro()
