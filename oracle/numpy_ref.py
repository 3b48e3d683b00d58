"""
Independent numpy (float64) restatement of the PLM objective/gradient and scoring, used to
cross-check the C oracle at small sizes.  TEST INFRASTRUCTURE ONLY.

Follows SURVEY.md App. C.3 (objective; PARITY UNPINNED -- plmc unavailable) and
evcouplings/couplings/model.py:179-233, 744-827 (scores).
"""
import numpy as np


def unpack(x, L, q):
    """[h | J_ij (i<j)] -> h (L,q), dense symmetric J (L,L,q,q) with zero diagonal."""
    h = x[:L * q].reshape(L, q)
    blocks = x[L * q:].reshape(L * (L - 1) // 2, q, q)
    J = np.zeros((L, L, q, q))
    iu, ju = np.triu_indices(L, 1)
    J[iu, ju] = blocks
    J[ju, iu] = blocks.transpose(0, 2, 1)
    return h, J


def pack_grad(gh, gJ, L):
    iu, ju = np.triu_indices(L, 1)
    return np.concatenate([gh.ravel(), gJ[iu, ju].ravel()])


def plm_eval(msa, w, q, lambda_h, lambda_j, x):
    """-> (fx, nll, g) via dense one-hot algebra (einsum), float64."""
    msa = np.asarray(msa)
    N, L = msa.shape
    w = np.asarray(w, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    h, J = unpack(x, L, q)
    X = np.zeros((N, L, q))
    X[np.arange(N)[:, None], np.arange(L)[None, :], msa] = 1.0
    H = h[None] + np.einsum("sjb,ijab->sia", X, J)
    H -= H.max(axis=2, keepdims=True)
    logZ = np.log(np.exp(H).sum(axis=2, keepdims=True))
    logP = H - logZ
    nll = -(w[:, None] * (logP * X).sum(axis=2)).sum()
    R = w[:, None, None] * (np.exp(logP) - X)          # residuals (N, L, q)
    gh = R.sum(axis=0) + 2 * lambda_h * h
    G = np.einsum("sia,sjb->ijab", R, X)                # asymmetric slab
    gJ = G + G.transpose(1, 0, 3, 2) + 2 * lambda_j * J
    iu, ju = np.triu_indices(L, 1)
    fx = nll + lambda_h * (h ** 2).sum() + lambda_j * (J[iu, ju] ** 2).sum()
    return fx, nll, pack_grad(gh, gJ, L)


def brute_force_conditionals(msa, q, x):
    """log P(x_si | rest) by explicit enumeration of site energies; tiny cases only."""
    msa = np.asarray(msa)
    N, L = msa.shape
    h, J = unpack(np.asarray(x, dtype=np.float64), L, q)
    out = np.zeros((N, L))
    for s in range(N):
        for i in range(L):
            e = np.array([h[i, a] + sum(J[i, j, a, msa[s, j]] for j in range(L) if j != i)
                          for a in range(q)])
            out[s, i] = e[msa[s, i]] - np.log(np.exp(e).sum())
    return out


def scores(jij, L, q):
    """zero-sum gauge -> Frobenius -> APC (model.py:208-231, 792, 764-775)."""
    blocks = np.asarray(jij, dtype=np.float64).reshape(-1, q, q)
    z = (blocks - blocks.mean(axis=2, keepdims=True) - blocks.mean(axis=1, keepdims=True)
         + blocks.mean(axis=(1, 2), keepdims=True))
    fn = np.zeros((L, L))
    iu, ju = np.triu_indices(L, 1)
    fn[iu, ju] = np.sqrt((z ** 2).sum(axis=(1, 2)))
    fn = fn + fn.T
    col = fn.mean(axis=0) * L / (L - 1)
    mean = fn.mean() * L / (L - 1)
    cn = fn - np.outer(col, col) / mean
    cn[np.diag_indices(L)] = 0
    return fn, cn
