"""
Numpy (float64) restatement of mean-field direct coupling analysis.  TEST INFRASTRUCTURE ONLY: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows evcouplings/couplings/mean_field.py:
  regularize_frequencies       :717-743     rf_i  = (1-pc) f_i + pc/q
  regularize_pair_frequencies  :746-789     rf_ij = (1-pc) f_ij + pc/q^2 (i != j);  rf_ii = (1-pc) diag(f_i) + pc/q I
  compute_covariance_matrix    :897-940     C[(i,a),(j,b)] = rf_ij(a,b) - rf_i(a) rf_j(b),  a, b < q-1
  MeanFieldDCA.fit             :204-210     J = -inv(C)
  reshape_invC_to_4d           :943-975     dense L x L x q x q, last row/column of each block 0, diagonal blocks kept
  fields                       :977-1014    h_i = log(rf_i / rf_i[q-1]) - sum_{j != i} J_ij rf_j
  tilde_fields, direct_information :792-893
Pinned against the reference's own functions by tests/golden/meanfield_*.npz (tests/golden/make_golden.py).
Inputs are the weighted frequencies (f_i [L,q], f_ij as i<j blocks [pairs,q,q]).
"""
import numpy as np


def dense_pair_frequencies(fi, fij_pairs):
    """i<j blocks -> dense symmetric L x L x q x q with f_ii = diag(f_i) (alignment.py:1110-1153 layout)."""
    L, q = fi.shape
    f = np.zeros((L, L, q, q))
    iu, ju = np.triu_indices(L, 1)
    f[iu, ju] = fij_pairs
    f[ju, iu] = np.transpose(fij_pairs, (0, 2, 1))
    for i in range(L):
        f[i, i] = np.diag(fi[i])
    return f


def regularize(fi, fij_dense, pc):
    L, q = fi.shape
    rfi = (1.0 - pc) * fi + pc / q
    rfij = (1.0 - pc) * fij_dense + pc / (q * q)
    for i in range(L):
        rfij[i, i] = (1.0 - pc) * fij_dense[i, i] + (pc / q) * np.eye(q)
    return rfi, rfij


def mean_field(fi, fij_pairs, pseudo_count=0.5, want_di=True):
    """-> dict(rfi, hi [L,q], jij_full [L,L,q,q], jij [pairs,q,q], di [L,L])."""
    fi = np.asarray(fi, dtype=np.float64)
    L, q = fi.shape
    rfi, rfij = regularize(fi, dense_pair_frequencies(fi, np.asarray(fij_pairs, dtype=np.float64)), pseudo_count)
    n = L * (q - 1)
    C = (rfij[:, :, :q - 1, :q - 1] - rfi[:, None, :q - 1, None] * rfi[None, :, None, :q - 1])
    C = C.transpose(0, 2, 1, 3).reshape(n, n)
    inv = -np.linalg.inv(C)
    J = np.zeros((L, L, q, q))
    J[:, :, :q - 1, :q - 1] = inv.reshape(L, q - 1, L, q - 1).transpose(0, 2, 1, 3)
    hi = np.zeros((L, q))
    for i in range(L):
        s = np.zeros(q)
        for j in range(L):
            if j != i:
                s += J[i, j] @ rfi[j]
        hi[i] = np.log(rfi[i] / rfi[i, q - 1]) - s
    iu, ju = np.triu_indices(L, 1)
    out = dict(rfi=rfi, hi=hi, jij_full=J, jij=J[iu, ju])
    if want_di:
        out["di"] = direct_information(J, rfi)
    return out


def direct_information(J, rfi):
    L, q = rfi.shape
    di = np.zeros((L, L))
    for i in range(L - 1):
        for j in range(i + 1, L):
            W = np.exp(J[i, j])
            hti = np.full(q, 1.0 / q)
            htj = np.full(q, 1.0 / q)
            diff = 1.0
            while diff > 1e-4:
                ui = rfi[i] / (W @ htj)
                ui /= ui.sum()
                uj = rfi[j] / (hti @ W)
                uj /= uj.sum()
                diff = max(np.abs(ui - hti).max(), np.abs(uj - htj).max())
                hti, htj = ui, uj
            P = W * np.outer(hti, htj)
            P /= P.sum()
            F = np.outer(rfi[i], rfi[j])
            di[i, j] = di[j, i] = (P * np.log((P + 1e-100) / (F + 1e-100))).sum()
    return di
