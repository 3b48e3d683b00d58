"""
Pins the CPU oracle: against golden vectors produced by the reference's own Python
(tests/golden/make_golden.py) where reference code exists, and against first principles
(brute force, finite differences, convexity) where it does not (PLM objective, L-BFGS:
PARITY UNPINNED, see oracle/plm_oracle.c).
"""
import os

import numpy as np
import pytest

from evcouplings_amd.synthetic import synthetic_msa
from oracle import numpy_ref

Q = 21


def _golden_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "reweight_freqs.npz"))
    names = sorted({k.split("_")[0] for k in z.files})
    return {n: {f: z["%s_%s" % (n, f)] for f in ("msa", "theta", "counts", "fi", "fij")} for n in names}


def test_reweight_matches_reference_kernel(oracle64, golden_dir):
    # evcouplings/align/alignment.py:1193-1233; includes rows at T-1, T, T+1 identities
    for name, c in _golden_cases(golden_dir).items():
        counts = oracle64.reweight(c["msa"], float(c["theta"]))
        np.testing.assert_array_equal(counts, c["counts"], err_msg=name)


def test_threshold_rule_equals_in_repo_rule(oracle64):
    # integer rule == `pair_id / L >= theta` for all L, theta tested
    for theta in (0.8, 0.6, 0.3, 0.9, 0.5):
        for L in range(1, 700):
            ids = np.arange(L + 1)
            ref_min = ids[(ids / (1.0 * L)) >= theta].min() if (ids / (1.0 * L) >= theta).any() else L + 1
            assert oracle64.threshold(L, theta) == ref_min, (theta, L)


def test_frequencies_match_reference_kernels(oracle64, golden_dir):
    # evcouplings/align/alignment.py:1079-1153
    for name, c in _golden_cases(golden_dir).items():
        msa = c["msa"]
        L = msa.shape[1]
        w = 1.0 / c["counts"]
        fi, fij = oracle64.marginals(msa, w, Q)
        np.testing.assert_allclose(fi, c["fi"], rtol=0, atol=1e-14, err_msg=name)
        iu, ju = np.triu_indices(L, 1)
        np.testing.assert_allclose(fij, c["fij"][iu, ju], rtol=0, atol=1e-14, err_msg=name)


def test_scores_match_reference_couplingsmodel(oracle64, golden_dir):
    # evcouplings/couplings/model.py:179-233, 744-827 via the real CouplingsModel
    z = np.load(os.path.join(golden_dir, "scores_L12.npz"))
    L = z["hi"].shape[0]
    fn, cn = oracle64.scores(z["jij"].astype(np.float64), L, Q)
    np.testing.assert_allclose(fn, z["fn"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cn, z["cn"], rtol=1e-11, atol=1e-12)
    fn2, cn2 = numpy_ref.scores(z["jij"], L, Q)
    np.testing.assert_allclose(cn2, z["cn"], rtol=1e-11, atol=1e-12)


def test_eval_brute_force_tiny(oracle64):
    rng = np.random.default_rng(0)
    N, L, q = 7, 3, 3
    msa = rng.integers(0, q, size=(N, L)).astype(np.int8)
    w = rng.random(N) + 0.1
    x = rng.normal(size=L * q + L * (L - 1) // 2 * q * q)
    fx, nll, g = oracle64.eval(msa, w, q, 0.3, 0.7, x)
    logp = numpy_ref.brute_force_conditionals(msa, q, x)
    assert nll == pytest.approx(-(w[:, None] * logp).sum(), rel=1e-12)
    fx2, nll2, g2 = numpy_ref.plm_eval(msa, w, q, 0.3, 0.7, x)
    assert fx == pytest.approx(fx2, rel=1e-12)
    np.testing.assert_allclose(g, g2, rtol=1e-10, atol=1e-12)


def test_eval_matches_numpy_and_finite_differences(oracle64):
    rng = np.random.default_rng(1)
    N, L = 64, 8
    msa, _ = synthetic_msa(N, L, seed=11)
    w = 1.0 / oracle64.reweight(msa, 0.8)
    n = L * Q + L * (L - 1) // 2 * Q * Q
    x = 0.1 * rng.normal(size=n)
    lh, lj = 0.01, 0.01 * 20 * (L - 1)
    fx, nll, g = oracle64.eval(msa, w, Q, lh, lj, x)
    fx2, nll2, g2 = numpy_ref.plm_eval(msa, w, Q, lh, lj, x)
    assert fx == pytest.approx(fx2, rel=1e-12)
    np.testing.assert_allclose(g, g2, rtol=1e-9, atol=1e-11)
    # central finite differences along random directions and a few coordinates
    for _ in range(4):
        d = rng.normal(size=n)
        d /= np.linalg.norm(d)
        eps = 1e-5
        fp = oracle64.eval(msa, w, Q, lh, lj, x + eps * d)[0]
        fm = oracle64.eval(msa, w, Q, lh, lj, x - eps * d)[0]
        assert (fp - fm) / (2 * eps) == pytest.approx(g @ d, rel=1e-6, abs=1e-7)
    for k in rng.integers(0, n, size=6):
        e = np.zeros(n)
        e[k] = 1e-5
        fd = (oracle64.eval(msa, w, Q, lh, lj, x + e)[0] - oracle64.eval(msa, w, Q, lh, lj, x - e)[0]) / 2e-5
        assert fd == pytest.approx(g[k], rel=1e-5, abs=1e-6)


def test_f32_build_agrees_with_f64(oracle64, oracle32):
    msa, _ = synthetic_msa(200, 12, seed=5)
    w = 1.0 / oracle64.reweight(msa, 0.8)
    np.testing.assert_array_equal(oracle32.reweight(msa, 0.8), oracle64.reweight(msa, 0.8))
    n = 12 * Q + 66 * Q * Q
    x = 0.05 * np.random.default_rng(2).normal(size=n)
    fx64, _, g64 = oracle64.eval(msa, w, Q, 0.01, 2.2, x)
    fx32, _, g32 = oracle32.eval(msa, w, Q, 0.01, 2.2, x)
    assert fx32 == pytest.approx(fx64, rel=1e-5)
    np.testing.assert_allclose(g32, g64, rtol=2e-3, atol=2e-4)


def test_fit_converges_to_the_unique_optimum(oracle64):
    # strictly convex objective: our L-BFGS and scipy's L-BFGS-B (different code, different
    # start) must land on the same optimum (SURVEY.md section 8c golden vector iv)
    from scipy.optimize import minimize
    N, L = 300, 10
    msa, _ = synthetic_msa(N, L, seed=21)
    lj = 0.01 * 20 * (L - 1)
    res = oracle64.fit(msa, Q, lambda_h=0.01, lambda_j=lj, max_iter=2000, epsilon=1e-9)
    assert res["status"] in (0, 2)
    w = res["weights"]

    def fun(x):
        fx, _, g = oracle64.eval(msa, w, Q, 0.01, lj, x)
        return fx, g

    sp = minimize(fun, np.zeros_like(res["x"]), jac=True, method="L-BFGS-B",
                  options=dict(maxiter=5000, ftol=1e-15, gtol=1e-9, maxcor=20))
    assert abs(sp.fun - res["fx"]) <= 1e-8 * abs(sp.fun)
    np.testing.assert_allclose(res["jij"].ravel(), sp.x[L * Q:], atol=2e-6)
    fn_sp, cn_sp = numpy_ref.scores(sp.x[L * Q:], L, Q)
    np.testing.assert_allclose(res["cn"], cn_sp, atol=1e-5)
    # iteration table is monotone in fx (line search enforces sufficient decrease)
    fxs = [r[3] for r in res["table"]]
    assert all(b <= a + 1e-12 * abs(a) for a, b in zip(fxs, fxs[1:]))
    # planted structure shows up: weights, N_eff sane
    assert 1.0 <= res["n_eff"] <= N


def test_fit_respects_max_iter_and_reports_status(oracle64):
    msa, _ = synthetic_msa(120, 8, seed=3)
    res = oracle64.fit(msa, Q, max_iter=5, epsilon=1e-12)
    assert res["iters"] == 5 and res["status"] == 1
    assert len(res["table"]) == 5


# ------------------------------------------------------------------ gap-ignoring mode (plmc -g)
def _numpy_eval_gaps(msa, w, q, lh, lj, x):
    """independent float64 restatement of the gap-ignoring objective (masking formulation)."""
    N, L = msa.shape
    qn = q - 1
    h, J = numpy_ref.unpack(np.asarray(x, np.float64), L, qn)
    X = np.zeros((N, L, qn))
    nz = msa > 0
    s_idx, i_idx = np.nonzero(nz)
    X[s_idx, i_idx, msa[s_idx, i_idx] - 1] = 1.0           # all-zero rows for gaps
    H = h[None] + np.einsum("sjb,ijab->sia", X, J)
    H -= H.max(axis=2, keepdims=True)
    logP = H - np.log(np.exp(H).sum(axis=2, keepdims=True))
    M = nz[:, :, None] * np.asarray(w, np.float64)[:, None, None]     # weight only where site is not a gap
    nll = -(M[:, :, 0] * (logP * X).sum(axis=2)).sum()
    R = M * (np.exp(logP) - X)
    gh = R.sum(axis=0) + 2 * lh * h
    G = np.einsum("sia,sjb->ijab", R, X)
    gJ = G + G.transpose(1, 0, 3, 2) + 2 * lj * J
    iu, ju = np.triu_indices(L, 1)
    fx = nll + lh * (h ** 2).sum() + lj * (J[iu, ju] ** 2).sum()
    return fx, nll, numpy_ref.pack_grad(gh, gJ, L)


def test_gap_mode_eval_matches_numpy_and_reduces_to_plain_model(oracle64):
    rng = np.random.default_rng(4)
    N, L, q = 80, 7, 6
    msa = rng.integers(0, q, size=(N, L)).astype(np.int8)
    w = rng.random(N) + 0.2
    qn = q - 1
    x = 0.3 * rng.normal(size=L * qn + L * (L - 1) // 2 * qn * qn)
    fx, nll, g = oracle64.eval_gaps(msa, w, q, 0.05, 0.4, x)
    fx2, nll2, g2 = _numpy_eval_gaps(msa, w, q, 0.05, 0.4, x)
    assert fx == pytest.approx(fx2, rel=1e-12) and nll == pytest.approx(nll2, rel=1e-12)
    np.testing.assert_allclose(g, g2, rtol=1e-9, atol=1e-11)
    # without any gap the mode is the ordinary (q-1)-state model on the shifted alphabet
    msa_ng = rng.integers(1, q, size=(N, L)).astype(np.int8)
    a = oracle64.eval_gaps(msa_ng, w, q, 0.05, 0.4, x)
    b = oracle64.eval((msa_ng - 1).astype(np.int8), w, qn, 0.05, 0.4, x)
    assert a[0] == pytest.approx(b[0], rel=1e-13)
    np.testing.assert_allclose(a[2], b[2], rtol=1e-12, atol=1e-13)
    np.testing.assert_array_equal(oracle64.reweight_gaps(msa_ng, 0.5), oracle64.reweight((msa_ng - 1).astype(np.int8), 0.5))


def test_gap_mode_marginals_and_reweighting_rules(oracle64):
    msa = np.array([[1, 2, 0, 3], [1, 2, 0, 3], [0, 2, 0, 1], [1, 0, 0, 0], [0, 0, 0, 0]], dtype=np.int8)
    counts = oracle64.reweight_gaps(msa, 0.5)       # T = 2 identical non-gap positions
    assert counts.tolist() == [2, 2, 1, 1, 1]       # all-gap / sparse rows still count themselves
    w = np.array([1.0, 2.0, 1.0, 4.0, 1.0])
    fi, fij = oracle64.marginals_gaps(msa, w, 4)
    assert fi[0].tolist() == [1.0, 0.0, 0.0]                       # site 0: only state 1 among ungapped
    assert fi[2].tolist() == [0.0, 0.0, 0.0]                       # site 2: everybody gapped
    np.testing.assert_allclose(fi[3], [1 / 4, 0, 3 / 4])
    np.testing.assert_allclose(fij[0].sum(), 1.0)                  # pair (0,1): sequences 0 and 1 only
    assert fij[0][0, 1] == pytest.approx(1.0)


def test_gap_mode_fit_converges(oracle64):
    from scipy.optimize import minimize
    msa, _ = synthetic_msa(250, 9, seed=6)
    res = oracle64.fit(msa, Q, max_iter=2000, epsilon=1e-9, ignore_gaps=True)
    assert res["hi"].shape == (9, 20) and res["jij"].shape == (36, 20, 20) and res["status"] in (0, 2)
    w = res["weights"]
    lj = res["lambda_j"]
    assert lj == pytest.approx(0.01 * 19 * 8)
    sp = minimize(lambda x: oracle64.eval_gaps(msa, w, Q, 0.01, lj, x)[::2], np.zeros_like(res["x"]), jac=True,
                  method="L-BFGS-B", options=dict(maxiter=5000, ftol=1e-15, gtol=1e-9, maxcor=20))
    assert abs(sp.fun - res["fx"]) <= 1e-8 * abs(sp.fun)
    np.testing.assert_allclose(res["jij"].ravel(), sp.x[9 * 20:], atol=5e-6)


def test_field_objective_scaling_matches_reference_independent_model(oracle64, golden_dir):
    """App. D-2 pinned by reference code: to_independent_model (model.py:894-910) minimises
    N_eff (logZ - f.x) + lambda_h |x|^2 per site; its optimum must be stationary for our objective at J = 0."""
    z = np.load(os.path.join(golden_dir, "independent_model_a.npz"))
    c = _golden_cases(golden_dir)["a"]
    msa, w = c["msa"], 1.0 / c["counts"]
    N, L = msa.shape
    h_ref = z["h_ref"]
    x = np.concatenate([h_ref.ravel(), np.zeros(L * (L - 1) // 2 * Q * Q)])
    fx, nll, g = oracle64.eval(msa, w, Q, float(z["lambda_h"]), 7.0, x)
    gh = g[:L * Q]
    assert np.abs(gh).max() < 1e-4, np.abs(gh).max()          # fmin_bfgs stopped at gtol 1e-5 in f32-rounded inputs
    assert np.abs(gh).max() < 1e-3 * np.abs(2 * 0.01 * h_ref).max() + 1e-4
    # and the value is the reference's formula summed over sites (model.py:899-900)
    neff = w.sum()
    logZ = np.log(np.exp(h_ref).sum(axis=1))
    ref_val = (neff * (logZ - (c["fi"] * h_ref).sum(axis=1)) + 0.01 * (h_ref ** 2).sum(axis=1)).sum()
    assert fx == pytest.approx(ref_val, rel=1e-12)
    # moving away from it along any field direction increases the objective
    d = np.random.default_rng(0).normal(size=L * Q)
    xp = x.copy(); xp[:L * Q] += 1e-2 * d
    assert oracle64.eval(msa, w, Q, 0.01, 7.0, xp)[0] > fx


# ---------------------------------------------------------------- statistical energies (row N2)
def test_energies_match_reference_golden(oracle64, golden_dir):
    """hamiltonians / single-mutant matrix == the reference's own loops (couplings/model.py:25-109)."""
    z = np.load(os.path.join(golden_dir, "energies_L12.npz"))
    g = np.load(os.path.join(golden_dir, "scores_L12.npz"))
    x = np.concatenate([g["hi"].ravel(), g["jij"].ravel()]).astype(np.float64)
    H = oracle64.hamiltonians(z["seqs"], 21, x)
    np.testing.assert_allclose(H, z["hamiltonians"], rtol=0, atol=1e-12)
    S = oracle64.single_mutants(z["seqs"][0], 21, x)
    np.testing.assert_allclose(S, z["single_mutants"], rtol=0, atol=1e-12)
    # internal consistency: a single substitution changes the energy by the matrix entry
    seq = z["seqs"][0].copy()
    base = oracle64.hamiltonians(seq[None], 21, x)[0]
    for (i, a) in ((0, 3), (5, 20), (11, 0)):
        mut = seq.copy()
        mut[i] = a
        np.testing.assert_allclose(oracle64.hamiltonians(mut[None], 21, x)[0] - base, S[i, a], atol=1e-12)


# ---------------------------------------------------------------- mean-field DCA (row N4)
@pytest.mark.parametrize("case", ["a", "d"])
def test_meanfield_restatement_matches_reference_golden(golden_dir, case):
    """oracle/meanfield_ref.py == the reference's own mean_field.py functions (regularisation, covariance,
    inverse, fields, direct information) on the frequencies of golden alignments."""
    from oracle import meanfield_ref
    z = np.load(os.path.join(golden_dir, "meanfield_%s.npz" % case))
    out = meanfield_ref.mean_field(z["fi"], z["fij_pairs"], float(z["pseudo_count"]))
    np.testing.assert_allclose(out["rfi"], z["rfi"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(out["jij_full"], z["jij_full"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["hi"], z["hi"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["di"], z["di"], rtol=0, atol=1e-12)


def test_convention_switches_of_the_oracle_against_brute_force(oracle64):
    """The PLM_CONV_* rules (include/plm_hip.h), restated in plain Python on a small gapped alignment."""
    import math
    rng = np.random.default_rng(12)
    N, L, q, theta = 40, 25, 6, 0.6
    msa = rng.integers(0, q, size=(N, L)).astype(np.int8)
    msa[5:15] = msa[0]                                   # a cluster around sequence 0 ...
    for s in range(5, 15):
        msa[s, rng.integers(0, L, size=s - 2)] = 0      # ... with more and more gaps
    w = rng.uniform(0.2, 1.0, N)

    def counts(rule):
        out = np.zeros(N, np.int32)
        T = math.ceil(theta * L - 1e-9)
        if rule & 32:
            T = math.ceil(float((np.float32(1) - np.float32(1.0 - theta)) * np.float32(L)))   # float32 throughout
        for s in range(N):
            for t in range(N):
                a, b = msa[s], msa[t]
                if rule & 128:
                    both = (a != 0) & (b != 0)
                    hit = ((a == b) & both).sum() >= math.ceil(theta * both.sum() - 1e-9)
                elif rule & 64:
                    hit = (a == b).sum() >= T
                else:
                    hit = ((a == b) & (a != 0)).sum() >= T
                out[s] += 1 if (hit or s == t) else 0
        return out

    try:
        for rule in (0, 32, 64, 128, 32 | 64):
            oracle64.set_conventions(rule)
            np.testing.assert_array_equal(oracle64.reweight_gaps(msa, theta), counts(rule), err_msg="rule %d" % rule)
        for rule in (0, 256):
            oracle64.set_conventions(rule)
            fi, fij = oracle64.marginals_gaps(msa, w, q)
            for i in (0, 7, L - 1):
                live = msa[:, i] != 0
                tot = w.sum() if rule else w[live].sum()
                ref = np.array([w[msa[:, i] == a].sum() / tot for a in range(1, q)])
                np.testing.assert_allclose(fi[i], ref, rtol=1e-12)
            i, j = 3, 11
            live = (msa[:, i] != 0) & (msa[:, j] != 0)
            tot = w.sum() if rule else w[live].sum()
            blk = fij[i * (2 * L - i - 1) // 2 + (j - i - 1)]
            assert blk[1, 2] == pytest.approx(w[(msa[:, i] == 2) & (msa[:, j] == 3)].sum() / tot, rel=1e-12)
        jij = rng.normal(0, 1, (L * (L - 1) // 2, q, q))
        oracle64.set_conventions(512)
        fn, _ = oracle64.scores(jij, L, q)
        b = jij[0] - jij[0].mean(0, keepdims=True) - jij[0].mean(1, keepdims=True) + jij[0].mean()
        assert fn[0, 1] == pytest.approx(np.sqrt((b[1:, 1:] ** 2).sum()), rel=1e-12)
        oracle64.set_conventions(0)
        fn, _ = oracle64.scores(jij, L, q)
        assert fn[0, 1] == pytest.approx(np.sqrt((b ** 2).sum()), rel=1e-12)
    finally:
        oracle64.set_conventions(0)
