"""
Parity and optimality at BASELINE.json scale.

The oracle (oracle/plm_oracle.c, float64, OpenMP) is the checker: one objective+gradient evaluation of it costs
2-6 s on the GPU box's 16 host cores at these sizes, so every test here spends a handful of them.
  * the HIP evaluation (f16 hi/lo MFMA operands, f32 accumulation over up to 50 000 sequences) against the f64
    oracle at a far-from-optimal point and at the converged point, config 2 (L=200, N=20 000) and headline
    (L=300, N=50 000);
  * an optimality certificate for the fit: the ORACLE's gradient at the point the GPU fit stopped satisfies the
    stop rule (the fit does not merely believe it converged), and pushing the GPU fit 10x further does not move CN;
  * config 2 "EC scores within 1e-4 of the CPU solver": the GPU's CN at the SHIPPED stop rule (epsilon = 1e-3) against
    the CN of the point an independent float64 optimiser (scipy L-BFGS-B on the oracle's objective) reaches from there;
  * BASELINE.json configs 3 (N = 100 000), 4 (L = 500) and 5 (L = 600, two chains): one f64-oracle evaluation far from
    the optimum and one at the GPU's stop point each (same thresholds), the optimality certificate for config 3, and for
    config 5 the whole file round trip (A2M -> run_plmc_hip -> .model / _ECs.txt -> reader -> scores).
"""
import os

import numpy as np
import pytest

from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED

pytestmark = pytest.mark.gpu
Q = 21
CONFIGS = {"config2": (20000, 200, BASE_SEED + 2), "headline": (50000, 300, BASE_SEED + 1),
           "config3": (100000, 300, BASE_SEED + 3), "config4": (50000, 500, BASE_SEED + 4),
           "config5": (30000, 600, BASE_SEED + 5)}
# configurations that get the full treatment (two fits); the larger ones get the shipped fit only -- their vectors are
# 220-320 MB each and an oracle evaluation costs 10-20 s of the box's host cores
FULL = ("config2", "headline", "config3")
# config 3 (N = 100 000): the error of the f32-class evaluation reaches the size of the stop rule there (see
# grad_error_bound), a "much tighter" fit cannot be asked for: 0.8 of the stop rule
TIGHT_OF = {"config3": 8e-4}


def grad_error_bound(N, L):
    """|g_hip - g_f64| / |x| allowed at a point the fit stopped at.  The backward GEMM is exact (integer sums of 24-bit
    residuals); what is left is the f32 accumulation of the forward GEMM (~400 accumulation steps per potential, each
    rounding a running sum that carries the large reference-state part) and the f32 softmax: an error per (sequence,
    site, state) of ~1e-6 relative that the gradient sums do not average out completely.  Measured on the five BASELINE
    configurations (round 3): 2.7 - 4.4e-11 N L (config 2 1.3e-4, headline 4.0 - 4.7e-4, config 3 0.9 - 1.3e-3, config 4
    8.1e-4, config 5 5.7 - 6.3e-4) -- tests/probes/operand_grid_probe.py shows it collapse to 1.0e-4 at the headline when
    the potentials happen to be exactly representable.  Bound: 5e-11 N L, never below 2.5e-4."""
    return max(2.5e-4, 5e-11 * N * L)
# the "much tighter than the stop rule" fit: |g|/|x| < 4e-4.  At N = 50 000 the rounding noise of the f32-class
# gradient is ~1e-4 in these units and the fit crawls below 3e-4: about as far as the
# headline can be pushed.
TIGHT = 4e-4


@pytest.fixture(scope="module")
def plm():
    from evcouplings_amd import plm as _plm
    assert _plm.device_count() >= 1, "no gfx950 device: the HIP path has no fallback"
    return _plm


@pytest.fixture(scope="module", autouse=True)
def _oracle_threads(oracle64):
    """These evaluations are big enough to use every core the box grants (cgroup quota aware)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    old = oracle64.num_threads()
    oracle64.set_num_threads(max(1, min(n, 32)))
    yield
    oracle64.set_num_threads(old)


@pytest.fixture(scope="module")
def fits(plm):
    """Per configuration: alignment, weights, a far-from-optimal point (20 iterations) and the converged fits."""
    cache = {}

    def get(name):
        if name in cache:
            return cache[name]
        N, L, seed = CONFIGS[name]
        msa, _ = synthetic_msa(N, L, seed=seed)
        out = {"msa": msa, "N": N, "L": L, "lambda_j": plm.default_lambda_j(L, Q)}
        with plm.PlmContext(msa, Q, max_iter=20, epsilon=1e-3) as ctx:
            out["w"], _, out["n_eff"] = ctx.reweight()
            ctx.marginals(pairs=False)
            ctx.set_x(None)
            ctx.optimize()
            out["x_far"] = ctx.get_x()
            ctx.set_options(max_iter=3000, epsilon=1e-3)
            r = ctx.optimize()
            out["fit_1e-3"] = dict(r, x=ctx.get_x(), cn=ctx.scores()[1])
            if name in FULL:
                ctx.set_options(max_iter=1000, epsilon=TIGHT_OF.get(name, TIGHT))
                r = ctx.optimize()
                out["fit_tight"] = dict(r, x=ctx.get_x(), cn=ctx.scores()[1])
        if name in FULL:          # configs 4 / 5 are visited by one test each: not kept (their vectors are 220-320 MB)
            cache[name] = out
        return out
    return get


def _assert_stopped_properly(name, fit):
    """Every BASELINE configuration meets the stop rule (status 0).  Config 3 (N = 100 000) is the exception that is
    allowed a second outcome: the error of the f32-class evaluation is as large as the stop rule there
    (grad_error_bound: 1.5e-3 against epsilon = 1e-3), the last decade of |g|/|x| is a walk on that noise (318 - 748
    iterations over the runs of round 3) and about one run in ten ends with the line search giving up first -- the
    library reports that honestly as status 2, "converged to precision", with the |g|/|x| it reached, which must then
    be within a few times the stop rule."""
    if name == "config3" and fit["status"] == 2:
        assert fit["table"][-1][2] < 4e-3, fit["status_msg"]
        return
    assert fit["status"] == 0, fit["status_msg"]


def _oracle_eval(oracle64, f, x):
    return oracle64.eval(f["msa"], f["w"].astype(np.float64), Q, 0.01, f["lambda_j"], x.astype(np.float64))


@pytest.mark.parametrize("name", ["config2", "headline", "config3", "config4", "config5"])
def test_evaluation_matches_f64_oracle_at_scale(plm, oracle64, fits, name):
    f = fits(name)
    _assert_stopped_properly(name, f["fit_1e-3"])
    # far from the optimum: relative criteria (gradient entries are large)
    fx, nll, g = plm.evaluate(f["msa"], f["w"], Q, 0.01, f["lambda_j"], f["x_far"])
    fxo, nllo, go = _oracle_eval(oracle64, f, f["x_far"])
    gmax_far = np.abs(go).max()
    assert abs(fx - fxo) <= 2e-6 * abs(fxo) and abs(nll - nllo) <= 2e-6 * abs(nllo)
    assert np.abs(g - go).max() <= 2e-5 * gmax_far
    # at the converged point the gradient itself is tiny: the error must stay well inside the stop rule's scale
    x = f["fit_1e-3"]["x"]
    fx, nll, g = plm.evaluate(f["msa"], f["w"], Q, 0.01, f["lambda_j"], x)
    fxo, nllo, go = _oracle_eval(oracle64, f, x)
    assert abs(fx - fxo) <= 2e-6 * abs(fxo)
    err = np.linalg.norm(g - go) / max(1.0, np.linalg.norm(x))
    cond64 = np.linalg.norm(go) / max(1.0, np.linalg.norm(x))
    print("%s: |g_hip - g_f64|/|x| = %.3g at the stop point, oracle cond %.3g, %d iterations / %d evaluations" % (
        name, err, cond64, f["fit_1e-3"]["iters"], f["fit_1e-3"]["n_evals"]))
    bound = grad_error_bound(f["N"], f["L"])
    assert err <= bound, (err, bound)            # 7.5e-4 at the headline (measured 4.0 - 4.7e-4); eps = 1e-3 is the stop rule
    assert np.abs(g - go).max() <= 2e-5 * gmax_far
    # optimality as the ORACLE sees it (configs 2 / headline / 3 repeat this with more checks below): the stop rule
    # plus the evaluation error
    assert cond64 < max(1e-3, f["fit_1e-3"]["table"][-1][2]) + max(6e-4, bound), cond64
    f["cond64_1e-3"] = cond64


@pytest.mark.parametrize("name", ["config2", "headline", "config3"])
def test_fit_optimality_certificate(oracle64, fits, name):
    f = fits(name)
    a, b = f["fit_1e-3"], f["fit_tight"]
    tight = TIGHT_OF.get(name, TIGHT)
    _assert_stopped_properly(name, a)                                 # converged by its own rule
    if a["status"] != 0:
        pytest.skip("config 3 stopped 'converged to precision' in this run: no tighter fit to compare with")
    assert a["table"][-1][2] < 1e-3
    # the oracle agrees: its float64 gradient at the GPU's final point satisfies the rule up to the evaluation error
    # (computed by the evaluation test above when it ran on this configuration first)
    cond64 = f.get("cond64_1e-3")
    if cond64 is None:
        _, _, go = _oracle_eval(oracle64, f, a["x"])
        cond64 = np.linalg.norm(go) / max(1.0, np.linalg.norm(a["x"]))
    bound = grad_error_bound(f["N"], f["L"])
    assert cond64 < 1e-3 + max(6e-4, bound), cond64
    # 1.25 - 2.5 times tighter moves no EC score by more than 1e-4 (BASELINE.json's tolerance on EC scores)
    assert b["status"] == 0 and b["table"][-1][2] < tight, (b["status_msg"], b["table"][-1][2])
    assert np.abs(a["cn"] - b["cn"]).max() < 1e-4
    _, _, gob = _oracle_eval(oracle64, f, b["x"])
    cond64_tight = np.linalg.norm(gob) / max(1.0, np.linalg.norm(b["x"]))
    print("%s: oracle cond at the eps = 1e-3 point %.3g, at the eps = %.1g point %.3g" % (name, cond64, tight, cond64_tight))
    assert cond64_tight < tight + max(6e-4, bound), cond64_tight


def test_config2_cn_within_1e4_of_an_independent_f64_optimiser(plm, oracle64, fits):
    """BASELINE.json config 2: 'EC scores vs CPU plmc within 1e-4'.  plmc is unobtainable; the CPU
    side here is scipy's L-BFGS-B minimising the ORACLE's float64 objective, started from the answer the drop-in SHIPS
    (stop rule epsilon = 1e-3, not the tighter fit) and given 25 evaluations: whatever it still gains must not move a CN
    score by 1e-4."""
    import scipy.optimize as so
    f = fits("config2")
    shipped = f["fit_1e-3"]
    x0 = shipped["x"].astype(np.float64)
    w64 = f["w"].astype(np.float64)

    def fun(x):
        fx, _, g = oracle64.eval(f["msa"], w64, Q, 0.01, f["lambda_j"], x)
        return fx, g

    res = so.minimize(fun, x0, jac=True, method="L-BFGS-B", options=dict(maxfun=25, maxcor=10, ftol=0, gtol=0))
    assert res.fun <= fun(x0)[0] * (1 + 1e-12)
    L = f["L"]
    _, cn_cpu = oracle64.scores(res.x[L * Q:], L, Q)
    print("config2: scipy f64 from the shipped point: f %.6f -> %.6f, max |dCN| %.3g" % (
        fun(x0)[0], res.fun, np.abs(cn_cpu - shipped["cn"]).max()))
    assert np.abs(cn_cpu - shipped["cn"]).max() < 1e-4
    assert np.abs(cn_cpu - f["fit_tight"]["cn"]).max() < 1e-4


def test_config5_two_chain_file_round_trip(plm, tmp_path):
    """BASELINE.json config 5: EVcomplex-style concatenated alignment, L = 600 = 350 + 250, N = 30 000 -- 'inter-chain EC
    scoring, CouplingsModel round-trip'.  A2M file -> the run_plmc drop-in (fit to the stop rule) -> _ECs.txt and the
    634 MB plmc_v2 .model -> read back -> scores recomputed from the file's couplings must be the EC file's, and the
    planted inter-chain couplings must lead the inter-chain ranking (what the reference's complex protocol extracts,
    couplings/protocol.py:521-560)."""
    from evcouplings_amd import model_io, tools
    from evcouplings_amd.synthetic import msa_to_a2m
    N, L, seed = CONFIGS["config5"]
    L1 = 350
    msa, planted = synthetic_msa(N, L, seed=seed)
    ali = msa_to_a2m(msa, str(tmp_path / "c5.a2m"))
    ec_file, model_file = str(tmp_path / "c5_ECs.txt"), str(tmp_path / "c5.model")
    r = tools.run_plmc_hip(ali, ec_file, model_file, focus_seq="SYN/1-600", theta=0.8, iterations="max", lambda_h=0.01,
                           lambda_J=plm.default_lambda_j(L, Q))
    assert r.optimization_status.startswith("converged"), r.optimization_status
    assert r.num_valid_sites == L and r.num_valid_seqs == N
    assert os.path.getsize(model_file) == 40 + Q + 4 * N + 5 * L + 8 * L * Q + 4 * L * (L - 1) * Q * Q
    m = model_io.read_model_file(model_file)
    assert m["L"] == L and m["q"] == Q and m["jij"].shape == (L * (L - 1) // 2, Q, Q)
    _, cn = plm.scores(m["jij"], L, Q)
    ecs = np.loadtxt(ec_file, usecols=(0, 2, 5))
    iu, ju = np.triu_indices(L, 1)
    np.testing.assert_array_equal(ecs[:, 0].astype(int), iu + 1)
    np.testing.assert_array_equal(ecs[:, 1].astype(int), ju + 1)
    np.testing.assert_allclose(ecs[:, 2], cn[iu, ju], atol=2e-6)            # 6 decimals in the text file
    inter = (iu < L1) & (ju >= L1)
    planted_inter = {(i, j) for (i, j) in planted if i < L1 <= j}
    order = np.argsort(-cn[iu, ju][inter])
    top = set(zip(iu[inter][order[:len(planted_inter)]].tolist(), ju[inter][order[:len(planted_inter)]].tolist()))
    assert len(planted_inter) >= 20 and len(top & planted_inter) >= 0.9 * len(planted_inter), (len(top & planted_inter), len(planted_inter))
