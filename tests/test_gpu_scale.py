"""
Parity and optimality at BASELINE.json scale.

The oracle (oracle/plm_oracle.c, float64, OpenMP) is the checker: one objective+gradient evaluation of it costs
2-20 s on the GPU box's 16 host cores at these sizes, so every test here spends a handful of them.
  * all five BASELINE configurations, and the reference's DEFAULT mode -- plmc -g, `ignore_gaps: True`
    (config/sample_config_monomer.txt:155, couplings/protocol.py:159-165: 20 model states) -- at the headline and
    config-3 shapes: the HIP evaluation against the f64 oracle far from the optimum and at the point the fit SHIPS
    (stop rule epsilon = 1e-3), with the error of the f32 CPU build (oracle32: sequential float32 arithmetic, the
    stand-in for a plmc openmp32 build) printed beside the HIP error at every stop point -- the HIP path must not be
    less accurate than the reference's own arithmetic class;
  * optimality certificates: the ORACLE's gradient at the GPU's stop point satisfies the stop rule (status 0 is
    required everywhere: no "converged to precision" escape), a tighter fit moves no CN score by 1e-4;
  * the PLAIN evaluation (k_fwd_w + three-plane k_bwd_w: what bench.py times and a fit runs until its last iterations)
    against the same oracle values at config 2 / headline / config 3 / -g headline: inside the float32 CPU build's error
    and inside an absolute bound per configuration;
  * config 2, headline and config 5 (two chains, L = 600): scipy's L-BFGS-B on the oracle's float64 objective, started
    from the shipped point and given 25 / 10 / 4 evaluations, moves no CN score by 1e-4 ('EC scores vs CPU plmc
    within 1e-4' with the only CPU solver available);
  * config 3 (N = 100 000) converges from two different starts;
  * at the reference's default `iterations: 100` (sample_config_monomer.txt:149), where no solver is converged, the
    default solver's CN ranking is at least as close to the converged one as the plmc-like joint L-BFGS's;
  * config 5 and the -g headline: the whole file round trip (A2M -> run_plmc_hip -> .model / _ECs.txt -> reader).
"""
import os

import numpy as np
import pytest

from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED

pytestmark = pytest.mark.gpu
Q = 21
# name -> (N, L, seed, ignore_gaps)
CONFIGS = {"config2": (20000, 200, BASE_SEED + 2, False), "headline": (50000, 300, BASE_SEED + 1, False),
           "config3": (100000, 300, BASE_SEED + 3, False), "config4": (50000, 500, BASE_SEED + 4, False),
           "config5": (30000, 600, BASE_SEED + 5, False),
           "headline_g": (50000, 300, BASE_SEED + 1, True), "config3_g": (100000, 300, BASE_SEED + 3, True)}
# configurations that get the full treatment (two fits); the larger ones get the shipped fit only -- their vectors are
# 220-320 MB each and an oracle evaluation costs 10-20 s of the box's host cores
FULL = ("config2", "headline", "config3", "headline_g")
# the "much tighter than the stop rule" fit.  The stop rule is certified on the point that SHIPS -- fields rounded to
# float32 -- and at N = 100 000 that rounding alone costs 4.7e-4 |x| (config 3: 7e-4 is what float32 fields can carry)
TIGHT = 4e-4
TIGHT_OF = {"config3": 7e-4}
# |g_hip - g_f64| / |x| allowed at a point a fit stopped at -- a constant since round 4: the last iterations of a fit and
# plm_eval run the accurate evaluation (exact integer forward GEMM, exact softmax arguments, 32-bit residuals;
# DESIGN.md 4.3 / section 5), measured 2.3e-5 at the headline and 4.7e-5 at N = 100 000.
# (Rounds 2-3, f32 accumulation throughout: 3e-11 N L -- 4.5e-4 at the headline, 1e-3 at N = 100 000, the size of the stop rule.)
GRAD_ERR = 8e-5
# The PLAIN evaluation -- k_fwd_w (f16 hi + lo operand planes, f32 accumulation) + __expf softmax + three residual digit
# planes on k_bwd_w: what a fit runs until its last iterations and what bench.py times -- against the f64 oracle at the
# same stop points.  Its error is coherent across sequences and grows as ~3e-11 N L |x|; the bounds
# are the values measured on the MI355X (profiles/r06_error_table.txt: 1.1e-4 / 3.5e-4 / 6.6e-4 / 2.5e-4 at the stop points,
# 1.2e-4 / 3.9e-4 / 8.6e-4 / 2.5e-4 far from the optimum; configs 4 / 5 / 3-g: 5.5e-4 / 4.5e-4 / 5.5e-4) + 30 % (config 3:
# + 12 % over its far point).  The second bound every
# point has to meet is relative: not worse than the float32 CPU build (oracle32) at the same point.
# Configs 4 / 5 / 3-g (round 6): their plain evaluation is checked at the stop point inside
# test_evaluation_matches_f64_oracle_at_scale, against the oracle values that test computes anyway.
# The measured values of every configuration and point are committed as profiles/r06_error_table.txt.
PLAIN_ERR = {"config2": 1.5e-4, "headline": 4.7e-4, "config3": 9.6e-4, "headline_g": 3.3e-4,
             "config4": 7.1e-4, "config5": 5.9e-4, "config3_g": 7.1e-4}
# the oracle's |g|/|x| at a point the fit reported converged at epsilon: the stop rule, up to the evaluation error
COND_SLACK = 1.05


@pytest.fixture(scope="module")
def plm():
    from evcouplings_amd import plm as _plm
    assert _plm.device_count() >= 1, "no gfx950 device: the HIP path has no fallback"
    return _plm


@pytest.fixture(scope="module", autouse=True)
def _oracle_threads(oracle64, oracle32):
    """These evaluations are big enough to use every core the box grants (cgroup quota aware)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    old = oracle64.num_threads()
    for o in (oracle64, oracle32):
        o.set_num_threads(max(1, min(n, 32)))
    yield
    for o in (oracle64, oracle32):
        o.set_num_threads(old)


def _context(plm, f, **kw):
    return plm.PlmContext(f["msa"], Q, ignore_gaps=f["gaps"], **kw)


@pytest.fixture(scope="module")
def fits(plm):
    """Per configuration: alignment, weights, a far-from-optimal point (20 iterations) and the converged fits."""
    cache = {}

    def get(name):
        if name in cache:
            return cache[name]
        N, L, seed, gaps = CONFIGS[name]
        msa, planted = synthetic_msa(N, L, seed=seed)
        out = {"msa": msa, "N": N, "L": L, "gaps": gaps, "planted": planted, "qm": Q - 1 if gaps else Q,
               "lambda_j": plm.default_lambda_j(L, Q - 1 if gaps else Q)}
        with _context(plm, out, max_iter=20, epsilon=1e-3) as ctx:
            out["w"], _, out["n_eff"] = ctx.reweight()
            ctx.marginals(pairs=False)
            ctx.set_x(None)
            ctx.optimize()
            out["x_far"] = ctx.get_x()
            ctx.set_options(max_iter=3000, epsilon=1e-3)
            r = ctx.optimize()
            out["fit_1e-3"] = dict(r, x=ctx.get_x(), cn=ctx.scores()[1], solver=ctx.solver_stats())
            if name in FULL:
                ctx.set_options(max_iter=1000, epsilon=TIGHT_OF.get(name, TIGHT))
                r = ctx.optimize()
                out["fit_tight"] = dict(r, x=ctx.get_x(), cn=ctx.scores()[1])
        if name in FULL:          # the others are visited by one test each: not kept (their vectors are 220-320 MB)
            cache[name] = out
        return out
    return get


def _oracle_eval(oracle, f, x):
    fn = oracle.eval_gaps if f["gaps"] else oracle.eval
    return fn(f["msa"], f["w"].astype(oracle.real), Q, 0.01, f["lambda_j"], x.astype(oracle.real))


def _oracle_cached(oracle, f, point):
    """oracle evaluation at f["x_far"] / the shipped point, computed once per configuration and precision"""
    key = "oracle_%s_%s" % (point, np.dtype(oracle.real).name)
    if key not in f:
        f[key] = _oracle_eval(oracle, f, f["x_far"] if point == "far" else f["fit_1e-3"]["x"])
    return f[key]


def _hip_eval_plain(plm, f, x, monkeypatch):
    """one evaluation with the PLAIN arithmetic of the fit's first iterations / of bench.py's timed window (PLM_FWD_ACCURATE
    is read once, when the context is created): k_fwd_w + plain k_hpass + the three-plane k_bwd_w at 21 states"""
    monkeypatch.setenv("PLM_FWD_ACCURATE", "0")
    try:
        with _context(plm, f, lambda_j=f["lambda_j"], epsilon=1e-3) as ctx:
            ctx.set_weights(f["w"])
            ctx.set_x(x)
            fx, nll = ctx.eval()
            return fx, nll, ctx.get_g()
    finally:
        monkeypatch.delenv("PLM_FWD_ACCURATE")


def _hip_eval(plm, f, x):
    """one evaluation through the resident-context API (plm_eval has no -g flag); always the accurate forward GEMM"""
    if not f["gaps"]:
        return plm.evaluate(f["msa"], f["w"], Q, 0.01, f["lambda_j"], x)
    with _context(plm, f, lambda_j=f["lambda_j"]) as ctx:
        ctx.set_weights(f["w"])
        ctx.set_x(x)
        fx, nll = ctx.eval()
        return fx, nll, ctx.get_g()


@pytest.mark.parametrize("name", ["config2", "headline", "config3", "config4", "config5", "headline_g", "config3_g"])
def test_evaluation_matches_f64_oracle_at_scale(plm, oracle64, oracle32, fits, monkeypatch, name):
    f = fits(name)
    assert f["fit_1e-3"]["status"] == 0, f["fit_1e-3"]["status_msg"]          # converged by its own rule, everywhere
    # ... and not by burning passes at the noise floor of the field gradient (the chain's stall rule ends it
    # within 100x of the tolerance; were the floor to sit in that band, every chain would run out of positions and be
    # continued by the host): a handful of continuations per fit at most, also at N = 100 000
    sv = f["fit_1e-3"]["solver"]
    assert sv["chains_continued_by_host"] <= 6 and sv["passes_per_evaluation"] <= 6.0, sv
    # far from the optimum: relative criteria (gradient entries are large).  Not at config 5: its oracle evaluation
    # costs ~20 s of host cores, the far point is held at six other shapes (config 4, L = 500, among them);
    # the scale of the gradient entries comes from the GPU's own far-point gradient there.
    fx, nll, g = _hip_eval(plm, f, f["x_far"])
    if name == "config5":
        gmax_far = np.abs(g).max()
    else:
        fxo, nllo, go = _oracle_cached(oracle64, f, "far")
        gmax_far = np.abs(go).max()
        assert abs(fx - fxo) <= 2e-6 * abs(fxo) and abs(nll - nllo) <= 2e-6 * abs(nllo)
        assert np.abs(g - go).max() <= 2e-5 * gmax_far
    # at the converged point the gradient itself is tiny: the error must stay well inside the stop rule's scale
    x = f["fit_1e-3"]["x"]
    xn = max(1.0, np.linalg.norm(x))
    fx, nll, g = _hip_eval(plm, f, x)
    fxo, nllo, go = _oracle_cached(oracle64, f, "stop")
    assert abs(fx - fxo) <= 2e-6 * abs(fxo)
    err = np.linalg.norm(g - go) / xn
    cond64 = np.linalg.norm(go) / xn
    # the same point through the float32 CPU build: the arithmetic class of a plmc openmp32 binary
    _, _, g32 = _oracle_cached(oracle32, f, "stop")
    err32 = np.linalg.norm(g32.astype(np.float64) - go) / xn
    print("%s: at the stop point |g_hip - g_f64|/|x| = %.3g, |g_f32cpu - g_f64|/|x| = %.3g, oracle cond %.4g "
          "(fit reported %.4g), %d iterations / %d evaluations" % (
              name, err, err32, cond64, f["fit_1e-3"]["table"][-1][2], f["fit_1e-3"]["iters"], f["fit_1e-3"]["n_evals"]))
    assert err <= GRAD_ERR, (err, GRAD_ERR)
    assert err <= 1.5 * err32, (err, err32)      # not narrower than the reference's arithmetic class
    assert np.abs(g - go).max() <= 2e-5 * gmax_far
    # optimality as the ORACLE sees it: the stop rule, up to the evaluation error
    assert cond64 < COND_SLACK * 1e-3, cond64
    f["cond64_1e-3"] = cond64
    if name not in FULL:
        # the PLAIN kernels (what bench.py times) at the shapes test_plain_evaluation_... does not
        # visit, against the oracle values already in hand: one more HIP evaluation, no oracle time
        fxp, nllp, gp = _hip_eval_plain(plm, f, x, monkeypatch)
        errp = np.linalg.norm(gp - go) / xn
        print("%s (stop point): PLAIN evaluation |g_hip - g_f64|/|x| = %.3g (bound %.3g), |g_f32cpu - g_f64|/|x| = %.3g" % (
            name, errp, PLAIN_ERR[name], err32))
        assert abs(fxp - fxo) <= 2e-6 * abs(fxo) and abs(nllp - nllo) <= 2e-6 * abs(nllo)
        assert errp <= err32, (errp, err32)
        assert errp <= PLAIN_ERR[name], errp


@pytest.mark.parametrize("name", ["config2", "headline", "config3", "headline_g"])
def test_plain_evaluation_matches_f64_oracle_at_scale(plm, oracle64, oracle32, fits, monkeypatch, name):
    """the kernels bench.py times (and every fit runs until its last ~8 evaluations) are the PLAIN
    ones; test_evaluation_matches_f64_oracle_at_scale goes through the accurate evaluation.  Here the plain path is held
    against the f64 oracle at BASELINE scale, far from the optimum and at the shipped stop point: (a) not further from
    f64 than the float32 CPU build (the arithmetic class of a plmc openmp32 binary) at the same point, (b) inside the
    absolute bound of this configuration."""
    f = fits(name)
    for point in ("far", "stop"):
        x = f["x_far"] if point == "far" else f["fit_1e-3"]["x"]
        xn = max(1.0, np.linalg.norm(x))
        fx, nll, g = _hip_eval_plain(plm, f, x, monkeypatch)
        fxo, nllo, go = _oracle_cached(oracle64, f, point)
        _, _, g32 = _oracle_cached(oracle32, f, point)
        err = np.linalg.norm(g - go) / xn
        err32 = np.linalg.norm(g32.astype(np.float64) - go) / xn
        print("%s (%s point): PLAIN evaluation |g_hip - g_f64|/|x| = %.3g (bound %.3g), |g_f32cpu - g_f64|/|x| = %.3g, "
              "max |dg| / max |g64| = %.3g" % (name, point, err, PLAIN_ERR[name], err32,
                                                np.abs(g - go).max() / np.abs(go).max()))
        assert abs(fx - fxo) <= 2e-6 * abs(fxo) and abs(nll - nllo) <= 2e-6 * abs(nllo)
        assert err <= err32, (point, err, err32)            # (a) inside the reference's own arithmetic class
        assert err <= PLAIN_ERR[name], (point, err)         # (b)
        if point == "far":
            assert np.abs(g - go).max() <= 1e-4 * np.abs(go).max()


@pytest.mark.parametrize("name", ["config2", "headline", "config3", "headline_g"])
def test_fit_optimality_certificate(plm, oracle64, fits, name):
    f = fits(name)
    a, b = f["fit_1e-3"], f["fit_tight"]
    assert a["status"] == 0, a["status_msg"]                             # converged by its own rule
    assert a["table"][-1][2] < 1e-3
    # the oracle agrees: its float64 gradient at the GPU's final point satisfies the rule up to the evaluation error
    # (computed by the evaluation test above when it ran on this configuration first)
    cond64 = f.get("cond64_1e-3")
    if cond64 is None:
        _, _, go = _oracle_cached(oracle64, f, "stop")
        cond64 = np.linalg.norm(go) / max(1.0, np.linalg.norm(a["x"]))
    assert cond64 < COND_SLACK * 1e-3, cond64
    # 1.4 - 2.5 times tighter moves no EC score by more than 1e-4 (BASELINE.json's tolerance on EC scores)
    tight = TIGHT_OF.get(name, TIGHT)
    # (a fit that stopped below the tighter tolerance already has nothing left to do: empty table)
    last = b["table"][-1][2] if b["table"] else a["table"][-1][2]
    assert b["status"] == 0 and last < tight and "rounding" not in b["status_msg"], (b["status_msg"], last)
    assert np.abs(a["cn"] - b["cn"]).max() < 1e-4
    _, _, gob = _oracle_eval(oracle64, f, b["x"])
    cond64_tight = np.linalg.norm(gob) / max(1.0, np.linalg.norm(b["x"]))
    print("%s: oracle cond at the eps = 1e-3 point %.4g, at the eps = %.1g point %.4g (%d more iterations)" % (
        name, cond64, tight, cond64_tight, b["iters"]))
    assert cond64_tight < COND_SLACK * tight, cond64_tight


@pytest.mark.parametrize("name,maxfun", [("config2", 25), ("headline", 25), ("config5", 4)])
def test_cn_within_1e4_of_an_independent_f64_optimiser(plm, oracle64, fits, name, maxfun):
    """BASELINE.json config 2: 'EC scores vs CPU plmc within 1e-4' (and the same at the headline and at config 5's two-chain
    shape, where an oracle evaluation costs ~20 s of host cores: 4 of them; 10 at the headline -- the whole GPU suite is
    kept under ~8 minutes of a 16-core host, and a shipped point that already meets the stop rule moves most in the
    first steps of an independent optimiser).  plmc is
    unobtainable; the CPU side here is scipy's L-BFGS-B minimising the ORACLE's float64 objective,
    started from the answer the drop-in SHIPS (stop rule epsilon = 1e-3, not the tighter fit) and given 25 evaluations:
    whatever it still gains must not move a CN score by 1e-4."""
    import scipy.optimize as so
    f = fits(name)
    shipped = f["fit_1e-3"]
    x0 = shipped["x"].astype(np.float64)
    w64 = f["w"].astype(np.float64)

    last, first = {}, []

    def fun(x):
        key = x.tobytes()
        if key not in last:
            fx, _, g = oracle64.eval(f["msa"], w64, Q, 0.01, f["lambda_j"], x)
            last.clear()
            last[key] = (fx, g)
            first.append(fx)       # first[0] = the objective at the shipped point (scipy evaluates its start itself)
        return last[key]

    res = so.minimize(fun, x0, jac=True, method="L-BFGS-B", options=dict(maxfun=maxfun, maxcor=10, ftol=0, gtol=0))
    f0 = first[0]
    assert res.fun <= f0 * (1 + 1e-12)
    L = f["L"]
    _, cn_cpu = oracle64.scores(res.x[L * Q:], L, Q)
    print("%s: scipy f64 from the shipped point: f %.6f -> %.6f, max |dCN| %.3g" % (
        name, f0, res.fun, np.abs(cn_cpu - shipped["cn"]).max()))
    assert np.abs(cn_cpu - shipped["cn"]).max() < 1e-4
    if "fit_tight" in f:
        assert np.abs(cn_cpu - f["fit_tight"]["cn"]).max() < 1e-4


def test_config3_converges_from_two_starts(plm, oracle64, fits):
    """N = 100 000: rounds 2-3 needed 290-750 iterations here and one run in ten ended 'converged to precision' at
    |g|/|x| = 3.4e-3 -- the last decade was a walk on the error of the plain forward GEMM.  With the accurate forward
    GEMM in the last iterations the fit must meet the stop rule (status 0) from the standard start AND from a perturbed
    one, in a bounded number of iterations, at the same CN scores."""
    f = fits("config3")
    a = f["fit_1e-3"]
    assert a["status"] == 0 and a["iters"] <= 300, (a["status_msg"], a["iters"])
    rng = np.random.default_rng(3)
    with _context(plm, f, max_iter=3000, epsilon=1e-3) as ctx:
        ctx.set_weights(f["w"])
        x0 = f["x_far"] + (0.02 * rng.normal(size=f["x_far"].size)).astype(np.float32)
        ctx.set_x(x0)
        r = ctx.optimize()
        cn = ctx.scores()[1]
    print("config3: standard start %d iterations, perturbed start %d iterations (%s), max |dCN| %.3g" % (
        a["iters"], r["iters"], r["status_msg"], np.abs(cn - a["cn"]).max()))
    # (the perturbed start has no good fields to begin with: 260-330 iterations over this round's runs)
    assert r["status"] == 0 and r["iters"] <= 400, (r["status_msg"], r["iters"])
    assert np.abs(cn - a["cn"]).max() < 1e-4


def _ranking_agreement(cn, cn_ref, L, top):
    from scipy.stats import spearmanr
    iu, ju = np.triu_indices(L, 6)                      # |i - j| >= 6, the pairs EC lists are read for
    a, b = cn[iu, ju], cn_ref[iu, ju]
    ta, tb = set(np.argsort(-a)[:top].tolist()), set(np.argsort(-b)[:top].tolist())
    return spearmanr(a, b).correlation, len(ta & tb) / float(top)


def test_default_solver_at_100_iterations_is_no_worse_than_the_plmc_like_route(plm, fits):
    """The reference's call site passes iterations = 100 (config/sample_config_monomer.txt:149), where neither solver is
    converged and the two stop at different points.  The drop-in's default (variable projection) must then be at least
    as close to the CONVERGED answer as joint L-BFGS (the algorithm plmc runs) in what a user reads off the EC file:
    rank correlation of the long-range CN scores and the overlap of the top-L pairs."""
    f = fits("headline")
    L = f["L"]
    got = {}
    for solver in ("vp", "joint"):
        with _context(plm, f, max_iter=100, epsilon=1e-3, joint=(solver == "joint")) as ctx:
            ctx.set_weights(f["w"])
            ctx.marginals(pairs=False)
            ctx.set_x(None)
            r = ctx.optimize()
            got[solver] = _ranking_agreement(ctx.scores()[1], f["fit_tight"]["cn"], L, L) + (r["table"][-1][2],)
    print("headline at 100 iterations: vp spearman %.4f top-L overlap %.3f (|g|/|x| %.3g); joint %.4f %.3f (%.3g)" % (
        got["vp"] + got["joint"]))
    assert got["vp"][0] >= got["joint"][0] - 1e-3 and got["vp"][1] >= got["joint"][1] - 0.01
    assert got["vp"][1] >= 0.85


def _file_round_trip(plm, tmp_path, name, ignore_gaps):
    from evcouplings_amd import model_io, tools
    from evcouplings_amd.synthetic import msa_to_a2m
    N, L, seed, _ = CONFIGS[name]
    msa, planted = synthetic_msa(N, L, seed=seed)
    ali = msa_to_a2m(msa, str(tmp_path / "c.a2m"))
    ec_file, model_file = str(tmp_path / "c_ECs.txt"), str(tmp_path / "c.model")
    qm = Q - 1 if ignore_gaps else Q
    r = tools.run_plmc_hip(ali, ec_file, model_file, focus_seq="SYN/1-%d" % L, theta=0.8, iterations="max", lambda_h=0.01,
                           lambda_J=plm.default_lambda_j(L, qm), ignore_gaps=ignore_gaps)
    assert r.optimization_status.startswith("converged"), r.optimization_status
    assert r.num_valid_sites == L and r.num_valid_seqs == N
    assert os.path.getsize(model_file) == 40 + qm + 4 * N + 5 * L + 8 * L * qm + 4 * L * (L - 1) * qm * qm
    m = model_io.read_model_file(model_file)
    assert m["L"] == L and m["q"] == qm and m["jij"].shape == (L * (L - 1) // 2, qm, qm)
    if ignore_gaps:
        assert m["alphabet"] == "ACDEFGHIKLMNPQRSTVWY"              # plmc -g writes a 20-letter model
    _, cn = plm.scores(m["jij"], L, qm)
    ecs = np.loadtxt(ec_file, usecols=(0, 2, 5))
    iu, ju = np.triu_indices(L, 1)
    np.testing.assert_array_equal(ecs[:, 0].astype(int), iu + 1)
    np.testing.assert_array_equal(ecs[:, 1].astype(int), ju + 1)
    np.testing.assert_allclose(ecs[:, 2], cn[iu, ju], atol=2e-6)            # 6 decimals in the text file
    return cn, planted, iu, ju


def test_config5_two_chain_file_round_trip(plm, tmp_path):
    """BASELINE.json config 5: EVcomplex-style concatenated alignment, L = 600 = 350 + 250, N = 30 000 -- 'inter-chain EC
    scoring, CouplingsModel round-trip'.  A2M file -> the run_plmc drop-in (fit to the stop rule) -> _ECs.txt and the
    634 MB plmc_v2 .model -> read back -> scores recomputed from the file's couplings must be the EC file's, and the
    planted inter-chain couplings must lead the inter-chain ranking (what the reference's complex protocol extracts,
    couplings/protocol.py:521-560)."""
    L1 = 350
    cn, planted, iu, ju = _file_round_trip(plm, tmp_path, "config5", False)
    inter = (iu < L1) & (ju >= L1)
    planted_inter = {(i, j) for (i, j) in planted if i < L1 <= j}
    order = np.argsort(-cn[iu, ju][inter])
    top = set(zip(iu[inter][order[:len(planted_inter)]].tolist(), ju[inter][order[:len(planted_inter)]].tolist()))
    assert len(planted_inter) >= 20 and len(top & planted_inter) >= 0.9 * len(planted_inter), (len(top & planted_inter), len(planted_inter))


def test_headline_ignore_gaps_file_round_trip(plm, tmp_path):
    """The reference's default mode at the headline shape through the run_plmc drop-in: `ignore_gaps=True`
    (couplings/tools.py:222-224 passes -g) writes a 20-letter plmc_v2 model; the planted pairs lead the ranking."""
    cn, planted, iu, ju = _file_round_trip(plm, tmp_path, "headline_g", True)
    order = np.argsort(-cn[iu, ju])
    top = set(zip(iu[order[:len(planted)]].tolist(), ju[order[:len(planted)]].tolist()))
    assert len(top & set(planted)) >= 0.9 * len(planted), (len(top & set(planted)), len(planted))
