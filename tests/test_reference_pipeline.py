"""
Config-1 "plumbing" row: the REFERENCE's own couplings protocol (evcouplings/couplings/
protocol.py:363 `standard`) driven end to end on top of our backend, using the reference's
unmodified readers (pairs.read_raw_ec_file, CouplingsModel), segment mapping, rescoring and
post-processing.

The build container has the reference but no GPU; the GPU box has a GPU but no reference.  So
the solver output used here is a fixture produced ON an MI355X by scripts/make_gpu_fixture.py
(tests/golden/hip_fit_L24.*: alignment, raw fit arrays, and the two files the HIP path wrote),
and `plm.fit` is replaced by a function that returns those arrays -- everything else is the
real host code and the real reference code.  Skipped where /root/reference is absent.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refstubs  # noqa: E402

pytestmark = pytest.mark.skipif(not refstubs.reference_available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    refstubs.install()
    import evcouplings.couplings.protocol as cp
    import evcouplings.couplings.tools as ct
    from evcouplings.couplings.model import CouplingsModel
    return dict(cp=cp, ct=ct, CouplingsModel=CouplingsModel)


@pytest.fixture()
def gpu_fit(golden_dir, monkeypatch):
    z = np.load(os.path.join(golden_dir, "hip_fit_L24.npz"))
    fit = {k: z[k] for k in ("weights", "fi", "fij", "hi", "jij", "fn", "cn")}
    fit.update(n_eff=float(z["n_eff"]), iters=int(z["iters"]), n_evals=int(z["n_evals"]), status=int(z["status"]),
               status_msg=str(z["status_msg"]), fx=float(z["fx"]), lambda_j=float(z["lambda_j"]),
               table=[tuple(r) for r in z["table"]], seconds={})
    from evcouplings_amd import plm
    calls = []

    def fake_fit(msa, **kw):
        calls.append((msa.shape, kw))
        return dict(fit)

    monkeypatch.setattr(plm, "fit", fake_fit)
    return fit, calls, z


def test_files_written_on_the_gpu_load_in_the_reference(ref, golden_dir):
    """the .model and EC file the HIP path wrote on the MI355X, read by the reference's readers"""
    m = ref["CouplingsModel"](os.path.join(golden_dir, "hip_fit_L24.model"))
    z = np.load(os.path.join(golden_dir, "hip_fit_L24.npz"))
    assert (m.L, m.num_symbols, m.N_valid, m.N_invalid) == (24, 21, 500, 0)
    assert m.index_list.tolist() == list(range(10, 34)) and "".join(m.alphabet) == "-ACDEFGHIKLMNPQRSTVWY"
    iu = np.triu_indices(24, 1)
    np.testing.assert_array_equal(m.J_ij[iu].astype(np.float32), z["jij"])
    np.testing.assert_array_equal(m.h_i.astype(np.float32), z["hi"])
    assert m.N_eff == pytest.approx(float(z["n_eff"]), rel=1e-6) and m.lambda_h == pytest.approx(0.01)
    # CouplingsModel re-derives CN in float64 (model.py:777-827); the GPU scored in f32
    np.testing.assert_allclose(m.cn_scores, z["cn"], atol=5e-6)
    from evcouplings.couplings.pairs import read_raw_ec_file
    ecs = read_raw_ec_file(os.path.join(golden_dir, "hip_fit_L24_ECs.txt"))
    ref_ecs = m.ecs
    merged = ecs.merge(ref_ecs, on=["i", "j"], suffixes=("", "_ref"))
    assert len(merged) == 276
    np.testing.assert_allclose(merged["cn"], merged["cn_ref"], atol=5e-6)
    assert (merged["A_i"] == merged["A_i_ref"]).all()
    # the planted pairs top the reference's own ranking of our output
    planted = {(int(a) + 10, int(b) + 10) for a, b in z["planted"]}
    # (long-range only: terminal gap runs make sequence neighbours co-vary as well)
    lr = ecs[(ecs["j"] - ecs["i"]).abs() >= 6]
    top = set(zip(lr["i"].head(8).tolist(), lr["j"].head(8).tolist()))
    assert len(top & planted) >= 6, (top, planted)
    # mutation-effect consumers (mutate stage) work on it
    assert np.isfinite(m.to_independent_model().h_i).all()


def test_reference_standard_protocol_runs_on_our_backend(ref, gpu_fit, golden_dir, tmp_path):
    fit, calls, z = gpu_fit
    from evcouplings_amd import protocol as hip_protocol
    from evcouplings_amd import tools as hip_tools
    cp = ref["cp"]
    prefix = str(tmp_path / "couplings" / "job")
    kwargs = dict(
        prefix=prefix, alignment_file=os.path.join(golden_dir, "hip_fit_L24.a2m"), focus_mode=True,
        focus_sequence="SYN/10-33", segments=[["aa", "A_1", 10, 33, list(range(10, 34))]] if False else None,
        theta=0.8, alphabet=None, ignore_gaps=False, iterations=100, lambda_h=0.01, lambda_J=0.01,
        lambda_J_times_Lq=True, lambda_group=None, scale_clusters=None, cpu=2, plmc="plmc", reuse_ecs=False,
        min_sequence_distance=6, frequencies_file=None, scoring_model="skewnormal", save_model=True)
    hip_protocol.install()
    try:
        assert ref["ct"].run_plmc is hip_tools.run_plmc_hip
        outcfg = cp.run(protocol="standard", **kwargs)
    finally:
        hip_protocol.uninstall()
    assert ref["ct"].run_plmc is not hip_tools.run_plmc_hip
    # our backend was called with the lambda_J the reference scaled (protocol.py:159-179)
    (shape, kw), = calls
    assert shape == (500, 24) and kw["lambda_j"] == pytest.approx(0.01 * 20 * 23) and kw["max_iter"] == 100
    assert kw["theta_id"] == 0.8 and kw["q"] == 21
    assert outcfg["num_sites"] == 24 and outcfg["num_valid_sequences"] == 500 and outcfg["region_start"] == 10
    assert outcfg["effective_sequences"] == pytest.approx(float(z["n_eff"]), abs=0.06)
    for key in ("raw_ec_file", "model_file", "ec_file", "ec_longrange_file"):
        assert os.path.getsize(outcfg[key]) > 0, key
    import pandas as pd
    table = pd.read_csv(outcfg["ec_file"])
    assert {"i", "j", "cn", "probability"} <= set(table.columns) and len(table) == 276
    it = pd.read_csv(prefix + "_iteration_table.csv")
    assert len(it) == 100 and list(it.columns)[1:] == hip_tools.ITER_COLUMNS
    # restart path: reuse_ecs short-circuits the solver (protocol.py:186-199) and still works
    calls.clear()
    hip_protocol.install()
    try:
        out2 = cp.run(protocol="standard", **{**kwargs, "reuse_ecs": True})
    finally:
        hip_protocol.uninstall()
    assert calls == [] and out2["num_sites"] == 24


def test_registered_protocol_selects_the_gpu_path_and_carries_the_solver_options(ref, gpu_fit, golden_dir, tmp_path, monkeypatch):
    """`protocol: standard_hip` in a pipeline config (couplings/protocol.py:934-974 looks the key up in PROTOCOLS): the
    reference's `standard` runs with the HIP solver installed for the call only; the extra config keys hip_solver /
    hip_conventions reach the solver, the environment is left as it was."""
    fit, calls, z = gpu_fit
    from evcouplings_amd import protocol as hip_protocol
    from evcouplings_amd import tools as hip_tools
    cp = ref["cp"]
    monkeypatch.delenv("PLM_HIP_SOLVER", raising=False)
    monkeypatch.delenv("PLM_HIP_CONVENTIONS", raising=False)
    reg = hip_protocol.register_protocols()
    try:
        assert {"standard", "complex", "mean_field", "standard_hip", "complex_hip"} <= set(reg)
        kwargs = dict(
            prefix=str(tmp_path / "c" / "job"), alignment_file=os.path.join(golden_dir, "hip_fit_L24.a2m"), focus_mode=True,
            focus_sequence="SYN/10-33", segments=None, theta=0.8, alphabet=None, ignore_gaps=False, iterations=100,
            lambda_h=0.01, lambda_J=0.01, lambda_J_times_Lq=True, lambda_group=None, scale_clusters=None, cpu=2,
            plmc="plmc", reuse_ecs=False, min_sequence_distance=6, frequencies_file=None, scoring_model="skewnormal",
            save_model=True, hip_solver="joint", hip_conventions=512)
        outcfg = cp.run(protocol="standard_hip", **kwargs)
        assert ref["ct"].run_plmc is not hip_tools.run_plmc_hip                      # installed for the call only
        (shape, kw), = calls
        assert shape == (500, 24) and kw["joint"] is True and kw["conventions"] == 512
        assert "PLM_HIP_SOLVER" not in os.environ and "PLM_HIP_CONVENTIONS" not in os.environ
        assert outcfg["num_sites"] == 24 and os.path.getsize(outcfg["ec_file"]) > 0
        # a hook somebody installed process-wide survives a *_hip protocol call
        hip_protocol.install()
        try:
            kwargs["prefix"] = str(tmp_path / "d" / "job")
            cp.run(protocol="standard_hip", **kwargs)
            assert ref["ct"].run_plmc is hip_tools.run_plmc_hip and len(calls) == 2
        finally:
            hip_protocol.uninstall()
        assert ref["ct"].run_plmc is not hip_tools.run_plmc_hip
    finally:
        hip_protocol.unregister_protocols()
    assert "standard_hip" not in cp.PROTOCOLS


def test_fast_model_reader_is_a_drop_in_for_the_reference_reader(ref, golden_dir):
    """model_accel's vectorised plmc_v2 reader vs the reference's own (model.py:317-400): every attribute the
    reference reader sets, same dtype, same values; derived scores unchanged."""
    import evcouplings.couplings.model as ref_model
    from evcouplings_amd import model_accel
    for fname in ("tiny_L12.model", "hip_fit_L24.model"):
        path = os.path.join(golden_dir, fname)
        slow = ref["CouplingsModel"](path)
        model_accel.install(ref_model, reader=True)
        try:
            fast = ref["CouplingsModel"](path)
            with open(path, "rb") as fh:                      # file-object form (the recommended use, model.py:247-249)
                fast2 = ref["CouplingsModel"](fh)
        finally:
            model_accel.uninstall(ref_model)
        again = ref["CouplingsModel"](path)                   # uninstall restored the original reader
        for name in ("L", "num_symbols", "N_valid", "N_invalid", "num_iter", "theta", "lambda_h", "lambda_J",
                     "lambda_group", "N_eff", "alphabet", "weights", "_target_seq", "index_list", "f_i", "h_i",
                     "f_ij", "J_ij", "target_seq_mapped"):
            for other in (fast, fast2, again):
                a, b = getattr(slow, name), getattr(other, name)
                assert np.asarray(a).dtype == np.asarray(b).dtype and np.asarray(a).shape == np.asarray(b).shape, name
                np.testing.assert_array_equal(a, b, err_msg=name)
        np.testing.assert_array_equal(slow.cn_scores, fast.cn_scores)
        assert slow.has_target_seq == fast.has_target_seq


def test_mean_field_drop_in_behind_the_reference_classes(ref, golden_dir, monkeypatch, tmp_path):
    """evcouplings_amd.mean_field.install(): the reference's MeanFieldDCA / MeanFieldCouplingsModel run on top of
    our fit and DI functions.  No GPU here, so plm.mean_field / plm.direct_information are replaced by the numpy
    oracle fed with the reference's own weights and frequencies -- the glue (attributes, dense layouts, model
    object, ECs table, model file) is what is under test; the GPU arithmetic is tested in test_gpu_parity.py."""
    import evcouplings.couplings.mean_field as ref_mf
    from evcouplings.align.alignment import Alignment
    from evcouplings_amd import mean_field as our_mf, plm
    from oracle import meanfield_ref

    a2m = os.path.join(golden_dir, "hip_fit_L24.a2m")
    with open(a2m) as f:
        ali = Alignment.from_file(f, "fasta")

    def fake_mean_field(msa, q, theta_id=0.8, pseudo_count=0.5, **kw):
        from oracle.oracle import Oracle
        o = Oracle("f64")
        counts = o.reweight(msa, theta_id)
        w = 1.0 / counts
        fi, fij = o.marginals(msa, w, q)
        out = meanfield_ref.mean_field(fi, fij, pseudo_count, want_di=False)
        return dict(weights=w.astype(np.float32), n_eff=float(w.sum()), fi=fi.astype(np.float32),
                    fij=fij.astype(np.float32), hi=out["hi"], jij=out["jij"].astype(np.float32),
                    jij_full=out["jij_full"])

    monkeypatch.setattr(plm, "mean_field", fake_mean_field)
    monkeypatch.setattr(plm, "direct_information", lambda J, f: meanfield_ref.direct_information(np.asarray(J), np.asarray(f)))

    # the reference's num_cluster_members calls range(L) with a float L (alignment.py:1216,1225): legal under numba,
    # a TypeError in CPython -- stand in the oracle's counts, pinned equal to it by
    # tests/golden/reweight_freqs.npz
    import evcouplings.align.alignment as ref_ali
    from oracle.oracle import Oracle
    monkeypatch.setattr(ref_ali, "num_cluster_members",
                        lambda matrix, thr: Oracle("f64").reweight(np.asarray(matrix).astype(np.int8), thr).astype(float))
    slow_dca = ref_mf.MeanFieldDCA(ali)
    slow = slow_dca.fit(theta=0.8, pseudo_count=0.5)                           # the reference's own fit
    with open(a2m) as f:
        ali2 = Alignment.from_file(f, "fasta")
    our_mf.install(ref_mf)
    try:
        fast_dca = ref_mf.MeanFieldDCA(ali2)
        fast = fast_dca.fit(theta=0.8, pseudo_count=0.5)
        # the attributes the reference's fit leaves on the MeanFieldDCA object (mean_field.py:196-205)
        np.testing.assert_allclose(fast_dca.covariance_matrix, slow_dca.covariance_matrix, atol=1e-6)
        np.testing.assert_allclose(fast_dca.covariance_matrix_inv, slow_dca.covariance_matrix_inv,
                                   atol=2e-4 * np.abs(slow_dca.covariance_matrix_inv).max())
        np.testing.assert_array_equal(fast_dca.reshape_invC_to_4d(), fast.J_ij)
        ecs_fast = fast.ecs
        out_model = str(tmp_path / "mf.model")
        fast.to_file(out_model)
    finally:
        our_mf.uninstall(ref_mf)
    assert type(fast) is type(slow)
    np.testing.assert_allclose(fast.weights, slow.weights, rtol=1e-6)
    np.testing.assert_allclose(fast.f_i, slow.f_i, atol=1e-6)
    np.testing.assert_allclose(fast.f_ij, slow.f_ij, atol=1e-6)
    np.testing.assert_allclose(fast.J_ij, slow.J_ij, atol=2e-4 * np.abs(slow.J_ij).max())   # f32 frequencies in between
    np.testing.assert_allclose(fast.h_i, slow.h_i, atol=2e-4 * np.abs(slow.h_i).max())
    ecs_slow = slow.ecs
    assert list(ecs_fast.columns) == list(ecs_slow.columns) and "di" in ecs_fast.columns
    a = ecs_fast.sort_values(["i", "j"])
    b = ecs_slow.sort_values(["i", "j"])
    np.testing.assert_allclose(a["di"].values, b["di"].values, atol=2e-4 * b["di"].max())
    np.testing.assert_allclose(a["cn"].values, b["cn"].values, atol=2e-4 * np.abs(b["cn"]).max())
    # the file written through the reference's to_file comes back as a mean-field model
    back = ref["CouplingsModel"](out_model)
    assert type(back).__name__ == "MeanFieldCouplingsModel" and back.L == fast.L


def test_alignment_accel_installs_into_the_reference_module(ref, golden_dir, monkeypatch):
    """alignment_accel.install() on the real evcouplings.align.alignment: Alignment.set_weights / .frequencies /
    .pair_frequencies run through our wrappers (plm.* replaced by the oracle here: no GPU in this container)."""
    import evcouplings.align.alignment as ref_ali
    from evcouplings_amd import alignment_accel, plm
    from oracle.oracle import Oracle
    o = Oracle("f64")
    monkeypatch.setattr(plm, "reweight", lambda msa, thr: o.reweight(msa, thr))

    def fake_marginals(msa, w, q, pairs=True):
        fi, fij = o.marginals(msa, np.asarray(w, dtype=np.float64), q, pairs=pairs) if pairs else (o.marginals(msa, np.asarray(w, dtype=np.float64), q, pairs=False), None)
        fi = fi[0] if isinstance(fi, tuple) else fi
        return fi.astype(np.float32), (None if fij is None else fij.astype(np.float32))

    monkeypatch.setattr(plm, "marginals", fake_marginals)
    with open(os.path.join(golden_dir, "hip_fit_L24.a2m")) as f:
        ali = ref_ali.Alignment.from_file(f, "fasta")
    z = np.load(os.path.join(golden_dir, "hip_fit_L24.npz"))
    alignment_accel.install(ref_ali)
    try:
        ali.set_weights(identity_threshold=0.8)
        fi, fij = ali.frequencies, ali.pair_frequencies
    finally:
        alignment_accel.uninstall(ref_ali)
    np.testing.assert_allclose(ali.weights, z["weights"], rtol=1e-6)          # the MI355X fit's weights
    L = fi.shape[0]
    np.testing.assert_allclose(fi, z["fi"], atol=2e-6)
    np.testing.assert_allclose(fij[np.triu_indices(L, 1)], z["fij"], atol=2e-6)
    assert fij.shape == (L, L, 21, 21) and np.allclose(fij[3, 3], np.diag(fi[3]))


def test_align_stage_descriptions_through_the_drop_ins(ref, golden_dir, monkeypatch):
    """the reference's describe_frequencies / describe_seq_identities / describe_coverage and
    the two filters of modify_alignment (align/protocol.py:463-640, 900-943), run with alignment_accel installed,
    reproduce the tables the unmodified reference produced (tests/golden/align_stats.npz, make_golden_align.py).
    No GPU here: plm.reweight / marginals / alignment_stats are numpy stand-ins; map_matrix is the real drop-in."""
    import evcouplings.align.alignment as ref_ali
    import evcouplings.align.protocol as ref_prot
    from evcouplings_amd import alignment_accel, plm
    from oracle.oracle import Oracle
    o = Oracle("f64")
    z = np.load(os.path.join(golden_dir, "align_stats.npz"))
    monkeypatch.setattr(plm, "reweight", lambda msa, thr, **kw: o.reweight(msa, thr))

    def fake_marginals(msa, w, q, pairs=True):
        fi, fij = o.marginals(msa, np.asarray(w, dtype=np.float64), q, pairs=pairs)
        return fi.astype(np.float32), (None if fij is None else fij.astype(np.float32))

    def fake_stats(msa, gap_state=0, query=None, device=0):
        msa = np.asarray(msa)
        ident = None if query is None else (msa == np.asarray(query)[None, :]).sum(1).astype(np.int32)
        return (msa == gap_state).sum(1).astype(np.int32), (msa == gap_state).sum(0).astype(np.int32), ident

    monkeypatch.setattr(plm, "marginals", fake_marginals)
    monkeypatch.setattr(plm, "alignment_stats", fake_stats)
    chars = z["chars"].astype(str)
    alignment_accel.install(ref_ali)
    try:
        ali = ref_ali.Alignment(chars, np.array(["s%d" % k for k in range(len(chars))]), alphabet=str(z["alphabet"]))
        np.testing.assert_array_equal(ref_ali.map_matrix(ali.matrix, ali.alphabet_map), z["mapped"])
        np.testing.assert_array_equal(ref_ali.map_matrix(z["odd_chars"].astype(str), ali.alphabet_map), z["odd_mapped"])
        np.testing.assert_allclose(ali.identities_to(ali[0]), z["ident_to_target"], rtol=0, atol=0)
        keep, lc = alignment_accel.alignment_filters(z["mapped"], 0, int(z["min_seq"]), float(z["min_col"]))
        np.testing.assert_array_equal(keep, z["keep_seqs"])
        np.testing.assert_array_equal(lc, z["lc_cols"])
        # the reference's own filter expressions (align/protocol.py:906-912, 941) through the installed Alignment.count
        assert ref_ali.Alignment.count is alignment_accel.alignment_count
        np.testing.assert_array_equal(ali.count("-", axis="seq"), z["seq_gap_frac"])
        np.testing.assert_array_equal(ali.count("-", axis="pos"), z["col_gap_frac"])
        assert ali.count("-", axis="pos", normalize=False).dtype == np.int64
        keep2 = (1 - ali.count("-", axis="seq")) >= int(z["min_seq"]) / 100
        np.testing.assert_array_equal(keep2, z["keep_seqs"])
        kept = ali.select(sequences=keep2)
        np.testing.assert_array_equal(kept.count(kept._match_gap, axis="pos") > 1 - float(z["min_col"]), z["lc_cols"])
        with pytest.raises(ValueError):
            ali.count("-", axis="rows")
        kept.set_weights(0.8)
        np.testing.assert_allclose(kept.weights, z["weights"], rtol=1e-12)
        freq = ref_prot.describe_frequencies(kept, 10, target_seq_index=0)
        assert list(freq.columns) == list(z["freq_columns"]) and list(freq["A_i"]) == list(z["freq_target"])
        np.testing.assert_allclose(freq.drop(columns=["A_i"]).to_numpy(dtype=float), z["freq_values"], atol=2e-6)
        ids = ref_prot.describe_seq_identities(kept, target_seq_index=0)
        np.testing.assert_allclose(ids["identity_to_query"].to_numpy(dtype=float), z["identities_table"], atol=1e-15)
        cov = ref_prot.describe_coverage(kept, "p", 10, [0.5, 0.7, 90])
        assert list(cov.columns) == list(z["coverage_columns"])
        np.testing.assert_allclose(cov.drop(columns=["prefix"]).to_numpy(dtype=float), z["coverage_values"],
                                   atol=2e-6, equal_nan=True)
    finally:
        alignment_accel.uninstall(ref_ali)
    assert ref_ali.map_matrix.__module__.startswith("evcouplings.")      # restored
    assert ref_ali.Alignment.count.__module__.startswith("evcouplings.")


def _oracle_backed_plm(monkeypatch):
    """No GPU in this container: stand the numpy / C oracle in for the library calls the drop-ins make."""
    from evcouplings_amd import plm
    from oracle import meanfield_ref
    from oracle.oracle import Oracle
    o = Oracle("f64")

    def fake_mean_field(msa, q, theta_id=0.8, pseudo_count=0.5, **kw):
        counts = o.reweight(msa, theta_id)
        w = 1.0 / counts
        fi, fij = o.marginals(msa, w, q)
        out = meanfield_ref.mean_field(fi, fij, pseudo_count, want_di=False)
        return dict(weights=w.astype(np.float32), n_eff=float(w.sum()), fi=fi.astype(np.float32),
                    fij=fij.astype(np.float32), hi=out["hi"], jij=out["jij"].astype(np.float32),
                    jij_full=out["jij_full"])

    monkeypatch.setattr(plm, "mean_field", fake_mean_field)
    monkeypatch.setattr(plm, "direct_information",
                        lambda J, f: meanfield_ref.direct_information(np.asarray(J), np.asarray(f)))


def test_reference_mean_field_protocol_runs_on_our_drop_in(ref, golden_dir, monkeypatch, tmp_path):
    """the reference's second inference protocol (couplings/protocol.py:597 mean_field) end to end with
    evcouplings_amd.mean_field installed: model file, raw EC file with MI / DI / CN columns, score table."""
    import evcouplings.couplings.mean_field as ref_mf
    from evcouplings_amd import mean_field as our_mf
    _oracle_backed_plm(monkeypatch)
    cp = ref["cp"]
    prefix = str(tmp_path / "mf" / "job")
    kwargs = dict(prefix=prefix, alignment_file=os.path.join(golden_dir, "hip_fit_L24.a2m"), segments=None,
                  focus_mode=True, focus_sequence="SYN/10-33", theta=0.8, pseudo_count=0.5, alphabet=None,
                  min_sequence_distance=6, ec_score_type="cn", scoring_model="skewnormal", frequencies_file=None)
    our_mf.install(ref_mf)
    try:
        outcfg = cp.run(protocol="mean_field", **kwargs)
    finally:
        our_mf.uninstall(ref_mf)
    assert outcfg["num_sites"] == 24 and outcfg["num_valid_sequences"] == 500 and outcfg["region_start"] == 10
    for key in ("raw_ec_file", "model_file", "ec_file"):
        assert os.path.getsize(outcfg[key]) > 0, key
    import pandas as pd
    raw = pd.read_csv(outcfg["raw_ec_file"], sep=" ", names=["i", "A_i", "j", "A_j", "mi_raw", "mi_apc", "di", "cn"])
    assert len(raw) == 276 and np.isfinite(raw[["mi_raw", "mi_apc", "di", "cn"]].values).all()
    back = ref["CouplingsModel"](outcfg["model_file"])
    assert type(back).__name__ == "MeanFieldCouplingsModel" and back.L == 24


def test_reference_complex_protocol_runs_on_our_backend(ref, gpu_fit, golden_dir, tmp_path):
    """the second caller of infer_plmc (couplings/protocol.py:521 complex): two segments of 12 sites each on the
    MI355X fixture -- segment mapping of the ECs, intra/inter scoring, inter-EC table."""
    fit, calls, z = gpu_fit
    from evcouplings_amd import protocol as hip_protocol
    cp = ref["cp"]
    prefix = str(tmp_path / "complex" / "job")
    # list form of mapping.Segment: segment_id, segment_type, sequence_id, region_start, region_end, positions
    segments = [["A_1", "aa", "SYNA", 10, 21, list(range(10, 22))], ["B_1", "aa", "SYNB", 122, 133, list(range(122, 134))]]
    kwargs = dict(
        prefix=prefix, alignment_file=os.path.join(golden_dir, "hip_fit_L24.a2m"), focus_mode=True,
        focus_sequence="SYN/10-33", segments=segments, theta=0.8, alphabet=None, ignore_gaps=False, iterations=100,
        lambda_h=0.01, lambda_J=0.01, lambda_J_times_Lq=True, lambda_group=None, scale_clusters=None, cpu=None,
        plmc="plmc", reuse_ecs=False, min_sequence_distance=6, frequencies_file=None, scoring_model="skewnormal",
        use_all_ecs_for_scoring=False, save_model=True)
    hip_protocol.install()
    try:
        outcfg = cp.run(protocol="complex", **kwargs)
    finally:
        hip_protocol.uninstall()
    (shape, kw), = calls
    assert shape == (500, 24) and kw["lambda_j"] == pytest.approx(0.01 * 20 * 23)
    import pandas as pd
    ecs = pd.read_csv(outcfg["ec_file"])
    inter = pd.read_csv(outcfg["inter_ec_file"])
    assert len(ecs) == 276 and len(inter) == 144 and set(inter["segment_i"]) == {"A_1"} and set(inter["segment_j"]) == {"B_1"}
    # sites 13..24 of the model were renumbered into the second segment's coordinates
    assert inter["j"].min() == 122 and inter["j"].max() == 133 and inter["i"].max() == 21
    assert "probability" in ecs.columns


def test_model_accel_behind_the_reference_couplings_model(ref, golden_dir, monkeypatch):
    """install_all(): the reference's CouplingsModel methods (hamiltonians, smm, dmm, delta_hamiltonian) run on our
    wrappers and give the numbers of the reference's own loops (plm.* stands on the C oracle here: no GPU)."""
    import evcouplings.couplings.model as ref_model
    from evcouplings_amd import plm, protocol as hip_protocol
    from oracle.oracle import Oracle
    o = Oracle("f64")

    def canon(hi, jij):
        return np.concatenate([np.asarray(hi, np.float64).ravel(), np.asarray(jij, np.float64).ravel()])

    monkeypatch.setattr(plm, "hamiltonians", lambda seqs, q, hi, jij, device=0: o.hamiltonians(seqs, q, canon(hi, jij)))
    monkeypatch.setattr(plm, "single_mutant_matrix",
                        lambda t, q, hi, jij, device=0: o.single_mutants(np.asarray(t).ravel(), q, canon(hi, jij)))
    path = os.path.join(golden_dir, "hip_fit_L24.model")
    slow = ref["CouplingsModel"](path)
    seqs = ["".join(slow.target_seq)] + ["".join(np.roll(slow.target_seq, k)) for k in (1, 5)]
    H_slow, smm_slow, dmm_slow = slow.hamiltonians(seqs), slow.smm(), slow.dmm()
    hip_protocol.install_all()
    try:
        assert ref_model._hamiltonians.__module__ == "evcouplings_amd.model_accel"
        fast = ref["CouplingsModel"](path)
        H_fast, smm_fast, dmm_fast = fast.hamiltonians(seqs), fast.smm(), fast.dmm()
        muts = [(12, slow.target_seq[2], "W"), (20, slow.target_seq[10], "K")]     # positions in index_list numbering
        d_fast = fast.delta_hamiltonian(muts)
    finally:
        hip_protocol.uninstall_all()
    assert ref_model._hamiltonians.__module__ != "evcouplings_amd.model_accel"
    np.testing.assert_allclose(H_fast, H_slow, rtol=1e-6, atol=1e-6)       # float32 parameters either way
    np.testing.assert_allclose(smm_fast, smm_slow, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dmm_fast, dmm_slow, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(d_fast, slow.delta_hamiltonian(muts), rtol=1e-6, atol=1e-6)


def test_mutate_stage_calculations_on_the_energy_drop_ins(ref, golden_dir, monkeypatch):
    """the arithmetic of the reference's mutate stage (evcouplings/mutate/calculations.py: single_mutant_matrix,
    predict_mutation_table) with model_accel installed gives the table the unpatched reference computes."""
    import pandas as pd
    import evcouplings.mutate.calculations as mc
    from evcouplings_amd import model_accel, plm
    from oracle.oracle import Oracle
    o = Oracle("f64")

    def canon(hi, jij):
        return np.concatenate([np.asarray(hi, np.float64).ravel(), np.asarray(jij, np.float64).ravel()])

    monkeypatch.setattr(plm, "hamiltonians", lambda seqs, q, hi, jij, device=0: o.hamiltonians(seqs, q, canon(hi, jij)))
    monkeypatch.setattr(plm, "single_mutant_matrix",
                        lambda t, q, hi, jij, device=0: o.single_mutants(np.asarray(t).ravel(), q, canon(hi, jij)))
    path = os.path.join(golden_dir, "hip_fit_L24.model")
    slow = ref["CouplingsModel"](path)
    t = slow.target_seq
    data = pd.DataFrame({"mutant": ["%s12W" % t[2], "%s20K,%s31D" % (t[10], t[21]), "wild"]})
    want_singles = mc.single_mutant_matrix(slow, output_column="prediction_epistatic")
    want_table = mc.predict_mutation_table(slow, data, "prediction_epistatic")
    model_accel.install()
    try:
        fast = ref["CouplingsModel"](path)
        got_singles = mc.single_mutant_matrix(fast, output_column="prediction_epistatic")
        got_table = mc.predict_mutation_table(fast, data, "prediction_epistatic")
    finally:
        model_accel.uninstall()
    assert list(got_singles.columns) == list(want_singles.columns) and len(got_singles) == len(want_singles) == 24 * 19
    np.testing.assert_allclose(got_singles["prediction_epistatic"].values, want_singles["prediction_epistatic"].values,
                               rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(got_table["prediction_epistatic"].values.astype(float),
                               want_table["prediction_epistatic"].values.astype(float), rtol=1e-6, atol=1e-6,
                               equal_nan=True)
