"""
CPU tests of the host layer around the C ABI: alignment reader (focus-mode rules), the two
output formats, the plmc stderr grammar and the plmc-argv shim.  Golden files come from the
reference's own readers/parsers (tests/golden/make_golden.py).
"""
import json
import os

import numpy as np
import pandas as pd
import pytest

from evcouplings_amd import alignment_io, cli, model_io, tools
from evcouplings_amd.synthetic import ALPHABET_PROTEIN, msa_to_a2m, synthetic_msa


def _write(tmp_path, text, name="a.a2m"):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


# ------------------------------------------------------------------ alignment reader
def test_focus_mode_columns_numbering_and_invalid_sequences(tmp_path):
    # focus row: uppercase = model column, lowercase/'.'/'-' = dropped; numbering counts every residue
    path = _write(tmp_path, "\n".join([
        ">FOC/11-20 extra words",
        "ACd.E-FGhiKL",          # residues: A C d E F G h i K L -> 10 residues, region 11-20
        ">s2/1-9",
        "AC-.EWFG--KL",
        ">s3/5-9",
        "XCa.E-FGhiKL",          # X in kept column 0 -> invalid
        ">s4",
        "-Cd.EAFG..K.",          # '.' in kept columns -> gap
    ]) + "\n")
    enc = alignment_io.encode_alignment(path, focus_seq="FOC/11-20")
    assert enc.focus_index == 1 and enc.region_start == 11
    assert enc.target_seq == "ACEFGKL"
    assert enc.columns.tolist() == [0, 1, 4, 6, 7, 10, 11]
    assert enc.index_list.tolist() == [11, 12, 14, 15, 16, 19, 20]   # d, h, i skipped in numbering
    assert enc.n_total_sites == 10 and enc.n_total_seqs == 4 and enc.n_valid_seqs == 3
    assert enc.valid.tolist() == [True, True, False, True]
    A = ALPHABET_PROTEIN
    assert enc.msa.dtype == np.int8 and enc.msa.shape == (3, 7)
    assert enc.msa[0].tolist() == [A.index(c) for c in "ACEFGKL"]
    assert enc.msa[1].tolist() == [A.index(c) for c in "ACEFGKL"]
    assert enc.msa[2].tolist() == [0, A.index("C"), A.index("E"), A.index("F"), A.index("G"), A.index("K"), 0]
    # the name without the /range also selects the focus (tools.py:219)
    assert alignment_io.encode_alignment(path, focus_seq="FOC").focus_index == 1


def test_non_focus_mode_and_errors(tmp_path):
    path = _write(tmp_path, ">a\nACDE\n>b\nAC-E\n>c\nACBE\n")
    enc = alignment_io.encode_alignment(path)
    assert enc.focus_index is None and enc.index_list.tolist() == [1, 2, 3, 4]
    assert enc.n_valid_seqs == 2 and enc.valid.tolist() == [True, True, False]   # 'B' is not in the alphabet
    with pytest.raises(alignment_io.AlignmentFormatError):
        alignment_io.encode_alignment(path, focus_seq="nope")
    ragged = _write(tmp_path, ">a\nACDE\n>b\nACD\n", "r.a2m")
    with pytest.raises(alignment_io.AlignmentFormatError):
        alignment_io.encode_alignment(ragged)
    empty = _write(tmp_path, "", "e.a2m")
    with pytest.raises(alignment_io.AlignmentFormatError):
        alignment_io.encode_alignment(empty)
    dna = _write(tmp_path, ">a\nACGT-\n>b\nAC-TT\n", "d.fa")
    enc = alignment_io.encode_alignment(dna, alphabet="-ACGT")
    assert enc.msa.tolist() == [[1, 2, 3, 4, 0], [1, 2, 0, 4, 4]]


def test_native_alignment_input_equals_the_python_twin(tmp_path, monkeypatch):
    """plm_fasta_split + plm_encode_columns (libplm_hip's host-side input passes, used when the library is built) against
    alignment_io's pure-Python reader on files that exercise every framing rule: wrapped sequences, CRLF line ends,
    blank and whitespace-only lines, leading / trailing blanks, whitespace INSIDE a data line (kept: the record then
    has the wrong length), no final newline, headers with descriptions, '.', lowercase and symbols outside the alphabet,
    data before the first header, an empty file."""
    from evcouplings_amd import _lib
    lib = _lib.load()
    assert hasattr(lib, "plm_fasta_split") and hasattr(lib, "plm_encode_columns")
    rng = np.random.default_rng(5)
    letters = np.frombuffer(b"-ACDEFGHIKLMNPQRSTVWY.acdxBZ", dtype=np.uint8)

    def both(path, **kw):
        out = []
        for mode in ("1", ""):
            if mode:
                monkeypatch.setenv("PLM_IO_PYTHON", mode)
            else:
                monkeypatch.delenv("PLM_IO_PYTHON", raising=False)
            try:
                out.append(alignment_io.encode_alignment(path, **kw))
            except alignment_io.AlignmentFormatError as exc:
                out.append(str(exc))
        return out

    def same(a, b):
        if isinstance(a, str) or isinstance(b, str):
            assert a == b, (a, b)
            return
        for k in ("msa", "valid", "columns", "index_list"):
            np.testing.assert_array_equal(getattr(a, k), getattr(b, k), err_msg=k)
        for k in ("ids", "target_seq", "focus_index", "region_start", "n_total_sites", "n_total_seqs", "n_valid_seqs"):
            assert getattr(a, k) == getattr(b, k), k
        assert a.msa.dtype == b.msa.dtype == np.int8 and a.msa.flags["C_CONTIGUOUS"] and b.msa.flags["C_CONTIGUOUS"]

    for case in range(12):
        n, w = int(rng.integers(1, 40)), int(rng.integers(2, 90))
        rows = letters[rng.integers(0, letters.size if case % 3 else 21, (n, w))]
        rows[0] = letters[rng.integers(1, 21, w)]                       # a clean focus row
        eol = b"\r\n" if case % 2 else b"\n"
        chunks = []
        for r in range(n):
            chunks.append(b">s%d/%d-%d some description" % (r, 5, 4 + w) + eol)
            seq = rows[r].tobytes()
            wrap = int(rng.integers(1, w + 1)) if case % 4 == 1 else w
            for o in range(0, w, wrap):
                lead = b"  " if case % 5 == 2 else b""
                chunks.append(lead + seq[o:o + wrap] + (b" \t" if case % 5 == 3 else b"") + eol)
            if case % 3 == 0:
                chunks.append(b"   " + eol + eol)
        body = b"".join(chunks)
        if case % 4 == 2:
            body = body.rstrip()                                         # no final newline
        path = str(tmp_path / ("case%d.a2m" % case))
        open(path, "wb").write(body)
        same(*both(path, focus_seq="s0"))
        same(*both(path))
    ragged = str(tmp_path / "inner_space.fa")
    open(ragged, "wb").write(b">a\nAC DE\n>b\nACDE\n")              # the blank inside the line belongs to the record
    same(*both(ragged))
    early = str(tmp_path / "early.fa")
    open(early, "wb").write(b"ACDE\n>a\nACDE\n")
    a, b = both(early)
    assert isinstance(a, str) and a == b and "before the first" in a
    empty = str(tmp_path / "empty.fa")
    open(empty, "wb").write(b"\n\n")
    a, b = both(empty)
    assert isinstance(a, str) and a == b


def test_family_like_generator_is_deterministic_and_has_the_advertised_shape(oracle64):
    """synthetic.family_msa (the robustness alignments of tests/test_gpu_parity.py): same seed -> same matrix, states in
    range, row 0 gap-free, a fifth or more of the cells gaps, and strongly clustered -- N_eff well below N at theta = 0.8."""
    from evcouplings_amd.synthetic import family_msa
    a, pa = family_msa(1500, 90, seed=11, depth=4, row_mut=(1.0, 15.0))
    b, pb = family_msa(1500, 90, seed=11, depth=4, row_mut=(1.0, 15.0))
    c, _ = family_msa(1500, 90, seed=12, depth=4, row_mut=(1.0, 15.0))
    assert a.dtype == np.int8 and a.shape == (1500, 90) and np.array_equal(a, b) and pa == pb and not np.array_equal(a, c)
    assert a.min() == 0 and a.max() <= 20 and (a[0] > 0).all()
    assert 0.15 < (a == 0).mean() < 0.45
    counts = oracle64.reweight(a, 0.8)
    assert (1.0 / counts).sum() < 0.6 * a.shape[0] and counts.max() > 20
    assert all(0 <= i < j < 90 and j - i >= 6 for i, j in pa)


def test_synthetic_a2m_round_trip(tmp_path):
    msa, _ = synthetic_msa(50, 30, seed=3)
    path = msa_to_a2m(msa, str(tmp_path / "syn.a2m"), region_start=7)
    enc = alignment_io.encode_alignment(path, focus_seq="SYN/7-36")
    np.testing.assert_array_equal(enc.msa, msa)
    assert enc.index_list.tolist() == list(range(7, 37)) and enc.region_start == 7


# ------------------------------------------------------------------ output formats
def test_model_writer_matches_file_validated_by_reference_reader(tmp_path, golden_dir):
    z = np.load(os.path.join(golden_dir, "scores_L12.npz"))
    L, q, N = 12, 21, 30
    out = str(tmp_path / "m.model")
    model_io.write_model_file(
        out, L=L, q=q, n_valid=N, n_invalid=2, num_iter=100, theta=0.2, lambda_h=0.01, lambda_j=2.2,
        lambda_group=0.0, n_eff=17.25, alphabet=ALPHABET_PROTEIN,
        weights=np.concatenate([z["weights"], np.zeros(2, np.float32)]), target_seq=str(z["target"]),
        index_list=z["index_list"], fi=z["fi"], hi=z["hi"], fij=z["fij"], jij=z["jij"])
    golden = os.path.join(golden_dir, "tiny_L12.model")   # CouplingsModel read this one back exactly
    assert open(out, "rb").read() == open(golden, "rb").read()
    assert os.path.getsize(out) == model_io.model_file_size(L, q, N + 2)
    back = model_io.read_model_file(out)
    np.testing.assert_array_equal(back["jij"], z["jij"])
    np.testing.assert_array_equal(back["index_list"], z["index_list"])
    assert back["alphabet"] == ALPHABET_PROTEIN and back["n_invalid"] == 2
    with pytest.raises(ValueError):
        model_io.write_model_file(out, L=L, q=q, n_valid=N, n_invalid=2, num_iter=1, theta=0.2, lambda_h=-1.0,
                                  lambda_j=1, lambda_group=0, n_eff=1, alphabet=ALPHABET_PROTEIN,
                                  weights=np.zeros(N + 2), target_seq=str(z["target"]), index_list=z["index_list"],
                                  fi=z["fi"], hi=z["hi"], fij=z["fij"], jij=z["jij"])


def test_raw_ec_file_format(tmp_path, golden_dir):
    z = np.load(os.path.join(golden_dir, "scores_L12.npz"))
    out = str(tmp_path / "ecs.txt")
    model_io.write_raw_ec_file(out, z["index_list"], str(z["target"]), z["cn"])
    assert open(out).read() == open(os.path.join(golden_dir, "tiny_L12_ECs.txt")).read()
    first = open(out).readline().split(" ")
    assert len(first) == 6 and first[4] == "0" and len(first[5].strip().split(".")[1]) == 6
    tab = pd.read_csv(out, sep=" ", names=["i", "A_i", "j", "A_j", "fn", "cn"])   # pairs.py:55-58
    assert len(tab) == 66 and (np.diff(tab["i"].values) >= 0).all()
    np.testing.assert_allclose(tab["cn"].values, z["ecs_cn"], atol=5.1e-7)


@pytest.mark.parametrize("name,n_sites", [("PABP_YEAST", 82), ("test_b0.6", 151)])
def test_raw_ec_writer_reproduces_real_plmc_output_byte_for_byte(tmp_path, golden_dir, name, n_sites):
    """The only genuine plmc artefacts the reference holds (notebooks/example/PABP_YEAST_ECs.txt, test_b0.6_ECs.txt; copied
    as data by tests/golden/make_golden.py (9); their alignments are not in the reference).  Parse -> dense CN matrix ->
    write_raw_ec_file must give the file back bit for bit: line order (i ascending, then j), the literal 0 in column 5,
    '%.6f' including negative scores, non-contiguous site numbering (pairs.py:55-58 reads exactly this)."""
    src = os.path.join(golden_dir, "plmc_real_%s_ECs.txt" % name)
    tab = pd.read_csv(src, sep=" ", names=["i", "A_i", "j", "A_j", "fn", "cn"])                  # pairs.py:55-58
    sites = sorted(set(tab["i"]) | set(tab["j"]))
    L = len(sites)
    assert L == n_sites and len(tab) == L * (L - 1) // 2
    pos = {s: k for k, s in enumerate(sites)}
    letter = dict(zip(tab["i"], tab["A_i"]))
    letter.update(zip(tab["j"], tab["A_j"]))
    cn = np.zeros((L, L))
    cn[[pos[i] for i in tab["i"]], [pos[j] for j in tab["j"]]] = tab["cn"].values
    out = str(tmp_path / "ecs.txt")
    model_io.write_raw_ec_file(out, np.array(sites), "".join(letter[s] for s in sites), cn)
    assert open(out, "rb").read() == open(src, "rb").read()
    if name == "test_b0.6":
        assert (tab["cn"] < 0).any() and (np.diff(sites) > 1).any()      # the cases the tiny golden does not have
    # the library's writer (plm_write_raw_ec_file, the default) and its Python twin give the same bytes
    twin = str(tmp_path / "ecs_twin.txt")
    os.environ["PLM_IO_PYTHON"] = "1"
    try:
        model_io.write_raw_ec_file(twin, np.array(sites), "".join(letter[s] for s in sites), cn.astype(np.float32))
    finally:
        del os.environ["PLM_IO_PYTHON"]
    model_io.write_raw_ec_file(out, np.array(sites), "".join(letter[s] for s in sites), cn.astype(np.float32))
    assert open(out, "rb").read() == open(twin, "rb").read()


# ------------------------------------------------------------------ stderr grammar
def test_log_text_is_what_the_reference_parser_accepted(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "plmc_log.json")))
    for name, c in cases.items():
        kw = dict(c["inputs"])
        kw["table"] = [tuple(r) for r in kw["table"]]
        assert tools.format_plmc_log(**kw) == c["log"], name
        df = tools.iteration_dataframe(kw["table"])
        assert list(df.columns) == c["iter_columns"]
        assert df.values.tolist() == c["iter_rows"]      # same string cells parse_plmc_log produced
        import re
        for row in df.values.tolist():
            assert all(re.fullmatch(r"\d+\.\d+", cell) for cell in row[1:]) and re.fullmatch(r"\d+", row[0])


# ------------------------------------------------------------------ plmc argv shim
def test_cli_parses_the_argv_run_plmc_builds():
    # evcouplings/couplings/tools.py:202-262 for a typical monomer job
    argv = ["-c", "out/x_ECs.txt", "-o", "out/x.model", "-f", "RASH_HUMAN", "-m", "100", "-t",
            str(1.0 - 0.8), "-s", "1.0", "-lh", "0.01", "-le", "59.8", "-lg", "0.0", "-n", "2", "ali.a2m"]
    o = cli.parse_argv(argv)
    assert o["couplings_file"] == "out/x_ECs.txt" and o["param_file"] == "out/x.model"
    assert o["focus_seq"] == "RASH_HUMAN" and o["iterations"] == 100 and o["alignment"] == "ali.a2m"
    assert round(1.0 - o["theta_div"], 12) == 0.8 and o["lambda_J"] == 59.8 and not o["ignore_gaps"]
    assert cli.parse_argv(["-c", "e", "-g", "-m", "max", "-n", "max", "a"])["iterations"] == "max"
    for bad in (["-c", "e"], ["a"], ["-c", "e", "-zz", "a"], ["-c", "e", "a", "b"], ["-c"]):
        with pytest.raises(ValueError):
            cli.parse_argv(bad)
    assert cli.main(["-c"]) == 1


def test_run_plmc_hip_error_conventions(tmp_path):
    with pytest.raises(tools.ResourceError):
        tools.run_plmc_hip(str(tmp_path / "missing.a2m"), str(tmp_path / "e.txt"))
    msa, _ = synthetic_msa(20, 12, seed=1)
    ali = msa_to_a2m(msa, str(tmp_path / "s.a2m"))
    with pytest.raises(tools.ExternalToolError):
        tools.run_plmc_hip(ali, str(tmp_path / "e.txt"), lambda_g=-0.5)
    with pytest.raises(tools.ExternalToolError):
        tools.run_plmc_hip(ali, str(tmp_path / "e.txt"), iterations="lots")
    from evcouplings_amd import _lib
    if _lib.load().plm_device_count() <= 0:
        # no GPU here: the solver must fail loudly (as ExternalToolError), never fall back to a CPU path
        with pytest.raises(tools.ExternalToolError):
            tools.run_plmc_hip(ali, str(tmp_path / "e.txt"), str(tmp_path / "m.model"), focus_seq="SYN/1-12")


def test_cli_extensions_and_solver_selection(monkeypatch):
    """Options plmc does not have: --solver / --gpus / --epsilon / --conventions on the shim, PLM_HIP_SOLVER for an
    unmodified pipeline (same pattern as PLM_HIP_CONVENTIONS)."""
    o = cli.parse_argv(["-c", "e", "--solver", "joint", "--gpus", "2", "--epsilon", "1e-4", "--conventions", "0x140", "a"])
    assert o["solver"] == "joint" and o["gpus"] == "2" and o["epsilon"] == 1e-4 and o["conventions"] == 320
    monkeypatch.delenv("PLM_HIP_SOLVER", raising=False)
    assert tools.solver_from_env() == "vp" and tools.solver_from_env("JOINT") == "joint"
    monkeypatch.setenv("PLM_HIP_SOLVER", "joint")
    assert tools.solver_from_env() == "joint" and tools.solver_from_env("vp") == "vp"
    with pytest.raises(ValueError):
        tools.solver_from_env("newton")


def test_gpu_count_is_opt_in(monkeypatch):
    """run_plmc's `cpu` is plmc's thread count: it never starts GPU ranks by itself.  `gpus=` or
    PLM_HIP_GPUS do; "cpu" as the variable's value reads the cpu option."""
    from evcouplings_amd import dist, plm
    monkeypatch.setattr(plm, "device_count", lambda: 8)
    monkeypatch.delenv("PLM_HIP_GPUS", raising=False)
    monkeypatch.delenv("PLM_DIST_BACKEND", raising=False)
    assert dist.resolve_gpu_count(16) == 1 and dist.resolve_gpu_count("max") == 1 and dist.resolve_gpu_count(None) == 1
    assert dist.resolve_gpu_count(16, gpus=4) == 4 and dist.resolve_gpu_count(None, gpus="max") == 8
    assert dist.resolve_gpu_count(None, gpus=64) == 8                      # capped by the visible devices
    monkeypatch.setenv("PLM_HIP_GPUS", "2")
    assert dist.resolve_gpu_count(16) == 2 and dist.resolve_gpu_count(16, gpus=1) == 1
    monkeypatch.setenv("PLM_HIP_GPUS", "cpu")
    assert dist.resolve_gpu_count(4) == 4 and dist.resolve_gpu_count(None) == 1 and dist.resolve_gpu_count("max") == 8
    monkeypatch.setenv("PLM_HIP_GPUS", "many")
    with pytest.raises(ValueError):
        dist.resolve_gpu_count(1)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run.
    PLM_BENCH_LAUNCH_ONLY makes every rank join a gloo group and rank 0 report: no GPU needed."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PLM_BENCH_LAUNCH_ONLY="1")
    env.pop("WORLD_SIZE", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stderr[-2000:]
    line = [ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line) == {"launch_only": True, "n_gpus": 2, "steps": 3, "warmup": 1}


def test_pin_kit_against_a_stand_in_plmc(tmp_path):
    """scripts/pin_against_plmc.py with tests/fake_plmc.py (the CPU oracle behind plmc's command line) as the binary:
    the A2M export, the argv in run_plmc's order, the golden files and their record.  The HIP side of the script (the
    sweep over the convention switches) runs in the GPU suite."""
    import importlib.util
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pin_against_plmc", os.path.join(root, "scripts", "pin_against_plmc.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    fake = os.path.join(root, "tests", "fake_plmc.py")
    rep = pin.main(["--plmc", fake, "--small", "--no-hip", "--iterations", "60", "--out", str(tmp_path / "g"),
                    "--work", str(tmp_path / "w"), "--cpu", "2"])
    assert [r["golden"] for r in rep] == ["plmc_small_ECs.txt", "plmc_small_g_ECs.txt"]
    for r in rep:
        ecs = pin.read_cn(str(tmp_path / "g" / r["golden"]))
        L = r["L"]
        assert len(ecs) == L * (L - 1) // 2 and all(np.isfinite(v) for v in ecs.values())
        rec = json.load(open(str(tmp_path / "g" / ("plmc_small%s.json" % ("_g" if r["ignore_gaps"] else "")))))
        argv = rec["argv"]
        # the order run_plmc assembles (evcouplings/couplings/tools.py:202-262)
        keys = [a for a in argv if a.startswith("-") and not a[1:2].isdigit()]
        want = ["-c", "-o", "-f"] + (["-g"] if r["ignore_gaps"] else []) + ["-m", "-t", "-s", "-lh", "-le", "-lg", "-n"]
        assert keys == want, keys
        assert argv[argv.index("-f") + 1] == "SYN" and argv[-1].endswith("small.a2m")
        assert argv[argv.index("-t") + 1] == str(1.0 - 0.8)                       # the 1 - theta round trip, as upstream
        assert "valid sequences out of" in rec["plmc_stderr_tail"]
    # with and without -g the stand-in solved different models
    a, b = pin.read_cn(str(tmp_path / "g" / "plmc_small_ECs.txt")), pin.read_cn(str(tmp_path / "g" / "plmc_small_g_ECs.txt"))
    assert max(abs(a[k] - b[k]) for k in a) > 1e-3


def test_map_matrix_drop_in_handles_empty_cells():
    """np.vectorize hands the alphabet map '' for an empty cell of a U1 / S1 matrix; the lookup-table drop-in does too."""
    from collections import defaultdict
    from evcouplings_amd import alignment_accel
    amap = defaultdict(lambda: 0, {c: k for k, c in enumerate("-ACDE")})
    amap[""] = 7
    m = np.array([["A", "", "C"], ["-", "E", ""]], dtype="U1")
    np.testing.assert_array_equal(alignment_accel.map_matrix(m, amap), np.vectorize(amap.__getitem__)(m))
    with pytest.raises(ValueError):
        alignment_accel._int8_states(np.array([[0, 200]]), "matrix")


def _example_a2m(z, tmp_path):
    """the reference's example alignment as an A2M file, written from the character matrix the fixture holds"""
    path = str(tmp_path / "example_aln.a2m")
    with open(path, "w") as f:
        for name, row in zip(z["ids"].tolist(), z["chars_full"]):
            f.write(">%s\n%s\n" % (name, row.tobytes().decode("ascii")))
    return path


def test_real_alignment_through_the_host_layer_matches_the_reference(golden_dir, oracle64, tmp_path):
    """The one alignment the reference ships (notebooks/example/example_aln.a2m: 53 cadherin sequences, 423 columns, three
    of them inserts -- lowercase in the first sequence, '.' elsewhere --, real gap runs) through the A2M reader and
    the oracle, against what the reference's own Alignment class made of it (tests/golden/make_golden_align.py)."""
    z = np.load(os.path.join(golden_dir, "example_aln.npz"))
    a2m = _example_a2m(z, tmp_path)
    enc = alignment_io.encode_alignment(a2m, focus_seq="Q641K6_MOUSE")
    assert enc.msa.shape == (53, 420) and enc.n_total_sites == 423 and enc.n_valid_seqs == enc.n_total_seqs == 53
    np.testing.assert_array_equal(enc.msa, z["mapped"])                       # same columns kept, same encoding
    # index_list numbers the focus residues: the three insert positions leave gaps in the numbering
    kept_positions = 1 + np.flatnonzero(z["keep_cols"])
    np.testing.assert_array_equal(enc.index_list, kept_positions)
    assert enc.target_seq == "".join(z["freq_target"].tolist())
    counts = oracle64.reweight(enc.msa, 0.8)
    np.testing.assert_array_equal(counts, z["counts"])
    w = 1.0 / counts
    np.testing.assert_allclose(w, z["weights"], rtol=1e-14)
    fi, fij = oracle64.marginals(enc.msa, w, 21)
    np.testing.assert_allclose(fi, z["fi"], atol=1e-13)
    np.testing.assert_allclose(fij, z["fij_pairs"], atol=2e-7)                # stored as float32


def test_multi_gpu_launch_errors_stay_inside_the_error_conventions(tmp_path, monkeypatch):
    """Without a GPU the ranks of a multi-GPU job die at once: dist.launch_fit must report that as LaunchError (with the
    ranks' stderr), and the run_plmc drop-in -- whose single-GPU fallback cannot work here either -- must surface an
    ExternalToolError, as for every other solver failure; nothing may fall back to a CPU computation."""
    from evcouplings_amd import _lib, dist
    if _lib.load().plm_device_count() > 0:
        pytest.skip("needs a host without a GPU")
    msa, _ = synthetic_msa(40, 16, seed=2)
    monkeypatch.setenv("PLM_DIST_BACKEND", "gloo")
    with pytest.raises(dist.LaunchError) as err:
        dist.launch_fit(msa, 2, q=21, max_iter=3, timeout=600)
    assert "multi-GPU fit failed" in str(err.value)
    ali = msa_to_a2m(msa, str(tmp_path / "s.a2m"))
    with pytest.warns(UserWarning, match="running on one GPU"):
        with pytest.raises(tools.ExternalToolError):
            tools.run_plmc_hip(ali, str(tmp_path / "e.txt"), focus_seq="SYN/1-16", iterations=3, gpus=2)


# ---- L-BFGS two-loop recursion in coefficient space (plm_lbfgs_coefficients, the code plm_ctx_optimize runs) --------
def _two_loop_dense(g, pairs, dinv=None):
    """textbook two-loop recursion on explicit vectors; pairs = [(s, y)] oldest -> newest; H0 = gamma * diag(dinv)"""
    dinv = np.ones_like(g) if dinv is None else dinv
    q = g.copy()
    alphas = []
    for s, y in reversed(pairs):
        a = s.dot(q) / s.dot(y)
        alphas.append(a)
        q -= a * y
    if pairs:
        s, y = pairs[-1]
        q = q * dinv * (s.dot(y) / (y * dinv).dot(y))
    else:
        q = q * dinv
    for (s, y), a in zip(pairs, reversed(alphas)):
        b = y.dot(q) / s.dot(y)
        q += (a - b) * s
    return -q


@pytest.mark.parametrize("m,stored,end,precond", [(6, 0, 0, False), (6, 3, 3, False), (6, 6, 2, False), (6, 5, 2, False),
                                                  (6, 5, 5, True), (6, 5, 0, False), (4, 3, 1, True), (6, 2, 2, False)])
def test_lbfgs_coefficients_follow_the_ring(m, stored, end, precond):
    """after a noise-dominated pair is skipped on a full ring the live pairs are every slot except
    `end`, not the physical slots 0..stored-1 -- the direction must be the L-BFGS direction of exactly the live pairs.
    Case (6, 5, 2): ring full, pair in slot 2 skipped."""
    import ctypes as C
    from evcouplings_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(m * 100 + stored * 10 + end)
    n = 40
    A = rng.normal(size=(n, n))
    H = A @ A.T + n * np.eye(n)                     # SPD: s.y > 0 for every pair
    S = rng.normal(size=(m, n))
    Y = S @ H
    g = rng.normal(size=n)
    dinv = 1.0 / np.diag(H) if precond else np.ones(n)
    SY = S @ Y.T
    YDY = (Y * dinv) @ Y.T
    Sg, YDg, gDg = S @ g, (Y * dinv) @ g, (g * dinv).dot(g)
    live_new_to_old = [(end - 1 - i) % m for i in range(stored)]
    dead = [j for j in range(m) if j not in live_new_to_old]
    for j in dead:                                  # dead slots hold garbage that must never be read
        SY[j, :] = SY[:, j] = YDY[j, :] = YDY[:, j] = np.nan
        Sg[j] = YDg[j] = np.nan
    cs, cy = np.full(m, 7.0), np.full(m, 7.0)
    cg, dg = C.c_double(0), C.c_double(0)
    SYc, YDYc, Sgc, YDgc = [np.ascontiguousarray(a, dtype=np.float64) for a in (SY, YDY, Sg, YDg)]
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.plm_lbfgs_coefficients(m, stored, end, p(SYc), p(YDYc), p(Sgc), p(YDgc), float(gDg), p(cs), p(cy),
                               C.byref(cg), C.byref(dg))
    assert np.all(cs[dead] == 0) and np.all(cy[dead] == 0)
    live = live_new_to_old
    direction = dinv * (cg.value * g) + sum(cs[j] * S[j] + dinv * (cy[j] * Y[j]) for j in live)
    ref = _two_loop_dense(g, [(S[j], Y[j]) for j in reversed(live)], dinv if precond else None)
    np.testing.assert_allclose(direction, ref, rtol=1e-9, atol=1e-12)
    assert dg.value == pytest.approx(g.dot(ref), rel=1e-9)
    assert dg.value < 0


# ---- cancellation: an exception / a signal handler's SystemExit inside the iteration callback --------------------
def test_iteration_callback_turns_exceptions_and_signals_into_a_cancellation():
    """evcouplings/utils/pipeline.py:476-545 installs SIGTERM / SIGINT handlers that call sys.exit().  During a fit the
    only Python code the main thread runs is the iteration callback, so the handler's SystemExit is raised there; ctypes
    would swallow it.  The wrapper must hand the library a non-zero return (cancel) and re-raise afterwards."""
    import signal
    from evcouplings_amd import plm

    def handler(signum, frame):
        raise SystemExit(1)

    old = signal.signal(signal.SIGTERM, handler)
    try:
        seen = []
        icb = plm._IterationCallback(lambda it, *rest: (seen.append(it), signal.raise_signal(signal.SIGTERM) if it == 2 else None))
        assert icb.cfunc(1, 0.1, 0.5, 10.0, 9.0, 1.0, 2.0, None) == 0
        assert icb.pending is None
        assert icb.cfunc(2, 0.2, 0.4, 9.0, 8.0, 1.0, 2.0, None) == 1          # -> PLM_STATUS_INTERRUPTED in the library
        assert isinstance(icb.pending, SystemExit) and seen == [1, 2] and len(icb.table) == 2
        with pytest.raises(SystemExit):
            icb.reraise()
        icb.reraise()                                                          # raised once
        icb2 = plm._IterationCallback(lambda *a: 1 / 0)
        assert icb2.cfunc(1, 0.1, 0.5, 10.0, 9.0, 1.0, 2.0, None) == 1
        with pytest.raises(ZeroDivisionError):
            icb2.reraise()
    finally:
        signal.signal(signal.SIGTERM, old)
