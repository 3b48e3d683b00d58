#!/usr/bin/env python3
"""
Generate golden vectors from the REFERENCE's own Python code (run in the build
container only; /root/reference does not exist on the GPU box).

What is pinned here, and by which reference code:
  * reweighting counts        <- evcouplings/align/alignment.py:1193-1233 num_cluster_members
  * single / pair frequencies <- evcouplings/align/alignment.py:1079-1153
  * `.model` plmc_v2 reader   <- evcouplings/couplings/model.py:317-389 (CouplingsModel)
  * FN / CN (APC) scores      <- evcouplings/couplings/model.py:179-233, 744-827
  * raw EC file reader        <- evcouplings/couplings/pairs.py:34-65
  * mean-field DCA            <- evcouplings/couplings/mean_field.py:717-1014 (regularisation, covariance,
                                 reshape, fields, direct_information) on frequencies of golden alignments
  * raw EC file WRITER        <- notebooks/example/PABP_YEAST_ECs.txt, test_b0.6_ECs.txt (real plmc output, copied as data)
  * statistical energies      <- evcouplings/couplings/model.py:25-109 (_hamiltonians,
                                 _single_mutant_hamiltonians) and the CouplingsModel methods on top

numba is not installed, so the reference's @jit kernels run as plain Python under an
identity `numba.jit` stub.  `num_cluster_members` rebinds L to a
float and then calls range(L) (alignment.py:1216,1225) -- legal under numba, a TypeError
in CPython -- so it is executed from an in-memory copy with that one call patched to
range(int(L)).  No reference source is written into this repo.

Usage:  python tests/golden/make_golden.py      (writes tests/golden/*.npz, *.model ...)
"""
import importlib.util
import inspect
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("EVC_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)


def install_stubs():
    nb = types.ModuleType("numba")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.jit = jit
    nb.njit = jit
    nb.prange = range
    sys.modules["numba"] = nb


def load_reference_module(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reference_alignment_kernels():
    """frequencies / pair_frequencies / num_cluster_members without importing the whole
    evcouplings package (its __init__ chain needs ruamel/billiard/...)."""
    path = os.path.join(REF, "evcouplings/align/alignment.py")
    src = open(path).read()
    ns = {"np": np, "jit": sys.modules["numba"].jit}
    out = {}
    import ast
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in (
                "frequencies", "pair_frequencies", "num_cluster_members"):
            text = ast.get_source_segment(src, node)
            # strip decorator line(s)
            text = "\n".join(l for l in text.split("\n") if not l.strip().startswith("@jit"))
            if node.name == "num_cluster_members":
                assert "for k in range(L):" in text
                text = text.replace("for k in range(L):", "for k in range(int(L)):")
            exec(compile(text, path, "exec"), ns)
            out[node.name] = ns[node.name]
    assert len(out) == 3
    return out


def main():
    install_stubs()
    from evcouplings_amd.synthetic import synthetic_msa, ALPHABET_PROTEIN
    from evcouplings_amd import model_io

    kern = reference_alignment_kernels()
    model_mod = load_reference_module("evcouplings/couplings/model.py", "ref_model")
    q = 21

    # ---- (1) reweighting + frequencies on small alignments, incl. threshold edge cases
    cases = {}
    for name, (N, L, theta, seed) in {
        "a": (96, 20, 0.8, 1), "b": (150, 37, 0.8, 2), "c": (64, 25, 0.6, 3), "d": (80, 10, 0.3, 4),
    }.items():
        msa, _ = synthetic_msa(N, L, seed=seed)
        # craft rows that sit exactly on / one off the identity threshold
        T = int(np.ceil(theta * L - 1e-9))
        base = msa[1].copy()
        for k, nid in enumerate((T - 1, T, T + 1)):
            nid = max(0, min(L, nid))
            row = base.copy()
            mism = np.arange(L - nid)
            row[mism] = (row[mism] % 20) + 1  # force a different non-gap state
            assert (row == base).sum() == nid
            msa[2 + k] = row
        counts = kern["num_cluster_members"](msa.astype(np.int64), theta)
        w = 1.0 / counts
        fi = kern["frequencies"](msa.astype(np.int64), w, q)
        fij = kern["pair_frequencies"](msa.astype(np.int64), w, q, fi)
        cases[name] = dict(msa=msa, theta=theta, counts=counts.astype(np.int32), fi=fi, fij=fij)
    np.savez_compressed(
        os.path.join(HERE, "reweight_freqs.npz"),
        **{"%s_%s" % (k, f): v for k, d in cases.items() for f, v in d.items()})

    # ---- (2) scoring + file formats: random parameters -> our writer -> reference reader
    rng = np.random.default_rng(7)
    L, N = 12, 30
    npair = L * (L - 1) // 2
    hi = rng.normal(size=(L, q)).astype(np.float32)
    jij = (0.3 * rng.normal(size=(npair, q, q))).astype(np.float32)
    fi = rng.dirichlet(np.ones(q), size=L).astype(np.float32)
    fij = rng.dirichlet(np.ones(q * q), size=npair).reshape(npair, q, q).astype(np.float32)
    weights = rng.random(N).astype(np.float32)
    target = "".join(ALPHABET_PROTEIN[1 + k % 20] for k in range(L))
    index_list = np.arange(5, 5 + L, dtype=np.int32)
    index_list[6:] += 3  # numbering gap, as lowercase-skipped columns would leave
    model_path = os.path.join(HERE, "tiny_L12.model")
    model_io.write_model_file(
        model_path, L=L, q=q, n_valid=N, n_invalid=2, num_iter=100, theta=0.2, lambda_h=0.01,
        lambda_j=2.2, lambda_group=0.0, n_eff=17.25, alphabet=ALPHABET_PROTEIN,
        weights=np.concatenate([weights, np.zeros(2, np.float32)]), target_seq=target,
        index_list=index_list, fi=fi, hi=hi, fij=fij, jij=jij)
    m = model_mod.CouplingsModel(model_path)
    assert m.L == L and m.num_symbols == q and m.N_valid == N and m.N_invalid == 2
    iu = np.triu_indices(L, 1)
    np.testing.assert_array_equal(m.J_ij[iu].astype(np.float32), jij)
    np.testing.assert_array_equal(m.J_ij[iu[1], iu[0]].astype(np.float32), jij.transpose(0, 2, 1))
    np.testing.assert_array_equal(m.h_i.astype(np.float32), hi)
    np.testing.assert_array_equal(m.index_list, index_list)
    assert "".join(m.target_seq) == target
    ecs = m.ecs.sort_values(["i", "j"])
    np.savez_compressed(
        os.path.join(HERE, "scores_L12.npz"), hi=hi, jij=jij, fi=fi, fij=fij, weights=weights,
        index_list=index_list, target=np.array(target), fn=m.fn_scores, cn=m.cn_scores,
        ecs_i=ecs["i"].values, ecs_j=ecs["j"].values, ecs_cn=ecs["cn"].values,
        ecs_fn=ecs["fn"].values, theta=np.float32(0.2), lambda_h=np.float32(0.01),
        lambda_j=np.float32(2.2), n_eff=np.float32(17.25))

    # ---- (3) raw EC file: our writer -> reference reader semantics (pairs.py:55-58 is a
    # plain pandas read_csv with these column names; pairs.py itself needs sklearn/scipy
    # and the evcouplings package chain, so the call is restated on the spot)
    import pandas as pd
    ec_path = os.path.join(HERE, "tiny_L12_ECs.txt")
    model_io.write_raw_ec_file(ec_path, index_list, target, m.cn_scores)
    tab = pd.read_csv(ec_path, sep=" ", names=["i", "A_i", "j", "A_j", "fn", "cn"])
    assert len(tab) == npair and (tab["fn"] == 0).all()
    np.testing.assert_allclose(tab.sort_values(["i", "j"])["cn"].values, ecs["cn"].values, atol=5.1e-7)
    # ---- (4) stderr grammar: our log text -> the reference's parse_plmc_log (tools.py:20-108)
    import ast
    import json
    import re as _re
    tools_path = os.path.join(REF, "evcouplings/couplings/tools.py")
    tsrc = open(tools_path).read()
    fn = [n for n in ast.parse(tsrc).body if isinstance(n, ast.FunctionDef) and n.name == "parse_plmc_log"][0]
    ns = {"re": _re, "pd": pd}
    exec(compile(ast.get_source_segment(tsrc, fn), tools_path, "exec"), ns)
    from evcouplings_amd import tools as our_tools
    table = [(1, 0.0123, 2984.1234567, 31645564.4534, 31645153.2858, 187.448, 1.0),
             (2, 0.0381, 2857.0, 29529698.8806, 29525634.5811, 187.429, 7.88),
             (3, 1.5, 0.000912, 13235552.0132, 10736296.9835, 185.93, 204.421)]
    cases = {
        "focus": dict(focus_name="SYN", focus_index=1, n_valid=49990, n_total=50000, n_sites=300,
                      n_total_sites=312, region_start=17, n_eff=12345.678,
                      status_msg="converged (|g|/max(1,|x|) below epsilon)", table=table),
        "nofocus": dict(focus_name=None, focus_index=None, n_valid=10, n_total=12, n_sites=20,
                        n_total_sites=20, region_start=1, n_eff=7.0,
                        status_msg="maximum number of iterations reached", table=table[:1]),
    }
    out = {}
    for name, kw in cases.items():
        text = our_tools.format_plmc_log(**kw)
        iter_df, fields = ns["parse_plmc_log"](text)
        out[name] = dict(inputs={k: v for k, v in kw.items()}, log=text, parsed_fields=list(fields),
                         iter_columns=list(iter_df.columns), iter_rows=iter_df.values.tolist())
        assert fields[1] == kw["n_valid"] and fields[2] == kw["n_total"]
        assert fields[6] == float("%.1f" % kw["n_eff"]) and fields[7] == kw["status_msg"]
        assert len(iter_df) == len(kw["table"]) and list(iter_df.columns) == our_tools.ITER_COLUMNS
        if kw["focus_index"] is not None:
            assert fields[0] == kw["focus_index"] and fields[3:6] == (kw["n_sites"], kw["n_total_sites"], kw["region_start"])
        else:
            assert fields[0] is None and fields[3] is None and fields[5] == 1
    with open(os.path.join(HERE, "plmc_log.json"), "w") as f:
        json.dump(out, f, indent=1)

    # ---- (5) scaling convention of the objective: the reference states the field
    # part itself in CouplingsModel.to_independent_model (model.py:894-910):  N_eff (logZ - f.x) + lambda_h |x|^2.
    # Its optimum for the frequencies of golden alignment "a" must be a stationary point of our objective
    # at J = 0 (tests/test_oracle.py, tests/test_gpu_parity.py).
    za = np.load(os.path.join(HERE, "reweight_freqs.npz"))
    ca = {f: za["a_" + f] for f in ("msa", "counts", "fi", "fij")}
    La = ca["msa"].shape[1]
    wa = 1.0 / ca["counts"]
    iu2 = np.triu_indices(La, 1)
    ind_path = os.path.join(HERE, "_tmp_indep.model")
    model_io.write_model_file(
        ind_path, L=La, q=q, n_valid=len(wa), n_invalid=0, num_iter=1, theta=0.2, lambda_h=0.01, lambda_j=1.0,
        lambda_group=0.0, n_eff=float(wa.sum()), alphabet=ALPHABET_PROTEIN, weights=wa.astype(np.float32),
        target_seq="A" * La, index_list=np.arange(1, La + 1), fi=ca["fi"], hi=np.zeros((La, q)),
        fij=ca["fij"][iu2], jij=np.zeros((La * (La - 1) // 2, q, q)))
    mi = model_mod.CouplingsModel(ind_path).to_independent_model()
    os.remove(ind_path)
    np.savez_compressed(os.path.join(HERE, "independent_model_a.npz"), h_ref=mi.h_i, lambda_h=0.01,
                        n_eff=np.float32(wa.sum()).astype(np.float64), fi32=ca["fi"].astype(np.float32))
    # ---- (6) statistical energies: the reference's own loops on the tiny model of (2)
    rng = np.random.default_rng(11)
    seqs = rng.integers(0, q, size=(40, L)).astype(np.int64)
    seqs[0] = [ALPHABET_PROTEIN.index(c) for c in target]
    # h_i is a float32 array in the reference object; compiled by numba its sums are double (float64 +
    # float32 -> float64), but under the identity-jit stub NumPy 2 keeps `0.0 + float32` in float32.  Passing
    # the same values as float64 reproduces the compiled semantics (J_ij already is float64).
    h64 = m.h_i.astype(np.float64)
    H = model_mod._hamiltonians(seqs, m.J_ij, h64)
    smm = model_mod._single_mutant_hamiltonians(seqs[0], m.J_ij, h64)
    # the public methods built on them must give the same numbers (up to that float32 artefact)
    np.testing.assert_allclose(m.hamiltonians(["".join(ALPHABET_PROTEIN[k] for k in row) for row in seqs]), H,
                               rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(m.smm(), smm[:, :, 0], rtol=1e-6, atol=1e-6)
    np.savez_compressed(os.path.join(HERE, "energies_L12.npz"), seqs=seqs.astype(np.int8), hamiltonians=H,
                        single_mutants=smm)
    # ---- (7) mean-field DCA: the reference's own functions (couplings/mean_field.py) on the
    # frequencies of golden alignments "a" (L=20) and "d" (L=10, theta 0.3); the package imports need the stub
    # modules of tests/refstubs.py
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refstubs
    refstubs.install()
    import evcouplings.couplings.mean_field as mf
    for name, pc in (("a", 0.5), ("d", 0.2)):
        fi_r = za[name + "_fi"].astype(np.float64)
        fij_r = za[name + "_fij"].astype(np.float64)          # dense L x L x q x q from pair_frequencies
        Lm = fi_r.shape[0]
        rfi = mf.regularize_frequencies(fi_r, pseudo_count=pc)
        rfij = mf.regularize_pair_frequencies(fij_r, pseudo_count=pc)
        cov = mf.compute_covariance_matrix(rfi, rfij)
        inv = -np.linalg.inv(cov)
        J4 = mf.reshape_invC_to_4d(inv, Lm, q)
        h_mf = mf.fields(J4, rfi)
        di = mf.direct_information(J4, rfi)
        iu3 = np.triu_indices(Lm, 1)
        np.savez_compressed(os.path.join(HERE, "meanfield_%s.npz" % name), pseudo_count=pc, fi=fi_r,
                            fij_pairs=fij_r[iu3], rfi=rfi, cov=cov, jij_full=J4, hi=h_mf, di=di)
    # ---- (9) the only REAL plmc artefacts the reference holds: two raw EC files (data, not source; their input
    # alignments are not in the reference -- .MISSING_LARGE_BLOBS:1-2).  Kept as byte-exact pins of the a9 writer
    # (pairs.py:55-58): non-contiguous numbering and negative scores in the second one.
    import shutil
    for fn in ("PABP_YEAST_ECs.txt", "test_b0.6_ECs.txt"):
        shutil.copyfile(os.path.join(REF, "notebooks/example", fn), os.path.join(HERE, "plmc_real_" + fn))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
