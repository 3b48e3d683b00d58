#!/usr/bin/env python3
"""
Golden vectors for the align-stage statistics, produced by the REFERENCE's own code run
in the build container (tests/refstubs.py makes the package importable without numba / ruamel / bokeh):

  * Alignment.count(gap, axis="seq" | "pos")          evcouplings/align/alignment.py:707-747
  * Alignment.identities_to(target)                    :994-1016 -> identities_to_seq :1157-1190
  * map_matrix(matrix, alphabet_map)                   :479-495  (np.vectorize over a defaultdict)
  * describe_frequencies / describe_seq_identities / describe_coverage     evcouplings/align/protocol.py:463-640
  * the two coverage filters of modify_alignment       :900-914, 935-943

on tests/golden/hip_fit_L24.a2m with a few sequences turned into fragments and two columns made gappy, so that both
filters bite.  Output: tests/golden/align_stats.npz.

Real data: the one alignment the reference ships, notebooks/example/example_aln.a2m (53 cadherin
sequences x 423 columns, 3 insert columns in lowercase / '.', real gap runs), is read where it lies (its character
matrix and ids go into the fixture; the tests write an A2M file from them) and run through the reference's own
Alignment class the way the couplings stage sees it: match columns
of the first sequence (couplings/mean_field.py:103-109), sequence weights at 80 % identity, single-site and pair
frequencies, the describe_frequencies table.  Output: tests/golden/example_aln.npz.

Usage: python tests/golden/make_golden_align.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))                       # tests/ (refstubs)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))      # repository root (oracle)
import refstubs  # noqa: E402


def golden_alignment():
    """character matrix + ids: the fixture alignment with fragments and gappy columns"""
    from evcouplings.align.alignment import Alignment
    with open(os.path.join(HERE, "hip_fit_L24.a2m")) as f:
        ali = Alignment.from_file(f, "fasta")
    m = np.copy(ali.matrix)
    rng = np.random.default_rng(4)
    for s in rng.choice(np.arange(1, ali.N), size=40, replace=False):
        m[s, rng.integers(4, 12):] = "-"                      # fragments
    m[rng.random(ali.N) < 0.6, 7] = "-"                       # columns with many gaps
    m[rng.random(ali.N) < 0.45, 19] = "-"
    m[0] = ali.matrix[0]
    return Alignment(m, np.copy(ali.ids), alphabet=ali.alphabet)


def main():
    refstubs.install()
    import evcouplings.align.alignment as ra
    import evcouplings.align.protocol as rp
    from oracle.oracle import Oracle
    # num_cluster_members calls range(L) with a float L (alignment.py:1216, 1225): fine under numba, a TypeError in
    # CPython -- the oracle's counts stand in, pinned equal to it by reweight_freqs.npz
    ra.num_cluster_members = lambda matrix, thr: Oracle("f64").reweight(np.asarray(matrix).astype(np.int8), thr).astype(float)
    ali = golden_alignment()
    mapped = ra.map_matrix(ali.matrix, ali.alphabet_map)
    # symbols outside the alphabet map to the gap state (alignment.py:446-476): pinned on a separate little matrix (the
    # reference's count() works on characters, ours on states, so unknown symbols are kept out of the filter goldens)
    odd = np.array([list("AC-XB"), list("x.zYW")])
    out = dict(chars=ali.matrix.astype("S1"), mapped=mapped.astype(np.int8), alphabet=ali.alphabet,
               odd_chars=odd.astype("S1"), odd_mapped=ra.map_matrix(odd, ali.alphabet_map).astype(np.int8),
               seq_gap_frac=ali.count("-", axis="seq"), col_gap_frac=ali.count("-", axis="pos"),
               ident_to_target=ali.identities_to(ali[0]), ident_counts=ali.identities_to(ali[0], normalize=False))
    min_seq, min_col = 50, 0.7
    keep = (1 - ali.count("-", axis="seq")) >= min_seq / 100
    kept = ali.select(sequences=keep)
    lc = kept.count(kept._match_gap, axis="pos") > 1 - min_col
    out.update(min_seq=min_seq, min_col=min_col, keep_seqs=keep, lc_cols=lc)
    kept.set_weights(0.8)
    freq = rp.describe_frequencies(kept, 10, target_seq_index=0)
    out["freq_columns"] = np.array(list(freq.columns))
    out["freq_values"] = freq.drop(columns=["A_i"]).to_numpy(dtype=float)
    out["freq_target"] = np.array(list(freq["A_i"]))
    ids = rp.describe_seq_identities(kept, target_seq_index=0)
    out["identities_table"] = ids["identity_to_query"].to_numpy(dtype=float)
    cov = rp.describe_coverage(kept, "p", 10, [0.5, 0.7, 90])
    out["coverage_columns"] = np.array(list(cov.columns))
    out["coverage_values"] = cov.drop(columns=["prefix"]).to_numpy(dtype=float)
    out["weights"] = kept.weights
    np.savez_compressed(os.path.join(HERE, "align_stats.npz"), **out)
    print("wrote align_stats.npz:", {k: getattr(v, "shape", v) for k, v in out.items()})


def real_data():
    import evcouplings.align.alignment as ra
    import evcouplings.align.protocol as rp
    from oracle.oracle import Oracle
    src = os.path.join(refstubs.REFERENCE, "notebooks", "example", "example_aln.a2m")
    with open(src) as f:
        ali = ra.Alignment.from_file(f, "fasta")
    focus = ali.matrix[0]
    keep_cols = np.array([c.isupper() for c in focus])             # uppercase = match state, not a gap, not an insert
    sel = ali.select(columns=keep_cols)
    mapped = ra.map_matrix(sel.matrix, sel.alphabet_map)
    counts = Oracle("f64").reweight(mapped.astype(np.int8), 0.8)    # stand-in for num_cluster_members (App. D-9)
    ra.num_cluster_members = lambda matrix, thr: Oracle("f64").reweight(np.asarray(matrix).astype(np.int8), thr).astype(float)
    sel.set_weights(0.8)
    fi = sel.frequencies
    fij = sel.pair_frequencies                                      # dense L x L x q x q
    L = sel.L
    iu, ju = np.triu_indices(L, 1)
    freq = rp.describe_frequencies(sel, 1, target_seq_index=0)
    out = dict(ids=np.array(list(ali.ids)), chars_full=ali.matrix.astype("S1"), keep_cols=keep_cols, mapped=mapped.astype(np.int8), counts=counts.astype(np.int32),
               weights=sel.weights, fi=fi, fij_pairs=fij[iu, ju].astype(np.float32), n_eff=float(sel.weights.sum()),
               freq_columns=np.array(list(freq.columns)), freq_values=freq.drop(columns=["A_i"]).to_numpy(dtype=float),
               freq_target=np.array(list(freq["A_i"])), seq_gap_frac=sel.count("-", axis="seq"),
               col_gap_frac=sel.count("-", axis="pos"), ident_to_target=sel.identities_to(sel[0]))
    np.savez_compressed(os.path.join(HERE, "example_aln.npz"), **out)
    print("wrote example_aln.npz: N=%d L=%d (of %d columns), N_eff %.2f" % (sel.N, L, ali.L, out["n_eff"]))


if __name__ == "__main__":
    main()
    real_data()
