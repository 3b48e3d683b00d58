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
filters bite.  Output: tests/golden/align_stats.npz.  Usage: python tests/golden/make_golden_align.py
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


if __name__ == "__main__":
    main()
