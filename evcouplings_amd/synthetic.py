"""
Synthetic protein-like alignments for benchmarks and parity tests.

The reference ships no alignment/solver-output pair, so every
benchmark and parity case runs on alignments drawn by this generator, defined in
SURVEY.md section 8(d): ancestors drawn from column-specific Dirichlet profiles, rows
copied from a random ancestor and mutated at a per-row rate, terminal + internal gaps,
and planted pair couplings.  Row 0 is the gap-free focus sequence ``SYN/1-L``.

States follow the reference's default protein alphabet ``-ACDEFGHIKLMNPQRSTVWY``
(evcouplings/align/alignment.py:25-26), gap = 0.
"""
import numpy as np

ALPHABET_PROTEIN = "-ACDEFGHIKLMNPQRSTVWY"
BASE_SEED = 20260921


def synthetic_msa(n_seqs, n_sites, seed=BASE_SEED, q=21, n_pairs=None, couple_prob=0.7):
    """Return an int8 (n_seqs, n_sites) matrix with values in 0..q-1 (0 = gap)."""
    rng = np.random.default_rng(seed)
    N, L, A = int(n_seqs), int(n_sites), q - 1
    K = max(16, N // 64)
    profiles = rng.dirichlet(0.3 * np.ones(A), size=L)          # (L, A)
    cdf = np.cumsum(profiles, axis=1)
    cdf[:, -1] = 1.0

    def draw(shape_rows):
        u = rng.random((shape_rows, L))
        return 1 + (u[:, :, None] > cdf[None, :, :]).sum(axis=2).astype(np.int8)

    ancestors = draw(K)                                          # (K, L) states 1..A
    msa = np.empty((N, L), dtype=np.int8)
    which = rng.integers(0, K, size=N)
    mu = rng.beta(2.0, 3.0, size=N)
    chunk = 4096
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        base = ancestors[which[lo:hi]]
        resample = rng.random((hi - lo, L)) < mu[lo:hi, None]
        fresh = draw(hi - lo)
        msa[lo:hi] = np.where(resample, fresh, base)

    # planted couplings: disjoint pairs (i, j), |i-j| >= 6, x_sj = perm(x_si) with prob couple_prob
    if n_pairs is None:
        n_pairs = L // 2
    order = rng.permutation(L)
    used = np.zeros(L, dtype=bool)
    planted = []
    for i in order:
        if len(planted) >= n_pairs or used[i]:
            continue
        cand = [j for j in order if not used[j] and abs(int(j) - int(i)) >= 6 and j != i]
        if not cand:
            continue
        j = cand[0]
        used[i] = used[j] = True
        planted.append((int(min(i, j)), int(max(i, j))))
    for (i, j) in planted:
        perm = np.concatenate([[0], 1 + rng.permutation(A)]).astype(np.int8)
        force = rng.random(N) < couple_prob
        msa[force, j] = perm[msa[force, i]]

    # gaps: N-/C-terminal runs ~ Geom(0.05) capped at L/4, plus 2 % internal gaps
    cap = max(1, L // 4)
    nterm = np.minimum(rng.geometric(0.05, size=N) - 1, cap)
    cterm = np.minimum(rng.geometric(0.05, size=N) - 1, cap)
    cols = np.arange(L)[None, :]
    gap = (cols < nterm[:, None]) | (cols >= (L - cterm)[:, None]) | (rng.random((N, L)) < 0.02)
    msa[gap] = 0
    # row 0: gap-free focus sequence
    msa[0] = ancestors[0]
    return msa, planted


def msa_to_a2m(msa, path, alphabet=ALPHABET_PROTEIN, focus_id="SYN", region_start=1):
    """Write the matrix as an A2M/FASTA file whose first record is ``SYN/start-end``."""
    letters = np.frombuffer(alphabet.encode("ascii"), dtype=np.uint8)
    N, L = msa.shape
    with open(path, "w") as f:
        for s in range(N):
            name = "%s/%d-%d" % (focus_id, region_start, region_start + L - 1) if s == 0 else "seq%d/1-%d" % (s, L)
            f.write(">%s\n%s\n" % (name, letters[msa[s]].tobytes().decode("ascii")))
    return path
