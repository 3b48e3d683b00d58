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


def family_msa(n_seqs, n_sites, seed=BASE_SEED, q=21, depth=6, conserved=0.15, gap_frac=0.2, dup_frac=0.02,
               row_mut=(1.0, 6.0)):
    """A harder, protein-family-like alignment (robustness tests; `synthetic_msa` is the benign benchmark shape):
      * ancestors on a binary tree of `depth` levels below a root, each child a mutated copy of its parent (clades at
        30-60 % identity to each other), rows copied from a leaf with a small per-row mutation rate (Beta(*row_mut),
        default mean 0.14): most rows have many neighbours above 80 % identity, N_eff is a small fraction of N;
      * `conserved` of the columns nearly invariant (Dirichlet 0.02), the rest variable (Dirichlet 0.5);
      * gaps: long terminal runs and internal indel RUNS (not single sites) up to `gap_frac` of all cells;
      * exact duplicates (`dup_frac` of the rows), row 0 gap-free;
      * planted pair couplings as in `synthetic_msa`.
    Returns (msa int8 [n_seqs, n_sites], planted pairs)."""
    rng = np.random.default_rng(seed)
    N, L, A = int(n_seqs), int(n_sites), q - 1
    alpha = np.where(rng.random(L) < conserved, 0.02, 0.5)
    profiles = np.stack([rng.dirichlet(a * np.ones(A)) for a in alpha])
    cdf = np.cumsum(profiles, axis=1)
    cdf[:, -1] = 1.0

    def draw(rows):
        u = rng.random((rows, L))
        return 1 + (u[:, :, None] > cdf[None, :, :]).sum(axis=2).astype(np.int8)

    level = draw(1)
    for _ in range(depth):                                   # each node gets two children, 12 % of the sites redrawn
        kids = np.repeat(level, 2, axis=0)
        fresh = draw(kids.shape[0])
        level = np.where(rng.random(kids.shape) < 0.12, fresh, kids)
    leaves = level                                           # 2^depth leaves
    sizes = rng.dirichlet(0.5 * np.ones(leaves.shape[0]))    # uneven clades: a few big ones, many small
    which = rng.choice(leaves.shape[0], size=N, p=sizes)
    mu = rng.beta(row_mut[0], row_mut[1], size=N)
    msa = np.empty((N, L), dtype=np.int8)
    for lo in range(0, N, 4096):
        hi = min(N, lo + 4096)
        fresh = draw(hi - lo)
        msa[lo:hi] = np.where(rng.random((hi - lo, L)) < mu[lo:hi, None], fresh, leaves[which[lo:hi]])
    planted = []
    order = rng.permutation(L)
    used = np.zeros(L, dtype=bool)
    for i in order:
        if len(planted) >= L // 3 or used[i]:
            continue
        cand = [j for j in order if not used[j] and abs(int(j) - int(i)) >= 6]
        if not cand:
            break
        j = cand[0]
        used[i] = used[j] = True
        planted.append((int(min(i, j)), int(max(i, j))))
        perm = np.concatenate([[0], 1 + rng.permutation(A)]).astype(np.int8)
        force = rng.random(N) < 0.6
        msa[force, j] = perm[msa[force, i]]
    # gaps: terminal runs (geometric, mean L/8) and internal indel runs shared by a clade with probability 1/2
    cols = np.arange(L)[None, :]
    nterm = np.minimum(rng.geometric(8.0 / L, size=N) - 1, L // 3)
    cterm = np.minimum(rng.geometric(8.0 / L, size=N) - 1, L // 3)
    gap = (cols < nterm[:, None]) | (cols >= (L - cterm)[:, None])
    n_runs = max(1, int(gap_frac * L / 12))
    for _ in range(n_runs):
        start = rng.integers(0, L, size=N)
        length = rng.geometric(1.0 / 6.0, size=N)
        on = rng.random(N) < 0.5
        gap |= on[:, None] & (cols >= start[:, None]) & (cols < (start + length)[:, None])
    msa[gap] = 0
    ndup = int(dup_frac * N)
    if ndup:
        msa[rng.integers(1, N, size=ndup)] = msa[rng.integers(1, N, size=ndup)]
    msa[0] = leaves[0]
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
