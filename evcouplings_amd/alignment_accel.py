"""
GPU alignment statistics behind the reference's `Alignment` class (SURVEY.md section 8f, row N3 -- the part of
the upstream preprocessing that is arithmetic).

`evcouplings/align/alignment.py` computes sequence weights and frequencies with three numba loops that
`Alignment.set_weights`, `.frequencies`, `.pair_frequencies` (and through them the align stage's
`describe_frequencies`, the mean-field protocol and the compare stage) call:
    num_cluster_members(matrix, identity_threshold)      alignment.py:1193-1233   O(N^2 L)
    frequencies(matrix, seq_weights, num_symbols)        alignment.py:1079-1106
    pair_frequencies(matrix, seq_weights, num_symbols, fi)  alignment.py:1110-1153  O(N L^2)
plus `identities_to_seq(seq, matrix)` (:1157-1190, identity of every sequence to the query) and the
`np.vectorize` encoder `map_matrix` (:479-495).  `install()` rebinds them to wrappers around the kernels the solver
uses (`plm_reweight`, `plm_marginals`) and `plm_alignment_stats`, and the method `Alignment.count` (:707-747) to
`alignment_count`, which is how `modify_alignment`'s two coverage filters (align/protocol.py:900-943) and
`describe_coverage` reach the GPU with the reference code unchanged; `alignment_filters` is the same arithmetic as one
call on an integer matrix:
same arguments, same return shapes and dtypes (float64; the arithmetic is float32).  Any non-negative weights are
accepted (rescaled by a power of two into the library's range); alphabets of up to 21 symbols.  No CPU fallback.
"""
import numpy as np

_ORIGINAL = {}


def _gpu_weights(seq_weights):
    """float32 weights for the library.  Frequencies are ratios of weighted sums, so a power-of-two rescale changes
    nothing -- the reference accepts any non-negative weights (align/alignment.py:1078-1153) and so does this; large
    weights are brought into (0.5, 1] so that their float32 sum N_eff keeps its precision."""
    w = np.asarray(seq_weights, dtype=np.float64)
    top = float(w.max()) if w.size else 0.0
    if top >= 2.0:
        w = w * 2.0 ** -np.ceil(np.log2(top))      # now in (0.5, 1]
    return w.astype(np.float32)


def _int8_states(a, what):
    """Integer states as the int8 the library takes; values outside 0..127 would wrap silently in the cast (and the
    packed byte compares assume bytes < 0x80), so they are refused here."""
    a = np.asarray(a)
    if a.size and (a.min() < 0 or a.max() > 127):
        raise ValueError("%s holds states outside 0..127 (min %d, max %d)" % (what, a.min(), a.max()))
    return np.ascontiguousarray(a, dtype=np.int8)


def num_cluster_members(matrix, identity_threshold):
    """Drop-in for alignment.num_cluster_members: length-N float64 vector of cluster sizes (self included)."""
    from evcouplings_amd import plm
    return plm.reweight(_int8_states(matrix, "matrix"), float(identity_threshold)).astype(np.float64)


def frequencies(matrix, seq_weights, num_symbols):
    """Drop-in for alignment.frequencies: L x num_symbols float64 (the arithmetic is float32 on the GPU, ~1e-7
    relative; alphabet sizes up to 21, larger ones raise PlmError -- there is no CPU path)."""
    from evcouplings_amd import plm
    fi = plm.marginals(_int8_states(matrix, "matrix"), _gpu_weights(seq_weights), int(num_symbols), pairs=False)
    fi = fi[0] if isinstance(fi, tuple) else fi
    return fi.astype(np.float64)


def pair_frequencies(matrix, seq_weights, num_symbols, fi):
    """Drop-in for alignment.pair_frequencies: dense L x L x q x q float64, f_ij[j,i] = f_ij[i,j].T and
    f_ii = diag(f_i) (the given fi is used for the diagonal blocks, like the reference)."""
    from evcouplings_amd import plm
    from evcouplings_amd.mean_field import dense_pair_frequencies
    _, fij = plm.marginals(_int8_states(matrix, "matrix"), _gpu_weights(seq_weights), int(num_symbols), pairs=True)
    return dense_pair_frequencies(np.asarray(fi, dtype=np.float64), fij.astype(np.float64))


def identities_to_seq(seq, matrix):
    """Drop-in for alignment.identities_to_seq (alignment.py:1157-1190): number of positions at which every sequence
    of the mapped matrix equals the mapped `seq`, length-N float64 (the numba twin returns np.zeros((N,)) floats)."""
    from evcouplings_amd import plm
    _, _, ident = plm.alignment_stats(_int8_states(matrix, "matrix"), 0, query=_int8_states(seq, "seq"))
    return ident.astype(np.float64)


def map_matrix(matrix, map_):
    """Drop-in for alignment.map_matrix (alignment.py:479-495): the reference maps every element through a Python
    dict with np.vectorize -- one interpreter call per residue, minutes at N = 100 000.  Same result (an integer array
    of the same shape; unknown symbols get the defaultdict's default) through a code-point lookup table.  Host-side
    format conversion, no arithmetic: it runs in numpy, not on the GPU."""
    m = np.asarray(matrix)
    if m.dtype.kind == "S":
        codes = m.view(np.uint8).reshape(m.shape) if m.dtype.itemsize == 1 else None
    elif m.dtype.kind == "U" and m.dtype.itemsize == 4:
        codes = np.ascontiguousarray(m).view(np.uint32).reshape(m.shape)
    else:
        codes = None
    if codes is None or m.size == 0:
        return np.vectorize(map_.__getitem__)(matrix)          # exotic dtypes: exactly what the reference does
    top = int(codes.max()) + 1
    if top > 0x110000:
        return np.vectorize(map_.__getitem__)(matrix)
    present = np.unique(codes)
    lut = np.zeros(top, dtype=np.int64)
    for c in present.tolist():
        # code point 0 is numpy's padding of an EMPTY cell: np.vectorize hands the dict '' (b'') for it
        key = (chr(c) if c else "") if m.dtype.kind == "U" else (bytes([c]) if c else b"")
        lut[c] = map_[key]                                    # defaultdict: unknown symbols -> its default, as upstream
    return lut[codes]


def _char_codes(matrix):
    """Character matrix (dtype S1 / U1) -> its code points as int8, or None when that is not possible (other dtypes,
    code points above 127).  The alignment kernels only compare bytes, so they work on ASCII codes as well as on
    mapped states."""
    m = np.asarray(matrix)
    if m.ndim != 2 or m.size == 0:
        return None
    if m.dtype.kind == "S" and m.dtype.itemsize == 1:
        codes = np.ascontiguousarray(m).view(np.uint8).reshape(m.shape)
    elif m.dtype.kind == "U" and m.dtype.itemsize == 4:
        codes = np.ascontiguousarray(m).view(np.uint32).reshape(m.shape)
    else:
        return None
    if int(codes.max()) > 127:
        return None
    return codes.astype(np.int8)


def alignment_count(self, char, axis="pos", normalize=True):
    """Drop-in for the method `Alignment.count` (align/alignment.py:707-747): occurrences of `char` per column
    (axis "pos") or per sequence (axis "seq"), relative to the axis length unless normalize=False.  It is what
    `modify_alignment` filters fragments and gappy columns with (align/protocol.py:906-912, 941) and what
    `describe_coverage` tabulates; here the counts come from `plm_alignment_stats` (k_align_rows / k_align_cols) run
    on the ASCII codes of the character matrix.  Same return dtype as upstream (int64 counts, float64 fractions)."""
    if axis not in ("pos", "seq"):
        raise ValueError("Invalid axis: {}".format(axis))
    codes = _char_codes(self.matrix)
    ch = char.decode("latin-1") if isinstance(char, bytes) else char
    if codes is None or not isinstance(ch, str) or len(ch) != 1 or ord(ch) > 126:
        # exotic input (other dtypes, code points above 127): the upstream arithmetic (align/alignment.py:742-746)
        naxis = 0 if axis == "pos" else 1
        c = np.sum(self.matrix == char, axis=naxis)
        return c / self.matrix.shape[naxis] if normalize else c
    from evcouplings_amd import plm
    seq_counts, col_counts, _ = plm.alignment_stats(codes, ord(ch))
    c = (col_counts if axis == "pos" else seq_counts).astype(np.int64)
    if normalize:
        c = c / self.matrix.shape[0 if axis == "pos" else 1]
    return c


_ORIGINAL_COUNT = {}


def alignment_filters(matrix_mapped, gap_state=0, minimum_sequence_coverage=None, minimum_column_coverage=None):
    """The two coverage filters of modify_alignment (align/protocol.py:900-914, 935-943) on the integer matrix:
    keep_seqs = sequences whose non-gap fraction is >= minimum_sequence_coverage, and -- computed on the kept
    sequences -- lc_cols = columns whose gap fraction exceeds 1 - minimum_column_coverage (the reference lower-cases
    them).  Integers are read as percentages, as upstream.  Either threshold may be None (no filter)."""
    from evcouplings_amd import plm
    m = _int8_states(matrix_mapped, "matrix")
    n, L = m.shape
    keep = np.ones(n, dtype=bool)
    if minimum_sequence_coverage is not None:
        cov = minimum_sequence_coverage / 100 if isinstance(minimum_sequence_coverage, int) else minimum_sequence_coverage
        seq_gaps, _, _ = plm.alignment_stats(m, gap_state)
        keep = (1 - seq_gaps / L) >= cov
    lc_cols = None
    if minimum_column_coverage is not None:
        cov = minimum_column_coverage / 100 if isinstance(minimum_column_coverage, int) else minimum_column_coverage
        kept = m[keep] if not keep.all() else m
        _, col_gaps, _ = plm.alignment_stats(kept, gap_state)
        lc_cols = col_gaps / kept.shape[0] > 1 - cov
    return keep, lc_cols


_NAMES = ("num_cluster_members", "frequencies", "pair_frequencies", "identities_to_seq", "map_matrix")


def install(alignment_module=None):
    """Rebind the five functions in evcouplings.align.alignment (or the module given).  With them the arithmetic of
    the align stage's modify_alignment / describe_frequencies / describe_seq_identities / describe_coverage
    (align/protocol.py:463-640, 806-1016) -- weights, frequencies, identities to the query, the encoding of the
    character matrix -- runs through this package; the reference code around it is unchanged."""
    if alignment_module is None:
        import evcouplings.align.alignment as alignment_module
    if alignment_module not in _ORIGINAL:
        _ORIGINAL[alignment_module] = tuple(getattr(alignment_module, n) for n in _NAMES)
    alignment_module.num_cluster_members = num_cluster_members
    alignment_module.frequencies = frequencies
    alignment_module.pair_frequencies = pair_frequencies
    alignment_module.identities_to_seq = identities_to_seq
    alignment_module.map_matrix = map_matrix
    # the method behind modify_alignment's coverage filters and describe_coverage
    cls = getattr(alignment_module, "Alignment", None)
    if cls is not None and cls not in _ORIGINAL_COUNT:
        _ORIGINAL_COUNT[cls] = cls.count
        cls.count = alignment_count
    return alignment_module


def uninstall(alignment_module=None):
    if alignment_module is None:
        import evcouplings.align.alignment as alignment_module
    if alignment_module in _ORIGINAL:
        for name, fn in zip(_NAMES, _ORIGINAL.pop(alignment_module)):
            setattr(alignment_module, name, fn)
    cls = getattr(alignment_module, "Alignment", None)
    if cls in _ORIGINAL_COUNT:
        cls.count = _ORIGINAL_COUNT.pop(cls)
