"""
GPU alignment statistics behind the reference's `Alignment` class (SURVEY.md section 8f, row N3 -- the part of
the upstream preprocessing that is arithmetic).

`evcouplings/align/alignment.py` computes sequence weights and frequencies with three numba loops that
`Alignment.set_weights`, `.frequencies`, `.pair_frequencies` (and through them the align stage's
`describe_frequencies`, the mean-field protocol and the compare stage) call:
    num_cluster_members(matrix, identity_threshold)      alignment.py:1193-1233   O(N^2 L)
    frequencies(matrix, seq_weights, num_symbols)        alignment.py:1079-1106
    pair_frequencies(matrix, seq_weights, num_symbols, fi)  alignment.py:1110-1153  O(N L^2)
`install()` rebinds them to wrappers around the same kernels the solver uses (`plm_reweight`, `plm_marginals`):
same arguments, same return shapes and dtypes (float64; the arithmetic is float32).  Any non-negative weights are
accepted (rescaled by a power of two into the library's range); alphabets of up to 21 symbols.  No CPU fallback.
"""
import numpy as np

_ORIGINAL = {}


def _gpu_weights(seq_weights):
    """float32 weights in the range the library's f16 residual split accepts (< 4).  Frequencies are ratios of
    weighted sums, so a power-of-two rescale changes nothing -- the reference accepts any non-negative weights
    (align/alignment.py:1078-1153) and so does this."""
    w = np.asarray(seq_weights, dtype=np.float64)
    top = float(w.max()) if w.size else 0.0
    if top >= 2.0:
        w = w * 2.0 ** -np.ceil(np.log2(top))      # now in (0.5, 1]
    return w.astype(np.float32)


def num_cluster_members(matrix, identity_threshold):
    """Drop-in for alignment.num_cluster_members: length-N float64 vector of cluster sizes (self included)."""
    from evcouplings_amd import plm
    return plm.reweight(np.asarray(matrix).astype(np.int8), float(identity_threshold)).astype(np.float64)


def frequencies(matrix, seq_weights, num_symbols):
    """Drop-in for alignment.frequencies: L x num_symbols float64 (the arithmetic is float32 on the GPU, ~1e-7
    relative; alphabet sizes up to 21, larger ones raise PlmError -- there is no CPU path)."""
    from evcouplings_amd import plm
    fi = plm.marginals(np.asarray(matrix).astype(np.int8), _gpu_weights(seq_weights), int(num_symbols), pairs=False)
    fi = fi[0] if isinstance(fi, tuple) else fi
    return fi.astype(np.float64)


def pair_frequencies(matrix, seq_weights, num_symbols, fi):
    """Drop-in for alignment.pair_frequencies: dense L x L x q x q float64, f_ij[j,i] = f_ij[i,j].T and
    f_ii = diag(f_i) (the given fi is used for the diagonal blocks, like the reference)."""
    from evcouplings_amd import plm
    from evcouplings_amd.mean_field import dense_pair_frequencies
    _, fij = plm.marginals(np.asarray(matrix).astype(np.int8), _gpu_weights(seq_weights), int(num_symbols), pairs=True)
    return dense_pair_frequencies(np.asarray(fi, dtype=np.float64), fij.astype(np.float64))


def install(alignment_module=None):
    if alignment_module is None:
        import evcouplings.align.alignment as alignment_module
    if alignment_module not in _ORIGINAL:
        _ORIGINAL[alignment_module] = tuple(getattr(alignment_module, n) for n in
                                            ("num_cluster_members", "frequencies", "pair_frequencies"))
    alignment_module.num_cluster_members = num_cluster_members
    alignment_module.frequencies = frequencies
    alignment_module.pair_frequencies = pair_frequencies
    return alignment_module


def uninstall(alignment_module=None):
    if alignment_module is None:
        import evcouplings.align.alignment as alignment_module
    if alignment_module in _ORIGINAL:
        (alignment_module.num_cluster_members, alignment_module.frequencies,
         alignment_module.pair_frequencies) = _ORIGINAL.pop(alignment_module)
