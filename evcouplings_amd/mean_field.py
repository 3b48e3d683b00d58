"""
GPU mean-field DCA behind the reference's `MeanFieldDCA`.

The reference's second inference protocol (`evcouplings/couplings/protocol.py:597 mean_field`) builds
`MeanFieldDCA(alignment)` and calls `.fit(theta, pseudo_count)` (couplings/mean_field.py:163-222): sequence
weights, frequencies, pseudo-count regularisation, the L(q-1) x L(q-1) covariance matrix, its inverse, fields;
the returned `MeanFieldCouplingsModel` later computes direct information for every pair
(`direct_information`, :842-893).  `install()` rebinds `MeanFieldDCA.fit` and the module-level
`direct_information` to the functions below, which run the arithmetic in libplm_hip (`plm_meanfield`,
`plm_direct_information`) and hand the reference's own classes exactly the objects its code would have built.
No CPU fallback: without the library they raise.
"""
import numpy as np

_ORIGINAL = {}


def dense_pair_frequencies(fi, fij_pairs):
    """i<j blocks -> the dense symmetric L x L x q x q array of `Alignment.pair_frequencies`
    (align/alignment.py:1110-1153: f_ii = diag(f_i))."""
    L, q = fi.shape
    f = np.zeros((L, L, q, q))
    iu, ju = np.triu_indices(L, 1)
    f[iu, ju] = fij_pairs
    f[ju, iu] = np.transpose(fij_pairs, (0, 2, 1))
    idx = np.arange(L)
    f[idx, idx] = np.einsum("ia,ab->iab", fi, np.eye(q))
    return f


def fit(self, theta=0.8, pseudo_count=0.5):
    """Drop-in for MeanFieldDCA.fit (mean_field.py:163-222); `self` is the reference's MeanFieldDCA object."""
    import evcouplings.couplings.mean_field as ref_mf
    from evcouplings_amd import plm
    self._reset()
    ali = self.alignment
    q = int(ali.num_symbols)
    if ali.matrix_mapped is None:      # the reference maps lazily (alignment.py:890-897)
        from evcouplings.align.alignment import map_matrix
        ali.matrix_mapped = map_matrix(ali.matrix, ali.alphabet_map)
    out = plm.mean_field(np.asarray(ali.matrix_mapped).astype(np.int8), q, theta_id=theta,
                         pseudo_count=pseudo_count, want_di=False)
    # what alignment.set_weights / .frequencies / .pair_frequencies would have left behind
    w = out["weights"].astype(np.float64)
    ali.num_cluster_members = np.rint(1.0 / w)
    ali.weights = 1.0 / ali.num_cluster_members
    ali._frequencies = out["fi"].astype(np.float64)
    ali._pair_frequencies = dense_pair_frequencies(ali._frequencies, out["fij"].astype(np.float64))
    # regularised frequencies through the reference's own (cheap) functions, for bit-identical attributes
    self.regularized_frequencies = ref_mf.regularize_frequencies(ali._frequencies, pseudo_count=pseudo_count)
    self.regularized_pair_frequencies = ref_mf.regularize_pair_frequencies(ali._pair_frequencies,
                                                                           pseudo_count=pseudo_count)
    # the two documented attributes the reference's fit leaves behind (mean_field.py:196-205): the covariance matrix of
    # the regularised frequencies and the NEGATIVE of its inverse, both (L (q-1))^2 with index i * (q-1) + alpha
    # (_flatten_index, :23).  Built from what the GPU returned, vectorised; reshape_invC_to_4d() / fields() work.
    L, qm = ali.L, q - 1
    rf, rff = self.regularized_frequencies, self.regularized_pair_frequencies
    self.covariance_matrix = (
        rff[:, :, :qm, :qm] - rf[:, None, :qm, None] * rf[None, :, None, :qm]
    ).transpose(0, 2, 1, 3).reshape(L * qm, L * qm)
    self.covariance_matrix_inv = np.ascontiguousarray(
        out["jij_full"][:, :, :qm, :qm].transpose(0, 2, 1, 3)).reshape(L * qm, L * qm)
    return ref_mf.MeanFieldCouplingsModel(
        alignment=ali, index_list=self.index_list, regularized_f_i=self.regularized_frequencies,
        regularized_f_ij=self.regularized_pair_frequencies, h_i=out["hi"], J_ij=out["jij_full"], theta=theta,
        pseudo_count=pseudo_count)


def direct_information(J_ij, f_i):
    """Drop-in for mean_field.direct_information(J_ij, f_i) (:842-893): L x L float64."""
    from evcouplings_amd import plm
    return plm.direct_information(J_ij, f_i)


def install(mf_module=None):
    """Rebind MeanFieldDCA.fit and direct_information in evcouplings.couplings.mean_field (or the module given)."""
    if mf_module is None:
        import evcouplings.couplings.mean_field as mf_module
    if mf_module not in _ORIGINAL:
        _ORIGINAL[mf_module] = (mf_module.MeanFieldDCA.fit, mf_module.direct_information)
    mf_module.MeanFieldDCA.fit = fit
    mf_module.direct_information = direct_information
    return mf_module


def uninstall(mf_module=None):
    if mf_module is None:
        import evcouplings.couplings.mean_field as mf_module
    if mf_module in _ORIGINAL:
        mf_module.MeanFieldDCA.fit, mf_module.direct_information = _ORIGINAL.pop(mf_module)
