"""
GPU statistical energies behind the reference's `CouplingsModel`.

`evcouplings/couplings/model.py` computes Hamiltonians with two numba-compiled loops that its
`CouplingsModel.hamiltonians`, `.single_mut_mat_full`, `.smm`, `.dmm` and the mutate stage call:
`_hamiltonians(sequences, J_ij, h_i)` (model.py:25-60) and
`_single_mutant_hamiltonians(target_seq, J_ij, h_i)` (model.py:63-109).  `install()` rebinds those two
module attributes to wrappers around libplm_hip (`plm_hamiltonians`, `plm_potentials`); signatures, dtypes
and return layouts are the reference's.  There is no CPU fallback: without the library the wrappers raise.
`install(reader=True)` also swaps the class's plmc_v2 reader (L^2 tiny `np.fromfile` calls, model.py:364-389)
for a block-wise one -- host-side format code, same attributes; measured to be no faster (see read_plmc_v2).
"""
import numpy as np

_ORIGINAL = {}


def _pairs_from_dense(J_ij):
    """dense L x L x q x q (J[j,i] = J[i,j].T) -> the i<j blocks the C ABI takes."""
    L = J_ij.shape[0]
    iu = np.triu_indices(L, 1)
    return np.ascontiguousarray(J_ij[iu], dtype=np.float32)


def hamiltonians(sequences, J_ij, h_i):
    """Drop-in for model._hamiltonians: N x 3 float64 (total, couplings, fields)."""
    from evcouplings_amd import plm
    sequences = np.asarray(sequences)
    L, q = h_i.shape
    return plm.hamiltonians(sequences.astype(np.int8), q, h_i, _pairs_from_dense(J_ij))


def single_mutant_hamiltonians(target_seq, J_ij, h_i):
    """Drop-in for model._single_mutant_hamiltonians: L x q x 3 float64."""
    from evcouplings_amd import plm
    L, q = h_i.shape
    return plm.single_mutant_matrix(np.asarray(target_seq).astype(np.int8), q, h_i, _pairs_from_dense(J_ij))


def read_plmc_v2(self, f, precision):
    """
    Drop-in for `CouplingsModel.__read_plmc_v2` (model.py:317-400): same attributes, dtypes and layouts,
    but the L(L-1)/2 pair blocks are read as two contiguous arrays and copied row by row into the dense
    symmetric L x L x q x q float64 arrays instead of L^2 `np.fromfile` calls.  Measured against the
    reference reader: 2.9 s vs 2.8 s at L = 300, 9.8 s vs 10.6 s at L = 600 -- the time goes into filling the
    two dense float64 arrays (2.5 GB at L = 600) with transposed blocks, not into the small reads, so this is
    NOT installed by default (`install(reader=True)` opts in).
    """
    self.L, self.num_symbols, self.N_valid, self.N_invalid, self.num_iter = np.fromfile(f, "int32", 5)
    self.theta, self.lambda_h, self.lambda_J, self.lambda_group, self.N_eff = np.fromfile(f, precision, 5)
    self.alphabet = np.fromfile(f, "S1", self.num_symbols).astype("U1")
    self.weights = np.fromfile(f, precision, self.N_valid + self.N_invalid)
    self._target_seq = np.fromfile(f, "S1", self.L).astype("U1")
    self.index_list = np.fromfile(f, "int32", self.L)
    L, q = int(self.L), int(self.num_symbols)
    self.f_i = np.fromfile(f, precision, L * q).reshape(L, q)
    self.h_i = np.fromfile(f, precision, L * q).reshape(L, q)
    n_pairs = L * (L - 1) // 2
    for name in ("f_ij", "J_ij"):
        blocks = np.fromfile(f, precision, n_pairs * q * q)
        if blocks.size != n_pairs * q * q:
            raise ValueError("truncated plmc_v2 file: %s blocks incomplete" % name)
        blocks = blocks.reshape(n_pairs, q, q)
        dense = np.zeros((L, L, q, q))
        off = 0
        for i in range(L - 1):          # the file's pair order is row-major i<j: row i is one contiguous run
            run = blocks[off:off + L - 1 - i]
            dense[i, i + 1:] = run
            dense[i + 1:, i] = run.transpose(0, 2, 1)
            off += L - 1 - i
        setattr(self, name, dense)
    if self.lambda_h < 0:                # mean-field marker, as in the reference reader (model.py:393-400)
        from evcouplings.couplings.mean_field import MeanFieldCouplingsModel
        self.__class__ = MeanFieldCouplingsModel
        self.transform_from_plmc_model()


def install(model_module=None, reader=False):
    """
    Rebind, in evcouplings.couplings.model (or the module object given): the two Hamiltonian loops and,
    if the module has a CouplingsModel class and `reader` is true, its plmc_v2 reader.
    """
    if model_module is None:
        import evcouplings.couplings.model as model_module
    cls = getattr(model_module, "CouplingsModel", None)
    if model_module not in _ORIGINAL:
        _ORIGINAL[model_module] = (model_module._hamiltonians, model_module._single_mutant_hamiltonians,
                                   getattr(cls, "_CouplingsModel__read_plmc_v2", None))
    model_module._hamiltonians = hamiltonians
    model_module._single_mutant_hamiltonians = single_mutant_hamiltonians
    if reader and cls is not None:
        setattr(cls, "_CouplingsModel__read_plmc_v2", read_plmc_v2)
    return model_module


def uninstall(model_module=None):
    if model_module is None:
        import evcouplings.couplings.model as model_module
    if model_module in _ORIGINAL:
        ham, smm, rd = _ORIGINAL.pop(model_module)
        model_module._hamiltonians, model_module._single_mutant_hamiltonians = ham, smm
        cls = getattr(model_module, "CouplingsModel", None)
        if cls is not None and rd is not None:
            setattr(cls, "_CouplingsModel__read_plmc_v2", rd)
