"""
GPU statistical energies behind the reference's `CouplingsModel`.

`evcouplings/couplings/model.py` computes Hamiltonians with two numba-compiled loops that its
`CouplingsModel.hamiltonians`, `.single_mut_mat_full`, `.smm`, `.dmm` and the mutate stage call:
`_hamiltonians(sequences, J_ij, h_i)` (model.py:25-60) and
`_single_mutant_hamiltonians(target_seq, J_ij, h_i)` (model.py:63-109).  `install()` rebinds those two
module attributes to wrappers around libplm_hip (`plm_hamiltonians`, `plm_potentials`); signatures, dtypes
and return layouts are the reference's.  There is no CPU fallback: without the library the wrappers raise.
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


def install(model_module=None):
    """Rebind the two loops in evcouplings.couplings.model (or the module object given)."""
    if model_module is None:
        import evcouplings.couplings.model as model_module
    if model_module not in _ORIGINAL:
        _ORIGINAL[model_module] = (model_module._hamiltonians, model_module._single_mutant_hamiltonians)
    model_module._hamiltonians = hamiltonians
    model_module._single_mutant_hamiltonians = single_mutant_hamiltonians
    return model_module


def uninstall(model_module=None):
    if model_module is None:
        import evcouplings.couplings.model as model_module
    if model_module in _ORIGINAL:
        model_module._hamiltonians, model_module._single_mutant_hamiltonians = _ORIGINAL.pop(model_module)
