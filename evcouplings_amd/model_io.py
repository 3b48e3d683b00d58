"""
Writers (and a fast reader) for the two files the couplings stage hands to the rest of
EVcouplings: the raw EC text file and the binary ``.model`` parameter file.

Contracts followed:
  * ``.model`` "plmc_v2" layout  -- reader evcouplings/couplings/model.py:317-389,
    writer evcouplings/couplings/model.py:1200-1252
  * raw EC file                   -- reader evcouplings/couplings/pairs.py:55-58, sample
    notebooks/example/PABP_YEAST_ECs.txt:1-5 (``i A_i j A_j 0 cn``, 6 decimals, unsorted,
    i ascending then j)
"""
import numpy as np


def n_pairs(L):
    return L * (L - 1) // 2


def model_file_size(L, q, n_seqs):
    """Byte size of a plmc_v2 float32 file."""
    return 40 + q + 4 * n_seqs + 5 * L + 8 * L * q + 4 * L * (L - 1) * q * q


def write_model_file(path, L, q, n_valid, n_invalid, num_iter, theta, lambda_h, lambda_j,
                     lambda_group, n_eff, alphabet, weights, target_seq, index_list,
                     fi, hi, fij, jij):
    """
    Write a plmc_v2 float32 ``.model`` file.

    fij / jij : (L(L-1)/2, q, q) blocks for i<j in row-major pair order, [a][b] with a at
    site i -- exactly the order model.py:375-389 reads them back.  All f_ij blocks precede
    all J_ij blocks.  ``theta`` is the divergence as plmc sees it (1 - identity threshold).
    ``lambda_h`` must be >= 0 (a negative value flips the reference reader into
    mean-field mode, model.py:393-400).
    """
    if lambda_h < 0:
        raise ValueError("lambda_h must be non-negative in a plmc_v2 PLM model file")
    npair = n_pairs(L)
    weights = np.ascontiguousarray(weights, dtype="<f4")
    fi = np.ascontiguousarray(fi, dtype="<f4").reshape(L, q)
    hi = np.ascontiguousarray(hi, dtype="<f4").reshape(L, q)
    fij = np.ascontiguousarray(fij, dtype="<f4").reshape(npair, q, q)
    jij = np.ascontiguousarray(jij, dtype="<f4").reshape(npair, q, q)
    index_list = np.ascontiguousarray(index_list, dtype="<i4")
    if len(alphabet) != q or len(target_seq) != L or index_list.size != L:
        raise ValueError("alphabet / target_seq / index_list length mismatch")
    if weights.size != n_valid + n_invalid:
        raise ValueError("weights must hold one entry per sequence (valid + invalid)")
    with open(path, "wb") as f:
        np.array([L, q, n_valid, n_invalid, num_iter], dtype="<i4").tofile(f)
        np.array([theta, lambda_h, lambda_j, lambda_group, n_eff], dtype="<f4").tofile(f)
        f.write(alphabet.encode("ascii"))
        weights.tofile(f)
        f.write("".join(target_seq).encode("ascii"))
        index_list.tofile(f)
        fi.tofile(f)
        hi.tofile(f)
        fij.tofile(f)   # streamed straight from the contiguous block array
        jij.tofile(f)
    return path


def read_model_file(path):
    """
    Vectorised plmc_v2 float32 reader (two bulk reads for the pair blocks instead of the
    reference's L(L-1) np.fromfile calls, model.py:375-389).  Returns a dict with the
    triangular block arrays; no dense L x L x q x q expansion.
    """
    with open(path, "rb") as f:
        L, q, n_valid, n_invalid, num_iter = np.fromfile(f, "<i4", 5)
        theta, lambda_h, lambda_j, lambda_group, n_eff = np.fromfile(f, "<f4", 5)
        alphabet = f.read(int(q)).decode("ascii")
        weights = np.fromfile(f, "<f4", int(n_valid + n_invalid))
        target_seq = f.read(int(L)).decode("ascii")
        index_list = np.fromfile(f, "<i4", int(L))
        fi = np.fromfile(f, "<f4", int(L * q)).reshape(L, q)
        hi = np.fromfile(f, "<f4", int(L * q)).reshape(L, q)
        npair = n_pairs(int(L))
        fij = np.fromfile(f, "<f4", npair * q * q).reshape(npair, q, q)
        jij = np.fromfile(f, "<f4", npair * q * q).reshape(npair, q, q)
    return dict(L=int(L), q=int(q), n_valid=int(n_valid), n_invalid=int(n_invalid),
                num_iter=int(num_iter), theta=float(theta), lambda_h=float(lambda_h),
                lambda_j=float(lambda_j), lambda_group=float(lambda_group), n_eff=float(n_eff),
                alphabet=alphabet, weights=weights, target_seq=target_seq,
                index_list=index_list, fi=fi, hi=hi, fij=fij, jij=jij)


def write_raw_ec_file(path, index_list, target_seq, cn):
    """
    One line per i<j, i ascending then j: ``index_i A_i index_j A_j 0 cn`` with cn printed
    with 6 decimals (pairs.py:55-58 names the columns i, A_i, j, A_j, fn, cn; plmc leaves
    the 5th column as the literal 0).  cn : dense (L, L).
    """
    cn = np.asarray(cn)
    L = cn.shape[0]
    if L >= 2 and _native_writer(path, index_list, target_seq, cn):
        return path
    iu, ju = np.triu_indices(L, 1)
    idx = np.asarray(index_list)
    letters = np.array(list(target_seq))
    lines = [
        "%d %s %d %s 0 %.6f" % (idx[i], letters[i], idx[j], letters[j], cn[i, j])
        for i, j in zip(iu.tolist(), ju.tolist())
    ]
    with open(path, "w") as f:
        f.write("\n".join(lines))
        f.write("\n")
    return path


def _native_writer(path, index_list, target_seq, cn):
    """plm_write_raw_ec_file (host code of libplm_hip: one buffer, one write) when the library is built and every
    residue letter is one ASCII byte; False -> the Python lines above, which give the same bytes (PLM_IO_PYTHON=1 forces
    them: tests compare the two).  File formatting is plumbing, not the solver: like the alignment reader it may fall back."""
    import ctypes as C
    import os
    if os.environ.get("PLM_IO_PYTHON"):
        return False
    try:
        from evcouplings_amd import _lib
        lib = _lib.load()
        if not hasattr(lib, "plm_write_raw_ec_file"):
            return False
        seq = "".join(target_seq).encode("ascii")
        idx = np.ascontiguousarray(index_list, dtype=np.int32)
        if len(seq) != cn.shape[0] or idx.size != cn.shape[0]:
            return False
        dense = np.ascontiguousarray(cn, dtype=np.float64)
        return lib.plm_write_raw_ec_file(os.fsencode(path), cn.shape[0], idx.ctypes.data_as(C.c_void_p), seq,
                                         dense.ctypes.data_as(C.c_void_p)) == 0
    except Exception:       # noqa: BLE001 -- library missing / non-ASCII letters: the Python twin
        return False
