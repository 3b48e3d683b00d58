"""
Alignment input for the couplings stage: A2M/FASTA -> the int8 state matrix the HIP
solver consumes, following plmc's focus-mode conventions as the reference relies on them.

Rules restated here (in-repo statement of the same column rule:
evcouplings/couplings/mean_field.py:103-120):
  * focus mode: model columns = positions where the focus sequence has an UPPERCASE,
    non-gap character; everything else ('.', '-', lowercase in the focus row) is dropped.
  * index_list[i] = region_start + number of focus-sequence residues (letters of either
    case) before that column; region_start comes from the ``NAME/start-end`` header
    (evcouplings/couplings/tools.py:219 passes only NAME to plmc -f).
  * a sequence with a symbol outside the alphabet in a kept column is INVALID: it is
    excluded from the model and counted in N_invalid (plmc behaviour [recollection]; the
    in-repo Alignment class instead maps unknown symbols to gap, alignment.py:446-476 --
    deliberately not copied).  '.' in a kept column is treated as the gap character.
  * non-focus mode (focus_seq=None): every column is a model column, numbering 1..L,
    the first record supplies target_seq.
"""
import re

import numpy as np

ALPHABET_PROTEIN = "-ACDEFGHIKLMNPQRSTVWY"   # evcouplings/align/alignment.py:25-26


class AlignmentFormatError(ValueError):
    pass


def read_fasta_records(path):
    """-> (ids, sequences) with sequences as python bytes objects (A2M is FASTA-framed)."""
    ids, seqs, cur = [], [], []
    with open(path, "rb") as f:
        for raw in f:
            line = raw.strip()
            if not line:
                continue
            if line.startswith(b">"):
                if ids:
                    seqs.append(b"".join(cur))
                ids.append(line[1:].decode("ascii", "replace"))
                cur = []
            else:
                if not ids:
                    raise AlignmentFormatError("sequence data before the first '>' header in %s" % path)
                cur.append(line)
    if ids:
        seqs.append(b"".join(cur))
    if not ids:
        raise AlignmentFormatError("no sequences in %s" % path)
    return ids, seqs


def _native():
    """libplm_hip's host-side input functions (plm_fasta_split, plm_encode_columns), or None when the library is not built /
    PLM_IO_PYTHON=1 asks for the Python twin (tests compare the two).  Parsing is plumbing, not the solver: unlike every
    compute entry point it may fall back -- to code that gives the same arrays, byte for byte."""
    import os
    if os.environ.get("PLM_IO_PYTHON"):
        return None
    try:
        from evcouplings_amd import _lib
        lib = _lib.load()
        return lib if hasattr(lib, "plm_fasta_split") and hasattr(lib, "plm_encode_columns") else None
    except Exception:       # noqa: BLE001 -- library missing or from an older build
        return None


def read_fasta_matrix(path):
    """-> (ids, uint8 matrix [n_records, width]): read_fasta_records + the equal-length check + the byte matrix, through
    plm_fasta_split when the library is there (one pass over the file image instead of a Python loop over its lines)."""
    lib = _native()
    if lib is None:
        ids, seqs = read_fasta_records(path)
        width = len(seqs[0])
        for k, s in enumerate(seqs):
            if len(s) != width:
                raise AlignmentFormatError("sequence %d (%s) has length %d, expected %d" % (k + 1, ids[k], len(s), width))
        return ids, np.frombuffer(b"".join(seqs), dtype=np.uint8).reshape(len(seqs), width)
    import ctypes as C
    with open(path, "rb") as f:
        buf = f.read()
    n = len(buf)
    nrec, nbytes = C.c_int64(0), C.c_int64(0)
    if lib.plm_fasta_split(buf, n, C.byref(nrec), C.byref(nbytes), None, None, None, None) != 0:
        raise AlignmentFormatError("sequence data before the first '>' header in %s" % path)
    if nrec.value == 0:
        raise AlignmentFormatError("no sequences in %s" % path)
    hdr_off = np.zeros(nrec.value, np.int64)
    hdr_len = np.zeros(nrec.value, np.int32)
    seq_len = np.zeros(nrec.value, np.int64)
    seq = np.empty(max(1, nbytes.value), np.uint8)
    lib.plm_fasta_split(buf, n, C.byref(nrec), C.byref(nbytes), hdr_off.ctypes.data, hdr_len.ctypes.data,
                        seq_len.ctypes.data, seq.ctypes.data)
    ids = [buf[o:o + l].decode("ascii", "replace") for o, l in zip(hdr_off.tolist(), hdr_len.tolist())]
    width = int(seq_len[0])
    bad = np.flatnonzero(seq_len != width)
    if bad.size:
        k = int(bad[0])
        raise AlignmentFormatError("sequence %d (%s) has length %d, expected %d" % (k + 1, ids[k], int(seq_len[k]), width))
    return ids, seq[:nrec.value * width].reshape(nrec.value, width)


def parse_region(header):
    """'NAME/start-end ...' -> (NAME, start, end); start/end None if absent."""
    name = header.split()[0] if header.split() else header
    m = re.match(r"^(.*)/(\d+)-(\d+)$", name)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(3))
    return name, None, None


class EncodedAlignment:
    """Result of encode_alignment(); plain attributes, numpy arrays."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def encode_alignment(path, focus_seq=None, alphabet=None):
    """
    Read an A2M/FASTA alignment and encode it for the solver.

    Returns EncodedAlignment with
      msa            int8 (N_valid, L) states 0..q-1, valid sequences in file order
      valid          bool (N_total,)   which input sequences are valid
      focus_index    1-based index of the focus sequence among all sequences, or None
      columns        indices of the kept alignment columns
      index_list     int32 (L,) sequence numbering of the model columns
      target_seq     str of length L
      region_start   int
      n_total_sites  number of residues of the focus sequence (focus mode) / columns
      alphabet       the alphabet used (gap first)
    """
    alphabet = ALPHABET_PROTEIN if alphabet is None else alphabet
    if len(set(alphabet)) != len(alphabet) or len(alphabet) < 2 or len(alphabet) > 32:
        raise AlignmentFormatError("alphabet must hold 2..32 distinct symbols, gap first")
    ids, mat = read_fasta_matrix(path)
    n_total, width = mat.shape
    gap = ord(alphabet[0])

    focus_index = None
    region_start = 1
    if focus_seq is not None:
        want = focus_seq.split("/")[0]            # tools.py:219
        # (substring test first: the regular expression of parse_region on every header was 60 ms at 50 000 records)
        hit = [k for k, h in enumerate(ids) if want in h and (parse_region(h)[0] == want or h.split()[0] == focus_seq)]
        if not hit:
            raise AlignmentFormatError("focus sequence %r not found in %s" % (focus_seq, path))
        fk = hit[0]
        focus_index = fk + 1
        frow = mat[fk]
        is_upper = (frow >= ord("A")) & (frow <= ord("Z"))
        is_lower = (frow >= ord("a")) & (frow <= ord("z"))
        columns = np.nonzero(is_upper)[0]
        residues_before = np.cumsum(is_upper | is_lower) - 1      # 0-based residue index per column
        _, start, _ = parse_region(ids[fk])
        region_start = start if start is not None else 1
        index_list = (region_start + residues_before[columns]).astype(np.int32)
        n_total_sites = int((is_upper | is_lower).sum())
        target_seq = frow[columns].tobytes().decode("ascii")
    else:
        columns = np.arange(width)
        index_list = np.arange(1, width + 1, dtype=np.int32)
        n_total_sites = width
        target_seq = mat[0].tobytes().decode("ascii")

    if columns.size < 2:
        raise AlignmentFormatError("fewer than 2 model columns")
    lut = np.full(256, -1, dtype=np.int8)
    for k, ch in enumerate(alphabet):
        lut[ord(ch)] = k
    lut[ord(".")] = 0                               # insert-gap in a match column counts as gap
    lib = _native()
    if lib is None:
        enc = lut[mat[:, columns]]
        valid = (enc >= 0).all(axis=1)
    else:                                           # column selection, lookup and validity in one pass
        mat = np.ascontiguousarray(mat)
        cols64 = np.ascontiguousarray(columns, dtype=np.int64)
        enc = np.empty((n_total, cols64.size), np.int8)
        ok = np.empty(n_total, np.uint8)
        if lib.plm_encode_columns(mat.ctypes.data, n_total, width, cols64.ctypes.data, cols64.size, lut.ctypes.data,
                                  enc.ctypes.data, ok.ctypes.data) != 0:
            raise AlignmentFormatError("column encoding failed")
        valid = ok.astype(bool)
    if focus_index is not None and not valid[focus_index - 1]:
        raise AlignmentFormatError("focus sequence contains symbols outside the alphabet")
    if not valid.any():
        raise AlignmentFormatError("no valid sequences")
    if focus_index is None:
        target_seq = "".join(alphabet[k] if k >= 0 else "-" for k in enc[0]) if valid[0] else target_seq
    return EncodedAlignment(
        msa=np.ascontiguousarray(enc if valid.all() else enc[valid]), valid=valid, focus_index=focus_index,
        columns=columns, index_list=index_list, target_seq=target_seq, region_start=int(region_start),
        n_total_sites=int(n_total_sites), n_total_seqs=int(n_total), n_valid_seqs=int(valid.sum()),
        alphabet=alphabet, gap=chr(gap), ids=ids)
