"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol that
include/plm_hip.h declares.  No compute calls (there is no GPU here)."""
import os
import re

import pytest

from evcouplings_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "plm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(plm_[a-z_0-9]+)\s*\(", text)) - {"plm_iter_cb", "plm_exchange_cb"})


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    bound = {name for name, _, _ in _lib.SYMBOLS}
    for name in declared:
        assert hasattr(lib, name), "missing export %s" % name
        assert name in bound, "python binding missing for %s" % name
    assert bound <= set(declared), "binding for undeclared symbol: %s" % (bound - set(declared))


def test_version_and_error_strings():
    lib = _lib.load()
    assert lib.plm_version() == 1
    assert lib.plm_strerror(0) == b"ok"
    assert b"argument" in lib.plm_strerror(-1)


def test_no_cpu_fallback_without_a_gpu():
    """On a box without a gfx950 device every compute entry point must fail loudly."""
    import numpy as np
    from evcouplings_amd import plm
    lib = _lib.load()
    if lib.plm_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_lib.PlmError):
        plm.reweight(np.zeros((4, 8), np.int8), 0.8)
    with pytest.raises(_lib.PlmError):
        plm.fit(np.zeros((4, 8), np.int8), q=21, max_iter=1)
