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


def test_dynamic_symbol_table_is_exactly_the_header():
    """a foreign host links against include/plm_hip.h and nothing else -- no C++ internals, no HIP
    kernel handles (-fvisibility=hidden + csrc/plm_exports.map)."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == _declared_symbols(), (set(exported) ^ set(_declared_symbols()))


def test_version_and_error_strings():
    lib = _lib.load()
    assert lib.plm_version() == _lib.ABI_VERSION == 2
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


def _build_c_host(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_host")
    libdir = os.path.join(root, "evcouplings_amd")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
           os.path.join(root, "examples", "c_host.c"), "-L" + libdir, "-lplm_hip", "-Wl,-rpath," + libdir, "-o", exe]
    run = subprocess.run(cmd, capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    return exe


def test_plain_c_host_compiles_and_links_against_the_header(tmp_path):
    """include/plm_hip.h is valid C11 and the library satisfies a C host's link (examples/c_host.c)."""
    assert os.path.exists(_build_c_host(tmp_path))


@pytest.mark.gpu
def test_plain_c_host_fits_and_finds_the_planted_pair(tmp_path):
    import subprocess
    run = subprocess.run([_build_c_host(tmp_path)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "top long-range pair 7 29" in run.stdout and "iter 10" in run.stderr
