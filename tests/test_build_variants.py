"""The compile-time alternative of the kernel file must keep compiling for gfx950: the dense forward GEMM of rounds
1-2 (-DPLM_SPARSE_FWD=0, the A/B partner of the sparse one).  hipcc cross-compiles without a GPU.  (The 4-row-fragment
tiling k_fwd4 of round 2 was measured in round 3 -- 4.07-4.14 ms against 3.78-3.88 ms -- and removed.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("flag", ["-DPLM_SPARSE_FWD=0"])
def test_kernel_variant_compiles(flag, tmp_path):
    out = tmp_path / "k.o"
    run = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-value", flag, "-c",
                          os.path.join(ROOT, "evcouplings_amd", "csrc", "plm_kernels.hip"), "-o", str(out)],
                         capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    assert out.stat().st_size > 100000
