"""A/B of kernel variants on the GPU: a variant library built with scripts/build_variant.sh NAME "-D..." (it lands in
evcouplings_amd/libplm_NAME.so) must reproduce the default library's evaluation.  Skipped when no variant library is
present -- the product build ships none.  Each library runs in its own process (PLM_HIP_LIB selects it at load time)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = sorted(p for p in glob.glob(os.path.join(ROOT, "evcouplings_amd", "libplm_*.so"))
                  if not p.endswith("libplm_hip.so"))

_WORKER = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa
out = {}
for (N, L, q, gaps) in ((700, 70, 21, False), (600, 50, 21, True), (500, 40, 5, False), (3000, 300, 21, False)):
    msa, _ = synthetic_msa(N, L, seed=N + L, q=q)
    qm = q - 1 if gaps else q
    x = (0.05 * np.random.default_rng(L).normal(size=plm.n_params(L, qm))).astype(np.float32)
    with plm.PlmContext(msa, q=q, ignore_gaps=gaps, lambda_h=0.01, lambda_j=2.0) as ctx:
        ctx.reweight(); ctx.set_x(x)
        fx, nll = ctx.eval()
        out["fx_%%d_%%d_%%d" %% (N, L, q)] = np.array([fx, nll])
        out["g_%%d_%%d_%%d" %% (N, L, q)] = ctx.get_g()
np.savez(sys.argv[1], **out)
"""


def _run(lib, path):
    env = dict(os.environ)
    if lib:
        env["PLM_HIP_LIB"] = lib
    else:
        env.pop("PLM_HIP_LIB", None)
    run = subprocess.run([sys.executable, "-c", _WORKER % ROOT, path], env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    return np.load(path)


@pytest.mark.gpu
@pytest.mark.skipif(not VARIANTS, reason="no variant library built (scripts/build_variant.sh)")
@pytest.mark.parametrize("lib", VARIANTS, ids=[os.path.basename(v) for v in VARIANTS])
def test_variant_library_reproduces_the_default_evaluation(lib, tmp_path):
    ref = _run(None, str(tmp_path / "ref.npz"))
    got = _run(lib, str(tmp_path / "var.npz"))
    for k in ref.files:
        if k.startswith("fx"):
            np.testing.assert_allclose(got[k], ref[k], rtol=2e-7)
        else:
            scale = np.abs(ref[k]).max()
            np.testing.assert_allclose(got[k], ref[k], atol=3e-6 * scale, rtol=0)   # same arithmetic, other summation order
