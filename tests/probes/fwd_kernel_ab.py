#!/usr/bin/env python3
"""GPU-side probe: k_fwd (8 waves x 2 row fragments x 21 states) against k_fwd_w (4 waves x 8 row fragments x 7 states in AccVGPRs, K loop in
assembly) -- PLM_FWD_KERNEL = 0 | 1, read once per context.  The gradients must be bit-identical (same instruction, operands and K order per accumulator);
prints the HIP-event kernel times of both.  usage: fwd_kernel_ab.py [N L] [planes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
N = int(sys.argv[1]) if len(sys.argv) > 2 else 50000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
planes = sys.argv[3] if len(sys.argv) > 3 else "3"
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
rng = np.random.default_rng(5)
w = rng.uniform(0.05, 1.0, N).astype(np.float32)
out = {}
x0 = None
os.environ["PLM_BWD_PLANES"] = planes
os.environ["PLM_FWD_ACCURATE"] = "0"     # plm_eval would run the exact kernel otherwise; this probe is about the plain ones
for kern in ("0", "1"):
    os.environ["PLM_FWD_KERNEL"] = kern
    with plm.PlmContext(msa, 21, max_iter=3, epsilon=1e-3) as ctx:
        ctx.set_weights(w); ctx.marginals(pairs=False)
        if x0 is None:
            ctx.set_x(None); ctx.optimize(); x0 = ctx.get_x()
            x0 = x0 + rng.normal(0, 0.02, x0.shape).astype(np.float32)
        ctx.set_x(x0)
        fx, _ = ctx.eval(); g = ctx.get_g()
        km = ctx.time_kernels(reps=5)
    out[kern] = (fx, g, km)
    print("PLM_FWD_KERNEL=%s: fx %.10g  |g| %.8g  forward %.3f ms  (expand %.3f, hpass %.3f)" % (
        kern, fx, np.linalg.norm(g.astype(np.float64)), km["forward"], km.get("expand", 0), km.get("hpass", 0)), flush=True)
same = np.array_equal(out["0"][1], out["1"][1])
print("gradients bit-identical:", same, " max |diff| %.3g" % np.abs(out["0"][1].astype(np.float64) - out["1"][1]).max())
sys.exit(0 if same else 1)
