#!/usr/bin/env python3
"""GPU-side: run_plmc_hip end to end at the headline shape -- A2M file in, _ECs.txt + .model out -- with the reference's
default settings (100 iterations, -g) and to convergence; where the wall time goes outside the library."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import tools, alignment_io
from evcouplings_amd.synthetic import synthetic_msa, msa_to_a2m, BASE_SEED
N, L = 50000, 300
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
d = tempfile.mkdtemp()
a2m = msa_to_a2m(msa, os.path.join(d, "aln.a2m"))
tools.run_plmc_hip(a2m, os.path.join(d, "w.txt"), os.path.join(d, "w.model"), focus_seq="SYN", iterations=3)   # warm
for label, kw in (("default -g, 100 iterations", dict(ignore_gaps=True, iterations=100)),
                  ("-g to convergence", dict(ignore_gaps=True, iterations="max")),
                  ("21 states to convergence", dict(ignore_gaps=False, iterations="max"))):
    for io in ("native", "python"):
        if io == "python":
            os.environ["PLM_IO_PYTHON"] = "1"
        else:
            os.environ.pop("PLM_IO_PYTHON", None)
        t = time.time(); enc = alignment_io.encode_alignment(a2m, focus_seq="SYN"); t_enc = time.time() - t
        t = time.time()
        res, fit, _ = tools.infer_to_files(a2m, os.path.join(d, "ec.txt"), os.path.join(d, "p.model"), focus_seq="SYN",
                                           lambda_h=0.01, lambda_J=0.2 * (L - 1) if kw["ignore_gaps"] else None, **kw)
        dt = time.time() - t
        print("%-28s reader=%-6s wall %.3f s | encode %.3f | library total %.3f (optimize %.3f, %d iterations) | rest %.3f" % (
            label, io, dt, t_enc, fit["seconds"]["total"], fit["seconds"]["optimize"], fit["iters"],
            dt - t_enc - fit["seconds"]["total"]), flush=True)
