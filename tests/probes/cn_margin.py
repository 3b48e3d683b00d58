#!/usr/bin/env python3
"""GPU-side: margins of the CN parity tests (max |cn_hip - cn_oracle|) for several seeds."""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa
from oracle.oracle import Oracle
orc = Oracle("f64")
for seed in (31, 32, 33):
    for gaps in (False, True):
        msa, _ = synthetic_msa(600, 24, seed=seed)
        ref = orc.fit(msa, 21, lambda_h=0.01, max_iter=3000, epsilon=1e-7, ignore_gaps=gaps)
        res = plm.fit(msa, 21, lambda_h=0.01, max_iter=3000, epsilon=2e-6, ignore_gaps=gaps)
        print("seed %d gaps=%s: iters hip %d oracle %d | max|dcn|=%.3g max|dJ|=%.3g max|dh|=%.3g status=%d" % (
            seed, gaps, res["iters"], ref["iters"], np.abs(res["cn"] - ref["cn"]).max(),
            np.abs(res["jij"] - ref["jij"]).max(), np.abs(res["hi"] - ref["hi"]).max(), res["status"]))
