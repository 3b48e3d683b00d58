"""Shapes far outside the BASELINE table ('maximum sizes'): very long alignments with few sequences, very deep
ones with few sites, a large family -- the HIP evaluation against the f64 oracle at a random point, the cluster sizes of
a sample of rows against numpy, and a short fit that must run and descend.  Run on the GPU box:
    python tests/probes/extreme_shapes_probe.py [name ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from evcouplings_amd import plm                                   # noqa: E402
from evcouplings_amd.synthetic import synthetic_msa               # noqa: E402
from oracle.oracle import Oracle                                  # noqa: E402

Q = 21
SHAPES = {"wide": (700, 1536), "wider": (300, 2500), "deep": (300000, 48), "family": (8000, 1000), "one_block": (100000, 16),
          "tiny": (2, 2), "single_seq": (1, 40), "single_site": (5000, 1)}


def main():
    names = sys.argv[1:] or list(SHAPES)
    o64 = Oracle("f64")
    o64.set_num_threads(len(os.sched_getaffinity(0)))
    bad = 0
    for name in names:
      try:
        N, L = SHAPES[name]
        rng = np.random.default_rng(N + L)
        if min(N, L) < 8:
            msa = rng.integers(0, Q, size=(N, L)).astype(np.int8)
        else:
            msa, _ = synthetic_msa(N, L, seed=N + L)
        t = time.time()
        counts = plm.reweight(msa, 0.8)
        t_rw = time.time() - t
        w = (1.0 / counts).astype(np.float32)
        n_eff = float(w.sum())
        # cluster sizes of a sample of rows by numpy
        T = int(np.ceil(0.8 * L - 1e-9))
        rows = rng.choice(N, size=min(N, 20), replace=False)
        for s in rows:
            c = int(((msa == msa[s]).sum(axis=1) >= T).sum())
            if c != int(counts[s]):
                print("%s: cluster size of row %d: hip %d numpy %d" % (name, s, counts[s], c))
                bad += 1
        n = plm.n_params(L, Q)
        x = (0.05 * rng.normal(size=n)).astype(np.float32)
        lh, lj = 0.01, plm.default_lambda_j(L, Q)
        t = time.time()
        fx, nll, g = plm.evaluate(msa, w, Q, lh, lj, x)
        t_ev = time.time() - t
        t = time.time()
        fxo, nllo, go = o64.eval(msa, w.astype(np.float64), Q, lh, lj, x.astype(np.float64))
        t_or = time.time() - t
        scale = max(np.abs(go).max(), 1e-30)
        e_g = np.abs(g - go).max() / scale
        e_fx = abs(fx - fxo) / max(abs(fxo), 1e-30)
        ok = e_g <= 2e-5 and e_fx <= 2e-6
        bad += 0 if ok else 1
        print("%s N=%d L=%d: n_eff %.1f (reweight %.3f s), eval %.3f s (oracle %.1f s): fx rel err %.2e, max |dg| / max|g| %.2e %s"
              % (name, N, L, n_eff, t_rw, t_ev, t_or, e_fx, e_g, "ok" if ok else "FAIL"), flush=True)
        t = time.time()
        r = plm.fit(msa, q=Q, max_iter=25, want_fij=False)
        tab = r["table"]
        desc = len(tab) < 2 or tab[-1][3] < tab[0][3]
        fin = np.isfinite(r["cn"]).all() and np.isfinite(r["jij"]).all()
        bad += 0 if (desc and fin) else 1
        print("   fit: %d iterations / %d evaluations in %.2f s, fx %.6g -> %.6g, status %d (%s) %s"
              % (r["iters"], r["n_evals"], time.time() - t, tab[0][3] if tab else float("nan"),
                 tab[-1][3] if tab else float("nan"), r["status"], r["status_msg"][:60], "ok" if desc and fin else "FAIL"),
              flush=True)
      except Exception as e:      # keep going: the point is to see every shape
        bad += 1
        print("%s: EXCEPTION %r" % (name, e), flush=True)
    print("FAILURES: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
