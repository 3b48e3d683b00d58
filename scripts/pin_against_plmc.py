#!/usr/bin/env python3
"""
Pin the HIP solver against a real plmc binary -- the one command a host that HAS plmc runs (SURVEY.md section 8c,
golden (vi)).  plmc is not available where this repository is built (no source, no binary, no
network), so rows a6 / a7 / N1 of the scope table are parity-unpinned; every convention of plmc that could not be
checked is a switch (PLM_CONV_* of include/plm_hip.h).  This script

  1. writes the synthetic alignments of BASELINE.json (config 2 and the headline by default) as A2M files,
  2. runs `plmc` on each with exactly the argv `run_plmc` assembles (evcouplings/couplings/tools.py:202-262), with and
     without -g, run to convergence (-m max would take plmc hours: the iteration count is an option),
  3. runs the HIP solver on the same files for every combination of the convention switches that applies to the mode
     (solver "joint" at the same iteration count -- plmc's own algorithm, the like-for-like comparison when plmc was
     stopped by its iteration limit -- and solver "vp" to a 10x tighter stop rule: the optimum itself, which a converged
     plmc must sit within its own tolerance of),
  4. prints a table of max |dCN| per combination and names the combination that agrees best,
  5. stores plmc's _ECs.txt under tests/golden/plmc_<case>[_g]_ECs.txt with a JSON record of argv, plmc's stderr and the
     winning switches: from then on tests/ can pin the solver without the binary.

    python scripts/pin_against_plmc.py --plmc /path/to/plmc [--cases config2,headline] [--iterations 500]
                                       [--out tests/golden] [--small]

`--plmc` may be ANY program with plmc's command line; the CPU test of this script (tests/test_host_layer.py) passes a
stand-in that answers like plmc.  `--small` swaps the BASELINE shapes for a 400 x 30 alignment (that test, and a first
try on a new host).  The HIP side needs an MI355X; with `--no-hip` only steps 1, 2 and 5 run (golden files from a
CPU-only host that has plmc).
"""
import argparse
import itertools
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from evcouplings_amd.synthetic import BASE_SEED, ALPHABET_PROTEIN, msa_to_a2m, synthetic_msa  # noqa: E402

CASES = {"config2": (20000, 200, 2), "headline": (50000, 300, 1), "small": (400, 30, 7)}
# convention switches that matter per mode (include/plm_hip.h): without -g only the threshold rule and the norm;
# with -g also the three gap conventions
CONV_PLAIN = (32, 512)
CONV_GAPS = (32, 64, 128, 256, 512)


def plmc_argv(binary, alignment, ec_file, model_file, focus, ignore_gaps, iterations, theta, lambda_h, lambda_j, cpu=None):
    """The argv of evcouplings/couplings/tools.py:202-262, in its order."""
    cmd = [binary, "-c", ec_file, "-o", model_file, "-f", focus.split("/")[0]]
    if ignore_gaps:
        cmd += ["-g"]
    cmd += ["-m", str(iterations), "-t", str(1.0 - theta), "-s", "1.0", "-lh", str(lambda_h), "-le", str(lambda_j),
            "-lg", "0.0"]
    if cpu is not None:
        cmd += ["-n", str(cpu)]
    return cmd + [alignment]


def read_cn(ec_file):
    """CN column of a raw EC file (couplings/pairs.py:55-58: i A_i j A_j fn cn) as a dict keyed by (i, j)."""
    out = {}
    with open(ec_file) as f:
        for line in f:
            t = line.split()
            out[(int(t[0]), int(t[2]))] = float(t[5])
    return out


def combos(bits):
    for r in range(len(bits) + 1):
        for c in itertools.combinations(bits, r):
            yield sum(c)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--plmc", required=True, help="plmc binary (or any program with its command line)")
    ap.add_argument("--cases", default="config2,headline")
    ap.add_argument("--small", action="store_true", help="400 x 30 instead of the BASELINE shapes")
    ap.add_argument("--iterations", default="500", help="plmc -m (an integer or max)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--work", default=None, help="scratch directory (default: <out>/_pin_work)")
    ap.add_argument("--cpu", default=None, help="plmc -n")
    ap.add_argument("--no-hip", action="store_true", help="only produce the golden files")
    ap.add_argument("--modes", default="plain,gaps")
    args = ap.parse_args(argv)
    work = args.work or os.path.join(args.out, "_pin_work")
    os.makedirs(work, exist_ok=True)
    os.makedirs(args.out, exist_ok=True)
    cases = ["small"] if args.small else [c for c in args.cases.split(",") if c]
    theta, lambda_h = 0.8, 0.01
    report = []
    for case in cases:
        N, L, k = CASES[case]
        msa, _ = synthetic_msa(N, L, seed=BASE_SEED + k)
        ali = msa_to_a2m(msa, os.path.join(work, "%s.a2m" % case))
        focus = "SYN/1-%d" % L
        for mode in [m for m in args.modes.split(",") if m]:
            gaps = mode == "gaps"
            q = len(ALPHABET_PROTEIN) - (1 if gaps else 0)
            lambda_j = 0.01 * (q - 1) * (L - 1)                       # couplings/protocol.py:159-179
            tag = "%s%s" % (case, "_g" if gaps else "")
            ec, model = os.path.join(work, tag + "_plmc_ECs.txt"), os.path.join(work, tag + "_plmc.model")
            cmd = plmc_argv(args.plmc, ali, ec, model, focus, gaps, args.iterations, theta, lambda_h, lambda_j, args.cpu)
            t0 = time.time()
            run = subprocess.run(cmd, capture_output=True, text=True)
            secs = time.time() - t0
            if not os.path.exists(ec) or os.path.getsize(ec) == 0:
                raise SystemExit("plmc produced no EC file for %s (exit %d):\n%s" % (tag, run.returncode, run.stderr[-2000:]))
            ref = read_cn(ec)
            golden_ec = os.path.join(args.out, "plmc_%s_ECs.txt" % tag)
            with open(ec) as src, open(golden_ec, "w") as dst:
                dst.write(src.read())
            record = {"case": case, "ignore_gaps": gaps, "N": N, "L": L, "seed": BASE_SEED + k, "argv": cmd[1:],
                      "plmc_seconds": secs, "plmc_stderr_tail": run.stderr[-3000:], "golden": os.path.basename(golden_ec)}
            rows = []
            if not args.no_hip:
                from evcouplings_amd import tools
                keys = sorted(ref)
                refv = np.array([ref[kk] for kk in keys])
                for solver, iters in (("joint", args.iterations), ("vp", "max")):
                    for conv in combos(CONV_GAPS if gaps else CONV_PLAIN):
                        out_ec = os.path.join(work, "%s_hip_%s_%d_ECs.txt" % (tag, solver, conv))
                        t0 = time.time()
                        tools.infer_to_files(ali, out_ec, None, focus_seq=focus, theta=theta, ignore_gaps=gaps,
                                             iterations=int(iters) if str(iters).isdigit() else iters, lambda_h=lambda_h,
                                             lambda_J=lambda_j, conventions=conv, solver=solver,
                                             epsilon=1e-4 if solver == "vp" else None)
                        got = read_cn(out_ec)
                        d = float(np.abs(np.array([got[kk] for kk in keys]) - refv).max())
                        rows.append({"solver": solver, "iterations": str(iters), "conventions": conv, "max_abs_dCN": d,
                                     "seconds": time.time() - t0})
                rows.sort(key=lambda r: r["max_abs_dCN"])
                record["table"] = rows
                record["best"] = rows[0]
                print("\n%s: max |dCN| against plmc (%s iterations, %.1f s)" % (tag, args.iterations, secs))
                print("  %-6s %-10s %-12s %s" % ("solver", "iterations", "conventions", "max|dCN|"))
                for r in rows:
                    print("  %-6s %-10s %-12d %.3e" % (r["solver"], r["iterations"], r["conventions"], r["max_abs_dCN"]))
                print("  best: solver=%s conventions=%d (PLM_HIP_SOLVER / PLM_HIP_CONVENTIONS), within 1e-4: %s" % (
                    rows[0]["solver"], rows[0]["conventions"], rows[0]["max_abs_dCN"] < 1e-4))
            with open(os.path.join(args.out, "plmc_%s.json" % tag), "w") as f:
                json.dump(record, f, indent=1)
            report.append(record)
    print("\nwrote %d golden EC file(s) to %s" % (len(report), args.out))
    return report


if __name__ == "__main__":
    main()
