#!/usr/bin/env python3
"""Stand-in for the plmc binary in CPU tests of scripts/pin_against_plmc.py (TEST INFRASTRUCTURE): accepts plmc's command
line (the argv evcouplings/couplings/tools.py:202-262 builds), fits with the CPU oracle, writes the raw EC file and
prints the log lines parse_plmc_log expects.  It records its argv next to the EC file so the test can check the order."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv):
    from evcouplings_amd import alignment_io, cli, model_io, tools
    from oracle.oracle import Oracle
    o = cli.parse_argv(argv)
    enc = alignment_io.encode_alignment(o["alignment"], focus_seq=o.get("focus_seq"), alphabet=o.get("alphabet"))
    q = len(enc.alphabet)
    iters = o.get("iterations", 100)
    res = Oracle("f64").fit(enc.msa, q, theta_id=round(1.0 - o["theta_div"], 12), lambda_h=o["lambda_h"],
                            lambda_j=o["lambda_J"], max_iter=3000 if iters == "max" else int(iters), epsilon=1e-6,
                            ignore_gaps=o["ignore_gaps"], want_fij=False)
    model_io.write_raw_ec_file(o["couplings_file"], enc.index_list, enc.target_seq, res["cn"])
    with open(o["couplings_file"] + ".argv.json", "w") as f:
        json.dump(argv, f)
    sys.stderr.write(tools.format_plmc_log(o.get("focus_seq"), enc.focus_index, enc.n_valid_seqs, enc.n_total_seqs,
                                           enc.msa.shape[1], enc.n_total_sites, enc.region_start, res["n_eff"],
                                           "stand-in", res["table"][:3]))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
