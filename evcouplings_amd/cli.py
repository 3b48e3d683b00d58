"""
``plmc``-argv-compatible command line front end of the HIP solver.

An unmodified EVcouplings can use the GPU solver by pointing ``tools.plmc`` at the
``bin/plmc_hip`` launcher: ``run_plmc`` builds this argv (evcouplings/couplings/tools.py:202-262)

    plmc -c ECs [-o MODEL] [-f FOCUS] [-g] [-m ITER|max] [-a ALPHABET] [-t 1-THETA] [-s SCALE]
         [-lh LAMBDA_H] [-le LAMBDA_J] [-lg LAMBDA_G] [-n CPUS|max] ALIGNMENT

launches it and regex-parses stderr (tools.py:20-108, 286).  This shim accepts exactly that
argv, writes the two files and prints the log lines the parser needs on stderr.  Exit code
0 on success, 1 on failure (with the reason on stderr).

`-n CPUS` is plmc's thread count: accepted and ignored.  Options plmc does not have (an unmodified pipeline sets the
environment variables instead): `--solver vp|joint` (PLM_HIP_SOLVER: "joint" = L-BFGS over fields and couplings
together as libLBFGS-based plmc runs it; "vp" = variable projection, the default), `--gpus N|max` (PLM_HIP_GPUS: GPUs of
this node to shard the fit over), `--epsilon E` (stop rule |g|/max(1,|x|) < E, default 1e-3),
`--conventions BITS` (PLM_HIP_CONVENTIONS).
"""
import sys

USAGE = __doc__

_WITH_VALUE = {"-c": "couplings_file", "-o": "param_file", "-f": "focus_seq", "-m": "iterations",
               "-a": "alphabet", "-t": "theta_div", "-s": "scale", "-lh": "lambda_h", "-le": "lambda_J",
               "-lg": "lambda_g", "-n": "cpu", "--solver": "solver", "--gpus": "gpus", "--epsilon": "epsilon",
               "--conventions": "conventions"}


def parse_argv(argv):
    """-> dict of options; raises ValueError on malformed input."""
    opts = {"ignore_gaps": False}
    positional = []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "-g":
            opts["ignore_gaps"] = True
        elif a in _WITH_VALUE:
            if i + 1 >= len(argv):
                raise ValueError("option %s needs a value" % a)
            opts[_WITH_VALUE[a]] = argv[i + 1]
            i += 1
        elif a in ("-h", "--help"):
            opts["help"] = True
        elif a.startswith("-") and len(a) > 1:
            raise ValueError("unknown option %s" % a)
        else:
            positional.append(a)
        i += 1
    if opts.get("help"):
        return opts
    if len(positional) != 1:
        raise ValueError("expected exactly one alignment file, got %d" % len(positional))
    if "couplings_file" not in opts:
        raise ValueError("-c <couplings file> is required")
    opts["alignment"] = positional[0]
    if "conventions" in opts:
        opts["conventions"] = int(opts["conventions"], 0)
    for key in ("scale", "lambda_h", "lambda_J", "lambda_g", "theta_div", "epsilon"):
        if key in opts:
            opts[key] = float(opts[key])
    if "iterations" in opts and opts["iterations"].lower() != "max":
        opts["iterations"] = int(opts["iterations"])
    return opts


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    try:
        opts = parse_argv(argv)
    except ValueError as exc:
        sys.stderr.write("plmc_hip: %s\n" % exc)
        return 1
    if opts.get("help"):
        sys.stdout.write(USAGE)
        return 0
    from evcouplings_amd import tools
    theta = None
    if "theta_div" in opts:
        theta = round(1.0 - opts["theta_div"], 12)   # run_plmc sends 1 - theta (tools.py:236-239)
    try:
        _, _, log = tools.infer_to_files(
            opts["alignment"], opts["couplings_file"], opts.get("param_file"),
            focus_seq=opts.get("focus_seq"), alphabet=opts.get("alphabet"), theta=theta,
            scale=opts.get("scale"), ignore_gaps=opts["ignore_gaps"], iterations=opts.get("iterations"),
            lambda_h=opts.get("lambda_h"), lambda_J=opts.get("lambda_J"), lambda_g=opts.get("lambda_g"),
            cpu=opts.get("cpu"), solver=opts.get("solver"), gpus=opts.get("gpus"), epsilon=opts.get("epsilon"),
            conventions=opts.get("conventions"))
    except Exception as exc:
        sys.stderr.write("plmc_hip: %s: %s\n" % (type(exc).__name__, exc))
        return 1
    sys.stderr.write(log)
    return 0


if __name__ == "__main__":
    sys.exit(main())
