"""
In-process replacement for ``evcouplings.couplings.tools.run_plmc``.

``run_plmc_hip`` has the signature, argument meaning, return type and error behaviour of
``run_plmc`` (evcouplings/couplings/tools.py:126-307) but runs the inference on an MI355X
through libplm_hip.so instead of launching the plmc binary, and writes the same two files:
the raw EC file (pairs.py:55-58) and the plmc_v2 ``.model`` file (model.py:317-389).
``format_plmc_log`` renders what plmc would have printed on stderr, in the grammar that
``parse_plmc_log`` (tools.py:20-108) consumes -- used by the CLI shim (cli.py).
"""
import os
from collections import namedtuple

import numpy as np
import pandas as pd

from evcouplings_amd import alignment_io, model_io

try:  # reuse the reference's exception types when it is importable, so callers' except clauses work
    from evcouplings.utils.system import ExternalToolError, ResourceError
except Exception:  # pragma: no cover - reference package not installed

    class ResourceError(Exception):
        """Missing/empty input or output file (mirrors evcouplings.utils.system.ResourceError)."""

    class ExternalToolError(Exception):
        """Solver failure (mirrors evcouplings.utils.system.ExternalToolError)."""

# same fields, same order as evcouplings/couplings/tools.py:113-123
PlmcResult = namedtuple(
    "PlmcResult",
    ["couplings_file", "param_file", "iteration_table", "focus_seq_index", "num_valid_seqs",
     "num_total_seqs", "num_valid_sites", "num_total_sites", "region_start", "effective_samples",
     "optimization_status"])

ITER_COLUMNS = ["iter", "time", "cond", "fx", "-loglk", "||h||", "||e||"]

# plmc's own defaults for options run_plmc leaves unset [recollection, SURVEY.md App. C];
# the reference's configs always pass them explicitly (config/sample_config_monomer.txt:142-178)
DEFAULTS = dict(theta=0.8, scale=1.0, lambda_h=0.01, lambda_J=100.0, lambda_g=0.0, iterations=100,
                epsilon=1e-3)


def _iter_rows(table):
    """7-tuples -> list of string rows in the fixed-point format the stderr grammar needs
    (each numeric cell must match ``\\d+\\.\\d+``, tools.py:59-61)."""
    rows = []
    for it, secs, cond, fx, nll, nh, ne in table:
        rows.append(("%d" % it, "%.3f" % secs, "%.6f" % cond, "%.4f" % fx, "%.4f" % nll, "%.4f" % nh,
                     "%.4f" % ne))
    return rows


def iteration_dataframe(table):
    """DataFrame with the columns parse_plmc_log would produce (string cells, tools.py:80-81)."""
    return pd.DataFrame(_iter_rows(table), columns=ITER_COLUMNS)


def format_plmc_log(focus_name, focus_index, n_valid, n_total, n_sites, n_total_sites, region_start,
                    n_eff, status_msg, table, theta=None):
    """Text with every line ``parse_plmc_log`` looks for.  `theta` is the identity threshold
    the run used (default: plmc's own 0.8)."""
    theta = DEFAULTS["theta"] if theta is None else float(theta)
    lines = []
    if focus_index is not None:
        lines.append("Found focus %s as sequence %d" % (focus_name, focus_index))
    lines.append("%d valid sequences out of %d" % (n_valid, n_total))
    if focus_index is not None:
        lines.append("%d sites out of %d" % (n_sites, n_total_sites))
        lines.append("Region starts at %d" % region_start)
    lines.append("Effective number of samples: %.1f\t(%.0f%% identical neighborhood = 1.000 samples)"
                 % (n_eff, 100.0 * theta))
    lines.append("\t".join(ITER_COLUMNS))
    for row in _iter_rows(table):
        lines.append("\t".join(row))
    lines.append("Gradient optimization: %s" % status_msg)
    return "\n".join(lines) + "\n"


def _valid_file(path):
    try:
        return os.stat(path).st_size > 0
    except (OSError, TypeError):
        return False


SOLVERS = ("vp", "joint")


def solver_from_env(solver=None):
    """Which optimiser a run_plmc-style call uses: explicit value, else the environment variable PLM_HIP_SOLVER (for an
    unmodified pipeline; same pattern as PLM_HIP_CONVENTIONS), else "vp".
      "vp"     variable projection (default): fields solved by Newton for every trial couplings, L-BFGS over the
               couplings -- reaches the optimum in ~20x fewer iterations;
      "joint"  L-BFGS over fields and couplings together, the route libLBFGS-based plmc takes (PLM_FLAG_JOINT_LBFGS):
               choose it when the iterate at a FIXED iteration count (the reference default is 100,
               config/sample_config_monomer.txt:149) should be plmc-like rather than converged."""
    if solver is None:
        solver = os.environ.get("PLM_HIP_SOLVER", "") or "vp"
    solver = str(solver).lower()
    if solver not in SOLVERS:
        raise ValueError("solver must be one of %s, got %r" % ("/".join(SOLVERS), solver))
    return solver


def infer_to_files(alignment, couplings_file, param_file=None, focus_seq=None, alphabet=None, theta=None,
                   scale=None, ignore_gaps=False, iterations=None, lambda_h=None, lambda_J=None,
                   lambda_g=None, cpu=None, epsilon=None, lbfgs_m=6, device=0, distributed=False,
                   callback=None, conventions=None, solver=None, gpus=None):
    """Does the work of run_plmc_hip and additionally returns the raw fit dict and the log text.
    Extensions over run_plmc's parameters: `solver` ("vp" | "joint", default from PLM_HIP_SOLVER), `gpus` (number of
    GPUs of this node to shard the fit over, default from PLM_HIP_GPUS, else 1), `epsilon`, `lbfgs_m`, `conventions`."""
    from evcouplings_amd import plm   # imports the HIP library: fails loudly if it is not built

    if not _valid_file(alignment):
        raise ResourceError("Alignment file does not exist: {}".format(alignment))
    for path in (couplings_file, param_file):
        if path:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

    theta = DEFAULTS["theta"] if theta is None else float(theta)
    scale = DEFAULTS["scale"] if scale is None else float(scale)
    lambda_h = DEFAULTS["lambda_h"] if lambda_h is None else float(lambda_h)
    lambda_J = DEFAULTS["lambda_J"] if lambda_J is None else float(lambda_J)
    try:
        lambda_g = DEFAULTS["lambda_g"] if lambda_g is None else float(lambda_g)  # plmc -lg (tools.py:252-253)
    except (TypeError, ValueError):
        raise ExternalToolError("lambda_g must be a number, got {!r}".format(lambda_g))
    if lambda_g < 0:
        raise ExternalToolError("lambda_g must not be negative, got {!r}".format(lambda_g))
    if iterations is None:
        iterations = DEFAULTS["iterations"]
    elif isinstance(iterations, str):
        if iterations.lower() != "max":
            raise ExternalToolError("iterations must be an integer or 'max', got {!r}".format(iterations))
        iterations = 0                       # until converged (tools.py:226-228 lets "max" through)
    iterations = int(iterations)
    # `cpu` is plmc's OpenMP thread count (tools.py:257-259; int or "max"): validated, otherwise ignored -- a
    # pipeline-wide `cpu: N` must not start N GPU ranks.  Multi-GPU is opt-in: `gpus=` here, or PLM_HIP_GPUS=N|max
    # for an unmodified pipeline (PLM_HIP_GPUS=cpu reads the cpu option as the GPU count).
    from evcouplings_amd import dist as _dist
    if cpu is not None and not (isinstance(cpu, str) and cpu.lower() == "max"):
        try:
            int(cpu)
        except (TypeError, ValueError):
            raise ExternalToolError("cpu must be an integer or 'max', got {!r}".format(cpu))
    try:
        n_gpus = 1 if distributed else _dist.resolve_gpu_count(cpu, gpus)
        solver = solver_from_env(solver)
    except (TypeError, ValueError) as exc:
        raise ExternalToolError(str(exc))
    except Exception as exc:       # library missing / no device: the same error type every other solver failure has
        raise ExternalToolError("HIP PLM solver failed: {}".format(exc))

    import time
    t_read = time.perf_counter()
    try:
        enc = alignment_io.encode_alignment(alignment, focus_seq=focus_seq, alphabet=alphabet)
    except alignment_io.AlignmentFormatError as exc:
        raise ExternalToolError("Could not read alignment {}: {}".format(alignment, exc))
    q = len(enc.alphabet)
    N, L = enc.msa.shape
    t_read = time.perf_counter() - t_read

    fit_kwargs = dict(q=q, ignore_gaps=bool(ignore_gaps), theta_id=theta, scale=scale, lambda_h=lambda_h, lambda_j=lambda_J,
                      max_iter=iterations, epsilon=DEFAULTS["epsilon"] if epsilon is None else float(epsilon),
                      lbfgs_m=lbfgs_m, callback=callback, joint=(solver == "joint"), lambda_group=lambda_g,
                      # PLM_CONV_* switches: explicit, or the environment variable PLM_HIP_CONVENTIONS
                      conventions=plm.conventions_from_env(conventions))
    t_lib = time.perf_counter()
    try:
        if distributed:
            res = _dist.fit_distributed(enc.msa, **fit_kwargs)
        elif n_gpus > 1:
            try:
                res = _dist.launch_fit(enc.msa, n_gpus, **fit_kwargs)     # N ranks under torch.distributed.run
            except _dist.LaunchError as exc:
                # the job never produced a result (torch / RCCL missing, a rank died): the single-GPU fit computes
                # the same thing, only slower -- still the HIP path, never a CPU path
                import warnings
                warnings.warn("multi-GPU fit on %d GPUs failed, running on one GPU instead: %s" % (n_gpus, exc))
                res = plm.fit(enc.msa, device=device, **fit_kwargs)
        else:
            res = plm.fit(enc.msa, device=device, **fit_kwargs)
    except Exception as exc:   # PlmError, ImportError (library missing), ...
        raise ExternalToolError("HIP PLM solver failed: {}".format(exc))

    t_lib = time.perf_counter() - t_lib
    t_write = time.perf_counter()
    # weights in original sequence order; invalid sequences get weight 0 (App. A field 4)
    weights = np.zeros(enc.n_total_seqs, dtype=np.float32)
    weights[enc.valid] = res["weights"]
    model_io.write_raw_ec_file(couplings_file, enc.index_list, enc.target_seq, res["cn"])
    if param_file is not None:
        # plmc -g writes a model over the alphabet without its gap character (the PABP example model of the
        # reference's notebooks has a 20-letter alphabet)
        model_alphabet = enc.alphabet[1:] if ignore_gaps else enc.alphabet
        model_io.write_model_file(
            param_file, L=L, q=len(model_alphabet), n_valid=enc.n_valid_seqs,
            n_invalid=enc.n_total_seqs - enc.n_valid_seqs,
            # plmc stores its iteration setting here (SURVEY.md App. A field 1); "max" (0 = until converged) is
            # recorded as the number of iterations actually taken
            num_iter=iterations if iterations > 0 else int(res["iters"]), theta=1.0 - theta, lambda_h=lambda_h, lambda_j=lambda_J, lambda_group=lambda_g,
            n_eff=res["n_eff"], alphabet=model_alphabet, weights=weights, target_seq=enc.target_seq,
            index_list=enc.index_list, fi=res["fi"], hi=res["hi"], fij=res["fij"], jij=res["jij"])
    if not _valid_file(couplings_file):
        raise ResourceError("HIP PLM solver returned no couplings: file={}".format(couplings_file))
    if param_file and not _valid_file(param_file):
        raise ResourceError("HIP PLM solver returned no parameter file: file={}".format(param_file))

    # where the wall-clock of this call went (bench.py's run_plmc_hip_default leg reports it)
    res.setdefault("seconds", {}).update(read_alignment=t_read, library=t_lib, write_files=time.perf_counter() - t_write)
    focus_name = focus_seq.split("/")[0] if focus_seq is not None else None
    log = format_plmc_log(focus_name, enc.focus_index, enc.n_valid_seqs, enc.n_total_seqs, L,
                          enc.n_total_sites, enc.region_start, res["n_eff"], res["status_msg"], res["table"],
                          theta=theta)
    in_focus = enc.focus_index is not None
    result = PlmcResult(
        couplings_file, param_file, iteration_dataframe(res["table"]),
        int(enc.focus_index) if in_focus else None, int(enc.n_valid_seqs), int(enc.n_total_seqs),
        int(L) if in_focus else None, int(enc.n_total_sites) if in_focus else None,
        int(enc.region_start) if in_focus else 1, float("%.1f" % res["n_eff"]), str(res["status_msg"]))
    return result, res, log


def run_plmc_hip(alignment, couplings_file, param_file=None, focus_seq=None, alphabet=None, theta=None,
                 scale=None, ignore_gaps=False, iterations=None, lambda_h=None, lambda_J=None,
                 lambda_g=None, cpu=None, binary=None, solver=None, gpus=None, conventions=None):
    """
    Drop-in for ``run_plmc`` (evcouplings/couplings/tools.py:126-130): same parameters (``theta``
    is the identity threshold, e.g. 0.8 -- no 1-theta round trip, tools.py:236-239; ``binary``
    is ignored; ``cpu``, plmc's thread count, is validated and ignored), same ``PlmcResult``, ``ResourceError`` for
    missing inputs/outputs and ``ExternalToolError`` for solver failures.  Three keyword extensions, all also
    selectable from the environment of an unmodified pipeline: ``solver`` ("vp" | "joint"; PLM_HIP_SOLVER),
    ``gpus`` (GPUs of this node to shard the fit over; PLM_HIP_GPUS) and ``conventions`` (PLM_CONV_* bits;
    PLM_HIP_CONVENTIONS).
    """
    result, _, _ = infer_to_files(alignment, couplings_file, param_file, focus_seq=focus_seq, alphabet=alphabet,
                                  theta=theta, scale=scale, ignore_gaps=ignore_gaps, iterations=iterations,
                                  lambda_h=lambda_h, lambda_J=lambda_J, lambda_g=lambda_g, cpu=cpu, solver=solver,
                                  gpus=gpus, conventions=conventions)
    return result
