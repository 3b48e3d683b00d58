"""
Hooks that put the HIP solver behind EVcouplings' couplings stage without editing the
reference.

``install()`` rebinds ``evcouplings.couplings.tools.run_plmc`` to ``run_plmc_hip``.
``infer_plmc`` (evcouplings/couplings/protocol.py:203-218) calls it through the module
attribute ``ct.run_plmc``, so the ``standard`` (:363) and ``complex`` (:480) protocols, their
``reuse_ecs`` short-circuit (:186-199), segment mapping, rescoring and post-processing all run
unchanged on top of the GPU inference.  ``uninstall()`` restores the subprocess path.

The alternative that needs no Python hook at all is the CLI shim: ``tools: plmc: bin/plmc_hip``.
``register_protocols()`` adds ``standard_hip`` / ``complex_hip`` to the reference's protocol registry
(couplings/protocol.py:922-931), selectable with the config key ``protocol``.

``install_all()`` additionally installs the optional GPU drop-ins of the rows SURVEY.md section 8f lists:
statistical energies behind ``CouplingsModel`` (``model_accel``), mean-field DCA (``mean_field``) and the
alignment statistics behind ``Alignment`` (``alignment_accel``).
"""
from evcouplings_amd import tools

_original = None


def install():
    """Route evcouplings' plmc calls to the HIP solver.  Returns the patched module."""
    global _original
    import evcouplings.couplings.tools as ct
    if ct.run_plmc is not tools.run_plmc_hip:
        _original = ct.run_plmc
        ct.run_plmc = tools.run_plmc_hip
    return ct


def uninstall():
    global _original
    import evcouplings.couplings.tools as ct
    if _original is not None:
        ct.run_plmc = _original
        _original = None
    return ct


def infer_plmc(**kwargs):
    """``evcouplings.couplings.protocol.infer_plmc`` with the HIP solver bound for the call (a hook that was already
    installed stays installed)."""
    import evcouplings.couplings.protocol as cp
    import evcouplings.couplings.tools as ct
    before = ct.run_plmc
    ct.run_plmc = tools.run_plmc_hip
    try:
        return cp.infer_plmc(**kwargs)
    finally:
        ct.run_plmc = before


# ---- protocol registry entries ------------------------------------------------------------------------------------
# `couplings/protocol.py:922-931` keeps its inference protocols in the dict PROTOCOLS and `run()` (:934-974) looks the
# config key `protocol` up there.  register_protocols() adds "standard_hip" and "complex_hip": the reference's own
# `standard` / `complex` functions with the HIP solver bound to `ct.run_plmc` for the duration of the call, so a
# pipeline config selects the GPU path with `protocol: standard_hip` -- and may carry three keys plmc does not have,
# `hip_solver` ("vp" | "joint"), `hip_gpus` (N | "max") and `hip_conventions` (PLM_CONV_* bits).  They are bound into
# the callable the call installs (the reference's infer_plmc passes a fixed argument list on); nothing process-wide
# changes: the environment is not touched, and whatever `ct.run_plmc` was before the call -- the subprocess path or a
# hook somebody installed with install() -- is what it is afterwards.
_OPTION_KEYS = {"hip_solver": "solver", "hip_gpus": "gpus", "hip_conventions": "conventions"}


def _wrapped(name):
    def protocol(**kwargs):
        import functools
        import evcouplings.couplings.protocol as cp
        import evcouplings.couplings.tools as ct
        options = {arg: kwargs[key] for key, arg in _OPTION_KEYS.items() if kwargs.get(key) is not None}
        before = ct.run_plmc
        ct.run_plmc = functools.partial(tools.run_plmc_hip, **options) if options else tools.run_plmc_hip
        try:
            return cp.PROTOCOLS[name](**kwargs)
        finally:
            ct.run_plmc = before
    protocol.__name__ = name + "_hip"
    protocol.__doc__ = "evcouplings.couplings.protocol.%s on the MI355X solver (evcouplings_amd.tools.run_plmc_hip)" % name
    return protocol


def register_protocols():
    """Add "standard_hip" / "complex_hip" to evcouplings.couplings.protocol.PROTOCOLS.  Returns the registry."""
    import evcouplings.couplings.protocol as cp
    for name in ("standard", "complex"):
        cp.PROTOCOLS.setdefault(name + "_hip", _wrapped(name))
    return cp.PROTOCOLS


def unregister_protocols():
    import evcouplings.couplings.protocol as cp
    for name in ("standard_hip", "complex_hip"):
        cp.PROTOCOLS.pop(name, None)


def install_all():
    """run_plmc + the N2 / N3 / N4 drop-ins (model energies, alignment statistics, mean-field DCA)."""
    from evcouplings_amd import alignment_accel, mean_field, model_accel
    install()
    model_accel.install()
    mean_field.install()
    alignment_accel.install()


def uninstall_all():
    from evcouplings_amd import alignment_accel, mean_field, model_accel
    alignment_accel.uninstall()
    mean_field.uninstall()
    model_accel.uninstall()
    uninstall()
