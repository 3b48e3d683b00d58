"""
Hooks that put the HIP solver behind EVcouplings' couplings stage without editing the
reference.

``install()`` rebinds ``evcouplings.couplings.tools.run_plmc`` to ``run_plmc_hip``.
``infer_plmc`` (evcouplings/couplings/protocol.py:203-218) calls it through the module
attribute ``ct.run_plmc``, so the ``standard`` (:363) and ``complex`` (:480) protocols, their
``reuse_ecs`` short-circuit (:186-199), segment mapping, rescoring and post-processing all run
unchanged on top of the GPU inference.  ``uninstall()`` restores the subprocess path.

The alternative that needs no Python hook at all is the CLI shim: ``tools: plmc: bin/plmc_hip``.

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
    """``evcouplings.couplings.protocol.infer_plmc`` with the HIP solver installed for the call."""
    import evcouplings.couplings.protocol as cp
    install()
    try:
        return cp.infer_plmc(**kwargs)
    finally:
        uninstall()


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
