"""
evcouplings_amd -- MI355X-native pseudo-likelihood Potts solver behind the
``run_plmc`` / ``infer_plmc`` boundary of EVcouplings (evcouplings/couplings/tools.py:126,
evcouplings/couplings/protocol.py:56).  See DESIGN.md.
"""
__version__ = "0.1.0"
