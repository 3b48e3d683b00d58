"""
Stubs that let the REFERENCE package (read-only at /root/reference, build container only)
import without its optional third-party dependencies, so its own couplings protocol can be
driven on top of our backend in tests.  Nothing here is
used by the product; on the GPU box /root/reference does not exist and the tests that need
it are skipped.
"""
import os
import sys
import types

REFERENCE = os.environ.get("EVC_REFERENCE", "/root/reference")


class _Anything(types.ModuleType):
    """module whose every attribute is a harmless callable/class"""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        class _Dummy:
            def __init__(self, *a, **k):
                pass

            def __call__(self, *a, **k):
                return self

            def __getattr__(self, n):
                return _Dummy()

        _Dummy.__name__ = name
        return _Dummy


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE, "evcouplings"))


def install():
    """Insert the stubs and put the reference on sys.path.  Idempotent."""
    if not reference_available():
        raise RuntimeError("reference not available at %s" % REFERENCE)
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    def have(mod):
        try:
            __import__(mod)
            return True
        except Exception:
            return False

    if not have("numba"):
        nb = types.ModuleType("numba")
        nb.jit = nb.njit = jit
        nb.prange = range
        sys.modules["numba"] = nb
    if not have("ruamel.yaml"):
        import yaml

        class _YAML:   # the subset evcouplings/utils/config.py uses
            def __init__(self, typ=None, pure=False):
                self.default_flow_style = False

            def load(self, stream):
                return yaml.safe_load(stream)

            def dump(self, data, stream=None):
                return yaml.safe_dump(_plain(data), stream, default_flow_style=False)

        def _plain(x):
            if isinstance(x, dict):
                return {k: _plain(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return [_plain(v) for v in x]
            return x

        ry = types.ModuleType("ruamel")
        ryy = types.ModuleType("ruamel.yaml")
        ryy.YAML = _YAML
        ryy.safe_load = yaml.safe_load
        ryy.safe_dump = yaml.safe_dump
        ryy.load = lambda s, Loader=None: yaml.safe_load(s)
        ryy.dump = lambda d, f=None, **k: yaml.safe_dump(_plain(d), f, default_flow_style=False)
        ryy.RoundTripLoader = ryy.RoundTripDumper = ryy.Loader = ryy.Dumper = None
        comments = types.ModuleType("ruamel.yaml.comments")

        class CommentedBase:
            pass

        comments.CommentedBase = CommentedBase
        ryy.comments = comments
        ry.yaml = ryy
        sys.modules.update({"ruamel": ry, "ruamel.yaml": ryy, "ruamel.yaml.comments": comments})
    if not have("billiard"):
        import multiprocessing
        b = types.ModuleType("billiard")
        for k in dir(multiprocessing):
            if not k.startswith("_"):
                setattr(b, k, getattr(multiprocessing, k))
        sys.modules["billiard"] = b
    for name in ("bokeh", "bokeh.core", "bokeh.core.properties", "bokeh.models", "bokeh.io", "bokeh.plotting",
                 "bokeh.palettes", "bokeh.layouts", "Bio", "Bio.PDB", "Bio.PDB.binary_cif", "Bio.PDB.Polypeptide",
                 "Bio.PDB.MMCIF2Dict", "mmtf", "msgpack_numpy"):
        if name.split(".")[0] not in ("Bio",) or not have("Bio"):
            if name not in sys.modules and not have(name):
                sys.modules[name] = _Anything(name)
    return REFERENCE
