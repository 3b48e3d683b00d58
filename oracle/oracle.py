"""
ctypes front-end for the CPU oracle (oracle/plm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under evcouplings_amd/ may import it.

Parity status (details in plm_oracle.c): reweighting, frequencies and EC scoring are
pinned against the reference's own Python (tests/golden/); the PLM objective/gradient
and the L-BFGS driver are PARITY UNPINNED (plmc is not available).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ITER_CB = C.CFUNCTYPE(None, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                      C.c_double, C.c_double, C.c_void_p)


def build(force=False):
    """Compile both oracle libraries with gcc (no GPU needed)."""
    targets = [os.path.join(_HERE, n) for n in ("libplm_oracle.so", "libplm_oracle32.so")]
    src = os.path.join(_HERE, "plm_oracle.c")
    stale = force or any(
        (not os.path.exists(t)) or os.path.getmtime(t) < os.path.getmtime(src) for t in targets
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return targets


class Oracle:
    """precision='f64' (parity checker) or 'f32' (timed OpenMP baseline)."""

    def __init__(self, precision="f64"):
        build()
        if precision == "f64":
            self.lib = C.CDLL(os.path.join(_HERE, "libplm_oracle.so"))
            self.pre, self.real, self.creal = "plmo_", np.float64, C.c_double
        elif precision == "f32":
            self.lib = C.CDLL(os.path.join(_HERE, "libplm_oracle32.so"))
            self.pre, self.real, self.creal = "plmo32_", np.float32, C.c_float
        else:
            raise ValueError(precision)
        assert self._f("sizeof_real")() == np.dtype(self.real).itemsize

    def _f(self, name):
        return getattr(self.lib, self.pre + name)

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    @staticmethod
    def _msa(msa):
        msa = np.ascontiguousarray(msa, dtype=np.int8)
        assert msa.ndim == 2
        return msa

    def num_threads(self):
        return int(self._f("num_threads")())

    def set_num_threads(self, n):
        """libgomp may have read OMP_NUM_THREADS long before this library was loaded (torch bundles it)."""
        f = self._f("set_num_threads")
        f.argtypes = [C.c_int]
        f.restype = None
        f(int(n))

    def set_conventions(self, conv):
        """PLM_CONV_* bits of include/plm_hip.h (process-global in the library; reset with 0)."""
        f = self._f("set_conventions")
        f.argtypes = [C.c_int]
        f.restype = None
        f(int(conv))

    def set_lambda_group(self, lg):
        """group regulariser lambda_g * sum_{i<j} sqrt(|J_ij|^2 + 1e-8) of every later eval / fit (process-global; 0 = off)"""
        f = self._f("set_lambda_group")
        f.argtypes = [C.c_double]
        f.restype = None
        f(float(lg))

    def threshold(self, L, theta_id):
        f = self._f("threshold")
        f.argtypes = [C.c_int, C.c_double]
        return int(f(L, theta_id))

    def reweight(self, msa, theta_id):
        msa = self._msa(msa)
        N, L = msa.shape
        counts = np.zeros(N, dtype=np.int32)
        f = self._f("reweight")
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        rc = f(self._p(msa), N, L, float(theta_id), self._p(counts))
        if rc:
            raise RuntimeError("oracle reweight rc=%d" % rc)
        return counts

    def marginals(self, msa, w, q, pairs=True):
        msa = self._msa(msa)
        N, L = msa.shape
        w = np.ascontiguousarray(w, dtype=self.real)
        fi = np.zeros((L, q), dtype=self.real)
        fij = np.zeros((L * (L - 1) // 2, q, q), dtype=self.real) if pairs else None
        f = self._f("marginals")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        rc = f(self._p(msa), self._p(w), N, L, q, self._p(fi), self._p(fij))
        if rc:
            raise RuntimeError("oracle marginals rc=%d" % rc)
        return fi, fij

    def eval(self, msa, w, q, lambda_h, lambda_j, x):
        """-> (fx, nll, g).  x, g in the [h | J_ij (i<j)] layout."""
        msa = self._msa(msa)
        N, L = msa.shape
        w = np.ascontiguousarray(w, dtype=self.real)
        x = np.ascontiguousarray(x, dtype=self.real)
        assert x.size == L * q + L * (L - 1) // 2 * q * q
        g = np.zeros_like(x)
        fx, nll = C.c_double(0), C.c_double(0)
        f = self._f("eval")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                      C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        rc = f(self._p(msa), self._p(w), N, L, q, float(lambda_h), float(lambda_j), self._p(x),
               self._p(g), C.byref(fx), C.byref(nll))
        if rc:
            raise RuntimeError("oracle eval rc=%d" % rc)
        return fx.value, nll.value, g

    # ---- gap-ignoring mode (plmc -g): model over q-1 states, see plm_oracle.c eval_gaps ----
    def reweight_gaps(self, msa, theta_id):
        msa = self._msa(msa)
        N, L = msa.shape
        counts = np.zeros(N, dtype=np.int32)
        f = self._f("reweight_gaps")
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        rc = f(self._p(msa), N, L, float(theta_id), self._p(counts))
        if rc:
            raise RuntimeError("oracle reweight_gaps rc=%d" % rc)
        return counts

    def marginals_gaps(self, msa, w, q, pairs=True):
        msa = self._msa(msa)
        N, L = msa.shape
        qn = q - 1
        w = np.ascontiguousarray(w, dtype=self.real)
        fi = np.zeros((L, qn), dtype=self.real)
        fij = np.zeros((L * (L - 1) // 2, qn, qn), dtype=self.real) if pairs else None
        f = self._f("marginals_gaps")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        rc = f(self._p(msa), self._p(w), N, L, q, self._p(fi), self._p(fij))
        if rc:
            raise RuntimeError("oracle marginals_gaps rc=%d" % rc)
        return fi, fij

    def eval_gaps(self, msa, w, q, lambda_h, lambda_j, x):
        """x, g in the (q-1)-state layout; msa holds 0..q-1 with 0 = gap."""
        msa = self._msa(msa)
        N, L = msa.shape
        qn = q - 1
        w = np.ascontiguousarray(w, dtype=self.real)
        x = np.ascontiguousarray(x, dtype=self.real)
        assert x.size == L * qn + L * (L - 1) // 2 * qn * qn
        g = np.zeros_like(x)
        fx, nll = C.c_double(0), C.c_double(0)
        f = self._f("eval_gaps")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                      C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        rc = f(self._p(msa), self._p(w), N, L, q, float(lambda_h), float(lambda_j), self._p(x),
               self._p(g), C.byref(fx), C.byref(nll))
        if rc:
            raise RuntimeError("oracle eval_gaps rc=%d" % rc)
        return fx.value, nll.value, g

    def scores(self, jij, L, q):
        jij = np.ascontiguousarray(jij, dtype=self.real)
        fn = np.zeros((L, L))
        cn = np.zeros((L, L))
        f = self._f("scores")
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        rc = f(self._p(jij), L, q, self._p(fn), self._p(cn))
        if rc:
            raise RuntimeError("oracle scores rc=%d" % rc)
        return fn, cn

    def hamiltonians(self, seqs, q, x):
        """n x 3 (H, H_J, H_h) of the sequences under the canonical parameter vector x (model.py:25-60)."""
        seqs = self._msa(seqs)
        n, L = seqs.shape
        x = np.ascontiguousarray(x, dtype=self.real)
        out = np.zeros((n, 3))
        f = self._f("hamiltonians")
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        rc = f(self._p(seqs), n, L, q, self._p(x), self._p(out))
        if rc:
            raise RuntimeError("oracle hamiltonians rc=%d" % rc)
        return out

    def single_mutants(self, target, q, x):
        """L x q x 3 (dH, dH_J, dH_h) of every single substitution of `target` (model.py:63-109)."""
        target = np.ascontiguousarray(target, dtype=np.int8)
        L = target.size
        x = np.ascontiguousarray(x, dtype=self.real)
        out = np.zeros((L, q, 3))
        f = self._f("single_mutants")
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        rc = f(self._p(target), L, q, self._p(x), self._p(out))
        if rc:
            raise RuntimeError("oracle single_mutants rc=%d" % rc)
        return out

    def fit(self, msa, q, theta_id=0.8, scale=1.0, lambda_h=0.01, lambda_j=None, max_iter=100,
            epsilon=1e-3, lbfgs_m=6, want_fij=True, callback=None, ignore_gaps=False):
        """ignore_gaps=True: plmc -g semantics of plm_oracle.c eval_gaps; all outputs have q-1 states."""
        msa = self._msa(msa)
        N, L = msa.shape
        q_in = q
        q = q - 1 if ignore_gaps else q          # model states
        if lambda_j is None:
            lambda_j = 0.01 * (q - 1) * (L - 1)
        npair = L * (L - 1) // 2
        weights = np.zeros(N, dtype=self.real)
        fi = np.zeros((L, q), dtype=self.real)
        fij = np.zeros((npair, q, q), dtype=self.real) if want_fij else None
        x = np.zeros(L * q + npair * q * q, dtype=self.real)
        fn, cn = np.zeros((L, L)), np.zeros((L, L))
        neff, fx = C.c_double(0), C.c_double(0)
        iters, status, nevals = C.c_int(0), C.c_int(0), C.c_int(0)
        table = []

        def _cb(it, secs, cond, fxv, nll, nh, ne, user):
            table.append((it, secs, cond, fxv, nll, nh, ne))
            if callback is not None:
                callback(it, secs, cond, fxv, nll, nh, ne)

        cb = ITER_CB(_cb)
        f = self._f("fit2")
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                      C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p,
                      C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                      C.POINTER(C.c_double), ITER_CB, C.c_void_p]
        rc = f(self._p(msa), N, L, q_in, float(theta_id), float(scale), float(lambda_h),
               float(lambda_j), int(max_iter), float(epsilon), int(lbfgs_m), int(bool(ignore_gaps)), self._p(weights),
               C.byref(neff), self._p(fi), self._p(fij), self._p(x), self._p(fn), self._p(cn),
               C.byref(iters), C.byref(status), C.byref(nevals), C.byref(fx), cb, None)
        if rc:
            raise RuntimeError("oracle fit rc=%d" % rc)
        return dict(weights=weights, n_eff=neff.value, fi=fi, fij=fij, x=x,
                    hi=x[:L * q].reshape(L, q), jij=x[L * q:].reshape(npair, q, q), fn=fn, cn=cn,
                    iters=iters.value, status=status.value, nevals=nevals.value, fx=fx.value,
                    table=table, lambda_j=lambda_j)
