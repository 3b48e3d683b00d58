"""
Python host API over the C ABI of libplm_hip.so: numpy in, numpy out.

This is the in-process replacement for what the reference obtains from the plmc child
process (evcouplings/couplings/tools.py:202-307): sequence weights, frequencies, the
fitted fields/couplings and the CN scores.  All arithmetic happens in the HIP library on
an MI355X; nothing here computes on the CPU and there is no fallback path.
"""
import ctypes as C

import numpy as np

from evcouplings_amd import _lib
from evcouplings_amd._lib import PlmProblem, PlmResult, check


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _msa(msa):
    msa = np.ascontiguousarray(msa, dtype=np.int8)
    if msa.ndim != 2:
        raise ValueError("msa must be a 2-D (N, L) int8 matrix")
    return msa


def n_params(L, q):
    return L * q + L * (L - 1) // 2 * q * q


def default_lambda_j(L, q, base=0.01):
    """lambda_J * (q-1) * (L-1), the scaling of evcouplings/couplings/protocol.py:159-179."""
    return base * (q - 1) * (L - 1)


def device_count():
    return _lib.load().plm_device_count()


# convention switches (include/plm_hip.h PLM_CONV_*): selectable conventions of plmc that cannot be verified here
CONV_THRESHOLD_F32 = 32        # App. D-1: float32 evaluation of the cluster threshold
CONV_G_GAPS_IDENTICAL = 64     # -g: gap-gap positions count as identical in reweighting
CONV_G_UNGAPPED_LENGTH = 128   # -g: threshold on the positions where both sequences are ungapped
CONV_G_FREQ_TOTAL = 256        # -g: frequencies normalised by N_eff instead of the ungapped weight
CONV_FN_NO_GAP = 512           # App. D-3: Frobenius norm without the gap state
CONV_MASK = 32 | 64 | 128 | 256 | 512


def conventions_from_env(conventions=None):
    """Explicit value, else the environment variable PLM_HIP_CONVENTIONS (an integer, e.g. "320" or "0x140"), else 0:
    lets an unmodified pipeline select conventions for the run_plmc drop-in."""
    import os
    if conventions is None:
        conventions = int(os.environ.get("PLM_HIP_CONVENTIONS", "0"), 0)
    conventions = int(conventions)
    if conventions & ~CONV_MASK:
        raise ValueError("unknown convention bits in %d (known: %d)" % (conventions, CONV_MASK))
    return conventions


def reweight(msa, theta_id=0.8, ignore_gaps=False, conventions=0):
    """Cluster sizes (incl. self) at identity >= theta_id; twin of alignment.py:1193-1233.
    ignore_gaps / conventions: plmc -g semantics and PLM_CONV_* switches."""
    lib = _lib.load()
    msa = _msa(msa)
    counts = np.zeros(msa.shape[0], dtype=np.int32)
    flags = (FLAG_IGNORE_GAPS if ignore_gaps else 0) | int(conventions)
    check(lib.plm_reweight_ex(_ptr(msa), msa.shape[0], msa.shape[1], float(theta_id), flags, _ptr(counts)))
    return counts


def marginals(msa, weights, q, pairs=True):
    """f_i (L,q) and f_ij (L(L-1)/2,q,q) for i<j; twin of alignment.py:1079-1153."""
    lib = _lib.load()
    msa = _msa(msa)
    N, L = msa.shape
    w = np.ascontiguousarray(weights, dtype=np.float32)
    fi = np.zeros((L, q), dtype=np.float32)
    fij = np.zeros((L * (L - 1) // 2, q, q), dtype=np.float32) if pairs else None
    check(lib.plm_marginals(_ptr(msa), _ptr(w), N, L, q, _ptr(fi), _ptr(fij)))
    return fi, fij


def evaluate(msa, weights, q, lambda_h, lambda_j, x):
    """Objective, its unregularised part and the gradient at x (canonical layout)."""
    lib = _lib.load()
    msa = _msa(msa)
    N, L = msa.shape
    w = np.ascontiguousarray(weights, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.size != n_params(L, q):
        raise ValueError("x has %d entries, expected %d" % (x.size, n_params(L, q)))
    g = np.zeros_like(x)
    fx, nll = C.c_double(0), C.c_double(0)
    check(lib.plm_eval(_ptr(msa), _ptr(w), N, L, q, float(lambda_h), float(lambda_j), _ptr(x),
                       C.byref(fx), C.byref(nll), _ptr(g)))
    return fx.value, nll.value, g


def scores(jij, L, q, conventions=0):
    """FN and CN (APC) matrices from i<j coupling blocks; twin of model.py:179-233, 744-827.
    conventions & CONV_FN_NO_GAP: state 0 left out of the Frobenius norm."""
    lib = _lib.load()
    jij = np.ascontiguousarray(jij, dtype=np.float32)
    fn = np.zeros((L, L), dtype=np.float32)
    cn = np.zeros((L, L), dtype=np.float32)
    check(lib.plm_scores_ex(_ptr(jij), L, q, int(conventions), _ptr(fn), _ptr(cn)))
    return fn, cn


def _canonical(hi, jij, L, q):
    """h [L][q] followed by the i<j coupling blocks [q][q]: the canonical parameter vector."""
    hi = np.ascontiguousarray(hi, dtype=np.float32).reshape(L * q)
    jij = np.ascontiguousarray(jij, dtype=np.float32).reshape(L * (L - 1) // 2 * q * q)
    return np.concatenate([hi, jij])


def alignment_stats(msa, gap_state=0, query=None, device=0):
    """Per-sequence gap counts, per-column gap counts and (if `query` is given) identities of every sequence to the
    query, as int32 arrays (n,), (L,), (n,) -- the raw counts behind Alignment.count(gap, axis=...) and
    identities_to_seq of evcouplings/align/alignment.py:707-747, 1157-1190."""
    lib = _lib.load()
    msa = _msa(msa)
    n, L = msa.shape
    seq_gaps, col_gaps = np.zeros(n, np.int32), np.zeros(L, np.int32)
    ident = None
    if query is not None:
        query = np.ascontiguousarray(query, dtype=np.int8).reshape(L)
        ident = np.zeros(n, np.int32)
    check(lib.plm_alignment_stats(_ptr(msa), n, L, int(gap_state), _ptr(query), _ptr(seq_gaps), _ptr(col_gaps),
                                  _ptr(ident), int(device), None))
    return seq_gaps, col_gaps, ident


def hamiltonians(seqs, q, hi, jij, device=0):
    """
    Statistical energies of sequences under a model: n x 3 float64 (H, H_J, H_h) with
    H_J = sum_{i<j} J_ij(x_i, x_j), H_h = sum_i h_i(x_i) -- the return value of the reference's
    `_hamiltonians(sequences, J_ij, h_i)` (couplings/model.py:25-60), computed by the forward one-hot GEMM.
    seqs: n x L integer states; hi: L x q; jij: the i<j blocks [L(L-1)/2][q][q].
    """
    lib = _lib.load()
    seqs = _msa(seqs)
    n, L = seqs.shape
    out = np.zeros((n, 3))
    check(lib.plm_hamiltonians(_ptr(seqs), n, L, q, _ptr(_canonical(hi, jij, L, q)), device, None, _ptr(out)))
    return out


def potentials(seqs, q, hi, jij, device=0):
    """HJ[s, i, a] = sum_{j != i} J_ij(a, x_sj): n x L x q float32 (coupling part of every conditional)."""
    lib = _lib.load()
    seqs = _msa(seqs)
    n, L = seqs.shape
    out = np.zeros((n, L, q), dtype=np.float32)
    check(lib.plm_potentials(_ptr(seqs), n, L, q, _ptr(_canonical(hi, jij, L, q)), device, None, _ptr(out)))
    return out


def single_mutant_matrix(target, q, hi, jij, device=0):
    """
    Energy differences of every single substitution of `target`: L x q x 3 float64 (dH, dH_J, dH_h), the
    return value of the reference's `_single_mutant_hamiltonians` (couplings/model.py:63-109).
    """
    target = np.ascontiguousarray(target, dtype=np.int8).reshape(1, -1)
    L = target.shape[1]
    hj = potentials(target, q, hi, jij, device=device)[0].astype(np.float64)
    hi = np.asarray(hi, dtype=np.float64).reshape(L, q)
    rows = np.arange(L)
    dj = hj - hj[rows, target[0]][:, None]
    dh = hi - hi[rows, target[0]][:, None]
    return np.stack([dj + dh, dj, dh], axis=2)


def mean_field(msa, q=21, theta_id=0.8, pseudo_count=0.5, device=0, want_fij=True, want_full=True, want_di=True):
    """
    Mean-field direct coupling analysis of an alignment: the arithmetic of the reference's
    `MeanFieldDCA.fit` (couplings/mean_field.py:163-222) and `direct_information` (:842-893) on the GPU.
    Returns weights, n_eff, raw fi [L,q] / fij [pairs,q,q], fields hi [L,q] (float64), couplings as the i<j blocks
    jij [pairs,q,q] (float32) and, if want_full, the dense jij_full [L,L,q,q] (float64, diagonal blocks included,
    last row/column of every block zero), and di [L,L] (float64).
    """
    lib = _lib.load()
    msa = _msa(msa)
    N, L = msa.shape
    npair = L * (L - 1) // 2
    out = {
        "weights": np.zeros(N, np.float32), "fi": np.zeros((L, q), np.float32), "hi": np.zeros((L, q)),
        "jij": np.zeros((npair, q, q), np.float32),
    }
    if want_fij:
        out["fij"] = np.zeros((npair, q, q), np.float32)
    if want_full:
        out["jij_full"] = np.zeros((L, L, q, q))
    if want_di:
        out["di"] = np.zeros((L, L))
    res = _lib.PlmMfResult()
    for name in ("weights", "fi", "fij", "hi", "jij_full", "jij", "di"):
        if name in out:
            setattr(res, name, out[name].ctypes.data)
    check(lib.plm_meanfield(_ptr(msa), N, L, q, float(theta_id), float(pseudo_count), device, None, C.byref(res)))
    out["n_eff"] = float(res.n_eff)
    out["theta_id"], out["pseudo_count"] = theta_id, pseudo_count
    return out


def direct_information(jij_full, fi, device=0):
    """DI of every pair from dense couplings [L,L,q,q] and frequencies [L,q] (both float64): the reference's
    `direct_information(J_ij, f_i)` (couplings/mean_field.py:842-893).  Returns [L,L] float64."""
    lib = _lib.load()
    jij_full = np.ascontiguousarray(jij_full, dtype=np.float64)
    fi = np.ascontiguousarray(fi, dtype=np.float64)
    L, q = fi.shape
    assert jij_full.shape == (L, L, q, q)
    di = np.zeros((L, L))
    check(lib.plm_direct_information(_ptr(jij_full), _ptr(fi), L, q, device, None, _ptr(di)))
    return di


FLAG_IGNORE_GAPS = 2
FLAG_SHARDED_STATE = 4
FLAG_PRECOND = 8
FLAG_JOINT_LBFGS = 16
FLAG_COMPACT_GAPS = 1024


class _IterationCallback:
    """The plm_iter_cb of one fit: collects the iteration table, forwards to the user's callback, and turns an exception
    raised while Python code runs inside the callback -- the user's own, or the SystemExit / KeyboardInterrupt of a signal
    handler (evcouplings/utils/pipeline.py:476-545 installs handlers that call sys.exit; Python runs a pending handler
    the next time the main thread executes bytecode, which during a fit is here) -- into a cancellation: the library
    stops after this iteration (status "interrupted"), and `reraise()` raises the exception where the fit was called.
    ctypes would otherwise print and swallow it, and the fit would run on."""

    def __init__(self, callback=None):
        self.table, self.callback, self.pending = [], callback, None
        self.cfunc = _lib.ITER_CB(self._call)

    def _call(self, it, secs, cond, fx, nll, nh, ne, user):
        try:
            self.table.append((it, secs, cond, fx, nll, nh, ne))
            if self.callback is not None:
                self.callback(it, secs, cond, fx, nll, nh, ne)
            return 0
        except BaseException as exc:      # incl. SystemExit / KeyboardInterrupt: never let one cross the C boundary
            self.pending = exc
            return 1

    def reraise(self):
        if self.pending is not None:
            exc, self.pending = self.pending, None
            raise exc


def _wrap_collective(collective):
    """python callable(op, send_ptr, recv_ptr, send_counts, recv_counts, n_shards, shard) -> 0  =>  C callback"""
    def _cb(op, send, recv, scounts, rcounts, n, shard, user):
        try:
            return int(collective(int(op), send, recv, [int(scounts[k]) for k in range(n)],
                                  [int(rcounts[k]) for k in range(n)] if op in (_lib.COLL_ALLTOALL, _lib.COLL_BROADCAST) else None,
                                  int(n), int(shard)))
        except Exception as exc:   # never let an exception cross the C boundary
            import sys
            print("plm collective failed: %r" % (exc,), file=sys.stderr)
            return 1
    return _lib.COLLECTIVE_CB(_cb)


def _embed_gaps(x, L, q):
    """(q-1)-state canonical vector -> q-state layout with zeros for state 0."""
    qn = q - 1
    x = np.asarray(x, dtype=np.float32)
    npair = L * (L - 1) // 2
    out_h = np.zeros((L, q), np.float32)
    out_h[:, 1:] = x[:L * qn].reshape(L, qn)
    out_j = np.zeros((npair, q, q), np.float32)
    out_j[:, 1:, 1:] = x[L * qn:].reshape(npair, qn, qn)
    return np.concatenate([out_h.ravel(), out_j.ravel()])


def _strip_gaps(x, L, q):
    """q-state canonical vector -> (q-1)-state layout (drops every entry that involves state 0)."""
    npair = L * (L - 1) // 2
    h = x[:L * q].reshape(L, q)[:, 1:]
    j = x[L * q:].reshape(npair, q, q)[:, 1:, 1:]
    return np.concatenate([h.ravel(), j.ravel()]).astype(np.float32)


def _problem(msa, q, theta_id, scale, lambda_h, lambda_j, max_iter, epsilon, lbfgs_m, n_shards, shard,
             ignore_gaps=False, sharded_state=False, precond=False, joint=False, conventions=0, lambda_group=0.0):
    N, L = msa.shape
    p = PlmProblem()
    p.n_seqs, p.n_sites, p.n_states = N, L, q
    p.msa = msa.ctypes.data
    p.theta_id, p.scale = float(theta_id), float(scale)
    p.lambda_h, p.lambda_j = float(lambda_h), float(lambda_j)
    p.lambda_group = float(lambda_group or 0.0)
    p.max_iter, p.epsilon, p.lbfgs_m = int(max_iter), float(epsilon), int(lbfgs_m)
    p.n_shards, p.shard = int(n_shards), int(shard)
    p.flags = ((FLAG_IGNORE_GAPS if ignore_gaps else 0) | (FLAG_SHARDED_STATE if sharded_state else 0) |
               (FLAG_PRECOND if precond else 0) | (FLAG_JOINT_LBFGS if joint else 0) | (int(conventions) & CONV_MASK))
    return p


def fit(msa, q=21, theta_id=0.8, scale=1.0, lambda_h=0.01, lambda_j=None, max_iter=100,
        epsilon=1e-3, lbfgs_m=6, device=0, stream=0, callback=None, n_shards=1, shard=0,
        exchange=None, want_fij=True, ignore_gaps=False, collective=None, precond=False, joint=False, conventions=0,
        rccl_id=None, lambda_group=0.0):
    """
    Whole couplings inference: reweight -> marginals -> L-BFGS -> scores.

    callback(iter, secs, cond, fx, nll, norm_h, norm_e) is called once per iteration.
    exchange(dev_ptr, bytes_per_shard, n_shards, shard) -> 0 implements the all-gather of
    the site-sharded gradient slabs (see evcouplings_amd.dist) and is required iff n_shards > 1
    in the replicated mode; pass `collective` instead to run the sharded-state mode
    (parameters, gradient and optimiser state split across the shards, see evcouplings_amd.dist), or `rccl_id`
    (the bytes of rccl_unique_id() made on rank 0) to run that mode with the collectives issued by the library
    itself over RCCL on its own stream (one process per GPU, rank = shard).
    ignore_gaps=True is plmc -g (tools.py:222-224): state 0 is excluded from the model and every
    returned array has q-1 states (fi, hi: (L, q-1); fij, jij: (pairs, q-1, q-1)).
    joint=True optimises fields and couplings jointly with L-BFGS as libLBFGS-based plmc does
    (PLM_FLAG_JOINT_LBFGS) instead of the default variable projection (fields solved by Newton for every trial
    couplings, ~10-20x fewer iterations to the same optimum); precond=True gives L-BFGS a diagonal initial Hessian
    (PLM_FLAG_PRECOND).  conventions: PLM_CONV_* bits (CONV_* above), the selectable conventions of plmc.
    lambda_group: run_plmc's lambda_g (plmc -lg), the group regulariser lambda_group * sum_{i<j} sqrt(|J_ij|^2 + 1e-8).
    Returns a dict of numpy arrays and scalars.
    """
    lib = _lib.load()
    msa = _msa(msa)
    N, L = msa.shape
    if lambda_j is None:
        lambda_j = default_lambda_j(L, q - 1 if ignore_gaps else q)
    npair = L * (L - 1) // 2
    qo = q - 1 if ignore_gaps else q       # -g: the library returns the (q-1)-state arrays (PLM_FLAG_COMPACT_GAPS)
    out = dict(
        weights=np.zeros(N, np.float32), fi=np.zeros((L, qo), np.float32),
        fij=np.zeros((npair, qo, qo), np.float32) if want_fij else None,
        hi=np.zeros((L, qo), np.float32), jij=np.zeros((npair, qo, qo), np.float32),
        fn=np.zeros((L, L), np.float32), cn=np.zeros((L, L), np.float32))
    res = PlmResult()
    for k in ("weights", "fi", "fij", "hi", "jij", "fn", "cn"):
        setattr(res, k, None if out[k] is None else out[k].ctypes.data)
    icb = _IterationCallback(callback)
    cb, table = icb.cfunc, icb.table
    if exchange is not None:
        xcb = _lib.EXCHANGE_CB(lambda buf, nbytes, ns, sh, user: int(exchange(buf, nbytes, ns, sh)))
    else:
        xcb = C.cast(None, _lib.EXCHANGE_CB)
    prob = _problem(msa, q, theta_id, scale, lambda_h, lambda_j, max_iter, epsilon, lbfgs_m,
                    n_shards, shard, ignore_gaps, sharded_state=collective is not None or rccl_id is not None,
                    precond=precond, joint=joint, conventions=conventions, lambda_group=lambda_group)
    if ignore_gaps:
        prob.flags |= FLAG_COMPACT_GAPS
    if rccl_id is not None:
        idbuf = C.create_string_buffer(bytes(rccl_id), RCCL_ID_BYTES)
        check(lib.plm_fit_sharded_rccl(C.byref(prob), C.byref(res), int(device), C.c_void_p(int(stream) or None), cb,
                                       None, idbuf))
    elif collective is not None:
        ccb = _wrap_collective(collective)
        check(lib.plm_fit_sharded(C.byref(prob), C.byref(res), int(device), C.c_void_p(int(stream) or None), cb,
                                  None, ccb, None))
    else:
        check(lib.plm_fit(C.byref(prob), C.byref(res), int(device), C.c_void_p(int(stream) or None), cb,
                          None, xcb, None))
    icb.reraise()      # an exception (or a signal handler's SystemExit) met inside the iteration callback
    out.update(
        n_eff=float(res.n_eff), iters=int(res.iters_done), n_evals=int(res.n_evals),
        status=int(res.status), status_msg=res.status_msg.decode("ascii", "replace"),
        fx=float(res.fx), table=table, lambda_j=float(lambda_j),
        seconds=dict(reweight=res.seconds_reweight, marginals=res.seconds_marginals,
                     optimize=res.seconds_optimize, total=res.seconds_total))
    return out


RCCL_ID_BYTES = 128


def rccl_unique_id():
    """plm_rccl_unique_id: the communicator id rank 0 creates and every rank attaches (128 bytes)."""
    buf = C.create_string_buffer(RCCL_ID_BYTES)
    check(_lib.load().plm_rccl_unique_id(buf))
    return buf.raw


def rccl_version():
    """NCCL version code of the RCCL the library resolved at run time (0: none found)."""
    return int(_lib.load().plm_rccl_runtime_version())


def rccl_probe(rccl_id, nranks, rank, device=0, stream=0):
    """plm_rccl_probe: every rank forms the communicator, exchanges an all-to-all and an all-reduce, destroys it; raises
    on this rank if anything failed"""
    idbuf = C.create_string_buffer(bytes(rccl_id), RCCL_ID_BYTES)
    check(_lib.load().plm_rccl_probe(idbuf, int(nranks), int(rank), int(device), C.c_void_p(int(stream) or None)))


def rccl_probe_local(nranks, device=0, stream=0):
    """plm_rccl_probe_local: the steps of the probe that can fail on one rank alone (no communicator call)"""
    check(_lib.load().plm_rccl_probe_local(int(nranks), int(device), C.c_void_p(int(stream) or None)))


def rccl_selftest(device=0, stream=0):
    """every collective of the sharded-state mode on a one-rank communicator; raises on any failure"""
    check(_lib.load().plm_rccl_selftest(int(device), C.c_void_p(int(stream) or None)))


class PlmContext:
    """Alignment resident in HBM; step-wise access for benchmarks and the multi-GPU host."""

    def __init__(self, msa, q=21, theta_id=0.8, scale=1.0, lambda_h=0.01, lambda_j=None,
                 max_iter=100, epsilon=1e-3, lbfgs_m=6, device=0, stream=0, n_shards=1, shard=0,
                 ignore_gaps=False, sharded_state=False, precond=False, joint=False, conventions=0, lambda_group=0.0):
        self.lib = _lib.load()
        msa = _msa(msa)
        self.N, self.L = msa.shape
        self.q = q
        self.ignore_gaps = bool(ignore_gaps)
        self.qm = q - 1 if ignore_gaps else q      # model states (layout of x, g, fi, fij at this API)
        self.lambda_j = default_lambda_j(self.L, self.qm) if lambda_j is None else lambda_j
        prob = _problem(msa, q, theta_id, scale, lambda_h, self.lambda_j, max_iter, epsilon, lbfgs_m,
                        n_shards, shard, ignore_gaps, sharded_state, precond, joint, conventions, lambda_group)
        self._h = C.c_void_p()
        check(self.lib.plm_ctx_create(C.byref(prob), int(device), C.c_void_p(int(stream) or None),
                                      C.byref(self._h)))
        self._keep = []

    def close(self):
        if self._h:
            self.lib.plm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_exchange(self, exchange):
        cb = _lib.EXCHANGE_CB(lambda buf, nbytes, ns, sh, user: int(exchange(buf, nbytes, ns, sh)))
        self._keep.append(cb)
        check(self.lib.plm_ctx_set_exchange(self._h, cb, None))

    def set_collective(self, collective):
        cb = _wrap_collective(collective)
        self._keep.append(cb)
        check(self.lib.plm_ctx_set_collective(self._h, cb, None))

    def attach_rccl(self, rccl_id):
        """collectives of the sharded-state mode from the library itself (RCCL on the context's stream); a collective
        call: every rank, with the id rank 0 made (rccl_unique_id)"""
        idbuf = C.create_string_buffer(bytes(rccl_id), RCCL_ID_BYTES)
        check(self.lib.plm_ctx_attach_rccl(self._h, idbuf))

    def set_options(self, max_iter=-1, epsilon=-1.0, lbfgs_m=-1):
        check(self.lib.plm_ctx_set_options(self._h, int(max_iter), float(epsilon), int(lbfgs_m)))

    def _set_max_iter(self, k):
        self.set_options(max_iter=k)

    def native_size(self):
        return int(self.lib.plm_ctx_native_size(self._h))

    def reweight(self):
        check(self.lib.plm_ctx_reweight(self._h))
        return self.weights()

    def set_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.float32)
        assert w.size == self.N
        check(self.lib.plm_ctx_set_weights(self._h, _ptr(w)))

    def weights(self):
        w = np.zeros(self.N, np.float32)
        counts = np.zeros(self.N, np.int32)
        neff = C.c_float(0)
        check(self.lib.plm_ctx_get_weights(self._h, _ptr(w), _ptr(counts), C.byref(neff)))
        return w, counts, neff.value

    def marginals(self, pairs=True):
        fi = np.zeros((self.L, self.q), np.float32)
        fij = np.zeros((self.L * (self.L - 1) // 2, self.q, self.q), np.float32) if pairs else None
        check(self.lib.plm_ctx_marginals(self._h, _ptr(fi), _ptr(fij)))
        if self.ignore_gaps:
            fi = fi[:, 1:].copy()
            fij = None if fij is None else fij[:, 1:, 1:].copy()
        return fi, fij

    def set_x(self, x=None):
        if x is not None:
            x = np.ascontiguousarray(x, dtype=np.float32)
            assert x.size == n_params(self.L, self.qm)
            if self.ignore_gaps:
                x = _embed_gaps(x, self.L, self.q)
        check(self.lib.plm_ctx_set_x(self._h, _ptr(x)))

    def get_x(self):
        x = np.zeros(n_params(self.L, self.q), np.float32)
        check(self.lib.plm_ctx_get_x(self._h, _ptr(x)))
        return _strip_gaps(x, self.L, self.q) if self.ignore_gaps else x

    def get_g(self):
        g = np.zeros(n_params(self.L, self.q), np.float32)
        check(self.lib.plm_ctx_get_g(self._h, _ptr(g)))
        return _strip_gaps(g, self.L, self.q) if self.ignore_gaps else g

    def eval(self, sync=True):
        if not sync:
            check(self.lib.plm_ctx_eval(self._h, None, None))
            return None
        fx, nll = C.c_double(0), C.c_double(0)
        check(self.lib.plm_ctx_eval(self._h, C.byref(fx), C.byref(nll)))
        return fx.value, nll.value

    def optimize(self, callback=None):
        res = PlmResult()
        icb = _IterationCallback(callback)
        table = icb.table
        check(self.lib.plm_ctx_optimize(self._h, icb.cfunc, None, C.byref(res)))
        icb.reraise()
        return dict(iters=int(res.iters_done), n_evals=int(res.n_evals), status=int(res.status),
                    status_msg=res.status_msg.decode("ascii", "replace"), fx=float(res.fx),
                    seconds=float(res.seconds_optimize), table=table)

    def scores(self):
        fn = np.zeros((self.L, self.L), np.float32)
        cn = np.zeros((self.L, self.L), np.float32)
        check(self.lib.plm_ctx_scores(self._h, _ptr(fn), _ptr(cn)))
        return fn, cn

    def solver_stats(self):
        """field-solver statistics of the last optimize() on this context (plm_ctx_solver_stats)"""
        out = np.zeros(_lib.S_COUNT, np.float64)
        check(self.lib.plm_ctx_solver_stats(self._h, _ptr(out)))
        ev, gv = max(1.0, out[0]), max(1.0, out[4])
        return {"evaluations": int(out[0]), "field_ms_per_evaluation": out[1] / ev, "passes_per_evaluation": out[2] / ev,
                "chains_continued_by_host": int(out[3]),
                # the two GEMMs as the fit ran them (HIP events inside the fit; plain arithmetic only)
                "gemm_evaluations": int(out[4]), "forward_ms_per_evaluation": out[5] / gv, "backward_ms_per_evaluation": out[6] / gv}

    def time_field_positions(self, reps=5):
        """ms of one Hessian position and of the closing (residual-writing) position of the field solver's chain on this
        context's site blocks (plm_ctx_time_field_positions; call time_kernels first)"""
        ms = np.zeros(2, np.float32)
        check(self.lib.plm_ctx_time_field_positions(self._h, int(reps), _ptr(ms)))
        return {"hessian_position": float(ms[0]), "closing_position": float(ms[1])}

    def time_kernels(self, reps=5):
        ms = np.zeros(_lib.K_COUNT, np.float32)
        check(self.lib.plm_ctx_time_kernels(self._h, int(reps), _ptr(ms)))
        names = ["expand", "forward", "backward", "assemble", "total", "reweight", "fields", "forward_accurate", "lbfgs_vector"]
        return dict(zip(names, ms.tolist()))
