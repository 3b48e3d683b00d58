"""
Site-sharded inference across the GPUs of one node.

One process per GPU.  Shard r owns a contiguous range of 16-site column blocks; its forward
and backward GEMMs need no communication.  Collectives are issued by the C library through
callbacks implemented here with torch.distributed -- backend "nccl" is RCCL over xGMI on
ROCm -- directly on the library's device buffers.

Sharded-state mode (default, `make_torch_collective`): parameters, gradient and L-BFGS state
are split over the shards (a shard's fields, the block pairs inside its site blocks, half of the
rows of every rectangle of block pairs it shares with another shard); per evaluation two
all-to-alls (couplings, gradient fragments: every rank exchanges with every other) and one scalar
all-reduce, plus one scalar all-reduce per iteration for the L-BFGS Gram matrix (DESIGN.md
section 8); at the end of a fit ONE float32 all-reduce of the canonical parameter vector, in
which every entry has exactly one non-zero term (the all-gather of the J tensor).
Replicated mode (`make_torch_exchange`): one all-gather of the gradient slabs per evaluation,
every rank repeats the same L-BFGS step on the full vectors.
`ThreadedShards` / `LoopbackShards` run either mode with all shards on ONE GPU for tests.

The reference has nothing to compare with here: plmc parallelises with OpenMP threads
only (evcouplings/couplings/tools.py:257-259, the `cpu` option).
"""
import ctypes as C
import os

import numpy as np


def shard_blocks(n_sites, n_shards):
    """Column-block partition used by the library (plm_internal.h plm_shard_lo / plm_shard_cnt): list of (lo, hi)
    16-site block ranges, one per shard -- balanced: the LAST nb16 % n_shards shards own one block more
    (19 blocks on 8 GPUs: 2,2,2,2,2,3,3,3 -- the low shards own the long rows of the block-pair triangle, so the surplus
    column blocks go to the high ones); shards are empty only when there are more shards than blocks."""
    nb16 = (n_sites + 15) // 16
    base, rem = divmod(nb16, n_shards)
    out, lo = [], 0
    for r in range(n_shards):
        cnt = base + (1 if r >= n_shards - rem else 0)
        out.append((lo, lo + cnt))
        lo += cnt
    return out


def pair_owner(I, J, parts):
    """Which shard owns the block pair (I <= J) in sharded-state mode (plm_internal.h plm_pair_owner): the shard of both
    blocks if it is the same one; else, counting rows over the LOWER shard's blocks, even rows belong to the lower
    shard, odd rows to the higher one -- every shard owns half of each rectangle it shares with another."""
    s_i = next(r for r, (lo, hi) in enumerate(parts) if lo <= I < hi)
    s_j = next(r for r, (lo, hi) in enumerate(parts) if lo <= J < hi)
    if s_i == s_j:
        return s_i
    return s_j if (I - parts[s_i][0]) & 1 else s_i


def owned_block_pairs(n_sites, n_shards):
    """Number of block pairs (I <= J) every shard owns: its share of x, g and every L-BFGS vector."""
    parts = shard_blocks(n_sites, n_shards)
    nb16 = (n_sites + 15) // 16
    counts = [0] * n_shards
    for I in range(nb16):
        for J in range(I, nb16):
            counts[pair_owner(I, J, parts)] += 1
    return counts


def shard_sites(n_sites, n_shards):
    """Site ranges [lo, hi) owned by each shard."""
    return [(min(n_sites, 16 * lo), min(n_sites, 16 * hi)) for lo, hi in shard_blocks(n_sites, n_shards)]


def all_gather_inplace(buf, n_shards, shard, group=None):
    """
    buf: 1-D torch uint8 tensor of n_shards * bytes_per_shard (CPU for gloo, GPU for nccl),
    whose slice [shard] is filled in.  On return every slice is filled.
    """
    import torch.distributed as dist
    per = buf.numel() // n_shards
    mine = buf[shard * per:(shard + 1) * per]
    if buf.is_cuda:
        dist.all_gather_into_tensor(buf, mine.clone(), group=group)
    else:
        parts = [buf[r * per:(r + 1) * per] for r in range(n_shards)]
        tmp = [p.clone() for p in parts]
        dist.all_gather(tmp, mine.clone(), group=group)
        for p, t in zip(parts, tmp):
            p.copy_(t)
    return buf


class _DeviceBytes:
    """Zero-copy view of library-owned device memory for torch.as_tensor."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {
            "shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2,
            "strides": None,
        }


def make_torch_exchange(group=None):
    """Exchange callback for plm.fit(..., exchange=...) backed by RCCL."""
    import torch

    def exchange(dev_ptr, bytes_per_shard, n_shards, shard):
        try:
            buf = torch.as_tensor(_DeviceBytes(dev_ptr, bytes_per_shard * n_shards), device="cuda")
            all_gather_inplace(buf, n_shards, shard, group=group)
            torch.cuda.current_stream().synchronize()
            return 0
        except Exception as exc:  # never let an exception cross the C boundary
            import sys
            print("plm exchange failed: %r" % (exc,), file=sys.stderr)
            return 1

    return exchange


def collective_on_tensors(op, send, recv, send_counts, recv_counts, group=None):
    """
    The three collectives of the sharded-state mode on torch tensors (CPU/gloo or GPU/nccl):
    send/recv are 1-D uint8 tensors; counts are bytes per rank.
    """
    import torch
    import torch.distributed as dist
    from evcouplings_amd import _lib
    if op == _lib.COLL_ALLTOALL:
        dist.all_to_all_single(recv, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts),
                               group=group)
    elif op in (_lib.COLL_ALLREDUCE_F64, _lib.COLL_ALLREDUCE_F32):
        dtype = torch.float64 if op == _lib.COLL_ALLREDUCE_F64 else torch.float32
        dist.all_reduce(send.view(dtype), op=dist.ReduceOp.SUM, group=group)
    elif op == _lib.COLL_BROADCAST:
        # recv_counts[0] = root rank within the group
        root = int(recv_counts[0])
        dist.broadcast(send, src=dist.get_global_rank(group, root) if group is not None else root, group=group)
    else:
        raise ValueError("unknown collective op %r" % (op,))


def make_torch_collective(group=None):
    """Collective callback for plm.fit(..., collective=...): RCCL on the library's device buffers."""
    import torch
    from evcouplings_amd import _lib

    def collective(op, send_ptr, recv_ptr, send_counts, recv_counts, n_shards, shard):
        if op == _lib.COLL_ALLTOALL:
            send = torch.as_tensor(_DeviceBytes(send_ptr, max(1, sum(send_counts))), device="cuda")[:sum(send_counts)]
            recv = torch.as_tensor(_DeviceBytes(recv_ptr, max(1, sum(recv_counts))), device="cuda")[:sum(recv_counts)]
        else:     # all-reduce / broadcast: in place on send_counts[0] bytes
            send = torch.as_tensor(_DeviceBytes(send_ptr, send_counts[0]), device="cuda")
            recv = None
        collective_on_tensors(op, send, recv, send_counts, recv_counts, group=group)
        torch.cuda.current_stream().synchronize()
        return 0

    return collective


def make_host_staged_collective(group=None, device=0):
    """
    The same three collectives staged through host memory, for a process group whose backend cannot touch
    device buffers (gloo): ranks that share one GPU, or a host without RCCL peer access.  Slow path -- used to
    exercise the multi-process flow of bench.py / fit_distributed on a single-GPU box.
    """
    import torch
    from evcouplings_amd import _lib
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipSetDevice.argtypes = [C.c_int]

    def d2h(ptr, nbytes):
        host = np.empty(max(1, nbytes), np.uint8)
        if nbytes and hip.hipMemcpy(host.ctypes.data, C.c_void_p(ptr), nbytes, 2) != 0:
            raise RuntimeError("hipMemcpy device->host failed")
        return host[:nbytes]

    def h2d(ptr, host):
        if host.size and hip.hipMemcpy(C.c_void_p(ptr), host.ctypes.data, host.size, 1) != 0:
            raise RuntimeError("hipMemcpy host->device failed")

    def collective(op, send_ptr, recv_ptr, send_counts, recv_counts, n_shards, shard):
        hip.hipSetDevice(device)
        if op == _lib.COLL_ALLTOALL:
            send = torch.from_numpy(d2h(send_ptr, sum(send_counts)))
            recv = torch.empty(sum(recv_counts), dtype=torch.uint8)
            collective_on_tensors(op, send, recv, send_counts, recv_counts, group=group)
            h2d(recv_ptr, np.ascontiguousarray(recv.numpy()))
        else:     # all-reduce / broadcast: in place
            buf = torch.from_numpy(d2h(send_ptr, send_counts[0]).copy())
            collective_on_tensors(op, buf, None, send_counts, recv_counts, group=group)
            h2d(send_ptr, buf.numpy())
        return 0

    return collective


def share_rccl_id(group=None):
    """rank 0 creates the id of the library's own RCCL communicator, every rank of the group receives it"""
    import torch.distributed as dist
    from evcouplings_amd import plm
    box = [plm.rccl_unique_id() if dist.get_rank(group) == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return box[0]


def native_rccl_requested():
    """Collectives issued by the library itself (RCCL on its own stream: no host round trip around any of them) are
    OPT-IN (PLM_NATIVE_RCCL=1, or transport="native"): no multi-GPU node was available in any round, so the grouped
    send/recv all-to-all of plm_rccl.cpp has only ever run on a one-rank communicator.  The default
    transport of an nccl group is torch.distributed's own collectives on the library's device buffers.  Even when
    requested, the native transport is only taken after `negotiate_native_rccl` has seen it work on every rank."""
    return os.environ.get("PLM_NATIVE_RCCL", "0") not in ("", "0")


def negotiate_native_rccl(group=None, device=None):
    """Can this job run the library-issued RCCL transport?  Collective over the (nccl-backed) group: every rank checks
    that it can load RCCL, then all ranks form a probe communicator from an id rank 0 made, push an all-to-all and an
    all-reduce through it and destroy it (plm_rccl_probe); the verdicts are combined with a torch all-reduce, so every
    rank returns the same answer.  False = use the torch.distributed callbacks (and say why on rank 0)."""
    import sys
    import torch
    import torch.distributed as dist
    from evcouplings_amd import plm
    if not native_rccl_requested() or dist.get_backend(group) != "nccl":
        return False
    device = torch.cuda.current_device() if device is None else device

    def agree(ok):
        flag = torch.tensor([1 if ok else 0], device="cuda:%d" % device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return bool(int(flag.item()))

    why = ""
    try:
        # everything that can fail on one rank alone (device, librccl, buffer, upload) is settled BEFORE any rank
        # enters a communicator call: a rank that dropped out there would leave its peers blocked in ncclCommInitRank
        plm.rccl_probe_local(dist.get_world_size(group), device=device)
        ok = True
    except Exception as exc:       # noqa: BLE001 -- any failure means "not here"
        ok, why = False, repr(exc)
    if not agree(ok):
        if dist.get_rank(group) == 0:
            print("plm: library-issued RCCL transport unavailable (%s): torch.distributed callbacks" % (why or "another rank"),
                  file=sys.stderr)
        return False
    ident = share_rccl_id(group)
    try:
        plm.rccl_probe(ident, dist.get_world_size(group), dist.get_rank(group), device=device)
        ok = True
    except Exception as exc:       # noqa: BLE001
        ok, why = False, repr(exc)
    ok = agree(ok)
    if not ok and dist.get_rank(group) == 0:
        print("plm: RCCL probe communicator failed (%s): torch.distributed callbacks" % (why or "another rank"), file=sys.stderr)
    return ok


def fit_distributed(msa, q=21, group=None, sharded_state=True, transport=None, **kwargs):
    """
    plm.fit on every rank of an initialised process group, sites sharded across ranks.
    sharded_state=True (default): parameters, gradient and L-BFGS state are split by owning site block;
    per evaluation two all-to-alls of neighbour blocks + scalar all-reduces.  False: replicated state,
    one all-gather of the gradient slabs per evaluation.
    transport "rccl": torch.distributed collectives on the library's device buffers (backend nccl); "native": the
    library issues the RCCL calls itself on its stream (torch.distributed only carries the communicator id); "host":
    staged through host memory (backend gloo; sharded-state mode only).  Default:
    "rccl" on an nccl group ("native" is opt-in: PLM_NATIVE_RCCL=1, taken only when every rank's probe communicator
    worked, `negotiate_native_rccl`); "host" on gloo.
    """
    import torch
    import torch.distributed as dist
    from evcouplings_amd import plm
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    device = kwargs.pop("device", torch.cuda.current_device())
    if world == 1:
        return plm.fit(msa, q=q, device=device, **kwargs)
    if transport is None:
        if dist.get_backend(group) != "nccl":
            transport = "host"
        else:
            transport = "native" if (sharded_state and negotiate_native_rccl(group, device)) else "rccl"
    if sharded_state and transport == "native":
        return plm.fit(msa, q=q, n_shards=world, shard=rank, device=device, rccl_id=share_rccl_id(group), **kwargs)
    if sharded_state:
        coll = make_torch_collective(group) if transport == "rccl" else make_host_staged_collective(group, device)
        return plm.fit(msa, q=q, n_shards=world, shard=rank, device=device, collective=coll, **kwargs)
    if transport != "rccl":
        raise ValueError("replicated-state mode needs the rccl transport")
    return plm.fit(msa, q=q, n_shards=world, shard=rank, device=device, exchange=make_torch_exchange(group),
                   **kwargs)


def resolve_gpu_count(cpu=None, gpus=None):
    """How many GPUs a run_plmc-style call uses.  Multi-GPU is opt-in: `gpus` (explicit keyword), else the environment
    variable PLM_HIP_GPUS -- an integer, "max" (every visible GPU) or "cpu" (read plmc's `cpu` option, threads for
    plmc -n, tools.py:257-259, as the number of GPUs) -- else 1.  A pipeline-wide `cpu: N` therefore never starts GPU
    ranks by itself.  The count is capped by the visible devices."""
    from evcouplings_amd import plm
    env = os.environ.get("PLM_HIP_GPUS", "")
    want = gpus if gpus is not None else (env if env != "" else None)
    if isinstance(want, str) and want.lower() == "cpu":
        want = cpu
    if want is None:
        return 1
    if isinstance(want, str) and want.lower() == "max":
        return max(1, plm.device_count())
    n = max(1, int(want))
    # the gloo flow test folds ranks onto the visible GPUs: the count may exceed them there
    if os.environ.get("PLM_DIST_BACKEND", "nccl") != "nccl":
        return n
    return min(n, max(1, plm.device_count()))


class LaunchError(RuntimeError):
    """The multi-GPU child job ended without a result."""


def launch_fit(msa, n_gpus, q=21, timeout=None, **kwargs):
    """
    The multi-GPU fit as a child job: `n_gpus` ranks under torch.distributed.run on this node, one per GPU, sites and
    optimiser state sharded across them (evcouplings_amd/dist_worker.py).  Same keyword arguments and same result
    dictionary as plm.fit (no per-iteration callback: the iteration table comes back with the result).
    """
    import json
    import os
    import socket
    import subprocess
    import sys
    import tempfile
    kwargs = {k: v for k, v in kwargs.items() if k != "callback"}
    kwargs["q"] = int(q)
    with tempfile.TemporaryDirectory(prefix="plm_dist_") as tmp:
        src, dst = os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")
        np.savez(src, msa=np.ascontiguousarray(msa, dtype=np.int8), kwargs=json.dumps(kwargs))
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        run = None
        for attempt in range(3):
            # a free port is found by binding and releasing it; another process can take it in between, so a
            # rendezvous that dies with "address already in use" is retried on a fresh port
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_gpus)),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "evcouplings_amd.dist_worker", src,
                   dst]
            try:
                run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
            except (OSError, subprocess.TimeoutExpired) as exc:
                raise LaunchError("multi-GPU fit could not run: %r" % (exc,))
            if run.returncode == 0 or "ddress already in use" not in (run.stderr or ""):
                break
        if run.returncode != 0 or not os.path.exists(dst):
            raise LaunchError("multi-GPU fit failed (exit %d):\n%s\n%s" % (run.returncode, run.stdout[-2000:],
                                                                            run.stderr[-4000:]))
        z = np.load(dst)
        res = {k: z[k] for k in z.files if k not in ("table", "meta")}
        res.update(json.loads(str(z["meta"])))
        res["table"] = [tuple([int(r[0])] + [float(v) for v in r[1:]]) for r in z["table"]]
        if "fij" not in res:
            res["fij"] = None
        return res


# --------------------------------------------------------------------------------------------
# single-GPU stand-in for the multi-GPU path: all shards live on one device and run in turn
# --------------------------------------------------------------------------------------------
class LoopbackShards:
    """
    Runs the sharded evaluation with every shard on the same GPU (gpurun offers one GPU).
    Pass 1 records each shard's slab; pass 2 replays the evaluation on shard 0 with an
    exchange that fills in the recorded slabs -- the same code path a real all-gather feeds.
    """

    def __init__(self, msa, weights, q, lambda_h, lambda_j, n_shards, device=0):
        from evcouplings_amd import plm
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.n_shards = n_shards
        self.ctx = [plm.PlmContext(msa, q=q, lambda_h=lambda_h, lambda_j=lambda_j, device=device,
                                   n_shards=n_shards, shard=r) for r in range(n_shards)]
        for c in self.ctx:
            c.set_weights(weights)
        self.slabs = {}

    def _record(self, dev_ptr, nbytes, n_shards, shard):
        host = np.empty(nbytes, np.uint8)
        rc = self.hip.hipMemcpy(host.ctypes.data, C.c_void_p(dev_ptr + shard * nbytes), nbytes, 2)
        self.slabs[shard] = host
        return rc

    def _replay(self, dev_ptr, nbytes, n_shards, shard):
        for r, host in self.slabs.items():
            if r != shard:
                rc = self.hip.hipMemcpy(C.c_void_p(dev_ptr + r * nbytes), host.ctypes.data, nbytes, 1)
                if rc:
                    return rc
        return 0

    def evaluate(self, x):
        self.slabs = {}
        for c in self.ctx:
            c.set_exchange(self._record)
            c.set_x(x)
            c.eval()
        out = None
        for c in self.ctx:            # every shard must arrive at the same answer
            c.set_exchange(self._replay)
            fx, nll = c.eval()
            g = c.get_g()
            if out is None:
                out = (fx, nll, g)
            else:
                assert fx == out[0] and np.array_equal(g, out[2]), "shards disagree"
        return out


class ThreadedShards:
    """
    Sharded-state mode with every shard on the SAME GPU, one Python thread per shard, collectives
    emulated through host memory and barriers (gpurun offers one GPU).  The C library code that runs is
    exactly what runs under torch.distributed; only the transport differs.
    """

    def __init__(self, n_shards, device=0):
        import threading
        from evcouplings_amd import _lib
        self._lib = _lib
        self.n = n_shards
        self.device = device
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipSetDevice.argtypes = [C.c_int]
        self.barrier = threading.Barrier(n_shards)
        self.slots = [None] * n_shards
        self.n_calls = {"alltoall": 0, "allreduce": 0}

    def _d2h(self, ptr, nbytes):
        host = np.empty(max(1, nbytes), np.uint8)
        if nbytes:
            assert self.hip.hipMemcpy(host.ctypes.data, C.c_void_p(ptr), nbytes, 2) == 0
        return host[:nbytes]

    def _h2d(self, ptr, host):
        if host.size:
            host = np.ascontiguousarray(host)
            assert self.hip.hipMemcpy(C.c_void_p(ptr), host.ctypes.data, host.size, 1) == 0

    def collective_for(self, shard):
        lib = self._lib

        def collective(op, send_ptr, recv_ptr, send_counts, recv_counts, n_shards, me):
            assert me == shard and n_shards == self.n
            self.hip.hipSetDevice(self.device)
            if op == lib.COLL_ALLTOALL:
                self.slots[me] = (self._d2h(send_ptr, sum(send_counts)), list(send_counts))
                self.barrier.wait()
                parts = []
                for r in range(self.n):
                    buf, cnt = self.slots[r]
                    off = sum(cnt[:me])
                    assert cnt[me] == recv_counts[r], "all-to-all split mismatch %d -> %d" % (r, me)
                    parts.append(buf[off:off + cnt[me]])
                self._h2d(recv_ptr, np.concatenate(parts) if parts else np.empty(0, np.uint8))
                if me == 0:
                    self.n_calls["alltoall"] += 1
            elif op == lib.COLL_BROADCAST:
                root = int(recv_counts[0])
                if me == root:
                    self.slots[root] = self._d2h(send_ptr, send_counts[0])
                self.barrier.wait()
                if me != root:
                    self._h2d(send_ptr, self.slots[root])
                if me == 0:
                    self.n_calls["broadcast"] = self.n_calls.get("broadcast", 0) + 1
            else:
                dtype = np.float64 if op == lib.COLL_ALLREDUCE_F64 else np.float32
                self.slots[me] = self._d2h(send_ptr, send_counts[0]).view(dtype)
                self.barrier.wait()
                total = self.slots[0].copy()
                for r in range(1, self.n):
                    total += self.slots[r]
                self._h2d(send_ptr, total.view(np.uint8))
                if me == 0:
                    self.n_calls["allreduce"] += 1
            self.barrier.wait()   # everyone has read the slots before the next collective overwrites them
            return 0

        return collective

    def run(self, fn):
        """fn(shard, collective) is called in one thread per shard; returns the list of results."""
        import threading
        out, err = [None] * self.n, [None] * self.n

        def work(r):
            try:
                out[r] = fn(r, self.collective_for(r))
            except BaseException as exc:   # noqa: BLE001 - reported below, and the barrier is broken for the others
                err[r] = exc
                self.barrier.abort()

        threads = [threading.Thread(target=work, args=(r,)) for r in range(self.n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        real = [e for e in err if e is not None]
        if real:
            # prefer the first non-barrier error
            import threading as _t
            for e in real:
                if not isinstance(e, _t.BrokenBarrierError):
                    raise e
            raise real[0]
        return out

    def fit(self, msa, q=21, **kwargs):
        from evcouplings_amd import plm
        return self.run(lambda r, coll: plm.fit(msa, q=q, n_shards=self.n, shard=r, device=self.device,
                                                collective=coll, **kwargs))
