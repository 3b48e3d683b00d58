#!/usr/bin/env python3
"""GPU-side: what ONE rank of a G-GPU run computes per iteration, measured on one MI355X (gpurun offers one GPU), and the
iteration time that follows for G = 2, 4, 8.  PROJECTED, not a SCALE number: no multi-GPU node was reachable in any round.

1. The 1-GPU row is MEASURED the way bench.py measures `value`: W = 5 warm-up iterations, then 30 timed iterations of the
   default fit (same window, same synchronisation), with the in-fit statistics of the library (plm_ctx_solver_stats:
   field-chain ms and passes per evaluation, the two GEMMs inside the fit).
2. A MODEL of an iteration is built from isolated launches (plm_ctx_time_kernels, plm_ctx_time_field_positions):
       evaluation = expand + forward + (p - 1) * hessian_position + closing_position + backward + assemble
       iteration  = evals_per_iteration * evaluation + lbfgs_vector
   with p = passes per evaluation and evals_per_iteration taken from the measured window.  The model of the 1-GPU context is
   printed beside the measurement ("model_over_measured"): the projection is only as good as that ratio is close to 1.
3. For G in 2, 4, 8 every shard r of the site-sharded (sharded-state) layout gets its own context on this GPU; the same
   model is evaluated on ITS isolated launches (its column blocks, its share of the state), plus the collectives: two
   all-to-alls (largest per-peer message over one xGMI link + latency) and one scalar all-reduce per evaluation.  The
   iteration time of a G-GPU job is that of its slowest rank; speedup = model(1 GPU) / that.

usage: shard_compute.py [N L seed_offset] > profiles/r06_shard_compute.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.dist import shard_blocks, owned_block_pairs
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED

N = int(sys.argv[1]) if len(sys.argv) > 2 else 50000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 1
Q = 21
XGMI_LINK_GBS = 153.0          # per direction and link, MI355X_MICROARCH.md (7 links per GPU, fully connected node)
COLL_LATENCY_US = 20.0         # launch + completion latency of one RCCL collective on a stream (assumption, stated)
WARMUP, STEPS = 5, 30
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + SEED)
out = {"workload": "synthetic MSA N=%d L=%d q=%d" % (N, L, Q), "note": __doc__.split("usage")[0].strip(),
       "assumptions": {"xgmi_link_GBps": XGMI_LINK_GBS, "collective_latency_us": COLL_LATENCY_US}, "gpus": {}}


def timed(ctx):
    km = ctx.time_kernels(reps=3)
    km.update(ctx.time_field_positions(reps=3))
    return km


with plm.PlmContext(msa, q=Q, max_iter=WARMUP, epsilon=1e-3) as ctx:
    w, _, n_eff = ctx.reweight()
    ctx.marginals(pairs=False)
    ctx.set_x(None)
    ctx.optimize()
    ctx._set_max_iter(STEPS)
    t0 = time.perf_counter()
    res = ctx.optimize()                 # bench.py's timed window: iterations W + 1 .. W + K
    dt = time.perf_counter() - t0
    st = ctx.solver_stats()
    x0 = ctx.get_x()                     # a point with couplings of realistic size
    base = timed(ctx)
measured = {"ms_per_iteration": 1e3 * dt / res["iters"], "iterations": res["iters"], "evaluations": res["n_evals"],
            "evals_per_iteration": res["n_evals"] / res["iters"], "field_ms_per_evaluation": st["field_ms_per_evaluation"],
            "passes_per_evaluation": st["passes_per_evaluation"], "forward_ms_in_fit": st["forward_ms_per_evaluation"],
            "backward_ms_in_fit": st["backward_ms_per_evaluation"]}
P, EPI = measured["passes_per_evaluation"], measured["evals_per_iteration"]
nb16 = (L + 15) // 16
out["gpus"]["1"] = {"shards": [dict(base, blocks=nb16, block_pairs=nb16 * (nb16 + 1) // 2)]}
blk_bytes = Q * Q * 256 * 4
for G in (2, 4, 8):
    shards = []
    parts = shard_blocks(L, G)
    for r in range(G):
        with plm.PlmContext(msa, q=Q, n_shards=G, shard=r, sharded_state=True, max_iter=20, epsilon=1e-3) as c:
            c.set_weights(w)
            c.set_x(x0)
            km = timed(c)
        lo, hi = parts[r]
        own = hi - lo
        # block pairs of this rank: the triangle over its own blocks + its half of the rectangle shared with every other
        # rank (even rows -- blocks of the lower rank -- belong to the lower rank, odd rows to the higher: plm_pair_owner);
        # per evaluation it sends the couplings of its halves and receives the partners' gradient fragments for them, and
        # the other way round for the partners' halves
        cnt = [b - a for a, b in parts]
        mine = {p: ((own + 1) // 2) * cnt[p] if r < p else (cnt[p] // 2) * own for p in range(G) if p != r}
        theirs = {p: own * cnt[p] - mine[p] for p in mine}
        assert own * (own + 1) // 2 + sum(mine.values()) == owned_block_pairs(L, G)[r]
        km.update(blocks=own, block_pairs=own * (own + 1) // 2 + sum(mine.values()),
                  x_halo_recv_MB=sum(theirs.values()) * blk_bytes / 1e6, g_halo_recv_MB=sum(mine.values()) * blk_bytes / 1e6,
                  x_halo_send_MB=sum(mine.values()) * blk_bytes / 1e6, g_halo_send_MB=sum(theirs.values()) * blk_bytes / 1e6,
                  largest_peer_message_MB=max(max(mine.values()), max(theirs.values())) * blk_bytes / 1e6)
        shards.append(km)
    out["gpus"][str(G)] = {"shards": shards}


def eval_ms(km):
    return (km["expand"] + km["forward"] + (P - 1.0) * km["hessian_position"] + km["closing_position"] + km["backward"] +
            km["assemble"])


model1 = EPI * eval_ms(base) + base["lbfgs_vector"]
proj = {"1": {"ms_per_iteration_model": model1, "ms_per_iteration_measured": measured["ms_per_iteration"],
              "model_over_measured": model1 / measured["ms_per_iteration"], "speedup": 1.0}}
for G in (2, 4, 8):
    worst, who = 0.0, None
    for r, km in enumerate(out["gpus"][str(G)]["shards"]):
        # every peer is reached over its own xGMI link (full duplex): an all-to-all lasts as long as its largest
        # per-peer message (MB / (GB/s) = ms); + one scalar all-reduce per evaluation
        coll = 2 * (COLL_LATENCY_US * 1e-3 + km["largest_peer_message_MB"] / XGMI_LINK_GBS) + COLL_LATENCY_US * 1e-3
        t = EPI * (eval_ms(km) + coll) + km["lbfgs_vector"]
        km["model_ms_per_iteration"] = t
        km["collectives_ms_per_evaluation"] = coll
        if t > worst:
            worst, who = t, r
    proj[str(G)] = {"ms_per_iteration_model": worst, "slowest_rank": who, "speedup": model1 / worst}
out["measured_1gpu_window"] = measured
out["projection"] = {"per_gpus": proj,
                     "label": "PROJECTED from single-GPU measurements of every shard's isolated launches + message sizes / "
                              "link rate; not a SCALE number (no multi-GPU node is reachable)"}
print(json.dumps(out, indent=1))
