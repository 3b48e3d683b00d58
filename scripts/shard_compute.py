#!/usr/bin/env python3
"""GPU-side: what ONE rank of a G-GPU run computes per evaluation, measured on one MI355X (gpurun offers one GPU).

For G in 1, 2, 4, 8 every shard r of the site-sharded (sharded-state) layout gets its own context on this GPU and
plm_ctx_time_kernels times its kernels over ITS site blocks alone: expand, forward GEMM, one field-solver step +
residual pass, backward GEMM, assemble (+ pack of the gradient halo), and the L-BFGS vector kernels of one iteration on
its share of the state.  The halo buffers are read as they are -- the two all-to-alls that fill them and the scalar
all-reduce are NOT measured here; DESIGN.md section 8 adds them from message sizes and the xGMI link rate.  Output: one
JSON document (profiles/r05_shard_compute.json) with per-(G, shard) kernel times, the exchange volumes, and the
resulting projected evaluation time per G.  Projected, not a SCALE number.

usage: shard_compute.py [N L] > profiles/r05_shard_compute.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.dist import shard_blocks
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED

N = int(sys.argv[1]) if len(sys.argv) > 2 else 50000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
Q = 21
XGMI_LINK_GBS = 153.0          # per direction and link, MI355X_MICROARCH.md (7 links per GPU, fully connected node)
COLL_LATENCY_US = 20.0         # launch + completion latency of one RCCL collective on a stream (assumption, stated)
msa, _ = synthetic_msa(N, L, seed=BASE_SEED + 1)
out = {"workload": "synthetic MSA N=%d L=%d q=%d" % (N, L, Q), "note": __doc__.split("usage")[0].strip(),
       "assumptions": {"xgmi_link_GBps": XGMI_LINK_GBS, "collective_latency_us": COLL_LATENCY_US}, "gpus": {}}
with plm.PlmContext(msa, q=Q, max_iter=20, epsilon=1e-3) as ctx:
    w, _, n_eff = ctx.reweight()
    ctx.marginals(pairs=False)
    ctx.set_x(None)
    ctx.optimize()                       # a point with couplings of realistic size
    x0 = ctx.get_x()
    base = ctx.time_kernels(reps=3)
out["gpus"]["1"] = {"shards": [dict(base, blocks=(L + 15) // 16)]}
nb16 = (L + 15) // 16
blk_bytes = Q * Q * 256 * 4
for G in (2, 4, 8):
    shards = []
    parts = shard_blocks(L, G)
    for r in range(G):
        with plm.PlmContext(msa, q=Q, n_shards=G, shard=r, sharded_state=True, max_iter=20, epsilon=1e-3) as c:
            c.set_weights(w)
            c.set_x(x0)
            km = c.time_kernels(reps=3)
        lo, hi = parts[r]
        own = hi - lo
        # all-to-all volumes of this rank per evaluation (blocks of Q*Q*256 floats): couplings of lower shards' pairs
        # with own column blocks come in, gradient fragments of own pairs with higher column blocks come in
        peer_max = max((b - a) for k, (a, b) in enumerate(parts) if k != r)
        km.update(blocks=own, x_halo_recv_MB=lo * own * blk_bytes / 1e6, g_halo_recv_MB=own * (nb16 - hi) * blk_bytes / 1e6,
                  x_halo_send_MB=own * (nb16 - hi) * blk_bytes / 1e6, g_halo_send_MB=lo * own * blk_bytes / 1e6,
                  largest_peer_message_MB=peer_max * own * blk_bytes / 1e6)
        shards.append(km)
    out["gpus"][str(G)] = {"shards": shards}


def eval_ms(km, fields):
    return km["expand"] + km["forward"] + fields + km["backward"] + km["assemble"]


# projection: an evaluation on G GPUs = the slowest rank's kernels (with the field solver's measured share scaled by the
# rank's share of the site blocks) + two all-to-alls (largest per-peer message over one xGMI link + latency) + one scalar
# all-reduce (latency); an iteration = 1.05 evaluations + the vector kernels of the busiest rank
fields_1gpu = float(os.environ.get("PLM_FIELDS_MS", base["fields"]))
t1 = eval_ms(base, fields_1gpu) * 1.05 + base["lbfgs_vector"]
proj = {"1": {"ms_per_iteration": t1, "speedup": 1.0}}
for G in (2, 4, 8):
    sh = out["gpus"][str(G)]["shards"]
    worst = 0.0
    for km in sh:
        comp = eval_ms(km, fields_1gpu * km["blocks"] / nb16)
        # every peer is reached over its own xGMI link (full duplex): an all-to-all lasts as long as its largest
        # per-peer message (MB / (GB/s) = ms)
        a2a = 2 * (COLL_LATENCY_US * 1e-3 + km["largest_peer_message_MB"] / XGMI_LINK_GBS)
        worst = max(worst, (comp + a2a + COLL_LATENCY_US * 1e-3) * 1.05 + km["lbfgs_vector"])
    proj[str(G)] = {"ms_per_iteration": worst, "speedup": t1 / worst}
out["projection"] = {"per_gpus": proj, "fields_ms_per_evaluation_1gpu": fields_1gpu,
                     "label": "PROJECTED from single-GPU measurements of every shard's kernels + message sizes / link "
                              "rate; not a SCALE number (no multi-GPU node is reachable)"}
print(json.dumps(out, indent=1))
