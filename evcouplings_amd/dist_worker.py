"""
One rank of a multi-GPU fit launched by `evcouplings_amd.dist.launch_fit` (and through it by
`run_plmc_hip(..., cpu=N)` / `bin/plmc_hip -n N`):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... -m evcouplings_amd.dist_worker IN.npz OUT.npz

Every rank loads the encoded alignment, joins the process group (backend "nccl" = RCCL over xGMI; PLM_DIST_BACKEND=gloo
stages the collectives through host memory and folds the ranks onto the visible GPUs -- a flow test for single-GPU
boxes, never a benchmark) and runs the site- and state-sharded fit; rank 0 writes the result.
"""
import json
import os
import sys

import numpy as np


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    src, dst = argv[0], argv[1]
    import torch
    import torch.distributed as dist
    from evcouplings_amd.dist import fit_distributed
    backend = os.environ.get("PLM_DIST_BACKEND", "nccl")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend)
    try:
        z = np.load(src)
        kwargs = json.loads(str(z["kwargs"]))
        res = fit_distributed(z["msa"], transport=None if backend == "nccl" else "host", device=local_rank, **kwargs)
        if dist.get_rank() == 0:
            arrays = {k: v for k, v in res.items() if isinstance(v, np.ndarray)}
            meta = {k: v for k, v in res.items() if not isinstance(v, np.ndarray) and k != "table"}
            np.savez(dst, table=np.asarray(res["table"], dtype=np.float64).reshape(-1, 7), meta=json.dumps(meta), **arrays)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
