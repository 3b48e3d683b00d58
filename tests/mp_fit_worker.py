"""Worker of test_gpu_parity.py::test_two_process_fit_over_gloo: one rank of a torch.distributed.run launch.
Fits a small synthetic alignment with the sites sharded over the ranks (all ranks on GPU 0, collectives staged
through host memory over gloo) and lets rank 0 save the result."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    from evcouplings_amd import plm
    from evcouplings_amd.dist import fit_distributed
    from evcouplings_amd.synthetic import synthetic_msa
    out = sys.argv[1]
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    msa, _ = synthetic_msa(600, 70, seed=11)
    res = fit_distributed(msa, q=21, transport="host", device=0, lambda_h=0.01, lambda_j=plm.default_lambda_j(70, 21),
                          max_iter=25, epsilon=1e-9, want_fij=False)
    if dist.get_rank() == 0:
        np.savez(out, jij=res["jij"], hi=res["hi"], cn=res["cn"], fx=res["fx"], iters=res["iters"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
