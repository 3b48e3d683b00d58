import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa
msa, _ = synthetic_msa(20000, 1000, seed=5)
plm.reweight(msa[:2000], 0.8)
t = time.time(); c = plm.reweight(msa, 0.8); print("N=20000 L=1000 reweight (chunked kernel, incl. upload) %.1f ms" % (1e3 * (time.time() - t)), c.sum())
