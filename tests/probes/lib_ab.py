"""A/B of library builds on one box: the isolated launches of plm_ctx_time_kernels (forward / backward GEMM), many
repetitions, the builds interleaved.  usage: lib_ab.py REPS LIB [LIB ...]   (evcouplings_amd/libplm_LIB.so)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r"""
import sys, json
sys.path.insert(0, %r)
from evcouplings_amd import plm
from evcouplings_amd.synthetic import synthetic_msa, BASE_SEED
msa, _ = synthetic_msa(50000, 300, seed=BASE_SEED + 1)
with plm.PlmContext(msa, q=21, max_iter=3, epsilon=1e-3) as ctx:
    ctx.reweight(); ctx.marginals(pairs=False); ctx.set_x(None); ctx.optimize()
    out = []
    for r in range(int(sys.argv[1])):
        k = ctx.time_kernels(reps=20)
        out.append((k["forward"], k["backward"]))
    print(json.dumps(out))
""" % ROOT
reps, libs = int(sys.argv[1]), sys.argv[2:]
res = {l: [] for l in libs}
for rnd in range(2):
    for l in libs:
        env = dict(os.environ, PLM_HIP_LIB=os.path.join(ROOT, "evcouplings_amd", "libplm_%s.so" % l))
        o = subprocess.run([sys.executable, "-c", WORKER, str(reps)], env=env, capture_output=True, text=True)
        try:
            res[l] += json.loads(o.stdout.strip().splitlines()[-1])
        except Exception:
            print(l, "FAILED", o.stderr[-500:])
for l in libs:
    f = sorted(v[0] for v in res[l]); b = sorted(v[1] for v in res[l])
    if f:
        print("%-6s forward min %.4f median %.4f max %.4f | backward min %.4f median %.4f  (n=%d)" % (
            l, f[0], f[len(f) // 2], f[-1], b[0], b[len(b) // 2], len(f)))
