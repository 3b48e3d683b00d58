#!/usr/bin/env python3
"""Print a per-kernel register/LDS/occupancy table for a .hip file (gfx950)."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*) \[-Rpass", line) or re.search(r"remark: (.*) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
import shutil
for k, r in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() if shutil.which("c++filt") else k
    print("%-58s vgpr=%-4s agpr=%-4s sgpr=%-4s scratch=%-5s vspill=%-4s sspill=%-4s occ=%s lds=%s" % (
        name[:58], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize [bytes/lane]"),
        r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
