#!/usr/bin/env python3
"""Print a table of per-kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "rdis_amd/csrc/rdis_hip.hip"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_kres.so", src] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>6s} {'vspill':>6s} {'sspill':>6s}")
for r in rows:
    print(f"{r['name'][:70]:70s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('TotalSGPRs', 0):5d} "
          f"{r.get('ScratchSize', 0):8d} {r.get('Occupancy', 0):4d} {r.get('LDS Size', 0):6d} "
          f"{r.get('VGPRs Spill', 0):6d} {r.get('SGPRs Spill', 0):6d}")
