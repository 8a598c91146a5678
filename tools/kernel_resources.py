#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py lidar_odometry_demo_amd/csrc/match.hip [name-filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-fno-gpu-rdc",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(f"{k:70s} VGPR {v.get('VGPRs', -1):4d} AGPR {v.get('AGPRs', -1):3d} SGPR {v.get('TotalSGPRs', -1):4d} "
              f"scratch {v.get('ScratchSize', -1):4d} vspill {v.get('VGPRs Spill', -1):3d} sspill {v.get('SGPRs Spill', -1):3d} "
              f"occ {v.get('Occupancy', -1)} lds {v.get('LDS Size', -1)}")
