#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage
(run from the repo root: python microbench/kres.py [file.hip ...])."""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "implementing-jpeg-compression_amd", "csrc")
files = sys.argv[1:] or [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
                          "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", f], cwd=CSRC, capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+(Function Name|Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k in ("Function Name", "Name"):
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        else:
            cur[k.split(" ")[0]] = v
        if k.startswith("LDS"):
            n = re.sub(r"\(anonymous namespace\)::|\(.*$", "", cur["name"])
            print("%-58s vgpr %-4s sgpr %-4s scratch %-4s occ %-2s lds %s" % (n[:58], cur.get("VGPRs"), cur.get("TotalSGPRs"), cur.get("ScratchSize"), cur.get("Occupancy"), cur.get("LDS")))
