#!/bin/bash
# Per-kernel VGPR / SGPR / spill / LDS / occupancy table of fx_kernels.hip as hipcc compiles it for gfx950.
cd "$(dirname "$0")/../../gr-liquiddsp_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math $EXTRA \
  -x hip --cuda-device-only -c fx_kernels.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import sys, re
cur = None; rows = {}
for l in sys.stdin:
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", l) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", l)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = t.split(":",1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":",1); rows[cur][k.strip()] = v.strip()
import subprocess
for k, r in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-52s VGPR %-4s AGPR %-3s SGPR %-4s spillV %-4s spillS %-4s scratch %-5s LDS %-7s occ %s" % (name[:52], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"),
          r.get("VGPR Spill"), r.get("SGPR Spill"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
'
