#!/bin/bash
# Register / LDS / occupancy report of one kernel source (dev tool): tools/kernel_resources.sh conv_kernels.hip [regex]
src=hiddenpose_amd/csrc/$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I include -I hiddenpose_amd/csrc -munsafe-fp-atomics \
  -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 | python3 -c '
import re, sys
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark: \s*([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if pat.search(k):
        print(k[:110], {a: v.get(a) for a in ("VGPRs", "AGPRs", "Occupancy", "LDS Size", "VGPRs Spill", "SGPRs Spill")})
' "${2:-.}"
