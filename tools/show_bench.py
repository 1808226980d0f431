"""python tools/show_bench.py <bench json line file> [n]: step time and the n largest kernel families."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
print(d["ms_per_step"], "ms/step", d["value"], d["unit"], "| roofline:", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac")} if d.get("roofline") else None)
for k, v in sorted(d.get("hip_kernel_ms_per_step", {}).items(), key=lambda t: -t[1])[:n]:
    print("  %-26s %8.2f %s" % (k, v, d.get("mfma_tflops_by_kernel", {}).get(k, "")))
