"""Idle time of the GPU between consecutive kernels of one training step, from a rocprofv3 --kernel-trace CSV (dev tool).
usage: gap_analysis.py <kernel_trace.csv> <steps_in_trace> [step index, default last] [name of the step's first kernel]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
steps = int(sys.argv[2])
# last step = last 1/steps of the launches
n = len(rows) // steps
idx = int(sys.argv[3]) if len(sys.argv) > 3 else steps - 1
# a step starts at the first launch of the forward (FeatureExtraction's replicate-padded convolution)
marker = sys.argv[4] if len(sys.argv) > 4 else next((m for m in ("k_stencil_c1<1", "k_dconv3_mfma<1") if any(m in r[2] for r in rows)), "")
starts = [i for i, r in enumerate(rows) if marker in r[2] and (i == 0 or marker not in rows[i - 1][2])]
first = [i for j, i in enumerate(starts) if j == 0 or i - starts[j - 1] > n // 2]
last = rows[first[idx]:first[idx + 1]] if idx + 1 < len(first) else rows[first[idx]:]
n = len(last)
busy = sum(e - s for s, e, _ in last)
span = last[-1][1] - last[0][0]
gaps = []
end = last[0][1]
for s, e, name in last[1:]:
    if s > end:
        gaps.append((s - end, name))
    end = max(end, e)
idle = sum(g for g, _ in gaps)
print(f"kernels {n}  span {span/1e6:.2f} ms  busy(sum) {busy/1e6:.2f} ms  idle {idle/1e6:.2f} ms in {len(gaps)} gaps")
hist = {}
for g, _ in gaps:
    k = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"
    hist[k] = hist.get(k, [0, 0])
    hist[k][0] += 1
    hist[k][1] += g
for k, (c, t) in hist.items():
    print(f"  gaps {k:7s}: {c:5d}  total {t/1e6:.3f} ms")
print("largest gaps (us, before kernel):")
for g, name in sorted(gaps, reverse=True)[:15]:
    print(f"  {g/1e3:9.1f}  {name[:90]}")
