"""Per-stage summary of the ranges a `HP_ROCTX=1 rocprofv3 --kernel-trace --marker-trace` run recorded (hiddenpose_amd/ranges.py).
    python tools/marker_summary.py <rocprofv3 results.db> [skip_first_n_steps] > profiles/roundN_t512_marker_ranges.txt
The ranges bracket the host-side ENQUEUE of each stage (the device runs behind the host; the kernel trace of the same run has the
device times), so what the table shows is where the host thread and the autograd thread spend a step."""
import json
import sqlite3
import sys
from collections import OrderedDict

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = db.execute("select tid, start, end, extdata, name from regions order by start").fetchall()
steps = sum(1 for r in rows if json.loads(r[3]).get("message") == "optimizer")
stat = OrderedDict()
seen = {}
for tid, s, e, ext, kind in rows:
    name = json.loads(ext).get("message", "?")
    seen[name] = seen.get(name, 0) + 1
    if seen[name] <= skip:
        continue
    n, tot, mx, k, t = stat.get(name, (0, 0.0, 0.0, kind, tid))
    stat[name] = (n + 1, tot + (e - s) / 1e6, max(mx, (e - s) / 1e6), kind, tid)
print(f"# {steps} steps recorded, the first {skip} of every range left out (plan creation, allocator growth)")
print(f"# {'range':<26} {'api':<20} {'thread':>8} {'count':>6} {'mean ms':>10} {'max ms':>10}")
for name, (n, tot, mx, kind, tid) in stat.items():
    print(f"  {name:<26} {kind:<20} {tid:>8} {n:>6} {tot / n:>10.3f} {mx:>10.3f}")
kern = db.execute("select count(*), sum(end - start) / 1e6 from kernels").fetchone()
print(f"# kernel trace of the same run: {kern[0]} dispatches, {kern[1]:.1f} ms of kernel time")
