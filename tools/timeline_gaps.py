"""Idle time between kernels from a rocprofv3 --kernel-trace CSV (GPU box or afterwards): for the last N steps of the run, the
sequence of kernels of one step with mean duration and mean gap to the previous kernel, and the totals.
usage: python tools/timeline_gaps.py <kernel_trace.csv> [steps=10]"""
import csv, sys
from collections import defaultdict
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from summarize_profile import short

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [short(r["Kernel_Name"]) for r in rows]
# a step ends with the fused force launch
ends = [i for i, n in enumerate(names) if n == "k_forces_lists"]
ends = ends[-(steps + 1):]
seqs = defaultdict(lambda: [0.0, 0.0, 0])
tot_busy = tot_gap = 0.0
for a, b in zip(ends[:-1], ends[1:]):
    for pos, i in enumerate(range(a + 1, b + 1)):
        dur = int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
        gap = int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])
        k = (pos, names[i])
        seqs[k][0] += dur; seqs[k][1] += gap; seqs[k][2] += 1
        tot_busy += dur; tot_gap += gap
n = len(ends) - 1
print("steps analysed %d: busy %.1f us/step, gaps %.1f us/step, span %.1f us/step" % (n, tot_busy / n / 1e3, tot_gap / n / 1e3, (tot_busy + tot_gap) / n / 1e3))
for (pos, name), (d, g, c) in sorted(seqs.items()):
    print("%3d %-34s x%-3d dur %8.1f us   gap before %7.1f us" % (pos, name, c, d / c / 1e3, g / c / 1e3))
