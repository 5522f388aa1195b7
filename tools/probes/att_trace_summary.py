"""Per-queue kernel time per train step from the kernel trace tools/probes/att_trace.sh leaves (gpurun_out/att_trace):
python tools/probes/att_trace_summary.py [csv]"""
import collections
import csv
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/att_trace/t_kernel_trace.csv"
rows = list(csv.DictReader(open(path)))
idx = [i for i, r in enumerate(rows) if "clamp_adam" in r["Kernel_Name"]]
per_step = 2 if len(idx) > 20 else 1           # (two parameter groups: two launches per step)
a, b = idx[-1 - 6 * per_step], idx[-1]
sel, n = rows[a + 1:b + 1], 6
byq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in sel:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("capnet::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    e = byq[r["Queue_Id"]][nm or r["Kernel_Name"][:60]]
    e[0] += 1
    e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for q, m in byq.items():
    tot, cnt = sum(v[1] for v in m.values()) / n, sum(v[0] for v in m.values()) / n
    if cnt < 100:
        continue
    print("queue %s: %.0f launches, %.0f us of kernels per step" % (q, cnt, tot))
    for nm, v in sorted(m.items(), key=lambda kv: -kv[1][1])[:24]:
        print("   %-72s %6.1f calls %8.1f us  (%.1f us each)" % (nm[:72], v[0] / n, v[1] / n, v[1] / v[0]))
