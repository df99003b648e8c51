#!/usr/bin/env python3
"""Per-sweep summary of a rocprofv3 --kernel-trace of one Mode B run (tools/modeb_probe.py): merged / open-tile arms / classify
durations in blocks of 50 sweeps.  usage: modeb_trace_summary.py <kernel_trace.csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
big = max((r["Kernel_Name"] for r in rows if "pv_step_merged_kernel" in r["Kernel_Name"]), key=lambda n: sum(1 for r in rows if r["Kernel_Name"] == n))
ms = [r for r in rows if r["Kernel_Name"] == big or "pv_step_fused_kernel" in r["Kernel_Name"]]
fu = [r for r in ms if "pv_step_fused_kernel" in r["Kernel_Name"]]
if fu:
    print("of which one-launch sweeps (pv_step_fused_kernel): %d, %.1f ms" % (len(fu), sum(map(d, fu)) / 1e3))
def arm(r):
    m = re.search(r"pv_step_open_kernel<\d+, \d+(?:, (\d+))?>", r["Kernel_Name"])
    return None if not m else int(m.group(1) or 0)
ops = [(arm(r), r) for r in rows if arm(r) is not None]
cl = [r for r in rows if "classify" in r["Kernel_Name"]]
print("merged %d launches %.1f ms; classify %d, %.1f ms" % (len(ms), sum(map(d, ms)) / 1e3, len(cl), sum(map(d, cl)) / 1e3))
for a in sorted(set(a for a, _ in ops)):
    rs = [r for b, r in ops if b == a]
    print("open arm %d: %d launches, %.1f ms" % (a, len(rs), sum(map(d, rs)) / 1e3))
for i in range(0, len(ms), 50):
    seg = ms[i:i + 50]
    a, b = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    line = "sweeps %4d..: span/sweep %6.1f us, merged %6.1f" % (i, (b - a) / 1e3 / len(seg), sum(map(d, seg)) / len(seg))
    for k in sorted(set(x for x, _ in ops)):
        o = [r for x, r in ops if x == k and a <= int(r["Start_Timestamp"]) <= b]
        line += ", arm%d %6.1f" % (k, sum(map(d, o)) / len(seg))
    print(line)
    if not any(a <= int(r["Start_Timestamp"]) <= b for _, r in ops) and i > 100:
        print("(no open-tile launches from here on)")
        break
acc = [r for r in rows if "pv_stream_accum_kernel" in r["Kernel_Name"]]
if acc:
    print("accumulate passes: %d, %.1f ms in total; mean us per block of 20 passes: %s" % (
        len(acc), sum(map(d, acc)) / 1e3, " ".join("%.0f" % (sum(map(d, acc[i:i + 20])) / len(acc[i:i + 20])) for i in range(0, len(acc), 20))))
for nm in ("pv_stream_trace_kernel", "pv_stream_tilegate_kernel", "pv_stream_idle_kernel", "fillBuffer"):
    rs = [r for r in rows if nm in r["Kernel_Name"]]
    if rs:
        print("%s: %d launches, %.1f ms" % (nm, len(rs), sum(map(d, rs)) / 1e3))
