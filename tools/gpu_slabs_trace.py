#!/usr/bin/env python3
"""Where a decomposed run's time goes (one device, S slabs): run under
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o s -- python tools/gpu_slabs_trace.py run <grid> <S>
then  python tools/gpu_slabs_trace.py summary DIR/s  prints, per run, the span of the step launches, the span of everything behind
them (boundary histories, analysis, gathers) and the kernels / copies that fill it."""
import csv, os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import numpy as np
    import planeverb_amd.api as pv
    n, S = int(sys.argv[2]), int(sys.argv[3])
    dx = np.float32(343.21) / np.float32(275) / np.float32(3.5)
    size = float((n + 0.5) * dx)
    s = pv.Solver(size, size, 275, slabs=None if S == 1 else [0] * S)
    s.load_scene(os.path.join(ROOT, "tests", "scenes", "HugeRoom.pv"))
    for _ in range(3):
        s.run((5, 0, 4))
    t0 = time.perf_counter()
    for _ in range(5):
        s.run((5, 0, 4))
    print("grid %d slabs %d: %.3f ms per run (wall)" % (n, S, (time.perf_counter() - t0) / 5 * 1e3))
    s.close()
else:
    base = sys.argv[2]
    ev = []
    for r in csv.DictReader(open(base + "_kernel_trace.csv")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void pva::", "").replace("pva::", "")))
    if os.path.exists(base + "_memory_copy_trace.csv"):
        for r in csv.DictReader(open(base + "_memory_copy_trace.csv")):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
    ev.sort()
    # runs: separated by pv_begin_run_kernel launches (S of them per run, close together)
    begins = [e[0] for e in ev if "pv_begin_run_kernel" in e[2]]
    runs = []
    for b in begins:
        if not runs or b - runs[-1] > 500000:
            runs.append(b)
    runs.append(ev[-1][1] + 1)
    for i in range(len(runs) - 1):
        seg = [e for e in ev if runs[i] <= e[0] < runs[i + 1]]
        steps = [e for e in seg if "pv_step_" in e[2]]
        if not steps:
            continue
        s0, s1 = steps[0][0], max(e[1] for e in steps)
        end = max(e[1] for e in seg)
        busy = sum(e[1] - e[0] for e in steps)
        print("run %d: begin->first step %.0f us, step span %.0f us (%d launches, sum of durations %.0f us), behind the steps %.0f us" % (
            i, (s0 - runs[i]) / 1e3, (s1 - s0) / 1e3, len(steps), busy / 1e3, (end - s1) / 1e3))
        if i == len(runs) - 2:
            acc = collections.OrderedDict()
            for e in seg:
                if e[0] >= s1 - 1000:
                    k = acc.setdefault(e[2], [0, 0.0, None, None])
                    k[0] += 1
                    k[1] += (e[1] - e[0]) / 1e3
                    k[2] = (e[0] - s1) / 1e3 if k[2] is None else k[2]
                    k[3] = (e[1] - s1) / 1e3
            print("  behind the steps of the last run (name: calls, total us, first start, last end relative to the steps' end):")
            for n, k in acc.items():
                print("    %-44s %3d %8.1f %8.1f %8.1f" % (n[:44], k[0], k[1], k[2], k[3]))
            # the sweeps: gaps between consecutive step launches of one slab
            pushes = [e for e in seg if "halo_push" in e[2]]
            print("  step launches: mean %.1f us; halo pushes: %d, mean %.1f us" % (busy / 1e3 / len(steps), len(pushes), sum(e[1] - e[0] for e in pushes) / 1e3 / max(len(pushes), 1)))
