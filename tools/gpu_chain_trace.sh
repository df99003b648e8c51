#!/bin/bash
# Runs ON THE GPU BOX: the TIMELINE of one run's analysis chain -- every kernel behind the run's last stencil launch: start (us after
# that launch's end), duration, queue -- from a rocprofv3 kernel trace of a few lone runs (the last one is printed).
#   tools/gpu_chain_trace.sh <out-dir> <grid: 4096 | res:275 ...> [LIB.so]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$1; W=$2; LIB=${3:-}
mkdir -p $O
[ -n "$LIB" ] && export PLANEVERB_AMD_LIB=$PWD/$LIB
cat > $O/w.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import planeverb_amd.api as pv
w = sys.argv[1]
if w.startswith("res:"):
    s = pv.Solver(25.0, 25.0, int(w[4:]), no_free_grid=1); s.load_scene("tests/scenes/SmallRoomScene.pv")
else:
    dx = float(np.float32(343.21) / np.float32(275) / np.float32(3.5)); n = int(w)
    s = pv.Solver((n + 0.5) * dx, (n + 0.5) * dx, 275, no_free_grid=1); s.load_scene("tests/scenes/HugeRoom.pv")
for L in ((5.0, 0.0, 4.0), (8.0, 0.0, 8.0), (12.0, 0.0, 6.0), (15.0, 0.0, 15.0)):
    s.run(L)
print("analysis ms", s.timings().analysisMs, "fdtd ms", s.timings().fdtdMs)
s.close()
PY
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o c -- python $O/w.py $W > $O/out.txt 2>&1
python - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/c_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
steps = [i for i, r in enumerate(rows) if "pv_step" in r["Kernel_Name"] or "pv_resident" in r["Kernel_Name"] or "pv_small_grid" in r["Kernel_Name"]]
last = steps[-1]
t0 = int(rows[last]["End_Timestamp"])
print("%-58s %10s %10s %6s" % ("kernel (behind the last stencil launch of the last run)", "start us", "dur us", "queue"))
for r in rows[last + 1:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pva::", "").replace("(anonymous namespace)::", "")
    print("%-58s %10.1f %10.1f %6s" % (n[:58], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Queue_Id"]))
PY
cat $O/out.txt | tail -2
