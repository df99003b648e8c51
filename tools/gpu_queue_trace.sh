#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
for i in 1 2 3 4 5 6 7 8; do
  rm -rf /tmp/qt_$i
  rocprofv3 --kernel-trace --output-format csv -d /tmp/qt_$i -o q -- python bench.py --no-cpu-baseline --no-dense-leg --steps 4 --repeats 3 > /tmp/qt_$i.json 2>/dev/null
  python - /tmp/qt_$i /tmp/qt_$i.json <<'PY'
import csv, glob, json, sys, collections
f = glob.glob(sys.argv[1] + "/**/q_kernel_trace.csv", recursive=True)[0]
c = collections.Counter()
for r in csv.DictReader(open(f)):
    if "pv_step_merged_kernel" in r["Kernel_Name"]:
        c[(r["Queue_Id"], r.get("Stream_Id", "?"))] += 1
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%.4g" % d["value"], "pair launch %.4f" % d["roofline"]["launch_ms"], dict(c))
PY
done
