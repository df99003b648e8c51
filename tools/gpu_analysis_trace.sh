#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of the all-cells-reached analysis workload (tools/gpu_analysis_workload.py) -> per-kernel
# average durations of the analysis chain.  usage: tools/gpu_analysis_trace.sh <out-dir> [LIB.so] [res=2009 ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$1; shift
LIB=${1:-}; shift
mkdir -p $O
[ -n "$LIB" ] && export PLANEVERB_AMD_LIB=$PWD/$LIB
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o a -- python tools/gpu_analysis_workload.py 6 "$@" > $O/workload.txt 2>&1
python - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/a_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "step" in n or "rocclr" in n: continue
    print("%-60s calls %4s avg %9.1f us  min %9.1f" % (n.split("(")[0][-60:] if not n.startswith("void pva::(anon") else n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
cat $O/workload.txt
