#!/bin/bash
# Runs ON THE GPU BOX: per-kernel average durations of a preset's run (tools/gpu_presets.py <res>, batch API only).
# usage: tools/gpu_preset_trace.sh <out-dir> <res> [LIB.so] [option=value ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$1; R=$2; LIB=${3:-}; shift; shift; shift
mkdir -p $O
[ -n "$LIB" ] && export PLANEVERB_AMD_LIB=$PWD/$LIB
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python tools/gpu_presets.py $R analysis_fork=${FORK:-1} "$@" > $O/presets.txt 2>&1
python - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/p_kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "rocclr" in n or int(r["Calls"]) < 10: continue
    print("%-64s calls %4s avg %8.1f us" % (n.split("(")[0][-64:] if not n.startswith("void pva::(anon") else n[:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep -v "^#" $O/presets.txt
