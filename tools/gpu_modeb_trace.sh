#!/bin/bash
# Runs ON THE GPU BOX: kernel trace + stats of one Mode B run (tools/modeb_probe.py) -> per-kernel totals of the run.
#   tools/gpu_modeb_trace.sh <out dir> <res> [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$1; R=$2; shift 2
rm -rf $O && mkdir -p $O
env "$@" RUNS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o m -- python tools/modeb_probe.py $R > $O/run.txt 2>&1
python - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/m_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(open(sys.argv[1] + "/run.txt").read().splitlines()[0])
print("%-58s %8s %10s %9s %6s" % ("kernel", "calls", "total ms", "avg us", "%"))
for r in rows[:14]:
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:58]
    print("%-58s %8s %10.2f %9.2f %6.1f" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
