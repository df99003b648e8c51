#!/bin/bash
# Runs ON THE GPU BOX: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the step kernels on the raw stencil
# workload.   tools/pmc_hbm_seg.sh <tag> "<sq_workload args>"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_hbm_$1
rm -rf $O && mkdir -p $O
W="python tools/sq_workload.py --fields zero --inflight 1 --reps 1 $2"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- $W > /dev/null 2> $O/$c.err
done
python - "$O" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pv_step" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-50:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    f = sorted(d["FETCH_SIZE"]); w = sorted(d["WRITE_SIZE"])
    fm, wm = f[len(f) // 2], w[len(w) // 2]
    print("%s: FETCH_SIZE %.0f KiB x2 (gfx950) = %.1f MB read, WRITE_SIZE %.0f KiB = %.1f MB written, total %.1f MB per launch" % (
        k, fm, fm * 2 * 1024 / 1e6, wm, wm * 1024 / 1e6, (fm * 2 + wm) * 1024 / 1e6))
PY
