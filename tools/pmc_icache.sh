#!/bin/bash
# Runs ON THE GPU BOX: instruction-cache counters of the step kernel on the bench workload (development aid).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_ic_$1
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/p -o p -- python bench.py --no-cpu-baseline --inflight 1 --steps 2 --warmup 1 > /dev/null 2> $O/p.err
python - "$O" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pv_step" not in k: continue
        acc[k.split("(")[0][-50:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("==", k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print("  %-30s n=%4d median %.4g" % (c, len(v), v[len(v)//2]))
PY
