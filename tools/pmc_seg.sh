#!/bin/bash
# Runs ON THE GPU BOX: SQ / instruction-cache counters of the step kernels on the raw stencil workload
# (tools/sq_workload.py), row-streaming segments vs tile kernels.   tools/pmc_seg.sh <tag> <segments> [fields]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_seg_$1
rm -rf $O && mkdir -p $O
W="python tools/sq_workload.py --fields ${3:-zero} --inflight 1 --reps 1 --segments $2 $4"
$W > $O/wall.txt 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAIT_IFETCH" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- $W > /dev/null 2> $O/p$i.err
done
python - "$O" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pv_step" not in k: continue
        acc[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pv_step" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][-60:]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print(open(sys.argv[1] + "/wall.txt").read().strip())
for k, d in acc.items():
    print("==", k, "duration under pmc median %.1f us" % (sorted(dur[k])[len(dur[k]) // 2] / 1e3 if dur[k] else -1))
    for c, v in sorted(d.items()):
        v = sorted(v)
        print("  %-32s n=%4d median %.5g" % (c, len(v), v[len(v)//2]))
PY
