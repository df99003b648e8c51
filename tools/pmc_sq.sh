#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the step kernel on the bench workload (own passes, kernel trace only).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_sq
rm -rf $O && mkdir -p $O
ARGS="${@:---no-cpu-baseline --steps 2 --warmup 1}"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python bench.py $ARGS > /dev/null 2> $O/p$i.err
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pv_step" not in k: continue
        acc[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("==", k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print("  %-32s n=%4d median %.4g mean %.4g" % (c, len(v), v[len(v)//2], sum(v)/len(v)))
PY
