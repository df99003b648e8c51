#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): SQ / GRBM / TCC counter passes of the dominant kernel (own passes: --pmc with
# --kernel-trace only) for four workloads -- {zero, random} fields x {1, 2} runs in flight -- plus the un-profiled wall
# rates, and writes gpurun_out/r02_sq/summary.md (copied to profiles/r02_sq_pmc.md).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_sq
rm -rf $O && mkdir -p $O
SETS=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
      "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
      "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
      "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"
      "FETCH_SIZE"
      "WRITE_SIZE")
for f in zero random; do
  for n in 1 2; do
    W="$f$n"
    python tools/sq_workload.py --fields $f --inflight $n > $O/$W.wall.txt 2>&1
    # kernel trace without counters: the un-serialised launch durations
    rocprofv3 --kernel-trace --output-format csv -d $O/$W/trace -o t -- python tools/sq_workload.py --fields $f --inflight $n > /dev/null 2> $O/$W.trace.err
    [ $n = 2 ] && continue   # under --pmc dispatches are serialised: the counters of n = 2 would repeat n = 1
    i=0
    for set in "${SETS[@]}"; do
      i=$((i+1))
      rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$W/p$i -o p -- python tools/sq_workload.py --fields $f --inflight $n --reps 1 > /dev/null 2> $O/$W.p$i.err
    done
  done
done
python tools/summarize_sq.py $O > $O/summary.md
cat $O/summary.md
