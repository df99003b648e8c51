#!/bin/bash
# tools/first_contact.sh -- the FIRST lease on a node with more than one MI355X, as one command.
# Nothing with N > 1 GPUs could be measured on the 1-GPU boxes this was built on: RCCL has only ever seen one rank
# (tests/test_gpu_configs.py), the slab decomposition only one device (tests/test_gpu_slabs.py), and the N > 1 orchestration
# only gloo on CPU (tests/test_dist_cpu.py, world 2 / 3 / 8).  This script runs, in order of what can go wrong first,
#   1. bench.py --gpus 2, then --gpus <all>, BASELINE config 4 (HugeRoom.pv at 4096^2, one run per GPU and step) with the gather
#      inside libplaneverb_amd.so (ncclAllGather through PvAmdComm) and again through torch.distributed (PV_BENCH_GATHER=torch);
#   2. the same for config 5 (open field at 8192^2, 64 listeners over the ranks);
#   3. one grid as row slabs ACROSS two devices (peer access, pushed halos: tools/gpu_slabs.py with SLAB_DEVICES=0,1) against
#      the same slabs on one device, and the slab ranks as two processes over RCCL (tools/slab_ranks_two_procs.py);
#   4. the multi-rank GPU tests,
# keeps every rank's record (bench.py prints them in its JSON line: block times, communicator seconds, how it gathered) and
# tars the lot.  Each step has its own timeout and failure does not stop the next.
#   tools/first_contact.sh [outdir]        (from the repository root; ~10 minutes on 8 GPUs)
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/first_contact}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
N=$(python -c "import torch; print(torch.cuda.device_count())")
echo "devices: $N" | tee "$OUT/summary.txt"
rocm-smi --showtopo > "$OUT/topology.txt" 2>&1 || true
if [ "$N" -lt 2 ]; then echo "first_contact: needs at least two GPUs" | tee -a "$OUT/summary.txt"; exit 2; fi
port=29600
run() {  # run <name> <timeout s> <command ...>: stdout + stderr to $OUT/<name>.log, the JSON line (if any) to summary.txt
  local name=$1 to=$2; shift 2
  echo "== $name: $*" | tee -a "$OUT/summary.txt"
  timeout "$to" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a "$OUT/summary.txt"
  grep '^{' "$OUT/$name.log" | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('   value %.4e %s on %d GPU(s), ms/step %.3f, verified %s of %s runs, gather: %s' % (d['value'], d['unit'], d['n_gpus'], d['ms_per_step'], d.get('verified_runs'), d.get('timed_runs'), d['config'].get('gather')))
    for r in d.get('ranks', []):
        print('     rank %s: blocks %s s, communicator + first gather %s s' % (r.get('rank'), r.get('block_s'), r.get('comm_init_and_first_gather_s')))
" 2>/dev/null | tee -a "$OUT/summary.txt"
  return $rc
}
bench() {  # bench <name> <gpus> <bench args ...>
  local name=$1 n=$2; shift 2
  port=$((port + 1))
  run "$name" 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus "$n" --steps 4 --warmup 2 --repeats 3 --no-cpu-baseline --no-dense-leg "$@"
}
run bench_1gpu 600 python bench.py --steps 4 --warmup 2 --repeats 3 --no-cpu-baseline --no-dense-leg
for n in 2 $N; do
  [ "$n" -gt "$N" ] && continue
  bench "cfg4_n${n}_native" "$n" --inflight 1
  PV_BENCH_GATHER=torch bench "cfg4_n${n}_torch" "$n" --inflight 1
  bench "cfg4_n${n}_2inflight" "$n"
  bench "cfg5_n${n}_native" "$n" --grid 8192 --open-field
  PV_BENCH_GATHER=torch bench "cfg5_n${n}_torch" "$n" --grid 8192 --open-field
done
SLABS=1,2,4 SLAB_DEVICES=0 run slabs_one_device 900 python tools/gpu_slabs.py 4096
SLABS=2,4,8 SLAB_DEVICES=0,1 run slabs_two_devices 900 python tools/gpu_slabs.py 4096
[ "$N" -ge 4 ] && SLABS=4,8 SLAB_DEVICES=0,1,2,3 run slabs_four_devices 900 python tools/gpu_slabs.py 4096 8192
port=$((port + 1))
PV_SLAB_DEVICES=0,1 PV_SLAB_BACKEND=nccl run slab_ranks_two_procs 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port $port tools/slab_ranks_two_procs.py 1024
run gpu_tests_dist 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_slabs.py -x -q -m gpu
tar czf "$OUT.tar.gz" -C "$(dirname "$OUT")" "$(basename "$OUT")"
echo "first_contact: records in $OUT/ and $OUT.tar.gz"
cat "$OUT/summary.txt"
