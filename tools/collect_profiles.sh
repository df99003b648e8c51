#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + stats of the default bench command, the two
# HBM PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes), and the counter
# calibration.  Everything lands in gpurun_out/; tools/summarize_profiles.py turns it into profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r02}
O=gpurun_out/$R
rm -rf $O && mkdir -p $O
# (--steps 20: the kernel average below then is dominated by the timed, overlapped launches -- 2 of the 42 runs of the
# command are warm-up runs made one at a time)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python bench.py --no-cpu-baseline --no-dense-leg --steps 20 --warmup 1 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py --no-cpu-baseline --no-dense-leg --steps 3 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py --no-cpu-baseline --no-dense-leg --steps 3 --warmup 1 > /dev/null 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch8k -o f -- python bench.py --no-cpu-baseline --no-dense-leg --grid 8192 --steps 2 --warmup 1 > /dev/null 2> $O/pmc_fetch8k.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write8k -o w -- python bench.py --no-cpu-baseline --no-dense-leg --grid 8192 --steps 2 --warmup 1 > /dev/null 2> $O/pmc_write8k.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch2k -o f -- python bench.py --no-cpu-baseline --no-dense-leg --grid 2048 --scene BigRoom.pv --steps 3 --warmup 1 > /dev/null 2> $O/pmc_fetch2k.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write2k -o w -- python bench.py --no-cpu-baseline --no-dense-leg --grid 2048 --scene BigRoom.pv --steps 3 --warmup 1 > /dev/null 2> $O/pmc_write2k.err
hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o /tmp/hbm_calib 2>/dev/null
/tmp/hbm_calib > $O/hbm_calib.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -o c -- /tmp/hbm_calib > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -o c -- /tmp/hbm_calib > /dev/null 2>&1
# the un-profiled bench line, same box
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --grid 8192 --steps 4 --no-cpu-baseline > $O/bench_8192.json 2>> $O/bench.err
python bench.py --grid 8192 --steps 4 --open-field --no-cpu-baseline > $O/bench_8192_open.json 2>> $O/bench.err
python bench.py --grid 2048 --scene BigRoom.pv --no-cpu-baseline > $O/bench_2048.json 2>> $O/bench.err
python bench.py --grid 512 --scene Shoebox.pv --inflight 4 --no-cpu-baseline > $O/bench_512.json 2>> $O/bench.err
python bench.py --grid 512 --scene Shoebox.pv --inflight 2 --batch 0 --no-cpu-baseline > $O/bench_512_batch8.json 2>> $O/bench.err
python bench.py --grid 1024 --scene Shoebox.pv --inflight 2 --batch 8 --no-cpu-baseline > $O/bench_1024_batch8.json 2>> $O/bench.err
python bench.py --dense-history 1 --no-cpu-baseline > $O/bench_dense.json 2>> $O/bench.err
python bench.py --inflight 1 --no-cpu-baseline > $O/bench_inflight1.json 2>> $O/bench.err
# sizes, open fields, runs in flight, batched launches
python tools/gpu_probe.py perf > $O/sizes.txt 2>&1
python tools/gpu_openfield.py >> $O/sizes.txt 2>&1
(python tools/gpu_concurrent.py 4096 2; python tools/gpu_concurrent.py 2048 2; python tools/gpu_concurrent.py 1024 4; python tools/gpu_concurrent.py 512 4; python tools/gpu_concurrent.py 8192 2) > $O/concurrent.txt 2>&1
python tools/gpu_batch.py 512,1024,2048 "1x1 4x1 1x8 2x8" Shoebox.pv > $O/batch.txt 2>&1
python tools/gpu_batch.py 4096 "2x1 1x2 3x1" >> $O/batch.txt 2>&1
rm -f $O/trace/bench_kernel_trace.csv.bak
ls -la $O
# SQ / GRBM counters of the dominant kernel, zero vs random fields (-> gpurun_out/r02_sq, summarised into profiles/<round>_sq_pmc.md)
bash tools/pmc_r02.sh > $O/pmc_r02.log 2>&1
python tools/gpu_live_rate.py --out $O/live_rate.txt > /dev/null 2>&1
python tools/gpu_slabs.py 4096 2048 > $O/slabs_one_device.txt 2>&1
python tools/modeb_probe.py 2009 8034 16067 32134 > $O/modeB_streaming.txt 2>&1
# round 3 additions: the reference's resolution presets, 1-rank RCCL self-test of bench.py's distributed path, patch-kernel A/B
python tools/gpu_presets.py > $O/presets.txt 2>&1
PV_BENCH_FORCE_DIST=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err
# (the patch kernel lives in the experimental build: PLANEVERB_AMD_LIB=$PWD/planeverb_amd/libplaneverb_amd_exp.so python tools/gpu_patch.py 4096 3)
# round 4 additions: the all-cells-reached analysis workload (kernel stats + the two HBM PMC passes), the presets through the resident
# kernel (table + kernel trace), live module with one / two iterations in flight, the decay-time forms
python tools/gpu_analysis_workload.py 6 > $O/analysis_workload.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ana -o a -- python tools/gpu_analysis_workload.py 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_ana -o f -- python tools/gpu_analysis_workload.py 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_ana -o w -- python tools/gpu_analysis_workload.py 3 > /dev/null 2>&1
python tools/gpu_resident.py 275 375 500 750 1000 1250 1500 stress=10 > $O/presets.txt 2>&1
for r in 275 750; do rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_presets_$r -o p -- python tools/gpu_presets.py $r use_graph=0 > /dev/null 2>&1; done
(for p in 1 2; do echo "== PLANEVERB_AMD_LIVE_PIPELINE=$p"; PLANEVERB_AMD_LIVE_PIPELINE=$p python tools/gpu_presets.py 275 375 500 750 1000 2>&1 | grep -v "^#"; done) > $O/live_pipeline.txt 2>&1
python tools/gpu_run_times.py > $O/run_times.txt 2>&1
# round 5 additions: raw stencil on random vs zero fields (bench.py's roofline.dense leg) across tiles, per-kernel traces of two presets
python tools/gpu_dense.py 4096 2 12,36 10,36 8,40 > $O/dense.txt 2>&1
python tools/gpu_dense.py 4096 1 12,36 2>&1 | tail -1 >> $O/dense.txt
python tools/gpu_dense.py 8192 2 12,36 2>&1 | tail -1 >> $O/dense.txt
LANES=16,4,1,0 python tools/gpu_rt60.py 275 375 500 750 1000 1250 1500 2009 > $O/rt60.txt 2>&1
for r in 275 750 1000 1500; do tools/gpu_preset_trace.sh $O/preset_trace_$r $r "" > $O/preset_trace_$r.txt 2>/dev/null; done
tools/gpu_analysis_trace.sh $O/analysis_trace "" > $O/analysis_trace.txt 2>/dev/null
# phase stamps of the resident kernel (needs the trace build, made HERE before the call: see tools/gpu_resident_trace.py)
if [ -f planeverb_amd/libplaneverb_amd_trace.so ]; then PLANEVERB_AMD_LIB=$PWD/planeverb_amd/libplaneverb_amd_trace.so python tools/gpu_resident_trace.py 275 750 > $O/resident_trace.txt 2>&1; fi
# (round 5, later additions) slab groups in several process orders, the general arm of Mode B's geometry, the persistent-form probe
(for i in 1 2; do echo "== process $i: 4096 then 2048, S = 1, 2, 4, 8"; python tools/gpu_slabs.py 4096 2048 2>&1 | cut -c1-100; echo "== process: S = 2 only, 2048 4096 2048 4096"; SLABS=2 python tools/gpu_slabs.py 2048 4096 2048 4096 2>&1 | cut -c1-100; done) > $O/slabs_orders.txt 2>&1
(for m in "" 1; do MODEB=$m python tools/gpu_dense.py 4096 1 12,36 2>&1 | tail -1; MODEB=$m PV_PROBE_GENERAL_ONLY=1 python tools/gpu_dense.py 4096 1 12,36 2>&1 | tail -1; done) > $O/general_arm.txt 2>&1
tools/gpu_modeb_trace.sh $O/modeb_trace 16067 > $O/modeb_trace_4096.txt 2>/dev/null
hipcc --offload-arch=gfx950 -O3 tools/persist_probe.hip -o /tmp/persist_probe 2>/dev/null && timeout 120 /tmp/persist_probe > $O/persist_probe.txt 2>&1
# round 6 additions: the timeline of the analysis chain behind a lone run (bench grid, three presets), the decay-time forms' table
# with the re-measured thresholds, the presets' run times with event-timed runs beside (PLANEVERB_AMD_STAMP_TIMINGS=0)
(for w in 4096 res:275 res:750 res:1000; do echo "== $w"; tools/gpu_chain_trace.sh $O/chain_$w $w 2>/dev/null | grep -v simple_timer; done) > $O/analysis_chain.txt 2>&1
(echo "# stamps (default):"; python tools/gpu_run_times.py; echo "# PLANEVERB_AMD_STAMP_TIMINGS=0 (HIP events between the kernels of a resident run):"; PLANEVERB_AMD_STAMP_TIMINGS=0 python tools/gpu_run_times.py; echo "# analysis ms (tools/gpu_analysis_times.py):"; python tools/gpu_analysis_times.py; echo "# PLANEVERB_AMD_NEAR_BOX=0:"; PLANEVERB_AMD_NEAR_BOX=0 python tools/gpu_analysis_times.py) > $O/run_times_r06.txt 2>&1
