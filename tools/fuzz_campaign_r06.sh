#!/bin/bash
# Runs ON THE GPU BOX: round 6's parity campaign (tools/gpu_fuzz.py, gpu_fuzz_ref.py, gpu_slab_stress.py) -> gpurun_out/r06_fuzz_raw.txt
E=$PWD/planeverb_amd/libplaneverb_amd_exp.so
run() { echo "## $*"; local t0=$SECONDS; "$@" 2>&1 | tail -3; echo "   $((SECONDS - t0)) s"; }
{
PV_FUZZ_CONFIG=resident_kernel=0 run python tools/gpu_fuzz.py 500000 1500
PV_FUZZ_CONFIG=resident_kernel=-1 run python tools/gpu_fuzz.py 510000 1000
PLANEVERB_AMD_LIB=$E run python tools/gpu_fuzz.py 520000 800
run python tools/gpu_fuzz_ref.py 530000 1000
PLANEVERB_AMD_LIB=$E PV_FUZZ_CONFIG=fused_analysis=1 run python tools/gpu_fuzz.py 540000 500
PV_FUZZ_CONFIG=steps_per_launch=12,tile_rows=36 run python tools/gpu_fuzz.py 550000 500
PLANEVERB_AMD_LIB=$E PV_FUZZ_CONFIG=steps_per_launch=12,tile_rows=36 run python tools/gpu_fuzz.py 560000 500
PLANEVERB_AMD_NEAR_BOX=0 PLANEVERB_AMD_STAMP_TIMINGS=0 PV_FUZZ_CONFIG=resident_kernel=-1 run python tools/gpu_fuzz.py 570000 300
run python tools/gpu_slab_stress.py 8000 60 4
} > gpurun_out/r06_fuzz_raw.txt 2>&1
grep -c mismatch gpurun_out/r06_fuzz_raw.txt; tail -3 gpurun_out/r06_fuzz_raw.txt
