E=$PWD/planeverb_amd/libplaneverb_amd_exp.so
run() { echo "## $*"; local t0=$SECONDS; "$@" 2>&1 | tail -1; echo "   $((SECONDS - t0)) s"; }
{
PV_FUZZ_CONFIG=resident_kernel=-1 run python tools/gpu_fuzz.py 700000 1500
run python tools/gpu_fuzz_ref.py 710000 1000
PLANEVERB_AMD_LIB=$E run python tools/gpu_fuzz.py 720000 700
run python tools/gpu_fuzz_nearbox.py 730000 1500
run python tools/gpu_slab_stress.py 9000 60 4
} > gpurun_out/r06_fuzz_raw2.txt 2>&1
cat gpurun_out/r06_fuzz_raw2.txt | grep -v "^seed"
