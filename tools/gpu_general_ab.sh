for i in 1 2 3; do
 for so in libplaneverb_amd.so libplaneverb_amd_gp.so; do
  echo "$so modeB whole:   $(PLANEVERB_AMD_LIB=$PWD/planeverb_amd/$so MODEB=1 python tools/gpu_dense.py 4096 1 12,36 2>&1 | tail -1)"
  echo "$so modeB general: $(PLANEVERB_AMD_LIB=$PWD/planeverb_amd/$so MODEB=1 PV_PROBE_GENERAL_ONLY=1 python tools/gpu_dense.py 4096 1 12,36 2>&1 | tail -1)"
 done
done
