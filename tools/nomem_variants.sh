#!/bin/bash
# Runs HERE then on the GPU box: builds measurement variants of the library (air-tile arithmetic without its loads and
# stores, -DPV_PROBE_NOMEM=1, plus one knob each) and times them with one and two waves per SIMD (PV_PROBE_LDS).
#   tools/nomem_variants.sh build   (in the container)      tools/nomem_variants.sh run   (through gpurun)
cd "$(dirname "$0")/.."
VARIANTS=("base:" "g2:-DPV_MIRROR_G=2" "g6:-DPV_MIRROR_G=6" "nobar:-DPV_STEP_SCHEDBAR=0")
if [ "$1" = build ]; then
  for v in "${VARIANTS[@]}"; do
    n=${v%%:*}; f=${v#*:}
    touch planeverb_amd/csrc/pv_kernels.hip
    make -C planeverb_amd/csrc -j8 EXTRA="-DPV_PROBE_NOMEM=1 $f" > /dev/null 2>&1 && cp planeverb_amd/libplaneverb_amd.so planeverb_amd/libpv_nm_$n.so
  done
  touch planeverb_amd/csrc/pv_kernels.hip; make -C planeverb_amd/csrc -j8 > /dev/null 2>&1
else
  for v in "${VARIANTS[@]}"; do
    n=${v%%:*}
    for l in 0 90000; do
      echo -n "$n, extra LDS $l: "
      PLANEVERB_AMD_LIB=$PWD/planeverb_amd/libpv_nm_$n.so PV_PROBE_LDS=$l python tools/gpu_decompose.py 4096 2>&1 | head -1
    done
  done
fi
