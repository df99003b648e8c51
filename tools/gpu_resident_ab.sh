#!/bin/bash
# Runs ON THE GPU BOX: alternating A/B of library builds on the presets' stencil time (resident kernel).
#   tools/gpu_resident_ab.sh <a.so> <b.so> [rounds]
for i in $(seq 1 ${3:-3}); do
  for so in "$1" "$2"; do
    echo "$so: $(PLANEVERB_AMD_LIB=$PWD/$so python tools/gpu_resident_trace.py 275 375 500 750 1000 2>&1 | grep -E '## fdtd' | cut -c9-13 | tr '\n' ' ')"
  done
done
