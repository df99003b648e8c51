#!/bin/bash
# Runs ON THE GPU BOX: alternating A/B of two library builds (PLANEVERB_AMD_LIB) over a command that prints one line.
#   tools/gpu_ab_libs.sh <a.so> <b.so> <rounds> <command ...>
a=$1; b=$2; n=$3; shift 3
for i in $(seq 1 $n); do
  for so in "$a" "$b"; do
    echo "$(basename $so): $(PLANEVERB_AMD_LIB=$PWD/$so "$@" 2>/dev/null | tail -1)"
  done
done
