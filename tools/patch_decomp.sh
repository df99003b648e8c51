#!/bin/bash
# runs ON THE GPU BOX: the patch kernel with parts switched off (rocprofv3 per-kernel averages)
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 4 6 7 14 15 9; do
  rm -rf /tmp/pp; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python /root/repo/tools/gpu_patch_prof.py 1 4096 3 $dbg > /tmp/p.log 2>&1 </dev/null
  echo "dbg=$dbg  $(grep patch_kernel /tmp/pp/p_kernel_stats.csv | cut -d, -f2,4,6,7)"
done
