#!/bin/bash
# runs ON THE GPU BOX: HBM-side bytes per launch of the patch kernel / the tile kernel (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
for cfg in "1 4096 3" "0 4096 3"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pp; timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pp -o p -- python /root/repo/tools/gpu_patch_prof.py $cfg > /tmp/p.log 2>&1 </dev/null
    python3 - "$cfg" $c <<'PY'
import csv, sys, glob
f = glob.glob('/tmp/pp/*counter_collection.csv')
if not f:
    print(sys.argv[1], sys.argv[2], 'no counter file'); sys.exit()
rows = list(csv.DictReader(open(f[0])))
acc = {}
for r in rows:
    k = r['Kernel_Name'][:40]
    acc.setdefault(k, []).append(float(r['Counter_Value']))
for k, v in acc.items():
    if 'step' in k:
        v = sorted(v)
        print(sys.argv[1], sys.argv[2], k, 'n', len(v), 'median', v[len(v)//2], 'max', v[-1])
PY
  done
done
