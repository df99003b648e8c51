#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "512 Shoebox.pv 2 8 10" "4096 HugeRoom.pv 2 1 8"; do
set -- $cfg
O=gpurun_out/b$1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python bench.py --no-cpu-baseline --grid $1 --scene $2 --inflight $3 --batch $4 --steps $5 > $O/bench.json 2> $O/err.txt
find $O -name "*kernel_stats.csv" | head -1 | xargs cat | head -9 | cut -c1-120
tail -1 $O/bench.json | cut -c1-80
done
