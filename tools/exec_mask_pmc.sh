#!/bin/bash
# Runs ON THE GPU BOX: VALU counters of the dominant kernel with and without the EXEC trapezoid (PV_EXEC_TRAPEZOID, pv_kernels.hip)
# on seeded random fields, one launch at a time: SQ_INSTS_VALU (instructions), SQ_ACTIVE_INST_VALU (issue quad-cycles),
# SQ_THREAD_CYCLES_VALU (lanes x cycles that executed).  Own --pmc pass with --kernel-trace only.
#   tools/exec_mask_pmc.sh <a.so> <b.so> ...      (names under planeverb_amd/)  -> gpurun_out/r06_exec_pmc/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_exec_pmc
rm -rf $O && mkdir -p $O
for so in "$@"; do
  for f in random zero; do
    PLANEVERB_AMD_LIB=$PWD/planeverb_amd/$so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE \
      --output-format csv -d $O/${so%.so}_$f -o p -- python tools/sq_workload.py --fields $f --inflight 1 --reps 1 > /dev/null 2> $O/${so%.so}_$f.err
  done
done
python - "$O" "$@" > $O/summary.txt <<'PY'
import collections, csv, glob, os, sys
O = sys.argv[1]
print("%-28s %-7s %14s %20s %22s %16s %8s" % ("library", "fields", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE", "us"))
for so in sys.argv[2:]:
    for f in ("random", "zero"):
        acc, dur = collections.defaultdict(list), []
        for p in glob.glob(os.path.join(O, so[:-3] + "_" + f, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(p)):
                if "pv_step_merged_kernel" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
        print("%-28s %-7s %14.4e %20.4e %22.4e %16.4e %8.1f" % (so, f, med(acc["SQ_INSTS_VALU"]), med(acc["SQ_ACTIVE_INST_VALU"]),
                                                               med(acc["SQ_THREAD_CYCLES_VALU"]), med(acc["GRBM_GUI_ACTIVE"]), med(dur)))
PY
cat $O/summary.txt
