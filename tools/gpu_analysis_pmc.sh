cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" ; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r05/pmc_x -o p -- env FORK=0 python tools/gpu_analysis_workload.py 2 > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/r05/pmc_x/**/p_counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "rt60_tile" in k or "encode" in k or "onset" in k:
        kk="rt60_tile" if "rt60_tile" in k else ("encode" if "encode" in k else "onset")
        acc[kk][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(kk,r["Counter_Name"])]+=1
for kk in acc:
    print(kk, {c: "%.3g"%(v/n[(kk,c)]) for c,v in acc[kk].items()})
PY
rm -rf gpurun_out/r05/pmc_x
done
