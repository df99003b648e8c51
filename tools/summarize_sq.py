#!/usr/bin/env python3
"""Summarise the counter passes of tools/pmc_r02.sh (run on the GPU box) as markdown on stdout."""
import collections
import csv
import glob
import os
import subprocess
import sys

O = sys.argv[1]
KERNEL = "pv_step_merged_kernel"
CU_NUM, SIMDS = 256, 1024


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float("nan")


def counters(w):
    """per-dispatch counter values of the step kernel: {counter: [values]} (full K-step launches only)"""
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(O, w, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations(w):
    """kernel-trace durations (ns) of the step kernel, un-profiled passes: list"""
    out = []
    for f in glob.glob(os.path.join(O, w, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                out.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return out


def pmc_durations(w):
    out = []
    for f in glob.glob(os.path.join(O, w, "p1", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                out.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return out


try:
    head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    head = "(snapshot without .git)"
print("# SQ / GRBM / TCC counters of `pv_step_merged_kernel<12,36>` -- raw stencil, 4096^2, HugeRoom.pv geometry")
print()
print("Collected by `tools/pmc_r02.sh` (rocprofv3 `--pmc` in separate passes with `--kernel-trace` only).  Medians over "
      "the full 12-step launches of one pass.  Units: `SQ_*_CYCLES` / `SQ_ACTIVE_INST_*` / `SQ_WAIT_*` count "
      "quad-cycles summed over waves (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs (it reads 8 x "
      "clock x duration), so per-XCD cycles = GRBM_GUI_ACTIVE / 8 and effective clock = that / launch duration of the same "
      "pass (the guide's DVFS recipe); VALU issue utilisation = VALU quad-cycles x 4 / 1024 SIMDs / per-XCD cycles.  Under `--pmc` the profiler serialises dispatches, so counters exist for single launches only; the "
      "`2 in flight` rows come from the counter-free kernel trace and the wall rate.")
print()
rows = {}
for w in ("zero1", "random1"):
    c = counters(w)
    if not c:
        continue
    m = {k: med(v) for k, v in c.items()}
    dur = med(pmc_durations(w))
    rows[w] = (m, dur)
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES",
         "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA",
         "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INST_CYCLES_VMEM", "SQ_THREAD_CYCLES_VALU",
         "GRBM_GUI_ACTIVE", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "FETCH_SIZE",
         "WRITE_SIZE"]
print("| counter (per launch, median) | zero fields | random fields | random / zero |")
print("|---|---|---|---|")
for n in names:
    z = rows.get("zero1", ({}, 0))[0].get(n)
    r = rows.get("random1", ({}, 0))[0].get(n)
    if z is None and r is None:
        continue
    print("| %s | %s | %s | %s |" % (n, "%.4g" % z if z is not None else "-", "%.4g" % r if r is not None else "-",
                                    "%.3f" % (r / z) if z and r else "-"))
print()
print("| derived (single launch, profiled pass) | zero fields | random fields |")
print("|---|---|---|")


def derived(w):
    m, dur = rows[w]
    g = m.get("GRBM_GUI_ACTIVE", float("nan"))
    d = collections.OrderedDict()
    d["launch duration under --pmc (us)"] = dur / 1e3
    # GRBM_GUI_ACTIVE comes back SUMMED over the 8 XCDs (8 x 2.4 GHz x duration): per-XCD cycles = / 8
    d["effective clock = GRBM_GUI_ACTIVE / 8 / duration (GHz)"] = g / 8 / dur
    d["VALU issue utilisation = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)"] = (
        m.get("SQ_ACTIVE_INST_VALU", float("nan")) * 4 / SIMDS / (g / 8))
    d["any-instruction issue utilisation = SQ_ACTIVE_INST_ANY x 4 / 1024 / (GRBM / 8)"] = (
        m.get("SQ_ACTIVE_INST_ANY", float("nan")) * 4 / SIMDS / (g / 8))
    d["mean resident waves per SIMD = SQ_WAVE_CYCLES x 4 / 1024 / (GRBM / 8)"] = (
        m.get("SQ_WAVE_CYCLES", float("nan")) * 4 / SIMDS / (g / 8))
    d["VALU instructions per wave"] = m.get("SQ_INSTS_VALU", float("nan")) / m.get("SQ_WAVES", float("nan"))
    d["active lanes per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64... of 64)"] = (
        m.get("SQ_THREAD_CYCLES_VALU", float("nan")) / m.get("SQ_ACTIVE_INST_VALU", float("nan")))
    wc = m.get("SQ_WAVE_CYCLES", float("nan"))
    d["share of wave cycles: issuing (ACTIVE_INST_ANY)"] = m.get("SQ_ACTIVE_INST_ANY", float("nan")) / wc
    d["share of wave cycles: issue stall (WAIT_INST_ANY)"] = m.get("SQ_WAIT_INST_ANY", float("nan")) / wc
    d["share of wave cycles: parked (WAIT_ANY: s_waitcnt / barrier)"] = m.get("SQ_WAIT_ANY", float("nan")) / wc
    d["HBM read MB (FETCH_SIZE KiB x 1024 x 2, gfx950 correction)"] = m.get("FETCH_SIZE", float("nan")) * 1024 * 2 / 1e6
    d["HBM write MB (WRITE_SIZE KiB x 1024)"] = m.get("WRITE_SIZE", float("nan")) * 1024 / 1e6
    d["L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS)"] = m.get("TCC_HIT_sum", float("nan")) / (
        m.get("TCC_HIT_sum", float("nan")) + m.get("TCC_MISS_sum", float("nan")))
    return d


dz = derived("zero1") if "zero1" in rows else {}
dr = derived("random1") if "random1" in rows else {}
for k in (dz or dr):
    print("| %s | %s | %s |" % (k, "%.4g" % dz[k] if k in dz else "-", "%.4g" % dr[k] if k in dr else "-"))
print()
print("| un-profiled (kernel trace only) | launches | median launch us | p10 | p90 | wall rate |")
print("|---|---|---|---|---|---|")
for w in ("zero1", "zero2", "random1", "random2"):
    d = sorted(durations(w))
    if not d:
        continue
    wall = open(os.path.join(O, w + ".wall.txt")).read().strip().splitlines()[-1] if os.path.exists(os.path.join(O, w + ".wall.txt")) else ""
    print("| %s | %d | %.1f | %.1f | %.1f | %s |" % (w, len(d), med(d) / 1e3, d[len(d) // 10] / 1e3, d[9 * len(d) // 10] / 1e3, wall))
