#!/usr/bin/env python3
"""Turn gpurun_out/<round>/ (written by tools/collect_profiles.sh on the GPU box) into the committed summaries under
profiles/:  <round>_kernel_stats.csv (rocprofv3 --stats), <round>_hbm_pmc.md/json (FETCH_SIZE/WRITE_SIZE per launch
with the gfx950 correction and its calibration), <round>_bench*.json, and profiles/hbm_traffic.json (read by bench.py
for roofline.traffic)."""
import collections
import csv
import json
import os
import shutil
import sys


def short_name(k):
    """kernel name without return type, namespaces and argument list ('(anonymous namespace)::' holds the first parenthesis of
    the kernels of pv_rt60.hip: cut at it, round 4's tables had blank rows for them)"""
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("pva::", "")
    return k.split("(")[0]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
SRC = os.path.join(ROOT, "gpurun_out", R)
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)


def counters(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def steady(v):
    """per-launch values of the T-step loop: drop the short remainder launch and warm-up outliers -> median"""
    return med(v)


shutil.copy(os.path.join(SRC, "trace", "bench_kernel_stats.csv"), os.path.join(DST, R + "_kernel_stats.csv"))
# the dominant kernel's launches split by what ran beside them (the stats file above averages the warm-up's lone launches with
# the timed, paired ones: 217 us in round 5, which matched neither quoted figure): a launch counts as PAIRED when launches of the
# same kernel on another queue overlap more than half of it
try:
    tr = [r for r in csv.DictReader(open(os.path.join(SRC, "trace", "bench_kernel_trace.csv"))) if "pv_step_merged_kernel" in r["Kernel_Name"]]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in tr)
    full = sorted(e[1] - e[0] for e in ev)
    cut = 0.6 * full[len(full) // 2]  # (the short remainder launch of every run: 3 of 435 steps)
    groups = {"lone": [], "paired": []}
    for i, (b0, e0, q0) in enumerate(ev):
        if e0 - b0 < cut:
            continue
        ov = 0
        j = i - 1
        while j >= 0 and ev[j][0] > b0 - 2 * (e0 - b0):
            if ev[j][2] != q0:
                ov += max(0, min(e0, ev[j][1]) - max(b0, ev[j][0]))
            j -= 1
        j = i + 1
        while j < len(ev) and ev[j][0] < e0:
            if ev[j][2] != q0:
                ov += max(0, min(e0, ev[j][1]) - max(b0, ev[j][0]))
            j += 1
        groups["paired" if ov > 0.5 * (e0 - b0) else "lone"].append((e0 - b0) / 1e3)
    with open(os.path.join(DST, R + "_kernel_stats_split.txt"), "w") as fh:
        fh.write("# pv_step_merged_kernel, full K-step launches of %s_kernel_stats.csv's trace split by what ran beside them (us)\n" % R)
        fh.write("# class   launches   average   median   p10   p90\n")
        for k in ("lone", "paired"):
            v = sorted(groups[k])
            if v:
                fh.write("%-7s %9d %9.1f %8.1f %6.1f %6.1f\n" % (k, len(v), sum(v) / len(v), v[len(v) // 2], v[len(v) // 10], v[9 * len(v) // 10]))
except Exception as e:  # noqa: BLE001
    print("kernel_stats_split skipped:", e)
if os.path.exists(os.path.join(ROOT, "gpurun_out", "r02_sq", "summary.md")):
    shutil.copy(os.path.join(ROOT, "gpurun_out", "r02_sq", "summary.md"), os.path.join(DST, R + "_sq_pmc.md"))
for f in ("bench.json", "bench_8192.json", "bench_8192_open.json", "bench_2048.json", "bench_512.json", "bench_dense.json",
          "bench_inflight1.json", "bench_512_batch8.json", "bench_1024_batch8.json", "sizes.txt", "concurrent.txt",
          "batch.txt",
          "bench_under_rocprof.json", "hbm_calib.txt", "live_rate.txt", "modeB_streaming.txt", "slabs_one_device.txt"):
    p = os.path.join(SRC, f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, R + "_" + f))

# calibration: known 1 GiB per kernel
cal_f = counters(os.path.join(SRC, "calib_fetch", "c_counter_collection.csv"))
cal_w = counters(os.path.join(SRC, "calib_write", "c_counter_collection.csv"))
GiB = float(1 << 30)
cal = {}
for k, v in cal_f.items():
    if "calib_read_dword" in k:
        cal["fetch_dword_read_ratio"] = med(v) * 1024 / GiB
    if "calib_copy_x4" in k:
        cal["fetch_dwordx4_copy_ratio"] = med(v) * 1024 / GiB
for k, v in cal_w.items():
    if "calib_write_dword" in k:
        cal["write_dword_ratio"] = med(v) * 1024 / GiB
fetch_corr = 1.0 / cal["fetch_dword_read_ratio"]
write_corr = 1.0 / cal["write_dword_ratio"]

out = {"calibration": cal, "fetch_correction": fetch_corr, "write_correction": write_corr, "grids": {}}
traffic = {}
lines = ["# HBM-side traffic per kernel launch (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes)", "",
         "Counter units are KiB.  Calibration on this box with known 1 GiB streams in the same access pattern "
         "(tools/hbm_calib.hip): FETCH_SIZE reads %.4f of the true bytes (dword-per-lane reads; %.4f for dwordx4), "
         "WRITE_SIZE %.4f -> corrections x%.3f / x%.3f (MI355X_MICROARCH.md: FETCH_SIZE = 1/2 on gfx950)." % (
             cal["fetch_dword_read_ratio"], cal.get("fetch_dwordx4_copy_ratio", float("nan")),
             cal["write_dword_ratio"], fetch_corr, write_corr), ""]
for tag, fd, wd in (("4096", "pmc_fetch", "pmc_write"), ("8192", "pmc_fetch8k", "pmc_write8k"),
                    ("2048", "pmc_fetch2k", "pmc_write2k")):
    fp = os.path.join(SRC, fd, "f_counter_collection.csv")
    wp = os.path.join(SRC, wd, "w_counter_collection.csv")
    if not (os.path.exists(fp) and os.path.exists(wp)):
        continue
    F, W = counters(fp), counters(wp)
    lines += ["## grid %s^2" % tag, "", "| kernel | launches | FETCH_SIZE KiB (median) | corrected read MB | "
              "WRITE_SIZE KiB (median) | corrected write MB | total MB |", "|---|---|---|---|---|---|---|"]
    g = {}
    for k in sorted(F):
        if "pv_" not in k:
            continue
        f, w = steady(F[k]), steady(W.get(k, [0.0]))
        rb, wb = f * 1024 * fetch_corr, w * 1024 * write_corr
        short = short_name(k)
        g[short] = {"launches_profiled": len(F[k]), "fetch_kib": f, "write_kib": w, "read_bytes": rb,
                    "write_bytes": wb, "bytes_per_launch": rb + wb}
        lines.append("| %s | %d | %.0f | %.1f | %.0f | %.1f | %.1f |" % (short, len(F[k]), f, rb / 1e6, w, wb / 1e6,
                                                                         (rb + wb) / 1e6))
        # (the dominant one: the FreeGrid's windowed run launches a small instantiation of the same kernel)
        if ("pv_step_merged_kernel" in short or ("pv_step_air_kernel" in short and tag not in traffic)) and \
                rb + wb > traffic.get(tag, {}).get("bytes_per_launch", 0.0):
            traffic[tag] = {"bytes_per_launch": rb + wb, "read_bytes": rb, "write_bytes": wb, "kernel": short}
    out["grids"][tag] = g
    lines.append("")
# the analysis half of the metric: bytes of the analysis chain per RUN (sum over its kernels' launches / runs), for the bench's
# grid (from the bench passes above) and for the all-cells-reached workload (tools/gpu_analysis_workload.py)
ANALYSIS = ("pv_far_frame_kernel", "pv_far_cells_kernel", "pv_onset_kernel", "pv_encode_kernel", "pv_encode_groups_kernel",
            "pv_rt60_wave_kernel", "pv_rt60_blocked_kernel", "pv_rt60_groups_kernel", "pv_rt60_tile_kernel", "pv_direction_kernel", "pv_dir_init_kernel",
            "pv_dir_jump_kernel", "pv_dir_final_kernel", "pv_carry_results_kernel", "pv_analysis_fused_kernel", "pv_run_finish_kernel")




def analysis_bytes(fp, wp):
    if not (os.path.exists(fp) and os.path.exists(wp)):
        return None
    F, W = counters(fp), counters(wp)
    runs = max([len(v) for k, v in F.items() if "pv_onset_kernel" in k or "pv_encode_kernel" in k] + [0])
    if not runs:
        return None
    per = {}
    for k in sorted(F):
        if not any(a in k for a in ANALYSIS):
            continue
        short = short_name(k)
        rb = sum(F[k]) * 1024 * fetch_corr / runs
        wb = sum(W.get(k, [0.0])) * 1024 * write_corr / runs
        per[short] = {"launches_per_run": len(F[k]) / runs, "read_bytes_per_run": rb, "write_bytes_per_run": wb}
    tot = sum(v["read_bytes_per_run"] + v["write_bytes_per_run"] for v in per.values())
    return {"runs_profiled": runs, "bytes_per_run": tot, "kernels": per,
            "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections calibrated on the box "
                      "(profiles/%s_analysis_pmc.md)" % R}


alines = ["# HBM-side traffic of the analysis chain per run (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes, corrected as in "
          "%s_hbm_pmc.md)" % R, ""]
for tag, fd, wd, what in (("analysis_4096", "pmc_fetch", "pmc_write", "bench workload: HugeRoom.pv in a 4096^2 grid, T = 435 (a closed room: ~4 400 reached cells)"),
                          ("analysis_512B", "pmc_fetch_ana", "pmc_write_ana", "all-cells-reached workload: Shoebox.pv 25 m at 512^2, T = 3179 (tools/gpu_analysis_workload.py)")):
    ab = analysis_bytes(os.path.join(SRC, fd, "f_counter_collection.csv"), os.path.join(SRC, wd, "w_counter_collection.csv"))
    if ab is None:
        continue
    traffic[tag] = ab
    alines += ["## %s" % what, "", "| kernel | launches per run | read MB per run | written MB per run |", "|---|---|---|---|"]
    for k, v in ab["kernels"].items():
        alines.append("| %s | %.1f | %.2f | %.2f |" % (k, v["launches_per_run"], v["read_bytes_per_run"] / 1e6, v["write_bytes_per_run"] / 1e6))
    alines += ["", "total %.2f MB per run (%d runs profiled)" % (ab["bytes_per_run"] / 1e6, ab["runs_profiled"]), ""]
for f in ("analysis_workload.txt", "trace_ana/a_kernel_stats.csv"):
    pth = os.path.join(SRC, f)
    if os.path.exists(pth):
        alines += ["## %s" % f, "", "```", open(pth).read().strip(), "```", ""]
if len(alines) > 2:
    open(os.path.join(DST, R + "_analysis_pmc.md"), "w").write("\n".join(alines) + "\n")

# stamp: which device code these counters belong to (bench.py quotes them only for the same kernel sources)
sys.path.insert(0, ROOT)
from planeverb_amd.build import kernel_source_hash  # noqa: E402
import subprocess  # noqa: E402
try:
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    head = None
stamp = {"kernel_source_hash": kernel_source_hash(), "collected_at_head": head, "round": R,
         "box_note": "one MI355X box of the pool; boxes differ by up to 10 %"}
# SQ / GRBM counters of the dominant kernel (tools/pmc_r02.sh -> gpurun_out/r02_sq), single launches
sq_dir = os.path.join(ROOT, "gpurun_out", "r02_sq")
sq = {}
for w in ("zero1", "random1"):
    acc = collections.defaultdict(list)
    import glob
    for f in glob.glob(os.path.join(sq_dir, w, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "pv_step_merged_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    durs = []
    for f in glob.glob(os.path.join(sq_dir, w, "p1", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "pv_step_merged_kernel" in r["Kernel_Name"]:
                durs.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if acc and durs:
        m = {k: med(v) for k, v in acc.items()}
        d = med(durs)
        sq[w] = {"counters": m, "launch_ns_under_pmc": d,
                 # GRBM_GUI_ACTIVE is summed over the 8 XCDs
                 "effective_clock_ghz": m.get("GRBM_GUI_ACTIVE", 0) / 8.0 / d,
                 "valu_busy_single_launch": m.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / 1024.0 / (m.get("GRBM_GUI_ACTIVE", 1) / 8.0)}
for t in traffic.values():
    t.update(stamp)
for f in ("presets.txt", "live_pipeline.txt", "rt60.txt", "inflight.txt", "run_times.txt", "resident_trace.txt"):
    pth = os.path.join(SRC, f)
    if os.path.exists(pth):
        shutil.copy(pth, os.path.join(DST, R + "_" + f))
for res in ("275", "750"):
    pth = os.path.join(SRC, "trace_presets_" + res, "p_kernel_stats.csv")
    if os.path.exists(pth):
        shutil.copy(pth, os.path.join(DST, R + "_presets_%s_kernel_stats.csv" % res))
if sq:
    traffic["sq_4096"] = dict(sq, **stamp)
out["stamp"] = stamp
json.dump(out, open(os.path.join(DST, R + "_hbm_pmc.json"), "w"), indent=1)
open(os.path.join(DST, R + "_hbm_pmc.md"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(DST, "hbm_traffic.json"), "w"), indent=1)
print("\n".join(lines))
print(open(os.path.join(DST, R + "_kernel_stats.csv")).read())
