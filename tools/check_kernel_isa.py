#!/usr/bin/env python3
"""Guard of the dominant kernel's compiled form (CPU only: hipcc -S for gfx950, ~1.5 min).

pv_step_merged_kernel holds the air arm (one wave per tile, 180 registers of fields) and the general arm in ONE function.  Until
round 5 the air arm kept ~100 scalar row offsets alive through the kernel, more than the SGPR file holds, so some were parked in
VGPR lanes (v_writelane / v_readlane) -- WHICH ones, and how the 180 tile loads are scheduled against their first consumers,
changed with anything else in the kernel, down to the layout of its arguments (a never-read StepArgs::layoutPad steered it).
Since round 5 no row offset is alive during the steps (PV_ROWOFF2 in pv_kernels.hip) and the pad is gone; the guard stays, for
the two compiled forms that were measured slow on MI355X in round 3 (profiles/r03_palette_ab.txt):
  * parked offsets reloaded INSIDE the twelve steps (237 v_readlane within the packed arithmetic instead of 0-8): 3-4 % slower
    at 4096^2, 2-3 % at 2048^2;
  * the tile loads issued in groups with full waits (s_waitcnt vmcnt(0)) between them instead of all 180 before the first
    consumer: 18-23 % slower at 4096^2 / 8192^2.
This script compiles pv_kernels.hip with the Makefile's flags and checks the large-grid instantiations for both.
tests/test_host_cpu.py runs it, so that an unrelated edit cannot cost the headline silently.  The thresholds were taken on the
compiler named in THRESHOLDS_TAKEN_ON.  Both checks are STRUCTURAL -- "no full wait inside the tile-load phase" holds or does not,
and the measured-good / measured-bad counts of reloads inside the steps are 0-8 against 237 -- so another hipcc is judged by the
same bounds (round 6; round 5 printed "not judged" there and left the two slow forms unguarded on every other toolchain): the
verdict then says which compiler the bounds come from, and a violation is a reason to MEASURE, as it is on the known compiler
(the test only warns unless PV_ISA_GUARD_STRICT=1).

    python tools/check_kernel_isa.py [file.s]      (exit code 1 on a violation)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "planeverb_amd", "csrc")
THRESHOLDS_TAKEN_ON = "HIP version: 7.2.26015"  # hipcc --version, first line (prefix)
FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-fvisibility=hidden"]
# kernel -> (max v_readlane inside the packed-arithmetic span, as measured good: 0-8 / 13-86)
KERNELS = {
    "pv_step_merged_kernel<12, 36, 2, 9> (4096^2, 8192^2)": ("_ZN3pva21pv_step_merged_kernelILi12ELi36ELi2ELi9ELb0EEEvNS_8StepArgsE", 16),
    "pv_step_merged_kernel<12, 36, 2, 9, packed general arm> (scenes with >= 8 % wall tiles)": (
        "_ZN3pva21pv_step_merged_kernelILi12ELi36ELi2ELi9ELb1EEEvNS_8StepArgsE", 16),
    "pv_step_merged_kernel<10, 36, 2, 9> (2048^2)": ("_ZN3pva21pv_step_merged_kernelILi10ELi36ELi2ELi9ELb0EEEvNS_8StepArgsE", 100),
}


def kernel_instructions(asm, symbol):
    i = asm.index(symbol + ":")
    j = asm.index("\t.section\t.rodata", i)
    return [l for l in asm[i:j].splitlines() if l.startswith("\t") and l.split() and not l.strip().startswith((".", ";"))]


def check(asm):
    ok = True
    for name, (sym, max_reload) in KERNELS.items():
        ins = kernel_instructions(asm, sym)
        op = [l.split()[0] for l in ins]
        pk = [i for i, o in enumerate(op) if o == "v_pk_add_f32"]
        reload_in_steps = sum(1 for i, o in enumerate(op) if o == "v_readlane_b32" and pk[0] <= i <= pk[-1])
        loads = [i for i, o in enumerate(op[:pk[0]]) if o.startswith("buffer_load_dword")]
        lo = loads[-170] if len(loads) >= 170 else loads[0]
        full_waits = 0
        for l in ins[lo:loads[-1]]:
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if m and int(m.group(1)) <= 3:
                full_waits += 1
        c = collections.Counter(op)
        good = reload_in_steps <= max_reload and full_waits == 0
        ok = ok and good
        print("%s: %d instructions, %d v_readlane / %d v_writelane, %d of the v_readlane inside the steps (<= %d), "
              "%d full waits inside the tile-load phase (0): %s" % (
                  name, len(ins), c["v_readlane_b32"], c["v_writelane_b32"], reload_in_steps, max_reload, full_waits,
                  "ok" if good else "REGRESSION (see this file's header)"))
    return ok


def compiler_version():
    try:
        return subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0].strip()
    except Exception:  # noqa: BLE001
        return "unknown"


def main():
    ver = compiler_version()
    judged = ver.startswith(THRESHOLDS_TAKEN_ON)
    print("compiler: %s (thresholds taken on '%s...': %s)" % (ver, THRESHOLDS_TAKEN_ON,
                                                              "judged" if judged else "judged by the same structural bounds; another register allocator -- measure before trusting either verdict"))
    if len(sys.argv) > 1:
        asm = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "pv_kernels.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", out, "pv_kernels.hip"],
                                  cwd=CSRC, stderr=subprocess.DEVNULL)
            asm = open(out).read()
    ok = check(asm)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
