#!/usr/bin/env python3
"""Lint for the one software-managed pipeline hazard the inline-asm DPP subtract could hit on gfx9-family ISAs
(gfx950 included): a VGPR written by a VALU instruction must not be read by a DPP instruction within the next 2 wait
states.  The compiler inserts s_nop for DPP instructions it selects itself, but inline asm is opaque to its hazard
recogniser -- so this script scans the generated assembly: every `v_subrev_f32_dpp` (subLanePrev in pv_kernels.hip)
must have no writer of its source register among the 2 preceding instructions.  Run by tests/test_host_cpu.py.

usage: check_dpp_hazard.py file.s   (exit 1 and a listing when a hazard is found)"""
import re
import sys


def regs(tok):
    """register numbers named by an operand token such as v12 or v[12:13]"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main(path):
    insts = []  # (line number, mnemonic, [operands])
    for n, line in enumerate(open(path), 1):
        line = line.split(";")[0].strip()
        if not line or line.startswith(".") or line.endswith(":") or line.startswith("//"):
            continue
        parts = line.split(None, 1)
        ops = [o.strip() for o in re.split(r",\s*", parts[1])] if len(parts) > 1 else []
        insts.append((n, parts[0], ops))
    bad = 0
    checked = 0
    for i, (n, mn, ops) in enumerate(insts):
        if mn != "v_subrev_f32_dpp":
            continue
        checked += 1
        src = regs(ops[1].split()[0]) | regs(ops[2].split()[0])
        wait = 0
        for j in range(i - 1, max(i - 8, -1), -1):
            pn, pm, pops = insts[j]
            if wait >= 2:
                break
            if pm.startswith("v_") and pops and regs(pops[0].split()[0]) & src:
                print("%s:%d: %s reads a register written %d wait state(s) earlier at line %d (%s)" % (
                    path, n, mn, wait, pn, pm))
                bad += 1
                break
            m = re.fullmatch(r"s_nop", pm)
            wait += (int(pops[0], 0) + 1) if m else 1
    print("%s: %d v_subrev_f32_dpp checked, %d hazard(s)" % (path, checked, bad))
    return 1 if bad or not checked else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
