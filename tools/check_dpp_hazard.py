#!/usr/bin/env python3
"""Lint for the one software-managed pipeline hazard the inline-asm DPP subtract could hit on gfx9-family ISAs
(gfx950 included): a VGPR written by a VALU instruction must not be read by a DPP instruction within the next 2 wait
states.  The compiler inserts s_nop for DPP instructions it selects itself, but inline asm is opaque to its hazard
recogniser -- so this script scans the generated assembly: every `v_subrev_f32_dpp` (subLanePrev in pv_kernels.hip)
must have no writer of its source register among the 2 preceding wait states ALONG EVERY PATH into it: at a label the walk goes on
through the instruction in front of it and through every branch that names it (the first version was a linear scan that skipped
labels).  Run by the Makefile on the assembly of the very flags the objects are built with (the build
fails on a hazard) and by tests/test_host_cpu.py.

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
    insts = []  # (line number, mnemonic, [operands]); labels are ("<label>", [name])
    for n, line in enumerate(open(path), 1):
        line = line.split(";")[0].strip()
        m = re.match(r"^(\.LBB\d+_\d+|[A-Za-z_][\w$.]*):$", line)
        if m:
            insts.append((n, "<label>", [m.group(1)]))
            continue
        if not line or line.startswith(".") or line.endswith(":") or line.startswith("//"):
            continue
        parts = line.split(None, 1)
        ops = [o.strip() for o in re.split(r",\s*", parts[1])] if len(parts) > 1 else []
        insts.append((n, parts[0], ops))
    # branches by target: what can run right before a label is the instruction in front of it (unless that is an unconditional
    # branch) and every branch that names it
    jumps = {}
    for i, (n, mn, ops) in enumerate(insts):
        if mn.startswith(("s_cbranch", "s_branch")) and ops:
            jumps.setdefault(ops[0], []).append(i)

    def hazard(i, src, wait, depth):
        """walk backwards from instruction i (exclusive) along EVERY path until 2 wait states have passed; a description of the
        first hazard found, or None"""
        j = i - 1
        while j >= 0 and wait < 2:
            pn, pm, pops = insts[j]
            if pm == "<label>":
                if depth > 8:
                    return "line %d: more than 8 labels deep" % pn
                for b in jumps.get(pops[0], []):  # arrive by a branch: the branch instruction itself is one wait state
                    h = hazard(b, src, wait + 1, depth + 1)
                    if h:
                        return h
                if pops[0].startswith("_Z") or not pops[0].startswith(".LBB"):
                    return None  # a kernel's entry: nothing of this wave runs before it
                j -= 1  # fall through from above (checked next; an unconditional branch there ends the path)
                continue
            if pm in ("s_branch", "s_setpc_b64", "s_endpgm"):
                return None  # nothing falls through an unconditional branch
            if pm.startswith("v_") and pops and regs(pops[0].split()[0]) & src:
                return "written %d wait state(s) earlier at line %d (%s)" % (wait, pn, pm)
            wait += (int(pops[0], 0) + 1) if pm == "s_nop" else 1
            j -= 1
        return None

    bad = 0
    checked = 0
    for i, (n, mn, ops) in enumerate(insts):
        if mn != "v_subrev_f32_dpp":
            continue
        checked += 1
        src = regs(ops[1].split()[0]) | regs(ops[2].split()[0])
        h = hazard(i, src, 0, 0)
        if h:
            print("%s:%d: %s reads a register %s" % (path, n, mn, h))
            bad += 1
    print("%s: %d v_subrev_f32_dpp checked along every path into them, %d hazard(s)" % (path, checked, bad))
    return 1 if bad or not checked else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
