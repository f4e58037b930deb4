#!/usr/bin/env python3
"""Where the production kernels touch SCRATCH (spilled VGPRs, by-value kernel arguments kept on the stack), by source line.

    tools/hot_spills.py [--asm /tmp/api.s] [kernel-name-substring ...]

`make resource-usage` says HOW MANY registers a kernel spills, not WHERE.  This compiles the library's translation unit to assembly with
the Makefile's flags + -gline-tables-only (line tables do not change code generation; tools/algorithmic_valu.py's recipe), and lists
every scratch_load / scratch_store of the timed instantiations with the source line it was emitted for and its position in the kernel
(instructions from the entry: the prologue's stores sit in the first few hundred).  Round 6 found the pool-2 kernel's slot id this way:
one dword, stored by every refill and re-read by every finish -- 3 % of C4 (HISTORY.md).  Nothing here needs a GPU.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rtiow-rust_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function".split()
# the instantiations the bench configs run (mangled prefixes)
KERNELS = {
    "render_lean_pool<true, false, true, false>": "_ZN3rtg16render_lean_poolILb1ELb0ELb1ELb0E",
    "render_full_pool<1, true, false, false>": "_ZN3rtg16render_full_poolILi1ELb1ELb0ELb0E",
    "render_full_pool2<true, false>": "_ZN3rtg17render_full_pool2ILb1ELb0E",
    "render_full_sync<1, false, false, false>": "_ZN3rtg16render_full_syncILi1ELb0ELb0ELb0E",
}


def main():
    args = sys.argv[1:]
    asm = None
    if "--asm" in args:
        asm = args[args.index("--asm") + 1]
        args = [a for a in args if a not in ("--asm", asm)]
    if not asm or not os.path.exists(asm):
        asm = asm or os.path.join(tempfile.gettempdir(), "rtg_api_lines.s")
        subprocess.run(["hipcc"] + FLAGS + ["-gline-tables-only", "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, "rtg_api.hip")],
                       check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    for name, prefix in KERNELS.items():
        if args and not any(a in name for a in args):
            continue
        starts = [i for i, l in enumerate(lines) if l.startswith(prefix)]
        if not starts:
            print("%s: not in the assembly" % name)
            continue
        start = starts[0]
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
        cur, pos, n_valu, rows = ("?", 0), 0, 0, []
        for i in range(start, end):
            t = lines[i].strip()
            m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
            if m:
                cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
                continue
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                continue
            pos += 1
            n_valu += t.startswith("v_")
            if t.startswith("scratch_"):
                rows.append((pos, cur, t))
        lane_spill = sum(1 for i in range(start, end) if re.match(r"\s*v_(read|write)lane_b32", lines[i]))
        print("%s: %d instructions (%d VALU), %d scratch instructions, %d v_readlane / v_writelane (SGPR spill code included)" % (
            name, pos, n_valu, len(rows), lane_spill))
        for p, c, t in rows:
            print("   @%-6d %-22s %s" % (p, "%s:%d" % c, t))


if __name__ == "__main__":
    main()
