#!/usr/bin/env python3
"""Per-operation VALU costs of the reference's arithmetic, read from the ISA of the PRODUCTION kernels.

    tools/algorithmic_valu.py [--out profiles/r06_final/algorithmic_valu_costs.json] [--asm /tmp/api.s]

`roofline.frac` (rtiow-rust_amd/roofline.py) credits every issued VALU instruction -- services, list bookkeeping, SGPR-spill
v_readlane / v_writelane, schedule ballots.  `roofline.algorithmic_valu` prices only the reference's own work,

    lanes = c_box * N + c_prim * P + c_shade * H + c_cam * samples          (N, P, H: the oracle's counters)

with c_* = the VALU instructions ONE lane needs for one Aabb::hit (aabb.rs:16-27), one primitive test (object.rs:84-111 /
185-218), one shaded hit (hit record + Material::scatter + its random draws, material.rs:55-146) and one camera ray with the
booking of the sample before it (lib.rs:366-374, camera.rs:52-63), counted in the code the compiler emitted for the timed
instantiation of each kernel:
  * the library's translation unit is compiled to assembly with the Makefile's flags + -gline-tables-only (line tables do not
    change code generation), so every instruction carries the source line it was emitted for;
  * a REGION is a span of source lines (found by its anchor text below, so edits above it do not move it); its instructions
    form one or more contiguous clusters in the assembly (a lambda instantiated twice gives two); a cluster's cost = the VALU
    instructions between its first and last instruction, inlined helpers (vector arithmetic, Philox, libm) included;
  * the raw cost of a region = its LARGEST cluster, statically (`valu`).
A pass is SIMT code: its static count is the union of what ANY lane may execute, and it holds launch bookkeeping.  The cost of
ONE lane's reference work (`per_operation` of the shade and camera regions of the lean kernel, which prices every workload) is
built from the same assembly:
    useful  = the VALU instructions of the cluster that were emitted for lines of the region itself or of the arithmetic headers
              (rt_device.h: Vec3, the counter RNG, the rejection loops; rt_trace.h: hit tests, camera, schlick; rt_libm.h) --
              not for the helpers above the kernel (lane ranks, work-item maps, the cost-ordered queue) nor for HIP's headers
            - the code only a Dielectric lane runs (its branch + powf(x, 5): 5 % of book-1's spheres; the modal lane is Lambertian)
            - the Philox blocks as the assembly holds them (inlined copies x 10 rounds, 2 v_mad_u64_u32 each)
            + the blocks one lane GENERATES per event x the instructions of one block:
              scatter: in_unit_sphere accepts with p = pi / 6, 3 draws per try -> 5.73 draws -> 1.43 blocks (vec3.rs:19-26);
              camera:  2 draws (lib.rs:368-369) + 2 per in_unit_disc try (p = pi / 4) + 1 (camera.rs:55) -> 5.55 draws -> 1.39 blocks.
The box step is unrolled twice in every kernel: c_box = region / 2 (exact: no lane-dependent branch inside a step).  c_prim is the
static Sphere::hit region (both roots).  Nothing here needs a GPU.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "rtiow-rust_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function".split()

# kernel (mangled-name prefix, demangled for the report) -> regions: name -> (file, first-line anchor, which occurrence, end anchor or +n lines, divisor)
KERNELS = {
    "render_lean_pool": ("_ZN3rtg16render_lean_poolILb1ELb0ELb1ELb0E", "rtg::render_lean_pool<true, false, true, false>", {
        "box": ("rt_pool.h", "        RT_BOX_STEP();", 0, "        RT_BOX_STEP();  // lanes that left", 2),
        "prim": ("rt_pool.h", "      if (op == OP_SPHERE) {  // Sphere::hit", 0, "      if (COUNT) t_sph += RT_TICK", 1),
        "shade": ("rt_pool.h", "      while (s_count >= 64u", 0, "      // (2b) END pass", 1),
        "cam": ("rt_pool.h", "      while (e_count >= 64u", 0, "      // (3) refill idle lanes", 1),
    }),
    "render_full_pool": ("_ZN3rtg16render_full_poolILi1ELb1ELb0ELb0E", "rtg::render_full_pool<1, true, false, false>", {
        "box": ("rt_full_traverse.inc", "        RT_FULL_BOX_STEP();", 0, "        RT_FULL_BOX_STEP();  // lanes that left", 2),
        "prim": ("rt_full_ops.inc", "    if (op == OP_SPHERE) {  // Sphere::hit", 0, "    } else if (op == OP_RECT) {  // Rect::hit", 1),
        "shade": ("rt_pool_full.h", "      auto shade_pass = [&]", 0, "      auto gen_pass = [&]", 1),
        "cam": ("rt_pool_full.h", "      auto gen_pass = [&]", 0, "      // full passes first", 1),
    }),
    "render_full_pool2": ("_ZN3rtg17render_full_pool2ILb1ELb0E", "rtg::render_full_pool2<true, false>", {
        "box": ("rt_pool2.h", "          P2_BOX_STEP();", 0, "          P2_BOX_STEP();  // lanes that left", 2),
        "prim": ("rt_pool2.h", "        if (r_sph && op == OP_SPHERE) {", 0, "        if (r_pri && op == OP_PRISM) {", 1),
        "shade": ("rt_pool2.h", "      auto shade_pass = [&]", 0, "      auto gen_pass = [&]", 1),
        "cam": ("rt_pool2.h", "      auto gen_pass = [&]", 0, "      // full passes first", 1),
    }),
}


def find_lines(path, begin, end):
    src = open(path).read().split("\n")
    b = [i + 1 for i, l in enumerate(src) if l.startswith(begin)]
    if not b:
        raise SystemExit("anchor %r not found in %s" % (begin, path))
    b = b[0]
    e = [i + 1 for i, l in enumerate(src) if l.startswith(end) and i + 1 >= b]
    if not e:
        raise SystemExit("end anchor %r not found in %s" % (end, path))
    e = e[0]
    return b, (e if "BOX_STEP" in end else e - 1)  # (the two uses of the box-step macro: both lines; else up to the line before the end anchor)


def parse_kernel(lines, prefix):
    start = [i for i, l in enumerate(lines) if l.startswith(prefix)]
    if not start:
        raise SystemExit("kernel %s not in the assembly" % prefix)
    start = start[0]
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    out, cur = [], (None, 0)
    for l in lines[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        t = l.strip()
        if not t or t[0] in ".;" or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith(("v_", "s_", "ds_", "buffer_", "global_", "flat_", "scratch_")):
            out.append((op, cur))
    return out


def region_cost(instrs, fname, a, b):
    idx = [i for i, (_, (f, ln)) in enumerate(instrs) if f == fname and a <= ln <= b]
    if not idx:
        return None
    clusters, first, last = [], idx[0], idx[0]
    for i in idx[1:]:
        if i - last > 400:
            clusters.append((first, last))
            first = i
        last = i
    clusters.append((first, last))
    best = None
    for f, l in clusters:
        span = instrs[f:l + 1]
        valu = sum(1 for op, _ in span if op.startswith("v_"))
        if best is None or valu > best["valu"]:
            best = {"valu": valu, "instructions": len(span), "special_rate": sum(1 for op, _ in span if re.match(r"v_(rcp|sqrt|rsq|mad_u64|mul_hi|mul_lo|div_)", op))}
    best["clusters"] = len(clusters)
    return best


ARITH_FILES = ("rt_device.h", "rt_trace.h", "rt_libm.h")
PHILOX_LINES = ("rt_device.h", 71, 96)   # SampleRng::refill
E_BLOCKS = {"shade": (3.0 / (3.14159265358979 / 6.0)) / 4.0, "cam": (2.0 + 2.0 / (3.14159265358979 / 4.0) + 1.0) / 4.0}


def lane_model(instrs, fname, a, b, name, dielectric):
    """The per-lane cost of the lean kernel's shade / camera region (docstring above)."""
    idx = [i for i, (_, (f, ln)) in enumerate(instrs) if f == fname and a <= ln <= b]
    span = instrs[idx[0]:idx[-1] + 1]
    valu = [(op, loc) for op, loc in span if op.startswith("v_")]
    useful = [(op, (f, ln)) for op, (f, ln) in valu if (f == fname and a <= ln <= b) or f in ARITH_FILES]
    philox = [1 for op, (f, ln) in useful if f == PHILOX_LINES[0] and PHILOX_LINES[1] <= ln <= PHILOX_LINES[2]]
    copies = max(1, round(sum(1 for op, (f, ln) in useful if op.startswith("v_mad_u64_u32")) / 20.0))
    diel = [1 for op, (f, ln) in useful if (f == fname and dielectric and dielectric[0] <= ln <= dielectric[1]) or (dielectric and f == "rt_libm.h")]
    per_block = len(philox) / float(copies)
    cost = len(useful) - len(philox) - len(diel) + E_BLOCKS[name] * per_block
    return {"useful_static": len(useful), "bookkeeping_excluded": len(valu) - len(useful), "philox_static": len(philox), "philox_copies": copies,
            "philox_per_block": per_block, "dielectric_only": len(diel), "expected_blocks_per_lane": E_BLOCKS[name], "per_operation": cost}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--asm", default="", help="reuse an assembly file made with the flags above + -gline-tables-only -S --cuda-device-only")
    args = ap.parse_args()
    asm = args.asm
    if not asm:
        asm = os.path.join(tempfile.mkdtemp(prefix="valu_"), "api.s")
        subprocess.check_call(["hipcc"] + FLAGS + ["-gline-tables-only", "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, "rtg_api.hip")],
                              stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    res = {"method": "static VALU instructions of source regions in the ISA of the timed kernel instantiations (tools/algorithmic_valu.py): a lower bound, loops counted once",
           "kernels": {}}
    for key, (prefix, pretty, regions) in KERNELS.items():
        instrs = parse_kernel(lines, prefix)
        k = {"kernel": pretty, "instructions_total": len(instrs), "valu_total": sum(1 for op, _ in instrs if op.startswith("v_")), "regions": {}}
        for name, (fname, begin, _, end, div) in regions.items():
            a, b = find_lines(os.path.join(CSRC, fname), begin, end)
            c = region_cost(instrs, fname, a, b)
            if c is None:
                raise SystemExit("%s: no instruction for %s:%d-%d" % (pretty, fname, a, b))
            c.update({"file": fname, "lines": [a, b], "per_operation": c["valu"] / float(div)})
            if key == "render_lean_pool" and name in ("shade", "cam"):
                diel = None
                if name == "shade":
                    src = open(os.path.join(CSRC, fname)).read().split("\n")
                    d0 = next(i + 1 for i, l in enumerate(src) if i + 1 >= a and l.startswith("          } else if (kind == MAT_DIELECTRIC) {"))
                    d1 = next(i + 1 for i, l in enumerate(src) if i + 1 > d0 and l.startswith("          } else {  // Isotropic")) - 1
                    diel = (d0, d1)
                c["lane_model"] = lane_model(instrs, fname, a, b, name, diel)
                c["per_operation"] = c["lane_model"]["per_operation"]
            k["regions"][name] = c
            k["c_" + name] = c["per_operation"]
        res["kernels"][key] = k
        print("%-52s c_box %.1f  c_prim %.0f  c_shade %.0f  c_cam %.0f   (of %d VALU / %d instructions)" % (
            pretty, k["c_box"], k["c_prim"], k["c_shade"], k["c_cam"], k["valu_total"], k["instructions_total"]))
        for name, c in k["regions"].items():
            print("    %-6s %s:%d-%d  %d VALU in %d instructions (%d at quarter rate or slower), %d cluster(s)" % (
                name, c["file"], c["lines"][0], c["lines"][1], c["valu"], c["instructions"], c["special_rate"], c["clusters"]))
    import __graft_entry__ as graft
    graft.load_package()
    from rtiow_rust_amd import roofline as rl
    res["build"] = rl.source_stamp(ROOT)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)
        print("wrote", args.out)


if __name__ == "__main__":
    main()
