#!/usr/bin/env python3
"""Print the flat program of a test scene (host only, no GPU): op histogram and the list-level record sequence.
usage: tools/dump_program.py [case]"""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import scene_cases
pkg = g.load_package(); be = pkg.load()
name = sys.argv[1] if len(sys.argv) > 1 else "book2"
fn, nx, ny, ns = scene_cases.CASES[name]
b = be.builder()
world, cam, _ = fn(pkg, b, nx, ny)
words, feat = b.flatten(world)
OPS = ["END", "BOX", "SPHERE", "RECT", "PUSH", "POP", "MEDIUM", "PRISM", "BEND", "SEG", "SAVE", "MERGE", "EXT"]
ops = words[:, 7] & 0xff
print("%s: %d records, features 0x%x" % (name, len(words), feat), dict(collections.Counter(OPS[o] for o in ops)))
pc, depth = 0, 0
while pc < len(words):  # walk the list level: skip over Bvh subtrees
    w = words[pc]; op = int(w[7] & 0xff); fl = int(w[7]) >> 8
    extra = ""
    if op == 1:
        extra = "skip -> %d (subtree of %d records)" % (w[6], w[6] - pc - 1)
    if op == 6:
        extra = "end -> %d" % w[4]
    if op == 9:
        extra = "hoisted segment: records %d .. %d" % (pc + 1, w[6] - 1)
    print("%5d %-7s flags %06x %s %s" % (pc, OPS[op], fl, "GATHER" if (int(w[7]) & (1 << 15)) else "", extra))
    if op == 0:
        break
    pc = int(w[6]) if op == 1 else (int(w[4]) if op == 6 else pc + 1)
