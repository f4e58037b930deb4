#!/usr/bin/env python3
"""Kernel resource table (VGPRs, spills, scratch, LDS, occupancy) of the HIP library: parses
`make -C rtiow-rust_amd/csrc resource-usage` (clang's -Rpass-analysis=kernel-resource-usage remarks).
usage: tools/resource_usage.py [substring-of-kernel-name ...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-s", "-C", os.path.join(root, "rtiow-rust_amd", "csrc"), "resource-usage"] +
                     (["EXTRA=" + os.environ["EXTRA"]] if os.environ.get("EXTRA") else []),
                     capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass-analysis", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        name = t.split(":", 1)[1].strip()
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            pass
        cur = {"name": re.sub(r"\(.*", "", name).replace("void rtg::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
pats = sys.argv[1:]
print("%-58s %5s %5s %6s %6s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "vspill", "sspill", "scratch", "occ", "LDS"))
for r in rows:
    if pats and not any(p in r["name"] for p in pats):
        continue
    print("%-58s %5s %5s %6s %6s %7s %4s %7s" % (r["name"][:58], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("VGPRs Spill", "?"),
                                                 r.get("SGPRs Spill", "?"), r.get("ScratchSize [bytes/lane]", "?"),
                                                 r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))
