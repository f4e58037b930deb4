#!/usr/bin/env python3
"""Mint tests/golden/ref_rttnw_final_200.npz from a data file the REFERENCE publishes: img/rttnw-final.jpg, the book-2 final scene
its README shows (README.md:16-20: rendered by the Rust binary, scene built by `book_final_scene` with
`SmallRng::seed_from_u64(0xDEADBEEF)`, src/main.rs:333).  The 1000x1000 JPEG is box-filtered to 200x200 RGB u8 -- an output of the
reference itself, the only one this repository can hold a render against (no Rust toolchain).  Run in the build container
(needs /root/reference and PIL):  python tools/gen_ref_image_fixture.py"""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = "/root/reference/img/rttnw-final.jpg"
img = Image.open(src).convert("RGB")
small = np.asarray(img.resize((200, 200), Image.BOX), dtype=np.uint8)
out = os.path.join(ROOT, "tests", "golden", "ref_rttnw_final_200.npz")
np.savez_compressed(out, rgb=small, source=np.array("cbiffle/rtiow-rust img/rttnw-final.jpg, %dx%d, box-filtered to 200x200" % img.size))
print("wrote", out, os.path.getsize(out), "bytes")

# The second picture the reference publishes: img/demo-scene.jpg, its book-1 random-spheres scene (README.md:13-14, 1200x800x50,
# rendered by an older revision that still had a gradient sky; the scene code survives commented out in src/lib.rs:236-319).  The
# positions, materials and colours of its ~480 small spheres are a function of the construction RNG -- SmallRng::seed_from_u64(
# 0xDEADBEEF) -- and of the ORDER random_scene draws from it.  Box-filtered to 300x200.
src2 = "/root/reference/img/demo-scene.jpg"
img2 = Image.open(src2).convert("RGB")
small2 = np.asarray(img2.resize((300, 200), Image.BOX), dtype=np.uint8)
out2 = os.path.join(ROOT, "tests", "golden", "ref_demo_scene_300x200.npz")
np.savez_compressed(out2, rgb=small2, source=np.array("cbiffle/rtiow-rust img/demo-scene.jpg, %dx%d, box-filtered to 300x200" % img2.size))
print("wrote", out2, os.path.getsize(out2), "bytes")
