"""Randomised object graphs: the flattened HIP path against the recursive oracle, bit for bit."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from fuzz_scenes import random_camera, random_world

N_SCENES = 48


def _build(pkg, backend, seed, nx, ny):
    rs = np.random.RandomState(seed)
    b = backend.builder()
    world = random_world(pkg, b, rs)
    cam = random_camera(pkg, backend, rs, nx, ny)
    return b, world, cam


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_graphs_flatten_on_host(pkg, seed):
    """CPU-only: every random graph flattens (no GPU needed) and the program is well formed."""
    b, world, _ = _build(pkg, pkg.load(), 1000 + seed, 24, 16)
    words, feat = b.flatten(world)
    ops = words[:, 7] & 0xff
    assert ops[-1] == 0 and (ops[:-1] != 0).all()
    box = ops == 1
    assert (words[box, 6] > np.nonzero(box)[0]).all() and (words[box, 6] < len(words)).all()
    assert int((ops == 4).sum()) == int((ops == 5).sum())          # PUSH / POP balanced
    med = np.nonzero(ops == 6)[0]
    assert np.isin(ops[med + 1], (2, 3)).all()                     # a medium's boundary record follows it


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(N_SCENES))
def test_fuzz_graphs_bit_exact(pkg, gpu, oracle, seed):
    nx, ny, ns = 40, 24, 5
    bg, wg, cam_g = _build(pkg, gpu, 1000 + seed, nx, ny)
    bo, wo, cam_o = _build(pkg, oracle, 1000 + seed, nx, ny)
    assert bytes(cam_g) == bytes(cam_o)
    sg, so = bg.scene(wg), bo.scene(wo)
    img_o, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    img_g, st_g = sg.par_cast(cam_g, nx, ny, ns, stats=True)
    assert_bit_equal(img_g, img_o, "fuzz scene %d" % seed)
    for k in ("aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
        assert st_g[k] == st_o[k], (seed, k, st_g[k], st_o[k])
