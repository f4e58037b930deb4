"""The N>1 path on CPU: world_size-2 gloo.  Each rank renders its pixel-tile shard into a zero-filled
full frame, ONE reduce(sum) assembles it on rank 0 -- the same rtiow_rust_amd.parallel code bench.py's
multi-GPU leg uses over RCCL.  (The shard renderer here is the oracle: there is no GPU in this box.)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as graft
from scene_cases import build_case
pkg = graft.load_package(); ora = graft.load_oracle()
from rtiow_rust_amd import parallel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
scene, cam, nx, ny, ns = build_case(pkg, ora, "book1", 64, 48)
def render_shard(fb, rank, world):
    part = scene.par_cast(cam, nx, ny, ns, rank=rank, nranks=world, threads=2)
    fb.copy_(torch.from_numpy(part))
frame = parallel.render_sharded(render_shard, nx, ny, torch.device("cpu"), mode=sys.argv[2])
if rank == 0:
    np.save(sys.argv[1], frame.numpy())
    sf = parallel.ShardedFrame(nx, ny, torch.device("cpu"), mode=sys.argv[2])
    assert sf.bytes_per_rank() == (nx * ny * 12 if sys.argv[2] == "reduce" else max(sf.owned_pixels(r).numel() for r in range(world)) * 12)
    assert sum(sf.owned_pixels(r).numel() for r in range(world)) == nx * ny
dist.barrier(); dist.destroy_process_group()
'''


import pytest


@pytest.mark.parametrize("mode", ["reduce", "gather"])
def test_two_rank_gloo_reduce_equals_single_frame(tmp_path, pkg, oracle, mode):
    """mode "reduce": ONE reduce(sum) of zero-padded full frames; "gather": ONE gather of the ranks' packed tiles (1 / world of the bytes)."""
    from conftest import assert_bit_equal
    from scene_cases import build_case
    out = str(tmp_path / "frame.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    from conftest import retry_on_busy_port

    def run(port):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
        procs = [subprocess.Popen([sys.executable, str(script), out, mode], env=dict(env, RANK=str(r)), stderr=subprocess.PIPE, text=True)
                 for r in range(2)]
        errs = [p.communicate(timeout=300)[1] for p in procs]
        return max(abs(p.returncode) for p in procs), "\n".join(errs)
    rc, err = retry_on_busy_port(run)
    assert rc == 0, err[-2000:]
    scene, cam, nx, ny, ns = build_case(pkg, oracle, "book1", 64, 48)
    assert_bit_equal(np.load(out), scene.par_cast(cam, nx, ny, ns), "2-rank sharded frame, %s" % mode)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """bench.py never reports another N than --gpus asks for (round 2: `bench.py --gpus 8` run as ONE process rendered
    the 1-GPU frame and printed n_gpus: 1).  Checked before any GPU is touched, so it runs here."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # without a launcher it starts its own ranks -- and refuses when the host does not have N GPUs (none here)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
