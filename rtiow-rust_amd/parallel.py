"""Multi-GPU sharding of par_cast: one process per GPU, pixels (not samples) partitioned.

The per-pixel mean is an ORDERED left fold over samples (lib.rs:365-374), so splitting samples across
GPUs would change the f32 addition order; pixels are independent, so the image is cut into tiles (shard_tile) and
rank r renders the tiles with tile_index % world == r (rtg_params.rank/nranks).  Every rank writes its
pixels into a zero-filled full-frame buffer; ONE collective -- reduce(sum) to rank 0 over RCCL/xGMI --
assembles the frame.  x + 0 is exact, so the result is bit-identical to the single-GPU frame.

`ShardedFrame` is the ONE implementation of that: bench.py's N > 1 leg drives it over RCCL ("nccl"), the CPU
test (tests/test_dist_cpu.py) over gloo with the oracle as the shard renderer.  The single-process twin inside the
library is rtg_par_cast_multi (include/rtiow_gpu.h).
"""
import torch
import torch.distributed as dist


def shard_tile(world):
    """(tile_w, tile_h) of the interleave for `world` ranks: 16x16 up to 4 ranks, 8x8 from 8 ranks on.  With an eighth of the
    tiles per rank the slowest rank of the 16x16 interleave lies 5-6 % above the mean (every shard of C3's and of a C5-like
    frame timed alone: profiles/r04_experiments/r04x_shard_tiles.txt); 8x8 tiles bring book-1's to 1.6-2 % (slowest shard 10.57
    -> 10.18 ms) and book-2's to 4 % (29.2 -> 28.9); at 2 and 4 ranks the tile size changes nothing."""
    return (8, 8) if world >= 8 else (16, 16)


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class ShardedFrame:
    """A full-frame float3 buffer on `device` that this rank fills with ITS tiles and rank 0 receives whole.

    render_shard(fb, rank, world) must write this rank's pixels of fb ([ny, nx, 3] float32) and leave the others
    alone; `via_host` reduces through host memory (gloo cannot reduce device tensors of a GPU it shares with
    another rank: the one-GPU test hook of bench.py)."""

    def __init__(self, nx, ny, device, via_host=False):
        self.rank, self.world = world_info()
        self.tile = shard_tile(self.world)   # pass as rtg_params.tile_w / tile_h
        self.fb = torch.zeros((ny, nx, 3), dtype=torch.float32, device=device)
        self.via_host = via_host

    def render(self, render_shard, dst=0):
        """One frame: zero (other ranks' pixels must contribute +0), render this rank's tiles, ONE reduce(sum)."""
        if self.world > 1:
            self.fb.zero_()
        out = render_shard(self.fb, self.rank, self.world)
        self.reduce(dst)
        return out

    def reduce(self, dst=0):
        if self.world <= 1:
            return self.fb
        if self.via_host:
            host = self.fb.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            self.fb.copy_(host)
        else:
            dist.reduce(self.fb, dst=dst, op=dist.ReduceOp.SUM)   # the float3 framebuffer over RCCL / xGMI
        return self.fb


def render_sharded(render_shard, nx, ny, device, via_host=False):
    """Convenience: one sharded frame; returns the assembled frame on rank 0 (other ranks: their partial sums)."""
    frame = ShardedFrame(nx, ny, device, via_host=via_host)
    frame.render(render_shard)
    return frame.fb
