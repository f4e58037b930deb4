"""Multi-GPU sharding of par_cast: one process per GPU, pixels (not samples) partitioned.

The per-pixel mean is an ORDERED left fold over samples (lib.rs:365-374), so splitting samples across
GPUs would change the f32 addition order; pixels are independent, so the image is cut into tiles and
rank r renders the tiles with tile_index % world == r (rtg_params.rank/nranks).  Every rank writes its
pixels into a zero-filled full-frame buffer; ONE collective -- reduce(sum) to rank 0 over RCCL/xGMI --
assembles the frame.  x + 0 is exact, so the result is bit-identical to the single-GPU frame.
"""
import torch
import torch.distributed as dist


def reduce_framebuffer(fb, dst=0):
    """fb: full-frame float32 tensor, zero outside this rank's tiles.  In-place reduce to `dst`."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb


def render_sharded(render_shard, nx, ny, rank, world, device):
    """render_shard(fb, rank, world) must fill this rank's tiles of fb ([ny, nx, 3] float32 on `device`).
    Returns the assembled frame on rank 0 (other ranks: their partial frame)."""
    fb = torch.zeros((ny, nx, 3), dtype=torch.float32, device=device)
    render_shard(fb, rank, world)
    return reduce_framebuffer(fb)
