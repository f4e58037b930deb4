"""Multi-GPU sharding of par_cast: one process per GPU, pixels (not samples) partitioned.

The per-pixel mean is an ORDERED left fold over samples (lib.rs:365-374), so splitting samples across
GPUs would change the f32 addition order; pixels are independent, so the image is cut into tiles (shard_tile) and
rank r renders the tiles with tile_index % world == r (rtg_params.rank/nranks).  Every rank writes its
pixels into a zero-filled full-frame buffer; ONE collective -- reduce(sum) to rank 0 over RCCL/xGMI --
assembles the frame.  x + 0 is exact, so the result is bit-identical to the single-GPU frame.

A second, PACKED collective ships only what a rank owns (mode="gather"): every rank packs its pixels -- 1 / world of the frame --
into a dense buffer, ONE gather brings the buffers to rank 0, which scatters them into the frame.  Pure copies: bit-identical
by construction, and 1 / world of the bytes per rank (rank 0 receives one frame instead of `world` frames).  The default
stays the reduce north_star names; bench.py prints which one ran and the bytes per rank.

`ShardedFrame` is the ONE implementation of both: bench.py's N > 1 leg drives it over RCCL ("nccl"), the CPU
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

    def __init__(self, nx, ny, device, via_host=False, mode="reduce"):
        if mode not in ("reduce", "gather"):
            raise ValueError("ShardedFrame mode: 'reduce' or 'gather'")
        self.rank, self.world = world_info()
        self.tile = shard_tile(self.world)   # pass as rtg_params.tile_w / tile_h
        self.fb = torch.zeros((ny, nx, 3), dtype=torch.float32, device=device)
        self.via_host = via_host
        self.mode = mode
        self.nx, self.ny = nx, ny
        self._owned = None   # mode "gather": per rank, the flat pixel indices it owns (row-major), built on first use
        self._packed = self._parts = None   # ... and the buffers of the collective, allocated once

    def owned_pixels(self, rank):
        """Flat row-major indices of the pixels of tiles with tile_index % world == rank (rtg_params.rank / nranks / tile_w / tile_h)."""
        if self._owned is None:
            tw, th = self.tile
            tiles_x = (self.nx + tw - 1) // tw
            rows = torch.arange(self.ny, device=self.fb.device).unsqueeze(1) // th
            cols = torch.arange(self.nx, device=self.fb.device).unsqueeze(0) // tw
            owner = ((rows * tiles_x + cols) % self.world).reshape(-1)
            self._owned = [torch.nonzero(owner == r).reshape(-1) for r in range(self.world)]
        return self._owned[rank]

    def bytes_per_rank(self):
        """What one rank hands to the collective per frame."""
        if self.world <= 1:
            return 0
        if self.mode == "gather":
            return int(max(self.owned_pixels(r).numel() for r in range(self.world))) * 12
        return self.nx * self.ny * 12

    def render(self, render_shard, dst=0):
        """One frame: zero (other ranks' pixels must contribute +0), render this rank's tiles, ONE reduce(sum)."""
        if self.world > 1 and self.mode == "reduce":
            self.fb.zero_()
        out = render_shard(self.fb, self.rank, self.world)
        self.reduce(dst)
        return out

    def reduce(self, dst=0):
        if self.world <= 1:
            return self.fb
        if self.mode == "gather":
            return self._gather(dst)
        if self.via_host:
            host = self.fb.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            self.fb.copy_(host)
        else:
            dist.reduce(self.fb, dst=dst, op=dist.ReduceOp.SUM)   # the float3 framebuffer over RCCL / xGMI
        return self.fb

    def _gather(self, dst):
        """ONE gather of the ranks' packed pixels to `dst` (equal-sized buffers: the largest share, zero-padded), then a scatter
        into the frame there.  Copies only."""
        n = int(max(self.owned_pixels(r).numel() for r in range(self.world)))
        flat = self.fb.view(-1, 3)
        mine = self.owned_pixels(self.rank)
        if self._packed is None:
            where = torch.device("cpu") if self.via_host else self.fb.device
            self._packed = torch.zeros((n, 3), dtype=torch.float32, device=where)
            self._parts = [torch.zeros((n, 3), dtype=torch.float32, device=where) for _ in range(self.world)] if self.rank == dst else None
        packed, parts = self._packed, self._parts
        packed[:mine.numel()] = flat[mine].to(packed.device)
        dist.gather(packed, parts, dst=dst)   # 1 / world of the float3 framebuffer per rank over RCCL / xGMI
        if self.rank == dst:
            for r in range(self.world):
                idx = self.owned_pixels(r)
                flat[idx] = parts[r][:idx.numel()].to(self.fb.device)
        return self.fb


def render_sharded(render_shard, nx, ny, device, via_host=False, mode="reduce"):
    """Convenience: one sharded frame; returns the assembled frame on rank 0 (other ranks: their partial sums / own tiles)."""
    frame = ShardedFrame(nx, ny, device, via_host=via_host, mode=mode)
    frame.render(render_shard)
    return frame.fb
