"""Multi-GPU layer: one process per GPU, screen-tile sharding of ONE frame, RCCL all-gather of
the float framebuffer (torch.distributed backend "nccl" on ROCm = RCCL over xGMI; "gloo" on CPU
for tests).  The reference has no multi-GPU path; this layer is new.

Two layouts (gswt_hip.h GSWT_SHARD_*): interleaved 16-px tile ROWS (below), and contiguous tile-COLUMN bands
(`cols_padded`, `shard_of_frame_cols`, `unshard_cols`): rank r owns tile columns [r*b, (r+1)*b), b = ceil(tiles_x / world),
renders an [H, b*16, 4] image and, on the plain surface, only projects the draws that can reach its band.  bench.py uses
the column bands: pairs split evenly across them for a horizon-dominated view and projection shards with them.

Ownership: 16-px tile row `ty` belongs to rank `ty % world`; a rank renders its rows compacted in
row order into a buffer padded to `rows_padded = ceil(tiles_y / world) * 16` rows so every rank
contributes the same byte count (1920x1080, world 8: 9 tile rows = 144 px rows = 4.4 MB per rank).
After the all-gather the full frame is a pure index permutation of the gathered buffer.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

TILE = 16


def rows_padded(height: int, world: int) -> int:
    tiles_y = (height + TILE - 1) // TILE
    return ((tiles_y + world - 1) // world) * TILE


def shard_of_frame(frame: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """The padded shard image rank `rank` would render from a full [H, W, 4] frame (test helper and
    the definition of the layout the HIP compositor writes)."""
    H, W, C = frame.shape
    rp = rows_padded(H, world)
    out = torch.zeros((rp, W, C), dtype=frame.dtype, device=frame.device)
    tiles_y = (H + TILE - 1) // TILE
    for tyl, ty in enumerate(range(rank, tiles_y, world)):
        y0, y1 = ty * TILE, min(H, ty * TILE + TILE)
        out[tyl * TILE: tyl * TILE + (y1 - y0)] = frame[y0:y1]
    return out


def unshard(gathered: torch.Tensor, height: int, world: int) -> torch.Tensor:
    """[world * rows_padded, W, 4] (all-gather order) -> [H, W, 4]."""
    rp = gathered.shape[0] // world
    W, C = gathered.shape[1], gathered.shape[2]
    g = gathered.view(world, rp // TILE, TILE, W, C).permute(1, 0, 2, 3, 4)   # (tile row local, rank) -> global tile row
    return g.reshape(-1, W, C)[:height].contiguous()


def cols_padded(width: int, world: int) -> int:
    tiles_x = (width + TILE - 1) // TILE
    return ((tiles_x + world - 1) // world) * TILE


def shard_of_frame_cols(frame: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """The padded column-band image rank `rank` would render from a full [H, W, 4] frame."""
    H, W, C = frame.shape
    bw = cols_padded(W, world)
    out = torch.zeros((H, bw, C), dtype=frame.dtype, device=frame.device)
    x0, x1 = rank * bw, min(W, (rank + 1) * bw)
    if x1 > x0:
        out[:, : x1 - x0] = frame[:, x0:x1]
    return out


def unshard_cols(gathered: torch.Tensor, width: int, world: int) -> torch.Tensor:
    """[world * H, band_px, 4] (all-gather order) -> [H, W, 4]."""
    H = gathered.shape[0] // world
    bw, C = gathered.shape[1], gathered.shape[2]
    g = gathered.view(world, H, bw, C).permute(1, 0, 2, 3)          # (row, rank, column in band)
    return g.reshape(H, world * bw, C)[:, :width].contiguous()


class FrameGather:
    """All-gather of per-rank shard images into the full frame on every rank."""

    def __init__(self, height: int, width: int, device, group=None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.height, self.width, self.group = height, width, group
        self.rows_padded = rows_padded(height, self.world)
        self.gathered = torch.empty((self.world * self.rows_padded, width, 4), dtype=torch.float32, device=device)

    def __call__(self, shard: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return shard[: self.height]
        dist.all_gather_into_tensor(self.gathered, shard.contiguous(), group=self.group)
        return unshard(self.gathered, self.height, self.world)
