"""Second multi-GPU axis (BASELINE config 5): one view sharded over the ranks by rows of 16x16 tiles.
CPU part: the partition / halo / strip-exchange bookkeeping (world size 2, gloo).  The rendering side of it -- k strips
reassemble the full render bit for bit, their gradients add up -- is the GPU test test_gpu_tile_rows.py."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from trase_amd.dp import strip_pixel_rows, tile_row_partition


def test_partition_covers_every_tile_row_once():
    for H in (1, 15, 16, 17, 270, 960, 1014, 1080):
        rows = (H + 15) // 16
        for world in (1, 2, 3, 4, 8, 100):
            part = tile_row_partition(H, world)
            assert len(part) == world and part[0][0] == 0 and part[-1][1] == rows
            assert all(part[r][1] == part[r + 1][0] for r in range(world - 1))
            sizes = [e - b for b, e in part]
            assert max(sizes) - min(sizes) <= 1
            # pixel rows incl. the SSIM halo (5 px) stay inside the image and overlap only the neighbours
            for r in range(world):
                y0, y1 = strip_pixel_rows(part, r, H, halo_px=5)
                b0, b1 = strip_pixel_rows(part, r, H)
                assert 0 <= y0 <= b0 <= b1 <= y1 <= H
                assert b0 - y0 <= 5 and y1 - b1 <= 5


def test_load_balanced_partition_minimises_the_largest_strip():
    import itertools
    import random
    rnd = random.Random(3)
    for H, world in ((960, 2), (960, 4), (960, 8), (270, 3), (1080, 8), (40, 8)):
        rows = (H + 15) // 16
        loads = [int(1000 * rnd.random() ** 3 * (1 + (i > rows // 2))) for i in range(rows)]
        part = tile_row_partition(H, world, loads=loads)
        assert len(part) == world and part[0][0] == 0 and part[-1][1] == rows
        assert all(part[r][1] == part[r + 1][0] for r in range(world - 1)) and all(b <= e for b, e in part)
        mean = sum(loads) / rows
        cost = lambda p: max(sum(loads[b:e]) + 0.05 * mean * (e - b) for b, e in p)
        assert cost(part) <= cost(tile_row_partition(H, world)) + 1e-6      # never worse than the equal-rows split
        if rows <= 12 and world <= 4:                                          # small enough to enumerate every contiguous split
            k = min(world, rows)
            best = min(cost(list(zip((0,) + c, c + (rows,)))) for c in itertools.combinations(range(1, rows), k - 1))
            assert abs(cost(part[:k]) - best) < 1e-6
    # balanced strips have different heights: the exchange pads to the tallest
    part = tile_row_partition(75, 2, loads=[100, 1, 1, 1, 1])
    assert part == [(0, 1), (1, 5)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out, balanced=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trase_amd.dp import allgather_strips
    H, W, C = 75, 40, 3                                   # 5 tile rows (the last one ragged): ranks get 3 and 2
    part = tile_row_partition(H, world) if not balanced else tile_row_partition(H, world, loads=[100, 1, 1, 1, 1])   # 1 and 4 rows
    y0, y1 = strip_pixel_rows(part, rank, H)
    full_want = (torch.arange(H, dtype=torch.float32)[None, :, None] * 10 + torch.arange(W)[None, None, :]
                 + 1000 * torch.arange(C)[:, None, None])
    local = torch.full((C, H, W), 777.0)                  # rows outside the own strip hold junk: they must not leak
    local[:, y0:y1] = full_want[:, y0:y1]
    local.requires_grad_(True)
    full = allgather_strips(local, part, H)
    ok = torch.equal(full, full_want)
    weight = torch.randn(C, H, W, generator=torch.Generator().manual_seed(5))      # same on every rank: "the full-frame loss"
    (full * weight).sum().backward()
    g = local.grad
    ok = ok and torch.equal(g[:, y0:y1], weight[:, y0:y1]) and float(g[:, :y0].abs().sum()) == 0 and float(g[:, y1:].abs().sum()) == 0
    # halo rows of the neighbour are available in the gathered map for a strip-local 11x11 SSIM
    h0, h1 = strip_pixel_rows(part, rank, H, halo_px=5)
    ok = ok and torch.equal(full[:, h0:h1], full_want[:, h0:h1]) and (h1 - h0) > (y1 - y0)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_strip_exchange_world_size_2():
    world = 2
    mgr = mp.Manager()
    for balanced in (False, True):            # equal-rows strips, then load-balanced strips of unequal height
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out, balanced), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}
