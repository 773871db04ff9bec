"""N>1 path on CPU: world_size-2 gloo.  The partition/gather logic is exercised with the oracle as
the per-rank renderer (same pixel seeds as the GPU path), so the assembled frame must be
bit-identical to the single-rank oracle frame."""
import os
import socket
import sys

import numpy as np
import pytest

from common import M, O, REPO, oracle_scene
from minimaloptix_amd import dist as D


def test_tile_partition_covers_every_pixel_once():
    for (w, h) in ((1920, 1080), (100, 37), (8, 8), (1280, 720)):
        for n in (1, 2, 3, 8):
            allp = np.concatenate([D.tile_pixel_indices(w, h, r, n) for r in range(n)])
            assert len(allp) == w * h and len(np.unique(allp)) == w * h
            sizes = [len(D.tile_pixel_indices(w, h, r, n)) for r in range(n)]
            assert max(sizes) - min(sizes) <= 64 * ((w + 7) // 8 // n + 1)


def test_sample_split_seeds_partition():
    seeds = M.launch_seeds(10)
    parts = [D.sample_split_seeds(seeds, r, 4) for r in range(4)]
    assert sorted(np.concatenate(parts).tolist()) == sorted(seeds.tolist())


def _worker(rank, world, port, w, h, spp, q):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hs = M.HostScene("spheres", w, h, farg=0.5)
    sc = oracle_scene(hs)
    seeds = M.launch_seeds(spp)
    acc = np.zeros((h, w, 3), np.float32)
    tiles_x = (w + 7) // 8
    ntiles = tiles_x * ((h + 7) // 8)
    for t in (g * world + (rank + g) % world for g in range((ntiles + world - 1) // world)):      # this rank's tiles only
        if t >= ntiles:
            continue
        x0, y0 = (t % tiles_x) * 8, (t // tiles_x) * 8
        sc.render(seeds, accum=acc, region=(x0, y0, min(x0 + 8, w), min(y0 + 8, h)), threads=1)
    frame = D.gather_tiles(torch.from_numpy(acc), w, h, rank, world, dst=0)
    # sample split: each rank renders the full frame for its launches, then reduce(sum)
    acc2 = np.zeros((h, w, 3), np.float32)
    sc.render(D.sample_split_seeds(seeds, rank, world), accum=acc2, threads=1)
    red = D.reduce_frame(torch.from_numpy(acc2), dst=0)
    if rank == 0:
        q.put((frame.numpy().copy(), red.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,spp", [(2, 72, 40, 4), (3, 70, 37, 3)])       # 3 ranks: ragged tiles, a last group of one tile
def test_tile_split_and_sample_split_gloo(world, w, h, spp):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, spp, q)) for r in range(world)]
    for p in procs:
        p.start()
    frame, red = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    hs = M.HostScene("spheres", w, h, farg=0.5)
    ref, _ = oracle_scene(hs).render(M.launch_seeds(spp))
    assert np.array_equal(frame, ref)                          # tile split: bit-identical to one rank
    assert np.allclose(red, ref, rtol=0, atol=2e-6)            # sample split: equal up to summation order
