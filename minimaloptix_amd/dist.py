"""Multi-GPU plumbing (new relative to the reference): one process per GPU,
torch.distributed over RCCL ("nccl") or gloo.  Nothing here is on the per-ray path.

  tile split   : every rank renders one 8x8-pixel tile of every group of N tiles (raster order), tile
                 g * N + (rank + g) % N of group g (moptix_set_partition); per-pixel seeds depend only on the global pixel index
                 and the launch seed, so the gathered frame is bit-identical to a 1-GPU frame.
                 One exchange at the end: gather of the packed tiles to rank 0.
  sample split : every rank renders the whole frame for launches i with i % N == rank; one
                 reduce(sum) of the accumulators (equal to 1-GPU up to float summation order).
"""
import numpy as np

# Sample split reorders the float32 additions of a pixel's samples (per-rank partial sums, then the reduce);
# per-sample values are clamped to [0, 1], so the normalised image moves by a few ulps of 1: |delta| <= this bound.
SAMPLE_SPLIT_TOL = 2e-6


def tile_pixel_indices(width, height, rank, nranks):
    """Global pixel ids (y*W+x, row 0 = bottom) owned by `rank`, in work-item order."""
    tiles_x, tiles_y = (width + 7) // 8, (height + 7) // 8
    g = np.arange((tiles_x * tiles_y + nranks - 1) // nranks, dtype=np.int64)     # groups of nranks tiles in raster order
    t = g * nranks + (rank + g) % nranks                    # the deal rotates from group to group (megakernel.h item_to_pixel)
    t = t[t < tiles_x * tiles_y]
    tx, ty = t % tiles_x, t // tiles_x
    inn = np.arange(64, dtype=np.int64)
    x = (tx[:, None] * 8 + (inn & 7)[None, :]).reshape(-1)
    y = (ty[:, None] * 8 + (inn >> 3)[None, :]).reshape(-1)
    ok = (x < width) & (y < height)
    return (y * width + x)[ok]


def max_tile_pixels(width, height, nranks):
    return max(len(tile_pixel_indices(width, height, r, nranks)) for r in range(nranks))


_tile_cache = {}


def _tile_plan(width, height, rank, nranks, dst, device):
    """Index tensors and staging buffers of one (frame size, world) pair, built once: the gather sits inside the
    timed frame, so nothing in it may depend on host work that scales with the pixel count."""
    import torch
    key = (width, height, rank, nranks, dst, str(device))
    plan = _tile_cache.get(key)
    if plan is None:
        counts = [len(tile_pixel_indices(width, height, r, nranks)) for r in range(nranks)]
        nmax = max(counts)
        mine = torch.from_numpy(tile_pixel_indices(width, height, rank, nranks)).to(device)
        plan = dict(nmax=nmax, mine=mine, send=torch.zeros((nmax, 3), dtype=torch.float32, device=device))
        if rank == dst:
            plan["idx"] = [torch.from_numpy(tile_pixel_indices(width, height, r, nranks)).to(device) for r in range(nranks)]
            plan["recv"] = [torch.empty((nmax, 3), dtype=torch.float32, device=device) for _ in range(nranks)]
        _tile_cache[key] = plan
    return plan


def comm_init(ctx, rank, nranks, device):
    """RCCL communicator of a minimaloptix_amd.Context (C ABI moptix_comm_init): rank 0 makes the ncclUniqueId, the
    process group that is already up (torch.distributed) carries its 128 bytes to the other ranks."""
    import torch
    import torch.distributed as dist
    buf = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        buf.copy_(torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8))
    if nranks > 1:
        dist.broadcast(buf, src=0)
    ctx.comm_init(bytes(buf.cpu().numpy().tobytes()), rank, nranks)


def gather_tiles(accum, width, height, rank, nranks, dst=0, group=None, ctx=None):
    """accum: torch tensor [H*W, 3] (or [H, W, 3]) holding this rank's tiles (others untouched).
    Returns the assembled [H, W, 3] frame on `dst`, None elsewhere.  One collective: gather to `dst`.

    With `ctx` (a Context whose accuBuffer `accum` is and whose communicator is up, comm_init above) the exchange is the
    C ABI's moptix_gather_tiles: pack kernel, grouped ncclSend / ncclRecv over RCCL, unpack kernel -- nothing in Python
    touches the pixels.  Without it (CPU tensors: the gloo tests of the partition logic) the same packed layout goes
    through torch.distributed.gather."""
    import torch
    import torch.distributed as dist
    flat = accum.reshape(-1, 3)
    if ctx is not None:
        ctx.gather_tiles(dst)
        return flat.reshape(height, width, 3) if rank == dst else None
    if nranks == 1:
        return flat.reshape(height, width, 3)
    plan = _tile_plan(width, height, rank, nranks, dst, flat.device)
    send = plan["send"]
    send[:len(plan["mine"])] = flat[plan["mine"]]
    if rank == dst:
        dist.gather(send, plan["recv"], dst=dst, group=group)
        frame = torch.zeros((height * width, 3), dtype=flat.dtype, device=flat.device)
        for r in range(nranks):
            frame[plan["idx"][r]] = plan["recv"][r][:len(plan["idx"][r])]
        return frame.reshape(height, width, 3)
    dist.gather(send, None, dst=dst, group=group)
    return None


def sample_split_seeds(seeds, rank, nranks):
    """Launches i with i % N == rank (SURVEY 8d seed schedule)."""
    return np.asarray(seeds)[rank::nranks]


def reduce_frame(accum, dst=0, group=None, ctx=None):
    """Sample split: sum of the per-rank accumulators on `dst` (with `ctx`: the C ABI's moptix_reduce_frame = ncclReduce)."""
    if ctx is not None:
        ctx.reduce_frame(dst)
        return accum
    import torch.distributed as dist
    dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum
