#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline benchmark of the MinimalOptiX render path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one frame of the metric's configuration: coffee.obj scene (168,193 triangles, LBVH),
1920x1080, 256 spp = clear + 256 fused launches of the megakernel + ordered sample reduction
(+ for N > 1 one collective per frame: --split tile (default, north_star) gathers the tile-partitioned framebuffer
to rank 0; --split sample gives rank r the launches i = r mod N over the whole frame and sums the accumulators with
one reduce -- BASELINE.json configs[4]'s decomposition).  Scene, BVH and seeds are resident in HBM before the timed
region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, Chip-level parameters)
# What actually serves the traversal's bytes: the 3.4 MB of nodes and 16 MB of triangle records are L2 / Infinity-Cache
# resident.  Ceilings for independent random 64-byte gathers measured with tools/micro/gather.hip on MI355X.
L2_GATHER_GBS = 13200.0     # 3.6 MB table (one XCD's L2 holds it)
IC_GATHER_GBS = 4300.0      # 38 MB table (Infinity Cache)


def source_hash(repo):
    """Identifies the device code a profile belongs to: sha1 over minimaloptix_amd/csrc/* and the Makefile."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(repo, "minimaloptix_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../Makefile"]:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def algorithmic_bytes(st, pixels):
    """B = NODE*N_node + 48*N_tri + 108*N_hit + 72*N_lightLoads + 24*N_accum, with NODE = 128:
    the node record is a four-child 128-byte node (one L2 line) since the binary tree was widened; N_node counts
    those fetches.  N_accum is counted per pixel per launch batch (the per-sample buffer traffic is not claimed)."""
    return 128 * st.nodeFetches + 48 * st.triTests + 108 * st.closestHits + 72 * st.lightLoads + 24 * pixels


def cpu_baseline(width, height, target_s):
    """north_star / BASELINE.md section 2: "a CPU build of the same megakernel" -- the per-lane path code of
    minimaloptix_amd/csrc/pt_path.h and the same LBVH (tests/hostsim, test infrastructure; kind "port": the reference
    has no CPU path), compiled with g++ -O3 -fopenmp, all host cores, same seeds.  Bounded sample: whole frames for
    the first n of the 256 launch seeds, n chosen from the first launch so that the leg takes about `target_s`
    seconds.  The LBVH build (single thread) is excluded from the rate and reported next to it."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from common import hostsim_render
    import minimaloptix_amd as M
    hs = M.HostScene("file:coffee", width, height)
    seeds = M.launch_seeds(256)
    _, c = hostsim_render(hs, seeds[:1])
    rays, secs, build_s, n = c["primaryRays"] + c["bounceRays"] + c["shadowRays"], c["render_s"], c["build_s"], 1
    more = max(0, min(255, int(target_s / max(secs, 1e-3)) - 1))
    if more:
        _, c = hostsim_render(hs, seeds[1:1 + more])
        rays += c["primaryRays"] + c["bounceRays"] + c["shadowRays"]; secs += c["render_s"]; n += more
    return {"value": round(rays / secs / 1e6, 3), "unit": "Mrays/s", "cores": int(c["threads"]), "kind": "port",
            "sample": "CPU build of the same megakernel (pt_path.h + same LBVH, g++ -O3 -fopenmp): coffee %dx%d, %d of 256 "
                      "launches, %.1f s, %d rays; LBVH build %.2f s on one thread excluded" % (width, height, n, secs, rays, build_s)}


def read_traffic(repo):
    """Counters of the trace kernel from the committed rocprofv3 PMC passes (collected separately: profiles/README.md,
    tools/prof_bench.sh + tools/make_traffic_json.py).  Only used when they were collected on THIS device code
    (source_hash); otherwise the fields are null rather than stale."""
    p = os.path.join(repo, "profiles", "traffic.json")
    try:
        t = json.load(open(p))
    except Exception:
        return None
    if t.get("source_hash") != source_hash(repo):
        return None
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--scene", default="file:coffee")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--split", choices=("tile", "sample"), default="tile", help="multi-GPU decomposition (N > 1)")
    a = ap.parse_args()

    import torch
    import minimaloptix_amd as M
    from minimaloptix_amd import dist as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
    import torch.distributed as dist
    # MOPTIX_BENCH_FORCE_DIST=1 brings the RCCL group up for a single rank too (exercises init/barrier/all_reduce
    # on a 1-GPU box); the measured path is unchanged
    use_dist = world > 1 or os.environ.get("MOPTIX_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    dev = torch.device("cuda", local)
    if use_dist and world == 1:                 # forced single-rank group: the collectives of the N>1 path, once
        t = torch.arange(12, dtype=torch.float32, device=dev).reshape(4, 3)
        got = [torch.empty_like(t)]
        dist.gather(t, got, dst=0)
        assert torch.equal(got[0], t)
    ctx = M.Context(local)                      # raises when the HIP library / device is missing: no fallback
    hs = M.HostScene(a.scene, a.width, a.height)
    # MOPTIX_BENCH_EMULATE_RANKS=n (single process only): render rank 0's share of an n-way tile split, to study a
    # rank's launch on a 1-GPU box; the JSON line then describes that share, not the frame
    emu = int(os.environ.get("MOPTIX_BENCH_EMULATE_RANKS", "0")) if world == 1 else 0
    part_rank, part_n = (0, emu) if emu > 1 else (rank, world)
    sample_split = a.split == "sample" and part_n > 1
    if sample_split:
        if a.spp % part_n:
            raise SystemExit("--split sample needs spp divisible by the number of ranks")
    else:
        ctx.set_partition(part_rank, part_n)
    ctx.load(hs)
    info = ctx.accel_info()
    W, H = a.width, a.height
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    ctx.accum_bind(accum.data_ptr())
    seeds = M.launch_seeds(a.spp)
    if sample_split:
        seeds = D.sample_split_seeds(seeds, part_rank, part_n)      # launches i = rank mod N (SURVEY 8d seed schedule)
    # Frame pipelining (N > 1): a rank's launch is short (frame / N), and the last ~20 ms of every launch are a drain in
    # which a few deep paths finish while most of the GPU idles.  Two contexts render alternate frames on their own
    # streams, so the next frame fills the CUs the draining one has released; every frame is still completed and
    # gathered inside the timed region.
    # Opt-in (MOPTIX_BENCH_PIPELINE=1): measured +8 % on an emulated 8-way share of one GPU, but never run together with
    # RCCL on a real multi-GPU node, where the gather kernels would have to find room next to a persistent grid.
    pipeline = (part_n > 1) and os.environ.get("MOPTIX_BENCH_PIPELINE", "0") == "1"
    ctxs, accums = [ctx], [accum]
    if pipeline:
        ctx2 = M.Context(local)
        if not sample_split:
            ctx2.set_partition(part_rank, part_n)
        ctx2.load(hs)
        accum2 = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
        ctx2.accum_bind(accum2.data_ptr())
        ctxs.append(ctx2); accums.append(accum2)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # counting launch (untimed): rays and algorithmic bytes of one frame are deterministic
    accum.zero_(); torch.cuda.synchronize()
    st = ctx.render_counted(seeds)
    my_pixels = W * H if sample_split else len(D.tile_pixel_indices(W, H, part_rank, part_n))
    my_rays, my_bytes = st.rays, algorithmic_bytes(st, my_pixels)
    tot = torch.tensor([float(my_rays), float(my_bytes)], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tot)
    total_rays, total_bytes = float(tot[0].item()), float(tot[1].item())

    def gather(j):                                                  # the frame's one collective
        if world == 1:
            return None
        if sample_split:
            return D.reduce_frame(accums[j], dst=0)
        return D.gather_tiles(accums[j].view(H * W, 3), W, H, rank, world, dst=0)

    pending = [False] * len(ctxs)
    frame_no = [0]

    def finish(j):                                                  # frame in context j: wait for it, collect it
        if pending[j]:
            ctxs[j].sync()
            gather(j)
            pending[j] = False

    def step():
        if not pipeline:
            accum.zero_()
            torch.cuda.synchronize()
            ctx.render(seeds)                                       # blocking: launches + ordered reduction
            return gather(0)
        j = frame_no[0] % 2
        frame_no[0] += 1
        finish(j)                                                   # the frame launched two steps ago
        accums[j].zero_()
        torch.cuda.current_stream().synchronize()
        ctxs[j].render_async(seeds)
        pending[j] = True
        return None

    def flush():                                                    # oldest frame first: same collective order on every rank
        if pipeline:
            j = frame_no[0] % 2
            finish(j); finish(1 - j)

    for _ in range(a.warmup):
        step()
    flush()
    for c_ in ctxs:
        c_.kernel_time(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    flush()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    kms, nlaunch, reduce_ms = 0.0, 0, 0.0
    for c_ in ctxs:                                                 # both contexts when two frames are in flight
        k_, n_ = c_.kernel_time()
        kms += k_; nlaunch += n_; reduce_ms += c_.reduce_time()

    # The reference compiles its programs with -use_fast_math (utils_host.cpp:30-32).  `value` above is the exact mode (bit
    # parity with the oracle); the opt-in approximate BRDF arithmetic ("fast_shading": same rays, weights within ~1e-6) is
    # timed beside it on one GPU, outside the timed region, and reported as a separate field.
    fast = None
    if world == 1 and emu <= 1 and not pipeline:
        try:
            ctx.set_option("fast_shading", 1)
            accum.zero_(); torch.cuda.synchronize(); ctx.render(seeds)
            accum.zero_(); torch.cuda.synchronize()
            tf0 = time.perf_counter(); ctx.render(seeds); torch.cuda.synchronize(); tf = time.perf_counter() - tf0
            fast = {"value": round(total_rays / tf / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(tf * 1e3, 3), "steps": 1,
                    "note": "option fast_shading = 1 (v_rcp / v_sqrt inside disneyPdf / disneyEval only); not the mode `value` is measured in"}
        finally:
            ctx.set_option("fast_shading", 0)

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        launch_ms = kms / max(1, nlaunch)
        passes_per_step = max(1, nlaunch // max(1, a.steps))
        achieved = my_bytes / passes_per_step / (launch_ms * 1e-3) / 1e9          # GB/s, rank 0's trace kernel
        if world == 1:
            par = "single GPU" if emu <= 1 else "EMULATION: rank 0 of a %d-way %s split on one GPU" % (emu, a.split)
        elif sample_split:
            par = "sample-split x%d (rank r renders launches i = r mod %d) + RCCL reduce" % (world, world)
        else:
            par = "tile-split x%d (8x8 tiles dealt round-robin, rotating per group) + RCCL gather" % world
        if pipeline:
            par += ", two frames in flight"
        traffic = read_traffic(REPO) if (world == 1 and emu <= 1 and a.scene == "file:coffee" and (W, H, a.spp) == (1920, 1080, 256)) else None
        fabric_gb = traffic.get("traffic_GB_per_launch") if traffic else None
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": fabric_gb,
                "kernel": "pt_packetkernel (trace)" if ctx.get_option("kernel_variant_used") == 4 else "pt_queuekernel (trace)", "launch_ms": round(launch_ms, 3), "launches_timed": int(nlaunch),
                "algorithmic_bytes_per_launch": int(my_bytes // passes_per_step),
                "bytes_per_ray": round(my_bytes / max(1, my_rays), 1), "reduce_ms_total": round(reduce_ms, 3),
                # `achieved` counts ALGORITHMIC bytes and most of them never leave L2 / Infinity Cache: it is the
                # agreed figure of merit, not HBM bandwidth.  What the memory system really did, from the PMC passes:
                "served_from": "L2 / Infinity Cache (19 MB working set); see fabric_* for what crossed the fabric",
                "fabric_GBps": round(fabric_gb / (launch_ms * 1e-3), 1) if fabric_gb else None,
                "fabric_frac_of_hbm_peak": round(fabric_gb / (launch_ms * 1e-3) / HBM_PEAK_GBS, 4) if fabric_gb else None,
                "tcc_hit_rate": traffic.get("tcc_hit_rate") if traffic else None,
                "fabric_bytes_per_ray": round(fabric_gb * 1e9 / max(1, my_rays), 1) if fabric_gb else None,
                "l2_gather_ceiling_GBps": L2_GATHER_GBS, "frac_of_l2_gather_ceiling": round(achieved / L2_GATHER_GBS, 4),
                "infinity_cache_gather_ceiling_GBps": IC_GATHER_GBS}
        out = {
            "metric": "Mrays/s (primary+bounce+shadow rays traced per second, coffee.obj 1920x1080 256spp)",
            "value": round(total_rays / (dt / a.steps) / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "reference scene scenes/coffee (168,193 triangles; Mesh010 missing upstream), synthetic seed schedule tea16(i,0)",
            "config": {"workload": "coffee.obj LBVH build+traverse, %dx%d, %d spp (BASELINE.json configs[2])" % (W, H, a.spp),
                       "scene": a.scene, "width": W, "height": H, "spp": a.spp, "rays_per_frame": int(total_rays),
                       "parallelism": par, "split": a.split if world > 1 else None, "pipeline": bool(pipeline),
                       "kernel_variant": ctx.get_option("kernel_variant_used"), "bvh_nodes": int(info.nNodes), "bvh_depth": int(info.treeDepth),
                       "bvh_build_ms": round(float(info.buildMs), 3), "ms_per_frame": round(ms_per_step, 3),
                       "source_hash": source_hash(REPO)},
            "roofline": roof,
        }
        if fast is not None:
            out["fast_shading_mode"] = fast
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(W, H, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
