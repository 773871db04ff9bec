#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline benchmark of the MinimalOptiX render path on MI355X.

  python bench.py --gpus N --steps K --warmup W

runs as typed for every N: with N > 1 and no WORLD_SIZE in the environment it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py ...`
as a child process BEFORE anything touches the GPU and exits with the child's code (the driver's own torch.distributed.run
launch sets WORLD_SIZE and goes straight to the worker).

A step = one frame of the metric's configuration: coffee.obj scene (168,193 triangles, LBVH),
1920x1080, 256 spp = clear + 256 fused launches of the megakernel + ordered sample reduction
(+ for N > 1 one collective per frame: --split tile (default, north_star) gathers the tile-partitioned framebuffer
to rank 0; --split sample gives rank r the launches i = r mod N over the whole frame and sums the accumulators with
one reduce -- BASELINE.json configs[4]'s decomposition).  Scene, BVH and seeds are resident in HBM before the timed
region.  Prints ONE JSON line on rank 0.

N > 1 (no multi-GPU node has been available to this project: the first run on one must explain itself whatever happens):
  * pre-flight: every communicator carries one collective under a 30 s deadline before anything is timed; if that fails on any rank
    with the non-blocking communicator, all ranks fall back to a blocking one once; if it fails again rank 0 prints a JSON line with
    "error" and every rank's diagnosis and the process exits 3 (fresh processes only, never a re-exec);
  * from 8 ranks on BOTH modes are timed, K steps each: one frame at a time, then two frames in flight (two contexts, two
    communicators per rank); `config.modes` carries both, `value` is the better one, `config.mode` says which.  A mode that fails is
    reported under its name and the other one stands.  Below 8 ranks: one frame at a time (two in flight loses there).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

NODE_BYTES = 64             # the node record the packet kernel fetches on coffee (csrc/pt_types.h Node64; get_option "node_format_used" says which)
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, Chip-level parameters)
def _latest(pattern, fallback):
    """The newest committed round of a micro-benchmark's output (profiles/rNN_<pattern>)."""
    import glob
    hits = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_" + pattern)))
    return os.path.relpath(hits[-1], REPO) if hits else fallback


GATHER_CEILING_FILE = _latest("gather_ceiling.txt", os.path.join("profiles", "r03_gather_ceiling.txt"))    # output of tools/micro/gather on MI355X
VALU_CEILING_FILE = _latest("valu_ceiling.txt", os.path.join("profiles", "r04_valu_ceiling.txt"))          # output of tools/micro/valu_issue on MI355X
PIPELINE_FROM_RANKS = int(os.environ.get("MOPTIX_BENCH_PIPELINE_FROM", "8"))     # from this many ranks on the two-frames-in-flight mode is timed as well (after the one-frame mode)
DRAIN_BELOW_DEFAULT = 64    # option drain_below as moptix_create leaves it (csrc/moptix_api.hip)
PREFLIGHT_TIMEOUT_MS = int(os.environ.get("MOPTIX_BENCH_PREFLIGHT_MS", "30000"))
COMM_TIMEOUT_MS = 120000
KERNEL_WAVES_PER_SIMD = 3   # what the trace kernel's 168 registers and 53 KB of LDS allow (csrc/packetkernel.hip)


def source_hash(repo):
    """Identifies the device code a profile belongs to: sha1 over minimaloptix_amd/csrc/* and the Makefile."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(repo, "minimaloptix_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../Makefile"]:
        p = os.path.join(d, f)
        if not os.path.isfile(p):
            continue
        with open(p, "rb") as fh:
            h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def algorithmic_bytes(st, pixels, node_bytes=NODE_BYTES):
    """B = NODE*N_node + 48*N_tri + 108*N_hit + 72*N_lightLoads + 24*N_accum.  N_node counts fetches of
    four-child nodes; NODE is the record the kernel read: 64 bytes (csrc/pt_types.h Node64, round 3) where the scene's own
    paths say the compressed form is cheaper -- coffee --, else the 128-byte form of rounds 1-2.  N_accum is counted per
    pixel per launch batch (the per-sample buffer traffic is not claimed)."""
    return node_bytes * st.nodeFetches + 48 * st.triTests + 108 * st.closestHits + 72 * st.lightLoads + 24 * pixels


def gather_ceilings(repo):
    """Ceilings of the traversal's access pattern (every lane fetches its own record, next address dependent on the data)
    measured on MI355X with tools/micro/gather.hip; the committed output is the evidence (profiles/).  Per record size
    R in (64, 128): "l2_R" = best GB/s over all occupancies / chains from the 3.1 MB table (one XCD's L2 holds it),
    "l2_R_at_12_waves" = the dependent single-chain rate at the trace kernel's own 12 waves per CU, "ic_R" = best from the
    38 MB table (Infinity Cache).  None when the file is missing."""
    try:
        rows = [l.split() for l in open(os.path.join(repo, GATHER_CEILING_FILE)) if l.strip() and not l.startswith("#")]
    except OSError:
        return None
    best = {}
    for r in rows:
        rec, mb, waves, chains, tbs = int(r[0]), float(r[1]), int(r[2]), int(r[3]), float(r[6])
        for key, ok in (("l2_%d" % rec, mb < 4.0), ("l2_%d_at_12_waves" % rec, mb < 4.0 and waves == 12 and chains == 1),
                        ("ic_%d" % rec, 30.0 < mb < 50.0)):
            if ok:
                best[key] = max(best.get(key, 0.0), tbs * 1000.0)
    return best if "l2_128" in best and "l2_64" in best else None


def valu_ceilings(repo):
    """Measured issue rates of the vector ALUs (tools/micro/valu_issue.hip on MI355X; the committed output is the evidence):
    {instruction: {waves per SIMD: G wave-instructions/s, chip-wide}}.  None when the file is missing."""
    out = {}
    try:
        for l in open(os.path.join(repo, VALU_CEILING_FILE)):
            f = l.split()
            if len(f) == 6 and not l.startswith("#"):
                out.setdefault(f[0], {})[int(f[1])] = float(f[3])
    except (OSError, ValueError):
        return None
    return out if "v_fma_f32" in out and "v_min_f32" in out else None


def fp32_vector_peak_tflops(repo, waves_per_simd=8):
    """The FP32 vector peak as measured: v_fma_f32 wave-instructions/s x 64 lanes x 2 flop.  BASELINE config 2 (brute-force
    spheres) is priced against this number -- the same measurement the trace kernel's instruction rate is priced against."""
    c = valu_ceilings(repo)
    return None if not c else c["v_fma_f32"][waves_per_simd] * 128e-3


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


def read_traffic(repo, tag=None):
    """Counters of the trace kernel from the committed rocprofv3 PMC passes (collected separately: profiles/README.md,
    tools/prof_bench.sh + tools/make_traffic_json.py; tag = "c2" / "c4" / "c5": the passes of that BASELINE config,
    profiles/rNN_traffic_<tag>.json).  Only used when they were collected on THIS device code (source_hash); otherwise the
    fields are null rather than stale."""
    p = os.path.join(repo, "profiles", "traffic.json") if not tag else os.path.join(repo, _latest("traffic_%s.json" % tag, os.path.join("profiles", "r05_traffic_%s.json" % tag)))
    try:
        t = json.load(open(p))
    except Exception:
        return None
    if t.get("source_hash") != source_hash(repo):
        return None
    return t


def roofline_block(achieved_gbs, launch_ms, nlaunch, bytes_per_launch, bytes_per_ray, rays, reduce_ms, kernel, traffic, ceil, node_bytes=NODE_BYTES, vceil=None,
                   waves_per_simd=KERNEL_WAVES_PER_SIMD, flops_per_launch=None):
    """The dominant kernel against its roofs.

    Head of the block (the contract's form): SURVEY 8(d)'s ALGORITHMIC bytes per launch / the launch duration measured here,
    against the HBM peak.  Those bytes are served by the XCDs' L2s and the Infinity Cache (what really crossed the fabric is
    `traffic` / `hbm_frac`, from the PMC passes), so the fraction says how much traversal work per second the kernel does in
    the survey's currency, not that HBM is 0.9 busy: `algorithmic_frac_of_hbm` repeats it under its own name, and `binding` /
    `binding_frac` name the unit the counters say limits the kernel (the vector ALUs' issue rate) and how busy it is.
    A scene without triangles (BASELINE config 2; SURVEY 8(d): "FP32-VALU-bound -- report flops there, not bytes") gets the same
    block headed by its FP32 rate against the measured v_fma_f32 rate (`flops_per_launch`).

    `valu_issue`: what the counters say binds the kernel -- the issue rate of the vector ALUs at the three waves per SIMD its
    registers and LDS allow.  achieved = SQ_INSTS_VALU per launch (PMC pass of THIS device code, profiles/traffic.json) / the
    launch duration measured here; peak = the chip's measured rate for independent v_fma_f32 at three waves per SIMD
    (VALU_CEILING_FILE: the newest committed profiles/rNN_valu_ceiling.txt, tools/micro/valu_issue.hip; v_min / v_max / v_cvt / v_cmp / v_cndmask issue at about
    half of it, `peak_half_rate_class`); useful_lane_frac = frac x the share of lanes active in an issued instruction.
    `live_fields` / `replayed_fields` say which numbers were measured in this run and which come from the committed passes."""
    launch_s = launch_ms * 1e-3
    fabric_gb = traffic.get("traffic_GB_per_launch") if traffic else None
    gpeak = ceil["l2_%d" % node_bytes] if ceil else None
    sq = (traffic.get("SQ") or {}) if traffic else {}
    valu = sq.get("SQ_INSTS_VALU")
    lanes = round(sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"]), 4) if sq.get("SQ_THREAD_CYCLES_VALU") and sq.get("SQ_ACTIVE_INST_VALU") else None
    if flops_per_launch and vceil:
        tf = flops_per_launch / launch_s / 1e12
        fpeak = vceil["v_fma_f32"][waves_per_simd] * 0.128
        head = {"bound": "valu_fp32", "achieved": round(tf, 2), "peak": round(fpeak, 1), "unit": "TFLOP/s", "frac": round(tf / fpeak, 4), "traffic": fabric_gb,
                "achieved_source": "primitive tests of one launch (counting launch of this run) x flop per test (sphere 17, quad 20: csrc/pt_path.h trav_begin) / mean launch "
                                   "duration of the timed region (HIP events on the launch stream)",
                "peak_source": "%s: v_fma_f32 at the kernel's %d waves per SIMD x 64 lanes x 2 flop (measured; the 157.3 TFLOP/s of the data sheet assume 2.4 GHz and 2.0 clocks "
                               "per instruction)" % (VALU_CEILING_FILE, waves_per_simd),
                "frac_of_spec_157_TFLOPs": round(tf / 157.3, 4), "flops_per_launch": float(flops_per_launch)}
    else:
        head = {"bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                "traffic": fabric_gb,
                "algorithmic_frac_of_hbm": round(achieved_gbs / HBM_PEAK_GBS, 4),
                "achieved_source": "SURVEY 8(d) algorithmic bytes of one launch (counting launch of this run: %d B per node fetch, 48 per triangle test, 108 per closest "
                                   "hit, 72 per light record, 24 per pixel) / mean launch duration of the timed region (HIP events on the launch stream)" % node_bytes,
                "note": "head = the contract's form (algorithmic bytes against the HBM peak); those bytes are served by L2 / Infinity Cache -- what crossed the fabric is "
                        "`traffic` (`hbm_frac` of the peak) -- so `frac` is NOT HBM utilisation; what limits the kernel is named in `binding`"}
    head.update({"kernel": kernel, "node_bytes": int(node_bytes), "launch_ms": round(launch_ms, 3), "launches_timed": int(nlaunch),
                 "algorithmic_bytes_per_launch": int(bytes_per_launch), "bytes_per_ray": round(bytes_per_ray, 1)})
    vi = None
    if valu and vceil:
        ach = valu / launch_s / 1e9
        peak = vceil["v_fma_f32"][waves_per_simd]
        half = vceil["v_min_f32"][waves_per_simd]
        vi = {"achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instructions/s", "frac": round(ach / peak, 4),
              "peak_source": "%s: v_fma_f32, %d waves per SIMD, 256 CUs (measured, includes the clock the chip holds under that load)" % (VALU_CEILING_FILE, waves_per_simd),
              "peak_half_rate_class": round(half, 1), "frac_of_half_rate_class": round(ach / half, 4),
              "peak_best_occupancy": round(max(vceil["v_fma_f32"].values()), 1),
              "instructions_per_ray": round(valu / max(1, rays), 2), "lane_utilisation": lanes,
              "useful_lane_frac": round(ach / peak * lanes, 4) if lanes else None,
              "valu_active_frac": traffic.get("valu_active_frac")}
    head["valu_issue"] = vi
    # what the counters say limits this kernel: the vector ALUs.  binding_frac = share of all SIMD-cycles with a vector instruction in the pipe
    # (SQ_ACTIVE_INST_VALU x 4 / (1,024 SIMDs x cycles), PMC pass of this device code); null until such a pass exists for this workload
    head["binding"] = "valu_issue"
    head["binding_frac"] = min(1.0, traffic["valu_active_frac"]) if traffic and traffic.get("valu_active_frac") else None
    head["binding_source"] = ("vector pipes busy: SQ_ACTIVE_INST_VALU x 4 / (1,024 SIMDs x cycles), replayed from the committed PMC passes of this device code (the counter "
                              "sums the waves' own time with a vector instruction in flight, so dense code with four waves per SIMD reads above 1: capped); the issue "
                              "RATE against the measured ceiling is in `valu_issue`; NOTEBOOK.md round 5 has the experiments that say neither look-ups nor bytes bind it")
    head.update({
        "gather_peak_GBps": round(gpeak, 1) if gpeak else None, "gather_frac": round(achieved_gbs / gpeak, 4) if gpeak else None,
        "gather_peak_source": GATHER_CEILING_FILE + " (tools/micro/gather.hip: dependent per-lane gathers of %d-B records, 3.1 MB table, best over occupancies)" % node_bytes,
        "infinity_cache_gather_GBps": round(ceil["ic_%d" % node_bytes], 1) if ceil and ("ic_%d" % node_bytes) in ceil else None,
        "hbm_frac": round(fabric_gb / launch_s / HBM_PEAK_GBS, 4) if fabric_gb else None,
        # FETCH_SIZE tallies 64 B per fabric read request; a request of this kernel's gathers fills a 128-byte line (calibrated with
        # the gather micro-benchmark, profiles/r03_fetch_size_calibration.txt): upper figure with the read side doubled
        "hbm_frac_if_128B_reads": round((2.0 * traffic["FETCH_SIZE_KB_per_launch"] + traffic["WRITE_SIZE_KB_per_launch"]) * 1024 / 1e9 / launch_s / HBM_PEAK_GBS, 4)
        if traffic and traffic.get("FETCH_SIZE_KB_per_launch") else None,
        "fabric_GBps": round(fabric_gb / launch_s, 1) if fabric_gb else None,
        "fabric_bytes_per_ray": round(fabric_gb * 1e9 / max(1, rays), 1) if fabric_gb else None,
        "tcc_hit_rate": traffic.get("tcc_hit_rate") if traffic else None,
        # the CU's vector-memory address path (one L1 tag look-up per clock); PMC
        "l1_address_unit_busy_frac": traffic.get("ta_busy_frac") if traffic else None,
        "l1_lookups_per_cu_clock": traffic.get("l1_lookups_per_cu_clock") if traffic else None,
        "reduce_ms_total": round(reduce_ms, 3),
        "live_fields": ["achieved", "frac", "launch_ms", "launches_timed", "algorithmic_bytes_per_launch", "bytes_per_ray", "gather_frac", "reduce_ms_total",
                        "valu_issue.achieved / frac / useful_lane_frac (numerator replayed, launch duration live)"],
        "replayed_fields": ["traffic", "hbm_frac", "hbm_frac_if_128B_reads", "fabric_GBps", "fabric_bytes_per_ray", "tcc_hit_rate", "l1_address_unit_busy_frac",
                            "l1_lookups_per_cu_clock", "valu_issue.instructions_per_ray / lane_utilisation / valu_active_frac (profiles/traffic.json, "
                            "rocprofv3 --pmc passes of this device code: source_hash-gated, null otherwise)", "peaks (%s, %s, HBM spec)" % (VALU_CEILING_FILE, GATHER_CEILING_FILE)],
        # no hardware counter is read inside this run: everything under replayed_fields comes from the committed --pmc passes
        "pmc_live": False,
    })
    return head


# BASELINE.json configs other than the headline: scene kind -> (tag of its committed PMC passes, configs[] index, what it is)
OTHER_CONFIGS = {"random_spheres": ("c2", 1, "random_spheres (analytic spheres + quads, NoAccel)"),
                 "dining_standin": ("c4", 3, "dining-room stand-in (multi-mesh, Disney BRDF; the asset is absent upstream)"),
                 "million_standin": ("c5", 4, "1 M-triangle glass knot stand-in (the asset is absent upstream)")}
CONFIGS_FILE = _latest("configs.json", os.path.join("profiles", "r05_configs.json"))      # one bench.py line per BASELINE config (tools/prof_configs.sh <tag>)


def workload_text(a):
    if a.scene == "file:coffee":
        return "coffee.obj LBVH build+traverse, %dx%d, %d spp (BASELINE.json configs[2])" % (a.width, a.height, a.spp)
    tag, idx, what = OTHER_CONFIGS.get(a.scene, (None, None, a.scene))
    return "%s, %dx%d, %d spp on ONE GPU%s" % (what, a.width, a.height, a.spp, " (BASELINE.json configs[%d]'s scene; parity case, not the headline)" % idx if idx is not None else "")


def other_configs(repo):
    """The committed bench.py lines of BASELINE configs 2, 4, 5 (the newest profiles/rNN_configs.json, written by tools/prof_configs.sh on MI355X):
    their rates and roofline fractions ride along in the headline's block so that one line says what limits each kernel."""
    try:
        rows = json.load(open(os.path.join(repo, CONFIGS_FILE)))
    except Exception:
        return None
    out = {}
    for tag, d in rows.items():
        r = d.get("roofline", {})
        out[tag] = {"workload": d.get("config", {}).get("workload"), "value_Mrays_s": d.get("value"), "ms_per_step": d.get("ms_per_step"),
                    "bound": r.get("bound"), "frac": r.get("frac"), "binding": r.get("binding"), "binding_frac": r.get("binding_frac"),
                    "source_hash": d.get("config", {}).get("source_hash")}
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--scene", default="file:coffee")
    ap.add_argument("--iarg", type=int, default=0, help="integer argument of a stand-in scene (dining_standin: copies, million_standin: triangles, random_spheres: spheres)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-leg", action="store_true", help="skip the untimed fast_shading comparison leg (PMC passes)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--split", choices=("tile", "sample"), default="tile", help="multi-GPU decomposition (N > 1)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_workers(n, argv, script=None, env=None):
    """`python bench.py --gpus N` as typed: N fresh worker processes under torch.distributed.run on a port chosen
    now.  Called before this process has touched the GPU, and it only ever starts CHILD processes (a process that has
    initialised HIP must not be replaced).  Returns the launcher's exit code; rank 0's JSON line passes through stdout."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script or os.path.abspath(__file__)] + list(argv)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=e)


class GpuFrame:
    """One rank's share of the benchmark frame on its MI355X, through the C ABI (minimaloptix_amd.Context)."""
    backend = "nccl"

    def __init__(self, a, rank, world, local):
        import torch
        import minimaloptix_amd as M
        from minimaloptix_amd import dist as D
        self.torch, self.D, self.a = torch, D, a
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", local)
        torch.cuda.set_device(local)
        self.ctx = M.Context(local)                 # raises when the HIP library / device is missing: no fallback
        hs = M.HostScene(a.scene, a.width, a.height, iarg=a.iarg) if getattr(a, "iarg", 0) else M.HostScene(a.scene, a.width, a.height)
        self.hs_sizes = (int(hs.sizes.nSpheres), int(hs.sizes.nQuads), int(hs.sizes.nFaces))
        # MOPTIX_BENCH_EMULATE_RANKS=n (single process only): render rank 0's share of an n-way tile split, to study a
        # rank's launch on a 1-GPU box; the JSON line then describes that share, not the frame
        self.emu = int(os.environ.get("MOPTIX_BENCH_EMULATE_RANKS", "0")) if world == 1 else 0
        self.part_rank, self.part_n = (0, self.emu) if self.emu > 1 else (rank, world)
        self.sample_split = a.split == "sample" and self.part_n > 1
        if self.sample_split:
            if a.spp % self.part_n:
                raise SystemExit("--split sample needs spp divisible by the number of ranks")
        else:
            self.ctx.set_partition(self.part_rank, self.part_n)
        self.ctx.load(hs)
        self.build_ms_cold = self.ctx.accel_info().buildMs
        self.ctx.build_accel("Trbvh" if self.ctx.accel_info().nTriangles else "NoAccel")      # same tree again, code already loaded
        self.build_ms_warm = self.ctx.accel_info().buildMs
        W, H = a.width, a.height
        self.accum = torch.zeros(H * W * 3, dtype=torch.float32, device=self.device)
        self.ctx.accum_bind(self.accum.data_ptr())
        self.seeds = M.launch_seeds(a.spp)
        if self.sample_split:
            self.seeds = D.sample_split_seeds(self.seeds, self.part_rank, self.part_n)      # launches i = rank mod N
        # Two frames in flight (default from PIPELINE_FROM_RANKS ranks on; MOPTIX_BENCH_PIPELINE=1 / 0 forces it on / off): a rank's
        # launch is short (frame / N) and its last ~13 ms are a drain in which a few depth-capped paths finish while most of the GPU
        # idles.  Two contexts render alternate frames on their own streams (and gather them on their own communicators, oldest frame
        # first on every rank); every frame is still completed and gathered inside the timed region.  Measured on the shares of ONE
        # GPU (profiles/r05_scaling_emulation.txt): an 8-way share 43.4 -> 39.7 ms (89 % -> 98 % of ideal), a 4-way share
        # 77.1 -> 76.2 ms, the whole frame 310 -> 323 ms (worse: hence not below 8 ranks).  Never run beside RCCL's kernels on a real
        # multi-GPU node; the collectives have a deadline (csrc/moptix_api.hip comm_wait).
        # Which modes this run times (run_rank): MOPTIX_BENCH_PIPELINE=1 / 0 forces two frames in flight / one frame at a time only;
        # otherwise one frame at a time, and from PIPELINE_FROM_RANKS ranks on two in flight AS WELL (the better one is `value`).
        pl = os.environ.get("MOPTIX_BENCH_PIPELINE", "")
        if pl == "1":
            self.mode_list = ["two_in_flight"]
        elif pl == "0" or self.part_n < PIPELINE_FROM_RANKS or world == 1:
            self.mode_list = ["one_frame"]
        else:
            self.mode_list = ["one_frame", "two_in_flight"]
        self.M, self.hs, self.local = M, hs, local
        self.pipeline = False
        self.ctxs, self.accums = [self.ctx], [self.accum]
        self.pending = [False]
        self.frame_no = 0
        self.collective_s, self.collectives = 0.0, 0
        # the frame's collective runs behind the C ABI on the context's own RCCL communicator (moptix_gather_tiles /
        # moptix_reduce_frame); MOPTIX_BENCH_FORCE_DIST=1 brings a one-rank communicator up on a 1-GPU box
        self.use_comm = world > 1 or os.environ.get("MOPTIX_BENCH_FORCE_DIST") == "1"
        if self.use_comm:
            D.comm_init(self.ctx, rank, world, self.device)

    def modes(self):
        return list(self.mode_list)

    def set_mode(self, mode):
        """"one_frame": a frame is rendered and collected before the next starts.  "two_in_flight": a second context (own stream, own
        accuBuffer, own communicator) renders the alternate frames; created here, the first time the mode is asked for (collective:
        every rank makes its second communicator at the same point)."""
        self.flush()
        self.pipeline = mode == "two_in_flight"
        # Two frames in flight run WITHOUT the drain kernel: a frame's drain kernel is queued behind its packet kernel, and by the time that one has
        # ended the other frame's persistent grid has filled every CU (3 x 53 KB of LDS, 504 of 512 registers per SIMD) -- the drain kernel, and
        # with it the frame's collective, would wait for a whole launch (measured: an 8-way share 43.7 ms with it, 41.1 without; profiles/r06_scaling_emulation.txt)
        self.ctx.set_option("drain_below", 0 if self.pipeline else DRAIN_BELOW_DEFAULT)
        if self.pipeline and len(self.ctxs) == 1:
            a, torch = self.a, self.torch
            ctx2 = self.M.Context(self.local)
            if not self.sample_split:
                ctx2.set_partition(self.part_rank, self.part_n)
            if getattr(self, "comm_blocking", 0):
                ctx2.set_option("comm_blocking", 1)
            ctx2.set_option("drain_below", 0)
            ctx2.load(self.hs)
            accum2 = torch.zeros(a.height * a.width * 3, dtype=torch.float32, device=self.device)
            ctx2.accum_bind(accum2.data_ptr())
            self.ctxs.append(ctx2); self.accums.append(accum2); self.pending.append(False)
            if self.use_comm:
                self.D.comm_init(ctx2, self.rank, self.world, self.device)
        self.frame_no = 0

    def preflight(self, which=None):
        """One collective per communicator under a short deadline, before anything is timed: a run that cannot communicate says so
        within PREFLIGHT_TIMEOUT_MS instead of after comm_timeout_ms inside the timed region.  Returns None or what went wrong."""
        if not self.use_comm:
            return None
        for j, c in enumerate(self.ctxs):
            if which is not None and j != which:
                continue
            try:
                c.set_option("comm_timeout_ms", PREFLIGHT_TIMEOUT_MS)
                self.accums[j].zero_(); self.sync()
                self.collect(j)
                c.set_option("comm_timeout_ms", COMM_TIMEOUT_MS)
            except Exception as e:
                return "communicator %d: %s" % (j, e)
        return None

    def reinit_comm(self, blocking):
        """Every rank drops its communicators and makes new ones (collective), blocking or not."""
        self.comm_blocking = 1 if blocking else 0
        for c in self.ctxs:
            try:
                c.comm_destroy()
            except Exception:
                pass
            c.set_option("comm_blocking", self.comm_blocking)
            self.D.comm_init(c, self.rank, self.world, self.device)

    def comm_kind(self):
        if not self.use_comm:
            return None
        return "non-blocking (ncclCommInitRankConfig, calls polled against comm_timeout_ms)" if self.ctx.get_option("comm_nonblocking_used") else "blocking (ncclCommInitRank)"

    def verify_frame(self):
        """N > 1, rank 0, outside the timed region: the frame the ranks just rendered and collected against rank 0's OWN render of the whole frame
        (a second context, one rank, same seeds).  The tile split must give the same bits -- every pixel's samples are the same work wherever
        they run, and the gather only moves them --, the sample split the same sums within dist.SAMPLE_SPLIT_TOL.  The first run on a real
        multi-GPU node thereby checks the data path through RCCL, not only times it."""
        if self.world <= 1 or self.rank != 0:
            return None
        torch, a = self.torch, self.a
        self.flush(); self.sync()
        got = self.accums[0].clone()
        ctx1 = self.M.Context(self.local)
        try:
            ctx1.load(self.hs)
            ref = torch.zeros_like(got)
            ctx1.accum_bind(ref.data_ptr())
            ctx1.render(self.M.launch_seeds(a.spp))
            torch.cuda.synchronize()
            if self.sample_split:
                err = float((got - ref).abs().max().item()) / max(1, a.spp)
                ok = err <= self.D.SAMPLE_SPLIT_TOL
                return {"what": "sample split + reduce against rank 0's own one-GPU render of the frame", "ok": bool(ok), "max_abs_diff_per_sample": err, "bound": self.D.SAMPLE_SPLIT_TOL}
            same = bool(torch.equal(got.view(torch.int32), ref.view(torch.int32)))
            return {"what": "tile split + gather against rank 0's own one-GPU render of the frame", "ok": same, "bit_identical": same,
                    "pixels_differing": 0 if same else int((got.view(-1, 3) != ref.view(-1, 3)).any(dim=1).sum().item())}
        finally:
            ctx1.close()

    def first_frame(self):
        """A context's FIRST frame: no depth history orders its work yet, so the launch ends with the deepest paths walking alone
        (csrc/drainkernel.hip takes them over).  Trace-kernel ms of one frame rendered like that (HIP events on the launch stream: the
        host-side loading of the code object is not in it); the history is dropped again afterwards so that the counted launch that
        follows is a first frame, too."""
        self.ctx.set_option("forget_history", 1)
        self.accum.zero_(); self.sync()
        self.ctx.kernel_time(reset=True)
        self.ctx.render(self.seeds)
        ms, _n = self.ctx.kernel_time()
        self.ctx.set_option("forget_history", 1)
        return ms

    def sync(self):
        self.torch.cuda.synchronize()

    def count(self):
        """Counting launch (untimed): rays and algorithmic bytes of this rank's share of one frame (deterministic)."""
        a = self.a
        self.accum.zero_(); self.sync()
        st = self.ctx.render_counted(self.seeds)
        px = a.width * a.height if self.sample_split else len(self.D.tile_pixel_indices(a.width, a.height, self.part_rank, self.part_n))
        self.node_bytes = self.ctx.get_option("node_format_used") if self.ctx.get_option("kernel_variant_used") == 4 else 128
        # the counting launch stamps its own timeline: first wave in -> last wave out (drain kernel included), and how much of that came
        # after the last work item.  The first call describes a context's first frame, a later one the frames the timed region renders
        if not hasattr(self, "counted_tail_first_ms"):
            self.counted_tail_first_ms = self.ctx.get_option("counted_tail_us") * 1e-3
            self.counted_span_first_ms = self.ctx.get_option("counted_span_us") * 1e-3
        self.counted_span_ms = self.ctx.get_option("counted_span_us") * 1e-3
        self.counted_tail_ms = self.ctx.get_option("counted_tail_us") * 1e-3
        nS, nQ, nF = self.hs_sizes
        # scenes without an acceleration structure are FP32-VALU bound -- flops, not bytes: primitive tests x flop per test
        self.flops = float(st.analyticTests) * (17.0 * nS + 20.0 * nQ) / max(1, nS + nQ) if nF == 0 else None
        return st.rays, algorithmic_bytes(st, px, self.node_bytes)

    def collect(self, j):                                               # the frame's one collective
        a = self.a
        if not self.use_comm:
            return None
        t0 = time.perf_counter()                                        # the C ABI call returns after its stream has drained
        if self.sample_split:
            r = self.D.reduce_frame(self.accums[j], dst=0, ctx=self.ctxs[j])
        else:
            r = self.D.gather_tiles(self.accums[j].view(a.height * a.width, 3), a.width, a.height, self.rank, self.world, dst=0, ctx=self.ctxs[j])
        self.collective_s += time.perf_counter() - t0
        self.collectives += 1
        return r

    def _finish(self, j):                                               # frame in context j: wait for it, collect it
        if self.pending[j]:
            self.ctxs[j].sync()
            self.collect(j)
            self.pending[j] = False

    def step(self):
        if not self.pipeline:
            self.accum.zero_()
            self.sync()
            self.ctx.render(self.seeds)                                 # blocking: launches + ordered reduction
            return self.collect(0)
        j = self.frame_no % 2
        self.frame_no += 1
        self._finish(j)                                                 # the frame launched two steps ago
        self.accums[j].zero_()
        self.torch.cuda.current_stream().synchronize()
        self.ctxs[j].render_async(self.seeds)
        self.pending[j] = True
        return None

    def flush(self):                                                    # oldest frame first: same collective order on every rank
        if self.pipeline:
            j = self.frame_no % 2
            self._finish(j); self._finish(1 - j)

    def reset_kernel_time(self):
        for c in self.ctxs:
            c.kernel_time(reset=True)
        self.collective_s, self.collectives = 0.0, 0

    def kernel_times(self):
        kms, n, red = 0.0, 0, 0.0
        for c in self.ctxs:                                             # both contexts when two frames are in flight
            k_, n_ = c.kernel_time()
            kms += k_; n += n_; red += c.reduce_time()
        return kms, n, red

    def fast_leg(self, total_rays):
        """The reference compiles its programs with -use_fast_math (utils_host.cpp:30-32).  `value` is the exact mode (bit
        parity with the oracle); the opt-in approximate BRDF arithmetic ("fast_shading": same rays, weights within ~1e-6)
        is timed beside it on one GPU, outside the timed region, and reported as a separate field."""
        self.flush()
        if self.world != 1 or self.emu > 1 or self.pipeline:
            return None
        ctx, accum = self.ctx, self.accum
        try:
            ctx.set_option("fast_shading", 1)
            accum.zero_(); self.sync(); ctx.render(self.seeds)
            accum.zero_(); self.sync()
            t0 = time.perf_counter(); ctx.render(self.seeds); self.sync(); tf = time.perf_counter() - t0
            return {"value": round(total_rays / tf / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(tf * 1e3, 3), "steps": 1,
                    "note": "option fast_shading = 1 (v_rcp / v_sqrt inside disneyPdf / disneyEval only); not the mode `value` is measured in"}
        except Exception as e:                                          # never lose the primary metric to the side leg
            return {"error": str(e)}
        finally:
            try:
                ctx.set_option("fast_shading", 0)
            except Exception:
                pass

    def describe(self):
        info = self.ctx.accel_info()
        v = self.ctx.get_option("kernel_variant_used")
        return {"kernel_variant": v, "bvh_nodes": int(info.nNodes), "bvh_depth": int(info.treeDepth),
                "path_slots_per_workgroup": self.ctx.get_option("path_slots") if v == 4 else 512,
                # the context's first build loads the builder's code objects (~5 ms): both numbers, the warm one is the builder's
                "bvh_build_ms": round(float(self.build_ms_warm), 3), "bvh_build_ms_first_in_context": round(float(self.build_ms_cold), 3),
                "kernel": "pt_packetkernel (trace)" if v == 4 else "pt_queuekernel (trace)"}

    def parallelism(self):
        a, world = self.a, self.world
        if world > 1 and os.environ.get("MOPTIX_RCCL_LIB"):
            return "TEST PLUMBING: %d ranks on one device through %s (not RCCL, not a measurement)" % (world, os.path.basename(os.environ["MOPTIX_RCCL_LIB"]))
        if world == 1:
            par = "single GPU" if self.emu <= 1 else "EMULATION: rank 0 of a %d-way %s split on one GPU" % (self.emu, a.split)
        elif self.sample_split:
            par = "sample-split x%d (rank r renders launches i = r mod %d) + RCCL reduce" % (world, world)
        else:
            par = "tile-split x%d (8x8 tiles dealt round-robin, rotating per group) + RCCL gather" % world
        return par

    def traffic(self):
        a = self.a
        if self.world != 1 or self.emu > 1:
            return None
        if a.scene == "file:coffee" and (a.width, a.height, a.spp) == (1920, 1080, 256):
            return read_traffic(REPO)
        if a.scene in OTHER_CONFIGS:                       # the PMC passes of tools/prof_configs.sh belong to one size per config
            t = read_traffic(REPO, OTHER_CONFIGS[a.scene][0])
            if t and t.get("workload") == workload_text(a):
                return t
        return None


def run_rank(a, frame_cls=GpuFrame):
    """One rank of the benchmark: W untimed warm-up frames, then exactly K frames between barrier + synchronize on both
    sides, MAX over ranks; rank 0 prints the JSON line.  `frame_cls` is the renderer (GpuFrame; the CPU test of the
    N > 1 plumbing passes a stand-in that lives under tests/)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    # MOPTIX_BENCH_FORCE_DIST=1 brings the process group up for a single rank too (exercises init/barrier/all_reduce
    # on a 1-GPU box); the measured path is unchanged
    use_dist = world > 1 or os.environ.get("MOPTIX_BENCH_FORCE_DIST") == "1"
    # MOPTIX_BENCH_BACKEND=gloo + MOPTIX_BENCH_DEVICE=0 (+ MOPTIX_RCCL_LIB=tests/rccl_loopback/...): N ranks of THIS code on a one-GPU box -- the
    # control plane over gloo, every rank on one device, the frame's collective through the loop-back transport.  Test plumbing
    # (tests/test_gpu_rccl_loopback.py): it exercises the N > 1 branches of this file, it measures nothing.
    backend = os.environ.get("MOPTIX_BENCH_BACKEND", frame_cls.backend)
    if "MOPTIX_BENCH_DEVICE" in os.environ:
        local = int(os.environ["MOPTIX_BENCH_DEVICE"])
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise SystemExit("MASTER_PORT is not set (start N > 1 ranks through `python bench.py --gpus N` or torch.distributed.run)")
            os.environ["MASTER_PORT"] = str(free_port())
        if backend == "nccl":
            if torch.cuda.device_count() <= local:
                raise SystemExit("%d devices needed, %d present" % (world, torch.cuda.device_count()))
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    fr = frame_cls(a, rank, world, local)
    dev = fr.device
    if use_dist and world == 1:                     # forced single-rank group: the collectives of the N > 1 path, once
        t = torch.arange(12, dtype=torch.float32, device=dev).reshape(4, 3)
        got = [torch.empty_like(t)]
        dist.gather(t, got, dst=0)
        assert torch.equal(got[0], t)

    def barrier():
        if use_dist:
            dist.barrier()
        fr.sync()

    def everyone(text):
        """What every rank has to say (None = nothing), on every rank, over the control plane."""
        if not (use_dist and world > 1):
            return [text]
        got = [None] * world
        dist.all_gather_object(got, text)
        return got

    def give_up(what, per_rank_text):
        """The run cannot produce a number: rank 0 still prints ONE JSON line that says why, every rank exits 3."""
        if rank == 0:
            print(json.dumps({"metric": "Mrays/s (primary+bounce+shadow rays traced per second)", "value": None, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps,
                              "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                              "error": what, "config": {"workload": workload_text(a), "split": a.split if world > 1 else None,
                                                        "ranks": {"diagnosis": per_rank_text}}}), flush=True)
        if use_dist:
            try:
                dist.destroy_process_group()
            except Exception:
                pass
        raise SystemExit(3)

    # ---- pre-flight: one collective per communicator under a short deadline; one fall-back to a blocking communicator ----
    comm_note = None
    if hasattr(fr, "preflight"):
        said = everyone(fr.preflight())
        if any(said):
            comm_note = "pre-flight failed with the %s communicator (%s); every rank made a blocking one" % (
                "non-blocking" if not getattr(fr, "comm_blocking", 0) else "blocking", "; ".join("rank %d: %s" % (i, t) for i, t in enumerate(said) if t))
            try:
                fr.reinit_comm(blocking=True)
                said = everyone(fr.preflight())
            except Exception as e:
                said = everyone(str(e))
            if any(said):
                give_up("the ranks cannot complete a collective (pre-flight, %d ms deadline; non-blocking and blocking communicator both tried)" % PREFLIGHT_TIMEOUT_MS,
                        ["rank %d: %s" % (i, t or "ok") for i, t in enumerate(said)])

    first_ms = fr.first_frame() if hasattr(fr, "first_frame") and world == 1 and not os.environ.get("MOPTIX_BENCH_PIPELINE") else None
    my_rays, my_bytes = fr.count()
    tot = torch.tensor([float(my_rays), float(my_bytes)], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tot)
    total_rays, total_bytes = float(tot[0].item()), float(tot[1].item())

    # ---- the timed region, once per mode: W untimed frames, then exactly K frames between barrier + synchronize, MAX over ranks ----
    modes = fr.modes() if hasattr(fr, "modes") else ["one_frame"]
    results = {}
    frame_check = None
    for mi, mode in enumerate(modes):
        failed = None
        try:
            if hasattr(fr, "set_mode"):
                fr.set_mode(mode)
                if mode == "two_in_flight" and hasattr(fr, "preflight"):
                    failed = fr.preflight(which=1)                      # the second communicator, made just now
        except Exception as e:
            failed = "%s: %s" % (type(e).__name__, e)
        # every rank says how entering the mode went BEFORE anyone starts a frame: a rank that could not must not leave the others waiting in
        # a collective for it
        said = everyone(failed)
        if any(said):
            results[mode] = {"error": "; ".join("rank %d: %s" % (i, t) for i, t in enumerate(said) if t)}
            if not any("ms_per_step" in r for r in results.values()):
                break
            continue
        try:
            if True:
                for _ in range(a.warmup):
                    fr.step()
                fr.flush()
                if mi == 0 and a.warmup > 0 and hasattr(fr, "first_frame"):
                    fr.count()                                          # the counted launch again, in the state the timed frames are in
                fr.reset_kernel_time()
                barrier()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    fr.step()
                fr.flush()
                barrier()
                dt = time.perf_counter() - t0
        except Exception as e:                                          # a collective that hit its deadline (MOPTIX_ERR_COMM), a HIP error ...
            failed = "%s: %s" % (type(e).__name__, e)
        said = everyone(failed)
        if any(said):
            results[mode] = {"error": "; ".join("rank %d: %s" % (i, t) for i, t in enumerate(said) if t)}
            if mi + 1 < len(modes) or not any("ms_per_step" in r for r in results.values()):
                # the communicators of a failed mode are gone (aborted): nothing further can be timed with them
                break
            continue
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        kms, nlaunch, reduce_ms = fr.kernel_times()
        # per-rank diagnosis of an N > 1 run (one driver run has to explain itself): each rank's mean trace-kernel ms per frame, the
        # drain of its counted launch, its collective's wall ms per frame and its ray count, gathered on every rank
        mine = torch.tensor([kms / max(1, a.steps), getattr(fr, "counted_tail_ms", -1.0), getattr(fr, "counted_span_ms", -1.0),
                             getattr(fr, "collective_s", 0.0) * 1e3 / max(1, a.steps), float(my_rays), getattr(fr, "counted_tail_first_ms", -1.0)], dtype=torch.float64, device=dev)
        per_rank = [mine.clone() for _ in range(world)]
        if use_dist and world > 1:
            dist.all_gather(per_rank, mine)
        per_rank = [[float(x) for x in t.tolist()] for t in per_rank]
        results[mode] = {"ms_per_step": dt / a.steps * 1e3, "dt": dt, "kms": kms, "nlaunch": nlaunch, "reduce_ms": reduce_ms, "per_rank": per_rank}
        if mi == 0 and hasattr(fr, "verify_frame"):
            try:
                frame_check = fr.verify_frame()
            except Exception as e:
                frame_check = {"ok": False, "error": "%s: %s" % (type(e).__name__, e)}
    good = {m: r for m, r in results.items() if "ms_per_step" in r}
    if not good:
        give_up("no mode could be timed", ["%s: %s" % (m, r.get("error")) for m, r in results.items()])
    mode = min(good, key=lambda m: good[m]["ms_per_step"])
    best = good[mode]
    dt, kms, nlaunch, reduce_ms, per_rank = best["dt"], best["kms"], best["nlaunch"], best["reduce_ms"], best["per_rank"]
    if hasattr(fr, "set_mode") and len(modes) > 1:
        try:
            fr.set_mode("one_frame")
        except Exception:
            pass
    fast = None if a.no_fast_leg else fr.fast_leg(total_rays)

    if rank == 0:
        W, H = a.width, a.height
        ms_per_step = dt / a.steps * 1e3
        launch_ms = kms / max(1, nlaunch)
        passes_per_step = max(1, nlaunch // max(1, a.steps))
        achieved = my_bytes / passes_per_step / max(launch_ms * 1e-3, 1e-12) / 1e9        # GB/s, rank 0's trace kernel
        d = fr.describe()
        lean = d.get("kernel_variant") == 3 and getattr(fr, "flops", None) is not None      # scenes without triangles: queuekernel_lean.hip, four workgroups per CU
        roof = roofline_block(achieved, launch_ms, nlaunch, my_bytes // passes_per_step, my_bytes / max(1, my_rays), my_rays,
                              reduce_ms, d.pop("kernel"), fr.traffic(), gather_ceilings(REPO), getattr(fr, "node_bytes", NODE_BYTES), valu_ceilings(REPO),
                              waves_per_simd=4 if lean else KERNEL_WAVES_PER_SIMD,
                              flops_per_launch=(fr.flops / passes_per_step) if getattr(fr, "flops", None) else None)
        headline = a.scene == "file:coffee" and (W, H, a.spp) == (1920, 1080, 256)
        if headline and world == 1:
            roof["other_configs"] = other_configs(REPO)      # BASELINE configs 2, 4, 5: their committed lines' rates and fractions

        def mode_summary(r):
            if "ms_per_step" not in r:
                return r
            pr = r["per_rank"]
            return {"ms_per_frame": round(r["ms_per_step"], 3), "value_Mrays_s": round(total_rays / (r["dt"] / a.steps) / 1e6, 2),
                    "kernel_ms_per_frame_min_max": [round(min(x[0] for x in pr), 3), round(max(x[0] for x in pr), 3)],
                    "collective_ms_per_frame_max": round(max(x[3] for x in pr), 3)}
        out = {
            "metric": "Mrays/s (primary+bounce+shadow rays traced per second, %s)" % ("coffee.obj 1920x1080 256spp" if headline else "%s %dx%d %dspp" % (a.scene, W, H, a.spp)),
            "value": round(total_rays / (dt / a.steps) / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": getattr(fr, "data", "reference scene scenes/coffee (168,193 triangles; Mesh010 missing upstream), synthetic seed schedule tea16(i,0)" if a.scene == "file:coffee"
                            else "stand-in scene authored by this project (minimaloptix_amd/host/standin_scenes.cpp, scenes.cpp), synthetic seed schedule tea16(i,0)"),
            "pmc_live": False,
            "config": dict({"workload": workload_text(a),
                            "scene": a.scene, "width": W, "height": H, "spp": a.spp, "rays_per_frame": int(total_rays),
                            "parallelism": fr.parallelism() + (", two frames in flight" if mode == "two_in_flight" else ""), "split": a.split if world > 1 else None,
                            # which of the timed modes `value` is, and every mode that was timed (K steps each) or tried
                            "mode": mode, "pipeline": mode == "two_in_flight", "modes": {m: mode_summary(r) for m, r in results.items()},
                            "communicator": fr.comm_kind() if hasattr(fr, "comm_kind") else None, "communicator_note": comm_note,
                            # N > 1: the collected frame against rank 0's own one-GPU render (GpuFrame.verify_frame): bit-identical for the tile split
                            "frame_check": frame_check,
                            "ms_per_frame": round(ms_per_step, 3),
                            # a context's first frame (no depth history orders its work yet): trace-kernel ms, next to the timed frames' mean
                            "first_frame_kernel_ms": round(first_ms, 3) if first_ms is not None else None,
                            "source_hash": source_hash(REPO),
                            "ranks": {"kernel_ms_per_frame": [round(r[0], 3) for r in per_rank],
                                      "kernel_ms_per_frame_min_max": [round(min(r[0] for r in per_rank), 3), round(max(r[0] for r in per_rank), 3)],
                                      "counted_launch_tail_ms": [round(r[1], 3) for r in per_rank], "counted_launch_span_ms": [round(r[2], 3) for r in per_rank],
                                      "counted_launch_tail_ms_first_frame": [round(r[5], 3) for r in per_rank],
                                      "collective_ms_per_frame": [round(r[3], 3) for r in per_rank], "rays_per_frame": [int(r[4]) for r in per_rank],
                                      "comm_ranks_seen": (fr.ctx.get_option("comm_ranks") if getattr(fr, "use_comm", False) else 1) if hasattr(fr, "ctx") else world,
                                      "note": "kernel ms: HIP events round the trace kernel (packet + drain kernel) on the launch stream; tail / span: s_memrealtime stamps of a "
                                              "counting launch (work items ran out -> last wave of the drain kernel out; first wave in -> last wave out): "
                                              "counted_launch_tail_ms in the state of the timed frames (after the warm-up: the depth history orders the work), "
                                              "..._first_frame for a context's first launch; collective: wall time of moptix_gather_tiles / moptix_reduce_frame incl. "
                                              "its stream synchronisation"}}, **d),
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


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    a = parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # device_count() may initialise the HIP runtime in this (parent) process.  That is harmless only because the workers
        # are started as CHILD processes (launch_workers -> subprocess); never exec from here, and keep GPU work out of the parent
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus:
            raise SystemExit("%d devices needed, %d present" % (a.gpus, have))
        raise SystemExit(launch_workers(a.gpus, argv))
    run_rank(a)


if __name__ == "__main__":
    main()
