"""Worker script of tests/test_bench_helpers.py::test_gpus_2_launcher_path_on_cpu -- NOT part of the product.

bench.py's multi-rank plumbing (self-launch under torch.distributed.run, barrier + MAX-over-ranks timing, the frame's one
collective, rank 0's JSON line) driven on CPU: `gloo` instead of RCCL, and in place of the GPU renderer the CPU build of the
same per-lane code (tests/hostsim), which renders the whole small frame and keeps this rank's tiles / launches."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench                                   # noqa: E402
from common import M, hostsim_render           # noqa: E402
from minimaloptix_amd import dist as D         # noqa: E402


class StubFrame:
    backend = "gloo"
    data = "TEST STUB (tests/bench_stub_worker.py): CPU build of the per-lane code, not a measurement"
    pipeline = False

    def __init__(self, a, rank, world, local):
        import torch
        self.torch, self.a, self.rank, self.world = torch, a, rank, world
        self.device = torch.device("cpu")
        self.hs = M.HostScene(a.scene, a.width, a.height)
        self.sample_split = a.split == "sample" and world > 1
        seeds = M.launch_seeds(a.spp)
        self.seeds = D.sample_split_seeds(seeds, rank, world) if self.sample_split else seeds
        self.mine = None if self.sample_split else D.tile_pixel_indices(a.width, a.height, rank, world)
        self.accum = torch.zeros(a.height * a.width, 3)
        self.frames = []

    def sync(self):
        pass

    # ---- the hooks of bench.run_rank that GpuFrame has (modes, pre-flight, fall-back), driven by the environment of the test ----
    def modes(self):
        return ["one_frame", "two_in_flight"] if os.environ.get("BENCH_STUB_MODES") == "2" else ["one_frame"]

    def set_mode(self, mode):
        self.pipeline = mode == "two_in_flight"
        if self.pipeline and os.environ.get("BENCH_STUB_FAIL_MODE_B") and self.rank == 1:
            raise RuntimeError("stub: the second mode cannot start on rank 1")

    def preflight(self, which=None):
        fail = os.environ.get("BENCH_STUB_FAIL_PREFLIGHT", "")          # "once": the first communicator kind fails on rank 1; "always": every kind
        if self.rank == 1 and (fail == "always" or (fail == "once" and not getattr(self, "comm_blocking", 0))):
            return "stub: rank 1 cannot complete a collective"
        return None

    def reinit_comm(self, blocking):
        self.comm_blocking = 1 if blocking else 0

    def comm_kind(self):
        return "stub, %s" % ("blocking" if getattr(self, "comm_blocking", 0) else "non-blocking")

    def _render(self):
        img, c = hostsim_render(self.hs, self.seeds)
        flat = img.reshape(-1, 3)
        if self.mine is not None:                       # tile split: only this rank's tiles are "rendered"
            keep = np.zeros_like(flat); keep[self.mine] = flat[self.mine]; flat = keep
        return flat, c

    def count(self):
        flat, c = self._render()
        share = 1.0 if self.mine is None else len(self.mine) / float(self.a.width * self.a.height)
        rays = int((c["primaryRays"] + c["bounceRays"] + c["shadowRays"]) * share)
        return rays, 1000 * rays

    def step(self):
        flat, _ = self._render()
        self.accum.copy_(self.torch.from_numpy(flat))
        a = self.a
        if self.world == 1:
            out = self.accum
        elif self.sample_split:
            out = D.reduce_frame(self.accum, dst=0)
        else:
            out = D.gather_tiles(self.accum, a.width, a.height, self.rank, self.world, dst=0)
        if self.rank == 0:
            self.frames.append(np.asarray(out).reshape(a.height, a.width, 3).copy())
            np.save(os.environ["BENCH_STUB_FRAME"], self.frames[-1])
        return out

    def flush(self):
        pass

    def reset_kernel_time(self):
        pass

    def kernel_times(self):
        return 1.0 * self.a.steps, self.a.steps, 0.0

    def fast_leg(self, total_rays):
        return None

    def describe(self):
        return {"kernel_variant": -1, "bvh_nodes": 0, "bvh_depth": 0, "bvh_build_ms": 0.0, "kernel": "hostsim (CPU stub)"}

    def parallelism(self):
        return "CPU stub x%d (%s split, gloo)" % (self.world, self.a.split)

    def traffic(self):
        return None


if __name__ == "__main__":
    bench.run_rank(bench.parse_args(sys.argv[1:]), StubFrame)
