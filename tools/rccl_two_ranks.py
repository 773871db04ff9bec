#!/usr/bin/env python3
"""Two (or N) ranks of the C ABI's RCCL path as separate processes, without torch: moptix_comm_unique_id -> file ->
moptix_comm_init, tile split + moptix_gather_tiles and sample split + moptix_reduce_frame, compared with a one-rank render.

    python tools/rccl_two_ranks.py [N] [--same-device]      (parent: starts the ranks, prints the verdict, exit code 0 / 1)

--same-device puts every rank on device 0 (a 1-GPU box): RCCL refuses that ("Duplicate GPU detected") unless the loop-back
transport under tests/ stands in for it (MOPTIX_RCCL_LIB, tests/rccl_loopback).  Every rank has a deadline: a rank that
does not finish is killed and the run fails (no hang)."""
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
W, H, SPP = 256, 144, 8


def rank_main(rank, n, same, idfile, outdir):
    import numpy as np
    import minimaloptix_amd as M
    dev = 0 if same else rank
    ctx = M.Context(dev)
    hs = M.HostScene("file:coffee", W, H)
    seeds = M.launch_seeds(SPP)
    if rank == 0:
        uid = ctx.comm_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 60:
                raise SystemExit("rank %d: no unique id after 60 s" % rank)
            time.sleep(0.01)
        uid = open(idfile, "rb").read()
    ctx.comm_init(uid, rank, n)
    if os.environ.get("RCCL_TWO_RANKS_DEAD_PEER"):
        # a peer that joins the communicator and then never calls the collective: rank 0's gather must come back with
        # MOPTIX_ERR_COMM after "comm_timeout_ms" (csrc/moptix_api.hip comm_wait), not block for good
        if rank != 0:
            time.sleep(4.0)
            return
        ctx.set_option("comm_timeout_ms", 1500)
        ctx.set_partition(rank, n); ctx.load(hs); ctx.accum_clear(); ctx.render(seeds)
        t0 = time.perf_counter()
        try:
            ctx.gather_tiles(0)
        except M.MoptixError as e:
            dt = time.perf_counter() - t0
            ok = e.code == M.ERR_COMM and ctx.get_option("comm_ranks") == 0
            print("rank 0: moptix_gather_tiles came back after %.2f s with code %d (%s); communicator size now %d" % (dt, e.code, e, ctx.get_option("comm_ranks")), flush=True)
            ctx.set_partition(0, 1); ctx.accum_clear(); ctx.render(seeds[:1])      # the context still renders as a one-rank context
            ctx.close()
            raise SystemExit(0 if ok and dt < 30.0 else 1)
        raise SystemExit("rank 0: moptix_gather_tiles returned without an error although rank 1 never sent")
    # tile split + gather
    ctx.set_partition(rank, n)
    ctx.load(hs)
    ctx.accum_clear()
    ctx.render(seeds)
    t0 = time.perf_counter()
    ctx.gather_tiles(0)
    tg = time.perf_counter() - t0
    if rank == 0:
        np.save(os.path.join(outdir, "tile.npy"), ctx.accum_read())
    # sample split + reduce
    ctx.set_partition(0, 1)
    ctx.load(hs)
    ctx.accum_clear()
    ctx.render(seeds[rank::n])
    t0 = time.perf_counter()
    ctx.reduce_frame(0)
    tr = time.perf_counter() - t0
    if rank == 0:
        np.save(os.path.join(outdir, "sample.npy"), ctx.accum_read())
    print("rank %d of %d on device %d: gather %.2f ms, reduce %.2f ms" % (rank, n, dev, tg * 1e3, tr * 1e3), flush=True)
    ctx.comm_destroy()
    ctx.close()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    same = "--same-device" in sys.argv
    if "--rank" in sys.argv:
        i = sys.argv.index("--rank")
        rank_main(int(sys.argv[i + 1]), int(sys.argv[i + 2]), same, sys.argv[i + 3], sys.argv[i + 4])
        return 0
    n = int(args[0]) if args else 2
    deadline = float(os.environ.get("RCCL_TWO_RANKS_DEADLINE", "240"))
    import numpy as np
    with tempfile.TemporaryDirectory() as d:
        idfile = os.path.join(d, "uid")
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), str(n), idfile, d] + (["--same-device"] if same else []), env=env)
                 for r in range(n)]
        t0 = time.time()
        rcs = [None] * n
        while any(rc is None for rc in rcs) and time.time() - t0 < deadline:
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = p.poll()
            if any(rc not in (None, 0) for rc in rcs):
                break                                   # one rank failed: its peers would wait for it forever
            time.sleep(0.05)
        for i, p in enumerate(procs):
            if rcs[i] is None:
                p.kill(); p.wait(); rcs[i] = -9
        if any(rc != 0 for rc in rcs):
            print("FAIL: rank exit codes %s" % rcs)
            return 1
        if os.environ.get("RCCL_TWO_RANKS_DEAD_PEER"):
            print("dead peer: rank 0's collective was aborted at its deadline, every rank exited")
            return 0
        # the one-rank frame, in this process
        import minimaloptix_amd as M
        ctx = M.Context(0)
        hs = M.HostScene("file:coffee", W, H)
        ctx.load(hs); ctx.accum_clear(); ctx.render(M.launch_seeds(SPP))
        ref = ctx.accum_read(); ctx.close()
        tile = np.load(os.path.join(d, "tile.npy")); samp = np.load(os.path.join(d, "sample.npy"))
        bit = bool(np.array_equal(tile.view(np.uint32), ref.view(np.uint32)))
        err = float(np.max(np.abs(samp / SPP - ref / SPP)))
        print("tile split + moptix_gather_tiles over %d ranks: %s; sample split + moptix_reduce_frame: max |delta| %.3g" % (
            n, "bit-identical to the one-rank frame" if bit else "DIFFERS", err))
        return 0 if bit and err <= 2e-6 else 1


if __name__ == "__main__":
    sys.exit(main())
