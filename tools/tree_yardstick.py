"""How far is the device's tree from a good one?  (CPU only: tests/hostsim, the host mirror of the device builder and of the per-lane walk.)

The device builds a binned SAH over the Morton order (pt_lbvh.h: 3 x (kSahBins - 1) planes per node, four-wide collapse by surface area).  The yardstick is the
same pipeline with the EXACT object-split SAH in its place -- every one of the 3 x (count - 1) splits of a node's range, sorted by centroid, is evaluated
(HOSTSIM_SWEEP=1) -- walked by the same code on the same rays: node steps and triangle tests per ray.

    python tools/tree_yardstick.py [scene [iarg]]          scene = million_standin (default) | dining_standin | coffee
"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("YARDSTICK_CHILD"):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    from common import M, hostsim_render
    kind, iarg = sys.argv[1], int(sys.argv[2])
    w, h, spp = (int(x) for x in os.environ.get("YARDSTICK_SIZE", "320x180x2").split("x"))
    hs = M.HostScene("file:coffee" if kind == "coffee" else kind, w, h, **({"iarg": iarg} if iarg else {}))
    for fmt in (128, 64):
        _, c = hostsim_render(hs, M.launch_seeds(spp), node_format=fmt)
        rays = c["primaryRays"] + c["bounceRays"] + c["shadowRays"]
        print("%-34s %d-byte nodes: %9d rays  %6.2f node steps per ray  %6.2f triangle tests per ray  (build %.1f s)" % (
            os.environ.get("YARDSTICK_LABEL", ""), fmt, rays, c["nodeFetches"] / rays, c["triTests"] / rays, c["build_s"]), flush=True)
    sys.exit(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "million_standin"
iarg = int(sys.argv[2]) if len(sys.argv) > 2 else {"million_standin": 1000000, "dining_standin": 6}.get(kind, 0)
print("# %s iarg %d, %s, per-lane walk of tests/hostsim (the device builder's host mirror)" % (kind, iarg, os.environ.get("YARDSTICK_SIZE", "320x180x2")))
for label, env in (("device builder (binned SAH)", {}), ("exact-sweep SAH (yardstick)", {"HOSTSIM_SWEEP": "1"})):
    e = dict(os.environ, YARDSTICK_CHILD="1", YARDSTICK_LABEL=label, **env)
    subprocess.check_call([sys.executable, os.path.abspath(__file__), kind, str(iarg)], env=e)
