"""Acceleration build alone (for rocprofv3 --kernel-trace --stats): KIND [IARG] builds the scene's tree three times."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import minimaloptix_amd as M   # noqa: E402
kind = sys.argv[1] if len(sys.argv) > 1 else "file:coffee"
iarg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = M.Context(0)
hs = M.HostScene(kind, 256, 144, iarg=iarg)
for i in range(3):
    ctx.load(hs); a = ctx.accel_info()
    print("%s: %d triangles, %d nodes, depth %d, build %.3f ms" % (kind, a.nTriangles, a.nNodes, a.treeDepth, a.buildMs), flush=True)
