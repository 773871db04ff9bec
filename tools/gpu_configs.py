"""BASELINE.json configs other than the headline: parity vs oracle at reduced size + timing at full size."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M, O, oracle_scene, rmse   # noqa: E402
ctx = M.Context(0)
def parity(kind, kw, res, spp):
    hs = M.HostScene(kind, res[0], res[1], **kw); seeds = M.launch_seeds(spp)
    ctx.load(hs); ctx.accum_clear(); st = ctx.render_counted(seeds); g = ctx.accum_read()
    t0 = time.time(); o, ost = oracle_scene(hs).render(seeds); dt = time.time() - t0
    print("PARITY %-16s %dx%d spp %d: rmse %.3g rays %d/%d oracle %.1fs tris %d" % (kind, res[0], res[1], spp, rmse(g / spp, o / spp), st.rays, ost.rays, dt, hs.sizes.nFaces), flush=True)
def timing(kind, kw, res, spp):
    hs = M.HostScene(kind, res[0], res[1], **kw); seeds = M.launch_seeds(spp)
    ctx.load(hs); a = ctx.accel_info()
    ctx.accum_clear(); st = ctx.render_counted(seeds)
    nb = ctx.get_option("node_format_used") if ctx.get_option("kernel_variant_used") == 4 else 128      # the node record that launch fetched
    B = nb * st.nodeFetches + 48 * st.triTests + 108 * st.closestHits + 72 * st.lightLoads + 24 * res[0] * res[1]
    ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time()
    print("TIMING %-16s %dx%d spp %d: %.1f ms (%d launches) %.1f Mrays/s rays/sample %.2f  alg %.2f TB/s  analytic tests/s %.3g | tris %d nodes %d (%d B) depth %d build %.2f ms" % (
        kind, res[0], res[1], spp, ms, n, st.rays / ms / 1e3, st.rays / st.samples, B / ms / 1e9, st.analyticTests / ms * 1e3, a.nTriangles, a.nNodes, nb, a.treeDepth, a.buildMs), flush=True)
    if a.nTriangles == 0:
        # scenes without an acceleration structure are FP32-VALU bound -- flops, not bytes.  Per primitive test: sphere
        # 17 flop (oc 3, b = d.oc 5, c = oc.oc - r^2 7, disc 2; the roots only where disc >= 0 are not counted), quad 20 flop
        # (plane t 11 incl. the division as one, point 6 -> two projected-edge dots are not reached by most rays, 3 counted)
        nS, nQ = hs.sizes.nSpheres, hs.sizes.nQuads
        flops = st.analyticTests * (17.0 * nS + 20.0 * nQ) / max(1, nS + nQ)
        import bench
        v = bench.valu_ceilings(REPO)
        w = 4 if ctx.get_option("kernel_variant_used") == 3 else 3      # scenes without triangles run the lean queue kernel: four workgroups per CU
        p3, p8 = v["v_fma_f32"][w] * 0.128, v["v_fma_f32"][8] * 0.128      # measured: G v_fma_f32/s x 64 lanes x 2 flop (profiles/r04_valu_ceiling.txt)
        tf = flops / ms / 1e9
        print("       %-16s FP32: %.3g primitive tests x %.1f flop = %.2f TFLOP/s = %.3f of the measured v_fma_f32 rate at the kernel's %d waves per SIMD "
              "(%.1f TFLOP/s), %.3f of the best measured (8 waves per SIMD, %.1f), %.3f of the 157.3 TFLOP/s spec" % (
            kind, st.analyticTests, (17.0 * nS + 20.0 * nQ) / max(1, nS + nQ), tf, tf / p3, w, p3, tf / p8, p8, tf / 157.3), flush=True)
    return ctx.resolve_rgb8(spp)
parity("cornell_quads", {}, (256, 256), 4)
parity("random_spheres", dict(iarg=497), (160, 90), 2)
parity("dining_standin", dict(iarg=3), (96, 54), 1)
parity("million_standin", dict(iarg=200000), (96, 54), 1)
from PIL import Image
for name, kind, kw, res, spp in [("c1", "cornell_quads", {}, (256, 256), 64), ("c2", "random_spheres", dict(iarg=497), (1280, 720), 64),
                                 ("c2b", "random_spheres", dict(iarg=256), (1280, 720), 64),
                                 ("c4", "dining_standin", dict(iarg=6), (1920, 1080), 16), ("c5", "million_standin", dict(iarg=1000000), (1920, 1080), 16)]:
    img = timing(kind, kw, res, spp)
    Image.fromarray(img).resize((480, 270 if res[0] != res[1] else 480)).save(os.path.join(REPO, "gpurun_out", "cfg_%s.png" % name))
