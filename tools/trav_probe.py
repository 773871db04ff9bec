"""Traversal-only probe: how fast is the scheduler + node loop + leaf pass WITHOUT any shading, at three and at four workgroups per CU?

Generates a copy of csrc/packetkernel.hip in which the shading visit is replaced by "walk on from the hit in a rotated direction until
the path misses or is 6 deep" (no material code, no lights, no BRDF), builds it twice --
   libmoptix_pa.so : as the product is built (3 workgroups per CU, 512 path slots each, 157 VGPRs)
   libmoptix_pb.so : -DPT_WAVES_PER_SIMD=4 -DPT_KP=88 (4 workgroups per CU, 352 slots each: what fits 40 KB of LDS; 128 VGPRs, 16 spilled)
-- and prints the commands to time them (tools/probe_run.py on the GPU box).  Experiment only; nothing here is part of the product.

   python tools/trav_probe.py && gpurun -- 'MOPTIX_DEVICE_LIB=libmoptix_pa.so BPC=3 python tools/probe_run.py; MOPTIX_DEVICE_LIB=libmoptix_pb.so BPC=4 python tools/probe_run.py'
"""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "minimaloptix_amd", "csrc")
src = open(os.path.join(CSRC, "packetkernel.hip")).read()
old = "          on_result_packet<CNT, FAST, SlotSink>(sc, ps, pk, res, att, ct, sink);"
assert old in src
src = src.replace(old, """          // PROBE: no shading.  The path walks on from the hit in a rotated direction until it misses or is 6 deep.
          if (res.bestPrim < 0 || ps.depth >= 6) { ps.accum = mk3(0.5f, 0.5f, 0.5f); ps.mode = M_NEW_SAMPLE; }
          else { ps.o = ps.o + ps.d * (res.tbest * 0.999f); ps.d = mk3(-ps.d.y, ps.d.z, ps.d.x); ps.depth++; ps.mode = M_TRACE; pk.hasBounce = 1; pk.nShadow = 0; pk.hasScale = 0; cnt<CNT>(ct.bounceRays); }""")
for h in ("megakernel.h", "pt_path.h", "pt_packet.h"):
    src = src.replace('#include "%s"' % h, '#include "%s"' % os.path.join(CSRC, h))
os.makedirs(os.path.join(REPO, "build_probe"), exist_ok=True)
probe = os.path.join(REPO, "build_probe", "pk_probe.hip")
open(probe, "w").write(src)
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "-include", "cstring"]
subprocess.check_call(["make", "-C", REPO, "-s", "device"])
for tag, defs in (("pa", []), ("pb", ["-DPT_WAVES_PER_SIMD=4", "-DPT_KP=88"])):
    obj = os.path.join(REPO, "build_probe", "pk_%s.o" % tag)
    subprocess.check_call(base + defs + ["-c", probe, "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(REPO, "minimaloptix_amd", "lib", "libmoptix_%s.so" % tag)] +
                          [os.path.join(REPO, "build", o) for o in ("moptix_api.o", "megakernel.o", "queuekernel.o", "lbvh.o")] + [obj, "-ldl", "-Wl,-rpath,/opt/rocm/lib"])
    print("built libmoptix_%s.so" % tag)
