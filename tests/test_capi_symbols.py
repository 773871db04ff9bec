"""The C-ABI libraries load on a machine without a GPU, export every symbol the headers declare,
and fail loudly (no CPU fallback) when asked to compute without a device."""
import ctypes as C
import os
import re

import pytest

from common import M, REPO, have_gpu

K = M._capi


def _declared(header):
    txt = open(os.path.join(REPO, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(moptix_[a-z0-9_]+|mohost_[a-z0-9_]+)\s*\(", txt))


def test_device_library_exports_everything_the_header_declares():
    lib = K.device_lib()
    declared = _declared("moptix.h")
    assert declared == set(K.DEVICE_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.moptix_version()


def test_host_library_exports_everything_the_header_declares():
    lib = K.host_lib()
    declared = _declared("moptix_host.h") - {"mohost_scene", "mohost_render_result", "mohost_scene_sizes"}
    assert declared == set(K.HOST_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


@pytest.mark.skipif(have_gpu(), reason="checks the no-device error path")
def test_no_cpu_fallback_without_a_device():
    with pytest.raises(M.MoptixError) as e:
        M.Context(0)
    assert e.value.code == K.ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    res = K.RenderResult()
    rc = K.host_lib().mohost_render_scene(0, 0, None, 64, 36, 1, 0, 0, None, None, None, C.byref(res))
    assert rc != K.MOPTIX_OK


def test_null_arguments_are_rejected():
    lib = K.device_lib()
    assert lib.moptix_create(None, 0) == K.ERR_INVALID
    assert lib.moptix_destroy(None) == K.ERR_INVALID
    assert lib.moptix_set_params(None, None) == K.ERR_INVALID


def test_trace_kernel_resources_are_pinned():
    """The trace kernel lives at its register limit (168 VGPRs for three waves per SIMD) and its allocation is fragile: in round 4 a
    window test added to a sphere loop that the benchmark scene never runs took it from 5 to 69 spilled vector registers and 7 % of
    the frame without any test noticing.  The numbers of the shipped code object (tools/kernel_resources.py reads the AMDGPU metadata
    of the gfx950 code objects embedded in libmoptix.so): the timed instantiation pt_packetkernel<false,true,false,false,true>."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from kernel_resources import kernel_resources
    res = kernel_resources(os.path.join(REPO, "minimaloptix_amd", "lib", "libmoptix.so"))
    bench = [v for k, v in res.items() if "pt_packetkernelILb0ELb1ELb0ELb0ELb1E" in k]
    assert len(bench) == 1, sorted(res)[:5]
    r = bench[0]
    assert r["vgpr_count"] <= 168, r                       # three workgroups of four waves per CU
    # pinned at today's numbers (round 6: 4 vector, 61 scalar under LLVM's "max-ilp" scheduling strategy, which this file's 64-byte-node kernels are
    # built with since it measured 0.85 % faster than the default's 6 / 53; round 5: 2 / 57; round 3: 11 / 111): an edit that moves them has to say so
    # here and in profiles/r06_kernel_resources.txt, where the numbers are kept per round
    assert r["vgpr_spill_count"] <= 4, r
    assert r["sgpr_spill_count"] <= 61, r
    # LDS: three workgroups per CU.  The CU hands LDS out in blocks of 1,280 bytes: 42 blocks = 53,760 bytes each (54,128 bytes ran two
    # workgroups per CU in round 4 -- frame 105 instead of 79 ms -- although hipOccupancyMaxActiveBlocksPerMultiprocessor says 3 up to 54,592)
    assert r["group_segment_fixed_size"] <= 53760, r
    for k, v in res.items():                               # every variant of the packet kernel keeps three workgroups per CU
        if "pt_packetkernel" in k:
            assert v["vgpr_count"] <= 168 and v["group_segment_fixed_size"] <= 53760, (k, v)


def test_queue_and_drain_kernel_resources_are_pinned():
    """pt_queuekernel (variant 3: random_spheres' kernel and the fall-back for scenes outside the packet kernel's limits)
    ships with spilled registers nobody looked at -- pinned here at today's numbers so that at least a regression shows.  The drain kernel
    (csrc/drainkernel.hip) runs one wave per workgroup, eight per CU: it must stay under 256 registers (two waves per SIMD) without spills."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from kernel_resources import kernel_resources
    res = kernel_resources(os.path.join(REPO, "minimaloptix_amd", "lib", "libmoptix.so"))
    queue = {k: v for k, v in res.items() if "pt_queuekernelILb0E" in k}        # the uncounted instantiations
    assert len(queue) >= 4, sorted(res)[:5]
    for k, v in queue.items():
        # round 6: 24-26 vector / 59 scalar once the passes take their own view of the arguments (fresh_args, as in the packet kernel); round 5: 31-40 / 118-120
        assert v["vgpr_count"] <= 128 and v["vgpr_spill_count"] <= 26 and v["sgpr_spill_count"] <= 59, (k, v)
    drain = {k: v for k, v in res.items() if "pt_drainkernelILb0E" in k}
    assert len(drain) == 8, sorted(res)[:5]
    for k, v in drain.items():
        assert v["vgpr_count"] <= 256 and v["vgpr_spill_count"] == 0 and v["group_segment_fixed_size"] <= 20480, (k, v)
