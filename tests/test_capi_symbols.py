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
