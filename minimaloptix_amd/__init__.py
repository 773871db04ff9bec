"""minimaloptix_amd -- MI355X-native (gfx950) path-tracing hot path of CalciferZh/MinimalOptiX.

Layout:
  csrc/   hand-written HIP: megakernel (traversal + fused shading), LBVH build, C ABI
  host/   C++ host: .scene/.obj ingest, scene builders, class MinimalOptiX (renderScene)
  lib/    built artefacts: libmoptix.so (device layer), libmoptix_host.so, moptix_render
  api.py  thin ctypes wrappers over the C ABI (tests / bench / multi-GPU plumbing)
  dist.py tile-split / sample-split across ranks with torch.distributed (RCCL)
"""
from ._capi import MoptixError, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_LIMIT, ERR_COMM  # noqa: F401
from .api import Context, HostScene, launch_seeds, scenes_dir  # noqa: F401

__version__ = "0.1.0"
