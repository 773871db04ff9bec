"""Registers, spills and LDS of the kernels in a device library, read from the code objects embedded in it.

    python3 tools/kernel_resources.py [minimaloptix_amd/lib/libmoptix.so] [name filter]

The library's .hip_fatbin section holds one clang offload bundle per translation unit; each gfx950 entry is an ELF code object whose
AMDGPU metadata note lists per kernel: .vgpr_count, .sgpr_count, .vgpr_spill_count, .sgpr_spill_count, .group_segment_fixed_size (LDS),
.private_segment_fixed_size (scratch).  The trace kernel lives at its register limit and its allocation is fragile (NOTEBOOK.md: an
unrelated edit took it from 5 to 69 spilled vector registers and 7 % of the frame), so tests/test_capi_symbols.py pins these numbers."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path, arch="gfx950"):
    data = open(lib_path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if arch in triple and size:
                out.append(data[i + off:i + off + size])
        pos = i + len(MAGIC)
    return out


def kernel_resources(lib_path, arch="gfx950"):
    """{demangled-ish kernel symbol: {field: int}} over all code objects of `arch` in the library."""
    res = {}
    for blob in code_objects(lib_path, arch):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for block in txt.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            if not name:
                continue
            d = {}
            for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size"):
                m = re.search(r"\.%s:\s+(\d+)" % k, block)
                if m:
                    d[k] = int(m.group(1))
            res[name.group(1)] = d
    return res


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "minimaloptix_amd", "lib", "libmoptix.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else "kernel"
    for k, v in sorted(kernel_resources(lib).items()):
        if flt in k:
            print("%-100s %s" % (k[:100], " ".join("%s=%d" % (a.replace("_count", "").replace("_fixed_size", ""), b) for a, b in v.items())))
