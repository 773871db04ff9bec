"""The repo's own .obj loader against the reference's vendored tiny_obj_loader.h v1.4.0 (MinimalOptiX/tiny_obj_loader.h),
compiled where it lies under /root/reference by oracle/Makefile into oracle/_ref/tinyobj_dump (build container only; the
binary travels with the snapshot, the header does not).  Same dump program around both (oracle/obj_dump.cpp): the
attribute arrays and every (vertex, normal, texcoord) index must agree byte for byte -- this is MinimalOptiX.cpp:385's
LoadObj call, the one piece of the reference on this path that CAN be run here."""
import glob
import os
import subprocess

import pytest

from common import M, REPO

REF = os.path.join(REPO, "oracle", "_ref", "tinyobj_dump")
OURS = os.path.join(REPO, "oracle", "mobj_dump")

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/tinyobj_dump is built only where /root/reference exists")


def _dump(exe, obj, out):
    r = subprocess.run([exe, obj, out], capture_output=True, text=True)
    return r.returncode, (open(out, "rb").read() if r.returncode == 0 else r.stderr)


def test_every_coffee_mesh_loads_like_tinyobj(tmp_path):
    if not os.path.exists(OURS):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "-s", "mobj_dump"])
    meshes = sorted(glob.glob(os.path.join(M.scenes_dir(), "coffee", "Mesh*.obj")))
    assert len(meshes) == 19
    for m in meshes:
        rc1, a = _dump(REF, m, str(tmp_path / "a.bin"))
        rc2, b = _dump(OURS, m, str(tmp_path / "b.bin"))
        assert rc1 == 0 and rc2 == 0, m
        assert a == b, os.path.basename(m)


CASES = {
    "quad_fan": "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0.5 0.25\nf 1/1/1 2/1/1 3/1/1 4/1/1\n",
    "pentagon": "v 0 0 0\nv 1 0 0\nv 1.5 1 0\nv 0.5 2 0\nv -0.5 1 0\nf 1 2 3 4 5\n",
    "negative_indices": "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf -3//-1 -2//-1 -1//-1\nv 2 0 0\nf -1 -3 -2\n",
    "mixed_forms": "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvn 0 0 1\nf 1/1 2/2 3/3\nf 2//1 4//1 3//1\nf 1 2 4\n",
    "groups_and_objects": "o a\nv 0 0 0\nv 1 0 0\nv 0 1 0\ng g1\nf 1 2 3\ng g2\nv 0 0 1\nf 1 2 4\no b\nf 2 3 4\n",
    "comments_blank_crlf": "# c\r\n\r\nv 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nmtllib x.mtl\r\nusemtl m\r\ns off\r\nf 1 2 3\r\n",
    "whitespace": "v   0.5\t-1e-3   2.5E+1\nv 1 0 0\nv 0 1 0\nf   1   2   3  \n",
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_obj_syntax_cases_load_like_tinyobj(tmp_path, name):
    if not os.path.exists(OURS):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "-s", "mobj_dump"])
    p = tmp_path / (name + ".obj")
    p.write_bytes(CASES[name].encode())
    rc1, a = _dump(REF, str(p), str(tmp_path / "a.bin"))
    rc2, b = _dump(OURS, str(p), str(tmp_path / "b.bin"))
    assert rc1 == 0 and rc2 == 0, (a, b)
    assert a == b
