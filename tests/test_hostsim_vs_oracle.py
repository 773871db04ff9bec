"""The megakernel's per-lane code compiled for the host (tests/hostsim) against the recursive CPU
oracle: same paths, same ray counts.  This is what is checked before any GPU time is spent; the
GPU tests then check that the device runs the same code to the same bits."""
import os

import numpy as np
import pytest

from common import node64_boxes, M, hostsim_bvh, hostsim_render, oracle_scene, rmse

CASES = [("spheres", dict(farg=0.5), (128, 72), 3), ("cornell_quads", {}, (48, 48), 2),
         ("random_spheres", dict(iarg=97), (96, 54), 2), ("file:coffee", {}, (96, 54), 2)]


@pytest.mark.parametrize("kind,kw,res,spp", CASES)
def test_flattened_state_machine_equals_recursive_oracle(kind, kw, res, spp):
    hs = M.HostScene(kind, res[0], res[1], **kw)
    seeds = M.launch_seeds(spp)
    o, ost = oracle_scene(hs).render(seeds)
    h, hc = hostsim_render(hs, seeds)
    assert rmse(h / spp, o / spp) < 2e-6
    assert (hc["primaryRays"], hc["bounceRays"], hc["shadowRays"], hc["closestHits"]) == \
           (ost.primaryRays, ost.bounceRays, ost.shadowRays, ost.closestHits)


@pytest.mark.parametrize("builder", [1, 0])
@pytest.mark.parametrize("leaf", [1, 4, 8])
def test_lbvh_mirror_is_a_valid_bvh(leaf, builder):
    hs = M.HostScene("file:coffee", 64, 36)
    nodes, tris, prim, root, depth = hostsim_bvh(hs, leaf, builder)
    n = hs.sizes.nFaces
    assert sorted(prim.tolist()) == list(range(n))                       # every face exactly once
    assert root == 0 and 5 < depth < 32
    # Node128: words 0..23 = lox loy loz hix hiy hiz (4 children each), 24..27 = refs, 28 = count
    EMPTY = 0x7ffffffe
    c = nodes[:, 24:28].view(np.int32)
    cnt = nodes[:, 28].view(np.int32)
    assert cnt.min() >= 2 and cnt.max() == 4 and ((c != EMPTY).sum(axis=1) == cnt).all()
    assert cnt.mean() > 3.0                                              # mostly full four-wide nodes
    leaves = c[c < 0]
    first, count = (~leaves) >> 3, ((~leaves) & 7) + 1
    assert count.max() <= leaf and count.sum() == n
    covered = np.zeros(n, np.int32)
    for f, k in zip(first, count):
        covered[f:f + k] += 1
    assert (covered == 1).all()                                          # leaves partition the sorted records
    internal = c[(c >= 0) & (c != EMPTY)]
    assert len(internal) == len(nodes) - 1 and len(set(internal.tolist())) == len(internal)   # a tree
    # child boxes contain their triangles (first leaf of every node, spot check)
    boxes = nodes[:, :24].view(np.float32).reshape(len(nodes), 6, 4)       # [node, lox..hiz, child]
    p0 = tris[:, 0:3].view(np.float32); e0 = tris[:, 4:7].view(np.float32); e1 = tris[:, 8:11].view(np.float32)
    for i in range(0, len(nodes), 997):
        for side, ref in enumerate(c[i]):
            if ref >= 0:
                continue
            lo, hi = boxes[i, 0:3, side], boxes[i, 3:6, side]
            f, k = (~ref) >> 3, ((~ref) & 7) + 1
            v = np.concatenate([p0[f:f + k], p0[f:f + k] + e0[f:f + k], p0[f:f + k] - e1[f:f + k]])
            assert (v >= lo - 1e-6).all() and (v <= hi + 1e-6).all()


@pytest.mark.parametrize("scene,leaf", [("file:coffee", 4), ("file:coffee", 1), ("file:coffee", 8)])
def test_compressed_nodes_contain_the_nodes_they_stand_for(scene, leaf):
    """Node64 (what the kernels fetch): every child box, decoded the way the kernels decode it, contains the Node128 box --
    the traversal may enter more boxes than with the uncompressed nodes, never fewer -- and is at most two grid steps
    (2/255 of the node's extent) larger per side; the refs are carried over; unused children are inverted."""
    hs = M.HostScene(scene, 64, 36)
    nodes, tris, prim, root, depth, n64 = hostsim_bvh(hs, leaf, 1, want_nodes64=True)
    assert len(n64) == len(nodes) > 0
    boxes = nodes[:, :24].view(np.float32).reshape(len(nodes), 6, 4)
    refs = nodes[:, 24:28].view(np.int32)
    qb, qrefs, step = node64_boxes(n64)
    assert np.array_equal(qrefs, refs)
    used = (refs != 0x7ffffffe)[:, None, :].repeat(3, axis=1)
    lo, hi, qlo, qhi = boxes[:, 0:3], boxes[:, 3:6], qb[:, 0:3], qb[:, 3:6]
    assert (qlo[used] <= lo[used]).all() and (qhi[used] >= hi[used]).all()
    step4 = step.astype(np.float64)[:, :, None].repeat(4, axis=2)
    slack = 2.0 * step4 + 2e-6 * np.maximum(np.abs(lo), np.abs(hi)) + 1e-29
    assert ((lo - qlo)[used] <= slack[used]).all() and ((qhi - hi)[used] <= slack[used]).all()
    # the grid is as fine as it can be: 255 steps span the node's box and its margin, hardly more
    ext = (np.where(used, hi, -np.inf).max(axis=2) - np.where(used, lo, np.inf).min(axis=2)).astype(np.float64)
    assert (255.0 * step <= 1.001 * ext + 4e-6 * np.abs(boxes).max() + 1e-27).all()
    assert (qlo[~used] > qhi[~used]).all()


@pytest.mark.parametrize("scene,kw", [("file:coffee", {}), ("dining_standin", dict(iarg=1)), ("million_standin", dict(iarg=20000)),
                                      ("coffee_pot_standin", {})])
def test_walking_the_compressed_nodes_gives_the_same_image_and_rays(scene, kw):
    """The node format changes which boxes are entered, never a result: same accumulator bits, same rays, same hits;
    only the node / triangle test counts move (up: the quantised boxes are a little larger -- a lot more triangle tests only
    where rays leave large flat faces, the dining room's walls)."""
    hs = M.HostScene(scene, 96, 54, **kw)
    seeds = M.launch_seeds(2)
    a128, c128 = hostsim_render(hs, seeds, node_format=128)
    a64, c64 = hostsim_render(hs, seeds, node_format=64)
    assert np.array_equal(a128.view(np.uint32), a64.view(np.uint32))
    for k in ("samples", "primaryRays", "bounceRays", "shadowRays", "closestHits", "lightLoads"):
        assert c128[k] == c64[k], k
    assert c128["nodeFetches"] <= c64["nodeFetches"] < 1.35 * c128["nodeFetches"]
    assert c128["triTests"] <= c64["triTests"] < 1.35 * c128["triTests"]


def test_a_tree_too_wide_for_the_node_grid_has_no_compressed_form(tmp_path):
    from common import write_glass_over_opaque_scene
    hs = M.HostScene("file:cornell", 32, 24, base_folder=write_glass_over_opaque_scene(tmp_path, True, scale=1e10))
    nodes, tris, prim, root, depth, n64 = hostsim_bvh(hs, 4, 1, want_nodes64=True)
    assert len(nodes) > 0 and not n64.any()                              # nothing was written
    hs = M.HostScene("file:cornell", 32, 24, base_folder=write_glass_over_opaque_scene(tmp_path, True, scale=1e3))
    nodes, tris, prim, root, depth, n64 = hostsim_bvh(hs, 4, 1, want_nodes64=True)
    assert len(nodes) > 0 and n64.any()
    a128, _ = hostsim_render(hs, M.launch_seeds(2), node_format=128)
    a64, _ = hostsim_render(hs, M.launch_seeds(2), node_format=64)
    assert np.array_equal(a128.view(np.uint32), a64.view(np.uint32))


def test_sah_topology_needs_fewer_node_fetches_and_gives_the_same_image():
    """The binned-SAH topology (what the reference gets from "Trbvh") against the plain Morton radix tree: same bits
    (the hit is independent of the tree), at least 15 % fewer four-wide node fetches per ray on the coffee scene."""
    from common import hostsim_lib
    hs = M.HostScene("file:coffee", 160, 90)
    seeds = M.launch_seeds(2)
    try:
        hostsim_lib().hostsim_set_builder(0)
        a, ca = hostsim_render(hs, seeds)
    finally:
        hostsim_lib().hostsim_set_builder(1)
    b, cb = hostsim_render(hs, seeds)
    assert np.array_equal(a, b)
    assert cb["nodeFetches"] < 0.85 * ca["nodeFetches"] and cb["triTests"] <= ca["triTests"]


def test_leaf_size_does_not_change_the_image():
    hs = M.HostScene("file:coffee", 80, 45)
    seeds = M.launch_seeds(2)
    a, _ = hostsim_render(hs, seeds, leaf_size=1)
    b, _ = hostsim_render(hs, seeds, leaf_size=8)
    assert np.array_equal(a, b)        # equal-t rule makes the hit independent of the tree


def _write_soup_scene(root, rng, n, scale, offset):
    """n random triangles in a unit cube (some tiny, some spanning it, some axis-aligned flats), scaled and shifted, a floor
    under them and a quad light above: a stress scene for the node compression (large |coordinate| / extent ratios)."""
    d = os.path.join(str(root), "cornell")
    os.makedirs(d, exist_ok=True)
    c = rng.random((n, 1, 3)); size = 10.0 ** rng.uniform(-3, 0, (n, 1, 1))
    v = c + (rng.random((n, 3, 3)) - 0.5) * size
    flat = rng.random(n) < 0.2
    v[flat, :, rng.integers(0, 3)] = c[flat, :, 0][:, :1]          # a fifth of them flat on one axis
    v = v * scale + offset
    with open(os.path.join(d, "soup.obj"), "w") as f:
        for t in v:
            for p in t:
                f.write("v %.9g %.9g %.9g\n" % tuple(p))
        for i in range(n):
            f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    lo, hi = np.array([-0.5, -0.05, -0.5]) * scale + offset, np.array([1.5, -0.05, 1.5]) * scale + offset
    with open(os.path.join(d, "floor.obj"), "w") as f:
        f.write("v %.9g %.9g %.9g\nv %.9g %.9g %.9g\nv %.9g %.9g %.9g\nv %.9g %.9g %.9g\nf 1 3 2\nf 1 4 3\n" % (
            lo[0], lo[1], lo[2], hi[0], lo[1], lo[2], hi[0], lo[1], hi[2], lo[0], lo[1], hi[2]))
    L = np.array([[0.2, 1.6, 0.2], [0.8, 1.6, 0.2], [0.2, 1.6, 0.8]]) * scale + offset
    with open(os.path.join(d, "cornell.scene"), "w") as f:
        f.write("material Grey\n{\n    color 0.7 0.7 0.7\n    roughness 0.5\n}\nmaterial Shiny\n{\n    color 0.8 0.5 0.3\n    roughness 0.1\n    metallic 0.5\n}\n"
                "mesh\n{\n    file floor.obj\n    material Grey\n}\nmesh\n{\n    file soup.obj\n    material Shiny\n}\n"
                "light\n{\n    type Quad\n    position %.9g %.9g %.9g\n    v1 %.9g %.9g %.9g\n    v2 %.9g %.9g %.9g\n    emission 15 15 15\n}\n" % tuple(L.ravel()))
    return str(root) + "/"


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (1e-3, 0.0), (1e4, 0.0), (1.0, 1e4), (10.0, -3e5)])
def test_compressed_nodes_on_random_triangle_soups(tmp_path, scale, offset):
    """Containment and image equality of the two node formats where the grid is stressed: tiny and huge scenes, scenes far from
    the origin (few mantissa bits left inside the node), flat and needle triangles."""
    import os  # noqa: F401
    rng = np.random.default_rng(int(abs(scale) * 1000 + abs(offset)) % (2 ** 31))
    hs = M.HostScene("file:cornell", 48, 36, base_folder=_write_soup_scene(tmp_path, rng, 700, scale, offset))
    nodes, tris, prim, root, depth, n64 = hostsim_bvh(hs, 4, 1, want_nodes64=True)
    assert len(nodes) > 100 and n64.any()
    boxes = nodes[:, :24].view(np.float32).reshape(len(nodes), 6, 4)
    refs = nodes[:, 24:28].view(np.int32)
    qb, qrefs, step = node64_boxes(n64)
    used = (refs != 0x7ffffffe)[:, None, :].repeat(3, axis=1)
    assert np.array_equal(qrefs, refs)
    assert (qb[:, 0:3][used] <= boxes[:, 0:3][used]).all() and (qb[:, 3:6][used] >= boxes[:, 3:6][used]).all()
    seeds = M.launch_seeds(3)
    a128, c128 = hostsim_render(hs, seeds, node_format=128)
    a64, c64 = hostsim_render(hs, seeds, node_format=64)
    assert np.array_equal(a128.view(np.uint32), a64.view(np.uint32)) and a128.max() > 0
    assert (c128["bounceRays"], c128["shadowRays"], c128["closestHits"]) == (c64["bounceRays"], c64["shadowRays"], c64["closestHits"])


def test_host_mirror_trees_are_valid_and_the_check_sees_a_broken_box():
    """tests/common.py tree_containment_errors (used by the GPU tests and the fuzzers' tree-dependence proof) on the host mirror
    of the device builder: every child box -- 128-byte form and decoded 64-byte form -- contains the triangles below it; one shrunken box is found."""
    from common import tree_containment_errors
    hs = M.HostScene("million_standin", 64, 36, iarg=3000)
    for leaf, builder in ((1, 1), (4, 1), (8, 1), (4, 0)):
        n, t, _p, root, _depth, n64 = hostsim_bvh(hs, leaf, builder, want_nodes64=True)
        assert tree_containment_errors(n, t, root, n64) == 0, (leaf, builder)
    broken = n.copy(); broken.view(np.float32)[0, 12] -= 0.5          # hix of the root's child 0
    assert tree_containment_errors(broken, t, root) == 1
