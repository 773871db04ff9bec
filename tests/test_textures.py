"""albedo textures -- image ingest, rtTex2D restatement, textured Disney materials."""
import ctypes as C
import os

import numpy as np
import pytest

from common import M, O, hostsim_render, oracle_scene, rmse, textured_scene, test_textures as _textures, write_png

K = M._capi


def _read_image(path):
    w, h = C.c_int32(), C.c_int32()
    rc = K.host_lib().mohost_read_image(str(path).encode(), C.byref(w), C.byref(h), None, 0)
    if rc != K.MOPTIX_OK:
        raise RuntimeError(K.host_lib().mohost_last_error().decode())
    px = np.zeros((h.value, w.value, 3), np.uint8)
    rc = K.host_lib().mohost_read_image(str(path).encode(), None, None, px.ctypes.data_as(C.POINTER(C.c_uint8)), px.size)
    assert rc == K.MOPTIX_OK
    return px


def test_png_and_pnm_decoder(tmp_path):
    checker, ramp = _textures()
    for ft in (0, 1):
        write_png(tmp_path / ("c%d.png" % ft), checker, filter_type=ft)
        assert np.array_equal(_read_image(tmp_path / ("c%d.png" % ft)), checker)
    (tmp_path / "r.ppm").write_bytes(b"P6 16 16 255\n" + ramp.tobytes())
    assert np.array_equal(_read_image(tmp_path / "r.ppm"), ramp)
    (tmp_path / "g.pgm").write_bytes(b"P5\n4 2\n255\n" + bytes(range(8)))
    assert np.array_equal(_read_image(tmp_path / "g.pgm")[..., 1].reshape(-1), np.arange(8))
    (tmp_path / "bad.gif").write_bytes(b"GIF89a not really")
    with pytest.raises(RuntimeError, match="unsupported image format"):
        _read_image(tmp_path / "bad.gif")
    (tmp_path / "bad.jpg").write_bytes(b"\xff\xd8\xff\xe0 not really")
    with pytest.raises(RuntimeError):
        _read_image(tmp_path / "bad.jpg")
    with pytest.raises(RuntimeError, match="cannot open"):
        _read_image(tmp_path / "missing.png")
    (tmp_path / "trunc.png").write_bytes((tmp_path / "c0.png").read_bytes()[:120])
    with pytest.raises(RuntimeError):
        _read_image(tmp_path / "trunc.png")


def test_png_decoder_against_pil(tmp_path):
    """Every PNG colour type / bit depth / filter heuristic PIL can write, decoded by both."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(3)
    base = (rng.rand(37, 53, 3) * 255).astype(np.uint8)
    smooth = np.stack([np.add.outer(np.arange(37) * 3, np.arange(53) * 2) % 256] * 3, -1).astype(np.uint8)
    cases = []
    for name, arr in (("noise", base), ("smooth", smooth)):
        img = Image.fromarray(arr, "RGB")
        cases += [(name + "_rgb", img, {}), (name + "_rgb_opt", img, dict(optimize=True)), (name + "_rgb_c0", img, dict(compress_level=0)),
                  (name + "_rgba", img.convert("RGBA"), {}), (name + "_l", img.convert("L"), {}), (name + "_la", img.convert("LA"), {}),
                  (name + "_p", img.convert("P", palette=Image.ADAPTIVE, colors=200), {}),
                  (name + "_p16", img.convert("P", palette=Image.ADAPTIVE, colors=16), dict(bits=4)),
                  (name + "_1", img.convert("1"), {})]
    cases.append(("i16", Image.fromarray((np.add.outer(np.arange(37), np.arange(53)) * 700).astype(np.uint16)), {}))
    for name, img, kw in cases:
        p = tmp_path / (name + ".png")
        img.save(p, **kw)
        got = _read_image(p)
        back = Image.open(p)
        if back.mode == "I;16":
            want = np.stack([(np.asarray(back) >> 8).astype(np.uint8)] * 3, -1)      # png_set_strip_16: the high byte
        else:
            want = np.asarray(back.convert("RGB"))
        assert got.shape == want.shape and np.array_equal(got, want), name
    img = Image.fromarray(base, "RGB")
    img.save(tmp_path / "interlaced.png")          # a plain file whose IHDR claims Adam7: its data is too short for the seven passes
    raw = bytearray((tmp_path / "interlaced.png").read_bytes()); raw[28] = 1
    (tmp_path / "interlaced.png").write_bytes(bytes(raw))
    with pytest.raises(RuntimeError):
        _read_image(tmp_path / "interlaced.png")


def _write_adam7_png(path, arr, ctype, depth, palette=None, filt=0):
    """An Adam7-interlaced PNG (PIL reads them but cannot write them): arr is [H, W, C] (or [H, W] for grey / palette) of samples
    < 2^depth; every scanline of every pass gets filter type `filt` (0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth)."""
    import struct, zlib
    a = np.asarray(arr)
    if a.ndim == 2:
        a = a[..., None]
    H, W, Cn = a.shape
    bpp = max(1, Cn * depth // 8)
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    def paeth(x, y, z):
        pp = x + y - z; pa, pb, pc = abs(pp - x), abs(pp - y), abs(pp - z)
        return x if (pa <= pb and pa <= pc) else (y if pb <= pc else z)
    data = bytearray()
    for xs, ys, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = a[ys::dy, xs::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for row in sub:
            if depth == 16:
                line = b"".join(struct.pack(">H", int(v)) for v in row.reshape(-1))
            elif depth == 8:
                line = bytes(int(v) for v in row.reshape(-1))
            else:                                                   # packed samples, most significant first
                bits = "".join(format(int(v), "0%db" % depth) for v in row.reshape(-1))
                bits += "0" * (-len(bits) % 8)
                line = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
            pl = prev if prev is not None else bytes(len(line))
            out = bytearray()
            for i, v in enumerate(line):
                x = line[i - bpp] if i >= bpp else 0; y = pl[i]; z = pl[i - bpp] if i >= bpp else 0
                out.append((v - (0, x, y, (x + y) >> 1, paeth(x, y, z))[filt]) & 255)
            data += bytes([filt]) + bytes(out)
            prev = line
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 1))
    if palette is not None:
        png += chunk(b"PLTE", bytes(int(v) for v in np.asarray(palette).reshape(-1)))
    comp = zlib.compress(bytes(data), 6)
    png += chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b"")
    open(path, "wb").write(png)


def test_interlaced_png_against_pil(tmp_path):
    """Adam7 (QImage reads it; MinimalOptiX.cpp:447-471): seven passes with their own scanlines and filters, for every colour type
    and bit depth, at sizes that leave some passes empty -- decoded by this project's reader and by PIL, and equal to the source."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(11)
    for (h, w) in ((37, 53), (1, 1), (2, 3), (3, 2), (8, 8), (5, 1), (1, 9), (16, 17)):
        rgb = (rng.rand(h, w, 3) * 256).astype(np.uint8)
        pal = (rng.rand(16, 3) * 256).astype(np.uint8)
        cases = [("rgb8", rgb, 2, 8, None, rgb), ("rgba8", np.concatenate([rgb, rgb[..., :1]], -1), 6, 8, None, rgb),
                 ("g8", rgb[..., 0], 0, 8, None, np.stack([rgb[..., 0]] * 3, -1)),
                 ("ga8", rgb[..., :2], 4, 8, None, np.stack([rgb[..., 0]] * 3, -1)),
                 ("g4", rgb[..., 0] >> 4, 0, 4, None, np.stack([(rgb[..., 0] >> 4) * 17] * 3, -1)),
                 ("g1", rgb[..., 0] >> 7, 0, 1, None, np.stack([(rgb[..., 0] >> 7) * 255] * 3, -1)),
                 ("p4", rgb[..., 1] >> 4, 3, 4, pal, pal[rgb[..., 1] >> 4]),
                 ("p8", rgb[..., 1] >> 4, 3, 8, pal, pal[rgb[..., 1] >> 4]),
                 ("rgb16", rgb.astype(np.uint16) * 257, 2, 16, None, rgb)]
        for i, (name, arr, ctype, depth, palette, want) in enumerate(cases):
            p = tmp_path / ("a7_%s_%dx%d.png" % (name, w, h))
            _write_adam7_png(p, arr, ctype, depth, palette, filt=(i + h + w) % 5)
            got = _read_image(p)
            assert got.shape == want.shape and np.array_equal(got, want), (name, h, w)
            back = Image.open(p)
            assert back.info.get("interlace") == 1
            if name not in ("rgb16", "ga8"):                        # (PIL's own 16-bit RGB and LA conversions differ from png_set_strip_16 / grey replication)
                assert np.array_equal(np.asarray(back.convert("RGB")), want), (name, h, w)


def _tex2d(tex, u, v):
    t = O.OrcTexture()
    px = np.ascontiguousarray(tex, np.float32)
    t.height, t.width = px.shape[0], px.shape[1]
    t.rgba = px.ctypes.data_as(C.POINTER(C.c_float))
    out = (C.c_float * 4)()
    O.lib().orc_tex2d(C.byref(t), u, v, out)
    return np.array(list(out), np.float32)


def test_tex2d_known_answers():
    """rtTex2D with RT_WRAP_REPEAT / normalized coordinates / RT_FILTER_LINEAR (MinimalOptiX.cpp:449-474)."""
    rng = np.random.RandomState(5)
    tex = rng.rand(4, 8, 4).astype(np.float32)
    for j in range(4):
        for i in range(8):                                   # texel centres reproduce the texel
            assert np.array_equal(_tex2d(tex, (i + 0.5) / 8, (j + 0.5) / 4), tex[j, i])
    mid = _tex2d(tex, 2.0 / 8, 0.5 / 4)                      # halfway between texels 1 and 2 of row 0
    assert np.allclose(mid, 0.5 * (tex[0, 1] + tex[0, 2]), atol=1e-7)
    edge = _tex2d(tex, 0.0, 0.5 / 4)                         # u = 0: wraps between the last and the first column
    assert np.allclose(edge, 0.5 * (tex[0, 7] + tex[0, 0]), atol=1e-7)
    corner = _tex2d(tex, 0.0, 0.0)
    assert np.allclose(corner, 0.25 * (tex[3, 7] + tex[3, 0] + tex[0, 7] + tex[0, 0]), atol=1e-7)
    for (u, v) in ((0.3, 0.6), (0.91, 0.07)):                # repeat: integer shifts change nothing (up to u - floor(u))
        assert np.allclose(_tex2d(tex, u + 3, v - 2), _tex2d(tex, u, v), atol=2e-6)
    # weights have 8 fractional bits: 1/512 of a texel past a centre still returns the centre texel's neighbour mix 1/256 or 0
    a = _tex2d(tex, (1.5 + 1.0 / 1024) / 8, 0.5 / 4)
    assert np.array_equal(a, tex[0, 1])


def test_textured_scene_ingest(tmp_path):
    hs = textured_scene(tmp_path)
    assert hs.sizes.nTextures == 2 and hs.sizes.nMeshes == 5 and hs.sizes.nFaces == 10 and not hs.warnings
    d = hs.to_dict()
    ids = [m["albedoID"] for m in d["materials"] if m["kind"] == K.MAT_DISNEY]
    assert ids == [1, 2, 1, 0, 2]                                    # one sampler per file name (texNameSamplerMap)
    checker, ramp = _textures()
    t0, t1 = d["textures"]
    assert t0.shape == (24, 32, 4) and t1.shape == (16, 16, 4)
    # MinimalOptiX.cpp:459-472: buffer row j = image row H-1-j, channel = 8-bit value / 255, alpha 1
    assert np.allclose(t0[..., :3], checker[::-1].astype(np.float64) / 255.0, atol=1e-7) and (t0[..., 3] == 1).all()
    assert np.allclose(t1[..., :3], ramp[::-1].astype(np.float64) / 255.0, atol=1e-7)
    uv, has = hs.face_uvs()
    assert has.tolist() == [1, 1, 1, 1, 0, 0, 0, 0, 1, 1] and np.allclose(uv[0], [-0.7, -0.4, -0.7, 1.9, 2.2, 1.9])
    # a texture file that cannot be read is reported and the material keeps its colour
    os.remove(os.path.join(str(tmp_path), "cornell", "ramp.ppm"))
    hs2 = M.HostScene("file:cornell", 32, 24, base_folder=str(tmp_path) + "/")
    assert hs2.sizes.nTextures == 1 and any("ramp.ppm" in w for w in hs2.warnings)


def test_textured_scene_hostsim_matches_oracle(tmp_path):
    """The device code path (compiled for the host) against the oracle on the textured scene."""
    hs = textured_scene(tmp_path, 64, 48)
    seeds = M.launch_seeds(4, 11)
    ref, st = oracle_scene(hs).render(seeds)
    got, cnt = hostsim_render(hs, seeds)
    assert cnt["primaryRays"] == st.primaryRays and cnt["bounceRays"] == st.bounceRays and cnt["shadowRays"] == st.shadowRays
    assert rmse(got, ref) / len(seeds) < 1e-5
    # the textures matter: rendering with the constant colours instead gives a different image
    d = hs.to_dict()
    for m in d["materials"]:
        m["albedoID"] = 0
    plain, _ = O.Scene(d).render(seeds)
    assert rmse(plain, ref) / len(seeds) > 1e-2


def test_jpeg_decoder_against_libjpeg(tmp_path):
    """Baseline JPEG (what QImage reads through libjpeg): islow IDCT, fancy upsampling and the YCbCr tables are
    restated so that the pixels equal libjpeg-turbo's (via PIL) for 4:4:4 / 4:2:2 / 4:2:0, grayscale, optimised
    Huffman tables, restart intervals and odd sizes."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(1)
    yy, xx = np.mgrid[0:61, 0:83]
    smooth = np.stack([(xx * 3 + yy) % 256, (yy * 4) % 256, (xx * 2 + yy * 2) % 256], -1).astype(np.uint8)
    noise = (rng.rand(61, 83, 3) * 255).astype(np.uint8)
    p = tmp_path / "t.jpg"
    for name, arr in (("smooth", smooth), ("noise", noise), ("tiny", smooth[:1, :1]), ("thin", smooth[:17, :3]), ("m16", smooth[:32, :48])):
        for sub in (0, 1, 2):
            for q, extra in ((30, {}), (75, {"optimize": True}), (95, {"restart_marker_blocks": 3})):
                Image.fromarray(arr, "RGB").save(p, quality=q, subsampling=sub, **extra)
                assert np.array_equal(_read_image(p), np.asarray(Image.open(p).convert("RGB"))), (name, sub, q, extra)
        Image.fromarray(arr, "RGB").convert("L").save(p, quality=80)
        assert np.array_equal(_read_image(p), np.asarray(Image.open(p).convert("RGB"))), (name, "gray")
    # progressive files (SOF2: spectral selection + successive approximation, DC and AC refinement scans, end-of-band runs)
    for name, arr in (("smooth", smooth), ("noise", noise), ("tiny", smooth[:1, :1]), ("thin", smooth[:17, :3]), ("m16", smooth[:32, :48]), ("wide", noise[:9, :80])):
        for sub in (0, 1, 2):
            for q, extra in ((20, {}), (60, {"optimize": True}), (92, {"restart_marker_blocks": 2}), (100, {})):
                Image.fromarray(arr, "RGB").save(p, quality=q, subsampling=sub, progressive=True, **extra)
                assert b"\xff\xc2" in p.read_bytes()[:1200]
                assert np.array_equal(_read_image(p), np.asarray(Image.open(p).convert("RGB"))), ("progressive", name, sub, q, extra)
        Image.fromarray(arr, "RGB").convert("L").save(p, quality=70, progressive=True)
        assert np.array_equal(_read_image(p), np.asarray(Image.open(p).convert("RGB"))), ("progressive", name, "gray")
    Image.fromarray(smooth, "RGB").save(p, progressive=True)
    raw = p.read_bytes()
    (tmp_path / "cut.jpg").write_bytes(raw[:40])
    with pytest.raises(RuntimeError):
        _read_image(tmp_path / "cut.jpg")
