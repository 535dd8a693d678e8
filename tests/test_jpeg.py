"""JPEG texture decoding (rf_jpeg.cpp; SURVEY.md 8(f) row 1: the real Sponza.glb carries JPEG textures).

The reference decodes with stb_image (un-vendored, absent here) => PARITY UNPINNED against stb itself.
What is pinned:
  * against Pillow (libjpeg-turbo) on every coding variant: baseline / progressive, 4:4:4 / 4:2:2 /
    4:2:0 / grey, optimised Huffman tables, restart intervals, 16-bit quantisation tables, odd sizes.
    The entropy decode + dequantisation are standardised, so a wrong coefficient shows up as a gross
    error; the IDCT / chroma up-sampling / colour conversion are stb_image's integer algorithms and
    differ from libjpeg-turbo's by rounding only (tolerances below, per variant);
  * self-consistency that does not depend on any other decoder: the progressive and the baseline
    file of the same image hold the same quantised coefficients, so the two decodes must be
    IDENTICAL; same for restart-interval and optimised-table re-encodings;
  * known answers of the post-entropy stages (flat blocks, stb's up-sampling taps, its fixed-point
    YCbCr -> RGB) computed here from the published formulas.
"""
import io
import os
import struct

import numpy as np
import pytest
from PIL import Image

import rayfinder_amd as rf


def _img(h=67, w=93, seed=1):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 9.0), 128 + 90 * np.cos(yy / 7.0), 128 + 60 * np.sin((xx + yy) / 11.0)], -1)
    return np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)


def _enc(im, mode="RGB", **kw):
    b = io.BytesIO()
    Image.fromarray(im, mode).save(b, "JPEG", **kw)
    return b.getvalue()


def _rgb(data):
    px, w, h = rf.texture_from_memory(data)
    assert ((px >> 24) == 255).all()            # texture.cpp:46: alpha forced to 255
    return np.stack([(px >> 16) & 255, (px >> 8) & 255, px & 255], -1).reshape(h, w, 3).astype(int)


def _pil(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(int)


VARIANTS = {
    # name: (save kwargs, grey?, max |d| vs Pillow, mean |d| vs Pillow)
    "444_q90": (dict(quality=90, subsampling=0), False, 3, 0.08),
    "420_q85": (dict(quality=85, subsampling=2), False, 4, 0.3),
        "422_q85": (dict(quality=85, subsampling=1), False, 10, 1.0),   # libjpeg-turbo's h2v1 filter differs (taps at the row ends, rounding)
    "grey_q80": (dict(quality=80), True, 2, 0.05),
    "prog420": (dict(quality=85, subsampling=2, progressive=True), False, 4, 0.3),
    "prog444_q95": (dict(quality=95, subsampling=0, progressive=True), False, 3, 0.08),
    "optimised_q70": (dict(quality=70, optimize=True), False, 4, 0.3),
    "q5_16bit_tables": (dict(quality=5, subsampling=2), False, 4, 0.3),
    "restart_blocks": (dict(quality=85, subsampling=2, restart_marker_blocks=3), False, 4, 0.3),
    "restart_rows_prog": (dict(quality=85, subsampling=0, restart_marker_rows=1, progressive=True), False, 3, 0.08),
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
@pytest.mark.parametrize("size", [(67, 93), (8, 8), (1, 1), (17, 16), (33, 5)])
def test_against_pillow(name, size):
    kw, grey, max_d, mean_d = VARIANTS[name]
    im = _img(*size)
    data = _enc(im[..., 0], "L", **kw) if grey else _enc(im, **kw)
    got, ref = _rgb(data), _pil(data)
    assert got.shape == ref.shape == (size[0], size[1], 3)
    d = np.abs(got - ref)
    assert d.max() <= max_d and d.mean() <= mean_d, (name, size, int(d.max()), float(d.mean()))


def test_progressive_restart_and_optimised_files_decode_identically_to_baseline():
    im = _img(75, 131, seed=3)
    for sub in (0, 1, 2):
        base = _rgb(_enc(im, quality=88, subsampling=sub))
        for kw in (dict(progressive=True), dict(optimize=True), dict(restart_marker_blocks=2), dict(restart_marker_rows=1),
                   dict(progressive=True, restart_marker_rows=2)):
            other = _rgb(_enc(im, quality=88, subsampling=sub, **kw))
            assert np.array_equal(base, other), (sub, kw)
    g = _rgb(_enc(im[..., 1], "L", quality=60))
    assert np.array_equal(g, _rgb(_enc(im[..., 1], "L", quality=60, progressive=True)))
    assert (g[..., 0] == g[..., 1]).all() and (g[..., 1] == g[..., 2]).all()


def test_flat_images_known_answer():
    """A flat image is DC-only: stb's IDCT gives clamp((dc*4*4096*... ) i.e. exactly 128 + dc_dequant/8 rounded;
    with the colour conversion of a neutral chroma (cb = cr = 128) R = G = B = Y."""
    for v in (0, 1, 77, 128, 200, 255):
        im = np.full((24, 40, 3), v, np.uint8)
        got = _rgb(_enc(im, quality=100, subsampling=2))
        assert (got == got[0, 0]).all()
        assert abs(int(got[0, 0, 0]) - v) <= 1 and (got[0, 0] == got[0, 0, 0]).all()
        assert np.array_equal(got, _pil(_enc(im, quality=100, subsampling=2)))


def test_fixed_point_colour_conversion_known_answers():
    """stb_image's YCbCr -> RGB: 20-bit fixed point, constants rounded to 1/4096, the Cb term of green
    masked to its upper 16 bits -- restated here with Python integers and checked through the decoder
    on constant-colour 4:4:4 files (flat blocks: Y / Cb / Cr are the plane values, read back through
    Pillow's no-conversion YCbCr draft mode)."""
    def fixed(x):
        return int(np.float32(x) * np.float32(4096.0) + np.float32(0.5)) << 8

    def i32(x):
        x &= 0xFFFFFFFF
        return x - (1 << 32) if x & 0x80000000 else x

    def convert(y, cb, cr):
        yf = (y << 20) + (1 << 19)
        cr -= 128
        cb -= 128
        r = i32(yf + cr * fixed(1.40200))
        g = i32(yf + cr * -fixed(0.71414) + ((cb * -fixed(0.34414)) & 0xFFFF0000))
        b = i32(yf + cb * fixed(1.77200))
        return tuple(min(max(c >> 20, 0), 255) for c in (r, g, b))

    rng = np.random.default_rng(9)
    worst = 0
    for _ in range(60):
        colour = rng.integers(0, 256, 3)
        im = np.broadcast_to(colour.astype(np.uint8), (16, 16, 3)).copy()
        data = _enc(im, quality=100, subsampling=0)
        pim = Image.open(io.BytesIO(data))
        pim.draft("YCbCr", pim.size)
        assert pim.mode == "YCbCr"
        y, cb, cr = (int(v) for v in np.asarray(pim)[8, 8])
        got = _rgb(data)[8, 8]
        want = convert(y, cb, cr)
        worst = max(worst, max(abs(int(g) - w) for g, w in zip(got, want)))
    assert worst <= 1, worst       # <= 1: the two IDCTs may round a flat plane differently by one level


def test_upsampling_taps_known_answer():
    """stb's h2v2 filter on a chroma step: interior samples are (9a + 3b + 3c + d + 8) >> 4.  A 32x32
    image whose left half is one colour and right half another, 4:2:0: rows far from the vertical
    edges see a = c, b = d, so the two columns next to the step are (3a + b + 2) >> 2 blends."""
    a, b = np.array([200, 60, 60], np.uint8), np.array([60, 60, 200], np.uint8)
    im = np.zeros((32, 32, 3), np.uint8); im[:, :16] = a; im[:, 16:] = b
    got = _rgb(_enc(im, quality=100, subsampling=2))
    ref = _pil(_enc(im, quality=100, subsampling=2))
    assert np.abs(got - ref).max() <= 3
    # symmetric about the step, monotone across it
    row = got[16, :, 0]
    assert (np.diff(row[8:24]) <= 1).all() and row[8] > row[23]


def test_jpeg_texture_through_a_glb(tmp_path):
    """image/jpeg embedded in a GLB -> PtFormat texture, as Texture::fromMemory would (texture.cpp:12-54)."""
    import json
    from oracle import gltf_ref
    im = _img(48, 64, seed=5)
    jpg = _enc(im, quality=92, subsampling=2)
    blob = bytearray()
    views = []

    def add(data):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)})
        blob.extend(data)
        return len(views) - 1

    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], "<f4"); nrm = np.array([[0, 0, 1]] * 3, "<f4"); uv = np.zeros((3, 2), "<f4")
    acc = []
    for arr, typ in ((pos, "VEC3"), (nrm, "VEC3"), (uv, "VEC2")):
        acc.append({"bufferView": add(arr.tobytes()), "componentType": 5126, "count": 3, "type": typ})
    acc.append({"bufferView": add(np.array([0, 1, 2], "<u2").tobytes()), "componentType": 5123, "count": 3, "type": "SCALAR"})
    js = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
          "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0, "mode": 4}]}],
          "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}],
          "images": [{"bufferView": add(jpg), "mimeType": "image/jpeg"}], "textures": [{"source": 0}],
          "accessors": acc, "bufferViews": views, "buffers": [{"byteLength": 0}]}
    js["buffers"][0]["byteLength"] = len(blob) + (-len(blob) % 4)
    jb = json.dumps(js).encode(); jb += b" " * (-len(jb) % 4)
    while len(blob) % 4:
        blob.append(0)
    path = os.path.join(tmp_path, "jpg.glb")
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(jb) + 8 + len(blob)))
        f.write(struct.pack("<II", len(jb), 0x4E4F534A)); f.write(jb)
        f.write(struct.pack("<II", len(blob), 0x004E4942)); f.write(bytes(blob))
    px, w, h = rf.PtFormat.from_gltf(path).texture(0)
    assert (w, h) == (64, 48)
    direct, _, _ = rf.texture_from_memory(jpg)
    assert np.array_equal(px, direct)
    opx, ow, oh = gltf_ref.decode_image_bgra(jpg)      # the oracle's ingest decodes with Pillow
    assert (ow, oh) == (w, h)
    chan = lambda p, s: ((p >> s) & 255).astype(int)
    assert max(np.abs(chan(px, s) - chan(opx, s)).max() for s in (0, 8, 16)) <= 4


def test_malformed_jpegs_fail_cleanly():
    good = _enc(_img(16, 16), quality=80)
    for bad in (good[:20], good[:len(good) // 2].replace(b"\xff\xda", b"\xff\xda", 1)[:60], b"\xff\xd8\xff\xe0\x00\x02", b"\xff\xd8\xff\xc9\x00\x0b" + bytes(9)):
        with pytest.raises(rf.RayfinderError):
            rf.texture_from_memory(bad)
    # a truncated entropy segment decodes what is there (stb_image does the same) instead of crashing
    px, w, h = rf.texture_from_memory(good[:-40] + b"\xff\xd9")
    assert (w, h) == (16, 16)
