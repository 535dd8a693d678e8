"""CPU tests of the product's host core through the C ABI (no GPU needed): every declared symbol
is exported; BVH builder, glTF/PNG ingest, .pt reader/writer, camera and sky agree bit for bit
with the oracle / the reference's golden vectors; error behaviour matches the reference."""
import ctypes as C
import hashlib
import io
import json
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

import rayfinder_amd as rf
from conftest import DUCK, GOLDEN, ROOT, bits
from oracle import gltf_ref, orc


# ---------------------------------------------------------------- the C ABI itself
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rayfinder_amd.h")).read()
    declared = set(re.findall(r"RF_API\s+[\w\s\*]+?\b(rf_\w+)\s*\(", header))
    assert len(declared) >= 40
    assert declared == set(rf._ffi.SIGNATURES), declared ^ set(rf._ffi.SIGNATURES)
    lib = C.CDLL(rf._ffi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", rf._ffi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (rf_\w+)", out))
    assert declared <= exported
    assert rf.version().startswith("rayfinder_amd")


def test_no_oracle_in_product():
    # the product must not link, include or import anything under oracle/
    out = subprocess.check_output(["ldd", rf._ffi.LIB_PATH]).decode()
    assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rayfinder_amd")):
        if "build" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("test oracle", ""), os.path.join(dirpath, f)


def test_renderer_create_fails_loudly_without_gpu(duck_pt):
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    params = rf.make_render_parameters(64, 64, rf.fly_camera(64, 64), 4, 2)
    with pytest.raises(rf.RayfinderError) as e:
        rf.ReferencePathTracer(params, duck_pt.scene())
    assert e.value.status == rf._ffi.RF_ERROR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_invalid_arguments_are_reported():
    assert rf.lib.rf_renderer_create(None, None, None) == rf._ffi.RF_ERROR_INVALID_ARGUMENT
    assert rf.lib.rf_pt_format_load(None, None) == rf._ffi.RF_ERROR_INVALID_ARGUMENT
    with pytest.raises(rf.RayfinderError):
        rf.build_bvh(np.zeros((0, 9), np.float32))      # the reference asserts !triangles.empty()
    with pytest.raises(rf.RayfinderError):
        rf.tiles_for_rank(64, 64, 2, 2)


# ---------------------------------------------------------------- BVH builder
def test_build_bvh_matches_oracle_on_duck(duck_oracle):
    nodes, idx, depth = rf.build_bvh(duck_oracle.P)
    assert nodes.tobytes() == duck_oracle.nodes.tobytes()
    assert np.array_equal(idx, duck_oracle.idx) and depth == duck_oracle.depth
    g = np.load(os.path.join(GOLDEN, "duck_golden.npz"))
    assert hashlib.sha256(nodes.tobytes()).hexdigest() == str(g["nodes_sha256"])


@pytest.mark.parametrize("seed,n", [(1, 1), (2, 2), (3, 3), (4, 17), (5, 300), (6, 5000)])
def test_build_bvh_matches_oracle_on_random_soups(seed, n):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-10, 10, (n, 1, 3))
    P = (c + rng.normal(0, 0.3, (n, 3, 3))).astype(np.float32).reshape(n, 9)
    a, ai, ad = rf.build_bvh(P)
    b, bi, bd = orc.build_bvh(P)
    assert a.tobytes() == b.tobytes() and np.array_equal(ai, bi) and ad == bd
    assert sorted(ai.tolist()) == list(range(n))          # a permutation
    leaves = a[a["triangleCount"] > 0]
    assert int(leaves["triangleCount"].sum()) == n


def test_build_bvh_degenerate_inputs():
    # 300 identical triangles: centroid extent is degenerate -> ONE leaf of 300 (bvh.cpp:111-121),
    # even though 300 > maxTrianglesInNode
    tri = np.array([0, 0, 0, 1, 0, 0, 0, 1, 0], np.float32)
    P = np.tile(tri, (300, 1))
    a, ai, _ = rf.build_bvh(P)
    b, bi, _ = orc.build_bvh(P)
    assert len(a) == 1 and a[0]["triangleCount"] == 300 and a.tobytes() == b.tobytes()
    # zero-area node box (all vertices on one point)
    P = np.zeros((5, 9), np.float32)
    a, _, _ = rf.build_bvh(P)
    assert len(a) == 1 and a[0]["triangleCount"] == 5
    # 400 triangles on a line: forced splits above 255 primitives
    P = np.zeros((400, 9), np.float32)
    P[:, 0] = P[:, 3] = P[:, 6] = np.arange(400)
    P[:, 4] = 1; P[:, 8] = 1
    a, ai, _ = rf.build_bvh(P)
    b, bi, _ = orc.build_bvh(P)
    assert a.tobytes() == b.tobytes() and np.array_equal(ai, bi)
    assert a[a["triangleCount"] > 0]["triangleCount"].max() <= 255


# ---------------------------------------------------------------- glTF ingest + .pt
def test_duck_pt_matches_oracle_ingest(duck_pt, duck_oracle):
    a = duck_pt.arrays()
    d = duck_oracle
    assert a["bvhNodes"].tobytes() == d.nodes.tobytes()
    assert np.array_equal(bits(a["bvhPositionAttributes"]), bits(d.tris36))
    assert np.array_equal(bits(a["trianglePositionAttributes"]), bits(d.pos48))
    assert np.array_equal(bits(a["triangleVertexAttributes"]), bits(d.attr80))
    px, w, h = a["baseColorTextures"][0]
    assert (w, h) == (512, 512) and np.array_equal(px, d.texels)      # own PNG decoder == PIL
    g = np.load(os.path.join(GOLDEN, "duck_golden.npz"))
    assert hashlib.sha256(px.tobytes()).hexdigest() == str(g["tex_sha256"])
    # raster-mesh arrays (pt_format.cpp:85-148)
    m = d.model["meshes"][0]
    assert np.array_equal(bits(a["vertexPositions"][:, :3]), bits(m["positions"])) and (a["vertexPositions"][:, 3] == 1).all()
    assert np.array_equal(bits(a["vertexNormals"][:, :3]), bits(m["normals"])) and (a["vertexNormals"][:, 3] == 0).all()
    assert np.array_equal(a["vertexIndices"], m["indices"])
    assert a["modelVertexPositions"].tolist() == [[0, 2399]] and a["modelVertexIndices"].tolist() == [[0, 12636]]
    assert a["modelBaseColorTextureIndices"].tolist() == [0]
    # substitute pins from SURVEY 8(c): accessor min/max * 0.01 == root AABB
    js, _ = gltf_ref.load_container(DUCK)
    acc = js["accessors"][2]
    assert np.allclose(a["bvhNodes"][0]["min"], np.float32(0.01) * np.array(acc["min"], np.float32), rtol=1e-6)
    assert np.allclose(a["bvhNodes"][0]["max"], np.float32(0.01) * np.array(acc["max"], np.float32), rtol=1e-6)


def test_pt_format_round_trip_and_size(duck_pt, tmp_path):
    data = duck_pt.serialize()
    assert len(data) == 2288437 and data[:9] == b"PTFORMAT3"
    again = rf.PtFormat.deserialize(data)
    assert again.serialize() == data
    a, b = duck_pt.arrays(), again.arrays()
    for k in a:
        if k == "baseColorTextures":
            assert all(np.array_equal(x[0], y[0]) and x[1:] == y[1:] for x, y in zip(a[k], b[k]))
        else:
            assert a[k].tobytes() == b[k].tobytes(), k       # memcmp-equal, as tests/pt_format.cpp:18-176
    p = tmp_path / "Duck.pt"
    duck_pt.save(p)
    assert open(p, "rb").read() == data
    assert rf.PtFormat.load(p).serialize() == data


def test_pt_format_layout_is_the_documented_one(duck_pt):
    """Independent parse of the byte stream per pt_format.cpp:238-321."""
    data = duck_pt.serialize()
    off = 9
    sizes = [48, 36, 48, 80, 16, 16, 8, 4]
    counts = []
    for s in sizes:
        n = struct.unpack_from("<Q", data, off)[0]; off += 8 + n * s; counts.append(n)
    assert counts == [8383, 4212, 4212, 4212, 2399, 2399, 2399, 12636]
    for _ in range(4):
        n = struct.unpack_from("<Q", data, off)[0]; off += 8 + 16 * n
    n = struct.unpack_from("<Q", data, off)[0]; off += 8 + 4 * n
    ntex = struct.unpack_from("<Q", data, off)[0]; off += 8
    assert ntex == 1
    w, h, npx = struct.unpack_from("<IIQ", data, off); off += 16 + 4 * npx
    assert (w, h, npx) == (512, 512, 512 * 512) and off == len(data)


def test_pt_format_magic_errors_match_reference():
    with pytest.raises(rf.RayfinderError) as e:
        rf.PtFormat.deserialize(b"PTFORMAT0")
    assert str(e.value) == ("Mismatching PtFormat file version. Invalid version in magic bytes: expected "
                            "'PTFORMAT3', got 'PTFORMAT0'.")
    with pytest.raises(rf.RayfinderError) as e:
        rf.PtFormat.deserialize(b"INVALID  ")
    assert str(e.value) == "Invalid file format: expected PtFormat file."
    with pytest.raises(rf.RayfinderError):
        rf.PtFormat.deserialize(b"PTFORMAT3" + b"\x05\x00\x00\x00\x00\x00\x00\x00")   # truncated
    with pytest.raises(rf.RayfinderError):
        rf.PtFormat.load("/nonexistent/file.pt")
    with pytest.raises(rf.RayfinderError) as e:
        rf.PtFormat.from_gltf("/nonexistent/file.glb")
    assert "does not exist" in str(e.value)


def _make_glb(path, nodes, meshes_prims, materials, images=(), textures=(), samplers=()):
    """Tiny GLB writer for ingest tests. meshes_prims: list of lists of (pos, nrm, uv, idx, material)."""
    blob = bytearray()
    views, accessors, meshes = [], [], []

    def add(data, target=None):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)})
        blob.extend(data)
        return len(views) - 1

    for prims in meshes_prims:
        plist = []
        for (pos, nrm, uv, idx, mat) in prims:
            accs = {}
            for name, arr, typ in (("POSITION", pos, "VEC3"), ("NORMAL", nrm, "VEC3"), ("TEXCOORD_0", uv, "VEC2")):
                v = add(np.asarray(arr, "<f4").tobytes())
                accessors.append({"bufferView": v, "componentType": 5126, "count": len(arr), "type": typ})
                accs[name] = len(accessors) - 1
            idx = np.asarray(idx)
            ct = {np.dtype("uint8"): 5121, np.dtype("uint16"): 5123, np.dtype("uint32"): 5125}[idx.dtype]
            v = add(idx.tobytes())
            accessors.append({"bufferView": v, "componentType": ct, "count": idx.size, "type": "SCALAR"})
            plist.append({"attributes": accs, "indices": len(accessors) - 1, "material": mat, "mode": 4})
        meshes.append({"primitives": plist})
    imgs = []
    for png in images:
        v = add(png)
        imgs.append({"bufferView": v, "mimeType": "image/png"})
    js = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": nodes, "meshes": meshes,
          "materials": materials, "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(blob)}]}
    if imgs:
        js.update(images=imgs, textures=list(textures), samplers=list(samplers))
    jb = json.dumps(js).encode()
    jb += b" " * (-len(jb) % 4)
    while len(blob) % 4:
        blob.append(0)
    total = 12 + 8 + len(jb) + 8 + len(blob)
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, total))
        f.write(struct.pack("<II", len(jb), 0x4E4F534A)); f.write(jb)
        f.write(struct.pack("<II", len(blob), 0x004E4942)); f.write(bytes(blob))


def _png(arr, mode, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr, mode).save(buf, "PNG", **kw)
    return buf.getvalue()


def test_gltf_ingest_transforms_materials_and_png_variants(tmp_path):
    rng = np.random.default_rng(11)

    def prim(nv, mat, dtype):
        pos = rng.uniform(-1, 1, (nv, 3)); nrm = rng.normal(size=(nv, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        uv = rng.uniform(-2, 2, (nv, 2)); idx = rng.integers(0, nv, 3 * 7).astype(dtype)
        return (pos, nrm, uv, idx, mat)

    rgb = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    rgba = rng.integers(0, 256, (4, 6, 4), dtype=np.uint8)
    gray = rng.integers(0, 256, (5, 3), dtype=np.uint8)
    pal = rng.integers(0, 256, (16, 16), dtype=np.uint8)
    from PIL import Image
    pim = Image.fromarray(pal, "P"); pim.putpalette([int(x) for x in rng.integers(0, 256, 768)])
    buf = io.BytesIO(); pim.save(buf, "PNG"); pal_png = buf.getvalue()
    images = [_png(rgb, "RGB"), _png(rgba, "RGBA"), _png(gray, "L"), pal_png]
    q = np.array([0.1, 0.7, -0.2, 0.67]); q /= np.linalg.norm(q)
    nodes = [
        {"children": [1, 2, 3], "translation": [1.0, 2.0, 3.0], "rotation": q.tolist(), "scale": [2.0, 0.5, 1.5]},
        {"mesh": 0, "matrix": [0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 2, 0, 0.5, 0.25, -4, 1]},
        {"mesh": 1, "scale": [0.01, 0.01, 0.01]},
        {"mesh": 2, "children": [4], "translation": [0, -1, 0]},
        {"mesh": 3, "rotation": [0.5, 0.5, 0.5, 0.5]},
    ]
    materials = [
        {"pbrMetallicRoughness": {"baseColorTexture": {"index": 3}}},     # palette
        {"pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 0.6, 1.0]}},
        {"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}},
        {"pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 0.6, 1.0]}},  # same factor -> dedup
        {"pbrMetallicRoughness": {"baseColorTexture": {"index": 1}}},
        {"pbrMetallicRoughness": {"baseColorTexture": {"index": 2}}},
        {"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}},     # same image -> dedup
    ]
    meshes = [[prim(9, 0, np.uint16), prim(5, 1, np.uint8)], [prim(12, 2, np.uint32)], [prim(6, 3, np.uint16), prim(7, 4, np.uint16)],
              [prim(8, 5, np.uint16), prim(4, 6, np.uint8)]]
    textures = [{"source": i, "sampler": 0} for i in range(4)]
    p = tmp_path / "multi.glb"
    _make_glb(p, nodes, meshes, materials, images, textures, [{"wrapS": 10497, "wrapT": 10497}])

    pt = rf.PtFormat.from_gltf(p)
    a = pt.arrays()
    m = gltf_ref.load_model(str(p))
    P, N, T, I = gltf_ref.flatten(m)
    onodes, oidx, _ = orc.build_bvh(P)
    assert a["bvhNodes"].tobytes() == onodes.tobytes()
    pa, va = gltf_ref.gpu_layout(orc.reorder(P, oidx), orc.reorder(N, oidx), orc.reorder(T, oidx), orc.reorder(I, oidx))
    assert np.array_equal(bits(a["trianglePositionAttributes"]), bits(pa))
    assert np.array_equal(bits(a["triangleVertexAttributes"]), bits(va))
    assert len(a["baseColorTextures"]) == len(m["textures"]) == 5         # 4 images + 1 factor colour
    for (px, w, h), (opx, ow, oh) in zip(a["baseColorTextures"], m["textures"]):
        assert (w, h) == (ow, oh) and np.array_equal(px, opx)
    assert a["modelBaseColorTextureIndices"].tolist() == sorted(a["modelBaseColorTextureIndices"].tolist())  # sorted by texture
    # fromPixel truncation (texture.cpp:56-65): 0.2*255 = 51, 0.4*255 = 102, 0.6*255 = 153
    factor = [t for t in a["baseColorTextures"] if t[1:] == (1, 1)][0][0][0]
    assert factor == (153 | (102 << 8) | (51 << 16) | (255 << 24))


def test_png_decoder_interlaced_16bit_and_low_depth():
    rng = np.random.default_rng(5)
    from PIL import Image
    cases = []
    rgb = rng.integers(0, 256, (13, 11, 3), dtype=np.uint8)
    buf = io.BytesIO(); Image.fromarray(rgb, "RGB").save(buf, "PNG", interlace=1) if False else None
    cases.append((_png(rgb, "RGB"), rgb))
    g16 = rng.integers(0, 65536, (7, 9), dtype=np.uint16)
    b = io.BytesIO(); Image.fromarray(g16, "I;16").save(b, "PNG"); cases.append((b.getvalue(), None))
    bw = (rng.integers(0, 2, (10, 10)) * 255).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(bw, "L").convert("1").save(b, "PNG"); cases.append((b.getvalue(), None))
    la = rng.integers(0, 256, (6, 6, 2), dtype=np.uint8)
    cases.append((_png(la, "LA"), None))
    for png, _ in cases:
        # go through a GLB so the product decodes it
        want = gltf_ref.decode_image_bgra(png)
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            tri = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], float), np.array([[0, 0, 1]] * 3, float), np.zeros((3, 2)), np.array([0, 1, 2], np.uint8), 0)
            path = os.path.join(td, "t.glb")
            _make_glb(path, [{"mesh": 0}], [[tri]], [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}], [png], [{"source": 0}], [])
            px, w, h = rf.PtFormat.from_gltf(path).texture(0)
        if w * h and want is not None:
            opx, ow, oh = want
            assert (w, h) == (ow, oh)
            if png is cases[1][0]:
                # 16-bit: stb keeps the high byte; PIL's I;16 -> RGBA conversion clips instead.  Check against the spec'd rule.
                hi = (g16 >> 8).astype(np.uint32).reshape(-1)
                assert np.array_equal(px, hi | (hi << 8) | (hi << 16) | np.uint32(255 << 24))
            else:
                assert np.array_equal(px, opx)


def test_cli_pt_format_tool_writes_identical_bytes(duck_pt, tmp_path):
    import shutil
    tool = os.path.join(ROOT, "rayfinder_amd", "bin", "rf-pt-format-tool")
    src = tmp_path / "Duck.glb"
    shutil.copy(DUCK, src)
    out = subprocess.check_output([tool, str(src)]).decode()
    assert "8383 nodes" in out
    assert open(tmp_path / "Duck.pt", "rb").read() == duck_pt.serialize()
    assert subprocess.call([tool, str(tmp_path / "missing.glb")], stderr=subprocess.DEVNULL) == 1
    assert b"Usage" in subprocess.check_output([tool])


# ---------------------------------------------------------------- camera / sky
def test_cameras_match_oracle(duck_oracle):
    for (w, h) in [(800, 600), (1920, 1080), (256, 256)]:
        assert np.array_equal(bits(rf.camera_to_array(rf.fly_camera(w, h))), bits(orc.default_pt_camera(w, h)))
        aspect = np.float32(np.float32(w) / np.float32(h))
        assert np.array_equal(bits(rf.camera_to_array(rf.bvh_visualizer_camera(duck_oracle.nodes, aspect))),
                              bits(orc.bvh_visualizer_camera(duck_oracle.nodes, aspect)))
    a = rf.camera_to_array(rf.create_camera([1, 2, 3], [0, 0.5, -1], 0.3, 4.0, orc.degrees_to_radians(55.0), 1.5))
    b = orc.create_camera([1, 2, 3], [0, 0.5, -1], 0.3, 4.0, orc.degrees_to_radians(55.0), 1.5)
    assert np.array_equal(bits(a), bits(b)) and a[18] == np.float32(0.15)


def test_sky_matches_reference_vectors():
    from test_oracle_pins import _check_sky
    _check_sky(rf.sky_state_new, rf.sky_state_radiance)
    assert np.array_equal(bits(rf.aligned_sky_state(rf.make_sky())), bits(orc.aligned_sky_state()))
    s = rf.make_sky(3.5, (0.2, 0.5, 0.9), 71.0, 213.0)
    assert np.array_equal(bits(rf.aligned_sky_state(s)), bits(orc.aligned_sky_state(3.5, (0.2, 0.5, 0.9), 71.0, 213.0)))
    with pytest.raises(rf.RayfinderError) as e:
        rf.aligned_sky_state(rf.make_sky(turbidity=11.0))
    assert e.value.status == rf._ffi.RF_ERROR_OUT_OF_RANGE


# ---------------------------------------------------------------- tiles
@pytest.mark.parametrize("w,h,world", [(1920, 1080, 1), (1920, 1080, 8), (800, 600, 2), (100, 70, 3), (31, 33, 4)])
def test_tile_assignment_is_a_balanced_partition(w, h, world):
    tx, ty = (w + 31) // 32, (h + 31) // 32
    all_tiles = [rf.tiles_for_rank(w, h, r, world) for r in range(world)]
    flat = np.concatenate(all_tiles)
    assert sorted(flat.tolist()) == list(range(tx * ty))
    sizes = [len(t) for t in all_tiles]
    assert max(sizes) - min(sizes) <= 1
    for t in all_tiles:
        assert (np.diff(t.astype(np.int64)) > 0).all()
    if world == 1:
        assert all_tiles[0].tolist() == list(range(tx * ty))


def test_untile_inverts_the_tile_major_layout():
    w, h = 100, 70
    tiles = rf.tiles_for_rank(w, h, 1, 3)
    compact = np.zeros((len(tiles) * 1024, 4), np.float32)
    tx = (w + 31) // 32
    expect = np.zeros((h, w, 4), np.float32)
    for t, tid in enumerate(tiles):
        for k in range(1024):
            block, lane = k >> 6, k & 63
            x = (tid % tx) * 32 + (block & 3) * 8 + (lane & 7)
            y = (tid // tx) * 32 + (block >> 2) * 8 + (lane >> 3)
            compact[t * 1024 + k] = (x, y, tid, 1)
            if x < w and y < h:
                expect[y, x] = (x, y, tid, 1)
    img = rf.untile(compact, tiles, w, h)
    assert np.array_equal(img, expect)


# ---------------------------------------------------------------- untrusted input (round-1 advisor findings)
def test_build_bvh_without_any_finite_sah_cost_terminates_as_one_leaf():
    """Coordinates around 1e20: every surface area overflows, no split cost passes `<`; the reference asserts
    (bvh.cpp:215-216) and its release build would recurse forever.  Documented choice: the node becomes a leaf."""
    rng = np.random.default_rng(3)
    P = (rng.uniform(-1, 1, (600, 9)) * 1e20).astype(np.float32)
    a, ai, _ = rf.build_bvh(P)
    b, bi, _ = orc.build_bvh(P)
    assert len(a) == 1 and a[0]["triangleCount"] == 600
    assert a.tobytes() == np.ascontiguousarray(b).tobytes() and np.array_equal(ai, bi)
    # mixed: a sane cluster plus far-away giants still builds and matches
    P2 = np.concatenate([rng.uniform(-1, 1, (500, 9)), rng.uniform(-1, 1, (300, 9)) * 3e19]).astype(np.float32)
    a, ai, _ = rf.build_bvh(P2)
    b, bi, _ = orc.build_bvh(P2)
    assert a.tobytes() == np.ascontiguousarray(b).tobytes() and np.array_equal(ai, bi)


def _duck_offsets(data):
    n_nodes = struct.unpack_from("<Q", data, 9)[0]
    return n_nodes, 17


@pytest.mark.parametrize("what", ["self_link", "backward_link", "link_past_end", "leaf_range", "split_axis", "texture_index", "texture_size", "slice_wrap"])
def test_corrupted_pt_is_rejected_before_it_reaches_the_device(duck_pt, what):
    data = bytearray(duck_pt.serialize())
    n_nodes, at = _duck_offsets(data)
    nodes = np.frombuffer(bytes(data[at:at + 48 * n_nodes]), np.uint32).reshape(-1, 12)
    interior = int(np.nonzero(nodes[:, 10] == 0)[0][5]); leaf = int(np.nonzero(nodes[:, 10] > 0)[0][5])
    f = lambda k: at + 48 * k + 32   # trianglesOffset, secondChildOffset, triangleCount, splitAxis
    if what == "self_link":
        struct.pack_into("<I", data, f(interior) + 4, interior)
    elif what == "backward_link":
        struct.pack_into("<I", data, f(interior) + 4, interior + 1)     # second child == first child
    elif what == "link_past_end":
        struct.pack_into("<I", data, f(interior) + 4, n_nodes)
    elif what == "leaf_range":
        struct.pack_into("<I", data, f(leaf), 4212)                      # offset == numTriangles, count >= 1
    elif what == "split_axis":
        struct.pack_into("<I", data, f(interior) + 12, 7)
    elif what == "texture_index":
        va_at = at + 48 * n_nodes + 8 + 36 * 4212 + 8 + 48 * 4212 + 8
        struct.pack_into("<I", data, va_at + 80 * 100 + 72, 3)
    elif what == "texture_size":
        struct.pack_into("<II", data, len(data) - 4 * 512 * 512 - 16, 4096, 4096)   # declares 16 Mi texels, holds 256 Ki
    else:
        # first slice table follows the 8 arrays: offset = 2^64 - 1, count = 2 wrapped past the old check
        off = 9
        for s in [48, 36, 48, 80, 16, 16, 8, 4]:
            off += 8 + struct.unpack_from("<Q", data, off)[0] * s
        assert struct.unpack_from("<Q", data, off)[0] >= 1
        struct.pack_into("<QQ", data, off + 8, 2**64 - 1, 2)
    with pytest.raises(rf.RayfinderError):
        rf.PtFormat.deserialize(bytes(data))


def test_png_header_checks():
    """IHDR must be 13 bytes and the (colour type, bit depth) pair one the PNG specification allows."""
    import zlib
    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
    def png(depth, ctype, ihdr_len=13):
        ihdr = struct.pack(">IIBBBBB", 2, 2, depth, ctype, 0, 0, 0)[:ihdr_len]
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
    for depth, ctype, n in [(3, 0, 13), (32, 0, 13), (200, 2, 13), (4, 2, 13), (16, 3, 13), (8, 6, 9), (8, 5, 13)]:
        with pytest.raises(rf.RayfinderError):
            rf.texture_from_memory(png(depth, ctype, n))


# ---------------------------------------------------------------- a Sponza-shaped asset (round 2)
def _courtyard(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_test_asset
    return make_test_asset.write_courtyard(str(tmp_path / "courtyard"))


def test_multi_material_gltf_with_external_uris_and_jpegs_bakes_like_the_independent_ingest(tmp_path):
    """tools/make_test_asset.py: .gltf + external .bin, external PNG (RGB / RGBA / palette) and JPEG (baseline 4:2:0,
    progressive 4:4:4, grey) images -- one URI percent-encoded --, a three-level node hierarchy with T/R/S and a raw
    matrix, six meshes, ten materials (image-backed with shared images, factor-only with a repeated factor), u8 / u16 /
    u32 indices (gltf_model.cpp:74-121,266-465).  Product C++ ingest == oracle/gltf_ref.py (numpy + PIL) bit for bit on
    every geometry array and PNG texel; JPEG texels within 2 levels of libjpeg-turbo (stb's IDCT differs)."""
    path = _courtyard(tmp_path)
    pt = rf.PtFormat.from_gltf(path)
    a = pt.arrays()
    m = gltf_ref.load_model(path)
    P, N, T, I = gltf_ref.flatten(m)
    assert len(P) == len(a["bvhPositionAttributes"]) > 2500 and len(m["textures"]) == len(a["baseColorTextures"]) == 8
    nodes, idx, _ = orc.build_bvh(P)
    assert nodes.tobytes() == a["bvhNodes"].tobytes()
    tris = orc.reorder(P, idx)
    assert np.array_equal(bits(tris), bits(a["bvhPositionAttributes"]))
    pos48, attr80 = gltf_ref.gpu_layout(tris, orc.reorder(N, idx), orc.reorder(T, idx), orc.reorder(I, idx))
    assert np.array_equal(bits(pos48), bits(a["trianglePositionAttributes"]))
    assert np.array_equal(np.ascontiguousarray(attr80).view(np.uint32).reshape(-1, 20), a["triangleVertexAttributes"].view(np.uint32))
    tex_idx = a["triangleVertexAttributes"].view(np.uint32)[:, 18]
    assert set(tex_idx) == set(range(8))                       # every texture is used by some triangle
    exact = 0
    for (px, w, h), (opx, ow, oh) in zip(a["baseColorTextures"], m["textures"]):
        assert (w, h) == (ow, oh)
        ch = lambda p: ((np.asarray(p)[:, None] >> np.array([0, 8, 16, 24])) & 255).astype(int)
        d = np.abs(ch(px) - ch(opx)).max()
        assert d <= 2
        exact += d == 0
    assert exact >= 5                                          # three PNGs + two factor textures
    # the raster-mesh arrays of pt_format.cpp:85-148: one slice per primitive, sorted by texture index
    assert len(a["modelBaseColorTextureIndices"]) == 10 and (np.diff(a["modelBaseColorTextureIndices"].astype(int)) >= 0).all()
    # CLI bake (host builder) == library bake, and a .pt round trip
    tool = os.path.join(ROOT, "rayfinder_amd", "bin", "rf-pt-format-tool")
    out = subprocess.check_output([tool, path]).decode()
    assert f"{len(nodes)} nodes" in out and "8 textures" in out
    data = open(path.replace(".gltf", ".pt"), "rb").read()
    assert data == pt.serialize() and rf.PtFormat.deserialize(data).serialize() == data


# ------------------------------------------------------------------ round 2: the render path's record layouts (host side)
@pytest.mark.parametrize("seed,n", [(1, 1), (2, 2), (3, 7), (4, 300), (5, 5000)])
def test_wide_record_layouts_decode_to_the_same_planes(seed, n):
    """The compact-capable records (the node's own x planes carried by the lane) and the 32-byte records (all six planes carried)
    are re-encodings of the 64-byte "children in the parent" records: rf_check_wide_layouts rebuilds all three as
    rf_renderer_create does and checks, on the host, that every variant decodes to the same twelve child planes and the same
    child words, and that what a lane carries (the node's own planes) is the union of the children's."""
    rng = np.random.default_rng(seed)
    tris = (rng.uniform(-3, 3, (n, 1, 3)) + rng.normal(0, 0.3, (n, 3, 3))).astype(np.float32)
    tris[: n // 4] = np.round(tris[: n // 4] * 2) / 2               # lattice coordinates: shared planes, zero-thickness boxes
    nodes, _, _ = rf.build_bvh(tris.reshape(n, 9))
    got = rf.check_wide_layouts(nodes)
    assert got == {"regular": True, "compact": n > 1, "hot": n > 1, "quad": n > 1, "quad_half": n > 1, "quad_local": n > 1, "oct": n > 1}   # (a single-leaf tree has no interior record)


def test_wide_record_layouts_of_duck_and_of_a_tree_whose_boxes_are_not_unions(duck_oracle):
    assert rf.check_wide_layouts(duck_oracle.nodes) == {"regular": True, "compact": True, "hot": True, "quad": True, "quad_half": True, "quad_local": True, "oct": True}
    # a hand-made tree whose root box is LARGER than the union of its children: the lane cannot carry the node's planes,
    # so the renderer falls back to the plain records
    nodes = np.zeros(3, dtype=rf.NODE_DTYPE)
    nodes[0]["min"] = (-5, -5, -5); nodes[0]["max"] = (5, 5, 5); nodes[0]["secondChildOffset"] = 2; nodes[0]["splitAxis"] = 0
    for i, x in ((1, -1.0), (2, 1.0)):
        nodes[i]["min"] = (x - 0.5, -0.5, -0.5); nodes[i]["max"] = (x + 0.5, 0.5, 0.5)
        nodes[i]["trianglesOffset"] = i - 1; nodes[i]["triangleCount"] = 1; nodes[i]["splitAxis"] = 0xFFFFFFFF
    # (the quad records skip the level BELOW the node they belong to; here that level holds only leaves, so nothing is skipped)
    assert rf.check_wide_layouts(nodes) == {"regular": True, "compact": False, "hot": False, "quad": True, "quad_half": True, "quad_local": True, "oct": True}
    # ... but a CHILD whose box is larger than the union of its children cannot be skipped: "a grandchild passes" would no
    # longer imply "the child passes" with the same planes
    deep = np.zeros(5, dtype=rf.NODE_DTYPE)
    deep[0]["min"] = (-6, -6, -6); deep[0]["max"] = (6, 6, 6); deep[0]["secondChildOffset"] = 4; deep[0]["splitAxis"] = 0
    deep[1]["min"] = (-6, -6, -6); deep[1]["max"] = (0, 6, 6); deep[1]["secondChildOffset"] = 3; deep[1]["splitAxis"] = 1
    for i, (lo, hi) in ((2, ((-2, -2, -1), (-1, -1, 1))), (3, ((-2, 1, -1), (-1, 2, 1))), (4, ((1, -1, -1), (6, 6, 6)))):
        deep[i]["min"] = lo; deep[i]["max"] = hi; deep[i]["trianglesOffset"] = i - 2; deep[i]["triangleCount"] = 1; deep[i]["splitAxis"] = 0xFFFFFFFF
    assert rf.check_wide_layouts(deep)["quad"] is False
    # coordinates beyond the binary16 range: no half-precision quad records (the exact quad records serve); tiny and huge boxes inside
    # the range keep the conservative margin on every plane (checked record by record inside rf_check_wide_layouts)
    rng = np.random.default_rng(5)
    for scale, want in ((1.0, True), (3000.0, True), (9000.0, False), (1e-3, True)):
        tris = (rng.uniform(-9.0, 9.0, (300, 3, 3)) * scale).astype(np.float32)
        tris[:40] = np.round(tris[:40])                                  # flat, lattice-aligned boxes
        big, _, _ = rf.build_bvh(tris.reshape(300, 9))
        st = rf.wide_layout_stats(big)
        assert bool(st["flags"] & 16) is want and bool(st["flags"] & 8) and bool(st["flags"] & 32)      # (the local-grid records have no range limit)
        if want:
            assert 1.0 < st["quad_half_area_ratio"] < 3.0
    # non-finite boxes: the packed slab test is not used at all
    nodes[1]["max"] = (np.inf, 0.5, 0.5)
    assert rf.check_wide_layouts(nodes)["regular"] is False
    # a broken child link is refused before any layout is built
    nodes[0]["secondChildOffset"] = 7
    with pytest.raises(Exception):
        rf.check_wide_layouts(nodes)


def test_occluder_cache_entries_of_a_tree_with_an_orphan_node_and_of_a_tree_that_is_not_nested():
    """ADVICE r4.  (i) validateScene accepts interior nodes no parent reaches.  The occluder-cache entries leafBoxesIntoTriangles writes into the triangle
    records used to number quad records by node order and depth (an orphan counted as depth 0), buildWide only numbers what the root reaches: on a tree with
    an orphan in front of the right subtree every entry behind it named the wrong record -- or one past the end.  rf_check_wide_layouts walks down from every
    entry (levels 1..3) and must find the leaf below the record it names.  (ii) A child box that sticks out of its parent's breaks what the two-level
    records, the exact box at the leaf and the occluder cache rest on: such a tree keeps to the binary records."""
    rng = np.random.default_rng(11)
    n = 96
    tris = (rng.uniform(-3, 3, (n, 1, 3)) + rng.normal(0, 0.2, (n, 3, 3))).astype(np.float32)
    nodes, _, _ = rf.build_bvh(tris.reshape(n, 9))
    assert rf.check_wide_layouts(nodes)["quad"]
    S = int(nodes[0]["secondChildOffset"])
    orphan = np.zeros(3, dtype=rf.NODE_DTYPE)
    orphan[0]["min"] = (-1, -1, -1); orphan[0]["max"] = (1, 1, 1); orphan[0]["secondChildOffset"] = S + 2; orphan[0]["splitAxis"] = 0
    for k, x in ((1, -0.5), (2, 0.5)):
        orphan[k]["min"] = (x - 0.5, -1, -1); orphan[k]["max"] = (x + 0.5, 1, 1)
        orphan[k]["trianglesOffset"] = n + k - 1; orphan[k]["triangleCount"] = 1; orphan[k]["splitAxis"] = 0xFFFFFFFF
    grown = np.concatenate([nodes[:S], orphan, nodes[S:]])
    for i in range(grown.shape[0]):
        if S <= i < S + 3:
            continue
        if grown[i]["triangleCount"] == 0 and grown[i]["secondChildOffset"] >= S:
            grown[i]["secondChildOffset"] += 3
    got = rf.check_wide_layouts(grown)          # (throws "the leaf is not below the record it names" with the old numbering)
    assert got["quad"] and got["quad_half"] and got["quad_local"] and got["oct"]
    # (ii) the left child of the root pokes out of the root's box
    poke = nodes.copy()
    poke[1]["max"] = tuple(np.asarray(poke[0]["max"]) + np.float32(1.0))
    got = rf.check_wide_layouts(poke)
    assert got["regular"] and not got["quad"] and not got["quad_half"] and not got["quad_local"] and not got["oct"]


# ---------------------------------------------------------------- the product's CPU query (rf_query.cpp): config 1 without a GPU
def test_host_query_node_visits_equal_the_golden_map_and_the_oracle(duck_pt, duck_oracle):
    """BASELINE.json config 1 through the PRODUCT on a GPU-less box: Duck.glb -> .pt arrays (product ingest + builder) -> the
    bvh-visualizer pass of rf_bvh_visualizer_pass == the committed golden node-visit map == the oracle, bit for bit."""
    from oracle import orc
    g = np.load(os.path.join(GOLDEN, "duck_golden.npz"))
    a = duck_pt.arrays()
    cam = rf.bvh_visualizer_camera(a["bvhNodes"], 1.0)
    out = rf.bvh_visualizer_pass(cam, 256, 256, a["bvhNodes"], a["bvhPositionAttributes"], threads=3)
    assert np.array_equal(out["nodesVisited"], g["viz256_nodes_visited"].astype(np.uint32))
    assert np.array_equal(np.packbits(out["hit"]), g["viz256_hit"])
    assert int(out["nodesVisited"].sum()) == 1209382 and int(out["hit"].sum()) == 19462 and int(out["triTests"].sum()) == 69097
    # 36-byte Positions and 48-byte PositionAttribute records traverse alike; one thread == many threads
    out48 = rf.bvh_visualizer_pass(cam, 256, 256, a["bvhNodes"], a["trianglePositionAttributes"], threads=1)
    for k in ("nodesVisited", "hit", "triTests"):
        assert np.array_equal(out[k], out48[k])
    assert np.array_equal(bits(out["t"]), bits(out48["t"]))
    # the reference tool's own size, against the oracle (t included) and the golden row sums
    aspect = np.float32(np.float32(1280) / np.float32(720))
    cam = rf.bvh_visualizer_camera(a["bvhNodes"], aspect)
    big = rf.bvh_visualizer_pass(cam, 1280, 720, a["bvhNodes"], a["bvhPositionAttributes"])
    cpu = orc.bvh_visualize(duck_oracle.nodes, duck_oracle.tris36, rf.camera_to_array(cam), 1280, 720)
    assert np.array_equal(big["nodesVisited"], cpu["nodesVisited"]) and np.array_equal(big["hit"], cpu["hit"])
    assert np.array_equal(bits(big["t"]), bits(cpu["t"])) and np.array_equal(big["triTests"], cpu["triTests"])
    assert int(big["nodesVisited"].sum()) == 9979946 and int(big["nodesVisited"].max()) == 159
    assert np.array_equal(big["nodesVisited"].reshape(720, 1280).sum(axis=1), g["viz720_row_sums"])
    # a row range leaves the other rows untouched
    part = rf.bvh_visualizer_pass(cam, 1280, 720, a["bvhNodes"], a["bvhPositionAttributes"], row_begin=100, row_end=104)
    assert np.array_equal(part["nodesVisited"].reshape(720, 1280)[100:104], big["nodesVisited"].reshape(720, 1280)[100:104])
    assert part["nodesVisited"].reshape(720, 1280)[:100].sum() == 0 and part["nodesVisited"].reshape(720, 1280)[104:].sum() == 0


def test_host_query_reference_bvh_test_grid(duck_pt, duck_oracle):
    """src/tests/bvh.cpp:76-101: the 64 x 64 ray grid, tMax = 1000 -- hit, t, triangle == golden (== brute force), p / u / v /
    nodesVisited / stack high-water == oracle; single-ray entry point == batch."""
    from oracle import orc
    g = np.load(os.path.join(GOLDEN, "duck_golden.npz"))
    a = duck_pt.arrays()
    out = rf.intersect_bvh_batch(g["grid_rays"], a["bvhNodes"], a["bvhPositionAttributes"], 1000.0, threads=4)
    assert np.array_equal(out["hit"], g["grid_hit"]) and int(out["hit"].sum()) == 1216
    h = out["hit"] == 1
    assert np.array_equal(bits(out["t"][h]), bits(g["grid_t"][h])) and np.array_equal(out["tri"][h], g["grid_tri"][h])
    assert (out["tri"][~h] == 0xFFFFFFFF).all()
    ref = orc.intersect_bvh_batch(duck_oracle.nodes, duck_oracle.tris36, g["grid_rays"], 1000.0)
    assert np.array_equal(out["nodesVisited"], ref["nodesVisited"]) and np.array_equal(out["triTests"], ref["triTests"])
    assert np.array_equal(out["stackHigh"], ref["stackHigh"])
    assert np.array_equal(bits(out["p"][h]), bits(ref["p"][h])) and np.array_equal(bits(out["uv"][h]), bits(ref["uv"][h]))
    for i in (0, 777, 2048, 4095, int(np.flatnonzero(h)[0]), int(np.flatnonzero(h)[-1])):
        hit, rec, st = rf.intersect_bvh(g["grid_rays"][i], a["bvhNodes"], a["bvhPositionAttributes"], 1000.0)
        assert hit == bool(out["hit"][i]) and st["nodes_visited"] == out["nodesVisited"][i]
        if hit:
            assert rec["t"].view(np.uint32) == out["t"][i].view(np.uint32) and rec["triangle"] == out["tri"][i]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_host_query_on_random_soups_and_hostile_rays(seed):
    """Random triangle soups (degenerate and duplicate triangles included) with axis-parallel, +-0, denormal, NaN and 1e30 rays:
    host query == oracle on every output bit."""
    from oracle import orc
    rng = np.random.default_rng(seed)
    n = 700
    P = (rng.standard_normal((n, 3, 3)) * 0.15 + rng.uniform(-2, 2, (n, 1, 3))).astype(np.float32)
    P[::50, 1] = P[::50, 0]                       # degenerate
    P[1::60] = P[0:1]                             # duplicates
    P[::7] = np.round(P[::7] * 4) / 4             # vertices on a lattice: rays through box planes
    nodes, idx, _ = rf.build_bvh(P.reshape(n, 9))
    tris = orc.reorder(P.reshape(n, 9), idx)
    m = 3000
    rays = np.concatenate([rng.uniform(-3, 3, (m, 3)), rng.standard_normal((m, 3))], axis=1).astype(np.float32)
    rays[::9, 3] = 0.0; rays[1::9, 4] = -0.0; rays[2::9, 3:5] = 0.0
    rays[3::97, 5] = 1e-42                        # denormal component
    rays[4::101, 0] = np.nan; rays[5::103, 3] = np.nan; rays[6::107, 1] = 1e30
    rays[7::11, :3] = np.round(rays[7::11, :3] * 4) / 4
    for tmax in (np.float32(1000.0), np.finfo(np.float32).max):
        out = rf.intersect_bvh_batch(rays, nodes, tris, tmax, threads=2)
        ref = orc.intersect_bvh_batch(nodes, tris, rays, tmax)
        assert np.array_equal(out["hit"], ref["hit"]) and np.array_equal(out["nodesVisited"], ref["nodesVisited"])
        assert np.array_equal(out["triTests"], ref["triTests"]) and np.array_equal(out["stackHigh"], ref["stackHigh"])
        h = out["hit"] == 1
        assert h.sum() > 100
        assert np.array_equal(out["tri"][h], ref["tri"][h]) and np.array_equal(bits(out["t"][h]), bits(ref["t"][h]))
        assert np.array_equal(bits(out["p"][h]), bits(ref["p"][h])) and np.array_equal(bits(out["uv"][h]), bits(ref["uv"][h]))


def test_host_query_rejects_malformed_trees_and_bad_strides(duck_pt):
    a = duck_pt.arrays()
    nodes = a["bvhNodes"].copy()
    ray = np.array([0.1, 0.8, 5.0, 0.0, 0.0, -1.0], np.float32)
    bad = nodes.copy(); bad[0]["secondChildOffset"] = len(nodes) + 5
    with pytest.raises(rf.RayfinderError):
        rf.intersect_bvh(ray, bad, a["bvhPositionAttributes"], 1000.0)
    bad = nodes.copy(); bad["trianglesOffset"][nodes["triangleCount"] > 0] = len(a["bvhPositionAttributes"])   # every leaf points past the array
    with pytest.raises(rf.RayfinderError):
        rf.bvh_visualizer_pass(rf.bvh_visualizer_camera(nodes, 1.0), 64, 64, bad, a["bvhPositionAttributes"])
    out = np.zeros(1, rf.INTERSECTION_DTYPE); hit = rf._ffi.C.c_int(0)
    rc = rf.lib.rf_intersect_bvh(ray.ctypes.data, nodes.ctypes.data, len(nodes), a["bvhPositionAttributes"].ctypes.data, 40, len(a["bvhPositionAttributes"]),
                                 np.float32(1.0), out.ctypes.data, None, rf._ffi.C.byref(hit))
    assert rc == rf._ffi.RF_ERROR_INVALID_ARGUMENT


def test_host_query_deep_chain_needs_no_fixed_stack():
    """A 200-deep degenerate chain (every far child pending): the reference's 32-entry array would be overrun; the host query's
    pending list grows (stack high-water 200) and still finds the hit at the bottom."""
    depth = 200
    P = np.zeros((depth + 1, 9), np.float32)
    for i in range(depth + 1):
        z = np.float32(-1.0 - i)
        P[i] = [-1, -1, z, 1, -1, z, 0, 1, z]
    # layout: interiors at 0..depth-1 (node k's first child is k+1), then leaf of the bottom, then the pending leaves
    nodes = np.zeros(2 * depth + 1, rf.NODE_DTYPE)
    for k in range(depth):
        nodes[k]["min"] = [-1, -1, -1.0 - depth]; nodes[k]["max"] = [1, 1, -1.0 - k]
        nodes[k]["splitAxis"] = 0
        nodes[k]["secondChildOffset"] = 2 * depth - k
    def leaf(at, tri):
        nodes[at]["min"] = [-1, -1, P[tri][2]]; nodes[at]["max"] = [1, 1, P[tri][2]]
        nodes[at]["trianglesOffset"] = tri; nodes[at]["triangleCount"] = 1; nodes[at]["splitAxis"] = 0xFFFFFFFF
    leaf(depth, depth)                                         # first child of the last interior node: the farthest triangle
    for k in range(depth):
        leaf(2 * depth - k, k)                                 # second child of interior k: the triangle at z = -1-k
    ray = np.array([0.0, 0.0, 5.0, 1e-3, 1e-3, -1.0], np.float32)   # +x: first child first, every second child pending
    hit, rec, st = rf.intersect_bvh(ray, nodes, P, np.finfo(np.float32).max)
    assert hit and st["stack_high_water"] == depth and st["nodes_visited"] == 2 * depth + 1
    assert rec["triangle"] == 0 and abs(rec["t"] - 6.0) < 1e-2      # the nearest triangle wins in the end


def test_cli_bvh_visualizer_cpu_writes_the_reference_grey_map(duck_pt, tmp_path):
    """rf-bvh-visualizer --cpu (no GPU): the PNG it writes decodes to grey = u32(min(0.01 * nodesVisited, 1) * 255), alpha 255
    (src/bvh-visualizer/main.cpp:73-84) of the host pass's node-visit map -- at the tool's default 1280 x 720 and at 256 x 256
    (BASELINE.json config 1), from the .glb and from the .pt file."""
    import subprocess
    from PIL import Image
    exe = os.path.join(ROOT, "rayfinder_amd", "bin", "rf-bvh-visualizer")
    a = duck_pt.arrays()
    pt_path = tmp_path / "Duck.pt"
    duck_pt.save(pt_path)
    for (w, h, src, args) in ((256, 256, DUCK, ["256", "256"]), (1280, 720, str(pt_path), [])):
        out = tmp_path / f"viz_{w}.png"
        r = subprocess.run([exe, "--cpu", "--threads", "2", "--out", str(out), src] + args, capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0, r.stderr
        img = np.array(Image.open(out))
        assert img.shape == (h, w, 4) and (img[..., 3] == 255).all()
        cam = rf.bvh_visualizer_camera(a["bvhNodes"], np.float32(np.float32(w) / np.float32(h)))
        nv = rf.bvh_visualizer_pass(cam, w, h, a["bvhNodes"], a["bvhPositionAttributes"])["nodesVisited"]
        grey = rf.bvh_visualizer_grey(nv).reshape(h, w)
        for c in range(3):
            assert np.array_equal(img[..., c], grey)
        assert f"{int(nv.sum())} node visits" in r.stdout
    # the grey formula itself, against the oracle's restatement of main.cpp:73-76
    from oracle import orc
    for n in (0, 1, 49, 50, 99, 100, 101, 157, 5000):
        assert int(rf.bvh_visualizer_grey(np.array([n]))[0]) == orc.lib().orc_bvh_visualizer_pixel(n) & 0xFF


def test_gltf_sparse_accessors_and_normalized_integer_attributes(tmp_path):
    """What the reference gets from cgltf_accessor_unpack_floats (gltf_model.cpp:400-438): POSITION as a sparse accessor over a
    buffer view and as a sparse accessor over NOTHING (zeros + overlay), NORMAL as normalized int8 in an interleaved (byteStride)
    view, TEXCOORD_0 as normalized uint16 / int16 -- known answers, and product bake == oracle ingest bit for bit."""
    import base64
    blob = bytearray()
    views = []

    def view(data, stride=None):
        while len(blob) % 4:
            blob.append(0)
        v = {"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        views.append(v)
        blob.extend(data)
        return len(views) - 1

    nv = 6
    base_pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1]], "<f4")
    v_pos = view(base_pos.tobytes())
    sp_idx = np.array([1, 4, 1], "<u2")                       # index 1 twice: the later value wins
    sp_val = np.array([[9, 9, 9], [0.5, -2, 3], [2, 0.25, -1]], "<f4")
    v_spi, v_spv = view(sp_idx.tobytes()), view(sp_val.tobytes())
    sp2_idx = np.array([0, 2, 3, 5], "<u1")
    sp2_val = np.array([[1, 2, 3], [-1, 0, 2], [4, 4, -4], [0.5, 0.5, 0.5]], "<f4")
    v_sp2i, v_sp2v = view(sp2_idx.tobytes()), view(sp2_val.tobytes())
    # interleaved vertex buffer, 8-byte stride: int8 normal xyz (+1 pad), uint16 uv
    inter = bytearray()
    nrm8 = np.array([[127, 0, 0], [0, -128, 0], [0, 0, 127], [-127, 64, 3], [90, 90, 0], [1, 2, 3]], np.int8)
    uv16 = np.array([[0, 65535], [32768, 1], [65535, 65535], [100, 200], [40000, 3], [7, 7]], "<u2")
    for i in range(nv):
        inter += nrm8[i].tobytes() + b"\0" + uv16[i].tobytes()
    v_inter = view(bytes(inter), stride=8)
    suv16 = np.array([[32767, -32768], [0, 1], [-1, 16384], [5, 5], [-32767, 32767], [1000, -1000]], "<i2")
    v_suv = view(suv16.tobytes())
    idx = np.array([0, 1, 2, 2, 1, 3, 4, 5, 0], "<u2")
    v_idx = view(idx.tobytes())
    accessors = [
        {"bufferView": v_pos, "componentType": 5126, "count": nv, "type": "VEC3",
         "sparse": {"count": 3, "indices": {"bufferView": v_spi, "componentType": 5123}, "values": {"bufferView": v_spv}}},          # 0
        {"componentType": 5126, "count": nv, "type": "VEC3",
         "sparse": {"count": 4, "indices": {"bufferView": v_sp2i, "componentType": 5121}, "values": {"bufferView": v_sp2v}}},        # 1
        {"bufferView": v_inter, "componentType": 5120, "normalized": True, "count": nv, "type": "VEC3"},                             # 2
        {"bufferView": v_inter, "byteOffset": 4, "componentType": 5123, "normalized": True, "count": nv, "type": "VEC2"},            # 3
        {"bufferView": v_suv, "componentType": 5122, "normalized": True, "count": nv, "type": "VEC2"},                               # 4
        {"bufferView": v_idx, "componentType": 5123, "count": idx.size, "type": "SCALAR"},                                           # 5
    ]
    js = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 1]}], "nodes": [{"mesh": 0}, {"mesh": 1, "translation": [5, 0, 0]}],
          "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 2, "TEXCOORD_0": 3}, "indices": 5, "material": 0}]},
                     {"primitives": [{"attributes": {"POSITION": 1, "NORMAL": 2, "TEXCOORD_0": 4}, "indices": 5, "material": 0}]}],
          "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [1.0, 0.5, 0.25, 1.0]}}],
          "accessors": accessors, "bufferViews": views,
          "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode()}]}
    p = tmp_path / "sparse.gltf"
    p.write_text(json.dumps(js))
    # known answers of the oracle's unpack (cgltf's published algorithm)
    ojs, obufs = gltf_ref.load_container(str(p))
    pos0 = gltf_ref.unpack_floats(ojs, obufs, 0)
    want0 = base_pos.copy(); want0[1] = [2, 0.25, -1]; want0[4] = [0.5, -2, 3]
    assert np.array_equal(pos0, want0)
    pos1 = gltf_ref.unpack_floats(ojs, obufs, 1)
    want1 = np.zeros((nv, 3), np.float32); want1[[0, 2, 3, 5]] = sp2_val
    assert np.array_equal(pos1, want1)
    n8 = gltf_ref.unpack_floats(ojs, obufs, 2)
    assert n8[0, 0] == 1.0 and n8[1, 1] == np.float32(-128) / np.float32(127) and n8[1, 1] < -1.0        # no clamp in cgltf
    assert np.array_equal(n8, nrm8.astype(np.float32) / np.float32(127))
    assert np.array_equal(gltf_ref.unpack_floats(ojs, obufs, 3), uv16.astype(np.float32) / np.float32(65535))
    assert np.array_equal(gltf_ref.unpack_floats(ojs, obufs, 4), suv16.astype(np.float32) / np.float32(32767))
    # product bake == oracle ingest
    pt = rf.PtFormat.from_gltf(p)
    a = pt.arrays()
    m = gltf_ref.load_model(str(p))
    P, N, T, I = gltf_ref.flatten(m)
    onodes, oidx, _ = orc.build_bvh(P)
    assert a["bvhNodes"].tobytes() == onodes.tobytes()
    pa, va = gltf_ref.gpu_layout(orc.reorder(P, oidx), orc.reorder(N, oidx), orc.reorder(T, oidx), orc.reorder(I, oidx))
    assert np.array_equal(bits(a["trianglePositionAttributes"]), bits(pa))
    assert np.array_equal(bits(a["triangleVertexAttributes"]), bits(va))
    assert len(P) == 6                                            # 3 triangles per mesh
    # malformed sparse data is refused, not read out of bounds
    for mutate in ("index", "values", "sparse_indices_accessor"):
        bad = json.loads(json.dumps(js))
        if mutate == "index":
            bad["accessors"][0]["sparse"]["indices"]["componentType"] = 5125         # 3 x u32 do not fit the 6-byte view
        elif mutate == "values":
            bad["accessors"][1]["sparse"]["count"] = 5
            bad["bufferViews"][v_sp2i]["byteLength"] = 4
        else:
            bad["accessors"][5]["sparse"] = {"count": 1, "indices": {"bufferView": v_spi, "componentType": 5123}, "values": {"bufferView": v_idx}}
        q = tmp_path / f"bad_{mutate}.gltf"
        q.write_text(json.dumps(bad))
        with pytest.raises(rf.RayfinderError):
            rf.PtFormat.from_gltf(q)


def test_gltf_strides_and_counts_whose_product_wraps_are_rejected(tmp_path):
    """byteStride and count come straight from the JSON.  (count - 1) * stride formed in size_t wraps for byteStride = 2^54 and
    count = 1025 (advisor, round 3: the bounds check passed and the bake read base + i * 2^54).  Every such file must be an
    error with a message -- in a child process, so that a crash is a test failure and not the end of the test run."""
    import base64
    import subprocess
    blob = np.arange(6 * 3, dtype="<f4").tobytes() + np.arange(6 * 2, dtype="<f4").tobytes() + np.array([0, 1, 2, 3, 4, 5], "<u2").tobytes()

    def doc(stride=None, count=6, sparse_count=None, idx_count=6):
        views = [{"buffer": 0, "byteOffset": 0, "byteLength": 72}, {"buffer": 0, "byteOffset": 72, "byteLength": 48}, {"buffer": 0, "byteOffset": 120, "byteLength": 12}]
        if stride is not None:
            views[0]["byteStride"] = stride
            views[1]["byteStride"] = stride
        acc = [{"bufferView": 0, "componentType": 5126, "count": count, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": count, "type": "VEC2"},
               {"bufferView": 2, "componentType": 5123, "count": idx_count, "type": "SCALAR"}]
        if sparse_count is not None:
            acc[0] = {"componentType": 5126, "count": 6, "type": "VEC3",
                      "sparse": {"count": sparse_count, "indices": {"bufferView": 2, "componentType": 5123}, "values": {"bufferView": 0}}}
        return {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
                "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 0, "TEXCOORD_0": 1}, "indices": 2, "material": 0}]}],
                "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [1.0, 1.0, 1.0, 1.0]}}], "accessors": acc, "bufferViews": views,
                "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]}

    cases = {"ok": doc(), "stride_2p54": doc(stride=2 ** 54, count=1025), "stride_2p61": doc(stride=2 ** 61, count=9), "stride_2p63": doc(stride=2 ** 63, count=3),
             "stride_3": doc(stride=3), "stride_256": doc(stride=256, count=1), "stride_below_element": doc(stride=8),
             "count_2p60": doc(count=2 ** 60), "count_negative": doc(count=-6), "count_1e300": doc(count=1e300),
             "sparse_2p62": doc(sparse_count=2 ** 62), "sparse_2p63_plus": doc(sparse_count=2 ** 63 + 1), "index_count_2p63": doc(idx_count=3 * 2 ** 61)}
    for name, js in cases.items():
        (tmp_path / f"{name}.gltf").write_text(json.dumps(js))
    code = ("import sys; sys.path.insert(0, %r); import rayfinder_amd as rf\n"
            "for p in sys.argv[1:]:\n"
            "    try:\n"
            "        pt = rf.PtFormat.from_gltf(p); print('OK', p, pt.view().num_triangle_position_attributes)\n"
            "    except Exception as e:\n"
            "        print('ERR', p, str(e)[:120])\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code] + [str(tmp_path / f"{n}.gltf") for n in cases], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"the bake crashed (exit {r.returncode}): {r.stdout[-2000:]} {r.stderr[-2000:]}"
    lines = {os.path.basename(l.split()[1])[:-5]: l for l in r.stdout.splitlines() if l.startswith(("OK", "ERR"))}
    assert set(lines) == set(cases)
    assert lines["ok"].startswith("OK") and lines["ok"].rstrip().endswith(" 2")
    # strides that break the glTF text's rules (4..252, >= the element) but stay inside the view load, with the stride they state: the reference
    # reads through cgltf without cgltf_validate (gltf_model.cpp) and does the same (do not reject what the reference loads)
    loads_like_the_reference = {"stride_3", "stride_below_element"}   # ("stride_256" has one vertex for six indices: refused for that)
    for name in cases:
        if name in loads_like_the_reference:
            assert lines[name].startswith("OK"), lines[name]
        elif name != "ok":
            assert lines[name].startswith("ERR") and "glTF" in lines[name], lines[name]


# ---------------------------------------------------------------- round 4: what the GPU tests hand to the checker (also run under -m gpu)
def test_product_sky_states_cameras_and_atrium_bake_equal_the_oracles():
    import checker_inputs
    checker_inputs.check_sky_states()
    checker_inputs.check_cameras()
    nodes, info = checker_inputs.check_atrium_bake()
    assert nodes == 459645 and info["triangles"] == 265024


def test_clutter_atrium_is_the_harder_stand_in_and_leaves_the_plain_one_alone():
    """scenes.atrium(detail="clutter"): deterministic, bakes to the oracle builder's bytes like the plain atrium, closed props wound
    so that the reference's geometric-normal ray offset leaves them lit, and the plain atrium's digest -- the BENCH series' workload -- is unchanged."""
    import checker_inputs
    from rayfinder_amd import scenes
    pt, info = scenes.atrium(1, "clutter")
    assert info["triangles"] == 358504 and info["textures"] == 25 and info["digest"] == "da326758c181ba04" and "clutter" in info["name"]
    P, N, UV, T = scenes.atrium_triangles(1, "clutter")
    assert np.isfinite(P).all() and np.isfinite(N).all() and np.isfinite(UV).all() and int(T.max()) < 25
    assert checker_inputs.check_bake_against_oracle_builder(P, N, UV, T, pt) == 644457
    geo = np.cross(P[:, 3:6] - P[:, 0:3], P[:, 6:9] - P[:, 0:3])
    facing = (geo * (N[:, 0:3] + N[:, 3:6] + N[:, 6:9])).sum(axis=1)
    assert (facing[-200000:] > 0).mean() > 0.995               # the props (the last ~214 k triangles): geometric and shading normals on the same side
    assert scenes.atrium(1)[1]["digest"] == "206d06350b4893c2" and scenes.atrium(1)[1]["triangles"] == 265024
