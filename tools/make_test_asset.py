#!/usr/bin/env python3
"""tools/make_test_asset.py <out_dir>  -- writes a small "courtyard" in the shape real-world glTF assets (Sponza) have,
to exercise everything the bake needs beyond Duck.glb (src/common/gltf_model.cpp:74-121,266-465):

  courtyard.gltf   JSON with EXTERNAL buffer and image URIs (one of them percent-encoded), a node hierarchy three
                   levels deep with translation / rotation / scale on every level and one raw `matrix` node, six meshes,
                   several primitives per mesh, u8 / u16 / u32 indices, a mesh instanced by two nodes is avoided
                   (the reference indexes transforms by mesh, gltf_model.cpp:305) -- every mesh has one node
  courtyard.bin    vertex / index data
  tex_*.png        RGB, RGBA, palette and 16-bit grey PNGs
  tex_*.jpg        baseline 4:2:0, progressive 4:4:4 and grey JPEGs
  materials        eight: texture-backed (images shared between materials: texture dedup by image index,
                   gltf_model.cpp:170-243) and baseColorFactor-only (dedup by FNV-1a of the factor)

Deterministic (fixed seeds, Pillow encoders).  ~6 000 triangles.  Needs numpy + Pillow; nothing of the product."""
import io
import json
import os
import sys

import numpy as np


def _grid(nx, nz, size, height_fn):
    xs, zs = np.linspace(-size, size, nx), np.linspace(-size, size, nz)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = height_fn(X, Z)
    pos = np.stack([X, Y, Z], -1).reshape(-1, 3)
    uv = np.stack([(X / size + 1) * 2.0, (Z / size + 1) * 2.0], -1).reshape(-1, 2)          # tiles 4 x 4: REPEAT wrap exercised
    idx = []
    for i in range(nx - 1):
        for j in range(nz - 1):
            a, b, c, d = i * nz + j, (i + 1) * nz + j, (i + 1) * nz + j + 1, i * nz + j + 1
            idx += [a, d, c, a, c, b]
    nrm = np.zeros_like(pos)
    tri = np.array(idx).reshape(-1, 3)
    fn = np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]])
    for k in range(3):
        np.add.at(nrm, tri[:, k], fn)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return pos, nrm, uv, np.array(idx)


def _prism(sides, radius, height):
    ang = np.arange(sides) * 2 * np.pi / sides
    ring = np.stack([np.cos(ang) * radius, np.zeros(sides), np.sin(ang) * radius], -1)
    pos, nrm, uv, idx = [], [], [], []
    for s in range(sides):
        t = (s + 1) % sides
        n = np.array([np.cos((ang[s] + ang[s] + 2 * np.pi / sides) / 2), 0, np.sin((ang[s] + ang[s] + 2 * np.pi / sides) / 2)])
        base = len(pos)
        for (p, v) in ((ring[s], 0.0), (ring[t], 0.0), (ring[t] + [0, height, 0], 1.0), (ring[s] + [0, height, 0], 1.0)):
            pos.append(p); nrm.append(n); uv.append([(s + (p is ring[t])) / sides * 3.0, v * 2.0])
        idx += [base, base + 2, base + 1, base, base + 3, base + 2]
    return np.array(pos, float), np.array(nrm, float), np.array(uv, float), np.array(idx)


def _sphere(rings, segs, radius):
    pos, nrm, uv, idx = [], [], [], []
    for r in range(rings + 1):
        th = np.pi * r / rings
        for s in range(segs + 1):
            ph = 2 * np.pi * s / segs
            n = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
            pos.append(n * radius); nrm.append(n); uv.append([s / segs, r / rings])
    for r in range(rings):
        for s in range(segs):
            a, b = r * (segs + 1) + s, (r + 1) * (segs + 1) + s
            if r > 0:
                idx += [a, a + 1, b]          # skip the degenerate pole triangles
            if r < rings - 1:
                idx += [a + 1, b + 1, b]
    return np.array(pos), np.array(nrm), np.array(uv), np.array(idx)


def _texture(kind, seed, size=64):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    if kind == "bricks":
        img = np.zeros((size, size, 3), np.uint8)
        img[..., 0] = 150 + 40 * (((yy // 8) % 2) ^ ((xx // 16 + (yy // 8) % 2) % 2)); img[..., 1] = 70; img[..., 2] = 50
        img[(yy % 8 == 0) | ((xx + 8 * ((yy // 8) % 2)) % 16 == 0)] = (200, 200, 190)
    elif kind == "marble":
        v = (128 + 90 * np.sin(xx / 5.0 + 3 * np.sin(yy / 9.0)) + rng.normal(0, 6, (size, size))).clip(0, 255)
        img = np.stack([v, v * 0.95, v * 0.85], -1).astype(np.uint8)
    elif kind == "checker":
        v = (((xx // 8) + (yy // 8)) % 2) * 180 + 40
        img = np.stack([v, 255 - v, (v // 2) + 60], -1).astype(np.uint8)
    else:
        img = rng.integers(0, 256, (size, size, 3)).astype(np.uint8)
    return img


def write_courtyard(out_dir):
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    blob = bytearray()
    views, accessors, meshes = [], [], []

    def add(data):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)})
        blob.extend(data)
        return len(views) - 1

    def prim(pos, nrm, uv, idx, material, index_type):
        accs = {}
        for name, arr, typ in (("POSITION", pos, "VEC3"), ("NORMAL", nrm, "VEC3"), ("TEXCOORD_0", uv, "VEC2")):
            arr = np.asarray(arr, "<f4")
            acc = {"bufferView": add(arr.tobytes()), "componentType": 5126, "count": len(arr), "type": typ}
            if name == "POSITION":
                acc["min"], acc["max"] = arr.min(0).tolist(), arr.max(0).tolist()
            accessors.append(acc)
            accs[name] = len(accessors) - 1
        ct = {"u8": (np.uint8, 5121), "u16": (np.uint16, 5123), "u32": (np.uint32, 5125)}[index_type]
        idx = np.asarray(idx).astype(ct[0])
        accessors.append({"bufferView": add(idx.tobytes()), "componentType": ct[1], "count": int(idx.size), "type": "SCALAR"})
        return {"attributes": accs, "indices": len(accessors) - 1, "material": material, "mode": 4}

    # ---- images (external files)
    def save(name, img, fmt, **kw):
        Image.fromarray(img).save(os.path.join(out_dir, name), fmt, **kw)
    save("tex_bricks.png", _texture("bricks", 1), "PNG")
    rgba = np.concatenate([_texture("marble", 2), np.full((64, 64, 1), 200, np.uint8)], -1)
    save("tex_marble rgba.png", rgba, "PNG")                                                    # space in the name: percent-encoded URI
    Image.fromarray(_texture("checker", 3)).quantize(16).save(os.path.join(out_dir, "tex_checker_palette.png"), "PNG")
    save("tex_noise_baseline.jpg", _texture("noise", 4, 48), "JPEG", quality=88, subsampling=2)
    save("tex_marble_progressive.jpg", _texture("marble", 5, 80), "JPEG", quality=92, subsampling=0, progressive=True)
    Image.fromarray(_texture("marble", 6)[..., 0]).save(os.path.join(out_dir, "tex_grey.jpg"), "JPEG", quality=80)
    images = [{"uri": "tex_bricks.png"}, {"uri": "tex_marble%20rgba.png"}, {"uri": "tex_checker_palette.png"},
              {"uri": "tex_noise_baseline.jpg"}, {"uri": "tex_marble_progressive.jpg"}, {"uri": "tex_grey.jpg"}]
    textures = [{"source": i, "sampler": 0} for i in range(6)] + [{"source": 0, "sampler": 0}]   # texture 6 shares image 0
    samplers = [{"wrapS": 10497, "wrapT": 10497}]
    tex_mat = lambda t: {"pbrMetallicRoughness": {"baseColorTexture": {"index": t}}}
    fac_mat = lambda f: {"pbrMetallicRoughness": {"baseColorFactor": f}}
    materials = [tex_mat(0), tex_mat(1), tex_mat(2), tex_mat(3), tex_mat(4), tex_mat(5), tex_mat(6),
                 fac_mat([0.8, 0.2, 0.1, 1.0]), fac_mat([0.1, 0.6, 0.9, 1.0]), fac_mat([0.8, 0.2, 0.1, 1.0])]   # 7 and 9: same factor -> one texture

    # ---- meshes
    floor = _grid(25, 25, 6.0, lambda x, z: 0.05 * np.sin(x * 1.3) * np.cos(z * 1.7))
    meshes.append({"primitives": [prim(*floor, 0, "u16")]})
    col = _prism(12, 0.35, 3.0)
    cap = _prism(4, 0.55, 0.25)
    meshes.append({"primitives": [prim(*col, 1, "u8"), prim(cap[0] + [0, 3.0, 0], cap[1], cap[2], cap[3], 7, "u8")]})       # column + capital
    meshes.append({"primitives": [prim(*col, 4, "u8"), prim(cap[0] + [0, 3.0, 0], cap[1], cap[2], cap[3], 9, "u8")]})
    ball = _sphere(24, 32, 0.8)
    meshes.append({"primitives": [prim(*ball, 3, "u32")]})
    wall = _grid(9, 5, 1.0, lambda x, z: 0 * x)
    meshes.append({"primitives": [prim(*wall, 2, "u16"), prim(wall[0] + [0, 0.4, 0], wall[1], wall[2], wall[3], 8, "u16")]})
    roof = _grid(7, 7, 1.0, lambda x, z: 0.3 * (1 - x * x) * (1 - z * z))
    meshes.append({"primitives": [prim(*roof, 5, "u16"), prim(roof[0] * [1, 1, -1] + [0, 0.02, 0], -roof[1], roof[2], roof[3][::-1], 6, "u16")]})

    s2 = float(np.sqrt(0.5))
    nodes = [
        {"name": "root", "children": [1, 2, 5], "scale": [0.5, 0.5, 0.5], "translation": [1.0, 0.0, -1.0]},
        {"name": "floor", "mesh": 0},
        {"name": "colonnade", "children": [3, 4], "rotation": [0.0, 0.3826834, 0.0, 0.9238795], "translation": [0.0, 0.0, 1.5]},
        {"name": "column A", "mesh": 1, "translation": [-2.5, 0.0, 0.0], "scale": [1.0, 1.2, 1.0]},
        {"name": "column B", "mesh": 2, "translation": [2.5, 0.0, 0.0], "rotation": [0.0, s2, 0.0, s2]},
        {"name": "props", "children": [6, 7, 8]},
        {"name": "ball", "mesh": 3, "translation": [0.5, 0.9, -2.0], "scale": [1.0, 0.8, 1.3]},
        # a raw column-major matrix: rotate the vertical wall upright (x-rotation by 90 degrees) and push it back
        {"name": "wall", "mesh": 4, "matrix": [3.0, 0, 0, 0, 0, 0, 3.0, 0, 0, -3.0, 0, 0, 0.0, 3.0, -5.0, 1.0]},
        {"name": "roof", "mesh": 5, "translation": [0.0, 4.2, 0.5], "scale": [3.5, 1.0, 3.5], "rotation": [0.0871557, 0.0, 0.0, 0.9961947]},
    ]
    js = {"asset": {"version": "2.0", "generator": "rayfinder_amd tools/make_test_asset.py"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": nodes,
          "meshes": meshes, "materials": materials, "accessors": accessors, "bufferViews": views,
          "buffers": [{"byteLength": len(blob), "uri": "courtyard.bin"}], "images": images, "textures": textures, "samplers": samplers}
    open(os.path.join(out_dir, "courtyard.bin"), "wb").write(bytes(blob))
    path = os.path.join(out_dir, "courtyard.gltf")
    json.dump(js, open(path, "w"), indent=1)
    return path


if __name__ == "__main__":
    print(write_courtyard(sys.argv[1] if len(sys.argv) > 1 else "courtyard"))
