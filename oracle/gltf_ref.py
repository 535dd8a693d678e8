"""Oracle restatement (numpy) of the reference's glTF ingest.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/src/common/gltf_model.cpp:29-72 (node transforms), :170-243 (base colour
textures), :266-465 (mesh extraction, sort by texture index), flattened_model.cpp:22-43 (un-index),
texture.cpp:12-65 (BGRA packing) and pt-format/pt_format.cpp:36-79 (GPU layout arrays).

The reference reads files with cgltf 1.13 and decodes images with stb_image; both are
un-vendored third-party dependencies absent from the mount (external/CMakeLists.txt:8-10,36-38).
This module parses the GLB container / JSON with the Python stdlib and decodes images with PIL
(8-bit PNG/JPEG decoding to RGBA is lossless/standardised for PNG, so the texels are
implementation independent for PNG; JPEG decoders may differ by +-1).  glm 0.9.9.8 arithmetic
(mat4*mat4, mat4*vec4, inverseTranspose, quaternion cast) is restated in f32 with the published
operation order.  PARITY UNPINNED by any reference test beyond src/tests/gltf.cpp:5-18 (sizes).
"""
import io
import json
import os
import struct

import numpy as np

f = np.float32

_COMP = {5120: ("i1", 1), 5121: ("u1", 1), 5122: ("<i2", 2), 5123: ("<u2", 2), 5125: ("<u4", 4), 5126: ("<f4", 4)}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def load_container(path):
    data = open(path, "rb").read()
    if data[:4] == b"glTF":
        _, _, total = struct.unpack("<III", data[:12])
        off = 12
        js = None
        bins = []
        while off < total:
            clen, ctype = struct.unpack("<II", data[off:off + 8])
            chunk = data[off + 8:off + 8 + clen]
            if ctype == 0x4E4F534A:
                js = json.loads(chunk.decode("utf-8"))
            elif ctype == 0x004E4942:
                bins.append(chunk)
            off += 8 + clen
        buffers = []
        for i, b in enumerate(js.get("buffers", [])):
            if "uri" in b:
                buffers.append(_load_uri(b["uri"], path))
            else:
                buffers.append(bins[0])
        return js, buffers
    js = json.loads(data.decode("utf-8"))
    return js, [_load_uri(b["uri"], path) for b in js.get("buffers", [])]


def _load_uri(uri, gltf_path):
    if uri.startswith("data:"):
        import base64
        return base64.b64decode(uri.split(",", 1)[1])
    from urllib.parse import unquote
    return open(os.path.join(os.path.dirname(gltf_path), unquote(uri)), "rb").read()


def _gather(buf, base, count, stride, dt, sz, nc):
    out = np.zeros((count, nc), dtype=dt)
    raw = np.frombuffer(buf, dtype=np.uint8)
    for c in range(nc):
        idxs = base + np.arange(count) * stride + c * sz
        comp = np.stack([raw[idxs + k] for k in range(sz)], axis=1).copy().view(dt).reshape(count)
        out[:, c] = comp
    return out


def read_accessor(js, buffers, idx):
    acc = js["accessors"][idx]
    bv = js["bufferViews"][acc["bufferView"]]
    dt, sz = _COMP[acc["componentType"]]
    nc = _NCOMP[acc["type"]]
    base = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or sz * nc
    return _gather(buffers[bv["buffer"]], base, acc["count"], stride, dt, sz, nc)


_NORM = {5120: 127.0, 5121: 255.0, 5122: 32767.0, 5123: 65535.0}


def _as_float(raw, comp_type, normalized):
    """cgltf 1.13 cgltf_component_read_float (un-vendored dependency, published algorithm): f32 as is, normalized 8/16-bit integers
    divided by 127 / 255 / 32767 / 65535 in f32 (no clamp to -1), anything else converted as an integer."""
    if comp_type == 5126:
        return raw.astype(np.float32)
    if normalized and comp_type in _NORM:
        return (raw.astype(np.float32) / np.float32(_NORM[comp_type])).astype(np.float32)
    return raw.astype(np.float32)


def unpack_floats(js, buffers, idx):
    """cgltf_accessor_unpack_floats (what gltf_model.cpp:400-438 reads the vertex attributes through): the base accessor (zeros
    without a buffer view), then the sparse overlay -- value k, laid out with the accessor's stride, replaces element indices[k]."""
    acc = js["accessors"][idx]
    dt, sz = _COMP[acc["componentType"]]
    nc = _NCOMP[acc["type"]]
    count = acc["count"]
    normalized = bool(acc.get("normalized", False))
    stride = sz * nc
    if "bufferView" in acc:
        bv = js["bufferViews"][acc["bufferView"]]
        stride = bv.get("byteStride", 0) or stride
        out = _as_float(_gather(buffers[bv["buffer"]], bv.get("byteOffset", 0) + acc.get("byteOffset", 0), count, stride, dt, sz, nc), acc["componentType"], normalized)
    else:
        out = np.zeros((count, nc), np.float32)
    if "sparse" in acc:
        sp = acc["sparse"]
        n = sp["count"]
        ibv = js["bufferViews"][sp["indices"]["bufferView"]]
        idt, isz = _COMP[sp["indices"]["componentType"]]
        where = _gather(buffers[ibv["buffer"]], ibv.get("byteOffset", 0) + sp["indices"].get("byteOffset", 0), n, isz, idt, isz, 1).reshape(-1).astype(np.int64)
        vbv = js["bufferViews"][sp["values"]["bufferView"]]
        vals = _as_float(_gather(buffers[vbv["buffer"]], vbv.get("byteOffset", 0) + sp["values"].get("byteOffset", 0), n, stride, dt, sz, nc), acc["componentType"], normalized)
        for k in range(n):                      # in file order: a repeated index keeps the LAST value, as a sequential overlay does
            out[where[k]] = vals[k]
    return out


# ------------------------------------------------------------------ glm restatements (f32)
def mat4_identity():
    return np.eye(4, dtype=np.float32)  # m[col][row]


def mat4_mul(a, b):
    """glm operator*(mat4,mat4): Result[c] = ((A0*b[c][0] + A1*b[c][1]) + A2*b[c][2]) + A3*b[c][3]"""
    r = np.zeros((4, 4), np.float32)
    for c in range(4):
        r[c] = ((a[0] * b[c][0] + a[1] * b[c][1]) + a[2] * b[c][2]) + a[3] * b[c][3]
    return r


def mat4_mul_vec4(m, v):
    """glm operator*(mat4,vec4): (m0*v.x + m1*v.y) + (m2*v.z + m3*v.w); v is (N,4) or (4,)"""
    v = np.asarray(v, np.float32)
    if v.ndim == 1:
        return (m[0] * v[0] + m[1] * v[1]) + (m[2] * v[2] + m[3] * v[3])
    return (m[0][None, :] * v[:, 0:1] + m[1][None, :] * v[:, 1:2]) + (m[2][None, :] * v[:, 2:3] + m[3][None, :] * v[:, 3:4])


def mat4_scale(v):
    m = mat4_identity()
    r = np.zeros((4, 4), np.float32)
    r[0] = m[0] * f(v[0]); r[1] = m[1] * f(v[1]); r[2] = m[2] * f(v[2]); r[3] = m[3]
    return r


def mat4_translate(v):
    m = mat4_identity()
    r = m.copy()
    r[3] = ((m[0] * f(v[0]) + m[1] * f(v[1])) + m[2] * f(v[2])) + m[3]
    return r


def quat_to_mat4(q):
    """glm::toMat4(quat) = mat4(mat3_cast(q)); q = (x,y,z,w) as stored by cgltf / glm 0.9.9.8"""
    x, y, z, w = (f(q[0]), f(q[1]), f(q[2]), f(q[3]))
    qxx, qyy, qzz = x * x, y * y, z * z
    qxz, qxy, qyz = x * z, x * y, y * z
    qwx, qwy, qwz = w * x, w * y, w * z
    one, two = f(1), f(2)
    r = mat4_identity()
    r[0][0] = one - two * (qyy + qzz); r[0][1] = two * (qxy + qwz); r[0][2] = two * (qxz - qwy)
    r[1][0] = two * (qxy - qwz); r[1][1] = one - two * (qxx + qzz); r[1][2] = two * (qyz + qwx)
    r[2][0] = two * (qxz + qwy); r[2][1] = two * (qyz - qwx); r[2][2] = one - two * (qxx + qyy)
    return r


def mat4_inverse_transpose(m):
    """glm::inverseTranspose(mat4), gtc/matrix_inverse.inl (glm 0.9.9.8), f32, same association."""
    m = m.astype(np.float32)
    S = [None] * 18
    S[0] = m[2][2] * m[3][3] - m[3][2] * m[2][3]
    S[1] = m[2][1] * m[3][3] - m[3][1] * m[2][3]
    S[2] = m[2][1] * m[3][2] - m[3][1] * m[2][2]
    S[3] = m[2][0] * m[3][3] - m[3][0] * m[2][3]
    S[4] = m[2][0] * m[3][2] - m[3][0] * m[2][2]
    S[5] = m[2][0] * m[3][1] - m[3][0] * m[2][1]
    S[6] = m[1][2] * m[3][3] - m[3][2] * m[1][3]
    S[7] = m[1][1] * m[3][3] - m[3][1] * m[1][3]
    S[8] = m[1][1] * m[3][2] - m[3][1] * m[1][2]
    S[9] = m[1][0] * m[3][3] - m[3][0] * m[1][3]
    S[10] = m[1][0] * m[3][2] - m[3][0] * m[1][2]
    S[11] = m[1][0] * m[3][1] - m[3][0] * m[1][1]
    S[12] = m[1][2] * m[2][3] - m[2][2] * m[1][3]
    S[13] = m[1][1] * m[2][3] - m[2][1] * m[1][3]
    S[14] = m[1][1] * m[2][2] - m[2][1] * m[1][2]
    S[15] = m[1][0] * m[2][3] - m[2][0] * m[1][3]
    S[16] = m[1][0] * m[2][2] - m[2][0] * m[1][2]
    S[17] = m[1][0] * m[2][1] - m[2][0] * m[1][1]
    I = np.zeros((4, 4), np.float32)
    I[0][0] = +(m[1][1] * S[0] - m[1][2] * S[1] + m[1][3] * S[2])
    I[0][1] = -(m[1][0] * S[0] - m[1][2] * S[3] + m[1][3] * S[4])
    I[0][2] = +(m[1][0] * S[1] - m[1][1] * S[3] + m[1][3] * S[5])
    I[0][3] = -(m[1][0] * S[2] - m[1][1] * S[4] + m[1][2] * S[5])
    I[1][0] = -(m[0][1] * S[0] - m[0][2] * S[1] + m[0][3] * S[2])
    I[1][1] = +(m[0][0] * S[0] - m[0][2] * S[3] + m[0][3] * S[4])
    I[1][2] = -(m[0][0] * S[1] - m[0][1] * S[3] + m[0][3] * S[5])
    I[1][3] = +(m[0][0] * S[2] - m[0][1] * S[4] + m[0][2] * S[5])
    I[2][0] = +(m[0][1] * S[6] - m[0][2] * S[7] + m[0][3] * S[8])
    I[2][1] = -(m[0][0] * S[6] - m[0][2] * S[9] + m[0][3] * S[10])
    I[2][2] = +(m[0][0] * S[7] - m[0][1] * S[9] + m[0][3] * S[11])
    I[2][3] = -(m[0][0] * S[8] - m[0][1] * S[10] + m[0][2] * S[11])
    I[3][0] = -(m[0][1] * S[12] - m[0][2] * S[13] + m[0][3] * S[14])
    I[3][1] = +(m[0][0] * S[12] - m[0][2] * S[15] + m[0][3] * S[16])
    I[3][2] = -(m[0][0] * S[13] - m[0][1] * S[15] + m[0][3] * S[17])
    I[3][3] = +(m[0][0] * S[14] - m[0][1] * S[16] + m[0][2] * S[17])
    det = m[0][0] * I[0][0] + m[0][1] * I[0][1] + m[0][2] * I[0][2] + m[0][3] * I[0][3]
    return (I / det).astype(np.float32)


def _node_local(node):
    if "matrix" in node:
        return np.array([f(x) for x in node["matrix"]], np.float32).reshape(4, 4)
    s = node.get("scale", [1, 1, 1]); r = node.get("rotation", [0, 0, 0, 1]); t = node.get("translation", [0, 0, 0])
    return mat4_mul(mat4_mul(mat4_translate(t), quat_to_mat4(r)), mat4_scale(s))


def _traverse(js, node_idx, parent, transforms):
    node = js["nodes"][node_idx]
    m = mat4_mul(parent, _node_local(node))
    if "mesh" in node:
        transforms[node["mesh"]] = (m, mat4_inverse_transpose(m))
    for c in node.get("children", []):
        _traverse(js, c, m, transforms)


def decode_image_bgra(data):
    """texture.cpp:12-54: 4 channels forced, alpha forced to 255, packed b | g<<8 | r<<16 | 255<<24"""
    from PIL import Image
    im = Image.open(io.BytesIO(data)).convert("RGBA")
    a = np.asarray(im, dtype=np.uint32)
    px = a[..., 2] | (a[..., 1] << 8) | (a[..., 0] << 16) | np.uint32(255 << 24)
    return px.astype(np.uint32).reshape(-1), im.width, im.height


def pixel_texture(r, g, b, a):
    """texture.cpp:56-65 Texture::fromPixel (truncating f*255)"""
    r8, g8, b8, a8 = (np.uint32(f(c) * f(255.0)) for c in (r, g, b, a))
    return np.array([b8 | (g8 << 8) | (r8 << 16) | (a8 << 24)], np.uint32), 1, 1


def fnv1a(data):
    h = 2166136261
    for b in data:
        h = ((h ^ b) * 16777619) & 0xFFFFFFFF
    return h


def load_model(path):
    """-> dict(meshes=[{positions,normals,texCoords,indices,tex}], textures=[(pixels,w,h)])"""
    js, buffers = load_container(path)
    nmesh = len(js["meshes"])
    transforms = {i: (mat4_identity(), mat4_identity()) for i in range(nmesh)}
    scene = js["scenes"][js.get("scene", 0)]
    for n in scene["nodes"]:
        _traverse(js, n, mat4_identity(), transforms)

    textures, image_lookup, factor_lookup = [], {}, {}
    meshes = []
    for mi, mesh in enumerate(js["meshes"]):
        M, N = transforms[mi]
        for prim in mesh["primitives"]:
            assert prim.get("mode", 4) == 4
            pbr = js["materials"][prim["material"]].get("pbrMetallicRoughness", {})
            bct = pbr.get("baseColorTexture")
            if bct is not None:
                tex = js["textures"][bct["index"]]
                img_idx = tex["source"]
                if img_idx not in image_lookup:
                    image_lookup[img_idx] = len(textures)
                    img = js["images"][img_idx]
                    if "bufferView" in img:
                        bv = js["bufferViews"][img["bufferView"]]
                        o = bv.get("byteOffset", 0)
                        data = buffers[bv["buffer"]][o:o + bv["byteLength"]]
                    else:
                        data = _load_uri(img["uri"], path)
                    textures.append(decode_image_bgra(data))
                tidx = image_lookup[img_idx]
            else:
                fac = [f(x) for x in pbr.get("baseColorFactor", [1, 1, 1, 1])]
                h = fnv1a(np.array(fac, np.float32).tobytes())
                if h not in factor_lookup:
                    factor_lookup[h] = len(textures)
                    textures.append(pixel_texture(*fac))
                tidx = factor_lookup[h]
            indices = read_accessor(js, buffers, prim["indices"]).astype(np.uint32).reshape(-1)
            lp = unpack_floats(js, buffers, prim["attributes"]["POSITION"])
            ln = unpack_floats(js, buffers, prim["attributes"]["NORMAL"])
            uv = unpack_floats(js, buffers, prim["attributes"]["TEXCOORD_0"])
            p4 = np.concatenate([lp, np.ones((lp.shape[0], 1), np.float32)], axis=1)
            pos = mat4_mul_vec4(M, p4)[:, :3].astype(np.float32)                      # gltf_model.cpp:413
            n4 = np.concatenate([ln, np.zeros((ln.shape[0], 1), np.float32)], axis=1)
            nn = mat4_mul_vec4(N, n4).astype(np.float32)
            # glm::normalize(vec4) then truncated to vec3 (gltf_model.cpp:428): dot over 4 comps
            d = ((nn[:, 0] * nn[:, 0] + nn[:, 1] * nn[:, 1]) + (nn[:, 2] * nn[:, 2] + nn[:, 3] * nn[:, 3])).astype(np.float32)
            inv = (f(1.0) / np.sqrt(d)).astype(np.float32)
            nrm = (nn[:, :3] * inv[:, None]).astype(np.float32)
            meshes.append(dict(positions=pos, normals=nrm, texCoords=uv, indices=indices, tex=tidx))
    # std::sort by texture index (gltf_model.cpp:462); for the stable case this is a stable sort
    meshes.sort(key=lambda m: m["tex"])
    return dict(meshes=meshes, textures=textures)


def flatten(model):
    """flattened_model.cpp:22-43 -> positions (N,9), normals (N,9), texCoords (N,6), texIdx (N,)"""
    P, Nn, T, I = [], [], [], []
    for m in model["meshes"]:
        idx = m["indices"].reshape(-1, 3)
        P.append(m["positions"][idx].reshape(-1, 9))
        Nn.append(m["normals"][idx].reshape(-1, 9))
        T.append(m["texCoords"][idx].reshape(-1, 6))
        I.append(np.full(idx.shape[0], m["tex"], np.uint32))
    return (np.concatenate(P).astype(np.float32), np.concatenate(Nn).astype(np.float32),
            np.concatenate(T).astype(np.float32), np.concatenate(I))


def gpu_layout(positions9, normals9, uvs6, tex_idx):
    """pt_format.cpp:58-75 -> PositionAttribute (N,12) f32 and VertexAttributes (N,20) as u32 view"""
    n = positions9.shape[0]
    pa = np.zeros((n, 12), np.float32)
    pa[:, 0:3] = positions9[:, 0:3]; pa[:, 4:7] = positions9[:, 3:6]; pa[:, 8:11] = positions9[:, 6:9]
    va = np.zeros((n, 20), np.float32)
    va[:, 0:3] = normals9[:, 0:3]; va[:, 4:7] = normals9[:, 3:6]; va[:, 8:11] = normals9[:, 6:9]
    va[:, 12:18] = uvs6
    vau = va.view(np.uint32)
    vau[:, 18] = tex_idx
    return pa, vau.view(np.float32)


def flatten_textures(textures):
    """reference_path_tracer.cpp:210-245 -> descriptors (T,3) u32 {w,h,offset}, texels u32"""
    descs, blob, off = [], [], 0
    for px, w, h in textures:
        descs.append((w, h, off))
        blob.append(px)
        off += px.size
    return np.array(descs, np.uint32).reshape(-1, 3), np.concatenate(blob).astype(np.uint32)
