"""GPU BVH build (-m gpu): rf_build_bvh_gpu against the oracle's restatement of the reference builder
(src/common/bvh.cpp:81-291) and against the product's host builder.

Bar: the node array is BYTE-IDENTICAL (memcmp) and so is the triangle permutation: the order inside a
multi-triangle leaf decides closest-hit ties between coincident triangles (`t < tmax` is strict,
wgsl:508), so the GPU builder reproduces the permutation of libstdc++'s std::partition /
std::nth_element that the host builder and the oracle inherit (the reference's
own order is whatever ITS standard library leaves).  A scene baked with either builder is the same .pt,
byte for byte.
"""
import numpy as np
import pytest

import rayfinder_amd as rf
from conftest import bits
from oracle import orc

pytestmark = pytest.mark.gpu


def _check(tris36, label):
    tris36 = np.ascontiguousarray(tris36, np.float32).reshape(-1, 9)
    n = len(tris36)
    ref_nodes, ref_idx, ref_depth = orc.build_bvh(tris36)           # oracle (C restatement of the reference)
    host_nodes, host_idx, host_depth = rf.build_bvh(tris36)          # product, host
    gpu_nodes, gpu_idx, gpu_depth, ms = rf.build_bvh_gpu(tris36)     # product, GPU
    assert len(gpu_nodes) == len(ref_nodes) == len(host_nodes), (label, len(gpu_nodes), len(ref_nodes))
    assert gpu_nodes.tobytes() == host_nodes.tobytes(), label
    assert gpu_nodes.tobytes() == np.ascontiguousarray(ref_nodes).tobytes(), label
    assert gpu_depth == ref_depth == host_depth, label
    # the same permutation as the host builder and the oracle (both follow libstdc++), leaf interiors included
    gpu_idx = np.asarray(gpu_idx, np.int64); ref_idx = np.asarray(ref_idx, np.int64)
    assert np.array_equal(np.sort(gpu_idx), np.arange(n)), label
    assert np.array_equal(gpu_idx, np.asarray(host_idx, np.int64)), label
    assert np.array_equal(gpu_idx, ref_idx), label
    leaf_of_pos = np.full(n, -1, np.int64)
    leaves = np.nonzero(gpu_nodes["triangleCount"] > 0)[0]
    for li in leaves[gpu_nodes["triangleCount"][leaves] > 1]:
        o, c = int(gpu_nodes["trianglesOffset"][li]), int(gpu_nodes["triangleCount"][li])
        leaf_of_pos[o:o + c] = li
    single = leaf_of_pos < 0
    # single-triangle leaves: position identical; multi-triangle leaves: same leaf
    inv_g = np.empty(n, np.int64); inv_g[gpu_idx] = np.arange(n)     # position -> source triangle
    inv_r = np.empty(n, np.int64); inv_r[ref_idx] = np.arange(n)
    assert np.array_equal(inv_g[single], inv_r[single]), label
    for li in np.unique(leaf_of_pos[~single]):
        pos = np.nonzero(leaf_of_pos == li)[0]
        assert set(inv_g[pos]) == set(inv_r[pos]), (label, li)
    return gpu_nodes, gpu_idx, ms


def test_duck_nodes_byte_identical(duck_oracle):
    nodes, idx, ms = _check(duck_oracle.P, "duck")
    assert len(nodes) == 8383           # the survey probe's node count


def test_atrium_nodes_byte_identical_and_build_time():
    from rayfinder_amd import scenes
    pt, info = scenes.atrium()
    tris = pt.arrays()["bvhPositionAttributes"]
    rng = np.random.default_rng(5)
    tris = tris[rng.permutation(len(tris))]          # undo the leaf order the .pt stores
    nodes, idx, ms = _check(tris, "atrium")
    # second build: same bytes (the build is deterministic, atomics notwithstanding)
    nodes2, idx2, _, ms2 = rf.build_bvh_gpu(tris)
    assert nodes2.tobytes() == nodes.tobytes() and np.array_equal(idx2, idx)
    print(f"atrium: {len(tris)} triangles -> {len(nodes)} nodes, GPU build {ms:.2f} ms / {ms2:.2f} ms")


@pytest.mark.parametrize("case", ["random", "duplicates", "coplanar", "tiny", "one", "two", "grid", "big_leaf", "huge"])
def test_synthetic_soups(case):
    rng = np.random.default_rng(hash(case) % 2**32)
    if case == "random":
        c = rng.uniform(-10, 10, (20000, 1, 3)); tris = (c + rng.normal(0, 0.3, (20000, 3, 3))).reshape(-1, 9)
    elif case == "duplicates":     # many identical triangles: degenerate centroid extents, forced splits, big leaves
        base = rng.uniform(-1, 1, (50, 9)); tris = base[rng.integers(0, 50, 30000)]
    elif case == "coplanar":       # zero surface area boxes
        tris = rng.uniform(-1, 1, (5000, 3, 3)); tris[:, :, 1] = 0.25; tris[:1000, :, 0] = 0.5; tris = tris.reshape(-1, 9)
    elif case == "tiny":
        tris = rng.uniform(-1, 1, (37, 9))
    elif case == "one":
        tris = rng.uniform(-1, 1, (1, 9))
    elif case == "two":
        tris = rng.uniform(-1, 1, (2, 9))
    elif case == "grid":           # regular lattice: ties in bucket costs and centroid coordinates
        g = np.stack(np.meshgrid(np.arange(40), np.arange(40), np.arange(8), indexing="ij"), -1).reshape(-1, 1, 3).astype(np.float32)
        tris = (g + np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)).reshape(-1, 9)
    elif case == "huge":           # areas overflow to inf: no finite SAH cost -> one leaf of 600 (the reference asserts)
        tris = rng.uniform(-1, 1, (600, 9)) * 1e20
    else:                          # > 255 triangles with identical centroids along the split axis cannot be split
        tris = np.tile(rng.uniform(-1, 1, (1, 9)), (1000, 1))
        tris = np.concatenate([tris, rng.uniform(-5, 5, (3000, 9))])
    _check(tris.astype(np.float32), case)


def test_traversal_through_gpu_built_tree(duck_oracle):
    """Render-path parity does not care who built the tree: same hits, t, u, v through a scene whose BVH
    came from the GPU builder."""
    P = duck_oracle.P
    nodes, idx, depth, _ = rf.build_bvh_gpu(P)
    ref_nodes, ref_idx, _ = orc.build_bvh(P)
    tris_g = orc.reorder(P, idx); tris_r = orc.reorder(P, ref_idx)
    rng = np.random.default_rng(11)
    lo, hi = ref_nodes[0]["min"].astype(np.float64), ref_nodes[0]["max"].astype(np.float64)
    o = rng.uniform(lo - 0.5 * (hi - lo), hi + 0.5 * (hi - lo), (20000, 3)); d = rng.uniform(lo, hi, (20000, 3)) - o
    rays = np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], 1).astype(np.float32)
    a = orc.intersect_bvh_batch(nodes, tris_g, rays, 10000.0)
    b = orc.intersect_bvh_batch(ref_nodes, tris_r, rays, 10000.0)
    assert np.array_equal(a["hit"], b["hit"]) and np.array_equal(bits(a["t"]), bits(b["t"]))
    assert np.array_equal(a["nodesVisited"], b["nodesVisited"])


def test_coincident_triangles_with_different_attributes_bake_and_render_identically():
    """The `duplicates` situation that makes leaf order observable: stacks of coincident triangles whose copies carry
    different textures.  Host bake and GPU bake must be the same .pt byte for byte, and the GPU render of the
    GPU-baked scene must equal the oracle's render of the host-baked one bit for bit."""
    from conftest import oracle_scene_from_pt
    rng = np.random.default_rng(77)
    base = rng.uniform(-1, 1, (40, 3, 3)).astype(np.float32)
    base[:, :, 2] *= 0.2                                        # a shallow slab seen from above
    pick = rng.integers(0, 40, 6000)
    P = base[pick].reshape(-1, 9)
    nrm = np.cross(base[:, 1] - base[:, 0], base[:, 2] - base[:, 0]); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    N = np.repeat(nrm[pick][:, None, :], 3, 1).reshape(-1, 9).astype(np.float32)
    T = rng.uniform(0, 1, (6000, 6)).astype(np.float32)
    I = rng.integers(0, 5, 6000).astype(np.uint32)             # coincident copies get different textures
    textures = [(np.full(16, 0xFF000000 | int(c), np.uint32), 4, 4) for c in (0xFF0000, 0x00FF00, 0x0000FF, 0xFFFF00, 0xFFFFFF)]
    pt_host = rf.PtFormat.from_triangles(P, N, T, I, textures)
    rf.set_bake_bvh_builder(0)
    try:
        pt_gpu = rf.PtFormat.from_triangles(P, N, T, I, textures)
    finally:
        rf.set_bake_bvh_builder(None)
    assert max(pt_host.arrays()["bvhNodes"]["triangleCount"]) > 1
    assert pt_gpu.serialize() == pt_host.serialize()
    W, H, spp, bounces = 96, 64, 4, 3
    cam = rf.create_camera((0.2, 0.3, 3.0), (0.0, 0.0, 0.0), 0.0, 1.0, np.radians(60.0), W / H)
    params = rf.make_render_parameters(W, H, cam, spp, bounces, rf.make_sky(), 0.25)
    r = rf.ReferencePathTracer(params, pt_gpu.scene())
    r.render(spp)
    img = r.read_accumulation()[0]
    r.close()
    sc, _ = oracle_scene_from_pt(pt_host)
    rp = orc.make_render_params(W, H, rf.camera_to_array(cam), spp, bounces, 0.25, rf.aligned_sky_state(rf.make_sky()))
    ref, _ = orc.render(sc, rp, 0, spp)
    assert np.array_equal(bits(img[..., :3]), bits(ref[..., :3]))
    assert (img[..., :3] > 0).any()


def test_bake_with_the_gpu_builder_renders_the_same_image(duck_pt):
    """Duck.glb baked with the GPU builder: the same .pt as the host bake byte for byte, and (therefore) the
    path tracer's accumulation image is bit-identical."""
    from conftest import DUCK
    rf.set_bake_bvh_builder(0)
    try:
        pt_gpu = rf.PtFormat.from_gltf(DUCK)
    finally:
        rf.set_bake_bvh_builder(None)
    a, b = pt_gpu.arrays(), duck_pt.arrays()
    assert a["bvhNodes"].tobytes() == b["bvhNodes"].tobytes()
    assert pt_gpu.serialize() == duck_pt.serialize()
    W, H, spp, bounces = 160, 120, 4, 3
    imgs = []
    for pt in (pt_gpu, duck_pt):
        r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, bounces, rf.make_sky(), 0.25), pt.scene())
        r.render(spp)
        imgs.append(r.read_accumulation()[0])
        r.close()
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))
