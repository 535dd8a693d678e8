"""GPU BVH build (-m gpu): rf_build_bvh_gpu against the oracle's restatement of the reference builder
(src/common/bvh.cpp:81-291) and against the product's host builder.

Bar: the node array is BYTE-IDENTICAL (memcmp).  The triangle permutation may differ only inside
multi-triangle leaves (the reference's order there is whatever its
standard library's std::partition / std::nth_element leave), so leaves are compared as sets, and
traversal through the GPU-built tree gives the same hits and t bit for bit.
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
    # a permutation, and the same triangle SET in every leaf
    gpu_idx = np.asarray(gpu_idx, np.int64); ref_idx = np.asarray(ref_idx, np.int64)
    assert np.array_equal(np.sort(gpu_idx), np.arange(n)), label
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


@pytest.mark.parametrize("case", ["random", "duplicates", "coplanar", "tiny", "one", "two", "grid", "big_leaf"])
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


def test_bake_with_the_gpu_builder_renders_the_same_image(duck_pt):
    """Duck.glb baked with the GPU builder: same node bytes as the host bake, and the path tracer's
    accumulation image is bit-identical (Duck's multi-triangle leaves hold no exact ties)."""
    from conftest import DUCK
    rf.set_bake_bvh_builder(0)
    try:
        pt_gpu = rf.PtFormat.from_gltf(DUCK)
    finally:
        rf.set_bake_bvh_builder(None)
    a, b = pt_gpu.arrays(), duck_pt.arrays()
    assert a["bvhNodes"].tobytes() == b["bvhNodes"].tobytes()
    W, H, spp, bounces = 160, 120, 4, 3
    imgs = []
    for pt in (pt_gpu, duck_pt):
        r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, bounces, rf.make_sky(), 0.25), pt.scene())
        r.render(spp)
        imgs.append(r.read_accumulation()[0])
        r.close()
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))
