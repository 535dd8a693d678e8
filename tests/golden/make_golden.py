#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.  Run in the build container (needs /root/reference for the
reference-compiled sky model under oracle/_ref; everything else comes from the oracle).

  sky_ref.npz     outputs of the REFERENCE's own hw_skymodel.c (oracle/_ref/libhwsky_ref.so):
                  sky_state_new for a grid of (elevation, turbidity, albedo) and sky_state_radiance
                  samples.  These pin both the oracle's and the product's sky restatement.
  duck_golden.npz oracle outputs on tests/golden/Duck.glb: sha256 of the node array, per-pixel
                  nodesVisited for the bvh-visualizer camera at 256x256, hit/t of the reference's
                  own test grid (src/tests/bvh.cpp:76-101), summary counts.  The counts equal the
                  probe numbers recorded in SURVEY.md 8(c), which came from the reference's C++
                  compiled unmodified (against a glm stand-in) by the survey session.
  duck_render_golden.npz  oracle radiance for a few 16x16 crops of config 2 (Duck.pt, 800x600,
                  default camera/sky) at 8 spp / 4 bounces, f32 sums.
  duck_render_golden_64spp.npz  the same at config 2's FULL sample count (64 spp): two 16x16 crops.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import gltf_ref, orc  # noqa: E402


def sky():
    ref = orc.RefSky()
    rng = np.random.default_rng(20240807)
    cases, states, rcs = [], [], []
    for turb in [1.0, 1.25, 2.0, 3.7, 5.5, 9.0, 9.99, 10.0]:
        for elev in [0.0, 0.1, 0.5, float(np.float32(0.5 * np.float32(np.pi)) - orc.degrees_to_radians(30.0)), 1.3, 1.5707964]:
            for alb in [(1, 1, 1), (0, 0, 0), (0.3, 0.5, 0.9)]:
                rc, st = ref.state_new(elev, turb, alb)
                cases.append([elev, turb, *alb]); states.append(st); rcs.append(rc)
    bad = [[-0.1, 1, 1, 1, 1], [0.2, 0.5, 1, 1, 1], [0.2, 11, 1, 1, 1], [0.2, 2, 1, 1.5, 1], [4.0, 2, 1, 1, 1]]
    bad_rc = [ref.state_new(c[0], c[1], c[2:])[0] for c in bad]
    samples = []
    for i in rng.integers(0, len(states), 200):
        th = np.float32(rng.uniform(0, 1.57)); g = np.float32(rng.uniform(0, 3.14)); ch = int(rng.integers(0, 3))
        if rng.uniform() < 0.1:
            g = np.float32(rng.uniform(0, 0.0044))  # inside the solar disk
        samples.append([i, th, g, ch, ref.radiance(states[i], th, g, ch)])
    np.savez_compressed(os.path.join(HERE, "sky_ref.npz"), cases=np.array(cases, np.float32), states=np.array(states, np.float32),
                        rcs=np.array(rcs, np.int32), bad=np.array(bad, np.float32), bad_rc=np.array(bad_rc, np.int32),
                        samples=np.array(samples, np.float64))


def duck_scene():
    m = gltf_ref.load_model(os.path.join(HERE, "Duck.glb"))
    P, N, T, I = gltf_ref.flatten(m)
    nodes, idx, depth = orc.build_bvh(P)
    tris = orc.reorder(P, idx)
    pa, va = gltf_ref.gpu_layout(tris, orc.reorder(N, idx), orc.reorder(T, idx), orc.reorder(I, idx))
    descs, texels = gltf_ref.flatten_textures(m["textures"])
    return m, P, nodes, idx, depth, tris, pa, va, descs, texels


def duck():
    m, P, nodes, idx, depth, tris, pa, va, descs, texels = duck_scene()
    cam = orc.bvh_visualizer_camera(nodes, np.float32(1.0))
    viz = orc.bvh_visualize(nodes, tris, cam, 256, 256)
    cam720 = orc.bvh_visualizer_camera(nodes, np.float32(np.float32(1280) / np.float32(720)))
    viz720 = orc.bvh_visualize(nodes, tris, cam720, 1280, 720)
    # the reference's own test scenario, src/tests/bvh.cpp:46-101
    tcam = orc.bvh_test_camera(tris)
    rays = np.array([orc.generate_camera_ray(tcam, np.float32(i) / np.float32(64), np.float32(j) / np.float32(64))
                     for i in range(64) for j in range(64)], np.float32)
    r = orc.intersect_bvh_batch(nodes, tris, rays, 1000.0)
    leaves = nodes[nodes["triangleCount"] > 0]
    np.savez_compressed(
        os.path.join(HERE, "duck_golden.npz"),
        nodes_sha256=hashlib.sha256(nodes.tobytes()).hexdigest(), num_nodes=len(nodes), depth=depth,
        leaf_hist=np.bincount(leaves["triangleCount"]), root=nodes[:1],
        tex_sha256=hashlib.sha256(texels.tobytes()).hexdigest(), tex_dims=descs,
        pos48_sha256=hashlib.sha256(pa.tobytes()).hexdigest(), attr80_sha256=hashlib.sha256(va.tobytes()).hexdigest(),
        viz256_nodes_visited=viz["nodesVisited"].astype(np.uint16), viz256_hit=np.packbits(viz["hit"]),
        viz256_tri_tests=int(viz["triTests"].sum()), viz256_stack_high=int(viz["stackHigh"].max()),
        viz720_sum=int(viz720["nodesVisited"].sum()), viz720_max=int(viz720["nodesVisited"].max()), viz720_hits=int(viz720["hit"].sum()),
        viz720_row_sums=viz720["nodesVisited"].reshape(720, 1280).sum(axis=1).astype(np.uint32),
        grid_rays=rays, grid_hit=r["hit"], grid_t=r["t"], grid_tri=r["tri"])
    print("nodes", len(nodes), "sum256", viz["nodesVisited"].sum(), "sum720", viz720["nodesVisited"].sum(), "grid hits", r["hit"].sum())


def duck_render():
    m, P, nodes, idx, depth, tris, pa, va, descs, texels = duck_scene()
    sc = orc.OracleScene(nodes, pa, va, descs, texels)
    W, H, spp, bounces = 800, 600, 8, 4
    rp = orc.make_render_params(W, H, orc.default_pt_camera(W, H), spp, bounces, 0.25, orc.aligned_sky_state())
    crops = [(392, 292), (300, 200), (100, 500), (700, 60), (420, 330), (0, 0), (784, 584)]
    out = []
    for (x0, y0) in crops:
        img, st = orc.render(sc, rp, 0, spp, x0, y0, x0 + 16, y0 + 16)
        out.append(img[y0:y0 + 16, x0:x0 + 16, :3].copy())
        print((x0, y0), img[y0:y0 + 16, x0:x0 + 16, :3].mean(), st.as_dict())
    np.savez_compressed(os.path.join(HERE, "duck_render_golden.npz"), crops=np.array(crops, np.int32), sums=np.array(out, np.float32),
                        width=W, height=H, spp=spp, bounces=bounces)


def duck_render64():
    m, P, nodes, idx, depth, tris, pa, va, descs, texels = duck_scene()
    sc = orc.OracleScene(nodes, pa, va, descs, texels)
    W, H, spp, bounces = 800, 600, 64, 4          # BASELINE.json config 2
    rp = orc.make_render_params(W, H, orc.default_pt_camera(W, H), spp, bounces, 0.25, orc.aligned_sky_state())
    crops = [(392, 292), (300, 200)]
    out = []
    for (x0, y0) in crops:
        img, st = orc.render(sc, rp, 0, spp, x0, y0, x0 + 16, y0 + 16)
        out.append(img[y0:y0 + 16, x0:x0 + 16, :3].copy())
        print((x0, y0), img[y0:y0 + 16, x0:x0 + 16, :3].mean(), st.as_dict())
    np.savez_compressed(os.path.join(HERE, "duck_render_golden_64spp.npz"), crops=np.array(crops, np.int32), sums=np.array(out, np.float32),
                        width=W, height=H, spp=spp, bounces=bounces)


if __name__ == "__main__":
    which = sys.argv[1:] or ["sky", "duck", "duck_render", "duck_render64"]
    for w in which:
        globals()[w]()
