#!/usr/bin/env python3
"""What would a GLOBAL order of a deep bounce's rays be worth to the closest-hit launch?  (round 5: the one lever left that the kernel itself cannot pull)

Builds bounce-b ray sets the way the renderer meets them -- tiles of 32 x 32 pixels, `spp` samples of a pixel next to each other, misses compacted away, a cosine
lobe about the geometric normal at every hit -- and times the render path's own closest-hit kernel (query_variant 2, half-precision quad records; RF_DEBUG_QUERY_MS:
HIP events around the launch alone) on the SAME rays in several orders:

    natural        queue order as the renderer has it (kShade's tile-local sort is not applied: it regroups 1024 neighbours only)
    tile-sort      kShade<SORTED>'s order: every run of 1024 entries sorted by the triangle the ray starts on
    tri            all rays sorted by the triangle they start on (triangles are in BVH leaf order: an order by region of the scene)
    tri/B          ... binned only: by triangle range of B triangles, queue order inside a bin (what a one-pass counting sort gives)
    tri/B+oct      ... and by direction octant inside the bin
    morton+oct     30-bit Morton code of the origin, then direction octant

  RF_DEBUG_QUERY_MS=1 python tools/gpu_sort_potential.py [tiles = 128] [spp = 64] [scene detail = plain] [scene scale = 1]

RF_SORT_INDIRECT=1 (round 6): the rays STAY in kShade's order (every run of 1 024 sorted by triangle: where the renderer's path state would sit) and only an index list is put into
the order under test -- the launch visits ray list[i] at position i (RF_DEBUG_QUERY_LIST), gathering origin / direction from and scattering its hit record to the ray's own place:
what a global order costs when nothing but a list of queue positions is sorted.
"""
import os, sys, re, subprocess, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

if os.environ.get("RF_SORT_POTENTIAL_CHILD") != "1":
    # the C library reports the launch time on stderr: run the measurement as a child and parse it
    env = dict(os.environ, RF_SORT_POTENTIAL_CHILD="1", RF_DEBUG_QUERY_MS="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
    times = [float(m.group(2)) for m in re.finditer(r"\[rf-query\] closest-hit launch: (\d+) rays, ([0-9.]+) ms", p.stderr)]
    labels = [l[6:] for l in p.stdout.splitlines() if l.startswith("LABEL ")]
    for l in p.stdout.splitlines():
        if not l.startswith("LABEL "): print(l)
    if p.returncode != 0 or len(times) != len(labels):
        print(p.stderr[-3000:]); print("child failed / label count mismatch", p.returncode, len(times), len(labels)); sys.exit(1)
    best = {}
    for lab, ms in zip(labels, times):
        best[lab] = min(best.get(lab, 1e30), ms)
    base = {}
    for lab, ms in best.items():
        bounce, rays, order = lab.split("|")
        if order == "natural" or (order == "tile-sort" and bounce not in base): base[bounce] = ms
    for lab, ms in best.items():
        bounce, rays, order = lab.split("|")
        print(f"bounce {bounce}  {int(rays) / 1e6:6.2f} M rays  {order:14s} {ms:8.3f} ms  {int(rays) / ms / 1e6:7.2f} Grays/s   x{base[bounce] / ms:5.3f}")
    sys.exit(0)

import rayfinder_amd as rf
from rayfinder_amd import scenes

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 128
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
detail = sys.argv[3] if len(sys.argv) > 3 else "plain"
scale = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if scale > 1: rf.set_bake_bvh_builder(0)
pt, info = scenes.atrium(scale, detail)
W, H = 1920, 1080
cam = rf.fly_camera(W, H)
c = rf.camera_to_array(cam)
origin, llc, hor, ver = c[0:3], c[3:6], c[6:9], c[9:12]
tri_pos = pt.arrays()["trianglePositionAttributes"]
p0, p1, p2 = tri_pos[:, 0:3], tri_pos[:, 4:7], tri_pos[:, 8:11]
gn = np.cross(p1 - p0, p2 - p0).astype(np.float32)
gn /= np.maximum(np.linalg.norm(gn, axis=1, keepdims=True), 1e-30)
print(f"scene: {len(tri_pos)} triangles; {tiles} tiles x 1024 px x {spp} spp")

r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, cam, 1, 1, rf.make_sky(), 0.25), pt.scene())
r.set_option("query_variant", 2)
r.set_option("query_compact", 4 if scale == 1 else 5)

rng = np.random.default_rng(7)
tx, ty = W // 32, H // 32
chosen = rng.choice(tx * ty, size=min(tiles, tx * ty), replace=False); chosen.sort()
px = []
for t in chosen:
    x0, y0 = (t % tx) * 32, (t // tx) * 32
    yy, xx = np.mgrid[y0:y0 + 32, x0:x0 + 32]
    px.append(np.stack([xx.ravel(), yy.ravel()], 1))
px = np.repeat(np.concatenate(px), spp, axis=0).astype(np.float32)
n = len(px)
jit = rng.random((n, 2), dtype=np.float32)
s = (px[:, 0] + jit[:, 0]) / W; t = 1.0 - (px[:, 1] + jit[:, 1]) / H
d = llc[None, :] + s[:, None] * hor[None, :] + t[:, None] * ver[None, :] - origin[None, :]
d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
o = np.broadcast_to(origin, d.shape).astype(np.float32)
start_tri = np.zeros(n, np.uint32)


INDIRECT = os.environ.get("RF_SORT_INDIRECT") == "1"
LIST_PATH = f"/tmp/rf_sort_list_{os.getpid()}.bin"


def timed(o, d, label, order=None):
    """order (INDIRECT): the rays are passed as they are and visited in this order through an index list"""
    rays = np.ascontiguousarray(np.concatenate([o, d], 1), np.float32)
    out = None
    if order is not None:
        np.ascontiguousarray(order, np.uint32).tofile(LIST_PATH)
        os.environ["RF_DEBUG_QUERY_LIST"] = LIST_PATH
    try:
        for _ in range(2):
            print(f"LABEL {label}", flush=True)
            out = r.intersect_rays(rays, 10000.0)
    finally:
        os.environ.pop("RF_DEBUG_QUERY_LIST", None)
    return out


def morton3(q):
    def spread(v):
        v = v.astype(np.uint64) & 0x3FF
        v = (v | (v << 16)) & 0x30000FF
        v = (v | (v << 8)) & 0x300F00F
        v = (v | (v << 4)) & 0x30C30C3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


for bounce in range(1, 6):
    n = len(o)
    if bounce >= 3:
        octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << 1) | ((d[:, 2] < 0).astype(np.uint64) << 2))
        orders = {"natural": np.arange(n)}
        key = start_tri.astype(np.uint64)
        run = np.arange(n, dtype=np.uint64) // 1024
        orders["tile-sort"] = np.argsort((run << 32) | key, kind="stable")
        orders["tri"] = np.argsort(key, kind="stable")
        for B in (16, 256):
            orders[f"tri/{B}"] = np.argsort(key // B, kind="stable")
        orders["tri/16+oct"] = np.argsort(((key // 16) << 3) | octant, kind="stable")
        orders["tri/256+oct"] = np.argsort(((key // 256) << 3) | octant, kind="stable")
        lo, hi = o.min(0), o.max(0)
        q = np.clip((o - lo) / np.maximum(hi - lo, 1e-9) * 1023.0, 0, 1023).astype(np.uint32)
        orders["morton+oct"] = np.argsort((morton3(q) << 3) | octant, kind="stable")
        ref = None
        if INDIRECT:
            base = orders["tile-sort"]                       # where the path state sits: kShade's order
            ob, db = o[base], d[base]
            pos_of = np.empty(n, np.int64); pos_of[base] = np.arange(n)      # natural index -> position in the tile-sorted arrays
            for name, perm in orders.items():
                if name == "natural": continue
                out = timed(ob, db, f"{bounce}|{n}|{name}", order=pos_of[perm])            # visit order `perm`, expressed in positions of the tile-sorted arrays
                tri_nat = out["tri"][pos_of]
                if ref is None: ref = tri_nat
                elif not np.array_equal(ref, tri_nat): print("RESULT MISMATCH under order", name)
            out = {k: (v[pos_of] if hasattr(v, "__len__") and len(v) == n else v) for k, v in out.items()}
        else:
          for name, perm in orders.items():
            out = timed(o[perm], d[perm], f"{bounce}|{n}|{name}")
            inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
            tri_nat = out["tri"][inv]
            if ref is None: ref = tri_nat
            elif not np.array_equal(ref, tri_nat): print("RESULT MISMATCH under order", name)
          out = {k: (v[inv] if hasattr(v, "__len__") and len(v) == n else v) for k, v in out.items()}
    else:
        out = timed(o, d, f"{bounce}|{n}|natural")
    hit = out["tri"] != 0xFFFFFFFF
    print(f"bounce {bounce}: {n} rays, {hit.mean() * 100:.1f} % hit")
    tri = out["tri"][hit]
    nrm = gn[tri]
    flip = np.einsum("ij,ij->i", nrm, d[hit]) > 0
    nrm = np.where(flip[:, None], -nrm, nrm)
    u1, u2 = rng.random(len(tri), dtype=np.float32), rng.random(len(tri), dtype=np.float32)
    rr, phi = np.sqrt(u1), 2 * np.pi * u2
    lx, ly, lz = rr * np.cos(phi), rr * np.sin(phi), np.sqrt(np.maximum(0, 1 - u1))
    a = np.where(np.abs(nrm[:, 0:1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    bu = np.cross(nrm, a); bu /= np.linalg.norm(bu, axis=1, keepdims=True)
    bv = np.cross(nrm, bu)
    d = (lx[:, None] * bu + ly[:, None] * bv + lz[:, None] * nrm).astype(np.float32)
    o = (out["p"][hit] + 1e-4 * nrm).astype(np.float32)
    start_tri = tri
r.close()
