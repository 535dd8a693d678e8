"""Deterministic synthetic scenes.

`atrium()` is the stand-in for the reference's assets/Sponza.glb, which is NOT in the reference
mount (/root/reference/.MISSING_LARGE_BLOBS) and cannot be fetched.  It has Sponza's character as
far as the hot path is concerned: a long two-storey colonnaded court open to the sky, ~262 k
triangles (Sponza: 262 267), 25 BGRA8 textures totalling ~96 MiB, tiling UVs, smooth-shaded
columns and draperies, and it encloses the reference's default camera pose (1.22, 1.25, -1.25).
Every number reported on it is labelled "synthetic atrium".  A user-supplied Sponza.pt/.glb is
used instead when given (bench.py --scene).

Geometry uses float64 arithmetic with +,-,*,/ and sqrt only (own sin/cos polynomial), textures use
integer hashing, so the scene is bit-reproducible; `scene_digest()` is printed with results.
"""
import hashlib

import numpy as np

from . import PtFormat


# ----------------------------------------------------------------------------- deterministic math
def _sincos(theta):
    """sin, cos for float64 arrays with plain arithmetic (range reduced to [-pi/4, pi/4])."""
    theta = np.asarray(theta, np.float64)
    two_over_pi = 0.6366197723675814
    k = np.floor(theta * two_over_pi + 0.5)
    r = theta - k * 1.5707963267948966 - k * 6.123233995736766e-17
    r2 = r * r
    s = r * (1.0 + r2 * (-1.0 / 6 + r2 * (1.0 / 120 + r2 * (-1.0 / 5040 + r2 * (1.0 / 362880 + r2 * (-1.0 / 39916800 + r2 / 6227020800.0))))))
    c = 1.0 + r2 * (-0.5 + r2 * (1.0 / 24 + r2 * (-1.0 / 720 + r2 * (1.0 / 40320 + r2 * (-1.0 / 3628800 + r2 * (1.0 / 479001600 - r2 / 87178291200.0))))))
    q = (k.astype(np.int64) % 4 + 4) % 4
    sin = np.where(q == 0, s, np.where(q == 1, c, np.where(q == 2, -s, -c)))
    cos = np.where(q == 0, c, np.where(q == 1, -s, np.where(q == 2, -c, s)))
    return sin, cos


def _hash32(x):
    x = np.asarray(x, np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def _value_noise(w, h, cell, seed):
    """Integer lattice value noise in [0, 255] (bilinear, fixed point), tiles with period w,h."""
    y, x = np.meshgrid(np.arange(h, dtype=np.uint32), np.arange(w, dtype=np.uint32), indexing="ij")
    cx, cy = x // cell, y // cell
    fx, fy = (x % cell).astype(np.uint32), (y % cell).astype(np.uint32)
    nx, ny = np.uint32(max(w // cell, 1)), np.uint32(max(h // cell, 1))

    def lat(ix, iy):
        return (_hash32((ix % nx) * np.uint32(73856093) ^ (iy % ny) * np.uint32(19349663) ^ np.uint32(seed)) & np.uint32(255)).astype(np.uint32)

    a, b = lat(cx, cy), lat(cx + 1, cy)
    c, d = lat(cx, cy + 1), lat(cx + 1, cy + 1)
    top = a * (cell - fx) + b * fx
    bot = c * (cell - fx) + d * fx
    return ((top * (cell - fy) + bot * fy) // np.uint32(cell * cell)).astype(np.uint32)


def _texture(kind, size, seed):
    """-> u32 BGRA pixels (b | g<<8 | r<<16 | 255<<24), row-major, (w, h)"""
    w, h = size
    n1 = _value_noise(w, h, max(w // 8, 1), seed)
    n2 = _value_noise(w, h, max(w // 64, 1), seed + 101)
    y, x = np.meshgrid(np.arange(h, dtype=np.uint32), np.arange(w, dtype=np.uint32), indexing="ij")
    base = np.array(kind["rgb"], np.uint32)
    if kind["pattern"] == "brick":
        bw, bh = max(w // 8, 2), max(h // 16, 2)
        row = y // bh
        xx = x + (row % 2) * (bw // 2)
        mortar = ((xx % bw) < max(bw // 16, 1)) | ((y % bh) < max(bh // 8, 1))
        brick_id = _hash32((xx // bw) * np.uint32(7919) + row * np.uint32(104729) + np.uint32(seed)) & np.uint32(63)
        shade = 160 + brick_id + n2 // 8
        shade = np.where(mortar, 120 + n2 // 4, shade)
    elif kind["pattern"] == "stripes":
        period = max(w // 16, 2)
        shade = np.where((x // period) % 2 == 0, 235, 150) + n2 // 16 - n1 // 16
    elif kind["pattern"] == "tiles":
        tw = max(w // 4, 2)
        edge = ((x % tw) < max(tw // 32, 1)) | ((y % tw) < max(tw // 32, 1))
        checker = ((x // tw) + (y // tw)) % 2
        shade = np.where(edge, 90, 170 + checker * 50) + n1 // 8
    else:  # marble / plaster
        shade = 150 + n1 // 4 + n2 // 8
    shade = np.clip(shade, 0, 255).astype(np.uint32)
    r = np.clip(base[0] * shade // 255, 0, 255)
    g = np.clip(base[1] * shade // 255, 0, 255)
    b = np.clip(base[2] * shade // 255, 0, 255)
    return (b | (g << np.uint32(8)) | (r << np.uint32(16)) | np.uint32(255 << 24)).astype(np.uint32).reshape(-1), w, h


_TEXTURE_KINDS = [
    # 0-3: structural stone
    dict(pattern="brick", rgb=(214, 196, 170)), dict(pattern="marble", rgb=(225, 220, 205)),
    dict(pattern="tiles", rgb=(200, 185, 160)), dict(pattern="marble", rgb=(180, 170, 150)),
    # 4-9: draperies (saturated, as Sponza's red/green/blue curtains)
    dict(pattern="stripes", rgb=(200, 40, 40)), dict(pattern="stripes", rgb=(40, 160, 60)), dict(pattern="stripes", rgb=(50, 70, 200)),
    dict(pattern="marble", rgb=(190, 60, 50)), dict(pattern="marble", rgb=(60, 150, 80)), dict(pattern="marble", rgb=(70, 90, 190)),
    # 10-24: more masonry / plaster variants
] + [dict(pattern=p, rgb=c) for p, c in zip(
    ["brick", "marble", "tiles", "brick", "marble", "tiles", "brick", "marble", "tiles", "brick", "marble", "tiles", "brick", "marble", "marble"],
    [(205, 190, 165), (230, 225, 215), (190, 180, 170), (220, 200, 175), (210, 205, 195), (185, 175, 160), (215, 195, 180), (235, 230, 220),
     (195, 190, 180), (200, 180, 160), (225, 215, 200), (180, 170, 155), (210, 190, 170), (240, 235, 225), (170, 165, 160)])]


def _texture_sizes():
    # 18 x 1024^2 (72 MiB) + 1 x 2048^2 (16 MiB) + 6 x 512^2 (6 MiB) = 94 MiB
    sizes = [(1024, 1024)] * 25
    sizes[2] = (2048, 2048)  # floor
    for i in (3, 7, 8, 9, 23, 24):
        sizes[i] = (512, 512)
    return sizes


# ----------------------------------------------------------------------------- mesh helpers
# Tessellation factor of every surface grid (atrium(scale)): scale s multiplies both grid directions, i.e. s^2 times the
# triangles (265 k at 1; 17 M at 8 -- 2.2 GB of BVH records + triangles, an order of magnitude past the 256 MiB Infinity Cache:
# the out-of-cache regime of bench.py --scene-scale).  Same shapes, textures and camera; the surfaces' small bumps are re-sampled.
_SCALE = 1
# atrium(detail="clutter"): the architecture's flat grids at half the density in both directions and the columns at half the height
# segments, so that the props below make up most of the triangles at about the same total (see _clutter)
_THIN = 1


class _Mesh:
    def __init__(self):
        self.P, self.N, self.UV, self.T = [], [], [], []

    def add_grid(self, pos, nrm, uv, tex):
        """pos/nrm: (nu+1, nv+1, 3), uv: (nu+1, nv+1, 2) vertex grids -> 2 triangles per cell."""
        a = (slice(None, -1), slice(None, -1)); b = (slice(1, None), slice(None, -1))
        c = (slice(1, None), slice(1, None)); d = (slice(None, -1), slice(1, None))
        for tri in ((a, b, c), (a, c, d)):
            self.P.append(np.stack([pos[t].reshape(-1, 3) for t in tri], axis=1))
            self.N.append(np.stack([nrm[t].reshape(-1, 3) for t in tri], axis=1))
            self.UV.append(np.stack([uv[t].reshape(-1, 2) for t in tri], axis=1))
            self.T.append(np.full(self.P[-1].shape[0], tex, np.uint32))

    def arrays(self):
        P = np.concatenate(self.P).astype(np.float32).reshape(-1, 9)
        N = np.concatenate(self.N).astype(np.float32).reshape(-1, 9)
        UV = np.concatenate(self.UV).astype(np.float32).reshape(-1, 6)
        T = np.concatenate(self.T)
        return P, N, UV, T


def _normalize(v):
    return v / np.sqrt((v * v).sum(axis=-1, keepdims=True))


def _bump(u, v, seed, amp):
    """Small deterministic displacement so flat surfaces are not perfectly planar/regular."""
    h = _hash32((u * 4096).astype(np.int64).astype(np.uint32) * np.uint32(2654435761) ^ (v * 4096).astype(np.int64).astype(np.uint32) * np.uint32(40503) ^ np.uint32(seed))
    return ((h & np.uint32(1023)).astype(np.float64) / 1023.0 - 0.5) * amp


def _plane(mesh, origin, eu, ev, nu, nv, tex, uv_scale, bump_seed=0, bump=0.0):
    nu, nv = max(nu * _SCALE // _THIN, 1), max(nv * _SCALE // _THIN, 1)
    origin, eu, ev = (np.asarray(a, np.float64) for a in (origin, eu, ev))
    s, t = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="ij")
    n = _normalize(np.cross(eu, ev))
    pos = origin + s[..., None] * eu + t[..., None] * ev
    if bump:
        interior = np.zeros_like(s, bool); interior[1:-1, 1:-1] = True
        pos = pos + (np.where(interior, _bump(s, t, bump_seed, bump), 0.0))[..., None] * n
    nrm = np.broadcast_to(n, pos.shape)
    uv = np.stack([s * uv_scale[0], t * uv_scale[1]], axis=-1)
    mesh.add_grid(pos, nrm, uv, tex)


def _column(mesh, cx, cz, y0, y1, radius, sides, segs, tex, flutes=12):
    sides, segs = sides * _SCALE, max(segs * _SCALE // _THIN, 2)
    ang = np.arange(sides + 1, dtype=np.float64) * (6.283185307179586 / sides)
    sn, cs = _sincos(ang)
    fl, _ = _sincos(ang * flutes)
    t = np.linspace(0.0, 1.0, segs + 1)
    # entasis (slight bulge) and flutes
    rad = radius * (1.0 - 0.12 * t[:, None] * t[:, None]) * (1.0 - 0.04 * fl[None, :] * fl[None, :])
    x = cx + rad * cs[None, :]; z = cz + rad * sn[None, :]
    y = np.broadcast_to((y0 + (y1 - y0) * t)[:, None], x.shape)
    pos = np.stack([x, y, z], axis=-1)
    nrm = _normalize(np.stack([np.broadcast_to(cs[None, :], x.shape), np.zeros_like(x), np.broadcast_to(sn[None, :], x.shape)], axis=-1))
    uv = np.stack([np.broadcast_to((ang / 6.283185307179586 * 2.0)[None, :], x.shape), np.broadcast_to((t * 3.0)[:, None], x.shape)], axis=-1)
    mesh.add_grid(pos, nrm, uv, tex)
    # capital and base as short wide drums
    for (ya, yb, r) in ((y0, y0 + 0.12, radius * 1.35), (y1 - 0.15, y1, radius * 1.45)):
        tt = np.linspace(0.0, 1.0, 2 * _SCALE + 1)
        xx = cx + r * cs[None, :] * np.ones((tt.size, 1)); zz = cz + r * sn[None, :] * np.ones((tt.size, 1))
        yy = np.broadcast_to((ya + (yb - ya) * tt)[:, None], xx.shape)
        p2 = np.stack([xx, yy, zz], axis=-1)
        n2 = _normalize(np.stack([np.broadcast_to(cs[None, :], xx.shape), np.zeros_like(xx), np.broadcast_to(sn[None, :], xx.shape)], axis=-1))
        uv2 = np.stack([np.broadcast_to((ang / 6.283185307179586)[None, :], xx.shape), np.broadcast_to(tt[:, None], xx.shape)], axis=-1)
        mesh.add_grid(p2, n2, uv2, tex)


def _arch(mesh, p0, p1, y_spring, rise, thickness, depth_axis, depth, steps, tex):
    """Semi-elliptical arch band between two column tops, extruded along depth_axis."""
    p0, p1 = np.asarray(p0, np.float64), np.asarray(p1, np.float64)
    steps = steps * _SCALE
    a = np.linspace(0.0, 3.141592653589793, steps + 1)
    sn, cs = _sincos(a)
    mid = 0.5 * (p0 + p1); half = 0.5 * (p1 - p0)
    d = np.zeros(3); d[depth_axis] = depth
    w = np.linspace(-0.5, 0.5, 4 * _SCALE + 1)
    for r_scale, flip in ((1.0, 1.0), (1.0 + thickness, -1.0)):
        base = mid[None, :] - cs[:, None] * half[None, :] * r_scale
        base = base + np.array([0.0, 1.0, 0.0])[None, :] * (y_spring + rise * r_scale * sn)[:, None]
        pos = base[:, None, :] + w[None, :, None] * d[None, None, :]
        tang = np.gradient(base, axis=0)
        nrm = _normalize(np.cross(tang, d)) * flip
        nrm = np.broadcast_to(nrm[:, None, :], pos.shape)
        uv = np.stack([np.broadcast_to((a / 3.141592653589793 * 2.0)[:, None], pos.shape[:2]), np.broadcast_to((w + 0.5)[None, :], pos.shape[:2])], axis=-1)
        mesh.add_grid(pos, nrm, uv, tex)


def _curtain(mesh, p0, p1, y_top, y_bot, waves, nu, nv, tex, amp, seed):
    nu, nv = nu * _SCALE, nv * _SCALE
    p0, p1 = np.asarray(p0, np.float64), np.asarray(p1, np.float64)
    s, t = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="ij")
    along = p1 - p0
    perp = _normalize(np.cross(along, np.array([0.0, 1.0, 0.0])))
    sw, cw = _sincos(s * (6.283185307179586 * waves) + seed)
    sag, _ = _sincos(s * 3.141592653589793)
    off = amp * sw * (0.3 + 0.7 * t)
    pos = p0 + s[..., None] * along + off[..., None] * perp
    pos[..., 1] = y_top - (y_top - y_bot) * t - 0.25 * sag * (1.0 - t)
    # analytic-ish normal from finite differences
    du = np.gradient(pos, axis=0); dv = np.gradient(pos, axis=1)
    nrm = _normalize(np.cross(du, dv))
    uv = np.stack([s * 2.0 - 0.5, t * 1.5 - 0.25], axis=-1)  # includes negative u,v (fract wrap)
    mesh.add_grid(pos, nrm, uv, tex)


# ----------------------------------------------------------------------------- clutter (atrium(detail="clutter"))
def _add_grid_facing(mesh, pos, nrm, uv, tex):
    """add_grid with the winding that makes the GEOMETRIC normal agree with the shading normal: the reference offsets ray origins along the
    geometric normal whichever side a ray came from (ray_intersection.cpp:17-35), so a closed surface wound inside out shadows itself."""
    geo = np.cross(pos[1:, :-1] - pos[:-1, :-1], pos[1:, 1:] - pos[:-1, :-1])
    if float((geo * nrm[:-1, :-1]).sum()) < 0.0:
        pos, nrm, uv = pos[:, ::-1], nrm[:, ::-1], uv[:, ::-1]
    mesh.add_grid(np.ascontiguousarray(pos), np.ascontiguousarray(nrm), np.ascontiguousarray(uv), tex)



def _rand01(n, seed):
    """n deterministic uniforms in [0, 1) (integer hash of the index)."""
    return (_hash32(np.arange(n, dtype=np.uint32) * np.uint32(2654435761) + np.uint32(seed)) >> np.uint32(8)).astype(np.float64) / 16777216.0


def _displaced_sphere(mesh, centre, radius, nu, nv, tex, seed, amp):
    """A 'lion head': a latitude / longitude sphere whose radius carries three octaves of lattice noise -- curved, densely tessellated,
    with small concavities (what the heads, vases and capitals of a real asset look like to a BVH builder)."""
    lat = np.linspace(0.02, 3.141592653589793 - 0.02, nv + 1)
    lon = np.arange(nu + 1, dtype=np.float64) * (6.283185307179586 / nu)
    sl, cl = _sincos(lat); so, co = _sincos(lon)
    iu = (np.arange(nu + 1) % nu).astype(np.uint32)          # the seam closes: vertex nu == vertex 0
    disp = np.zeros((nv + 1, nu + 1))
    for octave, (cells, a) in enumerate(((6, 1.0), (13, 0.5), (29, 0.25))):
        gu = (iu[None, :].astype(np.float64) * cells / nu); gv = (np.arange(nv + 1)[:, None].astype(np.float64) * cells / nv)
        u0 = np.floor(gu).astype(np.uint32); v0 = np.floor(gv).astype(np.uint32); fu = gu - u0; fv = gv - v0

        def lattice(a_, b_):
            return (_hash32((a_ % np.uint32(cells)) * np.uint32(73856093) ^ b_ * np.uint32(19349663) ^ np.uint32(seed + 17 * octave)) & np.uint32(1023)).astype(np.float64) / 1023.0 - 0.5

        top = lattice(u0, v0) * (1 - fu) + lattice(u0 + 1, v0) * fu
        bot = lattice(u0, v0 + 1) * (1 - fu) + lattice(u0 + 1, v0 + 1) * fu
        disp = disp + a * (top * (1 - fv) + bot * fv)
    r = radius * (1.0 + amp * disp)
    c = np.asarray(centre, np.float64)
    pos = np.stack([c[0] + r * sl[:, None] * co[None, :], c[1] + r * cl[:, None] * np.ones_like(co)[None, :], c[2] + r * sl[:, None] * so[None, :]], axis=-1)
    du = np.gradient(pos, axis=1); dv = np.gradient(pos, axis=0)
    nrm = _normalize(np.cross(du, dv))          # outward
    uv = np.stack([np.broadcast_to((lon / 6.283185307179586 * 3.0)[None, :], r.shape), np.broadcast_to((lat / 3.141592653589793 * 2.0)[:, None], r.shape)], axis=-1)
    _add_grid_facing(mesh, pos, nrm, uv, tex)


def _foliage(mesh, centre, radii, count, size, tex, seed):
    """A plant: `count` small leaf triangles with random positions and orientations inside an ellipsoid.  Overlapping triangles in one
    volume are what a binned-SAH builder cannot separate: this is where multi-triangle leaves and triangle tests per ray come from."""
    c = np.asarray(centre, np.float64); rr = np.asarray(radii, np.float64)
    u = [_rand01(count, seed + 101 * k) for k in range(9)]
    # position: cube root of a uniform for the radius, direction from two uniforms
    rad = u[0] ** (1.0 / 3.0)
    cz = 2.0 * u[1] - 1.0; sz = np.sqrt(np.maximum(1.0 - cz * cz, 0.0))
    sp, cp = _sincos(6.283185307179586 * u[2])
    p = c + rad[:, None] * np.stack([sz * cp, cz, sz * sp], axis=-1) * rr
    # a random frame per leaf
    cz2 = 2.0 * u[3] - 1.0; sz2 = np.sqrt(np.maximum(1.0 - cz2 * cz2, 0.0))
    sp2, cp2 = _sincos(6.283185307179586 * u[4])
    n = np.stack([sz2 * cp2, cz2, sz2 * sp2], axis=-1)
    helper = np.where(np.abs(n[:, 1:2]) < 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    t1 = _normalize(np.cross(n, helper)); t2 = np.cross(n, t1)
    s1 = size * (0.5 + u[5]); s2 = size * (0.25 + 0.5 * u[6])
    a = p - 0.5 * s1[:, None] * t1; b = p + 0.5 * s1[:, None] * t1 + 0.3 * s2[:, None] * t2; d = p + s2[:, None] * t2 * (1.0 + u[7][:, None])
    mesh.P.append(np.stack([a, b, d], axis=1))
    mesh.N.append(np.stack([n, n, n], axis=1))
    uv0 = np.stack([u[7], u[8]], axis=-1)
    mesh.UV.append(np.stack([uv0, uv0 + np.array([0.2, 0.0]), uv0 + np.array([0.1, 0.25])], axis=1))
    mesh.T.append(np.full(count, tex, np.uint32))


def _chain(mesh, top, links, link_r, tube_r, tex, seed):
    """A hanging chain: `links` tori, alternately turned by 90 degrees (thin, curved, many small triangles in a tall thin box)."""
    nu, nv = 10, 6
    a = np.arange(nu + 1, dtype=np.float64) * (6.283185307179586 / nu); b = np.arange(nv + 1, dtype=np.float64) * (6.283185307179586 / nv)
    sa, ca = _sincos(a); sb, cb = _sincos(b)
    for k in range(links):
        c = np.asarray(top, np.float64) - np.array([0.0, 1.55 * link_r * k, 0.0])
        ring = (link_r + tube_r * cb[None, :]); y = tube_r * sb[None, :] * np.ones_like(sa)[:, None]
        x = ring * ca[:, None] * 0.62; z = ring * sa[:, None]          # an oval link, long axis vertical after the swap below
        if k % 2 == 0:
            pos = np.stack([c[0] + x, c[1] + z, c[2] + y], axis=-1)
        else:
            pos = np.stack([c[0] + y, c[1] + z, c[2] + x], axis=-1)
        du = np.gradient(pos, axis=0); dv = np.gradient(pos, axis=1)
        nrm = _normalize(np.cross(du, dv))
        uv = np.stack([np.broadcast_to((a / 6.283185307179586)[:, None], x.shape), np.broadcast_to((b / 6.283185307179586)[None, :], x.shape)], axis=-1)
        # (shading normal: away from the tube's centre line, the ring of radius link_r)
        ring0 = link_r * ca[:, None] * 0.62 * np.ones_like(cb)[None, :]; ring1 = link_r * sa[:, None] * np.ones_like(cb)[None, :]; zero = np.zeros_like(x)
        core = np.stack([c[0] + ring0, c[1] + ring1, c[2] + zero], axis=-1) if k % 2 == 0 else np.stack([c[0] + zero, c[1] + ring1, c[2] + ring0], axis=-1)
        out = _normalize(pos - core)
        _add_grid_facing(mesh, pos, out, uv, tex)


def _cable(mesh, p0, p1, segments, radius, sag, tex):
    """A sagging cable between two points as a three-sided tube of LONG segments: skinny triangles that run diagonally through space have
    bounding boxes thousands of times their own area -- the classic BVH stressor (rails, rods, stems, the chains' hangers), and what puts
    triangle tests per ray up in a real asset."""
    p0, p1 = np.asarray(p0, np.float64), np.asarray(p1, np.float64)
    t = np.linspace(0.0, 1.0, segments + 1)
    centre = p0[None, :] + t[:, None] * (p1 - p0)[None, :]
    centre[:, 1] -= sag * 4.0 * t * (1.0 - t)
    axis = _normalize((p1 - p0)[None, :])[0]
    helper = np.array([0.0, 1.0, 0.0]) if abs(axis[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
    e1 = _normalize(np.cross(axis, helper)[None, :])[0]; e2 = np.cross(axis, e1)
    ang = np.arange(4, dtype=np.float64) * (6.283185307179586 / 3.0)
    sa, ca = _sincos(ang)
    ring = radius * (ca[:, None] * e1[None, :] + sa[:, None] * e2[None, :])          # (4, 3): three sides, closed
    pos = centre[:, None, :] + ring[None, :, :]
    nrm = np.broadcast_to(_normalize(ring)[None, :, :], pos.shape)
    uv = np.stack([np.broadcast_to((t * 8.0)[:, None], pos.shape[:2]), np.broadcast_to((ang / 6.283185307179586)[None, :], pos.shape[:2])], axis=-1)
    _add_grid_facing(mesh, pos, nrm, uv, tex)


def _clutter(m):
    """The props of atrium(detail="clutter"): draped cloth, displaced spheres, chains and plants -- the curved, densely tessellated and
    overlapping detail the plain stand-in lacks.  Same 25 textures."""
    H1, H2 = 4.2, 8.4
    CZ0, CZ1 = -3.4, 3.4
    # cloth: finely folded sheets hung across the court and along the lower arcades
    for i, (x, wv) in enumerate(((-9.5, 7), (-4.5, 9), (0.5, 6), (5.0, 8), (9.5, 7))):
        _curtain(m, (x, 0, CZ0 + 0.3), (x, 0, CZ1 - 0.3), H1 + 2.4, H1 - 1.0, wv, 88, 56, 4 + i % 6, 0.16, 0.9 * i + 0.3)
    for i, x in enumerate((-10.0, -6.0, -2.0, 2.0, 6.0, 10.0)):
        _curtain(m, (x - 0.9, 0, CZ0 - 0.5), (x + 0.9, 0, CZ0 - 0.5), H1 - 0.8, 0.4, 5, 36, 42, 4 + (i + 3) % 6, 0.09, 0.5 * i)
    # heads / vases: displaced spheres on the plinths and on the parapet
    for i, (x, y, z, r) in enumerate(((-3.0, 1.35, 1.2, 0.42), (-5.5, 1.0, -1.5, 0.38), (3.5, 1.65, -0.8, 0.40), (6.5, 1.1, 1.6, 0.36), (-8.5, 0.95, 0.4, 0.45), (0.2, 1.9, 2.2, 0.33),
                                      (-7.0, H1 + 1.3, CZ0, 0.3), (-1.0, H1 + 1.3, CZ0, 0.3), (5.0, H1 + 1.3, CZ1, 0.3), (9.0, H1 + 1.3, CZ1, 0.3))):
        _displaced_sphere(m, (x, y, z), r, 56, 36, 20 + i % 5, 700 + 13 * i, 0.35)
    # chains hanging from the upper slab into the court
    for i, (x, z) in enumerate(((-6.5, -1.2), (-1.5, 1.4), (3.0, -1.6), (7.5, 0.9), (1.0, 0.2), (-9.0, 1.0))):
        _chain(m, (x, H2 - 0.1, z), 28, 0.09, 0.02, 3 if i % 2 else 1, 40 + i)
    # cables strung across the court between the galleries and down to the floor, at all sorts of angles
    for i in range(96):
        h = _rand01(6, 500 + 7 * i)
        x0 = -11.0 + 22.0 * h[0]; x1 = min(max(x0 + (h[1] - 0.5) * 14.0, -11.2), 11.2)
        ya = (H1 + 0.9) if i % 3 else (H2 + 1.4); yb = (0.05 if i % 5 == 0 else (H1 + 0.9 if i % 2 else H2 + 1.4))
        _cable(m, (x0, ya, CZ0 + 0.05 if i % 2 else CZ1 - 0.05), (x1, yb, CZ1 - 0.05 if i % 2 else CZ0 + 0.05), 4, 0.012, 0.25 + 0.5 * h[2], 3)
    # plants: clusters of overlapping leaves in planters on the court floor and on the first-floor gallery ...
    for i, (x, y, z, rr, n) in enumerate(((-3.0, 2.3, 1.2, (0.55, 0.75, 0.55), 4000), (3.5, 2.6, -0.8, (0.5, 0.8, 0.5), 4000), (6.5, 2.0, 1.6, (0.6, 0.7, 0.6), 4000),
                                          (-8.5, 1.9, 0.4, (0.7, 0.8, 0.7), 5000), (-11.0, H1 + 1.2, -5.0, (0.9, 0.9, 0.9), 5000), (11.0, H1 + 1.2, 5.0, (0.9, 0.9, 0.9), 5000),
                                          (0.0, H1 + 1.1, 5.3, (1.4, 0.8, 0.6), 6000))):
        _foliage(m, (x, y, z), rr, n, 0.13, 5 if i % 3 else 8, 9000 + 31 * i)
    # ... and trees in the court: wide, sparse canopies of larger leaves that a good part of the rays has to cross (a ray through a canopy
    # meets some tens of overlapping leaf boxes: the triangle tests per ray of a real asset's plants and drapes)
    for i, (x, z, y, r) in enumerate(((-9.5, -1.0, 3.6, 1.35), (-6.2, 1.3, 4.3, 1.5), (-2.6, -1.2, 3.9, 1.4), (1.2, 1.2, 4.6, 1.6), (4.6, -1.1, 3.8, 1.4), (8.4, 1.0, 4.2, 1.5),
                                      (10.4, -1.4, 3.4, 1.2), (-0.6, 0.3, 6.6, 1.7))):
        _foliage(m, (x, y, z), (r, 0.8 * r, r), 6500, 0.24, 5 if i % 2 else 8, 12000 + 57 * i)
        _plane(m, (x - 0.07, 0.0, z - 0.07), (0.14, 0, 0), (0, y - 0.5 * r, 0), 2, 24, 3, (0.2, 3.0), 71 + i, 0.0)      # the trunk: two crossed strips
        _plane(m, (x, 0.0, z - 0.07), (0, 0, 0.14), (0, y - 0.5 * r, 0), 2, 24, 3, (0.2, 3.0), 81 + i, 0.0)


def atrium_triangles(scale=1, detail="plain"):
    """-> positions (N,9), normals (N,9), uvs (N,6), texture index (N,) in source order."""
    global _SCALE, _THIN
    assert detail in ("plain", "clutter")
    _SCALE = int(scale)
    _THIN = 2 if detail == "clutter" else 1
    try:
        return _atrium_triangles(detail == "clutter")
    finally:
        _SCALE = 1
        _THIN = 1


def _atrium_triangles(clutter=False):
    m = _Mesh()
    X0, X1, Z0, Z1 = -15.0, 15.0, -7.0, 7.0   # outer walls
    CX0, CX1, CZ0, CZ1 = -11.5, 11.5, -3.4, 3.4  # open court
    H1, H2 = 4.2, 8.4                          # storey heights
    # floor (one big tiled texture) and ground under the aisles
    _plane(m, (X0, 0.0, Z0), (X1 - X0, 0, 0), (0, 0, Z1 - Z0), 160, 76, 2, (15.0, 7.0), 11, 0.004)
    # outer walls, two storeys each, finely tessellated brick
    wall_tex = [0, 10, 13, 16]
    for i, (o, eu) in enumerate((((X0, 0, Z0), (X1 - X0, 0, 0)), ((X1, 0, Z1), (X0 - X1, 0, 0)), ((X0, 0, Z1), (0, 0, Z0 - Z1)), ((X1, 0, Z0), (0, 0, Z1 - Z0)))):
        length = abs(eu[0]) + abs(eu[2])
        nu = int(length * 5)
        _plane(m, o, eu, (0, H2 + 1.5, 0), nu, 40, wall_tex[i], (length / 3.0, 3.3), 21 + i, 0.01)
    # aisle ceilings (first floor slab underside + top, second floor ceiling), with the court cut out:
    for y, tex in ((H1, 11), (H1 + 0.25, 12), (H2, 14)):
        _plane(m, (X0, y, Z0), (X1 - X0, 0, 0), (0, 0, CZ0 - Z0), 140, 18, tex, (10.0, 1.2), 31, 0.003)
        _plane(m, (X0, y, CZ1), (X1 - X0, 0, 0), (0, 0, Z1 - CZ1), 140, 18, tex, (10.0, 1.2), 32, 0.003)
        _plane(m, (X0, y, CZ0), (CX0 - X0, 0, 0), (0, 0, CZ1 - CZ0), 18, 34, tex, (1.2, 2.3), 33, 0.003)
        _plane(m, (CX1, y, CZ0), (X1 - CX1, 0, 0), (0, 0, CZ1 - CZ0), 18, 34, tex, (1.2, 2.3), 34, 0.003)
    # roof ring above the second storey (court stays open to the sky)
    _plane(m, (X0, H2 + 1.5, Z0), (X1 - X0, 0, 0), (0, 0, CZ0 - Z0), 60, 8, 15, (10.0, 1.2), 41, 0.0)
    _plane(m, (X0, H2 + 1.5, CZ1), (X1 - X0, 0, 0), (0, 0, Z1 - CZ1), 60, 8, 15, (10.0, 1.2), 42, 0.0)
    _plane(m, (X0, H2 + 1.5, CZ0), (CX0 - X0, 0, 0), (0, 0, CZ1 - CZ0), 8, 14, 15, (1.2, 2.3), 43, 0.0)
    _plane(m, (CX1, H2 + 1.5, CZ0), (X1 - CX1, 0, 0), (0, 0, CZ1 - CZ0), 8, 14, 15, (1.2, 2.3), 44, 0.0)
    # parapet faces around the court at each slab
    for y0, y1, tex in ((H1, H1 + 0.9, 17), (H2, H2 + 1.5, 18)):
        _plane(m, (CX0, y0, CZ0), (CX1 - CX0, 0, 0), (0, y1 - y0, 0), 110, 5, tex, (8.0, 0.5), 51, 0.004)
        _plane(m, (CX0, y0, CZ1), (CX1 - CX0, 0, 0), (0, y1 - y0, 0), 110, 5, tex, (8.0, 0.5), 52, 0.004)
        _plane(m, (CX0, y0, CZ0), (0, 0, CZ1 - CZ0), (0, y1 - y0, 0), 34, 5, tex, (2.4, 0.5), 53, 0.004)
        _plane(m, (CX1, y0, CZ0), (0, 0, CZ1 - CZ0), (0, y1 - y0, 0), 34, 5, tex, (2.4, 0.5), 54, 0.004)
    # colonnades: columns on the court edge, both storeys
    col_x = np.linspace(CX0, CX1, 13)
    col_z = np.linspace(CZ0, CZ1, 4)[1:-1]
    spots = [(x, CZ0) for x in col_x] + [(x, CZ1) for x in col_x] + [(CX0, z) for z in col_z] + [(CX1, z) for z in col_z]
    for storey, (y0, y1, r) in enumerate(((0.0, H1 - 0.7, 0.32), (H1 + 0.25, H2 - 0.6, 0.24))):
        for i, (x, z) in enumerate(spots):
            _column(m, x, z, y0, y1, r, 32, 20, 1 if (i + storey) % 3 else 3)
    # arches between neighbouring columns (long sides and short sides), both storeys
    for storey, (ys, rise) in enumerate(((H1 - 0.7, 0.62), (H2 - 0.6, 0.5))):
        for zc in (CZ0, CZ1):
            for i in range(len(col_x) - 1):
                _arch(m, (col_x[i], 0, zc), (col_x[i + 1], 0, zc), ys, rise, 0.22, 2, 0.6, 26, 19 + storey)
        zs = [CZ0] + list(col_z) + [CZ1]
        for xc in (CX0, CX1):
            for i in range(len(zs) - 1):
                _arch(m, (xc, 0, zs[i]), (xc, 0, zs[i + 1]), ys, rise, 0.22, 0, 0.6, 26, 19 + storey)
    # draperies hanging in the upper arcade openings and across the court
    k = 0
    for zc in (CZ0 + 0.15, CZ1 - 0.15):
        for i in range(0, len(col_x) - 1):
            if i % 2 == (0 if zc < 0 else 1):
                _curtain(m, (col_x[i] + 0.3, 0, zc), (col_x[i + 1] - 0.3, 0, zc), H2 - 0.9, H1 + 1.0, 3, 32, 34, 4 + k % 6, 0.12, 0.7 * k)
                k += 1
    for i, x in enumerate((-7.5, -2.5, 2.5, 7.5)):
        _curtain(m, (x, 0, CZ0 + 0.4), (x, 0, CZ1 - 0.4), H2 + 0.6, H2 - 2.8, 2, 44, 30, 4 + (i + 2) % 6, 0.2, 1.3 * i)
    # a few plinth blocks / planters on the court floor (near the default camera)
    for i, (x, z, s, h) in enumerate(((-3.0, 1.2, 0.6, 0.9), (-5.5, -1.5, 0.8, 0.6), (3.5, -0.8, 0.5, 1.2), (6.5, 1.6, 0.7, 0.7), (-8.5, 0.4, 0.9, 0.5), (0.2, 2.2, 0.4, 1.5))):
        tex = 20 + i % 5
        for (o, eu, ev) in (((x - s, 0, z - s), (2 * s, 0, 0), (0, h, 0)), ((x + s, 0, z + s), (-2 * s, 0, 0), (0, h, 0)),
                            ((x - s, 0, z + s), (0, 0, -2 * s), (0, h, 0)), ((x + s, 0, z - s), (0, 0, 2 * s), (0, h, 0)),
                            ((x - s, h, z - s), (2 * s, 0, 0), (0, 0, 2 * s))):
            _plane(m, o, eu, ev, 10, 10, tex, (1.0, 1.0), 61 + i, 0.002)
    if clutter:
        _clutter(m)
    return m.arrays()


def atrium_textures():
    return [_texture(kind, size, 1000 + 17 * i) for i, (kind, size) in enumerate(zip(_TEXTURE_KINDS, _texture_sizes()))]


_CACHE = {}


def atrium(scale=1, detail="plain"):
    """-> (PtFormat, info dict).  The BVH is built by the product's builder (host by default: rf.set_bake_bvh_builder).
    scale > 1: every surface grid tessellated scale x finer in both directions (scale^2 x the triangles).
    detail = "clutter": the harder stand-in -- the flat grids thinned, plus draped cloth, displaced spheres, chains and plants of
    overlapping leaves (curved, densely tessellated, not separable by a SAH split: multi-triangle leaves, several triangle tests per ray)."""
    key = ("atrium" if scale == 1 else f"atrium{scale}") + ("" if detail == "plain" else "_" + detail)
    if key not in _CACHE:
        P, N, UV, T = atrium_triangles(scale, detail)
        tex = atrium_textures()
        pt = PtFormat.from_triangles(P, N, UV, T, tex)
        h = hashlib.sha256()
        for a in (P, N, UV, T):
            h.update(np.ascontiguousarray(a).tobytes())
        for px, w, hh in tex:
            h.update(px.tobytes())
        name = "synthetic atrium (Sponza stand-in)" if scale == 1 else f"synthetic atrium x{scale} tessellation (out-of-cache variant of the Sponza stand-in)"
        if detail != "plain":
            name = name.replace("synthetic atrium", "synthetic atrium with clutter")
        info = dict(name=name, triangles=int(P.shape[0]), textures=len(tex),
                    texture_mib=sum(px.size for px, _, _ in tex) * 4 / 2 ** 20, digest=h.hexdigest()[:16])
        _CACHE[key] = (pt, info)
    return _CACHE[key]


def quad_scene(albedo_rgb=(255, 255, 255), size=2.0, y=0.0):
    """Two-triangle floor with a 1x1 texture: the analytic known-answer scene for shading tests."""
    s = size
    P = np.array([[-s, y, -s, s, y, s, s, y, -s], [-s, y, -s, -s, y, s, s, y, s]], np.float32)  # geometric normal +y
    N = np.tile(np.array([0, 1, 0], np.float32), (2, 3))
    UV = np.array([[0, 0, 1, 1, 1, 0], [0, 0, 0, 1, 1, 1]], np.float32)
    T = np.zeros(2, np.uint32)
    r, g, b = albedo_rgb
    tex = [(np.array([b | (g << 8) | (r << 16) | (255 << 24)], np.uint32), 1, 1)]
    return PtFormat.from_triangles(P, N, UV, T, tex)
