"""An INDEPENDENT restatement of the reference's integrator for scenes made of axis-aligned rectangles, in float64 numpy:
no BVH, no Moller-Trumbore, no shared code with oracle/rf_oracle.c or the product.  It follows
src/pt/reference_path_tracer.wgsl:180-234 (rayColor) equation by equation:

    bounce = 1
    loop:  hit?  albedo = texture colour ^ 2.2                                          (:303-307, 552-565)
                 l      = onb(sunDirection) * coneSample(u)                             (:287-292, 568-579, 309-319)
                 radiance += throughput * solarRadiance * (albedo / pi) * dot(n, l) * V * SOLAR_INV_PDF   (:194-203; no clamp)
                 if bounce == numBounces: break                                         (:205)
                 wi = onb(n) * cosineHemisphere(u);  throughput *= albedo;  ray = (p, wi)                 (:209-211)
           miss: radiance += throughput * sky(theta = acos d.y, gamma = acos clamp(d . s)); break          (:212-228, 247-275)

with ONE blue-noise pair u per path (wgsl:52-55,194,209).  Rectangles carry the shading normal of their vertex
attributes and, separately, the geometric normal of their winding: the hit point is pushed off the surface along the
GEOMETRIC normal whatever side the ray came from (wgsl:514-516, 523-544), which decides what the next rays see.

Used by tests/test_oracle_pins.py (pins the oracle's shading half, which no reference test covers) and by the GPU parity
tests.  Every decision (which rectangle, hit or miss, visible or not) is made with a margin; a sample whose decisions
fall inside the margin is reported as not robust and skipped by the tests."""
import math

import numpy as np

SOLAR_COS_MAX = float(np.frombuffer(np.uint32(0x3F7FFF5A).tobytes(), np.float32)[0])   # wgsl:79-83 in f32
SOLAR_INV_PDF = float(np.frombuffer(np.uint32(0x38826048).tobytes(), np.float32)[0])
T_MAX = 10000.0
MARGIN = 1e-3


class Rect:
    """Axis-aligned rectangle: plane `axis` = `value`, spanning [lo, hi] in the two other axes (ascending axis order)."""

    def __init__(self, axis, value, lo, hi, shading_normal, geometric_sign, srgb):
        self.axis, self.value = axis, float(value)
        self.others = [a for a in range(3) if a != axis]
        self.lo, self.hi = np.float64(lo), np.float64(hi)
        self.n = np.float64(shading_normal)
        self.ng = np.zeros(3); self.ng[axis] = geometric_sign
        self.albedo = (np.float64(srgb) / 255.0) ** 2.2
        self.srgb = tuple(int(c) for c in srgb)

    def triangles(self):
        """Two triangles (9 floats each) wound so that normalize(cross(p1 - p0, p2 - p0)) == self.ng."""
        a, b = self.others
        def pt(u, v):
            p = np.zeros(3); p[self.axis] = self.value; p[a] = u; p[b] = v
            return p
        c00, c10, c11, c01 = pt(self.lo[0], self.lo[1]), pt(self.hi[0], self.lo[1]), pt(self.hi[0], self.hi[1]), pt(self.lo[0], self.hi[1])
        tris = [(c00, c10, c11), (c00, c11, c01)]
        out = []
        for p0, p1, p2 in tris:
            if np.dot(np.cross(p1 - p0, p2 - p0), self.ng) < 0:
                p1, p2 = p2, p1
            out.append(np.concatenate([p0, p1, p2]))
        return out

    def intersect(self, o, d):
        """-> (t, margin) or None; margin = distance of the decision from flipping (relative to the rectangle size)."""
        if abs(d[self.axis]) < 1e-12:
            return None
        t = (self.value - o[self.axis]) / d[self.axis]
        if t <= 1e-5 or t >= T_MAX:
            return None
        p = o + t * d
        a, b = self.others
        m = min(p[a] - self.lo[0], self.hi[0] - p[a], p[b] - self.lo[1], self.hi[1] - p[b])
        return t, m


def scene_arrays(rects):
    """-> positions36 (N,9), normals36 (N,9), texcoords24 (N,6), textureIdx (N,), textures [(pixels, 1, 1)] in source order."""
    P, N, T, I, tex = [], [], [], [], []
    for k, r in enumerate(rects):
        for tri in r.triangles():
            P.append(tri); N.append(np.tile(r.n, 3)); T.append(np.zeros(6)); I.append(k)
        b, g, rr = r.srgb[2], r.srgb[1], r.srgb[0]
        tex.append((np.array([b | (g << 8) | (rr << 16) | (255 << 24)], np.uint32), 1, 1))
    return (np.array(P, np.float32), np.array(N, np.float32), np.array(T, np.float32), np.array(I, np.uint32), tex)


def onb(n):
    """Duff et al. (wgsl:309-319) -> columns (u, v, n)."""
    s = 1.0 if n[2] >= 0.0 else -1.0
    a = -1.0 / (s + n[2])
    b = n[0] * n[1] * a
    return np.array([1.0 + s * n[0] * n[0] * a, s * b, -s * n[0]]), np.array([b, s + n[1] * n[1] * a, -n[1]])


def sky_radiance(sky40, theta, gamma):
    """wgsl:247-275 in float64 from the 40-float AlignedSkyState (params 27, sky radiances 3, solar 3, pad 3, sun dir 3)."""
    out = np.zeros(3)
    for c in range(3):
        p = np.float64(sky40[9 * c:9 * c + 9]); r = float(sky40[27 + c])
        cg = math.cos(gamma); ct = abs(math.cos(theta))
        exp_m = math.exp(p[4] * gamma)
        mie = (1.0 + cg * cg) / math.pow(1.0 + p[8] * p[8] - 2.0 * p[8] * cg, 1.5)
        lhs = 1.0 + p[0] * math.exp(p[1] / (ct + 0.01))
        rhs = p[2] + p[3] * exp_m + p[5] * cg * cg + p[6] * mie + p[7] * math.sqrt(ct)
        out[c] = r * lhs * rhs
    return out


def closest(rects, o, d, skip_margin=False):
    best = None
    worst_margin = np.inf
    for k, r in enumerate(rects):
        h = r.intersect(o, d)
        if h is None:
            continue
        t, m = h
        if m < 0:
            worst_margin = min(worst_margin, -m)      # missed this rectangle by -m
            continue
        worst_margin = min(worst_margin, m)
        if best is None or t < best[0]:
            best = (t, k)
    return best, worst_margin


def path_sample(rects, cam19, sky40, width, height, x, y, u, num_bounces):
    """One sample of pixel (x, y) with blue-noise pair u -> (rgb float64, robust flag, trace list)."""
    cam = np.float64(cam19)
    origin, llc, horiz, vert = cam[0:3], cam[3:6], cam[6:9], cam[9:12]
    assert cam[18] == 0.0, "pinhole only"
    s = (x + 0.5) / width + u[0] / width                          # wgsl:42-54
    t = (1.0 - (y + 0.5) / height) + u[1] / height
    d = llc + s * horiz + t * vert - origin
    d /= np.linalg.norm(d)
    o = origin.copy()
    sun = np.float64(sky40[36:39]); solar = np.float64(sky40[30:33])
    # the light sample of this path (same u at every vertex)
    # cos / sin of the cone angle in f32, as the shader computes them: 1 - cos(theta) ~ 1e-5 sits three digits above
    # f32's resolution, so these four operations ARE the result; everything else is f64
    f = np.float32
    cos32 = f(1.0) - f(u[0]) * (f(1.0) - f(SOLAR_COS_MAX))
    sin32 = np.sqrt(f(1.0) - cos32 * cos32)
    cos_t, sin_t = float(cos32), float(sin32)
    phi = 2.0 * math.pi * u[1]
    su, sv = onb(sun)
    light = (math.cos(phi) * sin_t) * su + (math.sin(phi) * sin_t) * sv + cos_t * sun
    radiance, throughput = np.zeros(3), np.ones(3)
    robust, trace = True, []
    for bounce in range(1, num_bounces + 1):
        hit, m = closest(rects, o, d)
        robust &= m > MARGIN
        if hit is None:
            theta = math.acos(max(-1.0, min(1.0, d[1]))); gamma = math.acos(max(-1.0, min(1.0, float(np.dot(d, sun)))))
            radiance += throughput * sky_radiance(sky40, theta, gamma)
            trace.append(("sky", bounce))
            break
        th, k = hit
        r = rects[k]
        p = o + th * d
        p_off = p + 1e-4 * r.ng                                   # offsetRay: a few hundred ulps along the geometric normal
        occ, m2 = closest(rects, p_off, light)
        # a ray that leaves its surface at a grazing angle re-crosses the surface's own plane at a distance that depends on the
        # size of the offset (a few hundred ulps in the shader, 1e-4 here): not a robust decision
        robust &= m2 > MARGIN and abs(float(np.dot(r.ng, light))) > 0.02
        vis = 0.0 if occ is not None else 1.0
        radiance += throughput * solar * (r.albedo / math.pi) * float(np.dot(r.n, light)) * vis * SOLAR_INV_PDF
        trace.append(("hit", bounce, k, vis, float(np.dot(r.n, light))))
        if bounce == num_bounces:
            break
        bu, bv = onb(r.n)
        sq = math.sqrt(max(0.0, 1.0 - u[0]))
        wi = (math.cos(phi) * sq) * bu + (math.sin(phi) * sq) * bv + math.sqrt(u[0]) * r.n
        throughput = throughput * r.albedo
        robust &= abs(float(np.dot(r.ng, wi))) > 0.02
        o, d = p_off, wi
    return radiance, robust, trace
