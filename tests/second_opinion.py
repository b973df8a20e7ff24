"""A SECOND, independent restatement of the reference's shading path, in numpy binary32 scalars (TEST INFRASTRUCTURE).

oracle/yahr_oracle.c is pinned by every known-answer test of the reference's test/Spec.hs -- but those vectors pin the
integrator only with radiance 0 (Spec.hs:265-297: the eye ray travels along +n and collideTriangle back-face-culls it),
so the MAGNITUDES of Integrators.hs:37, Lights.hs:19 and BSDF.hs:15-30 rest on one reading of the Haskell.  This file
is a second reading, written from the .hs files alone (not from the C): every operator is applied in Haskell's order
with one IEEE binary32 rounding per operation (np.float32 scalars), lists are lists, `sum` is a left fold from 0.
tests/test_second_opinion.py demands that the two restatements agree BIT FOR BIT on non-zero radiance.

Scope: a scene whose culling mode is `BVH 0 _` (Culling.hs:33: buildTree 0 xs = multiLeaf xs) -- one joint box, then
every primitive in `objects >>= expand` order through collideAll -- which is all the traversal the shading needs.
"""
import math

import numpy as np

F = np.float32
ZERO, ONE, TWO = F(0), F(1), F(2)


# ---- Vectors.hs -----------------------------------------------------------------------------------------
def vec(x, y, z):
    return (F(x), F(y), F(z))


def vof(l):                                             # :20-21
    return (F(l), F(l), F(l))


def vadd(a, b):                                         # Num Vec3 (+) :24
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def vsub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def vmul(a, b):                                         # (*) is component-wise :26
    return (a[0] * b[0], a[1] * b[1], a[2] * b[2])


def vneg(a):
    return (-a[0], -a[1], -a[2])


def dot(a, b):                                          # (.*) :32-33: x1*y1 + x2*y2 + x3*y3, + is infixl 6
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def lensq(v):
    return dot(v, v)


def length(v):
    return F(np.sqrt(lensq(v)))                         # sqrt :: Float, correctly rounded


def scale(f, v):                                        # f @* v = vof f * v :42-43
    return vmul(vof(f), v)


def norm(v):                                            # (1 / len v) @* v :46-47
    return scale(ONE / length(v), v)


def cross(a, b):                                        # :49-53
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def hmin(x, y):                                         # Ord's default: min x y = if x <= y then x else y
    return x if x <= y else y


def hmax(x, y):                                         # max x y = if x <= y then y else x
    return y if x <= y else x


# ---- Shapes.hs ------------------------------------------------------------------------------------------
def collide_sphere(what, r, s, ray):                    # :13-29
    x0, u, tmax = ray
    d = vsub(x0, s)
    a = dot(u, u)
    b = dot(vmul(vof(TWO), d), u)                       # 2 * d .* u: `*` and `.*` are both infixl 7 -> (2*d) .* u
    c = dot(d, d) - r * r
    delta = b * b - F(4) * a * c                        # 4 * a * c = (4*a)*c
    if delta < 0:
        return None
    sq = F(np.sqrt(delta))
    for t in ((-b - sq) / (TWO * a), (-b + sq) / (TWO * a)):
        if t > 0 and t <= tmax:
            x = vadd(x0, scale(t, u))                   # rayAt
            n = norm(vsub(x, s))
            return (t, dict(point=x, normal=n, dpdu=cross(n, vec(0, 0, 1)), dpdv=cross(n, vec(0, 1, 0))), what)
    return None


def collide_triangle(what, p0, p1, p2, n0, n1, n2, ray):        # :36-59
    x0, u, tmax = ray
    e1 = vsub(p1, p0)
    e2 = vsub(p2, p0)
    s = vsub(x0, p0)
    s1 = cross(u, e2)
    s2 = cross(s, e1)
    with np.errstate(all="ignore"):
        inv = ONE / dot(s1, e1)
        t = dot(s2, e2) * inv
        b1 = dot(s1, s) * inv
        b2 = dot(s2, u) * inv
        b0 = ONE - b1 - b2
        ns = vadd(vadd(scale(b0, n0), scale(b1, n1)), scale(b2, n2))
        ok = (b0 >= 0 and b0 <= 1 and b1 >= 0 and b1 <= 1 and b2 >= 0 and b2 <= 1 and t > 0 and t <= tmax
              and dot(u, ns) < 0)
        if not ok:
            return None
        ss = norm(e2)
        ts = norm(cross(ss, ns))
    return (t, dict(point=vadd(x0, scale(t, u)), normal=ns, dpdu=cross(ts, ns), dpdv=ts), what)


def bound(prim):                                        # boundSphere :32-33 / boundTriangle :62-63
    if prim[0] == "sphere":
        _, c, r, _ = prim
        a, b = vadd(c, vof(r)), vsub(c, vof(r))
        return tuple(hmin(x, y) for x, y in zip(a, b)), tuple(hmax(x, y) for x, y in zip(a, b))
    pts = prim[1:4]
    lo = tuple(hmin(x, y) for x, y in zip(pts[0], pts[1]))
    hi = tuple(hmax(x, y) for x, y in zip(pts[0], pts[1]))
    return tuple(hmin(x, y) for x, y in zip(lo, pts[2])), tuple(hmax(x, y) for x, y in zip(hi, pts[2]))


# ---- AABBs.hs:29-40 -------------------------------------------------------------------------------------
def bb_ray(box, ray):
    (lo, hi), (x0, u, tmax) = box, ray
    tnear, tfar = ZERO, tmax
    with np.errstate(all="ignore"):
        for d in range(3):
            inv = ONE / u[d]
            t0 = (lo[d] - x0[d]) * inv
            t1 = (hi[d] - x0[d]) * inv
            tnear, tfar = hmax(tnear, hmin(t0, t1)), hmin(tfar, hmax(t0, t1))
    return not (tnear > tfar)


# ---- scene = what `cull (BVH 0 _) (buildCollisionModel scene)` is: Culling.hs:33,52-54, Rays.hs:39-46 ----------
class Scene:
    def __init__(self, prims, materials, lights, depth):
        """prims: ("sphere", centre, radius, mat) | ("triangle", p0, p1, p2, n0, n1, n2, mat) in expand order;
        materials: index -> (diffuse, specular, shininess); lights: [(pos, spectrum)]"""
        self.prims, self.materials, self.lights, self.depth = prims, materials, lights, depth
        boxes = [bound(p) for p in prims]
        inf = F(np.inf)
        lo, hi = vof(inf), vof(-inf)                     # foldl join empty bbs (AABBs.hs:10-11,25-27)
        for blo, bhi in boxes:
            lo = tuple(hmin(x, y) for x, y in zip(lo, blo))
            hi = tuple(hmax(x, y) for x, y in zip(hi, bhi))
        self.box = (lo, hi)

    def collide(self, ray):                              # wrapCollider (collideAll colliders) jointBB
        if len(self.prims) == 0:
            return None
        if len(self.prims) == 1:                         # buildTree _ [(bb, col)] = wrapCollider col bb
            if not bb_ray(bound(self.prims[0]), ray):
                return None
        elif not bb_ray(self.box, ray):
            return None
        x0, u, tmax = ray
        best = None
        for p in self.prims:                             # collideAll: later hits replace, tMax is cut (Rays.hs:39-46)
            if p[0] == "sphere":
                h = collide_sphere(p[3], p[2], p[1], (x0, u, tmax))
            else:
                h = collide_triangle(p[7], *p[1:7], (x0, u, tmax))
            if h is not None:
                best, tmax = h, h[0]
        return best

    def reachable(self, p0, p1):                         # Rays.hs:49-54
        d = vsub(p1, p0)
        return self.collide((p0, norm(d), length(d))) is None


# ---- BSDF.hs:12-46, Shaders.hs:12-14 ----------------------------------------------------------------------
PI = F(math.pi)                                          # pi :: Float


def powf(x, e):
    """`**` on Float is libm's powf; the double-precision pow rounded once to binary32 is the correctly rounded value,
    which glibc's powf (< 1 ulp, in practice correctly rounded) returns too"""
    return F(math.pow(float(x), float(e)))


def bsdf_at(material, dg, i, o):
    diffuse, specular, shininess = material
    nn = dg["normal"]
    sn = norm(dg["dpdu"])
    tn = cross(nn, sn)

    def to_loc(v):
        return (dot(v, sn), dot(v, tn), dot(v, nn))
    li, lo = to_loc(i), to_loc(o)

    def lambertian():
        return vof(ONE / PI) if (li[2] > 0 and lo[2] > 0) else vof(ZERO)

    def blinn(e):
        if not (li[2] > 0 and lo[2] > 0):
            return vof(ZERO)
        h = norm(vadd(li, lo))
        cos_o, cos_i = abs(lo[2]), abs(li[2])
        cos_h = dot(li, h)
        o_dot_h = dot(lo, h)
        d = (e + TWO) * powf(abs(h[2]), e) / (TWO * PI)
        g = hmin(ONE, hmin(TWO * cos_h * cos_o / o_dot_h, TWO * cos_h * cos_i / o_dot_h))
        f = ONE                                          # fresnel ct = 1
        return vof(d * g * f / (F(4) * cos_i * cos_o))
    # blinnPhong = diffuse &* Lambertian &+ specular &* Blinn shininess = Composite [Scaled d Lambertian, Scaled s (Blinn e)]
    # locAt (Composite bs) = sum (map ...): foldl (+) 0
    acc = vof(ZERO)
    acc = vadd(acc, vmul(diffuse, lambertian()))
    acc = vadd(acc, vmul(specular, blinn(shininess)))
    return acc


# ---- Lights.hs:15-24, Integrators.hs:22-61 ------------------------------------------------------------------
def radiance(scene, ray):
    def vcast(depth, r):
        if depth == 0:
            return vof(ZERO)
        hit = scene.collide(r)
        if hit is None:
            return vof(ZERO)
        return vhit(depth, r, hit)

    def vhit(depth, r, hit):
        _, dg, mat = hit
        x, n = dg["point"], dg["normal"]
        x0, u, _ = r
        material = scene.materials[mat]
        refl = vsub(u, scale(TWO * dot(u, n), n))        # reflectionDir u n = u - 2 * (u .* n) @* n
        f_r = bsdf_at(material, dg, refl, vneg(u))
        rs = vcast(depth - 1, (vadd(x, scale(F(0.001), refl)), refl, F(1e6)))
        # (n .* r) @* f r * rs + directIllumination ...:  .*, @*, * are infixl 7
        first = vmul(scale(dot(n, refl), f_r), rs)
        direct = vof(ZERO)                               # sum = foldl (+) 0
        for lpos, spectrum in scene.lights:
            p2l = vsub(lpos, x)
            ldir = norm(p2l)
            intensity = scale(ONE / lensq(p2l), spectrum)
            k = bsdf_at(material, dg, ldir, vneg(u))
            contrib = vof(ZERO)
            if lensq(k) > 0 and scene.reachable(vadd(x, scale(F(0.001), ldir)), lpos):       # && is lazy
                contrib = vmul(scale(abs(dot(ldir, n)), k), intensity)
            direct = vadd(direct, contrib)
        return vadd(first, direct)
    with np.errstate(all="ignore"):
        return vcast(scene.depth, ray)
