"""The argument behind the fat-leaf culling of the packet walker, checked WITHOUT a GPU (render_kernel.hip: leaf_culled,
device_math.hpp: tri_far_packet).

`tri_far_packet` evaluates tri_stage0's expression tree -- s = o - p0, s1 = u x e2, det = s1 . e1, q = s1 . s, the same
binary32 operations in the same order -- in interval arithmetic whose end points are rounded TO NEAREST, over the box
[ol, oh] x [ul, uh] that bounds a packet's rays, and claims: if min|q| > fl(1.5 * max|det|) on the intervals, then EVERY
ray of the packet computes |q| > 1.5 |det| (tri_far), so the primitive step may be skipped for the whole wavefront.
Rounding to nearest is monotone and the operations are monotone in each operand (products: extrema at the corners), so
the enclosure holds for the COMPUTED values; it needs no infinities / NaNs on the way, which the magnitude limits
(|o|, |p0|, |e| <= 2^32, |u| <= 2^20) guarantee.  This file restates both functions in numpy binary32, operation by
operation, and checks the claim on random and adversarial packets (subnormals, zeros, 1e-38 .. 4e9, degenerate
triangles, packets of one ray, packets that straddle the triangle), including that it is not vacuous."""
import numpy as np

F = np.float32


def stage0(p0, e1, e2, o, u):
    """device_math.hpp: tri_stage0 for rays [n, 3] against ONE triangle -> (det, q), binary32 operation by operation"""
    s = (o - p0).astype(F)
    s1 = np.stack([(u[:, 1] * e2[2]).astype(F) - (u[:, 2] * e2[1]).astype(F),
                   (u[:, 2] * e2[0]).astype(F) - (u[:, 0] * e2[2]).astype(F),
                   (u[:, 0] * e2[1]).astype(F) - (u[:, 1] * e2[0]).astype(F)], axis=1).astype(F)
    det = ((s1[:, 0] * e1[0]).astype(F) + (s1[:, 1] * e1[1]).astype(F)).astype(F) + (s1[:, 2] * e1[2]).astype(F)
    q = ((s1[:, 0] * s[:, 0]).astype(F) + (s1[:, 1] * s[:, 1]).astype(F)).astype(F) + (s1[:, 2] * s[:, 2]).astype(F)
    return det.astype(F), q.astype(F)


def tri_far(q, det):
    return np.abs(q) > (F(1.5) * np.abs(det)).astype(F)


def imul_point(xl, xh, y):
    a, b = F(xl * y), F(xh * y)
    return min(a, b), max(a, b)


def imul(xl, xh, yl, yh):
    c = [F(xl * yl), F(xl * yh), F(xh * yl), F(xh * yh)]
    return min(c), max(c)


def tri_far_packet(p0, e1, e2, ol, oh, ul, uh):
    """device_math.hpp: tri_far_packet, binary32 operation by operation"""
    sl, sh = (ol - p0).astype(F), (oh - p0).astype(F)
    s1l, s1h = [None] * 3, [None] * 3
    for c, (i1, i2) in enumerate(((1, 2), (2, 0), (0, 1))):
        al, ah = imul_point(ul[i1], uh[i1], e2[i2])
        bl, bh = imul_point(ul[i2], uh[i2], e2[i1])
        s1l[c], s1h[c] = F(al - bh), F(ah - bl)
    x, y, z = (imul_point(s1l[c], s1h[c], e1[c]) for c in range(3))
    dl, dh = F(F(x[0] + y[0]) + z[0]), F(F(x[1] + y[1]) + z[1])
    x, y, z = (imul(s1l[c], s1h[c], sl[c], sh[c]) for c in range(3))
    ql, qh = F(F(x[0] + y[0]) + z[0]), F(F(x[1] + y[1]) + z[1])
    min_abs_q = max(max(ql, F(-qh)), F(0.0))
    max_abs_det = max(abs(dl), abs(dh))
    return bool(min_abs_q > F(F(1.5) * max_abs_det)), (dl, dh, ql, qh)


MAGS = np.array([0.0, 1e-45, 1e-41, 1e-38, 1e-30, 1e-20, 1e-6, 1e-3, 1.0, 1.0, 3.0, 1e3, 1e6, 4e9], dtype=np.float64)


def extreme(rng, shape, top=4e9):
    m = MAGS[MAGS <= top]
    return (rng.choice(m, size=shape) * rng.choice([-1.0, 1.0], size=shape) * rng.uniform(0.5, 1.0, size=shape)).astype(F)


def check_packet(p0, e1, e2, o, u):
    ol, oh, ul, uh = o.min(0), o.max(0), u.min(0), u.max(0)
    with np.errstate(all="ignore"):
        proven, (dl, dh, ql, qh) = tri_far_packet(p0, e1, e2, ol, oh, ul, uh)
        det, q = stage0(p0, e1, e2, o, u)
    assert np.isfinite([dl, dh, ql, qh]).all() and np.isfinite(det).all() and np.isfinite(q).all()   # the magnitude limits
    # the enclosure itself (stronger than the claim): every computed value inside its interval
    assert (det >= dl).all() and (det <= dh).all() and (q >= ql).all() and (q <= qh).all(), (dl, dh, ql, qh)
    if proven:
        assert tri_far(q, det).all()
    return proven, bool(tri_far(q, det).all())


def test_enclosure_on_coherent_packets_and_it_is_not_vacuous():
    rng = np.random.default_rng(4)
    proven = allfar = 0
    for it in range(1500):
        c = rng.uniform(-3, 3, 3)
        size = 10.0 ** rng.uniform(-3, 1)
        p0, p1, p2 = (c + rng.uniform(-size, size, 3) for _ in range(3))
        p0 = p0.astype(F); e1 = (p1.astype(F) - p0).astype(F); e2 = (p2.astype(F) - p0).astype(F)
        n = int(rng.integers(1, 65))
        spread_o = 10.0 ** rng.uniform(-4, 0) * (it % 3 != 0)          # a third of the packets share their origin (camera rays)
        spread_u = 10.0 ** rng.uniform(-4, -0.5)
        o0, u0 = rng.uniform(-8, 8, 3), rng.normal(size=3)
        o = (o0 + rng.uniform(-spread_o, spread_o, (n, 3))).astype(F)
        u = u0 / np.linalg.norm(u0) + rng.uniform(-spread_u, spread_u, (n, 3))
        u = (u / np.linalg.norm(u, axis=1, keepdims=True)).astype(F)
        a, b = check_packet(p0, e1, e2, o, u)
        proven += a; allfar += b
    assert proven > 600 and allfar >= proven                 # most coherent packets are settled by the box test


def test_enclosure_on_adversarial_operands():
    rng = np.random.default_rng(5)
    proven = 0
    for it in range(3000):
        kind = it % 4
        if kind == 0:
            p0, e1, e2 = extreme(rng, 3), extreme(rng, 3), extreme(rng, 3)
        elif kind == 1:                                       # degenerate: zero edge / collinear edges
            p0 = rng.uniform(-2, 2, 3).astype(F); e1 = rng.uniform(-1, 1, 3).astype(F)
            e2 = (e1 * F(rng.choice([0.0, 1.0, 2.0, -1.0]))).astype(F)
        elif kind == 2:                                       # tiny triangle at an ordinary place
            p0 = rng.uniform(-2, 2, 3).astype(F); e1 = (extreme(rng, 3) * F(1e-30)).astype(F); e2 = (extreme(rng, 3) * F(1e-30)).astype(F)
        else:
            p0 = rng.uniform(-2, 2, 3).astype(F); e1 = rng.uniform(-1, 1, 3).astype(F); e2 = rng.uniform(-1, 1, 3).astype(F)
        n = int(rng.integers(1, 65))
        o = np.where(rng.random((n, 3)) < 0.3, extreme(rng, (n, 3)), rng.uniform(-4, 4, (n, 3))).astype(F)
        u = np.where(rng.random((n, 3)) < 0.3, extreme(rng, (n, 3), top=1e6), rng.normal(size=(n, 3))).astype(F)
        if it % 5 == 0:                                       # a packet of identical rays: intervals are points
            o[:] = o[0]; u[:] = u[0]
        a, _ = check_packet(p0, e1, e2, o, u)
        proven += a
    assert proven > 100


def test_packet_that_straddles_the_triangle_is_never_proven():
    """rays that hit the triangle are inside the packet: the proof must fail (b1 in range means |q| <= |det| up to rounding)"""
    rng = np.random.default_rng(6)
    for it in range(300):
        p0 = rng.uniform(-2, 2, 3).astype(F); e1 = rng.uniform(-1, 1, 3).astype(F); e2 = rng.uniform(-1, 1, 3).astype(F)
        n = 16
        b = rng.uniform(0.1, 0.4, (n, 2))
        target = p0 + b[:, :1] * e1 + b[:, 1:] * e2
        o = np.tile(rng.uniform(-6, 6, 3), (n, 1)).astype(F)
        u = target - o
        u = (u / np.linalg.norm(u, axis=1, keepdims=True)).astype(F)
        a, allfar = check_packet(p0, e1, e2, o, u)
        assert not a and not allfar
