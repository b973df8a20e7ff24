"""The argument behind the octant run's missing sleep index, checked WITHOUT a GPU (render_kernel.hip: oct_run).

In the NaN-free regime of the FAST loops (finite boxes with lo <= hi, finite origin, finite non-zero reciprocal direction)
the slab test of bbRayIntersection (AABBs.hs:29-40) is, per axis, tnear = max(.., (near plane - o) * inv) and
tfar = min(.., (far plane - o) * inv), with 0 and tMax as the starting values.  Claim: if a child's box lies inside its
parent's (lo_c >= lo_p, hi_c <= hi_p), the ray is the same and tMax has not grown, then a ray that FAILS the parent's test
fails the child's -- in binary32 as computed, not just over the reals -- because subtraction, multiplication and rounding to
nearest are monotone.  So a lane needs no memory of the subtrees it has left.  This file restates the test in numpy
binary32, operation by operation, and checks the implication on random and adversarial pairs of boxes: shared planes,
planes one ulp apart, degenerate (flat, point) boxes, magnitudes from subnormal to 1e30 (overflowing differences and
products: infinities are monotone too), rays that start inside / behind / on a plane, tMax from the subnormals to "no
limit" (a NaN), every direction octant -- and that it is not vacuous."""
import numpy as np

F = np.float32


def slab_fast(lo, hi, o, inv, tmax):
    """device_math.hpp: slab_hit_fast / the octant run's step, rows of boxes against rows of rays -> hit mask.
    (hardware min / max: a NaN operand is dropped; the only NaN the regime admits is tMax = "no limit")"""
    with np.errstate(over="ignore", invalid="ignore"):
        t0 = ((lo - o).astype(F) * inv).astype(F)
        t1 = ((hi - o).astype(F) * inv).astype(F)
        near = np.where(inv > 0, t0, t1)                       # the octant form: WHICH plane is near is the sign of inv
        far = np.where(inv > 0, t1, t0)
        tnear = np.maximum(np.maximum(near[:, 0], near[:, 1]), np.maximum(near[:, 2], F(0.0)))
        tfar = np.fmin(np.minimum(np.minimum(far[:, 0], far[:, 1]), far[:, 2]), tmax)       # fmin: v_min_f32 drops a NaN tMax
        return ~(tnear > tfar)


MAGS = np.array([1e-45, 1e-41, 1e-38, 1e-30, 1e-10, 1e-3, 0.1, 1.0, 1.0, 2.5, 1e3, 1e6, 4e9, 1e20, 1e30], dtype=np.float64)


def values(rng, shape, wild):
    v = rng.uniform(-4, 4, shape)
    if wild:
        v = np.where(rng.random(shape) < 0.35, rng.choice(MAGS, size=shape) * rng.choice([-1.0, 1.0], size=shape) * rng.uniform(0.5, 1.0, shape), v)
        v = np.where(rng.random(shape) < 0.05, 0.0, v)
    return v.astype(F)


def nested_pairs(rng, n, wild):
    a, b = values(rng, (n, 3), wild), values(rng, (n, 3), wild)
    plo, phi = np.minimum(a, b), np.maximum(a, b)
    # child: two points inside the parent (per axis), then the adversarial variants
    u, v = rng.random((n, 3)), rng.random((n, 3))
    with np.errstate(over="ignore", invalid="ignore"):
        c0 = (plo + (phi - plo) * u.astype(F)).astype(F)
        c1 = (plo + (phi - plo) * v.astype(F)).astype(F)
    c0 = np.where(np.isfinite(c0), c0, plo); c1 = np.where(np.isfinite(c1), c1, phi)
    c0 = np.clip(c0, plo, phi); c1 = np.clip(c1, plo, phi)
    clo, chi = np.minimum(c0, c1), np.maximum(c0, c1)
    kind = rng.integers(0, 6, (n, 3))
    clo = np.where(kind == 0, plo, clo); chi = np.where(kind == 0, phi, chi)                       # the same slab
    clo = np.where(kind == 1, plo, clo)                                                            # shares the low plane
    chi = np.where(kind == 2, phi, chi)                                                            # shares the high plane
    up = np.nextafter(plo, F(np.inf)); dn = np.nextafter(phi, F(-np.inf))
    ok = up <= dn
    clo = np.where((kind == 3) & ok, up, clo); chi = np.where((kind == 3) & ok, np.maximum(dn, up), chi)   # one ulp inside
    chi = np.where(kind == 4, clo, chi)                                                            # flat child
    clo = np.minimum(clo, chi)
    assert (clo >= plo).all() and (chi <= phi).all() and (clo <= chi).all()
    return plo, phi, clo, chi


def rays(rng, n, wild):
    o = values(rng, (n, 3), wild)
    d = values(rng, (n, 3), wild)
    d = np.where(d == 0, F(1.0), d)
    with np.errstate(over="ignore", divide="ignore"):
        inv = (F(1.0) / d).astype(F)
    good = np.isfinite(inv) & (inv != 0)
    inv = np.where(good, inv, F(1.0))
    tp = np.abs(values(rng, n, wild))
    tp = np.where(rng.random(n) < 0.15, F(np.nan), tp).astype(F)            # "no limit"
    shrink = rng.random(n).astype(F)
    with np.errstate(invalid="ignore"):
        tc = np.where(rng.random(n) < 0.5, tp, (tp * shrink).astype(F)).astype(F)          # tMax only shrinks; NaN -> NaN or a number
    tc = np.where(np.isnan(tp) & (rng.random(n) < 0.5), np.abs(values(rng, n, wild)), tc).astype(F)
    return o, inv, tp, tc


def test_a_ray_that_fails_a_box_fails_every_box_inside_it():
    rng = np.random.default_rng(20260)
    seen_parent_miss = seen_child_hit = 0
    for wild in (False, True):
        for _ in range(8):
            n = 250_000
            plo, phi, clo, chi = nested_pairs(rng, n, wild)
            o, inv, tp, tc = rays(rng, n, wild)
            # rays that matter: aim half of them at the box
            aim = rng.random(n) < 0.5
            with np.errstate(over="ignore", invalid="ignore"):
                mid = (F(0.5) * plo + F(0.5) * phi).astype(F)
            mid = np.where(np.isfinite(mid), mid, plo)
            o = np.where(aim[:, None] & (rng.random((n, 3)) < 0.7), mid, o).astype(F)
            hp = slab_fast(plo, phi, o, inv, tp)
            hc = slab_fast(clo, chi, o, inv, tc)
            bad = ~hp & hc
            assert not bad.any(), (int(bad.sum()), [(plo[j], phi[j], clo[j], chi[j], o[j], inv[j], tp[j], tc[j]) for j in np.flatnonzero(bad)[:3]])
            seen_parent_miss += int((~hp).sum()); seen_child_hit += int(hc.sum())
    assert seen_parent_miss > 500_000 and seen_child_hit > 300_000          # not vacuous either way


def test_the_claim_needs_its_premises():
    """(so that the test above is known to be able to fail) a child that sticks out by one ulp, or a tMax that GROWS, does
    produce a ray that fails the parent and enters the child"""
    rng = np.random.default_rng(7)
    n = 400_000
    plo, phi, clo, chi = nested_pairs(rng, n, False)
    o, inv, tp, tc = rays(rng, n, False)
    chi_out = np.nextafter(phi, F(np.inf))                                   # child's high plane one ulp outside
    o2 = np.where(inv > 0, np.nextafter(phi, F(np.inf)), o).astype(F)        # start just beyond the parent's high plane
    hp = slab_fast(plo, phi, o2, inv, F(np.nan) * np.ones(n, F))
    hc = slab_fast(plo, chi_out, o2, inv, F(np.nan) * np.ones(n, F))
    assert (~hp & hc).any()
    hp = slab_fast(plo, phi, o, inv, np.zeros(n, F))                         # parent seen with tMax = 0 ...
    hc = slab_fast(clo, chi, o, inv, np.full(n, 1e6, F))                     # ... the child with a larger one
    assert (~hp & hc).any()
