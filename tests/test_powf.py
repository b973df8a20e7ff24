"""`**` on Float (BSDF.hs:25) is the C library's powf.  The device evaluates glibc's routine restated in binary64
(yahr_amd/csrc/device_math.hpp: glibc_powf; glibc 2.35 sysdeps/ieee754/flt-32/e_powf.c in the form x86-64 machines with
FMA run); the oracle calls the C library itself.  Here, without a GPU: the SAME source compiled for the host
(yahr_debug_powf_host) against the oracle's call (oracle_powf_array), bit for bit -- random operands, the operands the
Blinn term sees (a cosine in [0, 1] raised to a shininess), raw bit patterns, and every special case of the routine's
front end.  tests/test_gpu_parity.py::test_powf_is_the_hosts does the same with the device's evaluation."""
import ctypes as C

import numpy as np

from oracle_py import lib as oracle_lib
from yahr_amd._lib import lib

SHININESS = np.float32([1, 2, 3, 4, 5, 8, 10, 16, 20, 25, 32, 40, 50, 64, 100, 128, 200, 256, 500, 1000, 0.5, 1.5, 2.5, 17.3, 0.1, 33.3])


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def powf_operands(n, seed):
    rng = np.random.default_rng(seed)
    q = n // 4
    bits = lambda m, hi: rng.integers(0, hi, m, dtype=np.uint64).astype(np.uint32).view(np.float32)
    x = np.concatenate([rng.random(q, dtype=np.float32), bits(q, 2 ** 31), bits(q, 2 ** 32), rng.random(q, dtype=np.float32)])
    y = np.concatenate([rng.choice(SHININESS, q), rng.choice(SHININESS, q), bits(q, 2 ** 32), (rng.integers(0, 2 ** 20, q) / 1024.0).astype(np.float32)])
    return np.ascontiguousarray(x), np.ascontiguousarray(y)


def special_operands():
    """the cross product of everything the front end of the routine branches on"""
    u = lambda b: np.uint32(b).view(np.float32)
    v = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, u(0x7fc00000), u(0xffc00000), u(0x7f800001), u(0x7fa00000),   # quiet and signalling NaNs
         u(1), u(0x007fffff), u(0x00800000), -u(1), 0.5, -0.5, 2.0, -2.0, 3.0, -3.0, 2.5, -2.5, 1e-30, 1e30, -1e30, 0.99999994, 1.0000001,
         16777216.0, 16777217.0, 33554432.0, -16777216.0, 126.0, 127.0, 128.0, -149.0, -150.0, -151.0, 1024.0, 3.4e38, -3.4e38]
    v = np.array(v, dtype=np.float32)
    x, y = np.meshgrid(v, v, indexing="ij")
    return np.ascontiguousarray(x.reshape(-1)), np.ascontiguousarray(y.reshape(-1))


def same_floats(a, b):
    """bit for bit; a NaN for a NaN (the sign and payload of an invalid result are not part of any contract upstream)"""
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def libm_powf(x, y):
    out = np.zeros(len(x), np.float32)
    oracle_lib().oracle_powf_array(len(x), _fp(x), _fp(y), _fp(out))
    return out


def test_restated_powf_is_the_c_librarys():
    for x, y in (powf_operands(2_000_000, 5), special_operands()):
        ref = libm_powf(x, y)
        got = np.zeros(len(x), np.float32)
        assert lib.yahr_debug_powf_host(len(x), _fp(x), _fp(y), _fp(got)) == 0
        bad = np.flatnonzero(~same_floats(ref, got))
        assert len(bad) == 0, [(x[j].view(np.uint32), y[j].view(np.uint32), ref[j], got[j]) for j in bad[:8]]
    assert np.isfinite(ref).any() and np.isnan(ref).any()
    # the routine's own known answers
    x = np.float32([2, 4, 0.25, 0.0, 0.0, np.inf, 1.0, np.nan, 0.5]); y = np.float32([10, 0.5, 2, 3, -1, -2, np.nan, 0.0, 1074])
    got = np.zeros(len(x), np.float32)
    lib.yahr_debug_powf_host(len(x), _fp(x), _fp(y), _fp(got))
    assert got.tolist() == [1024.0, 2.0, 0.0625, 0.0, np.inf, 0.0, 1.0, 1.0, 0.0]
