/* Which powf does the host's C library run?  glibc's routine (sysdeps/ieee754/flt-32/e_powf.c, from ARM's
 * optimized-routines) restated twice -- every a * b + c FUSED, as the build x86-64 machines with FMA select at run time
 * compiles it (sysdeps/x86_64/fpu/multiarch/e_powf-fma.c), and UNFUSED -- and held against powf() itself on random
 * operands of four kinds (a cosine raised to a shininess; any positive float raised to a shininess; raw bit patterns for
 * both; a cosine raised to a random exponent).  On this image (glibc 2.35) and on the GPU boxes:
 *     n=200000000 fused mismatches 0 plain mismatches 2
 * so the fused form is what yahr_amd/csrc/device_math.hpp restates for the device (glibc_powf).
 *   build:  gcc -O2 -mfma -ffp-contract=off -o /tmp/powf_check tools/powf_check.c -lm     run:  /tmp/powf_check 200000000
 * (exp2 table: bits(2^(i/32)) - (i << 47), generated with 60-digit decimal arithmetic; log2 table and coefficients: the
 * routine's published data) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
static const double LT[16][2] = {
  {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
  {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
  {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
  {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
  {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
  {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
  {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
  {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
static const double PA[5] = {0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2, -0x1.7154748bef6c8p-1, 0x1.71547652ab82bp0};
static const double PC[3] = {0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1};
static const uint64_t ET[32] = {
  0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
  0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
  0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
  0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
  0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
  0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
  0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
  0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};
static inline int checkint(uint32_t iy) {
  int e = iy >> 23 & 0xff;
  if (e < 0x7f) return 0;
  if (e > 0x7f + 23) return 2;
  if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
  if (iy & (1u << (0x7f + 23 - e))) return 1;
  return 2;
}
static inline int zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000 - 1; }
static inline int signalling(uint32_t ix) { return 2 * (ix ^ 0x00400000) > 2u * 0x7fc00000; }
#define DEFINE_MODEL(NAME, MADD)                                                                              \
  static float NAME(float x, float y) {                                                                       \
    uint32_t sign_bias = 0, ix = asuint(x), iy = asuint(y);                                                   \
    if (ix - 0x00800000 >= 0x7f800000 - 0x00800000 || zeroinfnan(iy)) {                                       \
      if (zeroinfnan(iy)) {                                                                                   \
        if (2 * iy == 0) return signalling(ix) ? x + y : 1.0f;                                                \
        if (ix == 0x3f800000) return signalling(iy) ? x + y : 1.0f;                                           \
        if (2 * ix > 2u * 0x7f800000 || 2 * iy > 2u * 0x7f800000) return x + y;                               \
        if (2 * ix == 2 * 0x3f800000) return 1.0f;                                                            \
        if ((2 * ix < 2 * 0x3f800000) == !(iy & 0x80000000)) return 0.0f;                                     \
        return y * y;                                                                                         \
      }                                                                                                       \
      if (zeroinfnan(ix)) {                                                                                   \
        float x2 = x * x;                                                                                     \
        if (ix & 0x80000000 && checkint(iy) == 1) x2 = -x2;                                                   \
        return iy & 0x80000000 ? 1 / x2 : x2;                                                                 \
      }                                                                                                       \
      if (ix & 0x80000000) {                                                                                  \
        int yint = checkint(iy);                                                                              \
        if (yint == 0) return (x - x) / (x - x);                                                              \
        if (yint == 1) sign_bias = 1u << (5 + 11);                                                            \
        ix &= 0x7fffffff;                                                                                     \
      }                                                                                                       \
      if (ix < 0x00800000) { ix = asuint(x * 0x1p23f); ix &= 0x7fffffff; ix -= 23 << 23; }                    \
    }                                                                                                         \
    uint32_t tmp = ix - 0x3f330000; int i = (tmp >> 19) % 16; uint32_t top = tmp & 0xff800000;                \
    uint32_t iz = ix - top; int k = (int32_t)top >> 23;                                                       \
    double invc = LT[i][0], logc = LT[i][1], z = (double)asfloat(iz);                                         \
    double r = MADD(z, invc, -1.0), y0 = logc + (double)k, r2 = r * r;                                        \
    double yy = MADD(PA[0], r, PA[1]), p = MADD(PA[2], r, PA[3]), r4 = r2 * r2, q = MADD(PA[4], r, y0);       \
    q = MADD(p, r2, q); yy = MADD(yy, r4, q);                                                                 \
    double ylogx = (double)y * yy;                                                                            \
    if ((asuint64(ylogx) >> 47 & 0xffff) >= asuint64(126.0) >> 47) {                                          \
      if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();              \
      if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;                                                   \
    }                                                                                                         \
    double kd = ylogx + 0x1.8p+52 / 32; uint64_t ki = asuint64(kd); kd -= 0x1.8p+52 / 32;                     \
    double rr = ylogx - kd; uint64_t t = ET[ki % 32]; t += (ki + sign_bias) << 47; double s = asdouble(t);    \
    double zz = MADD(PC[0], rr, PC[1]), rr2 = rr * rr, o = MADD(PC[2], rr, 1.0);                              \
    o = MADD(zz, rr2, o); o = o * s;                                                                          \
    return (float)o;                                                                                          \
  }
#define MADD_FUSED(a, b, c) __builtin_fma((a), (b), (c))
#define MADD_PLAIN(a, b, c) ((a) * (b) + (c))
DEFINE_MODEL(model_fused, MADD_FUSED)
DEFINE_MODEL(model_plain, MADD_PLAIN)
static uint64_t rng = 88172645463325252ULL;
static inline uint64_t nxt(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
int main(int argc, char **argv) {
  long n = argc > 1 ? atol(argv[1]) : 20000000, bf = 0, bp = 0;
  const float ys[] = {1, 2, 3, 4, 5, 8, 10, 16, 20, 25, 32, 40, 50, 64, 100, 128, 200, 256, 500, 1000, 0.5f, 1.5f, 2.5f, 17.3f, 0.1f, 33.3f};
  for (long j = 0; j < n; j++) {
    uint64_t r = nxt(); float x, y; int mode = j & 3;
    if (mode == 0) { x = (float)((r >> 11) * (1.0 / 9007199254740992.0)); y = ys[(r & 1023) % 26]; }
    else if (mode == 1) { uint32_t u = (uint32_t)(r >> 32) & 0x7fffffff; memcpy(&x, &u, 4); y = ys[(r & 1023) % 26]; }
    else if (mode == 2) { uint32_t u = (uint32_t)(r >> 32), v = (uint32_t)r; memcpy(&x, &u, 4); memcpy(&y, &v, 4); }
    else { x = (float)((r >> 11) * (1.0 / 9007199254740992.0)); y = (float)((r & 0xfffff) / 1024.0); }
    float ref = powf(x, y), a = model_fused(x, y), b = model_plain(x, y);
    int nanr = ref != ref;
    if (!(nanr && a != a) && asuint(ref) != asuint(a)) { if (bf < 5) printf("fused  mismatch x=%a y=%a ref=%a got=%a\n", x, y, ref, a); bf++; }
    if (!(nanr && b != b) && asuint(ref) != asuint(b)) { if (bp < 5) printf("plain  mismatch x=%a y=%a ref=%a got=%a\n", x, y, ref, b); bp++; }
  }
  printf("n=%ld fused mismatches %ld plain mismatches %ld\n", n, bf, bp);
  return bf != 0;
}
