/*
 * srt_defs.h — the two definitions this project had to INVENT because the reference
 * does not pin them; shared verbatim by the HIP kernel, the C++ host and the oracle
 * so that all three produce the same bits.
 *
 *  1. The random stream.  The reference draws from C rand() (Raytracer.cpp:93-95,
 *     165,182) with MSVC's RAND_MAX = 32767; its per-thread LCG state makes the
 *     stream depend on which worker visits which pixel, i.e. it is not reproducible
 *     by construction.  The draw ORDER is pinned by the reference; the generator is
 *     not.  We define a counter-based stream keyed by (seed, pixel, sample, draw#)
 *     that yields 15-bit integers, consumed exactly like rand(): (float)r / 32767.
 *
 *  2. powf.  GetEnvironmentColor calls powf (Raytracer.cpp:81,87).  MSVC UCRT, glibc
 *     and ROCm OCML disagree in the last ulp, so a bit-exact GPU/CPU comparison needs
 *     ONE definition.  srt_powf below uses only IEEE-754 double +,-,*,/, explicit fma and
 *     integer bit operations in a fixed order, so it gives identical bits on x86 and gfx950
 *     provided the translation unit is compiled with -ffp-contract=off.  It is
 *     faithfully rounded (tests bound it to <= 1 ulp of libm's powf on every float of (0,1]).
 *
 * Plain C99 / C++ / HIP.  The only libm symbol used on the host is fma().
 */
#ifndef SRT_DEFS_H
#define SRT_DEFS_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define SRT_HD __host__ __device__ static inline
#else
#define SRT_HD static inline
#endif

#define SRT_RAND_MAX 32767 /* MSVC RAND_MAX, the reference's platform */

/* ---- random stream --------------------------------------------------------------- */

/* 32-bit finalizer ("lowbias32" constants). */
SRT_HD uint32_t srt_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

/* Key of one path sample. pixel = x + y*W in scene coordinates (absolute, so any
 * partition of the image over wavefronts or GPUs reproduces the same frame);
 * sample = 1-based frame index (ACCUMULATIONFRAMES). */
SRT_HD uint32_t srt_rng_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t k = srt_mix32(seed ^ 0xA511E9B3U);
    k = srt_mix32(k + pixel);
    k = srt_mix32(k + sample);
    return k;
}

/* draw-th value (0-based) of the sample's stream, in [0, SRT_RAND_MAX] — stands for
 * one rand() call. */
SRT_HD uint32_t srt_rng_draw(uint32_t key, uint32_t draw) {
    return srt_mix32(key + draw * 0x9E3779B9U) >> 17;
}

/* ---- portable powf --------------------------------------------------------------- */

/* Explicit fused multiply-add in double.  IEEE-754 fma is correctly rounded, so libm's
 * fma() on the host and v_fma_f64 on gfx950 give the same bits; this is NOT implicit
 * contraction (the translation units are still compiled with -ffp-contract=off). */
#if defined(__HIP_DEVICE_COMPILE__)
#define SRT_FMA(a, b, c) __builtin_fma((a), (b), (c))
#else
#ifdef __cplusplus
extern "C" double fma(double, double, double);
#else
double fma(double, double, double);
#endif
#define SRT_FMA(a, b, c) fma((a), (b), (c))
#endif

typedef union srt_f64bits {
    double d;
    uint64_t u;
} srt_f64bits;
typedef union srt_f32bits {
    float f;
    uint32_t u;
} srt_f32bits;

/* ln(m) for m in [1,2) comes from a 32-entry table: i = the top five mantissa bits, c_i = 1 + (i + 0.5)/32,
 * pair (invc_i, lnc_i) = (RN(1/c_i), RN(-ln(invc_i))), so that ln(m) = lnc_i + ln(1 + s) with s = m*invc_i - 1
 * (one exact-enough FMA, |s| < 1/64) — no division, a degree-7 series.  Generated with 60-digit arithmetic by
 * tools/gen_pow_table.py. */
#define SRT_POW_TABLE_INIT { \
    0x1.f81f81f81f820p-1, 0x1.fc0a8b0fc03c4p-7,  /* c = 1 + 0.5/32 */ \
    0x1.e9131abf0b767p-1, 0x1.77458f632dcffp-5,  /* c = 1 + 1.5/32 */ \
    0x1.dae6076b981dbp-1, 0x1.341d7961bd1d0p-4,  /* c = 1 + 2.5/32 */ \
    0x1.cd85689039b0bp-1, 0x1.a926d3a4ad562p-4,  /* c = 1 + 3.5/32 */ \
    0x1.c0e070381c0e0p-1, 0x1.0d77e7cd08e5bp-3,  /* c = 1 + 4.5/32 */ \
    0x1.b4e81b4e81b4fp-1, 0x1.44d2b6ccb7d1cp-3,  /* c = 1 + 5.5/32 */ \
    0x1.a98ef606a63bep-1, 0x1.7ab890210d907p-3,  /* c = 1 + 6.5/32 */ \
    0x1.9ec8e951033d9p-1, 0x1.af3c94e80bff3p-3,  /* c = 1 + 7.5/32 */ \
    0x1.948b0fcd6e9e0p-1, 0x1.e27076e2af2e8p-3,  /* c = 1 + 8.5/32 */ \
    0x1.8acb90f6bf3aap-1, 0x1.0a324e27390e2p-2,  /* c = 1 + 9.5/32 */ \
    0x1.8181818181818p-1, 0x1.22941fbcf7966p-2,  /* c = 1 + 10.5/32 */ \
    0x1.78a4c8178a4c8p-1, 0x1.3a64c556945eap-2,  /* c = 1 + 11.5/32 */ \
    0x1.702e05c0b8170p-1, 0x1.51aad872df82ep-2,  /* c = 1 + 12.5/32 */ \
    0x1.6816816816817p-1, 0x1.686c81e9b14adp-2,  /* c = 1 + 13.5/32 */ \
    0x1.6058160581606p-1, 0x1.7eaf83b82afc2p-2,  /* c = 1 + 14.5/32 */ \
    0x1.58ed2308158edp-1, 0x1.947941c2116fbp-2,  /* c = 1 + 15.5/32 */ \
    0x1.51d07eae2f815p-1, 0x1.a9cec9a9a084ap-2,  /* c = 1 + 16.5/32 */ \
    0x1.4afd6a052bf5bp-1, 0x1.beb4d9da71b7ap-2,  /* c = 1 + 17.5/32 */ \
    0x1.446f86562d9fbp-1, 0x1.d32fe7e00ebd5p-2,  /* c = 1 + 18.5/32 */ \
    0x1.3e22cbce4a902p-1, 0x1.e744261d68789p-2,  /* c = 1 + 19.5/32 */ \
    0x1.3813813813814p-1, 0x1.faf588f78f31dp-2,  /* c = 1 + 20.5/32 */ \
    0x1.323e34a2b10bfp-1, 0x1.0723e5c1cdf41p-1,  /* c = 1 + 21.5/32 */ \
    0x1.2c9fb4d812ca0p-1, 0x1.109f39e2d4c96p-1,  /* c = 1 + 22.5/32 */ \
    0x1.27350b8812735p-1, 0x1.19ee6b467c96fp-1,  /* c = 1 + 23.5/32 */ \
    0x1.21fb78121fb78p-1, 0x1.23130d7bebf43p-1,  /* c = 1 + 24.5/32 */ \
    0x1.1cf06ada2811dp-1, 0x1.2c0e9ed448e8cp-1,  /* c = 1 + 25.5/32 */ \
    0x1.1811811811812p-1, 0x1.34e289d9ce1d2p-1,  /* c = 1 + 26.5/32 */ \
    0x1.135c81135c811p-1, 0x1.3d9026a7156fbp-1,  /* c = 1 + 27.5/32 */ \
    0x1.0ecf56be69c90p-1, 0x1.4618bc21c5ec2p-1,  /* c = 1 + 28.5/32 */ \
    0x1.0a6810a6810a7p-1, 0x1.4e7d811b75bb0p-1,  /* c = 1 + 29.5/32 */ \
    0x1.0624dd2f1a9fcp-1, 0x1.56bf9d5b3f399p-1,  /* c = 1 + 30.5/32 */ \
    0x1.0204081020408p-1, 0x1.5ee02a9241676p-1,  /* c = 1 + 31.5/32 */ \
}
/* On the device the table lives in constant memory (and, for the kernels, as a copy in LDS: srt_powf_tab); host code of a HIP
 * translation unit reads the _host copy — a __constant__ variable has no usable host value. */
/* The coefficients of srt_powf_tab's two polynomials and its range-reduction constants, as a table too: the kernels read them
 * from their LDS copy (one broadcast read per term) instead of building each 64-bit literal with two moves in front of its FMA.
 * [0..5] ln(1+s): 1/7, -1/6, 1/5, -1/4, 1/3, -1/2 | [6] LN2_HI (0x3fe62e42fee00000: e * LN2_HI is exact) | [7] LN2_LO
 * (0x3dea39ef35793c76) | [8] 1/ln 2 | [9..17] e^r: 1/11!, 1/10!, ..., 1/3! | [18], [19] unused */
#define SRT_POW_COEF_INIT { \
    1.0 / 7.0, -1.0 / 6.0, 1.0 / 5.0, -1.0 / 4.0, 1.0 / 3.0, -0.5, \
    6.93147180369123816490e-01, 1.90821492927058770002e-10, 1.44269504088896338700e+00, \
    1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, \
    0.0, 0.0 \
}
#if defined(__HIPCC__) || defined(__HIP__)
__device__ __constant__ const double srt_pow_table[64] = SRT_POW_TABLE_INIT;
__device__ __constant__ const double srt_pow_coef[20] = SRT_POW_COEF_INIT;
static const double srt_pow_table_host[64] = SRT_POW_TABLE_INIT;
static const double srt_pow_coef_host[20] = SRT_POW_COEF_INIT;
#else
static const double srt_pow_table[64] = SRT_POW_TABLE_INIT;
static const double srt_pow_coef[20] = SRT_POW_COEF_INIT;
#define srt_pow_table_host srt_pow_table
#define srt_pow_coef_host srt_pow_coef
#endif

/* x^y for x >= 0 (x<0 -> NaN), finite y > 0.  Covers every call the path makes:
 * powf(upd, 0.1f) with upd in (0,1] and powf(|upd|, .05f) with |upd| in [0,1].
 * ln x = e*ln2 + lnc_i + ln(1+s) and e^t = 2^k * e^r in double, every step an IEEE add / multiply / fma in a
 * fixed order: identical bits on x86 and gfx950.  Relative error before the final rounding < 2^-45, i.e. the
 * result is the correctly rounded x^y except for about one input in a million (tests/test_defs.py: <= 1 ulp
 * from libm's powf on EVERY float of (0,1] for both exponents). */
/* `table`: the 32 (invc, lnc) pairs above — srt_pow_table itself, or a copy of its bytes somewhere cheaper to read (the kernels
 * keep one in LDS next to the scene image; same values, same arithmetic, same bits).  `coef`: srt_pow_coef or a copy of it. */
SRT_HD float srt_powf_tab(float xf, float yf, const double* table, const double* coef) {
    srt_f32bits xb;
    xb.f = xf;
    if ((xb.u & 0x7fffffffU) > 0x7f800000U) return xf; /* NaN */
    if (xb.u == 0U || xb.u == 0x80000000U) return 0.0f;  /* +-0 ^ (y>0) */
    if (xb.u & 0x80000000U) {                            /* negative base */
        srt_f32bits n;
        n.u = 0x7fc00000U;
        return n.f;
    }
    if (xb.u == 0x7f800000U) return xf; /* +inf */
    if (xb.u == 0x3f800000U) return 1.0f;

    /* x = m * 2^e, m in [1,2); float subnormals are normal doubles. */
    srt_f64bits b;
    b.d = (double)xf;
    const int e = (int)((b.u >> 52) & 0x7ffU) - 1023;
    const int i = (int)((b.u >> 47) & 31U);
    b.u = (b.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    const double m = b.d;
    const double s = SRT_FMA(m, table[2 * i], -1.0);
    /* ln(1+s) = s + s^2 * (-1/2 + s/3 - s^2/4 + s^3/5 - s^4/6 + s^5/7) */
    double p = coef[0];
    p = SRT_FMA(p, s, coef[1]);
    p = SRT_FMA(p, s, coef[2]);
    p = SRT_FMA(p, s, coef[3]);
    p = SRT_FMA(p, s, coef[4]);
    p = SRT_FMA(p, s, coef[5]);
    const double lnm = SRT_FMA(s * s, p, s);
    const double LN2_HI = coef[6], LN2_LO = coef[7];
    const double ed = (double)e;
    const double lnx = SRT_FMA(ed, LN2_HI, table[2 * i + 1] + SRT_FMA(ed, LN2_LO, lnm));
    const double t = (double)yf * lnx;

    /* e^t = 2^k * e^r, |r| <= ln2/2, Taylor series through r^11 */
    if (t > 89.0) {
        srt_f32bits inf;
        inf.u = 0x7f800000U;
        return inf.f;
    }
    if (t < -104.0) return 0.0f;
    const double kd = t * coef[8];
    const int k = (int)(kd < 0.0 ? kd - 0.5 : kd + 0.5);
    const double kk = (double)k;
    const double r = SRT_FMA(-kk, LN2_LO, SRT_FMA(-kk, LN2_HI, t));
    double q = coef[9]; /* 1/11! */
    q = SRT_FMA(q, r, coef[10]);
    q = SRT_FMA(q, r, coef[11]);
    q = SRT_FMA(q, r, coef[12]);
    q = SRT_FMA(q, r, coef[13]);
    q = SRT_FMA(q, r, coef[14]);
    q = SRT_FMA(q, r, coef[15]);
    q = SRT_FMA(q, r, coef[16]);
    q = SRT_FMA(q, r, coef[17]);
    q = SRT_FMA(q, r, 0.5);
    q = SRT_FMA(q, r, 1.0);
    q = SRT_FMA(q, r, 1.0);
    /* scale by 2^k in two exact steps so that the only rounding is the final
     * double->float conversion (also correct when the float result is subnormal). */
    srt_f64bits sc;
    const int k1 = k / 2, k2 = k - k1;
    sc.u = (uint64_t)(1023 + k1) << 52;
    q = q * sc.d;
    sc.u = (uint64_t)(1023 + k2) << 52;
    q = q * sc.d;
    return (float)q;
}

SRT_HD float srt_powf(float xf, float yf) { return srt_powf_tab(xf, yf, srt_pow_table, srt_pow_coef); }

#endif /* SRT_DEFS_H */
