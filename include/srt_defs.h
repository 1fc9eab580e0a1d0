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
 *     faithfully rounded (tests bound it to <= 1 ulp of libm's powf).
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

/* x^y for x >= 0 (x<0 -> NaN), finite y > 0.  Covers every call the path makes:
 * powf(upd, 0.1f) with upd in (0,1] and powf(|upd|, .05f) with |upd| in [0,1]. */
SRT_HD float srt_powf(float xf, float yf) {
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

    /* x = m * 2^e, m in [sqrt(1/2), sqrt(2)); float subnormals are normal doubles. */
    srt_f64bits b;
    b.d = (double)xf;
    int e = (int)((b.u >> 52) & 0x7ffU) - 1023;
    b.u = (b.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL; /* m in [1,2) */
    double m = b.d;
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e = e + 1;
    }
    /* ln m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716 */
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 23.0;
    p = SRT_FMA(p, z, 1.0 / 21.0);
    p = SRT_FMA(p, z, 1.0 / 19.0);
    p = SRT_FMA(p, z, 1.0 / 17.0);
    p = SRT_FMA(p, z, 1.0 / 15.0);
    p = SRT_FMA(p, z, 1.0 / 13.0);
    p = SRT_FMA(p, z, 1.0 / 11.0);
    p = SRT_FMA(p, z, 1.0 / 9.0);
    p = SRT_FMA(p, z, 1.0 / 7.0);
    p = SRT_FMA(p, z, 1.0 / 5.0);
    p = SRT_FMA(p, z, 1.0 / 3.0);
    p = SRT_FMA(p, z, 1.0);
    double lnm = 2.0 * s * p;
    const double LN2_HI = 6.93147180369123816490e-01; /* 0x3fe62e42fee00000 */
    const double LN2_LO = 1.90821492927058770002e-10; /* 0x3dea39ef35793c76 */
    double ed = (double)e;
    double lnx = SRT_FMA(ed, LN2_HI, SRT_FMA(ed, LN2_LO, lnm));
    double t = (double)yf * lnx;

    /* e^t = 2^k * e^r */
    if (t > 89.0) {
        srt_f32bits inf;
        inf.u = 0x7f800000U;
        return inf.f;
    }
    if (t < -104.0) return 0.0f;
    double kd = t * 1.44269504088896338700e+00;
    int k = (int)(kd < 0.0 ? kd - 0.5 : kd + 0.5);
    double kk = (double)k;
    double r = SRT_FMA(-kk, LN2_LO, SRT_FMA(-kk, LN2_HI, t));
    double q = 1.0 / 6227020800.0; /* 1/13! */
    q = SRT_FMA(q, r, 1.0 / 479001600.0);
    q = SRT_FMA(q, r, 1.0 / 39916800.0);
    q = SRT_FMA(q, r, 1.0 / 3628800.0);
    q = SRT_FMA(q, r, 1.0 / 362880.0);
    q = SRT_FMA(q, r, 1.0 / 40320.0);
    q = SRT_FMA(q, r, 1.0 / 5040.0);
    q = SRT_FMA(q, r, 1.0 / 720.0);
    q = SRT_FMA(q, r, 1.0 / 120.0);
    q = SRT_FMA(q, r, 1.0 / 24.0);
    q = SRT_FMA(q, r, 1.0 / 6.0);
    q = SRT_FMA(q, r, 0.5);
    q = SRT_FMA(q, r, 1.0);
    q = SRT_FMA(q, r, 1.0);
    /* scale by 2^k in two exact steps so that the only rounding is the final
     * double->float conversion (also correct when the float result is subnormal). */
    srt_f64bits sc;
    int k1 = k / 2, k2 = k - k1;
    sc.u = (uint64_t)(1023 + k1) << 52;
    q = q * sc.d;
    sc.u = (uint64_t)(1023 + k2) << 52;
    q = q * sc.d;
    return (float)q;
}

#endif /* SRT_DEFS_H */
