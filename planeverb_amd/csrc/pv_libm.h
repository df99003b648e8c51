// pv_libm.h -- log10f and powf with the exact results of glibc 2.35's libm (the libm the strict-IEEE reference
// build links), usable from device and host code.
//
// The analysis evaluates log10f once per tail sample (Analyzer.cpp:311) and powf once per cell (Analyzer.cpp:230).
// ROCm's device functions are within 1-2 ulp of glibc's, and neither is correctly rounded, so to get the
// reference's bits the published algorithms glibc uses are restated here:
//   * log10f: sysdeps/ieee754/flt-32/e_log10f.c (fdlibm lineage): x = 2^k*m, result = (k*log10_2lo + ivln10*logf(m))
//     + k*log10_2hi in float arithmetic;
//   * logf, powf: sysdeps/ieee754/flt-32/e_logf.c, e_powf.c + e_powf_log2_data.c, e_exp2f_data.c (ARM Optimized
//     Routines, Szabolcs Nagy): 16-entry log tables, 32-entry exp2 table, polynomials evaluated in double.
// tools/libm_check.cpp compares them with the host libm: every positive finite float for log10f (2 139 095 039
// values) and every non-negative float for powf(x, 0.8f) (2 139 095 041 values) match bit for bit on this image.
// Must be compiled without FP contraction (-ffp-contract=off); results do not depend on it for these inputs (checked
// both ways), but the step kernels need the flag anyway.
//
// Licences of the restated material (full notices in /LICENSE): ARM Optimized Routines -- Copyright (c) 2017-2018, Arm
// Limited, MIT; fdlibm -- Copyright (C) 1993 by Sun Microsystems, Inc. ("Permission to use, copy, modify, and distribute
// this software is freely granted, provided that this notice is preserved").
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define PV_HD __host__ __device__
#else
#define PV_HD
#endif

namespace pva {

PV_HD inline uint32_t pvBitsF(float f) { return __builtin_bit_cast(uint32_t, f); }
PV_HD inline float pvFloatBits(uint32_t u) { return __builtin_bit_cast(float, u); }

// logf for a positive NORMAL argument (the only kind log10f passes in)
PV_HD inline float pvLogfNormal(float x) {
    constexpr double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
        {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
        {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
        {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
        {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    const uint32_t ix = pvBitsF(x);
    if (ix == 0x3f800000u) return 0.f;
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = T[i][0], logc = T[i][1];
    const double z = (double)pvFloatBits(iz);
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * 0x1.62e42fefa39efp-1;
    const double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

PV_HD inline float pvLog10f(float x) {
    const float two25 = 3.3554432000e+07f, ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f,
                log10_2lo = 7.9034151668e-07f;
    int hx = (int)pvBitsF(x);
    int k = 0;
    if (hx < 0x00800000) {                                                          // x < 2^-126
        if ((hx & 0x7fffffff) == 0) return -two25 / pvFloatBits((uint32_t)hx & 0x7fffffffu);  // log(+-0) = -inf
        if (hx < 0) return (x - x) / (x - x);                                       // log(-#) = NaN
        k -= 25;
        x *= two25;                                                                 // subnormal: scale up
        hx = (int)pvBitsF(x);
    }
    if (hx >= 0x7f800000) return x + x;
    k += (hx >> 23) - 127;
    const int i = (int)(((unsigned)k & 0x80000000u) >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
    const float y = (float)(k + i);
    const float m = pvFloatBits((uint32_t)hx);
    const float z = y * log10_2lo + ivln10 * pvLogfNormal(m);
    return z + y * log10_2hi;
}

// The same function without control flow, for arguments >= +0 (what the RT60 loop feeds it: a running sum of
// squares): every special case is a select, so that several evaluations written one after the other form ONE basic
// block and the compiler interleaves their dependent chains and table reads.  With one thread per cell and fewer
// waves than SIMDs (the live module's grids) the loop is bound by exactly that latency: ~1800 cycles per sample with
// the branching form.  Bit-identical to pvLog10f for every x >= +0, inf and NaN included (tools/libm_check.cpp).
// (the table fetch is a functor: the RT60 kernels of pv_rt60.hip keep the 16 entries in LDS -- a per-lane index into a
// constant array is a global load whose latency sits at the head of every evaluation's dependent chain -- and the host
// check of tools/libm_check.cpp runs the very same arithmetic through pvLog10fNonNeg below)
// FUSED multiply-adds.  The reference's libm evaluates logf's polynomial in double and rounds once to float; whether the five
// multiply-add pairs of that evaluation are fused or not changes no float result for ANY argument log10f can hand to logf (the
// normalised mantissas: exponent fields 0x7e / 0x7f, 2^24 values -- checked exhaustively, tools/libm_fma_check.cpp, each pair
// alone and all together: 0 of 16 777 216 differ; glibc's own x86-64 build selects an FMA variant of logf at run time for the
// same reason).  So the branch-free form below fuses them: 10 double operations per logarithm instead of 15 -- and y0 = logc +
// kk ln2 (kk in {-1, 0, 1}) comes out of the table functor, which may hold the 48 values ready-made (pv_rt60.hip: LDS): 8.
// The decay-time kernels are bound by exactly these operations (profiles/r04_rt60.txt).
PV_HD inline double pvFma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return __builtin_fma(a, b, c);  // (host: libm's fma where the target has no instruction -- correct either way)
#endif
}

struct PvLogTabConst {
    PV_HD void operator()(int i, int kk, double* invc, double* y0) const {
        constexpr double T[16][2] = {
            {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
            {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
            {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
            {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
            {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
            {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
            {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
            {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
        *invc = T[i][0];
        *y0 = pvFma((double)kk, 0x1.62e42fefa39efp-1, T[i][1]);
    }
};

template <class TabF>
PV_HD inline float pvLog10fNonNegT(float x, const TabF& tab) {
    const float two25 = 3.3554432000e+07f, ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f,
                log10_2lo = 7.9034151668e-07f;
    const int hx0 = (int)pvBitsF(x);
    const bool tiny = hx0 < 0x00800000;                 // +0 or subnormal (x is not negative)
    const bool zero = hx0 == 0;
    const bool infnan = hx0 >= 0x7f800000;
    const float xs = tiny ? x * two25 : x;              // subnormal: scale up
    int hx = (int)pvBitsF(xs);
    int k = (tiny ? -25 : 0) + (hx >> 23) - 127;
    const int i10 = (int)(((unsigned)k & 0x80000000u) >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i10) << 23);
    const float yk = (float)(k + i10);
    // logf of the normalised mantissa m in [1/2, 2) (pvLogfNormal, without its early return for m == 1)
    const uint32_t ix = (uint32_t)hx;
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int kk = (int)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    double invc, y0;
    tab(i, kk, &invc, &y0);  // y0 = logc + kk ln2
    const double z = (double)pvFloatBits(iz);
    const double r = pvFma(z, invc, -1.0);
    const double r2 = r * r;
    double y = pvFma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = pvFma(-0x1.00ea348b88334p-2, r2, y);
    y = pvFma(y, r2, y0 + r);
    const float lm = ix == 0x3f800000u ? 0.f : (float)y;
    const float zf = yk * log10_2lo + ivln10 * lm;
    const float res = zf + yk * log10_2hi;
    return zero ? -two25 / 0.f : infnan ? x + x : res;
}

PV_HD inline float pvLog10fNonNeg(float x) { return pvLog10fNonNegT(x, PvLogTabConst{}); }

// ... and for a positive NORMAL finite argument only (0x00800000 <= bits < 0x7f800000): the selects for zero, subnormals,
// inf and NaN are gone, and so is logf's early return for a mantissa of exactly 1 -- table entry 9 is {1, 0}, so the
// polynomial gives r = 0 and y = y0 = +0 there by itself.  This is what the lane-per-cell decay-time kernel (pv_rt60.hip)
// evaluates per sample; a wave whose running energy is not normal in some lane (it is a sum of squares: zero before the
// first non-zero sample, subnormal for a few samples after it) takes pvLog10fNonNegT for that chunk instead.  Bit-identical
// to pvLog10f on its whole domain (tools/libm_check.cpp, every normal float).
PV_HD inline bool pvIsNormalPositive(float x) { return pvBitsF(x) - 0x00800000u < 0x7f000000u; }

template <class TabF>
PV_HD inline float pvLog10fNormalT(float x, const TabF& tab) {
    const float ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f;
    const int hx = (int)pvBitsF(x);
    const int k = (hx >> 23) - 127;
    const int i10 = (int)((unsigned)k >> 31);
    const uint32_t ix = ((uint32_t)hx & 0x007fffffu) | ((uint32_t)(0x7f - i10) << 23);
    const float yk = (float)(k + i10);
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int kk = (int)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    double invc, y0;
    tab(i, kk, &invc, &y0);
    const double z = (double)pvFloatBits(iz);
    const double r = pvFma(z, invc, -1.0);
    const double r2 = r * r;
    double y = pvFma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = pvFma(-0x1.00ea348b88334p-2, r2, y);
    y = pvFma(y, r2, y0 + r);
    const float lm = (float)y;
    const float zf = yk * log10_2lo + ivln10 * lm;
    return zf + yk * log10_2hi;
}

PV_HD inline float pvLog10fNormal(float x) { return pvLog10fNormalT(x, PvLogTabConst{}); }

// N of them STAGE BY STAGE: the same operations as pvLog10fNormalT on N independent arguments, written so that every stage
// is N independent instructions and (on the device) kept in that order by scheduling barriers.  One evaluation is a chain of
// ~25 dependent operations, eight of them in double precision; the compiler's scheduler, left alone, runs two or three
// evaluations at a time to save registers, and a kernel with one or two waves per SIMD (pv_rt60_tile_kernel: 1.6 on average
// for 100 000 cells) then issues one instruction per ~12 cycles instead of one per 4.
#if defined(__HIP_DEVICE_COMPILE__)
#define PV_STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define PV_STAGE_FENCE() \
    do {                 \
    } while (0)
#endif
template <int N, class TabF>
PV_HD inline void pvLog10fNormalBatch(const float (&x)[N], float (&out)[N], const TabF& tab) {
    const float ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f;
    float yk[N];
    uint32_t iz[N];
    double invc[N], y0[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int hx = (int)pvBitsF(x[n]);
        const int k = (hx >> 23) - 127;
        const int i10 = (int)((unsigned)k >> 31);
        const uint32_t ix = ((uint32_t)hx & 0x007fffffu) | ((uint32_t)(0x7f - i10) << 23);
        yk[n] = (float)(k + i10);
        const uint32_t tmp = ix - 0x3f330000u;
        iz[n] = ix - (tmp & (0x1ffu << 23));
        tab((int)((tmp >> 19) & 15u), (int)tmp >> 23, &invc[n], &y0[n]);
    }
    PV_STAGE_FENCE();
    double r[N], r2[N], y[N];
#pragma unroll
    for (int n = 0; n < N; ++n) r[n] = pvFma((double)pvFloatBits(iz[n]), invc[n], -1.0);
    PV_STAGE_FENCE();
#pragma unroll
    for (int n = 0; n < N; ++n) {
        r2[n] = r[n] * r[n];
        y[n] = pvFma(0x1.5575b0be00b6ap-2, r[n], -0x1.ffffef20a4123p-2);
        y0[n] = y0[n] + r[n];
    }
    PV_STAGE_FENCE();
#pragma unroll
    for (int n = 0; n < N; ++n) y[n] = pvFma(-0x1.00ea348b88334p-2, r2[n], y[n]);
    PV_STAGE_FENCE();
#pragma unroll
    for (int n = 0; n < N; ++n) y[n] = pvFma(y[n], r2[n], y0[n]);
    PV_STAGE_FENCE();
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float zf = yk[n] * log10_2lo + ivln10 * (float)y[n];
        out[n] = zf + yk[n] * log10_2hi;
    }
    PV_STAGE_FENCE();
}

// powf for x >= 0 (zero, subnormal, inf and NaN included) and a positive finite y with |y * log2(x)| < 126 -- the
// analysis calls it with y = 0.8f, for which the overflow / underflow branches of glibc's powf cannot be taken.
PV_HD inline float pvPowf(float x, float y) {
    constexpr double LT[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
        {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
        {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
        {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
        {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
        {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
    constexpr uint64_t ET[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    uint32_t ix = pvBitsF(x);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {          // zero, subnormal, inf, NaN
        if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) return x * x;  // +0 -> 0, inf -> inf, NaN -> NaN (y > 0)
        ix = pvBitsF(x * 0x1p23f);
        ix &= 0x7fffffffu;
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int)top >> 23;
    const double invc = LT[i][0], logc = LT[i][1];
    const double z = (double)pvFloatBits(iz);
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double yy = 0x1.27616c9496e0bp-2 * r + -0x1.71969a075c67ap-2;
    const double p = 0x1.ec70a6ca7baddp-2 * r + -0x1.7154748bef6c8p-1;
    const double r4 = r2 * r2;
    double q = 0x1.71547652ab82bp+0 * r + y0;
    q = p * r2 + q;
    yy = yy * r4 + q;                       // log2(x)
    const double xd = (double)y * yy;
    double kd = xd + 0x1.8p+47;             // round to a multiple of 1/32
    const uint64_t ki = __builtin_bit_cast(uint64_t, kd);
    kd -= 0x1.8p+47;
    const double rr = xd - kd;
    uint64_t t = ET[ki & 31ull];
    t += ki << (52 - 5);
    const double sc = __builtin_bit_cast(double, t);
    const double zz = 0x1.c6af84b912394p-5 * rr + 0x1.ebfce50fac4f3p-3;
    const double rr2 = rr * rr;
    double e = 0x1.62e42ff0c52d6p-1 * rr + 1.0;
    e = zz * rr2 + e;
    e = e * sc;
    return (float)e;
}

}  // namespace pva
