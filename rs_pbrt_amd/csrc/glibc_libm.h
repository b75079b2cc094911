// Rust's f32::sin / cos / ln / log2 / exp / acos / atan2 are the host libm's sinf / cosf / logf / log2f / expf / acosf / atan2f.  This file
// restates what glibc 2.35 (x86-64) does for each of them, operation by operation, so that the device returns the same float for every
// argument and the reference's transcendental functions stop being a source of last-ulp differences:
//   sinf, cosf, logf, log2f, expf   ARM optimized-routines algorithms in double precision, libm's own tables (__logf_data, __log2f_data,
//                                   __exp2f_data, read from the installed libm.so.6), fused multiply-adds where the x86-64 build has them;
//   acosf, atanf, atan2f            the fdlibm float routines in plain float arithmetic.
// The file is compiled twice: by hipcc into every kernel (included from dev_math.h; no contraction, correctly rounded / and sqrt), and by
// gcc into tools/libm_exhaustive.c, which compares every function with the host's libm over all 2^32 floats (2^30 pairs for atan2f):
// 0 mismatches (tests/test_host.py).  tests/test_gpu_trace.py then compares the device's results with the GPU box's libm.
// Third-party algorithms restated here (not the reference's code), with their upstream licences:
//   * sinf / cosf / logf / log2f / expf follow glibc 2.35 sysdeps/ieee754/flt-32/{s_sincosf.h, s_sincosf_data.c, e_logf.c, e_log2f.c, e_expf.c},
//     which glibc took from ARM's Optimized Routines — Copyright (c) Arm Limited, MIT OR Apache-2.0 WITH LLVM-exception upstream; distributed by
//     glibc under the GNU Lesser General Public License v2.1 or later (Copyright (C) Free Software Foundation, Inc.).  The polynomial
//     coefficients below are those routines'; the lookup tables are NOT copied: they are read at run time from the installed libm.so.6.
//   * acosf / atanf / atan2f follow fdlibm's e_acosf.c / s_atanf.c / e_atan2f.c as shipped in glibc — Copyright (C) 1993 by Sun Microsystems,
//     Inc. All rights reserved.  Developed at SunPro, a Sun Microsystems, Inc. business.  Permission to use, copy, modify, and distribute this
//     software is freely granted, provided that this notice is preserved.
// The includer defines GL_FN / GL_FN_COLD (function qualifiers: sinf and cosf sit on the shading path and are inlined, the others are called),
// GL_TABLE (table qualifiers) and the bit casts GL_F2U / GL_U2F / GL_D2U / GL_U2D.
#pragma once
#include <math.h>
#include <stdint.h>

// ---- sinf / cosf as glibc computes them -------------------------------------------------------------------------------------
// Rust's f32::sin / f32::cos are the host libm's sinf / cosf.  glibc (2.28 and later) evaluates them in double precision: argument
// reduction by n = round(x * 2 / pi) (a scaled float-to-int conversion), a degree-7 sine or degree-8 cosine polynomial in the reduced
// argument, one rounding to float at the end (sysdeps/ieee754/flt-32/s_sincosf.h, from ARM's optimized routines; x86-64 selects the
// build with fused multiply-adds).  The same operations in the same order here give the same float for every argument — checked
// exhaustively on the CPU against the host's libm for all 2^32 inputs with |x| < 120 (tools/sincos_exhaustive.c: 0 mismatches) and on
// the GPU box in tests/test_gpu_trace.py — which takes sin / cos out of the list of things that differ from the reference by an ulp.
// Arguments of 120 and beyond (never produced on this path: angles are 2 pi u or smaller) go to the device library.
GL_FN float glibc_sincos_poly(double x, double x2, bool neg_cos, int n) {
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
        const double x7 = x3 * x2;
        const double s = fma(x3, -0x1.555545995a603p-3, x);
        return (float)fma(x7, s1, s);
    }
    const double sg = neg_cos ? -1.0 : 1.0;  // the second table row holds the cosine polynomial negated (quadrants 2, 3)
    const double x4 = x2 * x2;
    const double c2 = fma(x2, sg * 0x1.99343027bf8c3p-16, sg * -0x1.6c087e89a359dp-10);
    const double c1 = fma(x2, sg * -0x1.ffffffd0c621cp-2, sg * 0x1p0);
    const double x6 = x4 * x2;
    const double c = fma(x4, sg * 0x1.55553e1068f19p-5, c1);
    return (float)fma(x6, c2, c);
}
GL_FN uint32_t glibc_abstop12(float x) { return (GL_F2U(x) >> 20) & 0x7ffu; }
GL_FN float glibc_sincosf(float y, int cosine) {
    double x = (double)y;
    if (glibc_abstop12(y) < glibc_abstop12(0x1.921FB6p-1f)) {  // |y| < pi / 4
        if (glibc_abstop12(y) < glibc_abstop12(0x1p-12f)) return cosine ? 1.0f : y;
        return glibc_sincos_poly(x, x * x, false, cosine);
    }
    if (glibc_abstop12(y) < glibc_abstop12(120.0f)) {
        const double r = x * 0x1.45F306DC9C883p+23;          // 2 / pi, prescaled by 2^24
        const int32_t n = ((int32_t)r + 0x800000) >> 24;
        x = fma(-(double)n, 0x1.921FB54442D18p0, x);          // x - n * pi / 2
        const int32_t q = n + cosine;
        const double sgn = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;   // sign[q & 3] = {1, -1, -1, 1}
        return glibc_sincos_poly(x * sgn, x * x, (q & 2) != 0, n ^ cosine);
    }
    return cosine ? cosf(y) : sinf(y);
}
GL_FN float rspt_sinf(float y) { return glibc_sincosf(y, 0); }
GL_FN float rspt_cosf(float y) { return glibc_sincosf(y, 1); }

// ---- logf / log2f / expf / acosf / atan2f as glibc 2.35 computes them ------------------------------------------------------------
// Same idea as sinf / cosf above.  logf, log2f, expf: ARM optimized-routines algorithms in double precision with libm's own tables
// (__logf_data, __log2f_data, __exp2f_data), fused multiply-adds where the x86-64 build has them; acosf, atanf, atan2f: the fdlibm
// float routines in plain float arithmetic (this file is compiled without contraction, with correctly rounded division and square
// root).  tools/libm_exhaustive.c is this code compiled for the host: 0 mismatches against the host libm over all 2^32 floats (2^30
// pairs for atan2f); tests/test_gpu_trace.py compares the device with the GPU box's libm.
GL_TABLE double GLIBC_LOGF_TAB[32] = {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2, 0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2, 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3, 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3, 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4, 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1.0000000000000p+0, 0x0.0p+0, 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5, 0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4, 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3, 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3, 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2, 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2};
GL_TABLE double GLIBC_LOG2F_TAB[32] = {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2, 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2, 0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2, 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2, 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2, 0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3, 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3, 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4, 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5, 0x1.0000000000000p+0, 0x0.0p+0, 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4, 0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3, 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3, 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2, 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2, 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2};
GL_TABLE uint64_t GLIBC_EXP2F_TAB[32] = {0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
GL_FN_COLD float rspt_logf(float x) {
    uint32_t ix = GL_F2U(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {  // x < 0x1p-126, inf or nan
        if (ix * 2u == 0u) return -__builtin_huge_valf();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return GL_U2F(0x7fc00000u);
        ix = GL_F2U(x * 0x1p23f);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) % 16u;
    const int32_t k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = GLIBC_LOGF_TAB[2 * i], logc = GLIBC_LOGF_TAB[2 * i + 1];
    const double z = (double)GL_U2F(iz);
    const double r = fma(z, invc, -1.0);
    const double y0 = fma((double)k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = r * r;
    double y = fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = fma(-0x1.00ea348b88334p-2, r2, y);
    y = fma(y, r2, y0 + r);
    return (float)y;
}
GL_FN float rspt_log2f(float x) {
    uint32_t ix = GL_F2U(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -__builtin_huge_valf();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return GL_U2F(0x7fc00000u);
        ix = GL_F2U(x * 0x1p23f);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) % 16u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int32_t k = (int32_t)tmp >> 23;
    const double invc = GLIBC_LOG2F_TAB[2 * i], logc = GLIBC_LOG2F_TAB[2 * i + 1];
    const double z = (double)GL_U2F(iz);
    const double r = fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double y = fma(0x1.ecabf496832e0p-2, r, -0x1.715479ffae3dep-1);
    y = fma(-0x1.712b6f70a7e4dp-2, r2, y);
    const double p = fma(0x1.715475f35c8b8p+0, r, y0);
    y = fma(y, r2, p);
    return (float)y;
}
GL_FN_COLD float rspt_expf(float x) {
    const double xd = (double)x;
    const uint32_t abstop = (GL_F2U(x) >> 20) & 0x7ffu;
    if (abstop >= (0x42b00000u >> 20)) {  // |x| >= 88 or nan
        if (GL_F2U(x) == 0xff800000u) return 0.0f;
        if (abstop >= (0x7f800000u >> 20)) return x + x;
        if (x > 0x1.62e42ep6f) return __builtin_huge_valf();
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    const double z = 0x1.71547652b82fep+5 * xd;
    double kd = z + 0x1.8p52;
    const uint64_t ki = GL_D2U(kd);
    kd -= 0x1.8p52;
    const double r = fma(0x1.71547652b82fep+5, xd, -kd);   // (the host build fuses z - kd with the product z = InvLn2N * xd)
    uint64_t t = GLIBC_EXP2F_TAB[ki % 32u];
    t += ki << 47;
    const double s = GL_U2D(t);
    const double zz = fma(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
    const double r2 = r * r;
    double y = fma(0x1.62e42ff0c52d6p-6, r, 1.0);
    y = fma(zz, r2, y);
    y = y * s;
    return (float)y;
}
GL_FN_COLD float rspt_acosf(float x) {  // e_acosf.c
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f, pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f,
                pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
                qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    float z, p, q, r, w, s, c, df;
    const int32_t hx = (int32_t)GL_F2U(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) { if (hx > 0) return 0.0f; return pi + 2.0f * pio2_lo; }
    else if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {
        if (ix <= 0x23000000) return pio2_hi + pio2_lo;
        z = x * x;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx < 0) {
        z = (one + x) * 0.5f;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        s = sqrtf(z);
        r = p / q;
        w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    } else {
        z = (one - x) * 0.5f;
        s = sqrtf(z);
        df = GL_U2F(GL_F2U(s) & 0xfffff000u);
        c = (z - df * df) / (s + df);
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        w = r * s + c;
        return 2.0f * (df + w);
    }
}
GL_FN_COLD float rspt_atanf(float x) {  // s_atanf.c
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;   // atanhi / atanlo: atan(0.5), atan(1), atan(1.5), atan(inf)
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f,
                          -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    float w, s1, s2, z;
    int32_t id;
    const int32_t hx = (int32_t)GL_F2U(x), ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        if (hx > 0) return hi3 + lo3;
        return -hi3 - lo3;
    }
    if (ix < 0x3ee00000) {  // |x| < 0.4375
        if (ix < 0x31000000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float h = id == 0 ? hi0 : (id == 1 ? hi1 : (id == 2 ? hi2 : hi3)), l = id == 0 ? lo0 : (id == 1 ? lo1 : (id == 2 ? lo2 : lo3));
    z = h - ((x * (s1 + s2) - l) - x);
    return hx < 0 ? -z : z;
}
GL_FN_COLD float rspt_atan2f(float y, float x) {  // e_atan2f.c
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    float z;
    const int32_t hx = (int32_t)GL_F2U(x), ix = hx & 0x7fffffff, hy = (int32_t)GL_F2U(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return rspt_atanf(y);
    const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) { if (m < 2) return y; return m == 2 ? pi + tiny : -pi - tiny; }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) { if (m == 0) return pi_o_4 + tiny; if (m == 1) return -pi_o_4 - tiny; if (m == 2) return 3.0f * pi_o_4 + tiny; return -3.0f * pi_o_4 - tiny; }
        if (m == 0) return 0.0f; if (m == 1) return -0.0f; if (m == 2) return pi + tiny; return -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = rspt_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return GL_U2F(GL_F2U(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

