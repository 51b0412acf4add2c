// Per-element math of the five KGE score functions in the query / candidate decomposition
// (see oracle/closed.py for the float64 statement that pins it, and DESIGN.md section 3).
//
//   score(i, x) = c0 - m * sum_k f(q_i[k], x[k])        q_i = query built from the two fixed operands
//
// Reference formulas: models/transe.py:65-76, rotate.py:69-99, complex.py:65-85, distmult.py:63-75,
// protate.py:74-93 (paths relative to /root/reference).  fp32 throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mkb_hip.h"

namespace mkb {

// A "unit" is the smallest group of floats the pair function couples: one float for the real models,
// one complex number (re at k, im at d+k) for RotatE.  ComplEx's pair function is a plain dot product
// over the 2d floats of the query, so it only needs complex units when BUILDING the query.
template <int MODEL> struct ModelTraits;
template <> struct ModelTraits<MKB_TRANSE>   { static constexpr bool cplx_query = false, cplx_pair = false, uses_gamma = true;  };
template <> struct ModelTraits<MKB_DISTMULT> { static constexpr bool cplx_query = false, cplx_pair = false, uses_gamma = false; };
template <> struct ModelTraits<MKB_PROTATE>  { static constexpr bool cplx_query = false, cplx_pair = false, uses_gamma = true;  };
template <> struct ModelTraits<MKB_COMPLEX>  { static constexpr bool cplx_query = true,  cplx_pair = false, uses_gamma = false; };
template <> struct ModelTraits<MKB_ROTATE>   { static constexpr bool cplx_query = true,  cplx_pair = true,  uses_gamma = true;  };

struct Cplx { float re, im; };

// sin and cos of one argument in ~25 VALU instructions: Cody-Waite reduction by pi/2 (three constants whose products with the
// quadrant number are exact) + the Cephes single-precision polynomials on [-pi/4, pi/4].  Max abs error 9.2e-8 for
// |x| <= 1e5 (libm's sinf / cosf: 7e-8; checked over 2 M points per range in numpy), growing like |x| * 6e-8 beyond that --
// the arguments here are phases r / (range / pi) or (h + r - t) / (range / pi) of table entries, O(pi).  libm's sinf + cosf
// inline ~200 instructions EACH with their large-argument paths; they sat in every row kernel (per-launch code is paid per
// byte: the instruction cache is cold at every launch) and, unrolled per pair term, made pRotatE's pooled kernels 6 MB of
// the library.  NaN / Inf give NaN like libm.
__device__ __forceinline__ void sincos_f32(float x, float &s, float &c) {
    const float q = rintf(x * 0.63661977236758134308f);
    float r = fmaf(-q, 1.5703125f, x);
    r = fmaf(-q, 4.837512969970703125e-4f, r);
    r = fmaf(-q, 7.54978995489188216e-8f, r);
    const float z = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                          fmaf(-0.5f, z, 1.0f));
    const int n = (int)q;
    const float a = (n & 1) ? pc : ps, b = (n & 1) ? ps : pc;  // quadrant: sin, cos swap in the odd ones
    s = (n & 2) ? -a : a;
    c = ((n + 1) & 2) ? -b : b;
}
__device__ __forceinline__ float sin_f32(float x) { float s, c; sincos_f32(x, s, c); return s; }

// ---------------------------------------------------------------- query build (real models)
// a, b = the two fixed operands: tail-style (h, r), head-style (r, t).
template <int MODEL, bool HEAD>
__device__ __forceinline__ float build_q_real(float a, float b, float kd) {
    if constexpr (MODEL == MKB_TRANSE) return HEAD ? (a - b) : (a + b);            // r - t | h + r
    if constexpr (MODEL == MKB_DISTMULT) return a * b;                              // r * t | h * r
    if constexpr (MODEL == MKB_PROTATE) return HEAD ? (a / kd - b / kd) : (a / kd + b / kd);
    return 0.f;
}

// ---------------------------------------------------------------- query build (complex models)
// tail-style: e = h (entity), r = relation.  head-style: e = t.
// RotatE: r_re holds the phase source (relation value), r_im unused.
template <int MODEL, bool HEAD>
__device__ __forceinline__ Cplx build_q_cplx(Cplx e, Cplx r, float kd) {
    float c, s;
    if constexpr (MODEL == MKB_ROTATE) {
        sincos_f32(r.re / kd, s, c);
    } else {
        c = r.re;
        s = r.im;
    }
    Cplx q;
    if constexpr (HEAD) {  // conj(rot) (x) t     rotate.py:84-85, complex.py:74-75
        q.re = c * e.re + s * e.im;
        q.im = c * e.im - s * e.re;
    } else {               // h (x) rot           rotate.py:89-90, complex.py:79-80
        q.re = e.re * c - e.im * s;
        q.im = e.re * s + e.im * c;
    }
    return q;
}

// ---------------------------------------------------------------- pair terms (forward)
template <int MODEL, bool HEAD>
__device__ __forceinline__ float pair_term_real(float q, float x, float kd) {
    if constexpr (MODEL == MKB_TRANSE) return fabsf(HEAD ? (x + q) : (q - x));
    if constexpr (MODEL == MKB_DISTMULT || MODEL == MKB_COMPLEX) return q * x;
    if constexpr (MODEL == MKB_PROTATE) return fabsf(sin_f32(HEAD ? (x / kd + q) : (q - x / kd)));
    return 0.f;
}

// v_sqrt_f32 / v_rsq_f32 directly (1 ulp): the correctly-rounded sqrtf() expands to ~20 instructions and was
// 3x the cost of the whole pair term; the error budget is 1e-4 on a sum of <= 1000 terms of size ~1e-2.
__device__ __forceinline__ float pair_term_cmod(Cplx q, Cplx x) {  // rotate.py:86-87 / 91-96
    const float a = q.re - x.re, b = q.im - x.im;
    return __builtin_amdgcn_sqrtf(a * a + b * b);
}

template <int MODEL>
__device__ __forceinline__ float finish_score(float sum, float gamma, float modulus) {
    if constexpr (MODEL == MKB_TRANSE || MODEL == MKB_ROTATE) return gamma - sum;
    if constexpr (MODEL == MKB_PROTATE) return gamma - sum * modulus;
    return sum;
}

// ---------------------------------------------------------------- pair terms (backward)
// Given upstream g = dL/dscore, returns the contribution to dq and to dx.
// extra accumulates sum_k |sin z| for pRotatE's modulus gradient.
template <int MODEL, bool HEAD>
__device__ __forceinline__ void pair_bwd_real(float q, float x, float g, float kd, float modulus, float &dq,
                                              float &dx, float &extra) {
    if constexpr (MODEL == MKB_TRANSE) {
        // d|z| / dz = sign(z) in {-1, 0, 1} (torch's abs backward: 0 at the origin).  As a SIGNED-INTEGER median of z's bit
        // pattern: positive floats are positive integers, negative floats negative ones, +0 is 0 -- one v_med3_i32 and a
        // conversion instead of two compare / select pairs (the pair term is a sign and two adds: round 4's pooled TransE
        // backward spent 7 VALU operations per element, 5 now).
        // ACCEPTED DEVIATIONS from torch's abs backward at two edge values: z = -0 reads as negative (sign -1 where torch
        // gives 0) -- a sum or difference of finite numbers is +0 when it is zero EXCEPT (-0) + (-0) and (-0) - (+0), i.e. only when
        // a table entry is itself a negative zero, which the reference's uniform_ initialisation and Adam updates do not produce; and
        // z = NaN gives +-1 instead of propagating NaN into this element's gradient (the loss is NaN then anyway: bench.py and
        // Pipeline stop on a non-finite loss).  Canonicalising with z + 0.0f would cost a sixth VALU operation per element in a
        // kernel bound by exactly that count.
        const float z = HEAD ? (x + q) : (q - x);
        int si;  // (written as min(max(zi, -1), 1) the compiler emits two compare / select pairs again)
        asm("v_med3_i32 %0, %1, -1, 1" : "=v"(si) : "v"(__float_as_uint(z)));
        const float sg = (float)si;
        dq = -g * sg;
        dx = HEAD ? (-g * sg) : (g * sg);
    } else if constexpr (MODEL == MKB_DISTMULT || MODEL == MKB_COMPLEX) {
        dq = g * x;
        dx = g * q;
    } else if constexpr (MODEL == MKB_PROTATE) {
        const float z = HEAD ? (x / kd + q) : (q - x / kd);
        float sz, cz;
        sincos_f32(z, sz, cz);
        const float sg = (sz > 0.f) ? 1.f : ((sz < 0.f) ? -1.f : 0.f);
        const float c = cz * sg * modulus;
        dq = -g * c;
        dx = (HEAD ? (-g * c) : (g * c)) / kd;
        extra += fabsf(sz);
    }
}

__device__ __forceinline__ void pair_bwd_cmod(Cplx q, Cplx x, float g, Cplx &dq, Cplx &dx) {
    const float a = q.re - x.re, b = q.im - x.im;
    const float n2 = a * a + b * b;
    // torch's norm backward gives 0 at the origin: with the clamp, a = b = 0 yields w * 0 = 0 as well
    const float w = g * __builtin_amdgcn_rsqf(fmaxf(n2, 1e-30f));
    dx.re = w * a;
    dx.im = w * b;
    dq.re = -dx.re;
    dq.im = -dx.im;
}

// Two complex units at once, written on 2-vectors so that the compiler emits v_pk_add/mul/fma_f32 without
// reshuffling registers: 5 VALU + 2 transcendental (backward: 8 + 2) instructions per TWO pairs instead of 6 (9) + 1
// per pair.  (re, im) of the two units live in separate 2-vectors: qr = (re_k, re_k+1), qi = (im_k, im_k+1).
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pair_term_cmod2(f2 qr, f2 qi, f2 xr, f2 xi) {
    const f2 a = qr - xr, b = qi - xi;
    const f2 n2 = a * a + b * b;
    return f2{__builtin_amdgcn_sqrtf(n2.x), __builtin_amdgcn_sqrtf(n2.y)};
}

// acc_r / acc_i accumulate  -g * u  (the dq sign); dx = -dq, so the x pass accumulates the same and negates once.
__device__ __forceinline__ void pair_bwd_cmod2(f2 qr, f2 qi, f2 xr, f2 xi, float g, f2 &acc_r, f2 &acc_i) {
    const f2 a = qr - xr, b = qi - xi;
    // a = b = 0 -> w * 0 = 0 (torch's sub-gradient at the origin).  Two packed fmas: the epsilon rides the first one
    // (a*a + b*b + eps as written costs mul + fma + add = one more packed op per pair, 8 -> 7).
    const f2 n2 = __builtin_elementwise_fma(b, b, __builtin_elementwise_fma(a, a, f2{1e-30f, 1e-30f}));
    const f2 w = f2{__builtin_amdgcn_rsqf(n2.x), __builtin_amdgcn_rsqf(n2.y)} * g;
    acc_r -= w * a;
    acc_i -= w * b;
}

// One evaluation of the pair term feeds BOTH gradients (single-pass backward kernel): dq accumulates -g*u, dx +g*u.
// 9 packed ops + 2 v_rsq per two complex dims (two separate passes cost 2 x (7 + 2)).
__device__ __forceinline__ void pair_bwd_cmod2_both(f2 qr, f2 qi, f2 xr, f2 xi, float g, f2 &dq_r, f2 &dq_i, f2 &dx_r,
                                                    f2 &dx_i) {
    const f2 a = qr - xr, b = qi - xi;
    const f2 n2 = __builtin_elementwise_fma(b, b, __builtin_elementwise_fma(a, a, f2{1e-30f, 1e-30f}));
    const f2 w = f2{__builtin_amdgcn_rsqf(n2.x), __builtin_amdgcn_rsqf(n2.y)} * g;
    dq_r = __builtin_elementwise_fma(-w, a, dq_r);
    dq_i = __builtin_elementwise_fma(-w, b, dq_i);
    dx_r = __builtin_elementwise_fma(w, a, dx_r);
    dx_i = __builtin_elementwise_fma(w, b, dx_i);
}

// Two rows against the same candidate, written stage by stage so that the two dependent chains (sub -> fma -> fma -> rsq ->
// mul -> fma) interleave: a lone chain leaves an idle issue slot after most of its packed ops (the compiler pads them
// with s_nop), two chains fill each other's slots.
__device__ __forceinline__ void pair_bwd_cmod2_both_x2(f2 qrA, f2 qiA, f2 qrB, f2 qiB, f2 xr, f2 xi, float gA, float gB,
                                                       f2 &dqrA, f2 &dqiA, f2 &dqrB, f2 &dqiB, f2 &dx_r, f2 &dx_i) {
    const f2 eps = f2{1e-30f, 1e-30f};
    const f2 aA = qrA - xr, aB = qrB - xr, bA = qiA - xi, bB = qiB - xi;
    f2 nA = __builtin_elementwise_fma(aA, aA, eps), nB = __builtin_elementwise_fma(aB, aB, eps);
    nA = __builtin_elementwise_fma(bA, bA, nA);
    nB = __builtin_elementwise_fma(bB, bB, nB);
    const f2 rA = f2{__builtin_amdgcn_rsqf(nA.x), __builtin_amdgcn_rsqf(nA.y)};
    const f2 rB = f2{__builtin_amdgcn_rsqf(nB.x), __builtin_amdgcn_rsqf(nB.y)};
    const f2 wA = rA * gA, wB = rB * gB;
    dqrA = __builtin_elementwise_fma(-wA, aA, dqrA);
    dqrB = __builtin_elementwise_fma(-wB, aB, dqrB);
    dqiA = __builtin_elementwise_fma(-wA, bA, dqiA);
    dqiB = __builtin_elementwise_fma(-wB, bB, dqiB);
    dx_r = __builtin_elementwise_fma(wA, aA, dx_r);
    dx_i = __builtin_elementwise_fma(wA, bA, dx_i);
    dx_r = __builtin_elementwise_fma(wB, aB, dx_r);
    dx_i = __builtin_elementwise_fma(wB, bB, dx_i);
}

// ---------------------------------------------------------------- query backward
// Chain dq through build_q into the two fixed operands (a, b as in build_q_*).
template <int MODEL, bool HEAD>
__device__ __forceinline__ void query_bwd_real(float a, float b, float dq, float kd, float &da, float &db) {
    if constexpr (MODEL == MKB_TRANSE) { da = dq; db = HEAD ? -dq : dq; }
    if constexpr (MODEL == MKB_DISTMULT) { da = dq * b; db = dq * a; }
    if constexpr (MODEL == MKB_PROTATE) { da = dq / kd; db = (HEAD ? -dq : dq) / kd; }
}

// e = entity operand (h tail-style, t head-style), r = relation operand.  Returns de (complex) and
// dr: ComplEx -> complex gradient of r; RotatE -> dr.re = gradient of the phase source, dr.im = 0.
template <int MODEL, bool HEAD>
__device__ __forceinline__ void query_bwd_cplx(Cplx e, Cplx r, Cplx dq, float kd, Cplx &de, Cplx &dr) {
    float c, s;
    if constexpr (MODEL == MKB_ROTATE) {
        sincos_f32(r.re / kd, s, c);
    } else {
        c = r.re;
        s = r.im;
    }
    float dc, ds;
    if constexpr (HEAD) {  // q.re = c e.re + s e.im ; q.im = c e.im - s e.re
        de.re = dq.re * c - dq.im * s;
        de.im = dq.re * s + dq.im * c;
        dc = dq.re * e.re + dq.im * e.im;
        ds = dq.re * e.im - dq.im * e.re;
    } else {               // q.re = e.re c - e.im s ; q.im = e.re s + e.im c
        de.re = dq.re * c + dq.im * s;
        de.im = -dq.re * s + dq.im * c;
        dc = dq.re * e.re + dq.im * e.im;
        ds = -dq.re * e.im + dq.im * e.re;
    }
    if constexpr (MODEL == MKB_ROTATE) {
        dr.re = (ds * c - dc * s) / kd;  // d/dphase then / kd
        dr.im = 0.f;
    } else {
        dr.re = dc;
        dr.im = ds;
    }
}

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace mkb
