// The 1-D transforms of Winograd F(4x4,3x3) and the lane sums of its pooling epilogue, shared by conv_wino4.hip (the stride-1 encoder
// layers) and conv_enc12.hip (pconv1_1 computed inside pconv1_2's block).  Include inside an anonymous namespace.
#pragma once
#include "common.h"

// ---- the 1-D transforms (interpolation points 0, +-1, +-2, infinity), on PACKED f32 pairs (v_pk_fma_f32 / v_pk_add_f32 /
// v_pk_mul_f32): a packed instruction costs the issuing wave what a scalar one does (tools/micro/pk_valu.hip: 5.3 against 5.9
// cycles alone, 7.4 both beside MFMAs, one wave per SIMD), so every pair halves the transform's share of the pipe.
// B^T x:  t0 = 4x0 - 5x2 + x4, t1 = -4x1 - 4x2 + x3 + x4, t2 = 4x1 - 4x2 - x3 + x4, t3 = -2x1 - x2 + 2x3 + x4,
//         t4 = 2x1 - x2 - 2x3 + x4, t5 = 4x1 - 5x3 + x5                                          (12 VALU)
template <class V>
__device__ __forceinline__ void bt6(const V x[6], V t[6]) {
    const V a = x[4] - 4.f * x[2], b = x[3] - 4.f * x[1];
    const V c = x[4] - x[2], e = x[3] - x[1];
    t[0] = 4.f * x[0] + (x[4] - 5.f * x[2]);
    t[1] = a + b;
    t[2] = a - b;
    t[3] = c + 2.f * e;
    t[4] = c - 2.f * e;
    t[5] = 4.f * x[1] + (x[5] - 5.f * x[3]);
}
// the same transform along a ROW whose six values sit in the pairs (x0,x5) (x1,x2) (x3,x4) the column pass leaves: 8 VALU
//   (b,a) = (x3,x4) - 4 (x1,x2);  (t1,t2) = (a+b, a-b);  (e,c) = (x3,x4) - (x1,x2);  (t3,t4) = (c+2e, c-2e);  t0, t5 scalar
__device__ __forceinline__ void bt6_row(f32x2 p05, f32x2 p12, f32x2 p34, float v[6]) {
    const f32x2 ba = p34 - 4.f * p12;
    const f32x2 ec = p34 - p12;
    // (b + a, -b + a) and (2e + c, -2e + c): broadcasts of a pair's halves ride in the instruction's op_sel bits (v_pk_fma_f32 ...
    // op_sel:[0,0,1] op_sel_hi:[0,1,1]); plain vector code, so the compiler places the wait states a VALU result needs in front of an MFMA
    const f32x2 k1 = {1.f, -1.f}, k2 = {2.f, -2.f};
    const f32x2 t12 = __builtin_shufflevector(ba, ba, 0, 0) * k1 + __builtin_shufflevector(ba, ba, 1, 1);
    const f32x2 t34 = __builtin_shufflevector(ec, ec, 0, 0) * k2 + __builtin_shufflevector(ec, ec, 1, 1);
    v[0] = __builtin_fmaf(4.f, p05[0], __builtin_fmaf(-5.f, p12[1], p34[1]));
    v[1] = t12[0]; v[2] = t12[1]; v[3] = t34[0]; v[4] = t34[1];
    v[5] = __builtin_fmaf(4.f, p12[0], __builtin_fmaf(-5.f, p34[0], p05[1]));
}
// A^T m:  y0 = m0 + m1 + m2 + m3 + m4, y1 = m1 - m2 + 2(m3 - m4), y2 = m1 + m2 + 4(m3 + m4), y3 = m1 - m2 + 8(m3 - m4) + m5   (10 VALU)
template <class V>
__device__ __forceinline__ void at6(const V m[6], V y[4]) {
    const V s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = (m[0] + s12) + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = (d12 + 8.f * d34) + m[5];
}

template <int SW>
__device__ __forceinline__ float window_sum(float v) {          // sum over SW adjacent tiles (lanes) and the two tile rows (lane ^ 8)
    static_assert(SW == 2 || SW == 4 || SW == 8, "pool window in tiles");
    v = dpp_add<0xB1>(v);                       // quad_perm [1,0,3,2]
    if (SW >= 4) v = dpp_add<0x4E>(v);          // quad_perm [2,3,0,1]
    if (SW >= 8) v = dpp_add<0x141>(v);         // row_half_mirror: lane i <-> 7-i
    v = dpp_add<0x128>(v);                      // row_ror:8 - the other tile row of the group
    return v;
}

