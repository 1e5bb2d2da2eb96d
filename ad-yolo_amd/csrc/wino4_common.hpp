// Shared by the two F(4x4,3x3) forward kernels (wino4.hip: one patch per workgroup; wino4p.hpp: persistent workgroups): tile
// geometry, the six-point data transform, the output transform, the VGPR-form MFMA chain.  See wino4.hip for the algorithm.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "wino_common.hpp"

namespace adyolo {
namespace w4 {

constexpr float PA = 0.75f, PB = 1.5f;                 // interpolation points +-PA, +-PB (besides 0 and infinity)
constexpr float A2 = PA * PA, B2 = PB * PB, S2 = A2 + B2, P2 = A2 * B2;
constexpr float A3 = PA * PA * PA, B3 = PB * PB * PB;

typedef unsigned int u32x4_t __attribute__((__vector_size__(16)));

template <int TC>
struct Cfg {
    static constexpr int TR = 32 / TC;                  // tile rows of a workgroup
    static constexpr int LOG_TC = TC == 1 ? 0 : TC == 2 ? 1 : TC == 4 ? 2 : 3;
    static constexpr int PR = 4 * TR + 2;               // patch rows
    static constexpr int PS = PR * TC + 2;              // plane stride in 16-byte slots (== 2 mod 8: conflict-free writes)
    static constexpr int CBS = 12 * PS + 4;             // slots per 8-channel buffer: 6 nu x 2 channel quads (== 4 mod 8)
    static constexpr int CBUF = CBS * 4;                // floats per buffer
    static constexpr int RS = 256 / (4 * TC);           // patch rows per full staging round (16 channels = 4 quads per pixel)
    static constexpr int CBP = 72;                      // epilogue exchange row (64 channels + pad)
    static constexpr int EXCH = 8 * 32 * CBP;
    static constexpr int LDS_FLOATS = 4 * CBUF > 2 * EXCH ? 4 * CBUF : 2 * EXCH;   // the epilogue exchanges two output rows per round
};


// six-point data transform B^T along one direction, per component
#define ADYOLO_W4_BT(F)                                                                   \
    {                                                                                     \
        const float e12 = fmaf(-B2, c[2].F, c[4].F), o12 = fmaf(-B2, c[1].F, c[3].F);      \
        const float e34 = fmaf(-A2, c[2].F, c[4].F), o34 = fmaf(-A2, c[1].F, c[3].F);      \
        const float t0 = fmaf(P2, c[0].F, fmaf(-S2, c[2].F, c[4].F));                     \
        const float t5 = fmaf(P2, c[1].F, fmaf(-S2, c[3].F, c[5].F));                     \
        t[0].F = t0;                                                                      \
        t[1].F = fmaf(PA, o12, e12);                                                      \
        t[2].F = fmaf(-PA, o12, e12);                                                     \
        t[3].F = fmaf(PB, o34, e34);                                                      \
        t[4].F = fmaf(-PB, o34, e34);                                                     \
        t[5].F = t5;                                                                      \
    }
// (t may alias c)
__device__ __forceinline__ void bt6(const float4 (&c)[6], float4 (&t)[6]) {
    ADYOLO_W4_BT(x) ADYOLO_W4_BT(y) ADYOLO_W4_BT(z) ADYOLO_W4_BT(w)
}
// Packed fp32 fma d = k * a + c on <4 x float> (two v_pk_fma_f32), k wave-uniform.  INLINE ASSEMBLY on purpose: this LLVM unpacks
// packed fp32 instructions that sit in the shadow of an MFMA into two scalar ones (it assumes they co-issue with the matrix
// pipe); next to the fp32 MFMA nothing co-issues (tools/micro/mfma32_coissue.hip) and with one wave per SIMD a packed
// instruction issues in the time of a scalar one (profiles/r03_valu_rate.txt: 6.0 vs 6.1 cycles), so the unpacking doubled the
// transform cost of the pair loop.  (a - b: there is no
// v_pk_sub_f32 and the backend scalarises a packed subtraction -- pkfma4(-1, b, a).)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 pkfma4(float k, f32x4 a, f32x4 c) {
    const f32x2 kk = {k, k};                      // (an aligned SGPR pair: a lone 32-bit SGPR may be odd-numbered, which the packed
    f32x2 lo, hi;                                 //  encoding rejects)
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a.lo), "s"(kk), "v"(c.lo));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a.hi), "s"(kk), "v"(c.hi));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// d = a * b + c, all per-lane
__device__ __forceinline__ f32x4 pkfma4v(f32x4 a, f32x4 b, f32x4 c) {
    f32x2 lo, hi;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a.lo), "v"(b.lo), "v"(c.lo));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a.hi), "v"(b.hi), "v"(c.hi));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// the same, in place, on ext-vector pixels (wino4p.hpp)
__device__ __forceinline__ void bt6v(f32x4 (&c)[6]) {
    auto fm = [](float k, f32x4 a, f32x4 b) { return pkfma4(k, a, b); };
    const f32x4 e12 = fm(-B2, c[2], c[4]), o12 = fm(-B2, c[1], c[3]);
    const f32x4 e34 = fm(-A2, c[2], c[4]), o34 = fm(-A2, c[1], c[3]);
    const f32x4 t0 = fm(P2, c[0], fm(-S2, c[2], c[4]));
    const f32x4 t5 = fm(P2, c[1], fm(-S2, c[3], c[5]));
    c[0] = t0;
    c[1] = fm(PA, o12, e12);
    c[2] = fm(-PA, o12, e12);
    c[3] = fm(PB, o34, e34);
    c[4] = fm(-PB, o34, e34);
    c[5] = t5;
}
// out of place
__device__ __forceinline__ void bt6v2(const f32x4 (&c)[6], f32x4 &t0, f32x4 &t1, f32x4 &t2, f32x4 &t3, f32x4 &t4, f32x4 &t5) {
    auto fm = [](float k, f32x4 a, f32x4 b) { return pkfma4(k, a, b); };
    const f32x4 e12 = fm(-B2, c[2], c[4]), o12 = fm(-B2, c[1], c[3]);
    const f32x4 e34 = fm(-A2, c[2], c[4]), o34 = fm(-A2, c[1], c[3]);
    t0 = fm(P2, c[0], fm(-S2, c[2], c[4]));
    t5 = fm(P2, c[1], fm(-S2, c[3], c[5]));
    t1 = fm(PA, o12, e12);
    t2 = fm(-PA, o12, e12);
    t3 = fm(PB, o34, e34);
    t4 = fm(-PB, o34, e34);
}
// ... and the half column (see bt3 below) on ext vectors
__device__ __forceinline__ void bt3v(const f32x4 (&c)[4], const f32x4 (&z)[3], f32x4 &t0, f32x4 &t1, f32x4 &t2, float K2, float KP) {
    auto fm = [](float k, f32x4 a, f32x4 b) { return pkfma4(k, a, b); };
    const f32x4 e = fm(-K2, c[1], c[3]), o = fm(-K2, c[0], c[2]);
    t0 = fm(P2, z[0], fm(-S2, z[1], z[2]));
    t1 = fm(KP, o, e);
    t2 = fm(-KP, o, e);
}
__device__ __forceinline__ void bt6s(const float (&c)[6], float (&t)[6]) {      // the same on scalars
    const float e12 = fmaf(-B2, c[2], c[4]), o12 = fmaf(-B2, c[1], c[3]);
    const float e34 = fmaf(-A2, c[2], c[4]), o34 = fmaf(-A2, c[1], c[3]);
    t[0] = fmaf(P2, c[0], fmaf(-S2, c[2], c[4]));
    t[1] = fmaf(PA, o12, e12);
    t[2] = fmaf(-PA, o12, e12);
    t[3] = fmaf(PB, o34, e34);
    t[4] = fmaf(-PB, o34, e34);
    t[5] = fmaf(P2, c[1], fmaf(-S2, c[3], c[5]));
}
#undef ADYOLO_W4_BT
// half of it for a half column: xi = 0, 1, 2 (hh = 0) or xi = 5, 3, 4 (hh = 1), in that order.  The pair terms use rows 1..4 in
// both cases (K2 = B2, KP = PA or A2, PB: wave-uniform scalars), the single term rows z = (0, 2, 4) or (1, 3, 5): the caller
// reads z through wave-uniform addresses, so there is no select and no branch here (two of the seven rows are read twice)
__device__ __forceinline__ void bt3(const float4 (&c)[4], const float4 (&z)[3], float4 (&t)[3], float K2, float KP) {
#define ADYOLO_W4_BT3(F)                                                                  \
    {                                                                                     \
        const float e = fmaf(-K2, c[1].F, c[3].F), o = fmaf(-K2, c[0].F, c[2].F);          \
        t[0].F = fmaf(P2, z[0].F, fmaf(-S2, z[1].F, z[2].F));                             \
        t[1].F = fmaf(KP, o, e);                                                          \
        t[2].F = fmaf(-KP, o, e);                                                         \
    }
    ADYOLO_W4_BT3(x) ADYOLO_W4_BT3(y) ADYOLO_W4_BT3(z) ADYOLO_W4_BT3(w)
#undef ADYOLO_W4_BT3
}

// output transform A^T along one direction: y[p] = sum_k AT[p][k] m[k]
__device__ __forceinline__ void at4(float m0, float m1, float m2, float m3, float m4, float m5, float &y0, float &y1,
                                    float &y2, float &y3) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    y0 = m0 + s12 + s34;
    y1 = fmaf(PA, d12, PB * d34);
    y2 = fmaf(A2, s12, B2 * s34);
    y3 = fmaf(A3, d12, fmaf(B3, d34, m5));
}

#ifndef W4_BRING
#define W4_BRING 9       // B fragments in flight per wave (9: 2304 matrix cycles ahead, 18: 4608)
#endif
#ifndef W4P_EXP
#define W4P_EXP 0
#endif
#ifndef W4_WHATIF
#define W4_WHATIF 0       // timing-only builds (results invalid): bit 0 no staging in the loop, 1 no B refills, 2 no A reads / transforms,
                          // 3 no epilogue (one store per lane), 4 no MFMAs, 5 no staging loads, 6 no staging transforms, 7 no staging writes,
                          // 8 no leftover-row round, 9 epilogue without its register->LDS half, 10 epilogue without its stores
#endif

// Four chained MFMAs on an accumulator tile that lives in ARCHITECTURAL registers.  A wave owns 18 tiles = 288 registers, the
// accumulator half of the file holds 256: hipcc keeps the other two tiles in VGPRs but issues every MFMA in the AccVGPR form,
// copying the tile in and out around each use (32 v_accvgpr moves per use, each read waiting for the MFMA to drain).  The "+v"
// constraint pins the VGPR form.  s_nop 1: a just-written VGPR operand needs two wait states before an MFMA reads it, and hipcc
// pads nothing inside an asm statement (cdna_hip_programming.md 5.7 item 2); the chain on one accumulator needs none, and
// the tile's next reader is the epilogue, thousands of cycles later.
__device__ __forceinline__ void mfma32x4_vgpr(f32x16 &c, const float4 &a, const float4 &b) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %1, %5, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %2, %6, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %3, %7, %0\n\t"
        "v_mfma_f32_32x32x2_f32 %0, %4, %8, %0"
        : "+v"(c)
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
}

}  // namespace w4
}  // namespace adyolo
