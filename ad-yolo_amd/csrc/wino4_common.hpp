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
typedef unsigned int u32x2_t __attribute__((__vector_size__(8)));

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
// p[j] = p[j] * sc + sh on the lanes whose byte offset vo[j] lies inside the buffer (vo[j] < nrec, unsigned); the other lanes keep
// p[j], which is the 0 an out-of-range buffer load returned.  The comparison writes EXEC (v_cmpx) and the two packed FMAs run under
// it: one vector instruction per pixel on top of the FMAs instead of a compare and four selects of the shift (round 6).  Every
// lane of the wave must be active at the call (the staging code is).
__device__ __forceinline__ void aff6_inrange(f32x4 (&p)[6], f32x4 sc, f32x4 sh, const int (&vo)[6], unsigned nrec) {
    f32x2 l0 = p[0].lo, h0 = p[0].hi, l1 = p[1].lo, h1 = p[1].hi, l2 = p[2].lo, h2 = p[2].hi;
    f32x2 l3 = p[3].lo, h3 = p[3].hi, l4 = p[4].lo, h4 = p[4].hi, l5 = p[5].lo, h5 = p[5].hi;
    unsigned long long sv;
#define ADYOLO_AFF1(L, H, O)                                                                                                      \
    "v_cmpx_gt_u32_e32 vcc, %[n], %[" O "]\n\tv_pk_fma_f32 %[" L "], %[" L "], %[scl], %[shl]\n\t"                                   \
    "v_pk_fma_f32 %[" H "], %[" H "], %[sch], %[shh]\n\ts_mov_b64 exec, %[sv]\n\t"
    asm volatile("s_mov_b64 %[sv], exec\n\t" ADYOLO_AFF1("l0", "h0", "o0") ADYOLO_AFF1("l1", "h1", "o1") ADYOLO_AFF1("l2", "h2", "o2")
                     ADYOLO_AFF1("l3", "h3", "o3") ADYOLO_AFF1("l4", "h4", "o4") ADYOLO_AFF1("l5", "h5", "o5")
                 : [l0] "+v"(l0), [h0] "+v"(h0), [l1] "+v"(l1), [h1] "+v"(h1), [l2] "+v"(l2), [h2] "+v"(h2), [l3] "+v"(l3),
                   [h3] "+v"(h3), [l4] "+v"(l4), [h4] "+v"(h4), [l5] "+v"(l5), [h5] "+v"(h5), [sv] "=&s"(sv)
                 : [scl] "v"(sc.lo), [sch] "v"(sc.hi), [shl] "v"(sh.lo), [shh] "v"(sh.hi), [n] "s"(nrec), [o0] "v"(vo[0]),
                   [o1] "v"(vo[1]), [o2] "v"(vo[2]), [o3] "v"(vo[3]), [o4] "v"(vo[4]), [o5] "v"(vo[5])
                 : "vcc");
#undef ADYOLO_AFF1
    p[0] = __builtin_shufflevector(l0, h0, 0, 1, 2, 3);
    p[1] = __builtin_shufflevector(l1, h1, 0, 1, 2, 3);
    p[2] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3);
    p[3] = __builtin_shufflevector(l3, h3, 0, 1, 2, 3);
    p[4] = __builtin_shufflevector(l4, h4, 0, 1, 2, 3);
    p[5] = __builtin_shufflevector(l5, h5, 0, 1, 2, 3);
}
// sum += v, sq = v * w + sq on the lanes with off >= 0 (an output pixel inside the image; out-of-image pixels carry the sign bit in
// their store offset), under an EXEC mask made by the comparison: no selects of v (round 6).  Every lane active at the call.
__device__ __forceinline__ void stat_acc_inimage(f32x4 &sum, f32x4 &sq, f32x4 v, f32x4 w, int off) {
    f32x2 sl = sum.lo, sh = sum.hi, ql = sq.lo, qh = sq.hi;
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\tv_cmpx_le_i32_e32 vcc, 0, %[o]\n\tv_pk_add_f32 %[sl], %[sl], %[vl]\n\t"
                 "v_pk_add_f32 %[sh], %[sh], %[vh]\n\tv_pk_fma_f32 %[ql], %[vl], %[wl], %[ql]\n\t"
                 "v_pk_fma_f32 %[qh], %[vh], %[wh], %[qh]\n\ts_mov_b64 exec, %[sv]"
                 : [sl] "+v"(sl), [sh] "+v"(sh), [ql] "+v"(ql), [qh] "+v"(qh), [sv] "=&s"(sv)
                 : [vl] "v"(v.lo), [vh] "v"(v.hi), [wl] "v"(w.lo), [wh] "v"(w.hi), [o] "v"(off)
                 : "vcc");
    sum = __builtin_shufflevector(sl, sh, 0, 1, 2, 3);
    sq = __builtin_shufflevector(ql, qh, 0, 1, 2, 3);
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
