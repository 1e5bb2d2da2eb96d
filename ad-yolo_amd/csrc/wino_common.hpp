// Shared by the Winograd F(2x2) forward kernel (wino.hip) and its experimental bf16x3 variant (tools/experiments/bf16x3):
// patch geometry constants, the per-template configuration and the epilogue (output transform, bias / masked addend / ReLU,
// per-patch BatchNorm sums).
#pragma once
#include <type_traits>
#include "common.hpp"

namespace adyolo {

constexpr int WKC = 32;                    // input channels per chunk
constexpr int WAS = 36;                    // floats per staged pixel (144 B)
constexpr int WHALF = 10;                  // slots per (row, parity) half row (9 used)
constexpr int WPATCH = 10 * 2 * WHALF * WAS;   // floats per staged patch (28.8 KB)

// ONE = the whole Cin fits one chunk (Cin == 32: stage 1): a single patch buffer, 44 KB of LDS and <= 168 VGPRs, so THREE
// workgroups per CU cover each other's prologue / epilogue (one 32-channel chunk is only 16 steps of matrix work)
template <int NT, bool ONE>
struct WinoCfg {
    static constexpr int CB = 32 * NT;
    static constexpr int CBP = CB + 8;                       // epilogue exchange row (conflict-free b32 writes)
    static constexpr int PBUF = 8 * 32 * CBP;                // [wave][b][tile][CBP]
    static constexpr int NBUF = ONE ? 1 : 2;
    static constexpr int LDS_FLOATS = (NBUF * WPATCH > PBUF) ? NBUF * WPATCH : PBUF;
    static constexpr int WG_PER_CU = (ONE && NT == 1) ? 3 : 2;
};
constexpr int WMAXC = 512;                 // largest Cin of the Winograd forward kernel (affine table in LDS)

__device__ __forceinline__ float4 f4_fma(float4 a, float s, float4 b) {      // b + s * a
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// Epilogue of a forward workgroup: acc[v][nt] = M[pos = 4 wave + v][32 tiles][32 nt + cout] in MFMA accumulator layout.
// The nu-sum of A^T . A is done in registers, the xi-sum through LDS (which the caller has finished reading), then
// bias / masked addend / ReLU / per-patch BatchNorm sums as in conv.hip, stored as float4 along channels.
template <int NT, bool ONE>
__device__ __forceinline__ void wino_epilogue(f32x16 (&acc)[4][NT], float *lds, int tid, int lane, int wave, int li,
                                              const float *__restrict__ bias, const float *__restrict__ addend,
                                              const float *__restrict__ addend_mask, float *__restrict__ y,
                                              float *__restrict__ stats, const float *__restrict__ stat_aux,
                                              const float *__restrict__ stat_mean, const float *__restrict__ stat_invstd,
                                              const float *__restrict__ stat_mask, int n, int H, int W, int Cout, int co0,
                                              int ty0, int tx0, int nsp, int sp, int relu, int mask_bits) {
    using Cfg = WinoCfg<NT, ONE>;
    constexpr int CB = Cfg::CB, CBP = Cfg::CBP;
    // ---- output transform.  nu-sum in registers: P[b] = sum_nu A^T[b][nu] M[w][nu];  xi-sum through LDS
    float *Pb = lds;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mfma_row(r, lane);
            const float p0 = acc[0][nt][r] + acc[1][nt][r] + acc[2][nt][r];
            const float p1 = acc[1][nt][r] - acc[2][nt][r] - acc[3][nt][r];
            Pb[((wave * 2 + 0) * 32 + m) * CBP + nt * 32 + li] = p0;
            Pb[((wave * 2 + 1) * 32 + m) * CBP + nt * 32 + li] = p1;
        }
    __syncthreads();

    constexpr int C4 = CB / 4;                  // float4 pieces per pixel
    constexpr int MPT = 32 * C4 / 256;          // tiles per thread (1 or 2)
    const int c4 = tid % C4, m0 = tid / C4;
    const int co = co0 + c4 * 4;
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = ssum;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), smean = bv, sinv = bv;
    if (bias) bv = *reinterpret_cast<const float4 *>(bias + co);
    if (stat_aux) {
        smean = *reinterpret_cast<const float4 *>(stat_mean + co);
        sinv = *reinterpret_cast<const float4 *>(stat_invstd + co);
    }
    // One body per combination of (statistics, addend, addend mask), chosen by wave-uniform branches ONCE: inside it there is
    // no branch per pixel and operand (round 4: on this chip a taken / not-taken scalar branch around a few vector instructions
    // costs more than the instructions; the per-pixel form of this epilogue was 12-19 % of a stage-1 / stage-2 launch), the
    // fused operands of the thread's pixels are requested together, out-of-image pixels are stored through the output's buffer
    // descriptor at an out-of-range offset (dropped by the hardware) and counted with weight 0.
    const bool rl = relu != 0;                            // ReLU branch-free: maximum + wave-uniform select (a maximum against a
                                                          // -inf floor, round 4, turned NaN results into -inf)
    const size_t sbase = (size_t)n * H * W * Cout;        // floats: the sample's base in y / addend / masks / stat_aux
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y + sbase, 0, H * W * Cout * 4, 0x00020000);
    typedef unsigned int u32x4_t_ __attribute__((__vector_size__(16)));
    auto body = [&](auto ST_, auto AD_, auto MK_) {
        constexpr bool ST = decltype(ST_)::value, AD = decltype(AD_)::value, MK = decltype(MK_)::value;
        const bool has_smk = ST && stat_mask != nullptr, has_aux = ST && stat_aux != nullptr;
        auto keep4 = [&](const float *mptr, bool bits, int o_) {
            bool kx, ky, kz, kw;
            if (bits) {
                mask_bits4(reinterpret_cast<const unsigned long long *>(mptr), (sbase + (size_t)(o_ >> 2)) >> 2, kx, ky, kz, kw);
            } else {
                const float4 mk = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(mptr + sbase) + o_);
                kx = mk.x > 0.f; ky = mk.y > 0.f; kz = mk.z > 0.f; kw = mk.w > 0.f;
            }
            return (kx ? 1u : 0u) | (ky ? 2u : 0u) | (kz ? 4u : 0u) | (kw ? 8u : 0u);
        };
#pragma unroll
        for (int it = 0; it < MPT; ++it) {
            const int m = m0 + it * (256 / C4);
            const int mr = m >> 3, mc = m & 7;
            int off[4];                                   // byte offset of the pixel's channel quad inside the sample
            bool ok[4];
            float4 ad[4], ax[4];
            unsigned amk[4], smk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gy = ty0 + 2 * mr + (q >> 1), gx = tx0 + 2 * mc + (q & 1);
                ok[q] = gy < H && gx < W;
                off[q] = (__mul24(__mul24(min(gy, H - 1), W) + min(gx, W - 1), Cout) + co) * 4;      // (full-rate 24-bit multiplies)
            }
            if (AD) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    ad[q] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(addend + sbase) + off[q]);
                if (MK) {
                    const bool bits = (mask_bits & 1) != 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) amk[q] = keep4(addend_mask, bits, off[q]);
                }
            }
            if (has_smk) {
                const bool bits = (mask_bits & 2) != 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) smk[q] = keep4(stat_mask, bits, off[q]);
            }
            if (has_aux) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    ax[q] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(stat_aux + sbase) + off[q]);
            }
            float4 P[4][2];
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    P[w][b] = *reinterpret_cast<const float4 *>(&Pb[((w * 2 + b) * 32 + m) * CBP + c4 * 4]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int a = q >> 1, b = q & 1;
                float4 v = a == 0 ? f4_add(f4_add(P[0][b], P[1][b]), P[2][b]) : f4_sub(f4_sub(P[1][b], P[2][b]), P[3][b]);
                v = f4_add(v, bv);
                if (AD) {
                    float4 a_ = ad[q];
                    if (MK) {
                        const unsigned k = amk[q];
                        a_ = make_float4((k & 1u) ? a_.x : 0.f, (k & 2u) ? a_.y : 0.f, (k & 4u) ? a_.z : 0.f, (k & 8u) ? a_.w : 0.f);
                    }
                    v = f4_add(v, a_);
                }
                v = make_float4(rl ? fmaxf(v.x, 0.f) : v.x, rl ? fmaxf(v.y, 0.f) : v.y, rl ? fmaxf(v.z, 0.f) : v.z, rl ? fmaxf(v.w, 0.f) : v.w);
                // (non-temporal loads/stores here: measured 1 % slower)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t_, v), yrs, ok[q] ? off[q] : (int)0x80000000, 0, 0);
                if (ST) {
                    unsigned k = ok[q] ? 15u : 0u;        // out-of-image pixels and masked-out components count 0
                    if (has_smk) k &= smk[q];
                    v = make_float4((k & 1u) ? v.x : 0.f, (k & 2u) ? v.y : 0.f, (k & 4u) ? v.z : 0.f, (k & 8u) ? v.w : 0.f);
                    ssum = f4_add(ssum, v);
                    float4 w_ = v;
                    if (has_aux)
                        w_ = make_float4((ax[q].x - smean.x) * sinv.x, (ax[q].y - smean.y) * sinv.y, (ax[q].z - smean.z) * sinv.z,
                                         (ax[q].w - smean.w) * sinv.w);
                    ssq.x = fmaf(v.x, w_.x, ssq.x);
                    ssq.y = fmaf(v.y, w_.y, ssq.y);
                    ssq.z = fmaf(v.z, w_.z, ssq.z);
                    ssq.w = fmaf(v.w, w_.w, ssq.w);
                }
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (stats) {
        if (addend) {
            if (addend_mask) body(T_{}, T_{}, T_{}); else body(T_{}, T_{}, F_{});
        } else {
            body(T_{}, F_{}, F_{});
        }
    } else {
        if (addend) {
            if (addend_mask) body(F_{}, T_{}, T_{}); else body(F_{}, T_{}, F_{});
        } else {
            body(F_{}, F_{}, F_{});
        }
    }
    if (stats) {
        // per-patch, per-channel sums of the stored output, layout [2][patches][Cout] (see conv.hip)
        __syncthreads();
        constexpr int G = 256 / C4;             // thread groups sharing a channel piece
        float *red = lds;                       // [2][G][CB]
        *reinterpret_cast<float4 *>(&red[(0 * G + m0) * CB + c4 * 4]) = ssum;
        *reinterpret_cast<float4 *>(&red[(1 * G + m0) * CB + c4 * 4]) = ssq;
        __syncthreads();
        if (tid < CB * 2) {
            const int c = tid % CB, which = tid / CB;
            float s = 0.f;
#pragma unroll 8
            for (int gI = 0; gI < G; ++gI) s += red[(which * G + gI) * CB + c];
            stats[(size_t)which * nsp * Cout + (size_t)sp * Cout + co0 + c] = s;
        }
    }
}

}  // namespace adyolo
