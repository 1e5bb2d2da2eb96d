// EXPERIMENT (round 2, moved out of the build in round 3): the wave-specialised persistent form of the stage-1 Winograd forward.
// Plain convolution 1.05 -> 0.90 ms, but with the training epilogues it LOSES (forward + statistics 1.08 -> 1.21 ms, fused
// data-gradient 1.6 -> 2.0 ms inside the step): see DESIGN.md, "Tried and rejected".  It was an opt-in of libadyolo_hip.so
// (ADYOLO_WINO_WS=1) with its own test in round 2; to rebuild it, add it back to build.py's SOURCES and restore the dispatch
// in adyolo_wino_fwd (git history: 033625c).
// K2w, stage-1 shape (Cin = 32 -> Cout = 32), wave-specialised persistent form of wino_fwd_kernel<1, true> (wino.hip).
// What-if builds put 22 % of that launch on the exposed first-patch load and 12 % on the epilogue
// (profiles/r02_whatif_wino_fwd_prologue.txt): one 32-channel chunk is only 16 steps of matrix work per patch, too short
// for two or three co-resident workgroups to cover each other's memory phases.  Here ONE 8-wave workgroup per CU walks its
// patches as a three-stage pipeline separated by one workgroup barrier per patch:
//   waves 4-7 (producers): patch i+3 is requested from HBM, patch i+1 (requested two iterations ago) goes from registers
//                          to LDS, and the output transform / stores of patch i-1 are done from the exchange buffer;
//   waves 0-3 (consumers): the 64 MFMAs of patch i (transform row xi = wave, as in wino_fwd_kernel), partial sums to the
//                          exchange buffer.
// The consumers keep their whole U slice in registers (64 per lane), so their loop has no global memory operation; the
// producers' long-latency loads have a vmcnt counter of their own.
#include "common.hpp"

namespace adyolo {

constexpr int SKC = 32;                         // channels (in and out)
constexpr int SAS = 36;                         // floats per staged pixel (144 B)
constexpr int SHALF = 10;                       // slots per (row, parity) half row
constexpr int SPATCH = 10 * 2 * SHALF * SAS;    // floats per staged patch (28.8 KB)
constexpr int SCBP = SKC + 8;                   // exchange row
constexpr int SPBUF = 8 * 32 * SCBP;            // [wave][b][tile][SCBP]  (41 KB)
constexpr int SAPT = 6;

__device__ __forceinline__ float4 s_fma(float4 a, float s, float4 b) {
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 s_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 s_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// makes a value opaque at this point of the instruction stream (a use of it cannot be scheduled above)
__device__ __forceinline__ void pin4(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// LDS writes of this wave done, then the workgroup barrier -- without the vmcnt(0) a __syncthreads() fence may add for
// outstanding global stores (it would drain the producers' prefetches, which share the in-order counter)
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// epilogue operands: the same set as wino_fwd_kernel (wino.hip)
struct WsEpi {
    const float *bias, *addend, *addend_mask, *in_scale, *in_shift, *stat_aux, *stat_mean, *stat_invstd, *stat_mask;
    float *stats;
    int relu, mask_bits;
};
constexpr int SRED = 2 * 32 * SKC;              // floats of one statistics staging buffer [2][32 tiles][32 channels]

__global__ __launch_bounds__(512, 1) void wino_ws_kernel(const float *__restrict__ x, const float *__restrict__ u,
                                                         float *__restrict__ y, WsEpi ep, int H, int W, int tilesW, int tilesH,
                                                         int nsp) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SPATCH + 2 * SPBUF + 2 * SRED];
    float *const Pbuf = lds;                    // two patch buffers
    float *const Xbuf = lds + 2 * SPATCH;       // two exchange buffers
    float *const Rbuf = Xbuf + 2 * SPBUF;       // two statistics staging buffers
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);       // 0: consumer waves, 1: producer waves
    const int tid = threadIdx.x & 255, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int first = blockIdx.x, stride = gridDim.x;
    const int count = first < nsp ? (nsp - first + stride - 1) / stride : 0;      // patches of this workgroup

    if (role == 1) {
        // ------------------------------------------------------------------------------------------ producers
        const int sq = tid & 7, spix0 = tid >> 3;
        // patch cursors: (sample, tile row, tile column) of patch `first + k * stride`, advanced by one stride at a time with
        // wave-uniform adds (no divisions in the loop); past the workgroup's last patch a cursor stays on it (its loads are
        // issued but never used)
        int dn, dth, dtw;
        {
            int s_ = stride;
            dtw = s_ % tilesW;
            s_ /= tilesW;
            dth = s_ % tilesH;
            dn = s_ / tilesH;
        }
        struct Cursor {
            int n, th, tw, k;
        };
        auto make_cursor = [&](int k) {
            Cursor c;
            int p = first + min(k, count - 1) * stride;
            c.tw = p % tilesW;
            p /= tilesW;
            c.th = p % tilesH;
            c.n = p / tilesH;
            c.k = k;
            return c;
        };
        auto advance = [&](Cursor &c) {
            c.k += 1;
            if (c.k >= count) return;
            c.tw += dtw;
            if (c.tw >= tilesW) {
                c.tw -= tilesW;
                c.th += 1;
            }
            c.th += dth;
            if (c.th >= tilesH) {
                c.th -= tilesH;
                c.n += 1;
            }
            c.n += dn;
        };
        auto load_patch = [&](float4 (&dst)[SAPT], const Cursor &c) {
            const int n = c.n, ty0 = c.th * 8, tx0 = c.tw * 16;
            const char *xs = reinterpret_cast<const char *>(x + (size_t)n * H * W * SKC);
#pragma unroll
            for (int ii = 0; ii < SAPT; ++ii) {
                const int pix = spix0 + ii * 32;
                const int hy = (pix * 3641) >> 16, hx = pix - hy * 18;
                const int cy = min(max(ty0 + hy - 1, 0), H - 1), cx = min(max(tx0 + hx - 1, 0), W - 1);
                dst[ii] = *reinterpret_cast<const float4 *>(xs + (unsigned)((cy * W + cx) * SKC + sq * 4) * 4u);
            }
        };
        // producer BatchNorm affine of the input (1 | 0 if none), applied while the patch is written to LDS; the zero padding
        // stays zero (bit mask)
        float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ep.in_scale) {
            isc = *reinterpret_cast<const float4 *>(ep.in_scale + sq * 4);
            ish = *reinterpret_cast<const float4 *>(ep.in_shift + sq * 4);
        }
        auto store_patch = [&](const float4 (&src)[SAPT], const Cursor &c, float *buf) {
            const int ty0 = c.th * 8, tx0 = c.tw * 16;
#pragma unroll
            for (int ii = 0; ii < SAPT; ++ii) {
                const int pix = spix0 + ii * 32;
                const bool real = pix < 180;
                const int hy0 = (pix * 3641) >> 16, hx0 = pix - hy0 * 18;
                const int gy = ty0 + hy0 - 1, gx = tx0 + hx0 - 1;
                const unsigned m = (real && gy >= 0 && gy < H && gx >= 0 && gx < W) ? 0xffffffffu : 0u;
                const int pw = real ? pix : spix0;
                const int hy = (pw * 3641) >> 16, hx = pw - hy * 18;
                const float4 v = src[ii];
                float4 o;
                o.x = __uint_as_float(__float_as_uint(fmaf(v.x, isc.x, ish.x)) & m);
                o.y = __uint_as_float(__float_as_uint(fmaf(v.y, isc.y, ish.y)) & m);
                o.z = __uint_as_float(__float_as_uint(fmaf(v.z, isc.z, ish.z)) & m);
                o.w = __uint_as_float(__float_as_uint(fmaf(v.w, isc.w, ish.w)) & m);
                *reinterpret_cast<float4 *>(&buf[((hy * 2 + (hx & 1)) * SHALF + (hx >> 1)) * SAS + (real ? sq * 4 : 32)]) = o;
            }
        };
        const int c4 = tid & 7, m = tid >> 3;                      // epilogue role: tile m, channels 4 c4 .. 4 c4 + 3
        const int mr = m >> 3, mc = m & 7;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), smean = bv, sinv = bv;
        if (ep.bias) bv = *reinterpret_cast<const float4 *>(ep.bias + c4 * 4);
        if (ep.stat_aux) {
            smean = *reinterpret_cast<const float4 *>(ep.stat_mean + c4 * 4);
            sinv = *reinterpret_cast<const float4 *>(ep.stat_invstd + c4 * 4);
        }
        // Epilogue operands of a patch (addend, its mask, the statistics' second factor and mask) are requested one
        // iteration before the patch's epilogue runs.  Out-of-image pixels read the (clamped) last pixel and are not used.
        // (the masks -- 1/32 of the bytes as bits, cache-friendly -- are read inside the epilogue: prefetching them too costs 64
        //  more registers and spills)
        float4 q_ad[4], q_ax[4];
        auto pixel_offset = [&](const Cursor &c, int a, int b) -> size_t {
            const int gy = min(c.th * 8 + 2 * mr + a, H - 1), gx = min(c.tw * 16 + 2 * mc + b, W - 1);
            return (((size_t)c.n * H + gy) * W + gx) * SKC + c4 * 4;
        };
        auto prefetch_operands = [&](const Cursor &c) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const size_t o = pixel_offset(c, a, b);
                    const int j = a * 2 + b;
                    if (ep.addend) q_ad[j] = *reinterpret_cast<const float4 *>(ep.addend + o);
                    if (ep.stat_aux) q_ax[j] = *reinterpret_cast<const float4 *>(ep.stat_aux + o);
                }
        };
        auto epilogue = [&](const Cursor &c, const float *Xb, float *Rb) {
            const int n = c.n, ty0 = c.th * 8, tx0 = c.tw * 16;
            float4 P[4][2];
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    P[w][b] = *reinterpret_cast<const float4 *>(&Xb[((w * 2 + b) * 32 + m) * SCBP + c4 * 4]);
            float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = ssum;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float4 v = a == 0 ? s_add(s_add(P[0][b], P[1][b]), P[2][b]) : s_sub(s_sub(P[1][b], P[2][b]), P[3][b]);
                    const int gy = ty0 + 2 * mr + a, gx = tx0 + 2 * mc + b;
                    const int j = a * 2 + b;
                    if (gy < H && gx < W) {
                        const size_t o = (((size_t)n * H + gy) * W + gx) * SKC + c4 * 4;
                        v = s_add(v, bv);
                        if (ep.addend) {
                            float4 ad = q_ad[j];
                            if (ep.addend_mask) {
                                bool kx, ky, kz, kw;
                                if (ep.mask_bits & 1) {
                                    mask_bits4(reinterpret_cast<const unsigned long long *>(ep.addend_mask), o >> 2, kx, ky, kz, kw);
                                } else {
                                    const float4 mk = *reinterpret_cast<const float4 *>(ep.addend_mask + o);
                                    kx = mk.x > 0.f; ky = mk.y > 0.f; kz = mk.z > 0.f; kw = mk.w > 0.f;
                                }
                                ad = make_float4(kx ? ad.x : 0.f, ky ? ad.y : 0.f, kz ? ad.z : 0.f, kw ? ad.w : 0.f);
                            }
                            v = s_add(v, ad);
                        }
                        if (ep.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                        *reinterpret_cast<float4 *>(y + o) = v;
                        if (ep.stats) {
                            if (ep.stat_mask) {
                                bool kx, ky, kz, kw;
                                if (ep.mask_bits & 2) {
                                    mask_bits4(reinterpret_cast<const unsigned long long *>(ep.stat_mask), o >> 2, kx, ky, kz, kw);
                                } else {
                                    const float4 mk = *reinterpret_cast<const float4 *>(ep.stat_mask + o);
                                    kx = mk.x > 0.f; ky = mk.y > 0.f; kz = mk.z > 0.f; kw = mk.w > 0.f;
                                }
                                v = make_float4(kx ? v.x : 0.f, ky ? v.y : 0.f, kz ? v.z : 0.f, kw ? v.w : 0.f);
                            }
                            ssum = s_add(ssum, v);
                            if (ep.stat_aux) {
                                const float4 ax = q_ax[j];
                                ssq.x += v.x * (ax.x - smean.x) * sinv.x;
                                ssq.y += v.y * (ax.y - smean.y) * sinv.y;
                                ssq.z += v.z * (ax.z - smean.z) * sinv.z;
                                ssq.w += v.w * (ax.w - smean.w) * sinv.w;
                            } else {
                                ssq.x += v.x * v.x;
                                ssq.y += v.y * v.y;
                                ssq.z += v.z * v.z;
                                ssq.w += v.w * v.w;
                            }
                        }
                    }
                }
            if (ep.stats) {                                // staged per tile; summed over the 32 tiles one iteration later
                *reinterpret_cast<float4 *>(&Rb[(0 * 32 + m) * SKC + c4 * 4]) = ssum;
                *reinterpret_cast<float4 *>(&Rb[(1 * 32 + m) * SKC + c4 * 4]) = ssq;
            }
        };
        // per-patch, per-channel sums of the stored output, layout [2][patches][32] (as wino_fwd_kernel): the staged tile
        // sums of patch `sp` are added in tile order (the same order as there: bit-identical statistics)
        auto flush_stats = [&](int sp, const float *Rb) {
            if (ep.stats && tid < 2 * SKC) {
                const int c = tid % SKC, which = tid / SKC;
                float sacc = 0.f;
#pragma unroll 8
                for (int gI = 0; gI < 32; ++gI) sacc += Rb[(which * 32 + gI) * SKC + c];
                ep.stats[(size_t)which * nsp * SKC + (size_t)sp * SKC + c] = sacc;
            }
        };
        float4 ra[SAPT], rb[SAPT];
        Cursor cl = make_cursor(0), cs = make_cursor(1), ce = make_cursor(0), cq = make_cursor(0);   // load / store / epilogue / operand cursors
        if (count > 0) {
            load_patch(ra, cl);                 // patch 0
            advance(cl);
            load_patch(rb, cl);                 // patch 1
            advance(cl);
            store_patch(ra, ce, Pbuf);          // (ce sits on patch 0)
            load_patch(ra, cl);                 // patch 2
            advance(cl);
        }
        ws_barrier();
        for (int i = 0; i <= count; i += 2) {
            // even iteration i: patch i+1 (in rb) -> P1, request patch i+3 into rb, finish patch i-1 from X1 (its operands were
            // requested in iteration i-1), request the operands of patch i, file the statistics of patch i-2
            if (i + 1 < count) store_patch(rb, cs, Pbuf + SPATCH);
            advance(cs);
            load_patch(rb, cl);
            advance(cl);
            if (i >= 1) {
                epilogue(ce, Xbuf + SPBUF, Rbuf + SRED);
                advance(ce);
            }
            if (i < count) prefetch_operands(cq);
            advance(cq);
            if (i >= 2) flush_stats(first + (i - 2) * stride, Rbuf);
            ws_barrier();
            if (i + 1 > count) {
                if (i >= 1) flush_stats(first + (i - 1) * stride, Rbuf + SRED);
                break;
            }
            // odd iteration i+1: patch i+2 (in ra) -> P0, request patch i+4 into ra, finish patch i from X0
            if (i + 2 < count) store_patch(ra, cs, Pbuf);
            advance(cs);
            load_patch(ra, cl);
            advance(cl);
            epilogue(ce, Xbuf, Rbuf);
            advance(ce);
            if (i + 1 < count) prefetch_operands(cq);
            advance(cq);
            if (i >= 1) flush_stats(first + (i - 1) * stride, Rbuf + SRED);
            ws_barrier();
            if (i + 2 > count) flush_stats(first + i * stride, Rbuf);
        }
    } else {
        // ------------------------------------------------------------------------------------------ consumers
        __builtin_amdgcn_s_setprio(3);          // the matrix wave of a SIMD is issued ahead of its producer wave
        const int ia = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
        const int ib = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
        const float sg = wave == 1 ? 1.f : -1.f;
        const int tr = li >> 3, tc = li & 7;
        const int offa = (((2 * tr + ia) * 2) * SHALF + tc) * SAS + lh * 4;
        const int offb = (((2 * tr + ib) * 2) * SHALF + tc) * SAS + lh * 4;
        constexpr int J1 = SHALF * SAS, J2 = SAS;
        constexpr int NKG = SKC / 8;                                          // four 8-channel groups
        const size_t ustride_pos = (size_t)NKG * 256;                         // floats per transform position (Cout / 32 = 1)
        const char *ubase = reinterpret_cast<const char *>(u + (size_t)(wave * 4) * NKG * 256);
        const unsigned ulane = lane * 16u;
        float4 r0, r1, r2, r3, da0, da1, db0, db1;
        auto issue_cols = [&](const float *As, int g, int half) {
            const float *pa = As + offa + g * 8 + half * J2, *pb = As + offb + g * 8 + half * J2;
            da0 = *reinterpret_cast<const float4 *>(pa);
            da1 = *reinterpret_cast<const float4 *>(pa + J1);
            db0 = *reinterpret_cast<const float4 *>(pb);
            db1 = *reinterpret_cast<const float4 *>(pb + J1);
        };
        auto combine_lo = [&]() { r0 = s_fma(db0, sg, da0); r1 = s_fma(db1, sg, da1); };
        auto combine_hi = [&]() { r2 = s_fma(db0, sg, da0); r3 = s_fma(db1, sg, da1); };
        // this wave's whole slice of U (4 positions x 32 x 32 = 64 registers per lane) is the same for every patch of the
        // launch: loaded once, the loop has no global loads at all
        float4 bq[NKG][4];
#pragma unroll
        for (int p = 0; p < NKG; ++p)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                bq[p][v] = *reinterpret_cast<const float4 *>(ubase + ((unsigned)((v * ustride_pos + (size_t)p * 256) * 4) + ulane));
        ws_barrier();
        for (int i = 0; i <= count; ++i) {
            if (i < count) {
                const float *As = Pbuf + (i & 1) * SPATCH;
                // One wave per SIMD has nobody to hide behind, so everything is software-pipelined by hand (measured with
                // s_memtime before: 5330 cycles for the 64 MFMAs of a patch + 900 for the exchange = 65 % matrix-pipe use):
                //  * the eight LDS reads of group g+1 are issued at the top of group g, combined into r0..r3 behind its first
                //    eight MFMAs, and the four A fragments of group g+1 are formed behind the next four;
                //  * the first MFMA of every accumulator takes a zero C operand instead of 64 register clears per patch;
                //  * the nu-sums of the exchange are formed and written as soon as their accumulators are final, under the
                //    MFMAs of the last group (only acc[3]'s share is exposed).
                // (The combines are pure arithmetic: without the opaque pins the compiler hoists them, and the wait for their
                //  LDS operands, to just behind a group's first MFMA.)
                f32x16 acc[4];
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                float *Xb = Xbuf + (i & 1) * SPBUF;
                float *x0 = Xb + ((wave * 2 + 0) * 32) * SCBP + li, *x1 = Xb + ((wave * 2 + 1) * 32) * SCBP + li;
                issue_cols(As, 0, 0);
                combine_lo();
                issue_cols(As, 0, 1);
                combine_hi();
                float4 a[4], an[4];
                a[0] = s_sub(r0, r2);
                a[1] = s_add(r1, r2);
                a[2] = s_sub(r2, r1);
                a[3] = s_sub(r1, r3);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NKG; ++g) {
                    const bool nextg = g + 1 < NKG;
                    float4 ea0, ea1, eb0, eb1;
                    const float *pa0 = As + offa + (g + 1) * 8, *pb0 = As + offb + (g + 1) * 8;
                    // accumulators interleaved (consecutive MFMAs are independent); the eight LDS reads of group g+1 are
                    // issued two at a time behind the first four MFMAs (a clump of eight holds the wave's issue slot long
                    // enough to leave the matrix pipe idle), their combines behind the 8th and 12th
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float av = q == 0 ? a[v].x : (q == 1 ? a[v].y : (q == 2 ? a[v].z : a[v].w));
                            const float bv_ = q == 0 ? bq[g][v].x : (q == 1 ? bq[g][v].y : (q == 2 ? bq[g][v].z : bq[g][v].w));
                            acc[v] = mfma32(av, bv_, (g == 0 && q == 0) ? zero : acc[v]);
                            if (nextg && q == 0) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (v == 0) {
                                    da0 = *reinterpret_cast<const float4 *>(pa0);
                                    da1 = *reinterpret_cast<const float4 *>(pa0 + J1);
                                } else if (v == 1) {
                                    db0 = *reinterpret_cast<const float4 *>(pb0);
                                    db1 = *reinterpret_cast<const float4 *>(pb0 + J1);
                                } else if (v == 2) {
                                    ea0 = *reinterpret_cast<const float4 *>(pa0 + J2);
                                    ea1 = *reinterpret_cast<const float4 *>(pa0 + J2 + J1);
                                } else {
                                    eb0 = *reinterpret_cast<const float4 *>(pb0 + J2);
                                    eb1 = *reinterpret_cast<const float4 *>(pb0 + J2 + J1);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        if (nextg && q == 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            pin4(da0); pin4(da1); pin4(db0); pin4(db1);
                            combine_lo();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (nextg && q == 2) {
                            __builtin_amdgcn_sched_barrier(0);
                            pin4(ea0); pin4(ea1); pin4(eb0); pin4(eb1);
                            r2 = s_fma(eb0, sg, ea0);
                            r3 = s_fma(eb1, sg, ea1);
                            an[0] = s_sub(r0, r2);
                            an[1] = s_add(r1, r2);
                            an[2] = s_sub(r2, r1);
                            an[3] = s_sub(r1, r3);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (nextg) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) a[v] = an[v];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int mm = mfma_row(r, lane) * SCBP;
                            x0[mm] = acc[0][r] + acc[1][r] + acc[2][r];
                            x1[mm] = acc[1][r] - acc[2][r] - acc[3][r];
                        }
                    }
                }
            }
            ws_barrier();
        }
    }
}

}  // namespace adyolo

using namespace adyolo;

static int launch_ws(const float *x, const float *u, float *y, const WsEpi &ep, int N, int H, int W, int blocks, void *stream) {
    const int tilesW = cdiv(W, 16), tilesH = cdiv(H, 8);
    const int nsp = N * tilesH * tilesW;
    if (blocks <= 0) blocks = 256;
    if (blocks > nsp) blocks = nsp;
    hipLaunchKernelGGL(wino_ws_kernel, dim3((unsigned)blocks), dim3(512), 0, as_stream(stream), x, u, y, ep, H, W, tilesW, tilesH,
                       nsp);
    return check_launch("wino_fwd_ws");
}

// The stage-1 shape of adyolo_wino_fwd (Cin = Cout = 32; same arguments and results): called from there.
int adyolo_wino_fwd_ws_full(const float *x, const float *u, const float *bias, const float *addend, const float *addend_mask,
                            const float *in_scale, const float *in_shift, float *y, float *stats, const float *stat_aux,
                            const float *stat_mean, const float *stat_invstd, const float *stat_mask, int N, int H, int W,
                            int relu, int mask_bits, void *stream) {
    WsEpi ep{bias, addend, addend_mask, in_scale, in_shift, stat_aux, stat_mean, stat_invstd, stat_mask, stats, relu, mask_bits};
    return launch_ws(x, u, y, ep, N, H, W, 256, stream);
}

// experimental entry point of tools/ws_bench.py (plain forward with optional bias / ReLU; Cin = Cout = 32)
extern "C" int adyolo_wino_fwd_ws(const float *x, const float *u, const float *bias, float *y, int N, int H, int W, int relu,
                                  int blocks, void *stream) {
    ADYOLO_REQUIRE(x && u && y && N > 0 && H > 0 && W > 0, ADYOLO_EINVAL, "wino_fwd_ws: bad arguments");
    WsEpi ep{bias, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, relu, 0};
    return launch_ws(x, u, y, ep, N, H, W, blocks, stream);
}
