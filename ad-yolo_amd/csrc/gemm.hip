// K7: dense GEMM on the exact-fp32 MFMA, C[m][n] = sum_k opA(m,k) opB(n,k) (+bias[n]).
// Replaces nn.Linear (linearheads.py:95-98; SE FCs resnet.py:96-98), the 1x1 downsample convolution
// (resnet.py:160-162), the GRU input projections (resnet.py:153) and all of their backward GEMMs.
// 128x64 output tile per workgroup (one 32-row strip per wave, two 32x32 accumulators), K tile 32.
// k-major operands are staged as [row][36] and read with one ds_read_b128 per 4 k; m-major
// (transposed) operands are staged as [k][rows+4] and read with four conflict-free ds_read_b32.
#include "common.hpp"

namespace adyolo {

constexpr int GBM = 128, GBN = 64, GBK = 32;
#ifndef GEMM_WHATIF
#define GEMM_WHATIF 0        // timing-only builds (results invalid)
#endif
#ifndef GEMM_ST_NT
#define GEMM_ST_NT 0         // 1: the vectorised epilogue's stores carry the non-temporal hint (A/B switch)
#endif
#ifndef GEMM_PF
#define GEMM_PF 1          // K tiles requested ahead of the one being staged (1 or 2)
#endif

// batched mode: blockIdx.z = outer * inner_count + inner; operand offsets = outer * o? + inner * i? (floats)
struct GemmBatch {
    int inner;                 // 0 = not batched (blockIdx.z is then the split-K slice)
    long oA, iA, oB, iB, oC, iC;
    float alpha;
};

// Implicit convolution operand (no im2col buffer): one GEMM operand is gathered from a channels-last tensor
// src [n][Hs][Ws][Cs] while it is staged.  The GEMM index that runs over pixels ((n, y, x) of an [n][Hm][Wm] grid) is m
// for G == 1 (k-major A: forward and data-gradient) and k for G == 2 (transposed B: weight-gradient); the other index of
// that operand is (kh, kw, c) = ((kh * KW + kw) * Cs + c).  dgrad == 0: source pixel (y*SH - PH + kh, x*SW - PW + kw);
// dgrad == 1 (transposed convolution): ((y + PH - kh) / SH, (x + PW - kw) / SW) where both divisions are exact.
struct ConvGather {
    int Hs, Ws, Cs, Hm, Wm, KH, KW, SH, SW, PH, PW, dgrad, kreal;
};

// Tile shape (round 4): WM x WN waves, each SM x SN accumulator tiles of 32 x 32: BM = 32 WM SM rows x BN = 32 WN SN columns.
//   4 x 1 x 1 x 2 = 128 x 64 on 256 threads (the form of rounds 1-3; skinny and split-K calls),
//   4 x 2 x 2 x 2 = 256 x 128 on 512 threads: per MFMA half the global loads (each costs 12-20 ns of the issue port the fp32
//   MFMA shares) and two thirds of the LDS reads.  Every thread stages 4 + 2 float4 per K tile in both.
template <bool TA, bool TB, int G = 0, int WM = 4, int WN = 1, int SM = 1, int SN = 2>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN > 4 ? 1 : 2)) void gemm_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                      const float *__restrict__ bias, float *__restrict__ C,
                                                      int M, int N, int K, int lda, int ldb, int ldc, int klen,
                                                      size_t slab_stride, int slab_ld, int accumulate, GemmBatch bt,
                                                      ConvGather cg, int fastg) {
    static_assert(G == 0 || (G == 1 && !TA) || (G == 2 && TB), "gathered operand: k-major A or transposed B");
    constexpr int BM = 32 * WM * SM, BN = 32 * WN * SN, NTHR = 64 * WM * WN;
    static_assert(BM * 8 == 4 * NTHR && BN * 8 == 2 * NTHR, "staging: 4 + 2 float4 per thread and K tile");
    constexpr int A_LD = TA ? (BM + 4) : (GBK + 4);
    constexpr int B_LD = TB ? (BN + 4) : (GBK + 4);
    // One LDS array: the two operand tiles of the K loop, and after it (SM == 1) the per-wave transposition buffer of the
    // vectorised epilogue -- [32 rows][32 SN + 4] floats per wave
    constexpr int A_FLOATS = TA ? GBK * (BM + 4) : BM * (GBK + 4), B_FLOATS = TB ? GBK * (BN + 4) : BN * (GBK + 4);
    constexpr bool VEPI = SM == 1 && !TA;               // (the transposed-A forms are the split-K weight gradients: measured 3-7 % slower with it)
    constexpr int T_LD = 32 * SN + 4, T_FLOATS = VEPI ? WM * WN * 32 * T_LD : 0;
    constexpr int SMEM_FLOATS = A_FLOATS + B_FLOATS > T_FLOATS ? A_FLOATS + B_FLOATS : T_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float *As = smem, *Bs = smem + A_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    int z = blockIdx.z;
    if (bt.inner > 0) {
        const long zo = z / bt.inner, zi = z - zo * bt.inner;
        A += zo * bt.oA + zi * bt.iA;
        B += zo * bt.oB + zi * bt.iB;
        C += zo * bt.oC + zi * bt.iC;
        z = 0;
    }
    const int kbeg = z * klen;
    const int kend = min(K, kbeg + klen);

    f32x16 acc[SM][SN];
#pragma unroll
    for (int sm = 0; sm < SM; ++sm)
#pragma unroll
        for (int nt = 0; nt < SN; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[sm][nt][r] = 0.f;

    // gathered A (G == 1): the pixel of each of this thread's four rows is fixed over the K loop
    int gy[4] = {0, 0, 0, 0}, gx[4] = {0, 0, 0, 0}, gb[4] = {-1, -1, -1, -1};
    if (G == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ((tid + i * NTHR) >> 3);
            if (m < M) {
                const int x_ = m % cg.Wm, t_ = m / cg.Wm;
                const int y_ = t_ % cg.Hm, n_ = t_ / cg.Hm;
                gy[i] = cg.dgrad ? y_ + cg.PH : y_ * cg.SH - cg.PH;
                gx[i] = cg.dgrad ? x_ + cg.PW : x_ * cg.SW - cg.PW;
                gb[i] = n_ * cg.Hs * cg.Ws;
            }
        }
    }
    // gathered B (G == 2): the (kh, kw, c) of this thread's columns is fixed over the K loop
    int bkh = 0, bkw = 0, bc = -1;
    if (G == 2) {
        const int nn = n0 + (tid % (BN / 4)) * 4;
        if (nn < cg.kreal) {
            const int tap = nn / cg.Cs;
            bc = nn - tap * cg.Cs;
            bkh = tap / cg.KW;
            bkw = tap - bkh * cg.KW;
        }
    }

    // gathered A: (kh, kw, c) of the K tile's first column.  With the channel count a multiple of the K tile, a tile never
    // straddles two taps and the triple is advanced with carries from tile to tile instead of two integer divisions per
    // fetch (~70 vector instructions per thread and tile beside 32 MFMAs)
    const bool fastk = G == 1 && (cg.Cs % GBK) == 0;
    int c_run = 0, kh_run = 0, kw_run = 0;
    if (fastk) {
        const int tap0 = kbeg / cg.Cs;
        c_run = kbeg - tap0 * cg.Cs;
        kh_run = tap0 / cg.KW;
        kw_run = tap0 - kh_run * cg.KW;
    }
    // gathered B (G == 2): pixel (n, y, x) of this thread's two K rows; with the map width dividing the K tile a tile step
    // moves y by GBK / Wm with at most one carry into n (Hm >= GBK / Wm is checked), instead of three divisions per row and fetch
    const bool fastp = G == 2 && cg.Wm > 0 && (GBK % cg.Wm) == 0 && cg.Hm >= GBK / cg.Wm;
    int px[2] = {0, 0}, py[2] = {0, 0}, pn[2] = {0, 0};
    if (fastp) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = kbeg + ((tid + i * NTHR) / (BN / 4));
            px[i] = k % cg.Wm;
            const int t_ = k / cg.Wm;
            py[i] = t_ % cg.Hm;
            pn[i] = t_ / cg.Hm;
        }
    }
    // Round 4, plain operands (G == 0) whose K range is whole tiles and whose extent fits a 31-bit byte offset (fastg, decided
    // by the host): fetched through buffer descriptors.  A thread keeps one byte offset per staged float4 (rows beyond M / N: an
    // out-of-range offset, answered with zeros), the K tile moves in the SCALAR offset: no vector address arithmetic, no bounds
    // branches, no zero-fill per tile -- the old form spent ~110 vector instructions and 6 branches per tile beside 32 MFMAs,
    // all of it on the issue port the fp32 MFMA uses ("Round 4").
    int offA[4] = {0, 0, 0, 0}, offB[2] = {0, 0};
    __amdgpu_buffer_rsrc_t rsA, rsB;
    // (k-major operands only: against the round-3 object the descriptor form of a TRANSPOSED operand measured 0-10 % slower --
    //  profiles/r04_gemm_fetch_ab.txt -- so those keep the old fetch; the gathered operand has its own path below)
    const bool fastA = G != 1 && !TA && (fastg & 1), fastB = G != 2 && !TB && (fastg & 2);
    if (fastA) {
        rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, (TA ? ((K - 1) * lda + M) : ((M - 1) * lda + K)) * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * NTHR;
            if (!TA) {
                const int m = m0 + (idx >> 3);
                offA[i] = m < M ? (m * lda + (idx & 7) * 4) * 4 : (int)0x80000000;
            } else {
                const int m = m0 + (idx % (BM / 4)) * 4;
                offA[i] = m < M ? ((idx / (BM / 4)) * lda + m) * 4 : (int)0x80000000;
            }
        }
    }
    if (fastB) {
        rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, (TB ? ((K - 1) * ldb + N) : ((N - 1) * ldb + K)) * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * NTHR;
            if (!TB) {
                const int nn = n0 + (idx >> 3);
                offB[i] = nn < N ? (nn * ldb + (idx & 7) * 4) * 4 : (int)0x80000000;
            } else {
                const int nn = n0 + (idx % (BN / 4)) * 4;
                offB[i] = nn < N ? ((idx / (BN / 4)) * ldb + nn) * 4 : (int)0x80000000;
            }
        }
    }
    // Gathered A through a buffer descriptor too (fastg bit 2; forward, or data-gradient with unit strides; channel count a
    // multiple of the K tile so that (kh, kw) is uniform per tile): the thread keeps the byte offset of its four pixels at tap
    // (0, 0) -- biased by the largest negative tap displacement so that it is never negative: the range check looks at the vector
    // offset alone -- the tap and channel of the tile go into the scalar offset, and a pixel / tap pair outside the source gets
    // an out-of-range offset (zeros).  6 vector instructions per load instead of ~20 and no branch.
    const bool fastGA = G == 1 && fastk && (fastg & 4);
    int rowoff[4] = {0, 0, 0, 0};
    __amdgpu_buffer_rsrc_t rsG;
    if (fastGA) {
        const int bias = cg.dgrad ? ((cg.KH - 1) * cg.Ws + (cg.KW - 1)) * cg.Cs : (cg.PH * cg.Ws + cg.PW) * cg.Cs;       // floats
        rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A) - bias, 0, (cg.Hs * cg.Ws * cg.Cs * (M / (cg.Hm * cg.Wm)) + bias) * 4,
                                                0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pix = cg.dgrad ? gb[i] + (gy[i] - (cg.KH - 1)) * cg.Ws + gx[i] - (cg.KW - 1) : gb[i] + gy[i] * cg.Ws + gx[i];
            rowoff[i] = gb[i] >= 0 ? (pix * cg.Cs + (tid & 7) * 4 + bias) * 4 : (int)0x80000000;
        }
    }
    // software pipeline: the next K tile is fetched into registers while the current one feeds the matrix cores
    // (GEMM_PF register sets: the tile being staged and, with 2, the one after it -- requested two K tiles ahead)
    auto fetch = [&](float4 (&ra)[4], float4 (&rb)[2], int k0) {
        if (fastA) {
            const int sa = TA ? k0 * lda * 4 : k0 * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, offA[i], sa, 0));
                ra[i] = make_float4(v.x, v.y, v.z, v.w);
            }
        }
        if (fastB) {
            const int sb = TB ? k0 * ldb * 4 : k0 * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, offB[i], sb, 0));
                rb[i] = make_float4(v.x, v.y, v.z, v.w);
            }
        }
        if (fastA) {
        } else if (fastGA) {
            const int kh = kh_run, kw = kw_run;
            const int so = cg.dgrad ? (((cg.KH - 1 - kh) * cg.Ws + (cg.KW - 1 - kw)) * cg.Cs + c_run) * 4
                                    : ((kh * cg.Ws + kw) * cg.Cs + c_run) * 4;
            c_run += GBK;
            if (c_run >= cg.Cs) {
                c_run = 0;
                if (++kw_run == cg.KW) {
                    kw_run = 0;
                    ++kh_run;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sy = cg.dgrad ? gy[i] - kh : gy[i] + kh, sx = cg.dgrad ? gx[i] - kw : gx[i] + kw;
                const bool ok = (unsigned)sy < (unsigned)cg.Hs && (unsigned)sx < (unsigned)cg.Ws;
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, ok ? rowoff[i] : (int)0x80000000, so, 0));
                ra[i] = make_float4(v.x, v.y, v.z, v.w);
            }
        } else if (G == 1) {
            const int k = k0 + (tid & 7) * 4;
            int c, kh, kw;
            if (fastk) {
                c = c_run + (tid & 7) * 4;
                kh = kh_run;
                kw = kw_run;
                c_run += GBK;
                if (c_run >= cg.Cs) {
                    c_run = 0;
                    if (++kw_run == cg.KW) {
                        kw_run = 0;
                        ++kh_run;
                    }
                }
            } else {
                const int tap = k / cg.Cs;
                c = k - tap * cg.Cs;
                kh = tap / cg.KW;
                kw = tap - kh * cg.KW;
            }
            const bool kok = k < kend && k < cg.kreal;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int sy, sx;
                bool ok = kok && gb[i] >= 0;
                if (cg.dgrad) {
                    const int ty = gy[i] - kh, tx = gx[i] - kw;
                    sy = ty / cg.SH;
                    sx = tx / cg.SW;
                    ok = ok && ty >= 0 && tx >= 0 && sy * cg.SH == ty && sx * cg.SW == tx;
                } else {
                    sy = gy[i] + kh;
                    sx = gx[i] + kw;
                    ok = ok && sy >= 0 && sx >= 0;
                }
                ok = ok && sy < cg.Hs && sx < cg.Ws;
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) ra[i] = *reinterpret_cast<const float4 *>(A + ((size_t)(gb[i] + sy * cg.Ws + sx) * cg.Cs + c));
            }
        } else if (!TA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + i * NTHR;
                const int row = idx >> 3, q = idx & 7;
                const int m = m0 + row, k = k0 + q * 4;
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < M && k < kend) ra[i] = *reinterpret_cast<const float4 *>(A + (size_t)m * lda + k);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + i * NTHR;
                const int kk = idx / (BM / 4), q = idx % (BM / 4);
                const int k = k0 + kk, m = m0 + q * 4;
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < kend && m < M) ra[i] = *reinterpret_cast<const float4 *>(A + (size_t)k * lda + m);
            }
        }
        if (fastB) {
        } else if (G == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = k0 + ((tid + i * NTHR) / (BN / 4));
                int x_, y_, n_;
                if (fastp) {                     // the pixel of this thread's K row, advanced by one K tile per fetch
                    x_ = px[i]; y_ = py[i]; n_ = pn[i];
                    py[i] += GBK / cg.Wm;
                    if (py[i] >= cg.Hm) {
                        py[i] -= cg.Hm;
                        ++pn[i];
                    }
                } else {
                    x_ = k % cg.Wm;
                    const int t_ = k / cg.Wm;
                    y_ = t_ % cg.Hm;
                    n_ = t_ / cg.Hm;
                }
                const int sy = y_ * cg.SH - cg.PH + bkh, sx = x_ * cg.SW - cg.PW + bkw;
                rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < kend && bc >= 0 && sy >= 0 && sy < cg.Hs && sx >= 0 && sx < cg.Ws)
                    rb[i] = *reinterpret_cast<const float4 *>(B + ((size_t)((n_ * cg.Hs + sy) * cg.Ws + sx) * cg.Cs + bc));
            }
        } else if (!TB) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * NTHR;
                const int row = idx >> 3, q = idx & 7;
                const int nn = n0 + row, k = k0 + q * 4;
                rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nn < N && k < kend) rb[i] = *reinterpret_cast<const float4 *>(B + (size_t)nn * ldb + k);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * NTHR;
                const int kk = idx / (BN / 4), q = idx % (BN / 4);
                const int k = k0 + kk, nn = n0 + q * 4;
                rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < kend && nn < N) rb[i] = *reinterpret_cast<const float4 *>(B + (size_t)k * ldb + nn);
            }
        }
    };
    auto stage = [&](const float4 (&ra)[4], const float4 (&rb)[2]) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * NTHR;
            if (!TA) *reinterpret_cast<float4 *>(&As[(idx >> 3) * A_LD + (idx & 7) * 4]) = ra[i];
            else *reinterpret_cast<float4 *>(&As[(idx / (BM / 4)) * A_LD + (idx % (BM / 4)) * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * NTHR;
            if (!TB) *reinterpret_cast<float4 *>(&Bs[(idx >> 3) * B_LD + (idx & 7) * 4]) = rb[i];
            else *reinterpret_cast<float4 *>(&Bs[(idx / (BN / 4)) * B_LD + (idx % (BN / 4)) * 4]) = rb[i];
        }
        __syncthreads();
    };
    auto compute = [&]() {
#pragma unroll
        for (int s = 0; s < GBK / 8; ++s) {
            float a[SM][4], b[SN][4];
            const int kk = s * 8 + lh * 4;
#pragma unroll
            for (int sm = 0; sm < SM; ++sm) {
                const int row = (wm * SM + sm) * 32 + li;
                if (!TA) {
                    const float4 v = *reinterpret_cast<const float4 *>(&As[row * A_LD + kk]);
                    a[sm][0] = v.x; a[sm][1] = v.y; a[sm][2] = v.z; a[sm][3] = v.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[sm][q] = As[(kk + q) * A_LD + row];
                }
            }
#pragma unroll
            for (int nt = 0; nt < SN; ++nt) {
                const int col = (wn * SN + nt) * 32 + li;
                if (!TB) {
                    const float4 v = *reinterpret_cast<const float4 *>(&Bs[col * B_LD + kk]);
                    b[nt][0] = v.x; b[nt][1] = v.y; b[nt][2] = v.z; b[nt][3] = v.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[nt][q] = Bs[(kk + q) * B_LD + col];
                }
            }
#pragma unroll
            for (int sm = 0; sm < SM; ++sm)
#pragma unroll
                for (int nt = 0; nt < SN; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[sm][nt] = mfma32(a[sm][q], b[nt][q], acc[sm][nt]);
        }
    };
#if GEMM_PF == 2
    float4 ra0[4], rb0[2], ra1[4], rb1[2];
    if (kbeg < kend) fetch(ra0, rb0, kbeg);
    if (kbeg + GBK < kend) fetch(ra1, rb1, kbeg + GBK);
    for (int k0 = kbeg; k0 < kend; k0 += 2 * GBK) {
        stage(ra0, rb0);
        if (k0 + 2 * GBK < kend) fetch(ra0, rb0, k0 + 2 * GBK);
        compute();
        if (k0 + GBK < kend) {
            stage(ra1, rb1);
            if (k0 + 3 * GBK < kend) fetch(ra1, rb1, k0 + 3 * GBK);
            compute();
        }
    }
#else
    float4 ra0[4], rb0[2];
    if (kbeg < kend) fetch(ra0, rb0, kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        stage(ra0, rb0);
        if (k0 + GBK < kend) fetch(ra0, rb0, k0 + GBK);
        compute();
    }
#endif
    // Vectorised epilogue (round 5): a wave's 32 x 32 SN block goes through LDS once (the MFMA result layout has a lane own ONE
    // column: 16 dword stores per accumulator, 256 bytes per wave-instruction) and leaves as 16-byte stores, 4 rows x 256 bytes per
    // wave-instruction -- 8 stores per lane instead of 32.  The dword form cost 9-43 % of a launch on the short-K shapes
    // (profiles/r05_gemm_epilogue_ab.txt: what-if without stores, then this).  Needs N, the leading dimension and the pointers
    // 16-byte aligned; anything else takes the scalar form below.
    const size_t out_ld = slab_stride ? (size_t)slab_ld : (size_t)ldc;
    const bool vec = VEPI && !(GEMM_WHATIF & 2) && !slab_stride && (N & 3) == 0 && (out_ld & 3) == 0 &&
                     (reinterpret_cast<size_t>(C) & 15) == 0 && (!bias || (reinterpret_cast<size_t>(bias) & 15) == 0);
    if (VEPI && vec) {
        __syncthreads();                                  // every wave is done with the operand tiles
        float *T = smem + wave * 32 * T_LD;
#pragma unroll
        for (int nt = 0; nt < SN; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) T[mfma_row(r, lane) * T_LD + nt * 32 + li] = acc[0][nt][r];
        constexpr int Q = 8 * SN, RPI = 64 / Q;           // float4 per row, rows per wave-instruction
        const int c4 = lane % Q, rsub = lane / Q;
        const int nn = n0 + wn * 32 * SN + c4 * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && !slab_stride && nn < N) bv = *reinterpret_cast<const float4 *>(bias + nn);
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int row = it * RPI + rsub;
            const int m = m0 + wm * 32 + row;
            float4 v = *reinterpret_cast<const float4 *>(&T[row * T_LD + c4 * 4]);
            if (m < M && nn < N) {
                if (slab_stride) {
                    *reinterpret_cast<float4 *>(C + (size_t)z * slab_stride + (size_t)m * slab_ld + nn) = v;
                } else {
                    float4 *o = reinterpret_cast<float4 *>(C + (size_t)m * ldc + nn);
                    v = make_float4(v.x * bt.alpha + bv.x, v.y * bt.alpha + bv.y, v.z * bt.alpha + bv.z, v.w * bt.alpha + bv.w);
                    if (accumulate) {
                        const float4 c0 = *o;
                        v = make_float4(v.x + c0.x, v.y + c0.y, v.z + c0.z, v.w + c0.w);
                    }
#if GEMM_ST_NT
                    const f32x4 t = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(o));
#else
                    *o = v;
#endif
                }
            }
        }
        return;
    }
#pragma unroll
    for (int sm = 0; sm < SM; ++sm)
#pragma unroll
        for (int nt = 0; nt < SN; ++nt) {
            const int nn = n0 + (wn * SN + nt) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * SM + sm) * 32 + mfma_row(r, lane);
                if (m < M && nn < N && (!(GEMM_WHATIF & 1) || r == 0)) {      // (what-if bit 0: one store per accumulator tile)
                    if (slab_stride) {
                        C[(size_t)z * slab_stride + (size_t)m * slab_ld + nn] = acc[sm][nt][r];
                    } else {
                        float v = acc[sm][nt][r] * bt.alpha;
                        if (bias) v += bias[nn];
                        const size_t o = (size_t)m * ldc + nn;
                        if (accumulate) v += C[o];
                        C[o] = v;
                    }
                }
            }
        }
}

__global__ void gemm_slab_reduce_kernel(const float *__restrict__ slabs, const float *__restrict__ bias,
                                        float *__restrict__ C, int M, int N, int ldc, int splits, int accumulate) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)M * N;
    if (idx >= total) return;
    const int m = (int)(idx / N), nn = (int)(idx - (size_t)m * N);
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(size_t)z * total + idx];
    if (bias) s += bias[nn];
    const size_t o = (size_t)m * ldc + nn;
    if (accumulate) s += C[o];
    C[o] = s;
}

// column sums: stage 1 partial[blk][c] over a strip of rows (64 columns x 4 row-lanes per workgroup, LDS
// combine), stage 2 sums the partials in double (32 columns x 8 part-groups per workgroup).
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ A, float *__restrict__ partial,
                                                             long R, int C, int lda, long rows_per_block) {
    __shared__ float red[256];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cx;
    const long rbeg = (long)blockIdx.x * rows_per_block;
    const long rend = rbeg + rows_per_block < R ? rbeg + rows_per_block : R;
    float s = 0.f;
    if (c < C)
        for (long r = rbeg + ry; r < rend; r += 4) s += A[(size_t)r * lda + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (ry == 0 && c < C) partial[(size_t)blockIdx.x * C + c] = red[cx] + red[64 + cx] + red[128 + cx] + red[192 + cx];
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ partial, float *__restrict__ out,
                                                           int nblk, int C, int accumulate) {
    __shared__ double red[256];
    const double s = block_colsum32(partial, nblk, (size_t)C, blockIdx.x * 32, C, red);
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) == 0 && c < C) out[c] = (accumulate ? out[c] : 0.f) + (float)s;
}

__global__ void add_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ y,
                           long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 u = a[i], v = b[i];
        y[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}
__global__ void mul_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ y,
                           long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 u = a[i], v = b[i];
        y[i] = make_float4(u.x * v.x, u.y * v.y, u.z * v.z, u.w * v.w);
    }
}

__global__ void scale_dev_kernel(const float4 *__restrict__ a, const float *__restrict__ s, float4 *__restrict__ y,
                                 long n4) {
    const float f = s[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 u = a[i];
        y[i] = make_float4(u.x * f, u.y * f, u.z * f, u.w * f);
    }
}

}  // namespace adyolo

using namespace adyolo;

// 256 x 128 tiles: built and measured in round 4 -- NOT faster: plain GEMMs 0.85-1.04 x of the
// 128 x 64 form (head dX 0.72 x), implicit-GEMM convolutions 0.89-1.0 x, only the split-K weight-gradient shapes gain 2-4 %
// (profiles/r04_gemm_tile_ab.txt).  One 8-wave workgroup per CU is the same two waves per SIMD as two 4-wave workgroups, but
// all eight meet at every barrier.  Kept as an opt-in for that A/B: ADYOLO_GEMM_TILE=big (auto = where >= 256 such tiles
// exist); the default is the 128 x 64 form everywhere.
// plain operands through buffer descriptors: whole K tiles per split and both operands addressable with 31-bit byte offsets
static int gemm_fast_fetch(int M, int N, int K, int lda, int ldb, int transA, int transB, int klen) {
    static const char *env = getenv("ADYOLO_GEMM_FETCH");             // "old": the round-3 fetch (A/B measurements)
    if (env && env[0] == 'o') return 0;
    if (K % GBK != 0 || klen % GBK != 0) return 0;
    const long ea = transA ? ((long)(K - 1) * lda + M) : ((long)(M - 1) * lda + K);
    const long eb = transB ? ((long)(K - 1) * ldb + N) : ((long)(N - 1) * ldb + K);
    const long lim = ((long)1 << 29) - 64;                            // floats: byte offsets (and scalar K offsets) stay below 2^31
    return ea < lim && eb < lim ? 3 : 0;                              // bit 0: operand A, bit 1: operand B
}

static bool gemm_big_tile(int M, int N, int splits, int batch) {
    static const char *env = getenv("ADYOLO_GEMM_TILE");
    if (!env || env[0] == 's') return false;
    if (env[0] == 'b') return M >= 32 && N >= 32;
    if (M < 256 || N < 128) return false;
    return (long)cdiv(M, 256) * cdiv(N, 128) * splits * batch >= 256;
}

extern "C" int adyolo_scale_dev(const float *a, const float *scalar_dev, float *y, long n, void *stream) {
    ADYOLO_REQUIRE(a && scalar_dev && y && n > 0 && n % 4 == 0, ADYOLO_EINVAL, "scale_dev: n must be a positive multiple of 4");
    const long n4 = n / 4;
    const int grid = (int)(n4 / 256 + 1 > 4096 ? 4096 : n4 / 256 + 1);
    hipLaunchKernelGGL(scale_dev_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (const float4 *)a, scalar_dev,
                       (float4 *)y, n4);
    return check_launch("scale_dev");
}

extern "C" int adyolo_gemm(const float *A, const float *B, const float *bias, float *C, float *slabs, int M,
                           int N, int K, int lda, int ldb, int ldc, int transA, int transB, int splits,
                           int accumulate, void *stream) {
    ADYOLO_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, ADYOLO_EINVAL, "gemm: bad arguments");
    // 16-byte vector loads run along the contiguous axis of each operand: k for k-major, m/n for transposed
    ADYOLO_REQUIRE(lda % 4 == 0 && ldb % 4 == 0, ADYOLO_ENOSUP, "gemm: lda=%d ldb=%d must be multiples of 4", lda, ldb);
    ADYOLO_REQUIRE((transA ? M : K) % 4 == 0 && (transB ? N : K) % 4 == 0, ADYOLO_ENOSUP,
                   "gemm: contiguous axis of each operand must be a multiple of 4 (M=%d N=%d K=%d tA=%d tB=%d)", M, N,
                   K, transA, transB);
    if (splits < 1) splits = 1;
    ADYOLO_REQUIRE(splits == 1 || slabs, ADYOLO_EINVAL, "gemm: splits > 1 needs a slab workspace");
    hipStream_t st = as_stream(stream);
    int klen = cdiv(cdiv(K, splits), GBK) * GBK;
    splits = cdiv(K, klen);
    const bool big = gemm_big_tile(M, N, splits, 1);
    const int fastg = gemm_fast_fetch(M, N, K, lda, ldb, transA, transB, klen);
    dim3 grid((unsigned)cdiv(N, big ? 128 : GBN), (unsigned)cdiv(M, big ? 256 : GBM), (unsigned)splits);
    const size_t slab_stride = splits > 1 ? (size_t)M * N : 0;
    float *out = splits > 1 ? slabs : C;
    GemmBatch bt{0, 0, 0, 0, 0, 0, 0, 1.0f};
    ConvGather cg{};
#define LAUNCH(TA_, TB_)                                                                                                     \
    do {                                                                                                                     \
        if (big)                                                                                                             \
            hipLaunchKernelGGL((gemm_kernel<TA_, TB_, 0, 4, 2, 2, 2>), grid, dim3(512), 0, st, A, B, bias, out, M, N, K, lda, ldb, \
                               ldc, klen, slab_stride, N, accumulate, bt, cg, fastg);                                        \
        else                                                                                                                 \
            hipLaunchKernelGGL((gemm_kernel<TA_, TB_>), grid, dim3(256), 0, st, A, B, bias, out, M, N, K, lda, ldb, ldc,       \
                               klen, slab_stride, N, accumulate, bt, cg, fastg);                                             \
    } while (0)
    if (transA && transB) LAUNCH(true, true);
    else if (transA) LAUNCH(true, false);
    else if (transB) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    int rc = check_launch("gemm");
    if (rc || splits == 1) return rc;
    const size_t total = (size_t)M * N;
    hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3(cdiv((long)total, 256)), dim3(256), 0, st, slabs, bias, C, M,
                       N, ldc, splits, accumulate);
    return check_launch("gemm_slab_reduce");
}

// batched GEMM for attention: batch = outer * inner problems, operand (o, i) at base + o*outer_stride + i*inner_stride;
// C = alpha * opA opB^T-style product as in adyolo_gemm (no bias, no split-K)
extern "C" int adyolo_gemm_batched(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb,
                                   int ldc, int transA, int transB, int outer, int inner, long oA, long iA, long oB,
                                   long iB, long oC, long iC, float alpha, int accumulate, void *stream) {
    ADYOLO_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && outer > 0 && inner > 0, ADYOLO_EINVAL, "gemm_batched: bad arguments");
    ADYOLO_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && (transA ? M : K) % 4 == 0 && (transB ? N : K) % 4 == 0 &&
                       oA % 4 == 0 && iA % 4 == 0 && oB % 4 == 0 && iB % 4 == 0,
                   ADYOLO_ENOSUP, "gemm_batched: contiguous axes, leading dimensions and batch strides must be multiples of 4");
    ADYOLO_REQUIRE((long)outer * inner <= 65535, ADYOLO_ENOSUP, "gemm_batched: more than 65535 problems");
    hipStream_t st = as_stream(stream);
    const int klen = cdiv(K, GBK) * GBK;
    dim3 grid((unsigned)cdiv(N, GBN), (unsigned)cdiv(M, GBM), (unsigned)(outer * inner));
    GemmBatch bt{inner, oA, iA, oB, iB, oC, iC, alpha};
    const float *bias = nullptr;
    const size_t slab_stride = 0;
    ConvGather cg{};
#define LAUNCHB(TA_, TB_)                                                                                     \
    hipLaunchKernelGGL((gemm_kernel<TA_, TB_>), grid, dim3(256), 0, st, A, B, bias, C, M, N, K, lda, ldb, ldc, \
                       klen, slab_stride, N, accumulate, bt, cg, 0)
    if (transA && transB) LAUNCHB(true, true);
    else if (transA) LAUNCHB(true, false);
    else if (transB) LAUNCHB(false, true);
    else LAUNCHB(false, false);
#undef LAUNCHB
    return check_launch("gemm_batched");
}

// General strided convolution on channels-last tensors as an implicit GEMM (no column buffer).
//   mode 0 (forward):        out y  [N*Ho*Wo][Cout] = gather(x)  . wk^T      src = x [N][H][W][Cin],    other = wk  [Cout][Kp]
//   mode 1 (data-gradient):  out dx [N*H*W][Cin]    = gather(dy) . wkT^T     src = dy [N][Ho][Wo][Cout], other = wkT [Cin][Kq]
//   mode 2 (weight-gradient):out dwk[Cout][Kp]      = dy^T . gather(x)       src = x,                    other = dy
// with Kp = roundup4(KH*KW*Cin), k = (kh*KW + kw)*Cin + ci and Kq = roundup4(KH*KW*Cout), k = (kh*KW + kw)*Cout + co.
extern "C" int adyolo_conv_gemm(const float *src, const float *other, float *out, float *slabs, int mode, int N, int H,
                                int W, int Cin, int Cout, int KH, int KW, int SH, int SW, int PH, int PW, int splits,
                                void *stream) {
    ADYOLO_REQUIRE(src && other && out && N > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && SH > 0 && SW > 0 && PH >= 0 &&
                       PW >= 0 && mode >= 0 && mode <= 2,
                   ADYOLO_EINVAL, "conv_gemm: bad arguments");
    ADYOLO_REQUIRE(Cin > 0 && Cout > 0 && Cin % 4 == 0 && Cout % 4 == 0, ADYOLO_ENOSUP,
                   "conv_gemm: Cin=%d and Cout=%d must be multiples of 4", Cin, Cout);
    const int Ho = (H + 2 * PH - KH) / SH + 1, Wo = (W + 2 * PW - KW) / SW + 1;
    ADYOLO_REQUIRE(Ho > 0 && Wo > 0, ADYOLO_EINVAL, "conv_gemm: empty output");
    ADYOLO_REQUIRE((size_t)N * H * W * Cin < ((size_t)1 << 31) && (size_t)N * Ho * Wo * Cout < ((size_t)1 << 31), ADYOLO_ENOSUP,
                   "conv_gemm: tensors must stay below 2^31 elements");
    hipStream_t st = as_stream(stream);
    const int Kp = cdiv(KH * KW * Cin, 4) * 4, Kq = cdiv(KH * KW * Cout, 4) * 4;
    GemmBatch bt{0, 0, 0, 0, 0, 0, 0, 1.0f};
    const float *bias = nullptr;
    int M, Nn, K;
    ConvGather cg;
    if (mode == 0) {
        M = N * Ho * Wo, Nn = Cout, K = Kp;
        cg = ConvGather{H, W, Cin, Ho, Wo, KH, KW, SH, SW, PH, PW, 0, KH * KW * Cin};
    } else if (mode == 1) {
        M = N * H * W, Nn = Cin, K = Kq;
        cg = ConvGather{Ho, Wo, Cout, H, W, KH, KW, SH, SW, PH, PW, 1, KH * KW * Cout};
    } else {
        M = Cout, Nn = Kp, K = N * Ho * Wo;
        cg = ConvGather{H, W, Cin, Ho, Wo, KH, KW, SH, SW, PH, PW, 0, KH * KW * Cin};
    }
    if (mode != 2 || splits < 1) splits = mode == 2 ? (splits < 1 ? 1 : splits) : 1;
    ADYOLO_REQUIRE(splits == 1 || slabs, ADYOLO_EINVAL, "conv_gemm: splits > 1 needs a slab workspace");
    const int klen = cdiv(cdiv(K, splits), GBK) * GBK;
    splits = cdiv(K, klen);
    const bool big = gemm_big_tile(M, Nn, splits, 1);
    // the operand that is NOT gathered goes through the buffer-descriptor fetch: mode 2: A = dy [pixels][Cout] (transposed,
    // lda = Cout); modes 0 / 1: B = the packed filter [Nn][K] (k-major, ldb = K)
    int fastg = mode == 2 ? (gemm_fast_fetch(M, 4, K, Cout, 4, 1, 1, klen) & 1)
                          : (gemm_fast_fetch(4, Nn, K, 4, mode == 0 ? Kp : Kq, 0, 0, klen) & 2);
    // bit 2: the gathered operand of a forward / unit-stride data-gradient launch through a descriptor as well (whole K tiles
    // inside one tap: source channels a multiple of the K tile; source below 2^29 floats incl. the tap bias)
    if (fastg && mode != 2 && cg.Cs % GBK == 0 && (mode == 0 || (SH == 1 && SW == 1)) &&
        (long)N * cg.Hs * cg.Ws * cg.Cs + (long)(KH * cg.Ws + KW) * cg.Cs < ((long)1 << 29))
        fastg |= 4;
    dim3 grid((unsigned)cdiv(Nn, big ? 128 : GBN), (unsigned)cdiv(M, big ? 256 : GBM), (unsigned)splits);
    const size_t slab_stride = splits > 1 ? (size_t)M * Nn : 0;
    float *dst = splits > 1 ? slabs : out;
    if (mode == 2) {
        if (big)
            hipLaunchKernelGGL((gemm_kernel<true, true, 2, 4, 2, 2, 2>), grid, dim3(512), 0, st, other, src, bias, dst, M, Nn, K,
                               Cout, 0, Nn, klen, slab_stride, Nn, 0, bt, cg, fastg);
        else
            hipLaunchKernelGGL((gemm_kernel<true, true, 2>), grid, dim3(256), 0, st, other, src, bias, dst, M, Nn, K, Cout, 0,
                               Nn, klen, slab_stride, Nn, 0, bt, cg, fastg);
    } else {
        if (big)
            hipLaunchKernelGGL((gemm_kernel<false, false, 1, 4, 2, 2, 2>), grid, dim3(512), 0, st, src, other, bias, dst, M, Nn, K,
                               0, mode == 0 ? Kp : Kq, Nn, klen, slab_stride, Nn, 0, bt, cg, fastg);
        else
            hipLaunchKernelGGL((gemm_kernel<false, false, 1>), grid, dim3(256), 0, st, src, other, bias, dst, M, Nn, K, 0,
                               mode == 0 ? Kp : Kq, Nn, klen, slab_stride, Nn, 0, bt, cg, fastg);
    }
    int rc = check_launch("conv_gemm");
    if (rc || splits == 1) return rc;
    const size_t total = (size_t)M * Nn;
    hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3(cdiv((long)total, 256)), dim3(256), 0, st, slabs, bias, out, M, Nn,
                       Nn, splits, 0);
    return check_launch("conv_gemm_reduce");
}

extern "C" int adyolo_colsum(const float *A, float *out, float *partial, int R, int C, int lda, int accumulate,
                             void *stream) {
    ADYOLO_REQUIRE(A && out && partial && R > 0 && C > 0, ADYOLO_EINVAL, "colsum: bad arguments");
    hipStream_t st = as_stream(stream);
    int nblk = cdiv(R, 64);
    if (nblk > 1024) nblk = 1024;
    const long rpb = ((long)R + nblk - 1) / nblk;
    nblk = cdiv(R, rpb);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk, cdiv(C, 64)), dim3(256), 0, st, A, partial, (long)R, C, lda, rpb);
    int rc = check_launch("colsum_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, partial, out, nblk, C, accumulate);
    return check_launch("colsum_final");
}

extern "C" int adyolo_add(const float *a, const float *b, float *y, long n, void *stream) {
    ADYOLO_REQUIRE(a && b && y && n > 0 && n % 4 == 0, ADYOLO_EINVAL, "add: n must be a positive multiple of 4");
    const long n4 = n / 4;
    const int grid = (int)(n4 / 256 + 1 > 4096 ? 4096 : n4 / 256 + 1);
    hipLaunchKernelGGL(add_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (const float4 *)a, (const float4 *)b,
                       (float4 *)y, n4);
    return check_launch("add");
}
extern "C" int adyolo_mul(const float *a, const float *b, float *y, long n, void *stream) {
    ADYOLO_REQUIRE(a && b && y && n > 0 && n % 4 == 0, ADYOLO_EINVAL, "mul: n must be a positive multiple of 4");
    const long n4 = n / 4;
    const int grid = (int)(n4 / 256 + 1 > 4096 ? 4096 : n4 / 256 + 1);
    hipLaunchKernelGGL(mul_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (const float4 *)a, (const float4 *)b,
                       (float4 *)y, n4);
    return check_launch("mul");
}
