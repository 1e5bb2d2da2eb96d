// K2: 3x3 convolution (stride 1, pad 1), channels-last, as an implicit GEMM on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32).  Replaces the cuDNN/MIOpen/oneDNN calls behind nn.Conv2d at
// /root/reference/src/models/backbones/resnet.py:16,18,142 (forward, data-gradient, weight-gradient).
//
// Forward / data-gradient (same kernel, different weight packing):
//   GEMM view  M = pixels, N = Cout, K = 9*Cin ordered (tap, cin).  One workgroup (4 waves) owns a
//   TH x TW = 256-pixel patch and BN output channels.  Per 32-channel input chunk the patch plus its
//   1-pixel halo is staged ONCE in LDS ([(TH+2)(TW+2)][KC+4] floats, 144-B rows: ds_read_b128 hits 16
//   distinct 16-B slots per lane group) and all 9 taps read it at shifted addresses; the per-tap weight
//   slice [BN][KC] is double-buffered in LDS and prefetched through registers under the previous tap's
//   MFMAs.  LDS = 48.9 KB + 2 x 9.2 KB = 67 KB -> two workgroups per CU overlap one's staging with the
//   other's matrix work.  K is walked 8 at a time: lane half h holds k = 4h..4h+3 of both operands (one
//   ds_read_b128 each), feeding 4 MFMAs -- the sum over k is order-free so A and B only need to agree.
// Weight-gradient:
//   GEMM view  M = Cout, N = (tap, Cin), K = pixels.  One workgroup owns a (32 cout x 32 cin) block of
//   every tap (9 accumulator tiles per wave) and walks pixel patches; the 4 waves split each patch's
//   rows and are summed through LDS at the end; each workgroup writes one slab, slabs are reduced
//   deterministically into the reference layout [Cout][Cin][3][3].
#include <stdlib.h>
#include "common.hpp"

namespace adyolo {

template <int KC, int BN, int TW>
struct ConvCfg {
    static constexpr int TH = 256 / TW;
    static constexpr int HW_ = TW + 2;
    static constexpr int HH_ = TH + 2;
    static constexpr int NPIX = HW_ * HH_;
    static constexpr int AS = KC + 4;
    static constexpr int NT = BN / 32;
    static constexpr int Q = KC / 4;
    static constexpr int A_FLOATS = NPIX * AS;
    static constexpr int W_FLOATS = BN * AS;
    static constexpr int WPT = (BN * Q + 255) / 256;
};

template <int KC, int BN, int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ wpk, const float *__restrict__ bias,
    const float *__restrict__ addend, const float *__restrict__ addend_mask, const float *__restrict__ in_scale,
    const float *__restrict__ in_shift, float *__restrict__ y, float *__restrict__ stats,
    const float *__restrict__ stat_aux, const float *__restrict__ stat_mean, const float *__restrict__ stat_invstd,
    const float *__restrict__ stat_mask, int H, int W, int Cin, int Cout, int tilesW, int tilesH, int relu,
    int mask_bits) {
    using Cfg = ConvCfg<KC, BN, TW>;
    constexpr int TH = Cfg::TH, HW_ = Cfg::HW_, NPIX = Cfg::NPIX, AS = Cfg::AS, NT = Cfg::NT, Q = Cfg::Q;
    constexpr int WPT = Cfg::WPT;
    __shared__ __attribute__((aligned(16))) float As[Cfg::A_FLOATS];
    __shared__ __attribute__((aligned(16))) float Ws[2][Cfg::W_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    const int tw = bid % tilesW;
    bid /= tilesW;
    const int th = bid % tilesH;
    const int n = bid / tilesH;
    const int co0 = blockIdx.y * BN;
    const int ty0 = th * TH, tx0 = tw * TW;

    int prow[2], pcol[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        if (TW == 32) {
            prow[mt] = wave * 2 + mt;
            pcol[mt] = li;
        } else {
            prow[mt] = (wave * 2 + mt) * 2 + (li >> 4);
            pcol[mt] = li & 15;
        }
    }
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int nchunks = Cin / KC;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * KC;
        __syncthreads();
        // stage the halo patch: issue every global load first (registers), then write LDS -- one exposed
        // memory latency per chunk instead of one per 16-byte piece
        constexpr int APT = (NPIX * Q + 255) / 256;
        constexpr int PSTEP = 256 / Q;                  // pixels advanced per pass; the 16-byte piece q is fixed per thread
        const int sq = tid % Q, spix0 = tid / Q;
        float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in_scale) {                                 // fused BatchNorm affine of the producer (padding stays exactly 0)
            isc = *reinterpret_cast<const float4 *>(in_scale + c0 + sq * 4);
            ish = *reinterpret_cast<const float4 *>(in_shift + c0 + sq * 4);
        }
        float4 av[APT];
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int pix = spix0 + i * PSTEP;
            av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pix < NPIX) {
                const int hy = pix / HW_, hx = pix - hy * HW_;
                const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    const float4 v = *reinterpret_cast<const float4 *>(x + (((size_t)n * H + gy) * W + gx) * Cin + c0 + sq * 4);
                    av[i] = make_float4(v.x * isc.x + ish.x, v.y * isc.y + ish.y, v.z * isc.z + ish.z, v.w * isc.w + ish.w);
                }
            }
        }
        // (named registers, not an array: hipcc would otherwise promote the tiny array to LDS)
        float4 w0a = make_float4(0.f, 0.f, 0.f, 0.f), w0b = w0a;
        const int wco_a = tid / Q, wq_a = tid - wco_a * Q;
        const int wco_b = (tid + 256) / Q, wq_b = (tid + 256) - wco_b * Q;
        if (tid < BN * Q) w0a = *reinterpret_cast<const float4 *>(wpk + ((size_t)(co0 + wco_a) * 9 + 0) * Cin + c0 + wq_a * 4);
        if (WPT > 1 && tid + 256 < BN * Q)
            w0b = *reinterpret_cast<const float4 *>(wpk + ((size_t)(co0 + wco_b) * 9 + 0) * Cin + c0 + wq_b * 4);
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int pix = spix0 + i * PSTEP;
            if (pix < NPIX) *reinterpret_cast<float4 *>(&As[pix * AS + sq * 4]) = av[i];
        }
        if (tid < BN * Q) *reinterpret_cast<float4 *>(&Ws[0][wco_a * AS + wq_a * 4]) = w0a;
        if (WPT > 1 && tid + 256 < BN * Q) *reinterpret_cast<float4 *>(&Ws[0][wco_b * AS + wq_b * 4]) = w0b;
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int cur = tap & 1;
            float4 wna = make_float4(0.f, 0.f, 0.f, 0.f), wnb = wna;
            if (tap < 8) {
                if (tid < BN * Q)
                    wna = *reinterpret_cast<const float4 *>(wpk + ((size_t)(co0 + wco_a) * 9 + tap + 1) * Cin + c0 + wq_a * 4);
                if (WPT > 1 && tid + 256 < BN * Q)
                    wnb = *reinterpret_cast<const float4 *>(wpk + ((size_t)(co0 + wco_b) * 9 + tap + 1) * Cin + c0 + wq_b * 4);
            }
#pragma unroll
            for (int s = 0; s < KC / 8; ++s) {
                float4 a[2], b[NT];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    a[mt] = *reinterpret_cast<const float4 *>(
                        &As[((prow[mt] + ky) * HW_ + pcol[mt] + kx) * AS + s * 8 + lh * 4]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    b[nt] = *reinterpret_cast<const float4 *>(&Ws[cur][(nt * 32 + li) * AS + s * 8 + lh * 4]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[mt][nt] = mfma32(a[mt].x, b[nt].x, acc[mt][nt]);
                        acc[mt][nt] = mfma32(a[mt].y, b[nt].y, acc[mt][nt]);
                        acc[mt][nt] = mfma32(a[mt].z, b[nt].z, acc[mt][nt]);
                        acc[mt][nt] = mfma32(a[mt].w, b[nt].w, acc[mt][nt]);
                    }
            }
            if (tap < 8) {
                if (tid < BN * Q) *reinterpret_cast<float4 *>(&Ws[cur ^ 1][wco_a * AS + wq_a * 4]) = wna;
                if (WPT > 1 && tid + 256 < BN * Q) *reinterpret_cast<float4 *>(&Ws[cur ^ 1][wco_b * AS + wq_b * 4]) = wnb;
                __syncthreads();
            }
        }
    }

    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pr = mfma_row(r, lane);
            int row, col;
            if (TW == 32) {
                row = wave * 2 + mt;
                col = pr;
            } else {
                row = (wave * 2 + mt) * 2 + (pr >> 4);
                col = pr & 15;
            }
            const int gy = ty0 + row, gx = tx0 + col;
            if (gy < H && gx < W) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int co = co0 + nt * 32 + li;
                    const size_t o = (((size_t)n * H + gy) * W + gx) * Cout + co;
                    float v = acc[mt][nt][r];
                    if (bias) v += bias[co];
                    if (addend) {
                        const float ad = addend[o];
                        const bool keep = !addend_mask ? true
                                          : ((mask_bits & 1) ? mask_bit1(reinterpret_cast<const unsigned long long *>(addend_mask), o)
                                                             : addend_mask[o] > 0.f);
                        v += keep ? ad : 0.f;
                    }
                    if (relu) v = fmaxf(v, 0.f);
                    y[o] = v;
                    if (stat_mask) {                                      // statistics of v * (mask > 0)
                        const bool keep = (mask_bits & 2) ? mask_bit1(reinterpret_cast<const unsigned long long *>(stat_mask), o)
                                                          : stat_mask[o] > 0.f;
                        if (!keep) v = 0.f;
                    }
                    ssum[nt] += v;
                    // second statistic: v^2 (BatchNorm forward of the consumer) or v * xhat(aux) (BatchNorm backward:
                    // sum dy * xhat, with aux = the BatchNorm input living at the same positions as this output)
                    ssq[nt] += stat_aux ? v * (stat_aux[o] - stat_mean[co]) * stat_invstd[co] : v * v;
                }
            }
        }
    }
    if (stats) {
        // per-tile, per-channel sum / sum of squares of the stored output: BatchNorm statistics (and the SE
        // squeeze) without a separate read pass.  layout [2][tiles][Cout], tile = blockIdx.x
        __syncthreads();
        float *red = As;                               // 4 waves x BN x 2 floats
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float a0 = ssum[nt] + __shfl_xor(ssum[nt], 32, 64);
            const float a1 = ssq[nt] + __shfl_xor(ssq[nt], 32, 64);
            if (lh == 0) {
                red[(wave * BN + nt * 32 + li) * 2 + 0] = a0;
                red[(wave * BN + nt * 32 + li) * 2 + 1] = a1;
            }
        }
        __syncthreads();
        if (tid < BN * 2) {
            const int c = tid >> 1, which = tid & 1;
            const float v = red[(0 * BN + c) * 2 + which] + red[(1 * BN + c) * 2 + which] +
                            red[(2 * BN + c) * 2 + which] + red[(3 * BN + c) * 2 + which];
            const size_t half = (size_t)gridDim.x * Cout;
            stats[which * half + (size_t)blockIdx.x * Cout + co0 + c] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ in_scale,
    const float *__restrict__ in_shift, float *__restrict__ slabs, int H, int W, int Cin, int Cout, int tilesW,
    int tilesH, int ntiles, int nsplit, int cinBlocks) {
    constexpr int TH = 256 / TW, HW_ = TW + 2, HH_ = TH + 2, NPIX = HW_ * HH_;
    constexpr int XS = 32, DS = 32;
    __shared__ __attribute__((aligned(16))) float Xs[NPIX * XS];   // 43.5 KB, reused for the cross-wave sum
    __shared__ __attribute__((aligned(16))) float Ds[256 * DS];    // 32 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x;
    const int cb = blockIdx.y / cinBlocks, ib = blockIdx.y - cb * cinBlocks;
    const int co0 = cb * 32, c0 = ib * 32;
    const int cvalid = min(32, Cin - c0);       // 8 for the stem (activations padded 7 -> 8)
    const int CinP = cinBlocks * 32;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int tile = split; tile < ntiles; tile += nsplit) {
        int b = tile;
        const int tw = b % tilesW;
        b /= tilesW;
        const int th = b % tilesH;
        const int n = b / tilesH;
        const int ty0 = th * TH, tx0 = tw * TW;
        __syncthreads();
        {
            constexpr int XPT = (NPIX * 8 + 255) / 256;
            const int sq = tid & 7, spix0 = tid >> 3;        // piece q fixed per thread, 32 pixels per pass
            float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in_scale && sq * 4 < cvalid) {
                isc = *reinterpret_cast<const float4 *>(in_scale + c0 + sq * 4);
                ish = *reinterpret_cast<const float4 *>(in_shift + c0 + sq * 4);
            }
            float4 xv[XPT];
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int pix = spix0 + i * 32;
                xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pix < NPIX) {
                    const int hy = pix / HW_, hx = pix - hy * HW_;
                    const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
                    if (sq * 4 < cvalid && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                        const float4 v = *reinterpret_cast<const float4 *>(x + (((size_t)n * H + gy) * W + gx) * Cin + c0 + sq * 4);
                        xv[i] = make_float4(v.x * isc.x + ish.x, v.y * isc.y + ish.y, v.z * isc.z + ish.z, v.w * isc.w + ish.w);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int pix = spix0 + i * 32;
                if (pix < NPIX) *reinterpret_cast<float4 *>(&Xs[pix * XS + sq * 4]) = xv[i];
            }
        }
        {
            float4 dv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = tid + i * 256;
                const int pix = idx >> 3, q = idx & 7;
                const int py = pix / TW, px = pix - py * TW;
                const int gy = ty0 + py, gx = tx0 + px;
                dv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gy < H && gx < W)
                    dv[i] = *reinterpret_cast<const float4 *>(dy + (((size_t)n * H + gy) * W + gx) * Cout + co0 + q * 4);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = tid + i * 256;
                *reinterpret_cast<float4 *>(&Ds[(idx >> 3) * DS + (idx & 7) * 4]) = dv[i];
            }
        }
        __syncthreads();
        constexpr int ROWS_PER_WAVE = TH / 4;     // 2 (TW=32) or 4 (TW=16)
        constexpr int KSTEPS = TW / 8;            // 4 or 2
#pragma unroll 1
        for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
            const int row = wave * ROWS_PER_WAVE + rr;
#pragma unroll 1
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int cbase = ks * 8 + lh * 4;
                float a[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = Ds[(row * TW + cbase + q) * DS + li];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float bq = Xs[((row + ky) * HW_ + cbase + q + kx) * XS + li];
                        acc[tap] = mfma32(a[q], bq, acc[tap]);
                    }
                }
            }
        }
    }
    // cross-wave sum through LDS (Xs has 10880 >= 9216 floats), then one slab per workgroup
    float *red = Xs;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(tap * 16 + r) * 64 + lane] = acc[tap][r];
    }
#pragma unroll 1
    for (int w = 1; w < 3; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(tap * 16 + r) * 64 + lane] += acc[tap][r];
                asm volatile("" ::: "memory");      // one tap's LDS traffic in flight at a time
            }
        }
    }
    __syncthreads();
    if (wave == 3) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + mfma_row(r, lane);
                slabs[(((size_t)split * Cout + co) * 9 + tap) * CinP + c0 + li] =
                    acc[tap][r] + red[(tap * 16 + r) * 64 + lane];
            }
            asm volatile("" ::: "memory");
        }
    }
}

// Forward of the 8-channel stem convolution (Cin = 8 padded from 7, Cout = 32, bias + ReLU + per-patch BatchNorm sums;
// reference resnet.py:142,183-184).  One (tap, 8 channels) slice of a pixel is exactly the 8 k-values four MFMAs consume,
// so the A operand is ONE float4 per lane and tap read straight from memory (lanes (pixel, half) cover 32 consecutive
// pixels x 32 bytes = 1 KB contiguous), the 9 x 8 x 32 filter lives in 36 registers, and nothing goes through LDS but the
// per-patch sums.  Same patch grid (8 x 32 pixels per workgroup, two image rows per wave) and statistics layout as
// conv3x3_fwd_kernel<8, 32, 32>, which staged a halo patch in LDS for a 72-deep contraction: 0.78 ms at B = 64 x 60 s for
// a 1.57 GB read + write.
__global__ __launch_bounds__(256, 4) void stem_fwd_kernel(const float *__restrict__ x, const float *__restrict__ wpk,
                                                          const float *__restrict__ bias, float *__restrict__ y,
                                                          float *__restrict__ stats, int H, int W, int tilesW, int tilesH,
                                                          int relu) {
    __shared__ float red[4 * 32 * 2];
    // output tiles of the waves (32 pixels x 32 channels each, rows padded to 36 floats): the MFMA result has a lane own one
    // CHANNEL of 16 pixels, so the direct epilogue was 16 dword stores per lane and row (256 B per wave-instruction) for a
    // 1.26 GB output; through LDS a lane stores four 16-byte pieces (1 KB per wave-instruction, whole 128-byte pixels; round 6)
    __shared__ __attribute__((aligned(16))) float otile[4][32 * 36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    const int tw = bid % tilesW;
    bid /= tilesW;
    const int th = bid % tilesH;
    const int n = bid / tilesH;
    const int ty0 = th * 8, tx0 = tw * 32;
    float4 bw[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) bw[tap] = *reinterpret_cast<const float4 *>(wpk + (li * 9 + tap) * 8 + lh * 4);
    const float bv = bias ? bias[li] : 0.f;
    const float *xn = x + (size_t)n * H * W * 8 + lh * 4;
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int gy = ty0 + wave * 2 + mt;
        float4 a[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = gy + tap / 3 - 1, xx = tx0 + li + tap % 3 - 1;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const int cy = min(max(yy, 0), H - 1), cx = min(max(xx, 0), W - 1);
            const float4 v = *reinterpret_cast<const float4 *>(xn + ((size_t)cy * W + cx) * 8);
            a[tap] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            acc = mfma32(a[tap].x, bw[tap].x, acc);
            acc = mfma32(a[tap].y, bw[tap].y, acc);
            acc = mfma32(a[tap].z, bw[tap].z, acc);
            acc = mfma32(a[tap].w, bw[tap].w, acc);
        }
        if (gy < H) {                                   // (wave-uniform)
            float *ot = otile[wave];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = mfma_row(r, lane);
                float v = acc[r] + bv;
                if (relu) v = fmaxf(v, 0.f);
                ot[px * 36 + li] = v;
                if (tx0 + px < W) {
                    ssum += v;
                    float sq = v * v;            // (a product and a sum, not an fma -- the empty asm keeps the compiler from contracting
                    asm volatile("" : "+v"(sq));  //  them: the statistics of every round of this kernel agree bit for bit)
                    ssq += sq;
                }
            }
            // (the tile is private to the wave: its LDS writes are ordered before its reads by the wait the compiler inserts)
            float *yrow = y + (((size_t)n * H + gy) * W + tx0) * 32;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int px = k * 8 + (lane >> 3), c4 = lane & 7;
                const f32x4 v4 = *reinterpret_cast<const f32x4 *>(&ot[px * 36 + c4 * 4]);
                if (tx0 + px < W) __builtin_nontemporal_store(v4, reinterpret_cast<f32x4 *>(yrow + (size_t)px * 32 + c4 * 4));
            }
        }
    }
    if (stats) {        // layout [2][tiles][32], tile = blockIdx.x (as conv3x3_fwd_kernel)
        const float a0 = ssum + __shfl_xor(ssum, 32, 64), a1 = ssq + __shfl_xor(ssq, 32, 64);
        if (lh == 0) {
            red[(wave * 32 + li) * 2 + 0] = a0;
            red[(wave * 32 + li) * 2 + 1] = a1;
        }
        __syncthreads();
        if (tid < 64) {
            const int c = tid >> 1, which = tid & 1;
            const float v = red[(0 * 32 + c) * 2 + which] + red[(1 * 32 + c) * 2 + which] + red[(2 * 32 + c) * 2 + which] +
                            red[(3 * 32 + c) * 2 + which];
            stats[which * ((size_t)gridDim.x * 32) + (size_t)blockIdx.x * 32 + c] = v;
        }
    }
}

// Weight gradient of the 8-channel stem convolution (Cin = 8 padded from 7, Cout = 32; reference resnet.py:142).  As a GEMM
// it is D[co][n = tap*8 + ci] = sum over pixels dy[pix][co] * x[pix + tap][ci]: M = 32 output channels is exactly one MFMA
// tile, N = 72 is three (the last a quarter full) and the contraction runs over pixels, so both operands ARE in MFMA layout
// in memory: lane (co, pixel parity) needs dy[pixel][co] and lane (n, pixel parity) needs x[pixel + tap][ci] -- no
// transposition.  Until round 6 the lanes read exactly that from global memory, four dword loads per three MFMAs, and the
// kernel was bound by the address path of those gathers (0.91 ms at B = 64 x 60 s for a 1.57 GB read; 0.38 ms of MFMAs).  Now a
// workgroup stages four image rows x 64 pixels of dy and the six x rows around them (one halo column each side, out-of-image
// pixels as zeros) in LDS with 16-byte loads -- ~12 per thread and chunk, requested one chunk ahead -- and the lanes take their
// operands from there with ds_read_b32 at immediate offsets: no predicate and no address arithmetic in the loop.  Rows, pixel
// pairs and accumulation order are the old kernel's: the result is bit-identical.
constexpr int SW_U = 8;                                   // pixel pairs read ahead of their MFMAs
constexpr int SW_T = 64;                                  // pixel columns per chunk
constexpr int SW_XS = (SW_T + 4) * 8;                     // floats per staged x row: 66 pixels (a halo column each side) + 2 of padding --
                                                          // 544 = 32 mod 64, which keeps the three tap rows of a B read in different banks
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                            float *__restrict__ slabs, int H, int W, int rows_total) {
    // dy [4 rows][64 pixels][32] | x [6 rows + a row of zeros][68 pixels][8]; the final reduction reuses the space
    __shared__ __attribute__((aligned(16))) float lds[4 * SW_T * 32 + 7 * SW_XS];
    static_assert(4 * SW_T * 32 + 7 * SW_XS >= 4 * 32 * 72, "the reduction buffer must fit");
    float *dyL = lds, *xL = lds + 4 * SW_T * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int ky[3], boff[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = t * 32 + li, tap = n < 72 ? n >> 3 : 8;        // (columns 72 .. 95 are never stored: any address will do)
        ky[t] = tap / 3 - 1;
        boff[t] = (lh + 1 + (tap % 3 - 1)) * 8 + (n & 7);            // floats inside a staged x row, pixel pair 0
    }
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int i = tid; i < SW_XS; i += 256) xL[6 * SW_XS + i] = 0.f;

    // rows are dealt to the workgroups in contiguous runs, one row of a four-row chunk per wave
    const int per = (rows_total + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * per, r1 = min(rows_total, r0 + per);
    const int ncol = (W + SW_T - 1) / SW_T;
    const int nchunk = r0 < r1 ? ((r1 - r0 + 3) / 4) * ncol : 0;
    f32x4 pd[8], px_[4];
    // chunk -> registers: 16-byte buffer loads against per-chunk resources (offsets stay small whatever the tensor's size);
    // a pixel outside the image, the workgroup's rows or the tensor gets the out-of-range offset and reads as zeros
    auto request = [&](int chunk) {
        const int cr = r0 + (chunk / ncol) * 4, c0 = (chunk % ncol) * SW_T;
        const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dy) + (size_t)cr * W * 32, 0,
                                                                              0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x) + ((ptrdiff_t)cr - 1) * W * 8, 0,
                                                                              0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = tid + 256 * i;
            const int j = f >> 9, col = c0 + ((f & 511) >> 3);
            const bool ok = cr + j < r1 && col < W;
            pd[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  drs, ok ? ((j * W + col) * 32 + (f & 7) * 4) * 4 : (int)0x80000000, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i;
            const int j = f / 132, rem = f - j * 132;
            const int row = cr - 1 + j, col = c0 - 1 + (rem >> 1);
            const bool ok = f < 6 * 132 && row >= 0 && row < rows_total && col >= 0 && col < W;
            px_[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   xrs, ok ? ((j * W + col) * 8 + (rem & 1) * 4) * 4 : (int)0x80000000, 0, 0));
        }
    };
    auto deposit = [&]() {                                // registers -> LDS
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4 *>(&dyL[(tid + 256 * i) * 4]) = pd[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i;
            const int j = f / 132;
            if (f < 6 * 132) *reinterpret_cast<f32x4 *>(&xL[j * SW_XS + (f - j * 132) * 4]) = px_[i];
        }
    };
    if (nchunk > 0) request(0);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        deposit();
        __syncthreads();
        if (chunk + 1 < nchunk) request(chunk + 1);
        const int row = r0 + (chunk / ncol) * 4 + wave;
        if (row < r1) {
            const int y = row % H;
            const float *ap = dyL + (wave * SW_T + lh) * 32 + li;
            const float *bp[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int yy = y + ky[t];
                bp[t] = xL + ((yy >= 0 && yy < H) ? wave + 1 + ky[t] : 6) * SW_XS + boff[t];
            }
            // the reads of group g + 1 are issued in front of the MFMAs of group g (two register sets; pinned, or the scheduler
            // sinks every read to just above its use and exposes the LDS latency 32 times per chunk)
            constexpr int NG = SW_T / (2 * SW_U);
            float a[2][SW_U], b[2][3][SW_U];
            auto reads = [&](int g) {
#pragma unroll
                for (int u = 0; u < SW_U; ++u) {
                    a[g & 1][u] = ap[(g * SW_U + u) * 64];
#pragma unroll
                    for (int t = 0; t < 3; ++t) b[g & 1][t][u] = bp[t][(g * SW_U + u) * 16];
                }
            };
            reads(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < NG) reads(g + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < SW_U; ++u)
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[t] = mfma32(a[g & 1][u], b[g & 1][t][u], acc[t]);
            }
        }
        __syncthreads();
    }
    // the four waves' partial sums are added in wave order; one slab [32][9][8] per workgroup
    float *red = lds;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = t * 32 + li;
            if (n < 72) red[(wave * 32 + mfma_row(r, lane)) * 72 + n] = acc[t][r];
        }
    __syncthreads();
    for (int i = tid; i < 32 * 72; i += 256)
        slabs[(size_t)blockIdx.x * (32 * 72) + i] = ((red[i] + red[32 * 72 + i]) + red[2 * 32 * 72 + i]) + red[3 * 32 * 72 + i];
}

// slabs [nslab][Cout][9][CinP] -> dw [Cout][Cin_real][3][3]: 32 outputs x 8 slab-groups per workgroup (coalesced 128-byte
// rows, partial sums in double), deterministic
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float *__restrict__ slabs,
                                                                   float *__restrict__ dw, int nslab, int Cout,
                                                                   int CinP, int Cin_real) {
    __shared__ double red[256];
    const int total = Cout * 9 * CinP;
    const double sum = block_colsum32(slabs, nslab, (size_t)total, blockIdx.x * 32, total, red);
    const int idx = blockIdx.x * 32 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) != 0 || idx >= total) return;
    const int ci = idx % CinP;
    const int tap = (idx / CinP) % 9;
    const int co = idx / (CinP * 9);
    if (ci < Cin_real) dw[((size_t)co * Cin_real + ci) * 9 + tap] = (float)sum;
}

__global__ void pack_w3x3_kernel(const float *__restrict__ w, float *__restrict__ wf,
                                 float *__restrict__ wd, int Cout, int Cin_real, int Cin) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // over [Cout][9][Cin]
    if (idx >= Cout * 9 * Cin) return;
    const int ci = idx % Cin;
    const int tap = (idx / Cin) % 9;
    const int co = idx / (Cin * 9);
    const float v = ci < Cin_real ? w[((size_t)co * Cin_real + ci) * 9 + tap] : 0.f;
    wf[idx] = v;
    if (wd) wd[((size_t)ci * 9 + (8 - tap)) * Cout + co] = v;
}

template <int KC, int BN, int TW>
static int launch_fwd(const float *x, const float *wpk, const float *bias, const float *addend,
                      const float *addend_mask, const float *in_scale, const float *in_shift, float *y, float *stats,
                      const float *stat_aux, const float *stat_mean, const float *stat_invstd, const float *stat_mask,
                      int N, int H, int W, int Cin, int Cout, int relu, int mask_bits, hipStream_t st) {
    // (a variant that walks several patches per workgroup and prefetches the next halo patch into registers under
    //  the current patch's MFMAs was measured 6-9 % SLOWER at every stage on MI355X -- 252 VGPRs, no gain over the
    //  overlap two resident workgroups per CU already give -- and was removed; see DESIGN.md "Tried and rejected")
    constexpr int TH = 256 / TW;
    const int tilesW = cdiv(W, TW), tilesH = cdiv(H, TH);
    dim3 grid((unsigned)(N * tilesH * tilesW), (unsigned)(Cout / BN));
    hipLaunchKernelGGL((conv3x3_fwd_kernel<KC, BN, TW>), grid, dim3(256), 0, st, x, wpk, bias, addend, addend_mask,
                       in_scale, in_shift, y, stats, stat_aux, stat_mean, stat_invstd, stat_mask, H, W, Cin, Cout, tilesW,
                       tilesH, relu, mask_bits);
    return check_launch("conv3x3_fwd");
}

static int wgrad_splits(int N, int H, int W, int Cin, int Cout, int *TWo) {
    const int TW = W >= 32 ? 32 : 16, TH = 256 / TW;
    const int ntiles = N * cdiv(H, TH) * cdiv(W, TW);
    const int blocks = (Cout / 32) * cdiv(Cin, 32);
    int nsplit = 1024 / blocks;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > ntiles) nsplit = ntiles;
    if (TWo) *TWo = TW;
    return nsplit;
}

}  // namespace adyolo

using namespace adyolo;

extern "C" int adyolo_pack_w3x3(const float *w, float *wpk_fwd, float *wpk_dgrad, int Cout, int Cin_real,
                                int Cin, void *stream) {
    ADYOLO_REQUIRE(w && wpk_fwd && Cout > 0 && Cin_real > 0 && Cin >= Cin_real, ADYOLO_EINVAL,
                   "pack_w3x3: bad arguments");
    const int total = Cout * 9 * Cin;
    hipLaunchKernelGGL(pack_w3x3_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, wpk_fwd,
                       wpk_dgrad, Cout, Cin_real, Cin);
    return check_launch("pack_w3x3");
}

extern "C" int adyolo_conv3x3_tiles(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return ADYOLO_EINVAL;
    const int TW = W >= 32 ? 32 : 16, TH = 256 / TW;
    return N * cdiv(H, TH) * cdiv(W, TW);
}

extern "C" int adyolo_conv3x3_fwd(const float *x, const float *wpk, const float *bias, const float *addend,
                                  const float *addend_mask, const float *in_scale, const float *in_shift, float *y,
                                  float *stats, const float *stat_aux, const float *stat_mean,
                                  const float *stat_invstd, const float *stat_mask, int N, int H, int W, int Cin,
                                  int Cout, int relu, int mask_bits, void *stream) {
    ADYOLO_REQUIRE(x && wpk && y && N > 0 && H > 0 && W > 0, ADYOLO_EINVAL, "conv3x3_fwd: bad arguments");
    ADYOLO_REQUIRE(!(mask_bits & ~3) && (!mask_bits || ((long)H * W * (Cout / 4)) % 64 == 0), ADYOLO_ENOSUP,
                   "conv3x3_fwd: mask bits need H*W*Cout/4 %% 64 == 0");
    ADYOLO_REQUIRE((Cin == 8 || Cin % 32 == 0) && Cout % 32 == 0, ADYOLO_ENOSUP,
                   "conv3x3_fwd: Cin=%d must be 8 or a multiple of 32, Cout=%d a multiple of 32", Cin, Cout);
    ADYOLO_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (!addend_mask || addend), ADYOLO_EINVAL,
                   "conv3x3_fwd: in_scale/in_shift come together; addend_mask needs addend");
    ADYOLO_REQUIRE(!stat_aux || (stats && stat_mean && stat_invstd), ADYOLO_EINVAL,
                   "conv3x3_fwd: stat_aux needs stats, stat_mean and stat_invstd");
    ADYOLO_REQUIRE(!stat_mask || stats, ADYOLO_EINVAL, "conv3x3_fwd: stat_mask needs stats");
    hipStream_t st = as_stream(stream);
    const bool wide = W >= 32;
#define ADYOLO_FWD(KC_, BN_, TW_) \
    launch_fwd<KC_, BN_, TW_>(x, wpk, bias, addend, addend_mask, in_scale, in_shift, y, stats, stat_aux, stat_mean, \
                              stat_invstd, stat_mask, N, H, W, Cin, Cout, relu, mask_bits, st)
    if (Cin == 8 && Cout == 32 && wide && !addend && !in_scale && !stat_aux && !stat_mask) {      // the stem
        const int tilesW = cdiv(W, 32), tilesH = cdiv(H, 8);
        hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)(N * tilesH * tilesW)), dim3(256), 0, st, x, wpk, bias, y, stats, H,
                           W, tilesW, tilesH, relu);
        return check_launch("stem_fwd");
    }
    if (Cin == 8) return wide ? ADYOLO_FWD(8, 32, 32) : ADYOLO_FWD(8, 32, 16);
    if (Cout % 64 == 0) return wide ? ADYOLO_FWD(32, 64, 32) : ADYOLO_FWD(32, 64, 16);
    return wide ? ADYOLO_FWD(32, 32, 32) : ADYOLO_FWD(32, 32, 16);
#undef ADYOLO_FWD
}

extern "C" int adyolo_conv3x3_wgrad_slabs(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cout % 32) return ADYOLO_EINVAL;
    return wgrad_splits(N, H, W, Cin, Cout, nullptr);
}

extern "C" int adyolo_conv3x3_wgrad(const float *x, const float *dy, const float *in_scale, const float *in_shift,
                                    float *slabs, float *dw, int N, int H, int W, int Cin, int Cin_real, int Cout,
                                    void *stream) {
    ADYOLO_REQUIRE(x && dy && slabs && dw && N > 0 && H > 0 && W > 0, ADYOLO_EINVAL, "conv3x3_wgrad: bad arguments");
    ADYOLO_REQUIRE((Cin == 8 || Cin % 32 == 0) && Cout % 32 == 0 && Cin_real <= Cin, ADYOLO_ENOSUP,
                   "conv3x3_wgrad: unsupported channels Cin=%d Cout=%d", Cin, Cout);
    hipStream_t st = as_stream(stream);
    int TW;
    const int nsplit = wgrad_splits(N, H, W, Cin, Cout, &TW);
    if (Cin == 8 && Cout == 32 && !in_scale && (long)N * H < (1L << 31)) {       // the stem: operands straight from memory
        const int nblk = nsplit < 512 ? nsplit : 512;       // (the slab workspace holds nsplit x 32 x 9 x 32 floats)
        hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nblk), dim3(256), 0, st, x, dy, slabs, H, W, N * H);
        int rcs = check_launch("stem_wgrad");
        if (rcs) return rcs;
        hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(cdiv(32 * 72, 32)), dim3(256), 0, st, slabs, dw, nblk, 32, 8,
                           Cin_real);
        return check_launch("stem_wgrad_reduce");
    }
    const int TH = 256 / TW;
    const int tilesW = cdiv(W, TW), tilesH = cdiv(H, TH);
    const int ntiles = N * tilesW * tilesH;
    const int cinBlocks = cdiv(Cin, 32);
    dim3 grid((unsigned)nsplit, (unsigned)((Cout / 32) * cinBlocks));
    if (TW == 32)
        hipLaunchKernelGGL((conv3x3_wgrad_kernel<32>), grid, dim3(256), 0, st, x, dy, in_scale, in_shift, slabs, H, W,
                           Cin, Cout, tilesW, tilesH, ntiles, nsplit, cinBlocks);
    else
        hipLaunchKernelGGL((conv3x3_wgrad_kernel<16>), grid, dim3(256), 0, st, x, dy, in_scale, in_shift, slabs, H, W,
                           Cin, Cout, tilesW, tilesH, ntiles, nsplit, cinBlocks);
    int rc = check_launch("conv3x3_wgrad");
    if (rc) return rc;
    const int CinP = cinBlocks * 32;
    const int total = Cout * 9 * CinP;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(cdiv(total, 32)), dim3(256), 0, st, slabs, dw, nsplit,
                       Cout, CinP, Cin_real);
    return check_launch("conv3x3_wgrad_reduce");
}
