// K2w4w: weight gradient of the 3x3 convolution (stride 1, pad 1, channels-last) in the Winograd F(4x4,3x3) domain on the exact-fp32
// MFMA (round 5).  Replaces the backward-weights half of nn.Conv2d at /root/reference/src/models/backbones/resnet.py:16,18, like
// wino.hip's wino_wgrad_kernel (F(2x2) domain: 16 multiplies per 2x2 tile and channel pair = 16/36 of the direct form's); here
//     dw = G^T [ sum over 4x4 output tiles of (B^T d B) (.) (A e A^T) ] G      d: 6x6 input tile, e: 4x4 tile of dy, A = (A^T)^T
// with the matrices of csrc/wino4.hip (interpolation points 0, +-3/4, +-3/2, infinity; every constant of B^T and A dyadic): 36
// multiplies per 16 outputs = 9/36 of the direct form's matrix FLOPs, 1.78x fewer MFMAs than wino_wgrad_kernel.  Error against a
// float64 weight gradient: 1.1-2.6e-6 of its absmax with fp32 accumulation in MFMA order (tools/wino4w/numerics.py; F(2x2): 0.7-2.0e-6).
// The 36 positions (xi, nu) are 36 GEMMs  dU[pos][ci][co] = sum_tiles V[pos][tile][ci] E[pos][tile][co]  with the contraction over TILES.
//
// One workgroup (4 waves, one per SIMD, 256 AccVGPR + 256 VGPR each: the register budget of wino4_fwd_kernel) per CU owns a
// (32 ci x 64 co) block of all 36 positions -- wave w one column nu of the 6 x 6 grid and half of another, as in wino4.hip --
// and walks pairs of RUNS (a run = four tiles side by side = 16 output columns; with W = 16 the two runs of a pair come from two
// samples, with W = 32 from one row) top to bottom, one tile row = 8 tiles = four MFMA k-steps per step (72 MFMAs per wave).
// Both transforms are separable and split like the forward kernel's input transform:
//   * W direction while staging: a thread owns (new row, run, tile, channel quad): six 16-byte loads (x; four for dy) fetch the
//     tile's pixels with the quad's four channels in one register quad (a wave-instruction moves 1 KB through the texture
//     addresser instead of the 256 B of the first version's dword loads, which kept the addresser busy for 3 200 of a step's
//     9 200 cycles: profiles/r05_w4w_timing.txt), the producer's BatchNorm affine and the transform run on <4 x float> = the four
//     channels, and the TRANSPOSE the GEMM side needs -- C[nu][row slot][channel][8 tiles], the tile index along a lane's
//     registers, so that a ds_read_b128 yields the operand of four k-steps -- happens in the LDS stores: 24 ds_write_b32 per
//     item (a ds_write_b32 costs 4 cycles of the store path, a ds_write_b128 13).  Channel c of a row lives in 32-byte slot
//     pi(c) = (c & ~3) | ((c + (c >> 2)) & 3): store k of the eight quads of a lane group then falls into four different
//     bank octets (2-way conflicts, which cost a ds_write_b32 nothing) instead of one; the reads keep their conflict-free
//     16-lane groups, pi permutes inside aligned groups of four channels.  The two tile quads of a channel are swapped when
//     (channel >> 3) is odd (wino.hip).  dy likewise (four pixels -> D[nu][row][channel][8 tiles], 4 -> 6 points with A).
//   * H direction in the GEMM waves: six (x) / four (dy) ds_read_b128 per column give the six A / B operand quads of the column.
// x rows live in a ten-slot ring (the six rows of a step's window + the four new rows of the next step, stored while this step
// computes), dy rows in two buffers: 60 + 2 x 48 KB of LDS, one barrier per step.  Slabs of dU (one per workgroup) are
// summed in a fixed order and G^T . G applied by one small kernel (deterministic).
#include "wino4_common.hpp"

namespace adyolo {
namespace w4 {

#ifndef W4W_WHATIF
#define W4W_WHATIF 0      // timing-only builds (results invalid): bit 0 no MFMAs, bit 1 no staging (loads, transforms, stores)
#endif
#ifndef W4W_LD_AUX
#define W4W_LD_AUX 0       // cache-policy bits of the x / dy loads (2 = non-temporal; A/B switch)
#endif
#ifndef W4W_TIMING
#define W4W_TIMING 0      // 1: wave 0 of one workgroup writes s_memtime stamps of the first steps of its first item to a buffer set
                          // with adyolo_w4w_timing_buffer (results stay valid; tools/wino4w/timing.py)
#endif

// A (6 x 4) along one direction on the four tiles of a run: t = A v
__device__ __forceinline__ void a6v(const f32x4 (&v)[4], f32x4 (&t)[6]) {
    auto fm = [](float k, f32x4 a, f32x4 b) { return pkfma4(k, a, b); };
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 ea = fm(A2, v[2], v[0]), oa = fm(A3, v[3], fm(PA, v[1], z));
    const f32x4 eb = fm(B2, v[2], v[0]), ob = fm(B3, v[3], fm(PB, v[1], z));
    t[0] = v[0];
    t[1] = ea + oa;
    t[2] = fm(-1.f, oa, ea);
    t[3] = eb + ob;
    t[4] = fm(-1.f, ob, eb);
    t[5] = v[3];
}
// half of it: rows xi = 0, 1, 2 (K = 3/4) or 5, 3, 4 (K = 3/2), in that order; z = v[0] or v[3] (read through a wave-uniform address)
__device__ __forceinline__ void a3v(const f32x4 (&v)[4], const f32x4 &zrow, f32x4 &t0, f32x4 &t1, f32x4 &t2, float K1, float K2,
                                    float K3) {
    auto fm = [](float k, f32x4 a, f32x4 b) { return pkfma4(k, a, b); };
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 e = fm(K2, v[2], v[0]), o = fm(K3, v[3], fm(K1, v[1], z));
    t0 = zrow;
    t1 = e + o;
    t2 = fm(-1.f, o, e);
}

#if W4W_TIMING
__device__ unsigned long long *g_w4w_stamps;
#endif

constexpr int XROWB = 1024;                 // bytes of an x row slot: 32 channels x 8 tiles
constexpr int XSLOTS = 10;               // the six rows of a step's window + the four new rows of the next step
constexpr int XNU = XSLOTS * XROWB;         // per nu plane
constexpr int XBYTES = 6 * XNU;             // 61 440
// LDS image: the two dy buffers FIRST, the x ring behind them (round 6).  Every dy store address is lane part + buffer + row +
// plane immediate; with the ring in front the 61 440 of XBYTES did not fit the 16-bit offset field beside the plane immediates
// and cost the step ~10 vector adds.  Behind the dy buffers the ring's start folds into values that are added anyway (the
// per-step slot offset of the stores, the per-item bases of the reads).
// dy rows: 32 NB channels x 8 tiles; NB = 2 (Cout % 64 == 0) or 1 (32-channel blocks: stage 1 -- half the MFMAs per staged byte)
template <int NB>
struct DCfg {
    static constexpr int DROWB = 1024 * NB;
    static constexpr int DNU = 4 * DROWB;
    static constexpr int DBUF = 6 * DNU;    // 49 152 | 24 576
};

template <bool AFF, int NB>
__global__ __launch_bounds__(256, 1) void wino4_wgrad_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ in_scale,
    const float *__restrict__ in_shift, float *__restrict__ slabs, int N, int H, int W, int Cin, int Cout, int runsW, int npairs,
    int nseg, int seg_steps, int nitems, int nsplit, int ciBlocks, int nblk) {
    constexpr int DROWB = DCfg<NB>::DROWB, DNU = DCfg<NB>::DNU, DBUF = DCfg<NB>::DBUF;
    constexpr int XOFF = 2 * DBUF;          // bytes: start of the x ring
    __shared__ __attribute__((aligned(16))) float lds[(XBYTES + 2 * DBUF) / 4 + (AFF ? 64 : 0)];
    char *ldsb = reinterpret_cast<char *>(lds);
    constexpr int AFFOFF = XBYTES + 2 * DBUF;   // bytes: the producer's BatchNorm scale | shift of the block's 32 input channels

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
#if W4W_TIMING
    unsigned long long *stamps = g_w4w_stamps;
    int tstep = 0;
    auto tstamp = [&](int k) {
        if (blockIdx.x == 8 && tid == 0 && tstep < 12) stamps[tstep * 16 + k] = __builtin_amdgcn_s_memtime();
    };
#else
    auto tstamp = [&](int) {};
#endif
    // workgroup -> (split, channel block): the 32 workgroups of an XCD are all channel blocks of 32 / nblk splits, so that one
    // XCD's L2 sees the x / dy rows of its splits once for every channel block that needs them
    int split, blk;
    if (32 % nblk == 0 && (int)gridDim.x == 8 * 32) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        blk = j % nblk;
        split = xcd * (32 / nblk) + j / nblk;
    } else {
        blk = blockIdx.x % nblk;
        split = blockIdx.x / nblk;
    }
    if (split >= nsplit) return;
    const int cbk = blk / ciBlocks, ibk = blk - cbk * ciBlocks;
    const int co0 = cbk * (32 * NB), c0 = ibk * 32;

    // ---- GEMM side: positions of the wave (wino4.hip): full column nuF (xi = 0..5 -> acc 0..5), half column nuH (acc 6..8 =
    // xi 0, 1, 2 for hh = 0, xi 5, 3, 4 for hh = 1)
    const int nuF = wave == 0 ? 0 : wave == 1 ? 2 : wave == 2 ? 3 : 5;
    const int nuH = wave < 2 ? 1 : 4;
    const int hh = wave & 1;
    const float K2x = hh ? A2 : B2, KPx = hh ? PB : PA;                       // x half column (bt3v)
    const float K1d = hh ? PB : PA, K2d = hh ? B2 : A2, K3d = hh ? B3 : A3;   // dy half column (a3v)
    const int lsw = (lh ^ ((li >> 3) & 1)) << 4;                              // tile quad of the lane's half, swizzled
    const int lpi = (li & ~3) | ((li + (li >> 2)) & 3);                        // slot of the lane's channel inside its row (see above)
    const int lx = lpi * 32 + lsw;                                            // lane part of an x read / a dy read inside a 32-channel block

    f32x16 acc[9][NB];
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][cb][r] = 0.f;

    // ---- staging roles.  A wave stages ONE run (the two runs of a pair may belong to different samples) and two rows:
    // row (wave >> 1) * 2 + lh of the step's four (x: new rows; dy: rows of the tile row), tile st of the run, channel quad sq
    // (dy: of each 32-channel block)
    const int run = wave & 1;
    const int sxr = (wave >> 1) * 2 + lh, sdr = sxr;
    const int st = li >> 3, sq = li & 7;
    // (the affine lives in LDS and is read where it is applied: held in registers across the unrolled step loop its eight
    //  registers were the ones that spilled, and a scratch reload waits for vmcnt(0) -- the staging loads in flight; round 6)
    if (AFF) {
        if (tid < 32) lds[AFFOFF / 4 + tid] = in_scale[c0 + tid];
        else if (tid < 64) lds[AFFOFF / 4 + tid] = in_shift[c0 + tid - 32];
    }
    const int xrowb = W * Cin * 4, drowb = W * Cout * 4, xpixb = Cin * 4, dpixb = Cout * 4;
    // whole-tensor buffer descriptors (the host checks that both tensors stay below 2 GiB); image-border pixels / rows are
    // redirected out of range (0x80000000 exceeds num_records: the load returns 0)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)((size_t)N * H * W * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dy), 0, (int)((size_t)N * H * W * Cout * 4), 0x00020000);
    // lane part of the four dword stores of a transformed value (channel 4 sq + k -> slot 4 sq + ((k + sq) & 3)); x and dy alike
    int wl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wl[k] = (4 * sq + ((k + sq) & 3)) * 32 + ((run ^ ((sq >> 1) & 1)) << 4) + st * 4;

    for (int item = split; item < nitems; item += nsplit) {
        const int seg = item % nseg;
        const int pair = item / nseg;
        // the wave's run of the pair: run index q over (sample, 16-column run inside a row); an odd total leaves the last pair half empty
        const int q = 2 * pair + run;
        // Narrow maps (runsW < 0: W = 4 or 8, -runsW tiles per image row -- the ResNet-Conformer's middle stages): a run is made
        // of 4 / tW SAMPLES side by side, the thread's tile st belongs to sample q (4 / tW) + st / tW, tile column st % tW; every
        // tile then touches the image's left and / or right border
        const int tW = runsW < 0 ? -runsW : 0;
        bool qok, offL, offR;
        int n, col0;                                                          // sample, first always-in-image column of the thread's tile
        if (tW) {
            const int sh = tW == 1 ? 0 : 1;
            n = q * (4 >> sh) + (st >> sh);
            const int tcol = st & (tW - 1);
            qok = n < N;
            n = qok ? n : 0;
            col0 = 4 * tcol;
            offL = tcol == 0;
            offR = tcol == tW - 1;
        } else {
            qok = q < N * runsW;
            n = qok ? q / runsW : 0;
            const int rw = qok ? q - (q / runsW) * runsW : 0;
            col0 = rw * 16 + 4 * st;
            // only column -1 (pixel 0 of tile 0) and column 16 (pixel 5 of tile 3) of a run can leave the image sideways
            offL = rw == 0 && st == 0;
            offR = rw == runsW - 1 && st == 3;
        }
        const int t0 = seg * seg_steps;                                       // first tile row of the segment
        const int nsteps = min(seg_steps, H / 4 - t0);
        // byte offset of the item's first always-in-image pixel (column col0) in row 0 of the sample, at the thread's quad
        const unsigned xbase = (unsigned)(((size_t)n * H * W + (size_t)col0) * Cin + c0 + 4 * sq) * 4u;
        const unsigned dbase = (unsigned)(((size_t)n * H * W + (size_t)col0) * Cout + co0 + 4 * sq) * 4u;

        // x pixels of image row gy (any, also -1 / H), columns 16 rw + 4 st - 1 .. + 4, of the thread's channel quad
        f32x4 xpx[6];
        // (pixels i0 .. i1 - 1: the step loop issues them two at a time, between different MFMA groups -- a burst of 16-byte loads
        //  from all four waves queues up behind the 64 B / clock of the L1 and stalls the issuing wave)
        auto x_load = [&](int gy, int i0 = 0, int i1 = 6) {
            const bool rowok = qok && gy >= 0 && gy < H;
            const int vrow = rowok ? (int)(xbase + (unsigned)gy * (unsigned)xrowb) : (int)0x80000000;
            // pixels 1 .. 4 through the scalar offset; the two outer ones have offsets of their own (the scalar offset is not
            // part of the range check)
            const int vL = (rowok && !offL) ? vrow - xpixb : (int)0x80000000;
            const int vR = (rowok && !offR) ? vrow : (int)0x80000000;
#pragma unroll
            for (int i = i0; i < i1; ++i) {
                if (i == 0)
                    xpx[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vL, 0, W4W_LD_AUX));
                else if (i == 5)
                    xpx[5] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vR, 4 * xpixb, W4W_LD_AUX));
                else
                    xpx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vrow, (i - 1) * xpixb, W4W_LD_AUX));
            }
        };
        // affine + W transform, in place: xpx[nu] afterwards
        auto x_prep = [&](int gy) {
            if (AFF) {
                // x' = scale x + shift inside the image, 0 outside (the loads returned 0 there)
                const bool rowok = qok && gy >= 0 && gy < H;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                const int sqo = (lane & 7) * 16;
                const f32x4 xsc = *reinterpret_cast<const f32x4 *>(ldsb + AFFOFF + sqo);
                const f32x4 xsh = *reinterpret_cast<const f32x4 *>(ldsb + AFFOFF + 128 + sqo);
                const f32x4 shv = rowok ? xsh : zero;
                xpx[0] = pkfma4v(xpx[0], xsc, offL ? zero : shv);
#pragma unroll
                for (int i = 1; i < 5; ++i) xpx[i] = pkfma4v(xpx[i], xsc, shv);
                xpx[5] = pkfma4v(xpx[5], xsc, offR ? zero : shv);
            }
            bt6v(xpx);
        };
        // the four transposing stores of plane nu (wx[k]: lane part + row slot of the thread's row)
        int wx[4];
        // (planes pairwise, the two stores of a lane address next to each other: they merge into one ds_write2st64_b32 --
        //  3 source dwords, 6 cycles of the store path instead of 2 x 4)
        auto x_wr2 = [&](int nu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *reinterpret_cast<float *>(ldsb + wx[k] + nu * XNU) = xpx[nu][k];
                *reinterpret_cast<float *>(ldsb + wx[k] + (nu + 1) * XNU) = xpx[nu + 1][k];
            }
        };
        f32x4 dpx[NB][4], de, dq;
        // dy row 4 trow + sdr (always inside the image).  The lane part of the address (sample, column, channel quad, row sdr of
        // the four) is fixed for the item; the tile row travels in the SCALAR offset of the load -- no vector arithmetic per request
        // (the step loop used to rebuild the 64-bit product four times per step: round 6).  Lanes of a missing run keep the
        // out-of-range marker in the vector offset, which is the part the range check sees.
        const int dlane = qok ? (int)(dbase + (unsigned)sdr * (unsigned)drowb) : (int)0x80000000;
        auto d_load = [&](int trow, int cb0 = 0, int cb1 = NB, int i0 = 0, int i1 = 4) {
            const int srow = __builtin_amdgcn_readfirstlane(4 * trow * drowb);
#pragma unroll
            for (int cb = cb0; cb < cb1; ++cb)
#pragma unroll
                for (int i = i0; i < i1; ++i)
                    dpx[cb][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, dlane, srow + cb * 128 + i * dpixb, W4W_LD_AUX));
        };
        // 4 -> 6 points (a6v) in three parts, each followed by its stores: planes (0, 1), (2, 3), (4, 5).  Plane 0 / 5 are pixels
        // 0 / 3 themselves; de / dq carry the even / odd sums from one part to the next (12 live registers instead of 24)
        int wd[4];                                                             // lane part + row + buffer of the step's dy stores
        auto d_put2 = [&](int cb, int nu, const f32x4 &v, const f32x4 &v1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *reinterpret_cast<float *>(ldsb + wd[k] + cb * 1024 + nu * DNU) = v[k];
                *reinterpret_cast<float *>(ldsb + wd[k] + cb * 1024 + (nu + 1) * DNU) = v1[k];
            }
        };
        auto d_part = [&](int cb, int part) {
            auto fm = [](float k, f32x4 a_, f32x4 b_) { return pkfma4(k, a_, b_); };
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4(&v)[4] = dpx[cb];
            if (part == 0) {
                de = fm(A2, v[2], v[0]);
                dq = fm(A3, v[3], fm(PA, v[1], z));
                d_put2(cb, 0, v[0], de + dq);
            } else if (part == 1) {
                const f32x4 t2 = fm(-1.f, dq, de);
                de = fm(B2, v[2], v[0]);
                dq = fm(B3, v[3], fm(PB, v[1], z));
                d_put2(cb, 2, t2, de + dq);
            } else {
                d_put2(cb, 4, fm(-1.f, dq, de), v[3]);
            }
        };

        // ---- prologue: the whole window of the first step (rows 4 t0 - 1 .. 4 t0 + 4 -> slots 0 .. 5) and its dy rows
        __syncthreads();                                                       // (the previous item's readers are done)
        x_load(4 * t0 - 1 + sxr);
        d_load(t0);
        x_prep(4 * t0 - 1 + sxr);
#pragma unroll
        for (int k = 0; k < 4; ++k) wx[k] = wl[k] + XOFF + sxr * XROWB;
#pragma unroll
        for (int nu = 0; nu < 6; nu += 2) x_wr2(nu);
        if (sxr < 2) {
            x_load(4 * t0 + 3 + sxr);
            x_prep(4 * t0 + 3 + sxr);
#pragma unroll
            for (int k = 0; k < 4; ++k) wx[k] = wl[k] + XOFF + (4 + sxr) * XROWB;
#pragma unroll
            for (int nu = 0; nu < 6; nu += 2) x_wr2(nu);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) wd[k] = wl[k] + sdr * DROWB;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            d_part(cb, 0);
            d_part(cb, 1);
            d_part(cb, 2);
        }
        // requests of step 1 (its four new x rows: window rows 2 .. 5; its dy rows) -- stored during step 0
        x_load(4 * (t0 + 1) + 1 + sxr);
        d_load(t0 + 1);
        __syncthreads();

        // ---- the step loop.  The wave's operands of a step: x rows of the window for its two columns (a), dy rows of its columns
        // (b: first 32-channel block, then the second); H transforms here, in the GEMM waves.
        f32x4 a[9], b[9];
        f32x4 vF[4], vH[4], zH;
        f32x4 cF[6], cP[4], cZ[3];
        auto d_reads_full = [&](int buf_, int cb) {
            const char *dp = ldsb + buf_ * DBUF + cb * 1024 + lx;
#pragma unroll
            for (int i = 0; i < 4; ++i) vF[i] = *reinterpret_cast<const f32x4 *>(dp + nuF * DNU + i * DROWB);
        };
        auto d_reads_half = [&](int buf_, int cb) {
            const char *dp = ldsb + buf_ * DBUF + cb * 1024 + lx;
#pragma unroll
            for (int i = 0; i < 4; ++i) vH[i] = *reinterpret_cast<const f32x4 *>(dp + nuH * DNU + i * DROWB);
            zH = *reinterpret_cast<const f32x4 *>(dp + nuH * DNU + (hh ? 3 : 0) * DROWB);
        };
        auto d_xform_full = [&]() {
            f32x4 t[6];
            a6v(vF, t);
#pragma unroll
            for (int s = 0; s < 6; ++s) b[s] = t[s];
        };
        auto d_xform_half = [&]() { a3v(vH, zH, b[6], b[7], b[8], K1d, K2d, K3d); };
        // window reads of a step whose window row 0 sits in ring slot ROT (a compile-time constant: the step loop below is unrolled
        // over the five ring positions 0, 4, 8, 2, 6, so every slot offset is an immediate of its ds_read -- with a run-time
        // position the 13 + 18 reads of a step cost 52 vector integer instructions and ~30 scalar ones, each an issue slot of the
        // wave that also issues the MFMAs: round 6): the full column, the half column
        auto x_reads_full = [&](auto ROT_) {
            constexpr int rot_ = decltype(ROT_)::value;
            const char *xp = ldsb + XOFF + lx + nuF * XNU;
#pragma unroll
            for (int i = 0; i < 6; ++i) cF[i] = *reinterpret_cast<const f32x4 *>(xp + ((rot_ + i) % XSLOTS) * XROWB);
        };
        auto x_reads_half = [&](auto ROT_) {
            constexpr int rot_ = decltype(ROT_)::value;
            constexpr int so0 = ((rot_ + 0) % XSLOTS) * XROWB, so1 = ((rot_ + 1) % XSLOTS) * XROWB, so2 = ((rot_ + 2) % XSLOTS) * XROWB,
                          so3 = ((rot_ + 3) % XSLOTS) * XROWB, so4 = ((rot_ + 4) % XSLOTS) * XROWB, so5 = ((rot_ + 5) % XSLOTS) * XROWB;
            const char *xp = ldsb + XOFF + lx + nuH * XNU;
            cP[0] = *reinterpret_cast<const f32x4 *>(xp + so1);
            cP[1] = *reinterpret_cast<const f32x4 *>(xp + so2);
            cP[2] = *reinterpret_cast<const f32x4 *>(xp + so3);
            cP[3] = *reinterpret_cast<const f32x4 *>(xp + so4);
            // rows hh, hh + 2, hh + 4 of the half column (the single term's operands; hh is wave-uniform)
            cZ[0] = *reinterpret_cast<const f32x4 *>(xp + (hh ? so1 : so0));
            cZ[1] = *reinterpret_cast<const f32x4 *>(xp + (hh ? so3 : so2));
            cZ[2] = *reinterpret_cast<const f32x4 *>(xp + (hh ? so5 : so4));
        };
        auto mfma4 = [&](int s, int cb) {
            if (W4W_WHATIF & 1) return;
            if (s < 8) {
                acc[s][cb] = mfma32(a[s][0], b[s][0], acc[s][cb]);
                acc[s][cb] = mfma32(a[s][1], b[s][1], acc[s][cb]);
                acc[s][cb] = mfma32(a[s][2], b[s][2], acc[s][cb]);
                acc[s][cb] = mfma32(a[s][3], b[s][3], acc[s][cb]);
                asm volatile("" : "+a"(acc[s][cb]));
            } else {
                mfma32x4_vgpr(acc[s][cb], make_float4(a[s][0], a[s][1], a[s][2], a[s][3]),
                              make_float4(b[s][0], b[s][1], b[s][2], b[s][3]));
            }
        };
        // The staging of step k + 1 (loads requested one step ago) rides between the MFMA groups of step k: its stores go to the
        // four ring slots OUTSIDE the step's window and to the other dy buffer, so nothing orders them against this step's reads
        // and ONE barrier per step is enough; the LDS store path (64 B / clock / CU: ~1 150 cycles of a step) and the waits for
        // the loads then run under the matrix work instead of in a phase of their own.  Side task t of a step (between MFMA
        // groups t - 1 and t; NB = 2: 18 groups, NB = 1: 9):
        //   NB = 2: 1 the half columns' H transforms | 2 x affine + W transform | 3, 4, 5 x planes (0, 1) (2, 3) (4, 5) | 4, 5, 6 the
        //   requests of step k + 2's x rows, two at a time as their registers retire | 6, 7, 8 dy block 0: 4 -> 6 points and its
        //   planes, pairwise | 8 also: this step's block-1 operands are read | 9, 10 requests of dy block 0 | 10 block 1's half
        //   column | 11, 12, 13 dy block 1 | 14 THE BARRIER (step k + 1's rows are in LDS; every read of this step's is done) |
        //   15 step k + 1's full columns are read (the registers of a[0 .. 5], b[0 .. 5] are free from group 14 on) | 16 their H
        //   transforms | 17 its half columns are read -- the next step starts with its operands in registers
        //   NB = 1: the same in nine groups: 0 x transform | 1, 2, 3 x planes | 3, 4, 5 dy | 6 the barrier and the full-column
        //   reads | 7 their transforms | 8 the half-column reads
        auto top_full = [&](auto ROT_, int buf_) {
            x_reads_full(ROT_);
            d_reads_full(buf_, 0);
        };
        auto top_half = [&](auto ROT_, int buf_) {
            x_reads_half(ROT_);
            d_reads_half(buf_, 0);
        };
        auto top_xform = [&]() {
            bt6v2(cF, a[0], a[1], a[2], a[3], a[4], a[5]);
            d_xform_full();
        };
        top_full(std::integral_constant<int, 0>{}, 0);
        top_half(std::integral_constant<int, 0>{}, 0);
        top_xform();
        // one step with window row 0 in ring slot ROT; the next step's is ROT + 4 (mod 10)
        auto step = [&](auto ROT_, int k) {
            constexpr int rot = decltype(ROT_)::value;
            constexpr std::integral_constant<int, (rot + 4) % XSLOTS> rotn{};
            // (k opaque: the five copies of the step must not share strength-reduced per-lane address chains -- kept in registers
            //  across the whole unrolled loop they spill, and a scratch reload sits in the same vmcnt queue as the staging loads)
            asm volatile("" : "+s"(k));
            const int buf = k & 1;
            const int gyn = 4 * (t0 + k + 1) + 1 + sxr;
            auto side = [&](int t) {
                // (requests of step k + 2, two at a time as their registers retire; rows past the end: clamped / out of range, unused)
                const int gy2 = 4 * (t0 + k + 2) + 1 + sxr, tr2 = min(t0 + k + 2, H / 4 - 1);
                if (t == 1) {
                    bt3v(cP, cZ, a[6], a[7], a[8], K2x, KPx);
                    d_xform_half();
                }
                if (NB == 2) {
                    if ((W4W_WHATIF & 2) && t != 1 && t != 8 && t != 10 && t < 14) return;    // timing only: no staging
                    if (t == 2) x_prep(gyn);
                    if (t >= 3 && t <= 5) x_wr2(2 * (t - 3));
                    if (t >= 4 && t <= 6) x_load(gy2, 2 * (t - 4), 2 * (t - 4) + 2);
                    if (t >= 6 && t <= 8) d_part(0, t - 6);
                    if (t == 8) {
                        d_reads_full(buf, 1);
                        d_reads_half(buf, 1);
                    }
                    if (t == 9) d_load(tr2, 0, 1, 0, 2);
                    if (t == 10) {
                        d_xform_half();
                        d_load(tr2, 0, 1, 2, 4);
                    }
                    if (t >= 11 && t <= 13) d_part(1, t - 11);
                    if (t == 14) __syncthreads();
                    if (t == 15) {
                        top_full(rotn, buf ^ 1);
                        d_load(tr2, 1, 2, 0, 2);
                    }
                    if (t == 16) {
                        top_xform();
                        d_load(tr2, 1, 2, 2, 4);
                    }
                    if (t == 17) top_half(rotn, buf ^ 1);
                } else {
                    if (t == 0) x_prep(gyn);
                    if (t >= 1 && t <= 3) x_wr2(2 * (t - 1));
                    if (t >= 2 && t <= 4) x_load(gy2, 2 * (t - 2), 2 * (t - 2) + 2);
                    if (t >= 3 && t <= 5) d_part(0, t - 3);
                    if (t == 5) d_load(tr2);
                    if (t == 6) {
                        __syncthreads();
                        top_full(rotn, buf ^ 1);
                    }
                    if (t == 7) top_xform();
                    if (t == 8) top_half(rotn, buf ^ 1);
                }
            };
            tstamp(0);
            {
                // slots of the new rows (window rows 6 .. 9 of this step: ring slots base + sxr, sxr = 2 (wave >> 1) + lh -- they wrap
                // around the ring only for base 8 and the waves 2, 3: a wave-uniform correction), the other dy buffer
                constexpr int base = (rot + 6) % XSLOTS;
                const int soff = ((base + 2 >= XSLOTS && wave >= 2) ? (base - XSLOTS) * XROWB : base * XROWB) + XOFF + sxr * XROWB;
#pragma unroll
                for (int k_ = 0; k_ < 4; ++k_) {
                    wx[k_] = wl[k_] + soff;
                    wd[k_] = wl[k_] + (buf ^ 1) * DBUF + sdr * DROWB;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            tstamp(1);
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                side(s);
                __builtin_amdgcn_sched_barrier(0);
                mfma4(s, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            tstamp(2);
            if (NB == 2) {
                d_xform_full();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    side(9 + s);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma4(s, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            tstamp(3);
            tstamp(4);
#if W4W_TIMING
            ++tstep;
#endif
        };
        // the ring positions of consecutive steps: 0, 4, 8, 2, 6, 0, ...
        for (int k = 0; k < nsteps;) {
            step(std::integral_constant<int, 0>{}, k);
            if (++k >= nsteps) break;
            step(std::integral_constant<int, 4>{}, k);
            if (++k >= nsteps) break;
            step(std::integral_constant<int, 8>{}, k);
            if (++k >= nsteps) break;
            step(std::integral_constant<int, 2>{}, k);
            if (++k >= nsteps) break;
            step(std::integral_constant<int, 6>{}, k);
            ++k;
        }
    }

    // ---- one slab per workgroup: [split][36 positions][Cin][Cout]
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int pos = wave * 9 + s;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mfma_row(r, lane);
                slabs[(((size_t)split * 36 + pos) * Cin + c0 + m) * Cout + co0 + cb * 32 + li] = acc[s][cb][r];
            }
    }
}

// Slab sum, then G^T . G -- two small launches.  (The first version did both in one kernel, 36 barrier-separated column sums per
// workgroup: 55 us per call -- latency-bound with 8 slabs, parallelism-bound with 32 x 32 channels; profiles/r05_bench_b64x60s_*.)
// reduce: grid (ceil(pairs / 32), 36): one position's 32 consecutive (ci, co) pairs, slabs summed in the fixed order of
// block_colsum32 (double) -> du [36][Cin][Cout].
__global__ __launch_bounds__(256) void wino4_wgrad_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ du, int nslab,
                                                                 int pairs) {
    __shared__ double red[256];
    const int p = blockIdx.y, total = 36 * pairs;
    const double s = block_colsum32(slabs, nslab, (size_t)total, p * pairs + blockIdx.x * 32, p * pairs + pairs, red);
    const int idx = blockIdx.x * 32 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) == 0 && idx < pairs) du[(size_t)p * pairs + idx] = (float)s;
}
// finish: dw[ky][kx] = sum G[xi][ky] dU[xi][nu] G[nu][kx] (in double), reference layout [Cout][Cin_real][3][3].  Position
// P = 9 w + s: s < 6: (xi = s, nu = nuF(w)); s >= 6: nu = nuH(w), xi = 0, 1, 2 (w even) or 5, 3, 4 (w odd) -- the order of wino4.hip.
__global__ __launch_bounds__(256) void wino4_wgrad_finish_kernel(const float *__restrict__ du, float *__restrict__ dw, int Cin,
                                                                 int Cin_real, int Cout) {
    const int pairs = Cin * Cout;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // over [Cin][Cout], co fastest
    if (idx >= pairs) return;
    const int co = idx % Cout, ci = idx / Cout;
    if (ci >= Cin_real) return;
    double d[6][6];
#pragma unroll
    for (int wv = 0; wv < 4; ++wv)
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int p = wv * 9 + s;
            const int nu = s < 6 ? (wv == 0 ? 0 : wv == 1 ? 2 : wv == 2 ? 3 : 5) : (wv < 2 ? 1 : 4);
            const int xi = s < 6 ? s : (wv & 1) ? (s == 6 ? 5 : s - 4) : s - 6;
            d[xi][nu] = (double)du[(size_t)p * pairs + idx];
        }
    const double a_ = 0.75, b_ = 1.5, a2 = a_ * a_, b2 = b_ * b_;
    const double na = 2.0 * a2 * (a2 - b2), nb = 2.0 * b2 * (b2 - a2);
    const double G[6][3] = {{1.0 / (a2 * b2), 0.0, 0.0}, {1.0 / na, a_ / na, a2 / na}, {1.0 / na, -a_ / na, a2 / na},
                            {1.0 / nb, b_ / nb, b2 / nb}, {1.0 / nb, -b_ / nb, b2 / nb}, {0.0, 0.0, 1.0}};
    float *o = dw + ((size_t)co * Cin_real + ci) * 9;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            double s_ = 0.0;
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) {
                double r_ = 0.0;
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) r_ += d[xi][nu] * G[nu][kx];
                s_ += G[xi][ky] * r_;
            }
            o[ky * 3 + kx] = (float)s_;
        }
}

// work split: items = (pair of runs, segment of tile rows); every workgroup (split) takes items split, split + nsplit, ...
static int wino4_wgrad_geometry(int N, int H, int W, int Cin, int Cout, int *npairs_o, int *nseg_o, int *seg_steps_o, int *nitems_o,
                                int *nblk_o) {
    // (narrow maps, W = 4 / 8: a run is 4 / (W / 4) samples side by side)
    const int nruns = W >= 16 ? N * (W / 16) : cdiv(N, 16 / W), npairs = (nruns + 1) / 2;
    const int nblk = (Cout / (Cout % 64 == 0 ? 64 : 32)) * (Cin / 32);
    int nsplit = 256 / nblk;
    if (nsplit < 1) nsplit = 1;
    const int steps = H / 4;
    // Segments per pair: the count that minimises the longest workgroup's work -- ceil(items / splits) items of seg_steps steps
    // plus a prologue worth about two steps each -- among the counts that leave segments of at least 8 tile rows.  (The first
    // version took the fewest segments that gave every split an item: 3 pairs x 3 segments on 8 splits left one workgroup
    // with two items beside seven with one.)
    int nseg = 1, seg_steps = steps;
    {
        long best = -1;
        const int maxseg = steps >= 16 ? steps / 8 : 1;
        for (int c = 1; c <= maxseg; ++c) {
            const int ss = (steps + c - 1) / c;
            const int ns = (steps + ss - 1) / ss;                  // segments that are not empty
            const long items = (long)npairs * ns;
            const int sp = nsplit < items ? nsplit : (int)items;
            const long cost = ((items + sp - 1) / sp) * (long)(ss + 2);
            if (best < 0 || cost < best) {
                best = cost;
                nseg = ns;
                seg_steps = ss;
            }
        }
    }
    const int nitems = npairs * nseg;
    if (nsplit > nitems) nsplit = nitems;
    if (npairs_o) *npairs_o = npairs;
    if (nseg_o) *nseg_o = nseg;
    if (seg_steps_o) *seg_steps_o = seg_steps;
    if (nitems_o) *nitems_o = nitems;
    if (nblk_o) *nblk_o = nblk;
    return nsplit;
}

}  // namespace w4
}  // namespace adyolo

using namespace adyolo;

#if W4W_TIMING
extern "C" int adyolo_w4w_timing_buffer(void *p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(w4::g_w4w_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

// > 0: the number of slabs the kernel writes for this shape; <= 0: shape not supported (use adyolo_wino_wgrad)
extern "C" int adyolo_wino4_wgrad_slabs(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return ADYOLO_EINVAL;
    if (Cin % 32 || Cout % 32 || (W % 16 && W != 4 && W != 8) || H % 4 || H < 8) return 0;
    if ((size_t)N * H * W * (size_t)(Cin > Cout ? Cin : Cout) * 4 >= ((size_t)1 << 31)) return 0;       // 31-bit byte offsets
    return w4::wino4_wgrad_geometry(N, H, W, Cin, Cout, nullptr, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int adyolo_wino4_wgrad(const float *x, const float *dy, const float *in_scale, const float *in_shift, float *slabs,
                                  float *du, float *dw, int N, int H, int W, int Cin, int Cin_real, int Cout, void *stream) {
    ADYOLO_REQUIRE(x && dy && slabs && du && dw && N > 0 && H > 0 && W > 0, ADYOLO_EINVAL, "wino4_wgrad: bad arguments");
    ADYOLO_REQUIRE(adyolo_wino4_wgrad_slabs(N, H, W, Cin, Cout) > 0 && Cin_real > 0 && Cin_real <= Cin, ADYOLO_ENOSUP,
                   "wino4_wgrad: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d (needs Cin %% 32 == 0, Cout %% 32 == 0, W %% 16 == 0 or W = 4 | 8, "
                   "H %% 4 == 0, tensors below 2 GiB)", N, H, W, Cin, Cout);
    ADYOLO_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), ADYOLO_EINVAL, "wino4_wgrad: in_scale/in_shift come together");
    // (16-byte loads of pixel quads and of the affine's channel quads)
    ADYOLO_REQUIRE(((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(dy) | reinterpret_cast<size_t>(in_scale) |
                     reinterpret_cast<size_t>(in_shift)) & 15) == 0, ADYOLO_ENOSUP, "wino4_wgrad: x, dy, in_scale, in_shift must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    int npairs, nseg, seg_steps, nitems, nblk;
    const int nsplit = w4::wino4_wgrad_geometry(N, H, W, Cin, Cout, &npairs, &nseg, &seg_steps, &nitems, &nblk);
    const unsigned grid = (unsigned)(nsplit * nblk);
#define ADYOLO_W4W(AFF_, NB_)                                                                                          \
    hipLaunchKernelGGL((w4::wino4_wgrad_kernel<AFF_, NB_>), dim3(grid), dim3(256), 0, st, x, dy, in_scale, in_shift, slabs, N, H, \
                       W, Cin, Cout, W >= 16 ? W / 16 : -(W / 4), npairs, nseg, seg_steps, nitems, nsplit, Cin / 32, nblk)
    if (Cout % 64 == 0) {
        if (in_scale) ADYOLO_W4W(true, 2); else ADYOLO_W4W(false, 2);
    } else {
        if (in_scale) ADYOLO_W4W(true, 1); else ADYOLO_W4W(false, 1);
    }
#undef ADYOLO_W4W
    int rc = check_launch("wino4_wgrad");
    if (rc) return rc;
    hipLaunchKernelGGL(w4::wino4_wgrad_reduce_kernel, dim3(cdiv(Cin * Cout, 32), 36), dim3(256), 0, st, slabs, du, nsplit, Cin * Cout);
    rc = check_launch("wino4_wgrad_reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(w4::wino4_wgrad_finish_kernel, dim3(cdiv(Cin * Cout, 256)), dim3(256), 0, st, du, dw, Cin, Cin_real, Cout);
    return check_launch("wino4_wgrad_finish");
}
