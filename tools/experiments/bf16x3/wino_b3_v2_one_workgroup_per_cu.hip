// EXPERIMENT (round 3, not built): wino_b3.hip with a second forward kernel, `wino_fwd_b3v2_kernel` -- 64 tiles x 64 output
// channels per workgroup, ONE workgroup per CU (512 registers per wave: 16 accumulators in AccVGPRs), every B fragment feeding
// two MFMAs, 18 x 18-pixel patches, two-round epilogue.  Bit-equal to the two-workgroup form in all tests/test_gpu_bf16x3.py
// cases; NOT faster: stage 4 (256 -> 256) 2.20-2.35 ms against 2.28, stages 2-3 slower (their prologue / epilogue sit exposed
// with one resident workgroup).  What-if builds (B3V2_WHATIF, stage 4, ms): as built 2.32; no staging of the next chunk 1.80;
// no B loads 1.85; no operand split 2.10; no LDS reads 2.24; two of six MFMAs 1.60; none of staging / B / split / LDS reads
// 1.33 (= prologue + MFMAs + epilogue); nothing but the skeleton 0.69.  Ring depth of B (2 or 4 steps), staging distance (1-3
// steps) and an explicit MFMA : VALU deal changed nothing.  See DESIGN.md section 5, "bf16x3, round 3".
// K2w-b3: the Winograd F(2x2, 3x3) forward / data-gradient convolution of wino.hip with its 16 GEMMs on the bf16 matrix pipe
// (v_mfma_f32_32x32x16_bf16, 14.7 x the rate of v_mfma_f32_32x32x2_f32 on this part) WITHOUT giving up fp32 operands: every
// fp32 operand value is split exactly into three bf16 terms, x = hi + mid + lo (8 + 8 + 8 mantissa bits, round-to-nearest at
// every step, the last remainder is exact), and a product is formed from six bf16 MFMAs with fp32 accumulation,
//     a b ~ lo hi' + hi lo' + mid mid' + mid hi' + hi mid' + hi hi'          (smallest terms first)
// -- the three dropped terms (mid lo', lo mid', lo lo') are below 2^-24 |a b|, the size of the fp32 MFMA's own rounding
// (tools/micro/split_bf16_gemm.hip: max error / sum |a b| 2.2e-7 against 2.8e-7 for the fp32 instruction).  Six K = 16
// instructions (6 x 8 passes) stand for eight K = 2 fp32 instructions (8 x 16 passes): 2.67 x less matrix time per channel.
// OPT-IN (ADYOLO_MATH=bf16x3, ops.py); the default path and the headline bench stay on the exact-fp32 instruction.
//
// Same workgroup shape, LDS patch image, staging, XCD mapping and epilogue as wino_fwd_kernel<NT, false> (wino_common.hpp).
// Differences: a K group is 16 channels (two 8-channel LDS reads per lane: k slot i of lane (tile, h) is channel
// 16 G + 8 (i >> 2) + 4 h + (i & 3), so the LDS addresses are those of the fp32 kernel); the filter U = G g G^T is packed
// pre-split, [16 pos][Cout/32][Cin/16][3 terms][64 lanes][8 bf16] (1.5 x the bytes of the fp32 pack), the activations are
// split in registers right after the input transform (4.5 VALU instructions per value, 44 per 12 MFMAs).
#include "wino_common.hpp"

namespace adyolo {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma_b16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// x = hi + mid + lo, two values at a time (v_cvt_pk_bf16_f32 rounds to nearest even).  The subtractions are single
// v_sub_f32 on purpose: left to itself the compiler pairs them into v_pk_add_f32, which costs ~13 issue cycles beside MFMAs
// against 2 x 4 for the two scalar instructions (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
__device__ __forceinline__ float sub1(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float add1(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float4 f4_add1(float4 a, float4 b) { return make_float4(add1(a.x, b.x), add1(a.y, b.y), add1(a.z, b.z), add1(a.w, b.w)); }
__device__ __forceinline__ float4 f4_sub1(float4 a, float4 b) { return make_float4(sub1(a.x, b.x), sub1(a.y, b.y), sub1(a.z, b.z), sub1(a.w, b.w)); }
__device__ __forceinline__ void split3_pair(float x0, float x1, bf16x2 &h, bf16x2 &m, bf16x2 &l) {
    const f32x2 a = {x0, x1};
    h = __builtin_convertvector(a, bf16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f32x2 r1 = {sub1(x0, hf[0]), sub1(x1, hf[1])};
    m = __builtin_convertvector(r1, bf16x2);
    const f32x2 mf = __builtin_convertvector(m, f32x2);
    const f32x2 r2 = {sub1(r1[0], mf[0]), sub1(r1[1], mf[1])};
    l = __builtin_convertvector(r2, bf16x2);
}
__device__ __forceinline__ void split3(float4 p, float4 q, bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {
    bf16x2 h, m, l;
    split3_pair(p.x, p.y, h, m, l);
    hi[0] = h[0]; hi[1] = h[1]; mid[0] = m[0]; mid[1] = m[1]; lo[0] = l[0]; lo[1] = l[1];
    split3_pair(p.z, p.w, h, m, l);
    hi[2] = h[0]; hi[3] = h[1]; mid[2] = m[0]; mid[3] = m[1]; lo[2] = l[0]; lo[3] = l[1];
    split3_pair(q.x, q.y, h, m, l);
    hi[4] = h[0]; hi[5] = h[1]; mid[4] = m[0]; mid[5] = m[1]; lo[4] = l[0]; lo[5] = l[1];
    split3_pair(q.z, q.w, h, m, l);
    hi[6] = h[0]; hi[7] = h[1]; mid[6] = m[0]; mid[7] = m[1]; lo[6] = l[0]; lo[7] = l[1];
}

__device__ __forceinline__ bf16x8 as_b8(float4 v) {
    union { float4 f; bf16x8 b; } c;
    c.f = v;
    return c.b;
}

#ifndef B3_DEAL
#define B3_DEAL 0
#endif
#ifndef B3_WHATIF
#define B3_WHATIF 0      // timing-only builds (results invalid): bit 0 all B loads hit one KB, 1 no operand split, 2 two of six MFMAs,
                         // 3 every other MFMA of a unit on a second accumulator, 4 no B loads in the loop
#endif
template <int NT>
__global__ __launch_bounds__(256, 2) void wino_fwd_b3_kernel(
    const float *__restrict__ x, const float *__restrict__ u, const float *__restrict__ bias,
    const float *__restrict__ addend, const float *__restrict__ addend_mask, const float *__restrict__ in_scale,
    const float *__restrict__ in_shift, float *__restrict__ y, float *__restrict__ stats,
    const float *__restrict__ stat_aux, const float *__restrict__ stat_mean, const float *__restrict__ stat_invstd,
    const float *__restrict__ stat_mask, int H, int W, int Cin, int Cout, int tilesW, int tilesH, int nsp, int ncb,
    int xcd_div, int relu, int mask_bits) {
    using Cfg = WinoCfg<NT, false>;
    constexpr int CB = Cfg::CB;
    constexpr int AFFC = WMAXC;
    __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) float aff[2 * AFFC];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    for (int c = tid; c < Cin; c += 256) {
        aff[c] = in_scale ? in_scale[c] : 1.f;
        aff[AFFC + c] = in_scale ? in_shift[c] : 0.f;
    }
    int sp, cb;                                           // block -> (patch, channel block): see wino_fwd_kernel
    if (xcd_div > 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        cb = xcd % ncb;
        sp = j * xcd_div + xcd / ncb;
    } else {
        cb = blockIdx.x % ncb;
        sp = blockIdx.x / ncb;
    }
    if (sp >= nsp) return;
    sp = nsp - 1 - sp;
    int t = sp;
    const int tw = t % tilesW;
    t /= tilesW;
    const int th = t % tilesH;
    const int n = t / tilesH;
    const int co0 = cb * CB;
    const int ty0 = th * 8, tx0 = tw * 16;

    f32x16 acc[4][NT];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[v][nt][r] = 0.f;

    const int ia = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int ib = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = wave == 1 ? 1.f : -1.f;
    const int tr = li >> 3, tc = li & 7;
    const int offa = (((2 * tr + ia) * 2) * WHALF + tc) * WAS + lh * 4;
    const int offb = (((2 * tr + ib) * 2) * WHALF + tc) * WAS + lh * 4;
    constexpr int J1 = WHALF * WAS, J2 = WAS;

    const int sq = tid & 7, spix0 = tid >> 3;
    constexpr int APT = 6;
    const int nchunks = Cin / WKC, nG = Cin / 16;
    const size_t ustride_pos = (size_t)(Cout / 32) * nG * 768;               // floats per transform position
    const char *ubase = reinterpret_cast<const char *>(u + ((size_t)(wave * 4) * (Cout / 32) + (size_t)cb * NT) * nG * 768);
    const unsigned ulane = lane * 16u;

    float4 pv;
    const char *xsamp = reinterpret_cast<const char *>(x + (size_t)n * H * W * Cin);
    auto opaque_zero = [&]() {
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        return z;
    };
    // staging is dealt out one 16-byte piece per thread and step: piece k (pixel spix0 + 32 k, channels 4 sq ..) of the next
    // chunk is requested in step k and written to the other LDS buffer in step k + 1 (one float4 in flight instead of three)
    auto load_piece = [&](int k, int c0) {
        const int pix = spix0 + k * 32 + opaque_zero();
        const int hy = (pix * 3641) >> 16, hx = pix - hy * 18;
        const int cy = min(max(ty0 + hy - 1, 0), H - 1), cx = min(max(tx0 + hx - 1, 0), W - 1);
        const unsigned off = (unsigned)((cy * W + cx) * Cin + sq * 4 + c0) * 4u;
        return *reinterpret_cast<const float4 *>(xsamp + off);
    };
    // Branch-free (see wino_fwd_kernel): out-of-image pixels are zeroed with a bit mask; threads without a 6th pixel write into
    // the 16-byte pad of their first pixel
    auto store_piece = [&](float4 v, int k, float *buf, int c0) {
        const float4 isc = *reinterpret_cast<const float4 *>(&aff[c0 + sq * 4]);
        const float4 ish = *reinterpret_cast<const float4 *>(&aff[AFFC + c0 + sq * 4]);
        const int pix = spix0 + k * 32 + opaque_zero();
        const bool real = pix < 180;
        const int hy0 = (pix * 3641) >> 16, hx0 = pix - hy0 * 18;
        const int gy = ty0 + hy0 - 1, gx = tx0 + hx0 - 1;
        const unsigned m = (real && gy >= 0 && gy < H && gx >= 0 && gx < W) ? 0xffffffffu : 0u;
        const int pw = real ? pix : spix0;
        const int hy = (pw * 3641) >> 16, hx = pw - hy * 18;
        float4 o;
        o.x = __uint_as_float(__float_as_uint(fmaf(v.x, isc.x, ish.x)) & m);
        o.y = __uint_as_float(__float_as_uint(fmaf(v.y, isc.y, ish.y)) & m);
        o.z = __uint_as_float(__float_as_uint(fmaf(v.z, isc.z, ish.z)) & m);
        o.w = __uint_as_float(__float_as_uint(fmaf(v.w, isc.w, ish.w)) & m);
        *reinterpret_cast<float4 *>(&buf[((hy * 2 + (hx & 1)) * WHALF + (hx >> 1)) * WAS + (real ? sq * 4 : 32)]) = o;
    };
    // r[j] = d[ia][j] + sg d[ib][j] for the 8-channel read g of the chunk (columns 0..3 of the 4x4 tile)
    auto read_rows = [&](const float *As, int g, float4 (&r)[4]) {
        const float *pa = As + offa + g * 8, *pb = As + offb + g * 8;
        const float4 a0 = *reinterpret_cast<const float4 *>(pa);
        const float4 a1 = *reinterpret_cast<const float4 *>(pa + J1);
        const float4 a2 = *reinterpret_cast<const float4 *>(pa + J2);
        const float4 a3 = *reinterpret_cast<const float4 *>(pa + J1 + J2);
        const float4 b0 = *reinterpret_cast<const float4 *>(pb);
        const float4 b1 = *reinterpret_cast<const float4 *>(pb + J1);
        const float4 b2 = *reinterpret_cast<const float4 *>(pb + J2);
        const float4 b3 = *reinterpret_cast<const float4 *>(pb + J1 + J2);
        r[0] = f4_fma(b0, sg, a0);
        r[1] = f4_fma(b1, sg, a1);
        r[2] = f4_fma(b2, sg, a2);
        r[3] = f4_fma(b3, sg, a3);
    };

    // B fragments (pre-split filter): a ring of RB units in registers; a unit = the three terms of one (transform position,
    // 32-channel output tile) of one 16-channel group = 6 MFMAs.  Unit u is requested when unit u - (RB - 1) starts.
#ifndef B3_RB
#define B3_RB 2
#endif
    constexpr int RB = B3_RB;
    float4 bq[RB][3];
    const int ntstride = nG * 768;                        // floats between the two output tiles of a workgroup
    auto load_unit = [&](int slot, int v, int Gi, int nt) {
        if (B3_WHATIF & 1) { v = 0; Gi = 0; }
        // uniform base + ONE 32-bit lane offset per unit (SGPR base, immediate offsets for the terms)
        const unsigned off = (unsigned)((v * ustride_pos + ((size_t)nt * nG + Gi) * 768) * 4) + ulane;
#pragma unroll
        for (int tm = 0; tm < 3; ++tm) bq[slot][tm] = *reinterpret_cast<const float4 *>(ubase + off + tm * 1024);
    };
    (void)ntstride;
    // unit index within a chunk: q = (2 Gl * 4 + v) * NT + nt; the first RB - 1 units are requested here
#pragma unroll
    for (int q = 0; q < ((B3_WHATIF & 16) ? RB : RB - 1); ++q) load_unit(q % RB, (q / NT) & 3, q / (4 * NT), q % NT);

    __syncthreads();                                      // affine table visible
    {                                                     // first patch: all six pieces in flight together
        float4 pw[APT];
#pragma unroll
        for (int k = 0; k < APT; ++k) pw[k] = load_piece(k, 0);
#pragma unroll
        for (int k = 0; k < APT; ++k) store_piece(pw[k], k, lds, 0);
    }
    __syncthreads();

    // The main loop is a pinned software pipeline (sched_barrier / sched_group_barrier; left alone the scheduler sinks every
    // load to just above its use).  A step = one transform position v of one 16-channel group: 6 NT MFMAs on A[v & 1] and
    // bq[step & 1].  Inside step s the wave also (a) requests the B fragments of step s + 1, (b) builds the A fragments of step
    // s + 1 -- input-transform column combination and the three-term split, ~52 VALU instructions; in the last step of a group
    // also the 16 LDS reads and row combinations of the next group -- and these are dealt out BETWEEN the MFMAs (one MFMA, then
    // a few VALU / LDS / VMEM instructions: a wave issues in order, and the matrix pipe takes one K = 16 instruction per 8 passes
    // whoever issues it).  The chunk barrier sits in front of the chunk's LAST step, whose LDS reads are the next chunk's.
    float4 r0[4], r1[4];
    bf16x8 Ah[2], Am[2], Al[2];
    auto build_a = [&](int v, int slot) {
        const float4 a0 = v == 0 ? f4_sub1(r0[0], r0[2]) : (v == 1 ? f4_add1(r0[1], r0[2]) : (v == 2 ? f4_sub1(r0[2], r0[1]) : f4_sub1(r0[1], r0[3])));
        const float4 a1 = v == 0 ? f4_sub1(r1[0], r1[2]) : (v == 1 ? f4_add1(r1[1], r1[2]) : (v == 2 ? f4_sub1(r1[2], r1[1]) : f4_sub1(r1[1], r1[3])));
        if (B3_WHATIF & 2) {
            Ah[slot] = as_b8(a0); Am[slot] = as_b8(a1); Al[slot] = as_b8(f4_add(a0, a1));
        } else {
            split3(a0, a1, Ah[slot], Am[slot], Al[slot]);
        }
    };
    read_rows(lds, 0, r0);
    read_rows(lds, 1, r1);
    build_a(0, 0);

    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        const float *An = lds + ((ch + 1) & 1) * WPATCH;
        float *Anw = lds + ((ch + 1) & 1) * WPATCH;
        const float *As = lds + (ch & 1) * WPATCH;
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int Gl = st >> 2, v = st & 3;
            const int Gi = ch * 2 + Gl;
            if (more) {                                   // staging of the next chunk, one piece per step
                if (st >= 1 && st <= APT) store_piece(pv, st - 1, Anw, (ch + 1) * WKC);
                if (st < APT) pv = load_piece(st, (ch + 1) * WKC);
            }
            if (st == 7) __syncthreads();                 // the chunk's last step reads the next chunk's patch
            // one unit per output tile: (a) request unit q + RB - 1, (b) a share of the next step's A fragments, (c) 6 MFMAs
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int q = st * NT + nt;               // unit index within the chunk (8 NT units per chunk)
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int qn = q + RB - 1;            // may run into the next chunk
                    const int cn = qn / (8 * NT), qq = qn % (8 * NT);
                    int Gn = (ch + cn) * 2 + (qq / NT) / 4;
                    Gn = Gn < nG ? Gn : nG - 1;
                    if (!(B3_WHATIF & 16)) load_unit(qn % RB, (qq / NT) & 3, Gn, qq % NT);
                }
                if (nt == 0) {
                    if (v < 3) {
                        build_a(v + 1, (st + 1) & 1);
                    } else if (st == 3) {
                        read_rows(As, 2, r0);
                        read_rows(As, 3, r1);
                        build_a(0, (st + 1) & 1);
                    } else if (more) {
                        read_rows(An, 0, r0);
                        read_rows(An, 1, r1);
                        build_a(0, (st + 1) & 1);
                    }
                }
                {
                    const bf16x8 ah = Ah[st & 1], am = Am[st & 1], al = Al[st & 1];
                    const bf16x8 bh = as_b8(bq[q % RB][0]), bm = as_b8(bq[q % RB][1]), bl = as_b8(bq[q % RB][2]);
                    constexpr int v2 = (B3_WHATIF & 8) ? 1 : 0;      // (what-if: alternate MFMAs on another accumulator)
                    if (!(B3_WHATIF & 4)) {
                        acc[v][nt] = mfma_b16(al, bh, acc[v][nt]);
                        acc[v ^ v2][nt] = mfma_b16(ah, bl, acc[v ^ v2][nt]);
                        acc[v][nt] = mfma_b16(am, bm, acc[v][nt]);
                        acc[v ^ v2][nt] = mfma_b16(am, bh, acc[v ^ v2][nt]);
                    }
                    acc[v][nt] = mfma_b16(ah, bm, acc[v][nt]);
                    acc[v ^ v2][nt] = mfma_b16(ah, bh, acc[v ^ v2][nt]);
                }
#if B3_DEAL
                __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, B3_DEAL, 0);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    wino_epilogue<NT, false>(acc, lds, tid, lane, wave, li, bias, addend, addend_mask, y, stats, stat_aux, stat_mean,
                             stat_invstd, stat_mask, n, H, W, Cout, co0, ty0, tx0, nsp, sp, relu, mask_bits);
}

// ---- second form: 64 tiles x 64 output channels per workgroup, ONE workgroup per CU (512 registers per wave) ----------------
// With the matrix time cut 2.67 x, the two-workgroup form above is limited by what the fp32 kernel hides under its MFMAs: a B
// fragment (1 KB per wave) feeds ONE MFMA, i.e. 64 B/clk/CU of L2 -> L1 traffic at matrix-pipe saturation (the whole fill
// bandwidth of a CU), and there is no register room for more than one unit of B prefetch.  Here a wave owns TWO 32-tile blocks
// (8 x 8 tiles = 16 x 16 output pixels per workgroup, 18 x 18-pixel patches: halo 1.27 x instead of 1.41 x) and two
// 32-channel tiles: 16 accumulators (256 registers), every B fragment feeds two MFMAs (32 B/clk/CU at saturation), every A
// fragment two as before, and the B ring holds four steps (2 300 matrix cycles of prefetch distance).  One wave per SIMD: the
// split / transform VALU work of step s + 1 is dealt between the MFMAs of step s by the scheduler (a wave issues in order; the
// matrix pipe accepts one K = 16 instruction per 32 cycles, ~5 vector instructions fit in each gap).  The epilogue runs in two
// rounds (one 32-tile block at a time) through the same exchange buffer as the other forward kernels.
#ifndef B3V2_WHATIF
#define B3V2_WHATIF 0   // timing-only (results invalid): 1 no staging, 2 no B loads in the loop, 4 no split, 8 no LDS reads in the loop, 16 two of six MFMAs
#endif
#ifndef B3V2_RB
#define B3V2_RB 2
#endif
#ifndef B3V2_PD
#define B3V2_PD 1     // steps between the request of a staged piece and its LDS write (6 request steps + PD <= 8)
#endif
#ifndef B3V2_DEAL
#define B3V2_DEAL 0
#endif
constexpr int W2ROWS = 18;
constexpr int W2PATCH = W2ROWS * 2 * WHALF * WAS;          // floats per staged 18 x 18 patch (51.8 KB)
constexpr int W2PIECES = 11;                                // 16-byte pieces per thread and chunk (324 pixels x 8 / 256)

__global__ __launch_bounds__(256, 1) void wino_fwd_b3v2_kernel(
    const float *__restrict__ x, const float *__restrict__ u, const float *__restrict__ bias,
    const float *__restrict__ addend, const float *__restrict__ addend_mask, const float *__restrict__ in_scale,
    const float *__restrict__ in_shift, float *__restrict__ y, float *__restrict__ stats,
    const float *__restrict__ stat_aux, const float *__restrict__ stat_mean, const float *__restrict__ stat_invstd,
    const float *__restrict__ stat_mask, int H, int W, int Cin, int Cout, int tilesW, int tilesH, int tilesH2, int nsp2,
    int nsp, int ncb, int xcd_div, int relu, int mask_bits) {
    constexpr int NT = 2, MT = 2;
    using Cfg = WinoCfg<NT, false>;
    constexpr int CB = Cfg::CB;
    constexpr int AFFC = WMAXC;
    constexpr int LDSF = 2 * W2PATCH > Cfg::PBUF ? 2 * W2PATCH : Cfg::PBUF;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    __shared__ __attribute__((aligned(16))) float aff[2 * AFFC];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    for (int c = tid; c < Cin; c += 256) {
        aff[c] = in_scale ? in_scale[c] : 1.f;
        aff[AFFC + c] = in_scale ? in_shift[c] : 0.f;
    }
    int sp2, cb;
    if (xcd_div > 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        cb = xcd % ncb;
        sp2 = j * xcd_div + xcd / ncb;
    } else {
        cb = blockIdx.x % ncb;
        sp2 = blockIdx.x / ncb;
    }
    if (sp2 >= nsp2) return;
    sp2 = nsp2 - 1 - sp2;
    int t = sp2;
    const int tw = t % tilesW;
    t /= tilesW;
    const int th2 = t % tilesH2;
    const int n = t / tilesH2;
    const int co0 = cb * CB;
    const int ty0 = th2 * 16, tx0 = tw * 16;

    f32x16 acc[MT][4][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][v][nt][r] = 0.f;

    const int ia = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int ib = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = wave == 1 ? 1.f : -1.f;
    const int tr = li >> 3, tc = li & 7;
    const int offa = (((2 * tr + ia) * 2) * WHALF + tc) * WAS + lh * 4;
    const int offb = (((2 * tr + ib) * 2) * WHALF + tc) * WAS + lh * 4;
    constexpr int J1 = WHALF * WAS, J2 = WAS;
    constexpr int MTOFF = 8 * 2 * WHALF * WAS;              // the second 32-tile block starts 8 patch rows further down

    const int sq = tid & 7, spix0 = tid >> 3;
    const int nchunks = Cin / WKC, nG = Cin / 16;
    const size_t ustride_pos = (size_t)(Cout / 32) * nG * 768;
    const char *ubase = reinterpret_cast<const char *>(u + ((size_t)(wave * 4) * (Cout / 32) + (size_t)cb * NT) * nG * 768);
    const unsigned ulane = lane * 16u;
    const char *xsamp = reinterpret_cast<const char *>(x + (size_t)n * H * W * Cin);
    auto opaque_zero = [&]() {
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        return z;
    };
    // staging descriptors of this thread's 11 pieces (pixel spix0 + 32 k, channels 4 sq ..), computed once: byte offset of
    // the pixel in the sample, float offset of its slot in a patch buffer, in-image bit (one wave per SIMD: the per-piece
    // index arithmetic of the other forward kernels would sit exposed between the MFMAs)
    unsigned goff[W2PIECES];
    short loff[W2PIECES];
    unsigned inimg = 0;
#pragma unroll
    for (int k = 0; k < W2PIECES; ++k) {
        const int pix = spix0 + k * 32;
        const bool real = pix < W2ROWS * 18;
        const int hy0 = (pix * 3641) >> 16, hx0 = pix - hy0 * 18;
        const int gy = ty0 + hy0 - 1, gx = tx0 + hx0 - 1;
        const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
        goff[k] = (unsigned)((cy * W + cx) * Cin + sq * 4) * 4u;
        if (real && gy >= 0 && gy < H && gx >= 0 && gx < W) inimg |= 1u << k;
        const int pw = real ? pix : spix0;
        const int hy = (pw * 3641) >> 16, hx = pw - hy * 18;
        loff[k] = (short)(((hy * 2 + (hx & 1)) * WHALF + (hx >> 1)) * WAS + (real ? sq * 4 : 32));
    }
    auto load_piece = [&](int k, int c0) { return *reinterpret_cast<const float4 *>(xsamp + goff[k] + (unsigned)c0 * 4u); };
    float4 isc, ish;                                      // producer affine of the chunk being staged (this thread's 4 channels)
    auto load_aff = [&](int c0) {
        isc = *reinterpret_cast<const float4 *>(&aff[c0 + sq * 4]);
        ish = *reinterpret_cast<const float4 *>(&aff[AFFC + c0 + sq * 4]);
    };
    auto store_piece = [&](float4 v, int k, float *buf, int c0) {
        (void)c0;
        const unsigned m = (inimg >> k) & 1u ? 0xffffffffu : 0u;
        float4 o;
        o.x = __uint_as_float(__float_as_uint(fmaf(v.x, isc.x, ish.x)) & m);
        o.y = __uint_as_float(__float_as_uint(fmaf(v.y, isc.y, ish.y)) & m);
        o.z = __uint_as_float(__float_as_uint(fmaf(v.z, isc.z, ish.z)) & m);
        o.w = __uint_as_float(__float_as_uint(fmaf(v.w, isc.w, ish.w)) & m);
        *reinterpret_cast<float4 *>(&buf[loff[k]]) = o;
    };
    auto read_rows = [&](const float *As, int g, float4 (&r)[4]) {
        const float *pa = As + offa + g * 8, *pb = As + offb + g * 8;
        const float4 a0 = *reinterpret_cast<const float4 *>(pa);
        const float4 a1 = *reinterpret_cast<const float4 *>(pa + J1);
        const float4 a2 = *reinterpret_cast<const float4 *>(pa + J2);
        const float4 a3 = *reinterpret_cast<const float4 *>(pa + J1 + J2);
        const float4 b0 = *reinterpret_cast<const float4 *>(pb);
        const float4 b1 = *reinterpret_cast<const float4 *>(pb + J1);
        const float4 b2 = *reinterpret_cast<const float4 *>(pb + J2);
        const float4 b3 = *reinterpret_cast<const float4 *>(pb + J1 + J2);
        r[0] = f4_fma(b0, sg, a0);
        r[1] = f4_fma(b1, sg, a1);
        r[2] = f4_fma(b2, sg, a2);
        r[3] = f4_fma(b3, sg, a3);
    };

    // B ring: two steps (a step = one transform position of one 16-channel group: both channel tiles, three terms each = 24
    // registers); the fragments of step s + 1 are requested when step s starts: 24 MFMAs = 768 matrix cycles ahead
    constexpr int RB2 = B3V2_RB;
    float4 bq[RB2][NT][3];
    auto load_step = [&](int slot, int v, int Gi) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const unsigned off = (unsigned)((v * ustride_pos + ((size_t)nt * nG + Gi) * 768) * 4) + ulane;
#pragma unroll
            for (int tm = 0; tm < 3; ++tm) bq[slot][nt][tm] = *reinterpret_cast<const float4 *>(ubase + off + tm * 1024);
        }
    };
#pragma unroll
    for (int s0 = 0; s0 < RB2 - 1; ++s0) load_step(s0, s0 & 3, (s0 >> 2) < nG ? (s0 >> 2) : nG - 1);

    __syncthreads();                                      // affine table visible
    {                                                     // first patch
        load_aff(0);
        float4 pw[W2PIECES];
#pragma unroll
        for (int k = 0; k < W2PIECES; ++k) pw[k] = load_piece(k, 0);
#pragma unroll
        for (int k = 0; k < W2PIECES; ++k) store_piece(pw[k], k, lds, 0);
    }
    __syncthreads();

    // A side.  r0 / r1[mt]: row combinations of the two 8-channel reads of the current group for tile block mt; A fragments are
    // built ONE unit ahead (a unit = one tile block of one step: 12 MFMAs), double-buffered: 24 registers
    float4 r0[MT][4], r1[MT][4];
    bf16x8 Ah[2], Am[2], Al[2];
    auto build_a = [&](int v, int mt, int slot) {
        const float4 a0 = v == 0 ? f4_sub1(r0[mt][0], r0[mt][2]) : (v == 1 ? f4_add1(r0[mt][1], r0[mt][2]) : (v == 2 ? f4_sub1(r0[mt][2], r0[mt][1]) : f4_sub1(r0[mt][1], r0[mt][3])));
        const float4 a1 = v == 0 ? f4_sub1(r1[mt][0], r1[mt][2]) : (v == 1 ? f4_add1(r1[mt][1], r1[mt][2]) : (v == 2 ? f4_sub1(r1[mt][2], r1[mt][1]) : f4_sub1(r1[mt][1], r1[mt][3])));
        if (B3V2_WHATIF & 4) {
            Ah[slot] = as_b8(a0); Am[slot] = as_b8(a1); Al[slot] = as_b8(f4_add(a0, a1));
        } else {
            split3(a0, a1, Ah[slot], Am[slot], Al[slot]);
        }
    };
    auto read_block = [&](const float *As, int Gl, int mt) {
        if ((B3V2_WHATIF & 8) && (Gl != 0 || As != lds)) return;
        read_rows(As + mt * MTOFF, 2 * Gl, r0[mt]);
        read_rows(As + mt * MTOFF, 2 * Gl + 1, r1[mt]);
    };
    read_block(lds, 0, 0);
    build_a(0, 0, 0);

    float4 pv[B3V2_PD][2];
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        const float *An = lds + ((ch + 1) & 1) * W2PATCH;
        float *Anw = lds + ((ch + 1) & 1) * W2PATCH;
        const float *As = lds + (ch & 1) * W2PATCH;
        if (more) load_aff((ch + 1) * WKC);
#pragma clang loop unroll(full)
        for (int q = 0; q < 16; ++q) {
            const int st = q >> 1, mt = q & 1;
            const int Gl = st >> 2, v = st & 3;
            const int Gi = ch * 2 + Gl;
            if (q == 15) __syncthreads();                 // the chunk's last unit reads the next chunk's patch
            __builtin_amdgcn_sched_barrier(0);
            if (mt == 0 && more && !(B3V2_WHATIF & 1)) {   // staging of the next chunk: two pieces per step (11 in all),
                constexpr int PD = B3V2_PD;               // written to LDS PD steps after they were requested
                if (st >= PD && st - PD < 6) {
                    const int k0 = 2 * (st - PD);
                    store_piece(pv[(st - PD) % PD][0], k0, Anw, (ch + 1) * WKC);
                    if (k0 + 1 < W2PIECES) store_piece(pv[(st - PD) % PD][1], k0 + 1, Anw, (ch + 1) * WKC);
                }
                if (st < 6) {
                    pv[st % PD][0] = load_piece(2 * st, (ch + 1) * WKC);
                    if (2 * st + 1 < W2PIECES) pv[st % PD][1] = load_piece(2 * st + 1, (ch + 1) * WKC);
                }
            }
            if (mt == 0) {                                // (a) B fragments RB2 - 1 steps ahead
                const int sn = st + RB2 - 1;
                int Gn = (ch + (sn >> 3)) * 2 + ((sn & 7) >> 2);
                Gn = Gn < nG ? Gn : nG - 1;
                if (!(B3V2_WHATIF & 2)) load_step(sn % RB2, sn & 3, Gn);
            }
            // (b) A fragments of the next unit
            if (mt == 0) {
                if (v == 0) read_block(As, Gl, 1);        // the group's second tile block (its first was read one unit earlier)
                build_a(v, 1, (q + 1) & 1);
            } else if (v < 3) {
                build_a(v + 1, 0, (q + 1) & 1);
            } else if (q == 7) {
                read_block(As, 1, 0);
                build_a(0, 0, (q + 1) & 1);
            } else if (more) {
                read_block(An, 0, 0);
                build_a(0, 0, (q + 1) & 1);
            }
            // (c) this unit's 12 MFMAs
            {
                const bf16x8 ah = Ah[q & 1], am = Am[q & 1], al = Al[q & 1];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8 bh = as_b8(bq[st % RB2][nt][0]), bm = as_b8(bq[st % RB2][nt][1]), bl = as_b8(bq[st % RB2][nt][2]);
                    if (!(B3V2_WHATIF & 16)) {
                        acc[mt][v][nt] = mfma_b16(al, bh, acc[mt][v][nt]);
                        acc[mt][v][nt] = mfma_b16(ah, bl, acc[mt][v][nt]);
                        acc[mt][v][nt] = mfma_b16(am, bm, acc[mt][v][nt]);
                        acc[mt][v][nt] = mfma_b16(am, bh, acc[mt][v][nt]);
                    }
                    acc[mt][v][nt] = mfma_b16(ah, bm, acc[mt][v][nt]);
                    acc[mt][v][nt] = mfma_b16(ah, bh, acc[mt][v][nt]);
                }
            }
#if B3V2_DEAL
            // the deal (one wave per SIMD: nothing else fills the gaps): memory requests first, then one MFMA followed by a
            // share of the vector work; whatever is left goes behind the last MFMA
            if (mt == 0) __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, B3V2_DEAL, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // epilogue, one 32-tile block at a time (statistics rows are the 8 x 16-pixel patches of the other forward kernels)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int th8 = 2 * th2 + mt;
        if (th8 < tilesH) {
            const int sp = (n * tilesH + th8) * tilesW + tw;
            wino_epilogue<NT, false>(acc[mt], lds, tid, lane, wave, li, bias, addend, addend_mask, y, stats, stat_aux, stat_mean,
                                     stat_invstd, stat_mask, n, H, W, Cout, co0, ty0 + 8 * mt, tx0, nsp, sp, relu, mask_bits);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void wino_pack_b3_kernel(const float *__restrict__ w, unsigned short *__restrict__ u,
                                                           int Cin_real, int K, int Nn, int mode) {
    const long total = (long)(Nn / 32) * (K / 16) * 512;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) wino_pack_b3_one(w, u, Cin_real, K, Nn, mode, idx);
}

}  // namespace adyolo

using namespace adyolo;

extern "C" int adyolo_wino_pack_w_b3(const float *w, float *u_fwd, float *u_dgrad, int Cout, int Cin_real, int Cin,
                                     void *stream) {
    ADYOLO_REQUIRE(w && (u_fwd || u_dgrad) && Cout > 0 && Cin_real > 0 && Cin >= Cin_real, ADYOLO_EINVAL,
                   "wino_pack_w_b3: bad arguments");
    ADYOLO_REQUIRE(Cout % 32 == 0 && Cin % 32 == 0, ADYOLO_ENOSUP,
                   "wino_pack_w_b3: Cin=%d and Cout=%d must be multiples of 32", Cin, Cout);
    const long total = (long)(Cout / 32) * (Cin / 16) * 512;      // same count for both packings
    if (u_fwd)
        hipLaunchKernelGGL(wino_pack_b3_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w,
                           reinterpret_cast<unsigned short *>(u_fwd), Cin_real, Cin, Cout, 0);
    if (u_dgrad)
        hipLaunchKernelGGL(wino_pack_b3_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w,
                           reinterpret_cast<unsigned short *>(u_dgrad), Cin_real, Cout, Cin, 1);
    return check_launch("wino_pack_w_b3");
}

extern "C" int adyolo_wino_fwd_b3(const float *x, const float *u, const float *bias, const float *addend,
                                  const float *addend_mask, const float *in_scale, const float *in_shift, float *y,
                                  float *stats, const float *stat_aux, const float *stat_mean, const float *stat_invstd,
                                  const float *stat_mask, int N, int H, int W, int Cin, int Cout, int relu, int mask_bits,
                                  void *stream) {
    ADYOLO_REQUIRE(x && u && y && N > 0 && H > 0 && W > 0, ADYOLO_EINVAL, "wino_fwd_b3: bad arguments");
    ADYOLO_REQUIRE(!(mask_bits & ~3) && (!mask_bits || ((long)H * W * (Cout / 4)) % 64 == 0), ADYOLO_ENOSUP,
                   "wino_fwd_b3: mask bits need H*W*Cout/4 %% 64 == 0");
    ADYOLO_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0 && Cin > 0 && Cout > 0 && Cin <= WMAXC, ADYOLO_ENOSUP,
                   "wino_fwd_b3: Cin=%d (<= 512) and Cout=%d must be multiples of 32", Cin, Cout);
    ADYOLO_REQUIRE((size_t)H * W * Cin * 4 < ((size_t)1 << 31), ADYOLO_ENOSUP, "wino_fwd_b3: one sample must stay below 2 GiB");
    ADYOLO_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (!addend_mask || addend), ADYOLO_EINVAL,
                   "wino_fwd_b3: in_scale/in_shift come together; addend_mask needs addend");
    ADYOLO_REQUIRE(!stat_aux || (stats && stat_mean && stat_invstd), ADYOLO_EINVAL,
                   "wino_fwd_b3: stat_aux needs stats, stat_mean and stat_invstd");
    ADYOLO_REQUIRE(!stat_mask || stats, ADYOLO_EINVAL, "wino_fwd_b3: stat_mask needs stats");
    const int tilesW = cdiv(W, 16), tilesH = cdiv(H, 8);
    const int nsp = N * tilesH * tilesW;
    const int nt = Cout % 64 == 0 ? 2 : 1;
    const int ncb = Cout / (32 * nt);
    int xcd_div = 0, blocks = nsp * ncb;
    if (ncb <= 8 && 8 % ncb == 0) {
        xcd_div = 8 / ncb;
        blocks = cdiv(nsp, xcd_div) * 8;
    }
    hipStream_t st = as_stream(stream);
    // the 64-tile x 64-channel form (one workgroup per CU) for channel counts in multiples of 64; ADYOLO_B3_KERNEL=v1 keeps
    // the two-workgroup form (A/B runs)
    static const int use_v2 = [] {
        const char *e = getenv("ADYOLO_B3_KERNEL");
        return (e && e[0] == 'v' && e[1] == '1') ? 0 : 1;
    }();
    if (nt == 2 && use_v2) {
        const int tilesH2 = cdiv(H, 16);
        const int nsp2 = N * tilesH2 * tilesW;
        int xd2 = 0, blocks2 = nsp2 * ncb;
        if (ncb <= 8 && 8 % ncb == 0) {
            xd2 = 8 / ncb;
            blocks2 = cdiv(nsp2, xd2) * 8;
        }
        hipLaunchKernelGGL(wino_fwd_b3v2_kernel, dim3((unsigned)blocks2), dim3(256), 0, st, x, u, bias, addend, addend_mask,
                           in_scale, in_shift, y, stats, stat_aux, stat_mean, stat_invstd, stat_mask, H, W, Cin, Cout, tilesW,
                           tilesH, tilesH2, nsp2, nsp, ncb, xd2, relu, mask_bits);
    } else if (nt == 2)
        hipLaunchKernelGGL((wino_fwd_b3_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, x, u, bias, addend, addend_mask,
                           in_scale, in_shift, y, stats, stat_aux, stat_mean, stat_invstd, stat_mask, H, W, Cin, Cout, tilesW,
                           tilesH, nsp, ncb, xcd_div, relu, mask_bits);
    else
        hipLaunchKernelGGL((wino_fwd_b3_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, x, u, bias, addend, addend_mask,
                           in_scale, in_shift, y, stats, stat_aux, stat_mean, stat_invstd, stat_mask, H, W, Cin, Cout, tilesW,
                           tilesH, nsp, ncb, xcd_div, relu, mask_bits);
    return check_launch("wino_fwd_b3");
}
