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
    if (nt == 2)
        hipLaunchKernelGGL((wino_fwd_b3_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, x, u, bias, addend, addend_mask,
                           in_scale, in_shift, y, stats, stat_aux, stat_mean, stat_invstd, stat_mask, H, W, Cin, Cout, tilesW,
                           tilesH, nsp, ncb, xcd_div, relu, mask_bits);
    else
        hipLaunchKernelGGL((wino_fwd_b3_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, x, u, bias, addend, addend_mask,
                           in_scale, in_shift, y, stats, stat_aux, stat_mean, stat_invstd, stat_mask, H, W, Cin, Cout, tilesW,
                           tilesH, nsp, ncb, xcd_div, relu, mask_bits);
    return check_launch("wino_fwd_b3");
}
