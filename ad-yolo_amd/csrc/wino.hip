// K2w: 3x3 convolution (stride 1, pad 1), channels-last, as Winograd F(2x2, 3x3) on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32).  Same operator as conv.hip (nn.Conv2d at /root/reference/src/models/backbones/
// resnet.py:16,18 -- forward and data-gradient), 2.25x fewer matrix FLOPs:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, 4x4 input tile d, 3x3 filter g
// The 16 transform positions (xi, nu) are 16 independent GEMMs  M[pos][tile][cout] = sum_cin V[pos][tile][cin] U[pos][cin][cout].
//
// One workgroup (4 waves) owns 32 tiles (4 x 8 tiles = 8 x 16 output pixels) x CB = 32*NT output channels; wave w owns
// the four positions of transform row xi = w (NT x 4 accumulator tiles).  Per 32-channel chunk the 10 x 18 input patch
// is staged in LDS (double-buffered, next chunk prefetched through registers under the MFMAs).  There is no V buffer:
// B^T has two non-zeros per row, so a lane builds its A fragments from 8 ds_read_b128 of patch pixels and 32 VALU ops
// per 8-channel group -- 16*NT MFMAs (1024*NT matrix cycles) of cover.  Patch columns are de-interleaved by parity
// with 10 slots per half row and 144-byte pixels, which makes every 16-lane ds_read_b128 group hit 16 distinct slots.
// U = G g G^T is packed once per launch in MFMA-fragment order, so the B operand is a fully coalesced 1 KB
// global_load_dwordx4 per wave straight from L2 (each byte is used by exactly one wave of the workgroup: LDS staging
// would buy nothing), loaded one 8-channel group ahead.  Blocks are dealt to XCDs so that an XCD keeps one output-
// channel slice of U in its L2.  Epilogue: the nu-sum of A^T . A is done in registers, the xi-sum through LDS, then
// bias / masked addend / ReLU / per-patch BatchNorm sums as in conv.hip, stored as float4 along channels.
#include "wino_common.hpp"

namespace adyolo {


template <int NT, bool ONE>
__global__ __launch_bounds__(256, (WinoCfg<NT, ONE>::WG_PER_CU)) void wino_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ u, const float *__restrict__ bias,
    const float *__restrict__ addend, const float *__restrict__ addend_mask, const float *__restrict__ in_scale,
    const float *__restrict__ in_shift, float *__restrict__ y, float *__restrict__ stats,
    const float *__restrict__ stat_aux, const float *__restrict__ stat_mean, const float *__restrict__ stat_invstd,
    const float *__restrict__ stat_mask, int H, int W, int Cin, int Cout, int tilesW, int tilesH, int nsp, int ncb,
    int xcd_div, int relu, int mask_bits) {
    // mask_bits: bit 0 -- addend_mask points to ReLU-mask BITS (common.hpp mask_bits4) instead of a float tensor;
    //            bit 1 -- the same for stat_mask (1/32 of the epilogue's read traffic for that operand)
    using Cfg = WinoCfg<NT, ONE>;
    constexpr int CB = Cfg::CB, CBP = Cfg::CBP;
    constexpr int AFFC = ONE ? WKC : WMAXC;               // channels in the affine table
    __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) float aff[2 * AFFC];      // producer BatchNorm scale | shift (1 | 0 if none)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform: keeps xi-dependent values in SGPRs
    const int li = lane & 31, lh = lane >> 5;
    for (int c = tid; c < Cin; c += 256) {
        aff[c] = in_scale ? in_scale[c] : 1.f;
        aff[AFFC + c] = in_scale ? in_shift[c] : 0.f;
    }
    // block -> (spatial patch, channel block): blocks are dealt round-robin to the 8 XCDs; with xcd_div = 8/ncb
    // an XCD always works on channel block (xcd % ncb), so its L2 keeps one slice of U
    int sp, cb;
    if (xcd_div > 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        cb = xcd % ncb;
        sp = j * xcd_div + xcd / ncb;
    } else {
        cb = blockIdx.x % ncb;
        sp = blockIdx.x / ncb;
    }
    if (sp >= nsp) return;
    // patches are walked from the LAST to the first: the elementwise producer in front of this launch wrote the tensor
    // front to back, so its tail is what the 256 MB memory-side cache still holds, and the consumer after this launch
    // reads front to back again (-0.3 % per step; results identical)
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

    // B^T row xi = wave:  r[j] = d[ia][j] + sg * d[ib][j]
    const int ia = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int ib = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = wave == 1 ? 1.f : -1.f;
    const int tr = li >> 3, tc = li & 7;
    const int offa = (((2 * tr + ia) * 2) * WHALF + tc) * WAS + lh * 4;
    const int offb = (((2 * tr + ib) * 2) * WHALF + tc) * WAS + lh * 4;
    constexpr int J1 = WHALF * WAS, J2 = WAS;                                // column j of the 4x4 tile: (j & 1) J1 + (j >> 1) J2

    // staging: thread owns 16-byte piece q of pixels spix0 + 32 i
    const int sq = tid & 7, spix0 = tid >> 3;
    constexpr int APT = 6;                                                    // 180 pixels / 32 per pass
    const int nchunks = Cin / WKC, nkg = Cin / 8;
    const size_t ustride_pos = (size_t)(Cout / 32) * nkg * 256;               // floats per transform position
    // uniform base + 32-bit lane offset: global_load with an SGPR base, no 64-bit address arithmetic per load
    const char *ubase = reinterpret_cast<const char *>(u + ((size_t)(wave * 4) * (Cout / 32) + (size_t)cb * NT) * nkg * 256);
    const unsigned ulane = lane * 16u;

    // staging registers: raw pixels of half of the next chunk's patch (clamped, unconditional loads so that they are
    // issued back to back); the affine and the zero padding are applied when they are written to LDS.  Source
    // offsets, LDS offsets and the in-image flags are RECOMPUTED per call from an opaque zero: hoisted out of the loop
    // they would cost 13 long-lived registers, which this kernel (256 VGPRs at 2 workgroups per CU) would spill, and a
    // scratch reload drains the B-operand prefetches (scratch and global loads share vmcnt).
    float4 pv[APT / 2];
    const char *xsamp = reinterpret_cast<const char *>(x + (size_t)n * H * W * Cin);
    auto opaque_zero = [&]() {
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        return z;
    };
    auto load_patch_into = [&](float4 (&dst)[APT / 2], int half, int c0) {
        const int z = opaque_zero();
#pragma unroll
        for (int ii = 0; ii < APT / 2; ++ii) {
            const int pix = spix0 + (half * (APT / 2) + ii) * 32 + z;
            const int hy = (pix * 3641) >> 16, hx = pix - hy * 18;            // pix / 18 for pix < 2^12
            const int cy = min(max(ty0 + hy - 1, 0), H - 1), cx = min(max(tx0 + hx - 1, 0), W - 1);
            const unsigned off = (unsigned)((cy * W + cx) * Cin + sq * 4 + c0) * 4u;
            dst[ii] = *reinterpret_cast<const float4 *>(xsamp + off);
        }
    };
    auto load_patch = [&](int half, int c0) { load_patch_into(pv, half, c0); };
    // Branch-free on purpose: a wait for pv[i] inside a divergent branch leaves the load "possibly outstanding" on the
    // other path, and the compiler then drains the B-operand prefetches at the top of every chunk to protect pv's
    // registers.  Out-of-image pixels are zeroed with a bit mask; threads without a 6th pixel write into the 16-byte
    // pad of their first pixel.
    auto store_patch_from = [&](const float4 (&src)[APT / 2], int half, float *buf, int c0) {
        const float4 isc = *reinterpret_cast<const float4 *>(&aff[c0 + sq * 4]);
        const float4 ish = *reinterpret_cast<const float4 *>(&aff[AFFC + c0 + sq * 4]);
        const int z = opaque_zero();
#pragma unroll
        for (int ii = 0; ii < APT / 2; ++ii) {
            const int pix = spix0 + (half * (APT / 2) + ii) * 32 + z;
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
            *reinterpret_cast<float4 *>(&buf[((hy * 2 + (hx & 1)) * WHALF + (hx >> 1)) * WAS + (real ? sq * 4 : 32)]) = o;
        }
    };
    auto store_patch = [&](int half, float *buf, int c0) { store_patch_from(pv, half, buf, c0); };
    float4 r0, r1, r2, r3;
    float4 da0, da1, db0, db1;                          // raw patch pixels in flight (two columns of the 4x4 tile at a time)
    auto issue_cols = [&](const float *As, int g, int half) {     // columns 2 half, 2 half + 1 of rows ia, ib
        const float *pa = As + offa + g * 8 + half * J2, *pb = As + offb + g * 8 + half * J2;
        da0 = *reinterpret_cast<const float4 *>(pa);
        da1 = *reinterpret_cast<const float4 *>(pa + J1);
        db0 = *reinterpret_cast<const float4 *>(pb);
        db1 = *reinterpret_cast<const float4 *>(pb + J1);
    };
    // r[j] = d[ia][j] + sg d[ib][j]
    auto combine_lo = [&]() { r0 = f4_fma(db0, sg, da0); r1 = f4_fma(db1, sg, da1); };
    auto combine_hi = [&]() { r2 = f4_fma(db0, sg, da0); r3 = f4_fma(db1, sg, da1); };

    // B fragments of the first 8-channel group
    // B fragments: PF 8-channel groups in flight.  NT = 2: one group = 4 steps = 2048 matrix cycles ahead (register-bound);
    // NT = 1: a step is only 4 MFMAs, so two groups keep the same 2048 cycles of cover (one group exposes the L2 latency)
    constexpr int PF = NT == 1 ? 2 : 1;
    float4 bq[PF][4][NT];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int kg0 = p < nkg ? p : nkg - 1;
                bq[p][v][nt] = *reinterpret_cast<const float4 *>(
                    ubase + ((unsigned)((v * ustride_pos + ((size_t)nt * nkg + kg0) * 256) * 4) + ulane));
            }

    __syncthreads();                                      // affine table visible
    {
        // First patch, all six pixels of the thread in flight together.  Round 4: its own code instead of the loop's
        // load_patch / store_patch pair -- those recompute every index from scratch on both sides (the loop has no registers to
        // keep them) and cost ~73 vector instructions per pixel incl. five quarter-rate integer multiplies; here the patch
        // coordinates advance by increments (32 pixels = one row + 14 columns of the 18-wide patch), the source offset is one
        // 24-bit multiply, out-of-image pixels are an out-of-range offset of a buffer descriptor (zeros) and offsets are kept
        // for the store: ~26 per pixel.  The stage-1 kernel executed 15 vector instructions per MFMA, 44 % of them here
        // (profiles/r04_pmc_stage1_vs_stage4.txt), and every one costs matrix-pipe time.
        const __amdgpu_buffer_rsrc_t xrs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (size_t)n * H * W * Cin), 0, H * W * Cin * 4, 0x00020000);
        const float4 isc = *reinterpret_cast<const float4 *>(&aff[sq * 4]);
        const float4 ish = *reinterpret_cast<const float4 *>(&aff[AFFC + sq * 4]);
        int hy = (spix0 * 3641) >> 16, hx = spix0 - hy * 18;
        const int lds_first = ((hy * 2 + (hx & 1)) * WHALF + (hx >> 1)) * WAS;
        f32x4 pf[APT];
        int voff[APT], loff[APT];
#pragma unroll
        for (int ii = 0; ii < APT; ++ii) {
            const bool real = ii < APT - 1 || spix0 + 32 * ii < 180;          // (only the sixth pixel can lie beyond the 180)
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const bool ok = real && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            voff[ii] = ok ? (__mul24(__mul24(gy, W) + gx, Cin) + sq * 4) * 4 : (int)0x80000000;
            loff[ii] = real ? ((hy * 2 + (hx & 1)) * WHALF + (hx >> 1)) * WAS + sq * 4 : lds_first + 32;   // (else: its first pixel's pad)
            pf[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff[ii], 0, 0));
            hx += 14;
            hy += 1;
            if (hx >= 18) {
                hx -= 18;
                hy += 1;
            }
        }
#pragma unroll
        for (int ii = 0; ii < APT; ++ii) {
            const bool ok = voff[ii] >= 0;
            float4 o;
            o.x = fmaf(pf[ii].x, isc.x, ok ? ish.x : 0.f);
            o.y = fmaf(pf[ii].y, isc.y, ok ? ish.y : 0.f);
            o.z = fmaf(pf[ii].z, isc.z, ok ? ish.z : 0.f);
            o.w = fmaf(pf[ii].w, isc.w, ok ? ish.w : 0.f);
            *reinterpret_cast<float4 *>(&lds[loff[ii]]) = o;
        }
    }
    __syncthreads();

    // The main loop is a hand-placed software pipeline; __builtin_amdgcn_sched_barrier(0) pins it (left alone, the
    // scheduler sinks every prefetch to just above its use and exposes the L2 latency 16 times per chunk):
    //   step (g, v):  8 MFMAs on bq[v], then the loads that refill bq[v] for group g+1 (4 steps = 2048 matrix
    //   cycles ahead of their use); the next chunk's pixels are requested at the top of the chunk and written to the
    //   other LDS buffer at its end; the LDS reads of group g+1 are issued under the MFMAs of step (g, 3).
    for (int ch = 0; ch < nchunks; ++ch) {
        const float *As = lds + (ONE ? 0 : (ch & 1) * WPATCH);
        const bool more = !ONE && ch + 1 < nchunks;
        float *An = lds + (ONE ? 0 : ((ch + 1) & 1) * WPATCH);
        if (more) load_patch(0, (ch + 1) * WKC);
        issue_cols(As, 0, 0);
        combine_lo();
        issue_cols(As, 0, 1);
        combine_hi();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < WKC / 8; ++g) {
            const int kg = ch * (WKC / 8) + g;
            const int kgn = kg + PF < nkg ? kg + PF : nkg - 1;
            const bool nextg = g + 1 < WKC / 8;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 a = v == 0 ? f4_sub(r0, r2) : (v == 1 ? f4_add(r1, r2) : (v == 2 ? f4_sub(r2, r1) : f4_sub(r1, r3)));
                // the LDS reads of the next group ride under this group's last step: columns 0,1 ahead of its first
                // MFMAs, columns 2,3 in the middle (r0..r3 are dead once `a` of step 3 exists)
                if (v == 3 && nextg) {
                    issue_cols(As, g + 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[v][nt] = mfma32(a.x, bq[g % PF][v][nt].x, acc[v][nt]);
                    acc[v][nt] = mfma32(a.y, bq[g % PF][v][nt].y, acc[v][nt]);
                    if (NT == 1 && v == 3 && nextg) {
                        __builtin_amdgcn_sched_barrier(0);
                        combine_lo();
                        issue_cols(As, g + 1, 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[v][nt] = mfma32(a.z, bq[g % PF][v][nt].z, acc[v][nt]);
                    acc[v][nt] = mfma32(a.w, bq[g % PF][v][nt].w, acc[v][nt]);
                    if (NT == 2 && nt == 0 && v == 3 && nextg) {
                        __builtin_amdgcn_sched_barrier(0);
                        combine_lo();
                        issue_cols(As, g + 1, 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bq[g % PF][v][nt] = *reinterpret_cast<const float4 *>(
                        ubase + ((unsigned)((v * ustride_pos + ((size_t)nt * nkg + kgn) * 256) * 4) + ulane));
                if (v == 3 && nextg) {
                    __builtin_amdgcn_sched_barrier(0);
                    combine_hi();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == 1 && more) {                       // mid-chunk: first half lands in the other buffer, second half requested
                store_patch(0, An, (ch + 1) * WKC);
                load_patch(1, (ch + 1) * WKC);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) store_patch(1, An, (ch + 1) * WKC);
        __syncthreads();
    }

    wino_epilogue<NT, ONE>(acc, lds, tid, lane, wave, li, bias, addend, addend_mask, y, stats, stat_aux, stat_mean, stat_invstd,
                           stat_mask, n, H, W, Cout, co0, ty0, tx0, nsp, sp, relu, mask_bits);
}

// U = G g G^T in fragment order [16 pos][Cout/32][Cin/8][64 lanes][4]: lane (n, h) element j = U_pos[cin 8g+4h+j][cout 32cb+n].
// mode 0: forward filter g = w[cout][cin];  mode 1: data-gradient filter g[ky][kx] = w[k][n][2-ky][2-kx]
// (the GEMM's "cin" runs over the forward Cout and its "cout" over the forward, padded, Cin).
__device__ __forceinline__ void wino_pack_one(const float *__restrict__ w, float *__restrict__ u, int Cin_real, int K,
                                              int Nn, int mode, long idx, long total) {
    // K = GEMM reduction channels, Nn = GEMM output channels
    const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const long rest = idx >> 8;
    const int g = (int)(rest % (K / 8)), cbk = (int)(rest / (K / 8));
    const int k = g * 8 + (lane >> 5) * 4 + j, nn = cbk * 32 + (lane & 31);
    float f[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float v = 0.f;
            if (mode == 0) {
                if (k < Cin_real) v = w[((size_t)nn * Cin_real + k) * 9 + a * 3 + b];
            } else {
                if (nn < Cin_real) v = w[((size_t)k * Cin_real + nn) * 9 + (2 - a) * 3 + (2 - b)];
            }
            f[a][b] = v;
        }
    // t = G f  (4x3), U = t G^T (4x4);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float tt[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        tt[0][b] = f[0][b];
        tt[1][b] = 0.5f * (f[0][b] + f[1][b] + f[2][b]);
        tt[2][b] = 0.5f * (f[0][b] - f[1][b] + f[2][b]);
        tt[3][b] = f[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float u0 = tt[a][0], u1 = 0.5f * (tt[a][0] + tt[a][1] + tt[a][2]),
                    u2 = 0.5f * (tt[a][0] - tt[a][1] + tt[a][2]), u3 = tt[a][2];
        const size_t ps = (size_t)total;
        u[(size_t)(a * 4 + 0) * ps + idx] = u0;
        u[(size_t)(a * 4 + 1) * ps + idx] = u1;
        u[(size_t)(a * 4 + 2) * ps + idx] = u2;
        u[(size_t)(a * 4 + 3) * ps + idx] = u3;
    }
}

__global__ __launch_bounds__(256) void wino_pack_kernel(const float *__restrict__ w, float *__restrict__ u, int Cout_f,
                                                        int Cin_real, int K, int Nn, int mode) {
    (void)Cout_f;
    const long total = (long)(Nn / 32) * (K / 8) * 256;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) wino_pack_one(w, u, Cin_real, K, Nn, mode, idx, total);
}

// Every 3x3 filter of a model in ONE launch (64 pack launches per train step of SE-ResNet34 otherwise: the packed filters
// change once per optimizer step, not per layer call).  table: [n][8] int64 = {w, u_fwd, u_dgrad (or 0), Cout, Cin_real, Cin,
// unused, unused}.  grid (ceil(largest total / 256), n)
__global__ __launch_bounds__(256) void wino_pack_many_kernel(const long long *__restrict__ table) {
    const long long *d = table + 8 * blockIdx.y;
    const float *w = reinterpret_cast<const float *>(d[0]);
    float *uf = reinterpret_cast<float *>(d[1]), *ud = reinterpret_cast<float *>(d[2]);
    const int Cout = (int)d[3], Cin_real = (int)d[4], Cin = (int)d[5];
    const long total = (long)(Cout / 32) * (Cin / 8) * 256;       // = Cout * Cin: one element per thread in every form
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    if (uf) wino_pack_one(w, uf, Cin_real, Cin, Cout, 0, idx, total);
    if (ud) wino_pack_one(w, ud, Cin_real, Cout, Cin, 1, idx, total);
}

}  // namespace adyolo

using namespace adyolo;

extern "C" int adyolo_wino_tiles(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return ADYOLO_EINVAL;
    return N * cdiv(H, 8) * cdiv(W, 16);
}

extern "C" int adyolo_wino_pack_w(const float *w, float *u_fwd, float *u_dgrad, int Cout, int Cin_real, int Cin,
                                  void *stream) {
    ADYOLO_REQUIRE(w && (u_fwd || u_dgrad) && Cout > 0 && Cin_real > 0 && Cin >= Cin_real, ADYOLO_EINVAL,
                   "wino_pack_w: bad arguments");
    ADYOLO_REQUIRE(Cout % 32 == 0 && Cin % 32 == 0, ADYOLO_ENOSUP,
                   "wino_pack_w: Cin=%d and Cout=%d must be multiples of 32", Cin, Cout);
    const long total = (long)(Cout / 32) * (Cin / 8) * 256;       // same count for both packings
    if (u_fwd)
        hipLaunchKernelGGL(wino_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, u_fwd, Cout,
                           Cin_real, Cin, Cout, 0);
    if (u_dgrad)
        hipLaunchKernelGGL(wino_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, u_dgrad, Cout,
                           Cin_real, Cout, Cin, 1);
    return check_launch("wino_pack_w");
}

extern "C" int adyolo_wino_pack_many(const int64_t *table, int n, int max_cout, int max_cin, void *stream) {
    ADYOLO_REQUIRE(table && n > 0 && max_cout > 0 && max_cin > 0 && max_cout % 32 == 0 && max_cin % 32 == 0, ADYOLO_EINVAL,
                   "wino_pack_many: bad arguments");
    const long total = (long)(max_cout / 32) * (max_cin / 8) * 256;
    hipLaunchKernelGGL(wino_pack_many_kernel, dim3(cdiv(total, 256), n), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const long long *>(table));
    return check_launch("wino_pack_many");
}

extern "C" int adyolo_wino_fwd(const float *x, const float *u, const float *bias, const float *addend,
                               const float *addend_mask, const float *in_scale, const float *in_shift, float *y,
                               float *stats, const float *stat_aux, const float *stat_mean, const float *stat_invstd,
                               const float *stat_mask, int N, int H, int W, int Cin, int Cout, int relu, int mask_bits,
                               void *stream) {
    ADYOLO_REQUIRE(x && u && y && N > 0 && H > 0 && W > 0, ADYOLO_EINVAL, "wino_fwd: bad arguments");
    ADYOLO_REQUIRE(!(mask_bits & ~3) && (!mask_bits || ((long)H * W * (Cout / 4)) % 64 == 0), ADYOLO_ENOSUP,
                   "wino_fwd: mask bits need H*W*Cout/4 %% 64 == 0");
    ADYOLO_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0 && Cin > 0 && Cout > 0 && Cin <= WMAXC, ADYOLO_ENOSUP,
                   "wino_fwd: Cin=%d (<= 512) and Cout=%d must be multiples of 32", Cin, Cout);
    // 31-bit byte offsets inside a sample's buffer descriptor (input AND output side: the epilogue's dropped-store offset
    // 0x80000000 must stay out of range) and 24-bit pixel indices (__mul24 in the staging / epilogue address arithmetic)
    ADYOLO_REQUIRE((size_t)H * W * Cin * 4 < ((size_t)1 << 31) && (size_t)H * W * Cout * 4 < ((size_t)1 << 31) &&
                       (long)H * W < (1L << 23),
                   ADYOLO_ENOSUP, "wino_fwd: one sample (input and output) must stay below 2 GiB and 2^23 pixels");
    ADYOLO_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (!addend_mask || addend), ADYOLO_EINVAL,
                   "wino_fwd: in_scale/in_shift come together; addend_mask needs addend");
    ADYOLO_REQUIRE(!stat_aux || (stats && stat_mean && stat_invstd), ADYOLO_EINVAL,
                   "wino_fwd: stat_aux needs stats, stat_mean and stat_invstd");
    ADYOLO_REQUIRE(!stat_mask || stats, ADYOLO_EINVAL, "wino_fwd: stat_mask needs stats");
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
#define ADYOLO_WINO_FWD(NT_, ONE_)                                                                                  \
    hipLaunchKernelGGL((wino_fwd_kernel<NT_, ONE_>), dim3((unsigned)blocks), dim3(256), 0, st, x, u, bias, addend,       \
                       addend_mask, in_scale, in_shift, y, stats, stat_aux, stat_mean, stat_invstd, stat_mask, H, W,   \
                       Cin, Cout, tilesW, tilesH, nsp, ncb, xcd_div, relu, mask_bits)
    if (Cin == WKC) {
        if (nt == 2) ADYOLO_WINO_FWD(2, true); else ADYOLO_WINO_FWD(1, true);
    } else {
        if (nt == 2) ADYOLO_WINO_FWD(2, false); else ADYOLO_WINO_FWD(1, false);
    }
#undef ADYOLO_WINO_FWD
    return check_launch("wino_fwd");
}

#ifndef WGRAD_WHATIF
#define WGRAD_WHATIF 0
#endif

namespace adyolo {

// ================================================================================================
// K2w weight-gradient, Winograd form (replaces the backward-weights half of nn.Conv2d, resnet.py:16,18):
//     dw = G^T [ sum over 2x2 output tiles of (B^T d B) (.) (A e A^T) ] G         d: 4x4 input tile, e: 2x2 tile of dy
// i.e. 16 GEMMs  dU[pos][ci][co] = sum_tiles V[pos][tile][ci] E[pos][tile][co]  with the contraction over tiles --
// 16 instead of 36 multiplies per tile, channel pair.  One workgroup owns a (32 ci x 32*NT co) block of all 16
// positions (wave w: transform row xi = w) and walks 16-pixel-wide column strips of the images top to bottom, TWO
// tile rows (2 x 8 tiles = two MFMA k-groups, 32*NT MFMAs per wave) per step and barrier.  Both operands need the TILE
// index along a lane's registers, so the four new x rows and dy rows of a step are fetched with channel-contiguous
// dword loads (SGPR base + 32-bit offsets, all issued before the step's MFMAs, one whole step = 4096 matrix cycles
// ahead of their use) and written to LDS transposed, [row][column phase j][channel][8 tiles]: a ds_read_b128 then
// yields 4 tiles of one channel.  The two tile quads of a channel are swapped when (channel >> 3) is odd, which puts
// every 16-lane ds_read_b128 group on 16 distinct 16-byte slots without padding.  x rows live in a 10-slot ring
// (6 read + 4 being written), dy rows in an 8-slot ring: 72 KB, two workgroups per CU.  Signs of the dy transform
// (A = [[1,0],[1,1],[1,-1],[0,-1]]) that are plain negations are folded into one sign flip of the accumulators at the
// end; waves 0 and 3 read one dy row instead of two.  Each workgroup writes one slab of dU; slabs are summed in a
// fixed order and G^T . G is applied by two small kernels (deterministic).

template <int NT, bool RAGGED>
__global__ __launch_bounds__(256, 2) void wino_wgrad_kernel(
    const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ in_scale,
    const float *__restrict__ in_shift, float *__restrict__ slabs, int H, int W, int Cin, int Cout, int tilesW,
    int tilesH, int nseg, int seg_rows, int nitems, int nsplit, int ciBlocks) {
    constexpr int CB = 32 * NT;
    constexpr int XROW = 4 * 32 * 8;              // floats per x row slot
    constexpr int DROW = 2 * CB * 8;              // floats per dy row slot
    constexpr int XSLOTS = 10, DSLOTS = 8;
    __shared__ __attribute__((aligned(16))) float Xs[XSLOTS * XROW];
    __shared__ __attribute__((aligned(16))) float Dsh[(DSLOTS + 1) * DROW];          // + one row of zeros (see substep)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x;
    const int cbk = blockIdx.y / ciBlocks, ibk = blockIdx.y - cbk * ciBlocks;
    const int co0 = cbk * CB, c0 = ibk * 32;

    f32x16 acc[4][NT];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[v][nt][r] = 0.f;

    // transform row xi = wave.  x side (B^T): r[j] = d[ia][j] + sg d[ib][j]
    const int ia = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int ib = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = wave == 1 ? 1.f : -1.f;
    const float sgd = wave == 2 ? -1.f : 1.f;
    const int aoff = li * 8 + ((lh ^ ((li >> 3) & 1)) << 2);        // channel li, tile quad lh (swizzled)
    for (int i = tid; i < DROW; i += 256) Dsh[DSLOTS * DROW + i] = 0.f;      // (made visible by the first barrier below)

    // staging roles.  x: channel xci, column phase xj, rows xr and xr + 2 of the step's four, all 8 tiles.
    // dy: channel dco, column phase dj, rows dr and dr + 2, DK tiles starting at tile dk0
    const int xci = tid & 31, xg = tid >> 5, xr = xg >> 2, xj = xg & 3;
    float xsc = 1.f, xsh = 0.f;
    if (in_scale) {
        xsc = in_scale[c0 + xci];
        xsh = in_shift[c0 + xci];
    }
    constexpr int DK = NT == 2 ? 8 : 4;
    const int dco = NT == 2 ? (tid & 63) : (tid & 31);
    const int dg = NT == 2 ? (tid >> 6) : (tid >> 5);
    const int dr = NT == 2 ? (dg >> 1) : (dg >> 2);
    const int dj = NT == 2 ? (dg & 1) : ((dg >> 1) & 1);
    const int dk0 = NT == 2 ? 0 : 4 * (dg & 1);
    const int xsw = (xci >> 3) & 1, dsw = (dco >> 3) & 1;
    float *xdst = Xs + (xj * 32 + xci) * 8;
    float *ddst = Dsh + (dj * CB + dco) * 8;
    constexpr bool ragged = RAGGED;                                 // (W & 15) != 0 || (H & 1) != 0: the general masking path
    const int xrowb = W * Cin * 4, drowb = W * Cout * 4, xpixb = Cin * 4, dpixb = Cout * 4;

    for (int item = split; item < nitems; item += nsplit) {
        const int seg = item % nseg;
        const int rest = item / nseg;
        const int tw = rest % tilesW, n = rest / tilesW;
        const int tr0 = seg * seg_rows;
        const int nrows = min(seg_rows, tilesH - tr0);
        const int nbig = (nrows + 1) >> 1;
        const int tx0 = tw * 16;
        const char *xn = reinterpret_cast<const char *>(x + (size_t)n * H * W * Cin + c0);        // uniform bases
        const char *dn = reinterpret_cast<const char *>(dy + (size_t)n * H * W * Cout + co0);
        const int xgx0 = tx0 - 1 + xj, dgx0 = tx0 + dj + 2 * dk0;     // image column of tile 0 of this thread's run

        float xraw[2][8], draw[2][DK];            // (never live together: see the step loop)
        // Round 4: the fetches go through buffer descriptors of the sample (base = this block's first channel).  The fp32 MFMA
        // shares its issue port with the vector ALU (tools/micro/mfma32_coissue.hip: every VALU instruction between two MFMAs
        // costs its full issue time), and the old form spent ~100 VALU instructions per step on addresses: a clamp pair and a
        // quarter-rate integer multiply per pixel, then a 64-bit add per load.  Now the thread holds three byte offsets per
        // item (tiles 1..6 / tile 0 / tile 7 of its run), a step adds the uniform row offset to them (6 adds per 16 loads)
        // and tile k is selected by the SCALAR offset operand (2 k pixels; no VALU).  Rows above / below the image give a
        // negative / too large vector offset, which the range check answers with 0 (the hardware checks the vector offset
        // only -- which is also why tile 0 and tile 7, the two that can leave the image sideways in the non-ragged case, keep
        // clamped offsets of their own instead of relying on it).  Ragged images (W % 16, H % 2) keep the general path.
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + (size_t)n * H * W * Cin + c0), 0, (H * W * Cin - c0) * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(dy + (size_t)n * H * W * Cout + co0), 0, (H * W * Cout - co0) * 4, 0x00020000);
        const int xthr = xr * xrowb + xci * 4;
        const int xcolA = xthr + (xgx0 + 2) * xpixb;
        const int xcol0 = xthr + max(xgx0, 0) * xpixb;
        const int xcol7 = xthr + min(xgx0 + 14, W - 1) * xpixb;
        const int dcol = dr * drowb + dco * 4 + dgx0 * dpixb;
        // (column offsets are recomputed per call from an opaque zero: hoisted out of the step loop they would hold
        //  16 more registers and this kernel would spill)
        auto opaque_zero = [&]() {
            int z;
            asm volatile("v_mov_b32 %0, 0" : "=v"(z));
            return z;
        };
        // rows rr0 + xr and rr0 + xr + 2 (relative to image row 2 tr0 - 1); the loads are unconditional
        auto load_x_into = [&](float (&xr_)[2][8], int rr0) {
            if (!ragged) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int so = (2 * tr0 - 1 + rr0 + 2 * p) * xrowb;         // uniform
                    const int vA = xcolA + so, v0 = xcol0 + so, v7 = xcol7 + so;
                    xr_[p][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, v0, 0, 0));
#pragma unroll
                    for (int k = 1; k < 7; ++k)
                        xr_[p][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vA, (2 * k - 2) * xpixb, 0));
                    xr_[p][7] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, v7, 0, 0));
                }
                return;
            }
            const int gxz = xgx0 + opaque_zero();
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int gy = 2 * tr0 - 1 + rr0 + xr + 2 * p;
                const int row = min(max(gy, 0), H - 1) * xrowb;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int gx = min(max(gxz + 2 * k, 0), W - 1);
                    xr_[p][k] = *reinterpret_cast<const float *>(xn + (unsigned)(row + gx * xpixb + xci * 4));
                }
            }
        };
        auto store_x_from = [&](const float (&xr_)[2][8], int rr0) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int rr = rr0 + xr + 2 * p;
                const int gy = 2 * tr0 - 1 + rr;
                const bool rowok = gy >= 0 && gy < H;
                float t[8];
                if (!ragged) {          // only the two outer tiles of a strip can leave the image
                    const float sc = rowok ? xsc : 0.f, sh = rowok ? xsh : 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = fmaf(xr_[p][k], sc, sh);
                    t[0] = xgx0 >= 0 ? t[0] : 0.f;
                    t[7] = xgx0 + 14 < W ? t[7] : 0.f;
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int gx = xgx0 + 2 * k;
                        t[k] = (rowok && gx >= 0 && gx < W) ? fmaf(xr_[p][k], xsc, xsh) : 0.f;
                    }
                }
                // ring slot (rr mod 10): rr0 + 2 p is even and uniform, xr is 0 or 1 -- no per-thread division
                float *q = xdst + (((rr0 + 2 * p) % XSLOTS) + xr) * XROW;
                *reinterpret_cast<float4 *>(q + (xsw << 2)) = make_float4(t[0], t[1], t[2], t[3]);
                *reinterpret_cast<float4 *>(q + ((xsw ^ 1) << 2)) = make_float4(t[4], t[5], t[6], t[7]);
            }
        };
        auto load_d = [&](int rd0) {              // dy rows rd0 + dr, rd0 + dr + 2 (relative to image row 2 tr0)
            if (!ragged) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int vd = dcol + (2 * tr0 + rd0 + 2 * p) * drowb;
#pragma unroll
                    for (int k = 0; k < DK; ++k)
                        draw[p][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, vd, 2 * k * dpixb, 0));
                }
                return;
            }
            const int gxz = dgx0 + opaque_zero();
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int gy = 2 * tr0 + rd0 + dr + 2 * p;
                const int row = min(gy, H - 1) * drowb;
#pragma unroll
                for (int k = 0; k < DK; ++k) {
                    const int gx = min(gxz + 2 * k, W - 1);
                    draw[p][k] = *reinterpret_cast<const float *>(dn + (unsigned)(row + gx * dpixb + dco * 4));
                }
            }
        };
        auto store_d = [&](int rd0) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int rd = rd0 + dr + 2 * p;
                float t[DK];
#pragma unroll
                for (int k = 0; k < DK; ++k) t[k] = draw[p][k];
                if (ragged) {
                    const bool rowok = 2 * tr0 + rd < H;
#pragma unroll
                    for (int k = 0; k < DK; ++k) t[k] = (rowok && dgx0 + 2 * k < W) ? t[k] : 0.f;
                }
                float *q = ddst + (((rd0 + 2 * p) & (DSLOTS - 1)) + dr) * DROW;     // (rd0 + 2 p is even and uniform, dr is 0 or 1)
                if (NT == 2) {
                    *reinterpret_cast<float4 *>(q + (dsw << 2)) = make_float4(t[0], t[1], t[2], t[3]);
                    *reinterpret_cast<float4 *>(q + ((dsw ^ 1) << 2)) = make_float4(t[DK - 4], t[DK - 3], t[DK - 2], t[DK - 1]);
                } else {
                    *reinterpret_cast<float4 *>(q + ((((dk0 >> 2) ^ dsw) & 1) << 2)) = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
        };

        auto load_x = [&](int rr0) { load_x_into(xraw, rr0); };
        auto store_x = [&](int rr0) { store_x_from(xraw, rr0); };
        {                                           // prologue: x rows 0..7 and dy rows 0..3, one exposed latency
            float xraw2[2][8];
            load_x_into(xraw, 0);
            load_d(0);
            load_x_into(xraw2, 4);
            store_x_from(xraw, 0);
            store_d(0);
            store_x_from(xraw2, 4);
        }
        __syncthreads();

        // Wave priorities (s_setprio): the staging code between the substeps runs at 0, the LDS reads and transforms of a
        // substep at 1, its MFMAs at 3, so that of the two waves sharing a SIMD (one of each resident workgroup) the one
        // with matrix work ready is issued first: measured -4 % at every stage (same box, stage 4: 3.29 -> 3.16 ms)
        auto substep = [&](int rb) {                  // one tile row: x rows rb .. rb + 3, dy rows rb, rb + 1 of the rings
            __builtin_amdgcn_s_setprio(1);
            const float *xa = Xs + ((rb + ia) % XSLOTS) * XROW + aoff;
            const float *xb = Xs + ((rb + ib) % XSLOTS) * XROW + aoff;
            // dy side, row xi of A e A^T up to sign: wave 0: e0, wave 1: e0 + e1, wave 2: e0 - e1, wave 3: e1 (negated at the end)
            const float *eP = Dsh + (wave == 3 ? DSLOTS : (rb & (DSLOTS - 1))) * DROW + aoff;
            const float *eQ = Dsh + (wave == 0 ? DSLOTS : ((rb + 1) & (DSLOTS - 1))) * DROW + aoff;
            float4 r0, r1, r2, r3;
            {
                const float4 a0 = *reinterpret_cast<const float4 *>(xa);
                const float4 a1 = *reinterpret_cast<const float4 *>(xa + 32 * 8);
                const float4 a2 = *reinterpret_cast<const float4 *>(xa + 64 * 8);
                const float4 a3 = *reinterpret_cast<const float4 *>(xa + 96 * 8);
                const float4 b0 = *reinterpret_cast<const float4 *>(xb);
                const float4 b1 = *reinterpret_cast<const float4 *>(xb + 32 * 8);
                const float4 b2 = *reinterpret_cast<const float4 *>(xb + 64 * 8);
                const float4 b3 = *reinterpret_cast<const float4 *>(xb + 96 * 8);
                r0 = f4_fma(b0, sg, a0);
                r1 = f4_fma(b1, sg, a1);
                r2 = f4_fma(b2, sg, a2);
                r3 = f4_fma(b3, sg, a3);
            }
            // dy side, row xi of A e A^T up to sign: wave 0: e0, wave 1: e0 + e1, wave 2: e0 - e1, wave 3: e1 (negated
            // at the end); columns: nu 0: s0, 1: s0 + s1, 2: s0 - s1, 3: s1 (negated at the end)
            // (round 4: branch-free.  Waves 0 and 3 used to read one dy row and the others two, under wave-uniform branches:
            //  twelve scalar branches and ~60 register moves per tile row to merge the paths.  Now every wave reads two rows --
            //  the missing one is a row of zeros kept behind the ring -- and forms P + sgd Q.)
            float4 s0[NT], s1[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 p0 = *reinterpret_cast<const float4 *>(eP + nt * 32 * 8);
                const float4 p1 = *reinterpret_cast<const float4 *>(eP + (CB + nt * 32) * 8);
                const float4 q0 = *reinterpret_cast<const float4 *>(eQ + nt * 32 * 8);
                const float4 q1 = *reinterpret_cast<const float4 *>(eQ + (CB + nt * 32) * 8);
                s0[nt] = f4_fma(q0, sgd, p0);
                s1[nt] = f4_fma(q1, sgd, p1);
            }
            __builtin_amdgcn_sched_barrier(0);
#if WGRAD_WHATIF
            // timing-only model of a F(4x4)-domain weight-gradient (results invalid): 9/16 of the MFMAs
            // (18 of the 32 per tile row) and the extra transform instructions of 6 x 6 tiles as dependent-free dummy VALU
            // (WGRAD_WHATIF = their number per tile row)
            {
                float dmy[4] = {r0.x, r1.y, r2.z, r3.w};
#pragma unroll
                for (int i = 0; i < WGRAD_WHATIF; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(dmy[i & 3]));
                r0.x = dmy[0]; r1.y = dmy[1]; r2.z = dmy[2]; r3.w = dmy[3];
            }
#endif
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 a = v == 0 ? f4_sub(r0, r2) : (v == 1 ? f4_add(r1, r2) : (v == 2 ? f4_sub(r2, r1) : f4_sub(r1, r3)));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 b = v == 0 ? s0[nt] : (v == 1 ? f4_add(s0[nt], s1[nt]) : (v == 2 ? f4_sub(s0[nt], s1[nt]) : s1[nt]));
                    acc[v][nt] = mfma32(a.x, b.x, acc[v][nt]);
                    acc[v][nt] = mfma32(a.y, b.y, acc[v][nt]);
#if WGRAD_WHATIF
                    if (v == 0 || (v == 1 && nt == 0)) acc[v][nt] = mfma32(a.z, b.z, acc[v][nt]);      // 2 per (v, nt) + 2 more = 18 of 32
#else
                    acc[v][nt] = mfma32(a.z, b.z, acc[v][nt]);
                    acc[v][nt] = mfma32(a.w, b.w, acc[v][nt]);
#endif
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        };

        // step T: dy rows of step T+1 are requested at the top and land in LDS after the first tile row; the x rows of
        // step T+1 are requested then and land after the second (16 prefetch registers at a time instead of 32)
        // (the prefetch is unconditional -- on the last step it fetches clamped rows nobody reads: a load/store pair
        //  under `if (more)` leaves the loads "possibly outstanding" in the compiler's vmcnt scoreboard at the loop
        //  header, and it then drains every prefetch right after issuing it)
#pragma unroll 1
        for (int T = 0; T < nbig; ++T) {
            load_d(4 * T + 4);
            __builtin_amdgcn_sched_barrier(0);
            substep(4 * T);
            store_d(4 * T + 4);
            load_x(4 * T + 6);
            __builtin_amdgcn_sched_barrier(0);
            if (2 * T + 1 < nrows) substep(4 * T + 2);
            store_x(4 * T + 6);
            __syncthreads();
        }
    }

    // one slab per workgroup: [split][pos][Cin][Cout];  folded signs: wave 3 and nu = 3 each negate
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int pos = wave * 4 + v;
        const float sgn = ((wave == 3) != (v == 3)) ? -1.f : 1.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mfma_row(r, lane);
                slabs[(((size_t)split * 16 + pos) * Cin + c0 + m) * Cout + co0 + nt * 32 + li] = sgn * acc[v][nt][r];
            }
    }
}

// slabs [nslab][total] -> du [total] (fixed order, partial sums in double).  (Round 5 tried ONE launch for the slab sum and G^T . G --
// a workgroup per 32 (ci, co) pairs, 16 barrier-separated column sums: bit-identical, but 34 us per call against 12 + 5 for the two
// launches below (latency-bound with few slabs, 16 x fewer workgroups): reverted; profiles/r05_small_shapes.json of that build.)
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ du,
                                                                int nslab, int total) {
    __shared__ double red[256];
    const double sum = block_colsum32(slabs, nslab, (size_t)total, blockIdx.x * 32, total, red);
    const int idx = blockIdx.x * 32 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) == 0 && idx < total) du[idx] = (float)sum;
}
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const float *__restrict__ du, float *__restrict__ dw,
                                                                int Cin, int Cin_real, int Cout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // over [Cin][Cout], co fastest
    if (idx >= Cin * Cout) return;
    const int co = idx % Cout, ci = idx / Cout;
    if (ci >= Cin_real) return;
    float d[4][4];
#pragma unroll
    for (int p = 0; p < 16; ++p) d[p >> 2][p & 3] = du[(size_t)p * Cin * Cout + idx];
    float t[3][4];                                // t = G^T d
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        t[0][v] = d[0][v] + 0.5f * (d[1][v] + d[2][v]);
        t[1][v] = 0.5f * (d[1][v] - d[2][v]);
        t[2][v] = 0.5f * (d[1][v] + d[2][v]) + d[3][v];
    }
    float *o = dw + ((size_t)co * Cin_real + ci) * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o[a * 3 + 0] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
        o[a * 3 + 1] = 0.5f * (t[a][1] - t[a][2]);
        o[a * 3 + 2] = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
    }
}

// Work items = (sample, 16-pixel column strip, segment of seg_rows tile rows), dealt STATICALLY to the nsplit workgroups of
// a (cin, cout) block pair (item = split, split + nsplit, ...); all pairs x nsplit = ~512 workgroups are resident together
// (2 per CU).  Every item pays one exposed prologue (8 x rows + 4 dy rows: about 1.5 steps' worth; measured: 32-row
// segments 3.53 ms, 64-row 3.31 ms at stage 4), so the launch takes  rounds x (seg_rows / 2 + 1.5)  steps with
// rounds = ceil(items / nsplit): the segment length is the one that minimises that.  Round 2 aimed at "about four items per
// workgroup" regardless: at the benchmark shape that cut the strips into 4-8 segments (2-7 % more steps than one item per
// workgroup), and at the reference's own shapes (16 x 20 s) into 16-row crumbs with partly idle rounds (20-30 % more).
static int wino_wgrad_geometry(int N, int H, int W, int Cin, int Cout, int *nt_o, int *nseg_o, int *seg_rows_o,
                               int *nitems_o) {
    const int nt = Cout % 64 == 0 ? 2 : 1;
    const int tilesH = cdiv(H, 2), tilesW = cdiv(W, 16);
    const int pairs = (Cout / (32 * nt)) * (Cin / 32);
    int nsplit0 = 512 / pairs;
    if (nsplit0 < 1) nsplit0 = 1;
    const int strips = N * tilesW;
    double best_cost = 0.0;
    int best_seg = 0, best_nseg = 1, best_split = 1;
    for (int want = 1; want <= tilesH / 4 + 1; ++want) {
        int seg_rows = cdiv(tilesH, want);
        seg_rows += seg_rows & 1;                         // even: a step is two tile rows
        if (seg_rows < 8) seg_rows = 8;
        const int nseg = cdiv(tilesH, seg_rows);
        const long nitems = (long)strips * nseg;
        const int nsplit = nitems < nsplit0 ? (int)nitems : nsplit0;
        const double cost = (double)cdiv(nitems, nsplit) * (0.5 * seg_rows + 1.5);
        if (best_seg == 0 || cost < best_cost - 1e-9) {
            best_cost = cost;
            best_seg = seg_rows;
            best_nseg = nseg;
            best_split = nsplit;
        }
        if (seg_rows == 8) break;
    }
    if (nt_o) *nt_o = nt;
    if (nseg_o) *nseg_o = best_nseg;
    if (seg_rows_o) *seg_rows_o = best_seg;
    if (nitems_o) *nitems_o = strips * best_nseg;
    return best_split;
}

}  // namespace adyolo

extern "C" int adyolo_wino_wgrad_slabs(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32) return ADYOLO_EINVAL;
    return adyolo::wino_wgrad_geometry(N, H, W, Cin, Cout, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int adyolo_wino_wgrad(const float *x, const float *dy, const float *in_scale, const float *in_shift,
                                 float *slabs, float *du, float *dw, int N, int H, int W, int Cin, int Cin_real,
                                 int Cout, void *stream) {
    ADYOLO_REQUIRE(x && dy && slabs && du && dw && N > 0 && H > 0 && W > 0, ADYOLO_EINVAL, "wino_wgrad: bad arguments");
    ADYOLO_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0 && Cin > 0 && Cout > 0 && Cin_real <= Cin && Cin_real > 0, ADYOLO_ENOSUP,
                   "wino_wgrad: unsupported channels Cin=%d Cout=%d", Cin, Cout);
    ADYOLO_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), ADYOLO_EINVAL, "wino_wgrad: in_scale/in_shift come together");
    ADYOLO_REQUIRE((size_t)H * W * (Cin > Cout ? Cin : Cout) * 4 < ((size_t)1 << 31), ADYOLO_ENOSUP,
                   "wino_wgrad: one sample must stay below 2 GiB");
    hipStream_t st = as_stream(stream);
    int nt, nseg, seg_rows, nitems;
    const int nsplit = wino_wgrad_geometry(N, H, W, Cin, Cout, &nt, &nseg, &seg_rows, &nitems);
    const int tilesH = cdiv(H, 2), tilesW = cdiv(W, 16), ciBlocks = Cin / 32;
    dim3 grid((unsigned)nsplit, (unsigned)((Cout / (32 * nt)) * ciBlocks));
    const bool ragged = (W & 15) != 0 || (H & 1) != 0;
#define ADYOLO_WINO_WGRAD(NT_, RG_)                                                                                     \
    hipLaunchKernelGGL((wino_wgrad_kernel<NT_, RG_>), grid, dim3(256), 0, st, x, dy, in_scale, in_shift, slabs, H, W, Cin, \
                       Cout, tilesW, tilesH, nseg, seg_rows, nitems, nsplit, ciBlocks)
    if (nt == 2) {
        if (ragged) ADYOLO_WINO_WGRAD(2, true); else ADYOLO_WINO_WGRAD(2, false);
    } else {
        if (ragged) ADYOLO_WINO_WGRAD(1, true); else ADYOLO_WINO_WGRAD(1, false);
    }
#undef ADYOLO_WINO_WGRAD
    int rc = check_launch("wino_wgrad");
    if (rc) return rc;
    const int total = 16 * Cin * Cout;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(cdiv(total, 32)), dim3(256), 0, st, slabs, du, nsplit, total);
    rc = check_launch("wino_wgrad_reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3(cdiv(Cin * Cout, 256)), dim3(256), 0, st, du, dw, Cin, Cin_real, Cout);
    return check_launch("wino_wgrad_finish");
}
