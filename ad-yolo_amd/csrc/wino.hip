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
#include "common.hpp"

namespace adyolo {

constexpr int WKC = 32;                    // input channels per chunk
constexpr int WAS = 36;                    // floats per staged pixel (144 B)
constexpr int WHALF = 10;                  // slots per (row, parity) half row (9 used)
constexpr int WPATCH = 10 * 2 * WHALF * WAS;   // floats per staged patch (28.8 KB)

template <int NT>
struct WinoCfg {
    static constexpr int CB = 32 * NT;
    static constexpr int CBP = CB + 8;                       // epilogue exchange row (conflict-free b32 writes)
    static constexpr int PBUF = 8 * 32 * CBP;                // [wave][b][tile][CBP]
    static constexpr int LDS_FLOATS = (2 * WPATCH > PBUF) ? 2 * WPATCH : PBUF;
};

__device__ __forceinline__ float4 f4_fma(float4 a, float s, float4 b) {      // b + s * a
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

template <int NT>
__global__ __launch_bounds__(256, 2) void wino_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ u, const float *__restrict__ bias,
    const float *__restrict__ addend, const float *__restrict__ addend_mask, const float *__restrict__ in_scale,
    const float *__restrict__ in_shift, float *__restrict__ y, float *__restrict__ stats,
    const float *__restrict__ stat_aux, const float *__restrict__ stat_mean, const float *__restrict__ stat_invstd,
    int H, int W, int Cin, int Cout, int tilesW, int tilesH, int nsp, int ncb, int xcd_div, int relu) {
    using Cfg = WinoCfg<NT>;
    constexpr int CB = Cfg::CB, CBP = Cfg::CBP;
    __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
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
    constexpr int J1 = WHALF * WAS, J2 = WAS, J3 = WHALF * WAS + WAS;       // column j of the 4x4 tile

    // staging: thread owns 16-byte piece q of pixels spix0 + 32 i
    const int sq = tid & 7, spix0 = tid >> 3;
    constexpr int APT = 6;                                                    // 180 pixels / 32 per pass
    const int nchunks = Cin / WKC, nkg = Cin / 8;
    const size_t ustride_pos = (size_t)(Cout / 32) * nkg * 256;               // floats per transform position
    const float *ubase = u + ((size_t)(wave * 4) * (Cout / 32) + (size_t)cb * NT) * nkg * 256 + lane * 4;

    // staging registers: raw pixels of the next chunk (clamped, unconditional loads so that they are issued back to
    // back), the affine and the zero padding are applied when they are written to LDS
    float4 pv[APT];
    unsigned psrc[APT];                                   // in 16-byte units from x
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int pix = spix0 + i * 32;
        const int hy = pix / 18, hx = pix - hy * 18;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const bool ok = pix < 180 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        okmask |= (ok ? 1u : 0u) << i;
        const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
        psrc[i] = (unsigned)(((((size_t)n * H + cy) * W + cx) * Cin) >> 2) + sq;
    }
    auto load_patch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < APT; ++i) pv[i] = reinterpret_cast<const float4 *>(x + c0)[psrc[i]];
    };
    auto store_patch = [&](float *buf, int c0) {
        float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in_scale) {
            isc = *reinterpret_cast<const float4 *>(in_scale + c0 + sq * 4);
            ish = *reinterpret_cast<const float4 *>(in_shift + c0 + sq * 4);
        }
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int pix = spix0 + i * 32;
            if (pix < 180) {
                const int hy = pix / 18, hx = pix - hy * 18;
                const bool ok = (okmask >> i) & 1u;
                const float4 v = pv[i];
                const float4 o = ok ? make_float4(fmaf(v.x, isc.x, ish.x), fmaf(v.y, isc.y, ish.y), fmaf(v.z, isc.z, ish.z),
                                                  fmaf(v.w, isc.w, ish.w))
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(&buf[((hy * 2 + (hx & 1)) * WHALF + (hx >> 1)) * WAS + sq * 4]) = o;
            }
        }
    };
    float4 r0, r1, r2, r3;
    float4 da0, da1, da2, da3, db0, db1, db2, db3;      // raw patch pixels of the next 8-channel group
    auto issue_rows = [&](const float *As, int g) {
        const float *pa = As + offa + g * 8, *pb = As + offb + g * 8;
        da0 = *reinterpret_cast<const float4 *>(pa);
        da1 = *reinterpret_cast<const float4 *>(pa + J1);
        da2 = *reinterpret_cast<const float4 *>(pa + J2);
        da3 = *reinterpret_cast<const float4 *>(pa + J3);
        db0 = *reinterpret_cast<const float4 *>(pb);
        db1 = *reinterpret_cast<const float4 *>(pb + J1);
        db2 = *reinterpret_cast<const float4 *>(pb + J2);
        db3 = *reinterpret_cast<const float4 *>(pb + J3);
    };
    auto combine_rows = [&]() {                         // r[j] = d[ia][j] + sg d[ib][j]
        r0 = f4_fma(db0, sg, da0);
        r1 = f4_fma(db1, sg, da1);
        r2 = f4_fma(db2, sg, da2);
        r3 = f4_fma(db3, sg, da3);
    };

    // B fragments of the first 8-channel group
    float4 bq[4][NT];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            bq[v][nt] = *reinterpret_cast<const float4 *>(ubase + v * ustride_pos + (size_t)nt * nkg * 256);

    load_patch(0);
    store_patch(lds, 0);
    __syncthreads();

    // The main loop is a hand-placed software pipeline; __builtin_amdgcn_sched_barrier(0) pins it (left alone, the
    // scheduler sinks every prefetch to just above its use and exposes the L2 latency 16 times per chunk):
    //   step (g, v):  8 MFMAs on bq[v], then the loads that refill bq[v] for group g+1 (4 steps = 2048 matrix
    //   cycles ahead of their use); the next chunk's pixels are requested at the top of the chunk and written to the
    //   other LDS buffer at its end; the LDS reads of group g+1 are issued under the MFMAs of step (g, 3).
    for (int ch = 0; ch < nchunks; ++ch) {
        const float *As = lds + (ch & 1) * WPATCH;
        if (ch + 1 < nchunks) load_patch((ch + 1) * WKC);
        issue_rows(As, 0);
        combine_rows();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < WKC / 8; ++g) {
            const int kg = ch * (WKC / 8) + g;
            const int kgn = kg + 1 < nkg ? kg + 1 : kg;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 a = v == 0 ? f4_sub(r0, r2) : (v == 1 ? f4_add(r1, r2) : (v == 2 ? f4_sub(r2, r1) : f4_sub(r1, r3)));
                if (v == 3 && g + 1 < WKC / 8) {        // LDS reads of the next group go out ahead of this step's MFMAs
                    issue_rows(As, g + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[v][nt] = mfma32(a.x, bq[v][nt].x, acc[v][nt]);
                    acc[v][nt] = mfma32(a.y, bq[v][nt].y, acc[v][nt]);
                    acc[v][nt] = mfma32(a.z, bq[v][nt].z, acc[v][nt]);
                    acc[v][nt] = mfma32(a.w, bq[v][nt].w, acc[v][nt]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bq[v][nt] = *reinterpret_cast<const float4 *>(ubase + v * ustride_pos + ((size_t)nt * nkg + kgn) * 256);
                if (v == 3 && g + 1 < WKC / 8) {
                    __builtin_amdgcn_sched_barrier(0);
                    combine_rows();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ch + 1 < nchunks) store_patch(lds + ((ch + 1) & 1) * WPATCH, (ch + 1) * WKC);
        __syncthreads();
    }

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
#pragma unroll
    for (int it = 0; it < MPT; ++it) {
        const int m = m0 + it * (256 / C4);
        float4 P[4][2];
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                P[w][b] = *reinterpret_cast<const float4 *>(&Pb[((w * 2 + b) * 32 + m) * CBP + c4 * 4]);
        const int mr = m >> 3, mc = m & 7;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float4 v = a == 0 ? f4_add(f4_add(P[0][b], P[1][b]), P[2][b]) : f4_sub(f4_sub(P[1][b], P[2][b]), P[3][b]);
                const int gy = ty0 + 2 * mr + a, gx = tx0 + 2 * mc + b;
                if (gy < H && gx < W) {
                    const size_t o = (((size_t)n * H + gy) * W + gx) * Cout + co;
                    v = f4_add(v, bv);
                    if (addend) {
                        float4 ad = *reinterpret_cast<const float4 *>(addend + o);
                        if (addend_mask) {
                            const float4 mk = *reinterpret_cast<const float4 *>(addend_mask + o);
                            ad = make_float4(mk.x > 0.f ? ad.x : 0.f, mk.y > 0.f ? ad.y : 0.f, mk.z > 0.f ? ad.z : 0.f,
                                             mk.w > 0.f ? ad.w : 0.f);
                        }
                        v = f4_add(v, ad);
                    }
                    if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    *reinterpret_cast<float4 *>(y + o) = v;
                    if (stats) {
                        ssum = f4_add(ssum, v);
                        if (stat_aux) {
                            const float4 ax = *reinterpret_cast<const float4 *>(stat_aux + o);
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

// U = G g G^T in fragment order [16 pos][Cout/32][Cin/8][64 lanes][4]: lane (n, h) element j = U_pos[cin 8g+4h+j][cout 32cb+n].
// mode 0: forward filter g = w[cout][cin];  mode 1: data-gradient filter g[ky][kx] = w[k][n][2-ky][2-kx]
// (the GEMM's "cin" runs over the forward Cout and its "cout" over the forward, padded, Cin).
__global__ __launch_bounds__(256) void wino_pack_kernel(const float *__restrict__ w, float *__restrict__ u, int Cout_f,
                                                        int Cin_real, int K, int Nn, int mode) {
    // K = GEMM reduction channels, Nn = GEMM output channels
    const long total = (long)(Nn / 32) * (K / 8) * 256;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
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
    (void)Cout_f;
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

extern "C" int adyolo_wino_fwd(const float *x, const float *u, const float *bias, const float *addend,
                               const float *addend_mask, const float *in_scale, const float *in_shift, float *y,
                               float *stats, const float *stat_aux, const float *stat_mean, const float *stat_invstd,
                               int N, int H, int W, int Cin, int Cout, int relu, void *stream) {
    ADYOLO_REQUIRE(x && u && y && N > 0 && H > 0 && W > 0, ADYOLO_EINVAL, "wino_fwd: bad arguments");
    ADYOLO_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0 && Cin > 0 && Cout > 0, ADYOLO_ENOSUP,
                   "wino_fwd: Cin=%d and Cout=%d must be multiples of 32", Cin, Cout);
    ADYOLO_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (!addend_mask || addend), ADYOLO_EINVAL,
                   "wino_fwd: in_scale/in_shift come together; addend_mask needs addend");
    ADYOLO_REQUIRE(!stat_aux || (stats && stat_mean && stat_invstd), ADYOLO_EINVAL,
                   "wino_fwd: stat_aux needs stats, stat_mean and stat_invstd");
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
        hipLaunchKernelGGL((wino_fwd_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, x, u, bias, addend,
                           addend_mask, in_scale, in_shift, y, stats, stat_aux, stat_mean, stat_invstd, H, W, Cin, Cout,
                           tilesW, tilesH, nsp, ncb, xcd_div, relu);
    else
        hipLaunchKernelGGL((wino_fwd_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, x, u, bias, addend,
                           addend_mask, in_scale, in_shift, y, stats, stat_aux, stat_mean, stat_invstd, H, W, Cin, Cout,
                           tilesW, tilesH, nsp, ncb, xcd_div, relu);
    return check_launch("wino_fwd");
}
