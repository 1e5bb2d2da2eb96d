// K2w4: 3x3 convolution (stride 1, pad 1), channels-last, as Winograd F(4x4, 3x3) on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32).  Same operator as conv.hip / wino.hip (nn.Conv2d at /root/reference/src/models/backbones/
// resnet.py:16,18 -- forward and data-gradient): 36 multiplies per 4x4 output tile and channel pair instead of 144
// (direct) or 64 (F(2x2,3x3)), i.e. 1.78x fewer matrix instructions than wino.hip:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 4x4 output tile, 6x6 input tile d, 3x3 filter g
// Interpolation points 0, +-3/4, +-3/2, infinity (not the textbook 0, +-1, +-2): every constant is a dyadic rational,
// exact in fp32, and the error against a float64 convolution is ~4x smaller (tools/wino4/numerics.py, scan_points.py).
// The 36 transform positions (xi, nu) are 36 independent GEMMs  M[pos][tile][cout] = sum_cin V[pos][tile][cin] U[pos][cin][cout].
//
// ONE workgroup (4 waves, one per SIMD) per CU on the 512-register budget: it owns 32 tiles (TR x TC tiles = 4 TR x 4 TC
// output pixels; TC = 4 for maps 16 pixels wide, 8 otherwise) x 64 output channels; wave w owns 9 positions -- one whole
// column nu of the 6 x 6 position grid and half of another -- for both 32-channel halves: 18 accumulator tiles = 288
// registers (the AccVGPR half of the file), which leaves the 256 architectural registers for operands in flight.  An A
// fragment feeds two MFMAs, and the patch is staged once per 64 output channels.  The input transform is separable:
//   * W direction while staging: a thread loads the six pixels of one (patch row, tile column, channel quad) straight from
//     global memory (buffer loads: out-of-image pixels are redirected out of range and read as 0, no masks), applies the
//     six-point transform and writes the six results to LDS -- there is no raw patch in LDS, only C[nu][y][tile column];
//     a producer BatchNorm affine is applied AFTER the transform (it is linear: scale * T(x) + shift * T(in-image mask));
//   * H direction in the GEMM waves: six ds_read_b128 of C (one per patch row of the lane's tile) + 48 VALU give the six
//     A fragments of a column nu for four k-steps; rows of a 4-row block are rotated by the block index, which puts every
//     16-lane ds_read_b128 group on 16 distinct 16-byte slots without padding (tools/wino4/formulas_check.py).
// C lives in four 8-channel buffers: a 16-channel pair is read while the next pair is staged, one barrier per pair
// (144 MFMAs per wave).  U = G g G^T is packed once per optimizer step in MFMA-fragment order and streamed from L2 through
// a nine-slot register ring (half a group = 2304 matrix cycles ahead).  Epilogue: the xi-sum of A^T . A in registers, the
// nu-sum through LDS in four rounds (one output row of the tiles per round), then bias / masked addend / ReLU / per-patch
// BatchNorm sums as in conv.hip.
#include "wino_common.hpp"

namespace adyolo {
namespace w4 {

constexpr float PA = 0.75f, PB = 1.5f;                 // interpolation points +-PA, +-PB (besides 0 and infinity)
constexpr float A2 = PA * PA, B2 = PB * PB, S2 = A2 + B2, P2 = A2 * B2;
constexpr float A3 = PA * PA * PA, B3 = PB * PB * PB;

template <int TC>
struct Cfg {
    static constexpr int TR = 32 / TC;                  // tile rows of a workgroup
    static constexpr int LOG_TC = TC == 4 ? 2 : 3;
    static constexpr int PR = 4 * TR + 2;               // patch rows
    static constexpr int PS = PR * TC + 2;              // plane stride in 16-byte slots (== 2 mod 8: conflict-free writes)
    static constexpr int CBS = 12 * PS + 4;             // slots per 8-channel buffer: 6 nu x 2 channel quads (== 4 mod 8)
    static constexpr int CBUF = CBS * 4;                // floats per buffer
    static constexpr int RS = 256 / (4 * TC);           // patch rows per full staging round (16 channels = 4 quads per pixel)
    static constexpr int CBP = 72;                      // epilogue exchange row (64 channels + pad)
    static constexpr int EXCH = 8 * 32 * CBP;
    static constexpr int LDS_FLOATS = 4 * CBUF > EXCH ? 4 * CBUF : EXCH;
};


// six-point data transform B^T along one direction, per component
#define ADYOLO_W4_BT(F)                                                                   \
    {                                                                                     \
        const float e12 = fmaf(-B2, c[2].F, c[4].F), o12 = fmaf(-B2, c[1].F, c[3].F);      \
        const float e34 = fmaf(-A2, c[2].F, c[4].F), o34 = fmaf(-A2, c[1].F, c[3].F);      \
        t[0].F = fmaf(P2, c[0].F, fmaf(-S2, c[2].F, c[4].F));                             \
        t[1].F = fmaf(PA, o12, e12);                                                      \
        t[2].F = fmaf(-PA, o12, e12);                                                     \
        t[3].F = fmaf(PB, o34, e34);                                                      \
        t[4].F = fmaf(-PB, o34, e34);                                                     \
        t[5].F = fmaf(P2, c[1].F, fmaf(-S2, c[3].F, c[5].F));                             \
    }
__device__ __forceinline__ void bt6(const float4 (&c)[6], float4 (&t)[6]) {
    ADYOLO_W4_BT(x) ADYOLO_W4_BT(y) ADYOLO_W4_BT(z) ADYOLO_W4_BT(w)
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
// half of it: rows xi = 0, 1, 2 from c0..c4 (hh = 0) or xi = 3, 4, 5 from c1..c5 (hh = 1); r[k] = c[hh + k]
#define ADYOLO_W4_BT3(F)                                                                  \
    if (hh == 0) {                                                                        \
        const float e12 = fmaf(-B2, r[2].F, r[4].F), o12 = fmaf(-B2, r[1].F, r[3].F);      \
        t[0].F = fmaf(P2, r[0].F, fmaf(-S2, r[2].F, r[4].F));                             \
        t[1].F = fmaf(PA, o12, e12);                                                      \
        t[2].F = fmaf(-PA, o12, e12);                                                     \
    } else {                                                                              \
        const float e34 = fmaf(-A2, r[1].F, r[3].F), o34 = fmaf(-A2, r[0].F, r[2].F);      \
        t[0].F = fmaf(PB, o34, e34);                                                      \
        t[1].F = fmaf(-PB, o34, e34);                                                     \
        t[2].F = fmaf(P2, r[0].F, fmaf(-S2, r[2].F, r[4].F));                             \
    }
__device__ __forceinline__ void bt3(const float4 (&r)[5], float4 (&t)[3], int hh) {
    ADYOLO_W4_BT3(x) ADYOLO_W4_BT3(y) ADYOLO_W4_BT3(z) ADYOLO_W4_BT3(w)
}
#undef ADYOLO_W4_BT3

// output transform A^T along one direction: y[p] = sum_k AT[p][k] m[k]
__device__ __forceinline__ void at4(float m0, float m1, float m2, float m3, float m4, float m5, float &y0, float &y1,
                                    float &y2, float &y3) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    y0 = m0 + s12 + s34;
    y1 = fmaf(PA, d12, PB * d34);
    y2 = fmaf(A2, s12, B2 * s34);
    y3 = fmaf(A3, d12, fmaf(B3, d34, m5));
}

template <int TC>
__global__ __launch_bounds__(256, 1) void wino4_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ u, const float *__restrict__ bias,
    const float *__restrict__ addend, const float *__restrict__ addend_mask, const float *__restrict__ in_scale,
    const float *__restrict__ in_shift, float *__restrict__ y, float *__restrict__ stats,
    const float *__restrict__ stat_aux, const float *__restrict__ stat_mean, const float *__restrict__ stat_invstd,
    const float *__restrict__ stat_mask, int H, int W, int Cin, int Cout, int patchesW, int patchesH, int nsp, int ncb,
    int xcd_div, int relu, int mask_bits) {
    using C = Cfg<TC>;
    constexpr int PS = C::PS, CBUF = C::CBUF, RS = C::RS, PR = C::PR, CBP = C::CBP;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS + 2 * WMAXC];     // (one array: see cdna_hip_programming.md 5, trap 4a)
    float *aff = lds + C::LDS_FLOATS;                     // producer BatchNorm scale | shift (1 | 0 if none)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const bool has_aff = in_scale != nullptr;
    if (has_aff)
        for (int c = tid; c < Cin; c += 256) {
            aff[c] = in_scale[c];
            aff[WMAXC + c] = in_shift[c];
        }
    // block -> (spatial patch, channel block), dealt to XCDs as in wino.hip: an XCD keeps one 64-channel slice of U in its L2
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
    sp = nsp - 1 - sp;                                   // last patch first (see wino.hip)
    int t = sp;
    const int pw = t % patchesW;
    t /= patchesW;
    const int ph = t % patchesH;
    const int n = t / patchesH;
    const int co0 = cb * 64;
    const int ty0 = ph * (4 * C::TR), tx0 = pw * (4 * TC);

    // ---- GEMM-side constants.  Wave w: full column nuF (xi = 0..5 -> acc 0..5) and half column nuH (xi = 3 hh + 0..2 -> acc 6..8)
    const int nuF = wave == 0 ? 0 : wave == 1 ? 2 : wave == 2 ? 3 : 5;
    const int nuH = wave < 2 ? 1 : 4;
    const int hh = wave & 1;
    const int tr = li >> C::LOG_TC, tc = li & (TC - 1);
    // read offsets (bytes) of patch rows 4 tr + j (j = 0..3) of plane (nu 0, quad lh); rows 4, 5 = rows 0, 1 of the next block,
    // whose rotation is one more: row 4 tr + 4 sits at o[1] + 4 rows, row 4 tr + 5 at o[2] + 4 rows
    int o_[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o_[j] = ((lh * PS + (4 * tr + ((tr + j) & 3)) * TC + tc) * 16);
    constexpr int ROW4 = 4 * TC * 16;
    const int planeF = nuF * 2 * PS * 16, planeH = nuH * 2 * PS * 16;      // wave-uniform byte offsets

    // ---- staging.  Item = (patch row y, tile column stc, channel quad sq4 of the 16-channel pair): six pixels -> six nu
    const int rowb = W * Cin * 4, pixb = Cin * 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x + (size_t)n * H * W * Cin), 0, H * W * Cin * 4, 0x00020000);
    auto opaque_zero = [&]() {
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        return z;
    };
    // idx = tid (full rounds: rows ybase + (tid >> (2 + LOG_TC))) or lane (the leftover rows, one wave)
    auto st_load = [&](float4 (&p)[6], int idx, int ybase, int pr) {
        const int i = idx + opaque_zero();
        const int sq4 = i & 3, stc = (i >> 2) & (TC - 1), yy = ybase + (i >> (2 + C::LOG_TC));
        const int gy = ty0 + yy - 1, gx0 = tx0 + 4 * stc - 1;
        const bool rowok = (unsigned)gy < (unsigned)H && yy < PR;
        const int vo = gy * rowb + gx0 * pixb + sq4 * 16;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const bool ok = rowok && (unsigned)(gx0 + j) < (unsigned)W;
            // (whole-vector bit cast: __builtin_bit_cast(float, v[i]) of one element makes hipcc narrow the load to one dword)
            // (the pixel's whole offset in the VGPR: the range check looks at it alone, and vo itself is negative at the image's
            //  first row / column)
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                          xrs, ok ? vo + j * pixb : (int)0x80000000, pr * 64, 0));
            p[j] = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    auto st_store = [&](const float4 (&p)[6], int idx, int ybase, float *Cn, int pr) {
        const int i = idx + opaque_zero();
        const int sq4 = i & 3, stc = (i >> 2) & (TC - 1), yy = ybase + (i >> (2 + C::LOG_TC));
        float4 tt[6];
        bt6(p, tt);
        if (has_aff) {
            // affine after the transform: scale * T(x) + shift * T(m), m = 1 on in-image pixels
            const int gy = ty0 + yy - 1, gx0 = tx0 + 4 * stc - 1;
            const bool rowok = (unsigned)gy < (unsigned)H;
            float m[6], tm[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) m[j] = (rowok && (unsigned)(gx0 + j) < (unsigned)W) ? 1.f : 0.f;
            bt6s(m, tm);
            const float4 isc = *reinterpret_cast<const float4 *>(&aff[pr * 16 + sq4 * 4]);
            const float4 ish = *reinterpret_cast<const float4 *>(&aff[WMAXC + pr * 16 + sq4 * 4]);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                tt[j].x = fmaf(tt[j].x, isc.x, ish.x * tm[j]);
                tt[j].y = fmaf(tt[j].y, isc.y, ish.y * tm[j]);
                tt[j].z = fmaf(tt[j].z, isc.z, ish.z * tm[j]);
                tt[j].w = fmaf(tt[j].w, isc.w, ish.w * tm[j]);
            }
        }
        if (yy < PR) {
            const int q = yy >> 2;
            const int brow = 4 * q + (((yy & 3) + q) & 3);       // storage row: rotation inside 4-row blocks
            // buffer (sq4 >> 1) of the pair, plane (nu, sq4 & 1)
            char *dst = reinterpret_cast<char *>(Cn) + (sq4 >> 1) * (CBUF * 4) + ((sq4 & 1) * PS + brow * TC + stc) * 16;
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<float4 *>(dst + j * 2 * PS * 16) = tt[j];
        }
    };

    const int nkg = Cin / 8, npairs = Cin / 16;
    const size_t ustride_pos = (size_t)(Cout / 32) * nkg * 256;               // floats per transform position
    const char *ubase = reinterpret_cast<const char *>(u + ((size_t)(wave * 9) * (Cout / 32) + (size_t)cb * 2) * nkg * 256);
    const unsigned ulane = lane * 16u;
    // B fragment of use uu = 2 s + nt of group kg
    auto bload = [&](int uu, int kg) {
        const int s = uu >> 1, nt = uu & 1;
        return *reinterpret_cast<const float4 *>(
            ubase + ((unsigned)((s * ustride_pos + ((size_t)nt * nkg + kg) * 256) * 4) + ulane));
    };
    // nine-slot ring: use uu of a group sits in slot uu % 9 (18 uses per group: the slot of a use is the same in every group)
    float4 bq[9];
#pragma unroll
    for (int uu = 0; uu < 9; ++uu) bq[uu] = bload(uu, 0);

    __syncthreads();                                      // affine table visible
    {                                                     // pair 0: all staging rounds in flight together
        float4 p0[6], p1[6];
        st_load(p0, tid, 0, 0);
        st_load(p1, tid, RS, 0);
        st_store(p0, tid, 0, lds, 0);
        st_load(p0, tid, 2 * RS, 0);                      // leftover rows (every wave redundantly here: prologue only)
        st_store(p1, tid, RS, lds, 0);
        st_store(p0, tid, 2 * RS, lds, 0);
    }
    float4 pv[6];
    st_load(pv, tid, 0, npairs > 1 ? 1 : 0);              // round 0 of pair 1
    __syncthreads();

    f32x16 acc[9][2];
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][nt][r] = 0.f;

    // A fragments of one 8-channel group from buffer Cs
    auto a_read_full = [&](float4 (&c)[6], const char *Cs) {
        const char *pf = Cs + planeF;
        c[0] = *reinterpret_cast<const float4 *>(pf + o_[0]);
        c[1] = *reinterpret_cast<const float4 *>(pf + o_[1]);
        c[2] = *reinterpret_cast<const float4 *>(pf + o_[2]);
        c[3] = *reinterpret_cast<const float4 *>(pf + o_[3]);
        c[4] = *reinterpret_cast<const float4 *>(pf + o_[1] + ROW4);
        c[5] = *reinterpret_cast<const float4 *>(pf + o_[2] + ROW4);
    };
    auto a_read_half = [&](float4 (&c)[5], const char *Cs) {     // rows hh .. hh + 4
        const char *ph_ = Cs + planeH;
        if (hh == 0) {
            c[0] = *reinterpret_cast<const float4 *>(ph_ + o_[0]);
            c[1] = *reinterpret_cast<const float4 *>(ph_ + o_[1]);
            c[2] = *reinterpret_cast<const float4 *>(ph_ + o_[2]);
            c[3] = *reinterpret_cast<const float4 *>(ph_ + o_[3]);
            c[4] = *reinterpret_cast<const float4 *>(ph_ + o_[1] + ROW4);
        } else {
            c[0] = *reinterpret_cast<const float4 *>(ph_ + o_[1]);
            c[1] = *reinterpret_cast<const float4 *>(ph_ + o_[2]);
            c[2] = *reinterpret_cast<const float4 *>(ph_ + o_[3]);
            c[3] = *reinterpret_cast<const float4 *>(ph_ + o_[1] + ROW4);
            c[4] = *reinterpret_cast<const float4 *>(ph_ + o_[2] + ROW4);
        }
    };

    float4 a0[9], a1[9];                                  // A fragments of the pair's two groups
    for (int pr = 0; pr < npairs; ++pr) {
        const char *Cs = reinterpret_cast<const char *>(lds) + (pr & 1) * (2 * CBUF * 4);
        float *Cn = lds + ((pr + 1) & 1) * (2 * CBUF);
        const int prn = pr + 1 < npairs ? pr + 1 : npairs - 1;                 // pair being staged (clamped)
        const int prn2 = pr + 2 < npairs ? pr + 2 : npairs - 1;
        const bool lwave = wave == (pr & 3);                                   // this wave stages the leftover rows of the pair
        {
            float4 cF[6], cH[5];
            a_read_full(cF, Cs);
            a_read_half(cH, Cs);
            float4 tF[6], tH[3];
            bt6(cF, tF);
            bt3(cH, tH, hh);
#pragma unroll
            for (int s = 0; s < 6; ++s) a0[s] = tF[s];
#pragma unroll
            for (int s = 0; s < 3; ++s) a0[6 + s] = tH[s];
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kg = 2 * pr + half;
            const int kgn = kg + 1 < nkg ? kg + 1 : nkg - 1;
            if (half == 0) {
                // A fragments of the second group, under the first group's MFMAs
                float4 cF[6], cH[5];
                a_read_full(cF, Cs + CBUF * 4);
                a_read_half(cH, Cs + CBUF * 4);
                float4 tF[6], tH[3];
                bt6(cF, tF);
                bt3(cH, tH, hh);
#pragma unroll
                for (int s = 0; s < 6; ++s) a1[s] = tF[s];
#pragma unroll
                for (int s = 0; s < 3; ++s) a1[6 + s] = tH[s];
            }
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                const float4 a = half == 0 ? a0[s] : a1[s];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int uu = 2 * s + nt, slot = uu % 9;
                    acc[s][nt] = mfma32(a.x, bq[slot].x, acc[s][nt]);
                    acc[s][nt] = mfma32(a.y, bq[slot].y, acc[s][nt]);
                    acc[s][nt] = mfma32(a.z, bq[slot].z, acc[s][nt]);
                    acc[s][nt] = mfma32(a.w, bq[slot].w, acc[s][nt]);
                    bq[slot] = uu + 9 < 18 ? bload(uu + 9, kg) : bload(uu - 9, kgn);
                }
                if (half == 0 && s == 4) {
                    st_store(pv, tid, 0, Cn, prn);
                    st_load(pv, tid, RS, prn);
                }
                if (half == 1 && s == 0) {
                    st_store(pv, tid, RS, Cn, prn);
                    if (lwave) st_load(pv, lane, 2 * RS, prn);
                }
                if (half == 1 && s == 6) {
                    if (lwave) st_store(pv, lane, 2 * RS, Cn, prn);
                    st_load(pv, tid, 0, prn2);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue.  xi-sum of A^T . A in registers: Q[p] (p = output row inside the tile) of the full column and the partial
    // one of the half column; nu-sum through LDS, one output row p per round: slot 2 w = full column of wave w, 2 w + 1 = half
    float *Pb = lds;
    constexpr int C4 = 16;                                // float4 pieces per pixel (64 channels)
    const int c4 = tid % C4, m0 = tid / C4;               // epilogue thread: tiles m0 and m0 + 16, channel quad c4
    const int co = co0 + c4 * 4;
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = ssum;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), smean = bv, sinv = bv;
    if (bias) bv = *reinterpret_cast<const float4 *>(bias + co);
    if (stat_aux) {
        smean = *reinterpret_cast<const float4 *>(stat_mean + co);
        sinv = *reinterpret_cast<const float4 *>(stat_invstd + co);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (p) __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mfma_row(r, lane);
                float q0, q1, q2, q3;
                at4(acc[0][nt][r], acc[1][nt][r], acc[2][nt][r], acc[3][nt][r], acc[4][nt][r], acc[5][nt][r], q0, q1, q2, q3);
                const float qf = p == 0 ? q0 : p == 1 ? q1 : p == 2 ? q2 : q3;
                float h0, h1, h2, h3;
                if (hh == 0) at4(acc[6][nt][r], acc[7][nt][r], acc[8][nt][r], 0.f, 0.f, 0.f, h0, h1, h2, h3);
                else at4(0.f, 0.f, 0.f, acc[6][nt][r], acc[7][nt][r], acc[8][nt][r], h0, h1, h2, h3);
                const float qh = p == 0 ? h0 : p == 1 ? h1 : p == 2 ? h2 : h3;
                Pb[((wave * 2 + 0) * 32 + m) * CBP + nt * 32 + li] = qf;
                Pb[((wave * 2 + 1) * 32 + m) * CBP + nt * 32 + li] = qh;
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int m_ = m0 + it * 16;
            const int etr = m_ >> C::LOG_TC, etc = m_ & (TC - 1);
            float4 S[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) S[k] = *reinterpret_cast<const float4 *>(&Pb[(k * 32 + m_) * CBP + c4 * 4]);
            // Q[p][nu]: nu0 = S0, nu1 = S1 + S3, nu2 = S2, nu3 = S4, nu4 = S5 + S7, nu5 = S6
            const float4 n1 = f4_add(S[1], S[3]), n4 = f4_add(S[5], S[7]);
            float4 Y[4];
            at4(S[0].x, n1.x, S[2].x, S[4].x, n4.x, S[6].x, Y[0].x, Y[1].x, Y[2].x, Y[3].x);
            at4(S[0].y, n1.y, S[2].y, S[4].y, n4.y, S[6].y, Y[0].y, Y[1].y, Y[2].y, Y[3].y);
            at4(S[0].z, n1.z, S[2].z, S[4].z, n4.z, S[6].z, Y[0].z, Y[1].z, Y[2].z, Y[3].z);
            at4(S[0].w, n1.w, S[2].w, S[4].w, n4.w, S[6].w, Y[0].w, Y[1].w, Y[2].w, Y[3].w);
            const int gy = ty0 + 4 * etr + p;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int gx = tx0 + 4 * etc + b;
                float4 v = Y[b];
                if (gy < H && gx < W) {
                    const size_t o = (((size_t)n * H + gy) * W + gx) * Cout + co;
                    v = f4_add(v, bv);
                    if (addend) {
                        float4 ad = *reinterpret_cast<const float4 *>(addend + o);
                        if (addend_mask) {
                            bool kx, ky, kz, kw;
                            if (mask_bits & 1) {
                                mask_bits4(reinterpret_cast<const unsigned long long *>(addend_mask), o >> 2, kx, ky, kz, kw);
                            } else {
                                const float4 mk = *reinterpret_cast<const float4 *>(addend_mask + o);
                                kx = mk.x > 0.f; ky = mk.y > 0.f; kz = mk.z > 0.f; kw = mk.w > 0.f;
                            }
                            ad = make_float4(kx ? ad.x : 0.f, ky ? ad.y : 0.f, kz ? ad.z : 0.f, kw ? ad.w : 0.f);
                        }
                        v = f4_add(v, ad);
                    }
                    if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    *reinterpret_cast<float4 *>(y + o) = v;
                    if (stats) {
                        if (stat_mask) {
                            bool kx, ky, kz, kw;
                            if (mask_bits & 2) {
                                mask_bits4(reinterpret_cast<const unsigned long long *>(stat_mask), o >> 2, kx, ky, kz, kw);
                            } else {
                                const float4 mk = *reinterpret_cast<const float4 *>(stat_mask + o);
                                kx = mk.x > 0.f; ky = mk.y > 0.f; kz = mk.z > 0.f; kw = mk.w > 0.f;
                            }
                            v = make_float4(kx ? v.x : 0.f, ky ? v.y : 0.f, kz ? v.z : 0.f, kw ? v.w : 0.f);
                        }
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
    }
    if (stats) {
        // per-patch, per-channel sums of the stored output, layout [2][patches][Cout] (see conv.hip)
        __syncthreads();
        float *red = lds;                                 // [2][16 thread groups][64]
        *reinterpret_cast<float4 *>(&red[(0 * 16 + m0) * 64 + c4 * 4]) = ssum;
        *reinterpret_cast<float4 *>(&red[(1 * 16 + m0) * 64 + c4 * 4]) = ssq;
        __syncthreads();
        if (tid < 128) {
            const int c = tid & 63, which = tid >> 6;
            float s = 0.f;
#pragma unroll 8
            for (int gI = 0; gI < 16; ++gI) s += red[(which * 16 + gI) * 64 + c];
            stats[(size_t)which * nsp * Cout + (size_t)sp * Cout + co0 + c] = s;
        }
    }
}

// U = G g G^T (6 x 6 positions, computed in double) in fragment order [36 P][Cout/32][Cin/8][64 lanes][4]: lane (n, h) element j
// = U_P[cin 8 g + 4 h + j][cout 32 cb + n]; P = 9 w + s is the position owned by accumulator s of wave w:
//   s < 6: (xi = s, nu = nuF(w)), nuF = 0, 2, 3, 5;   s >= 6: (xi = 3 (w & 1) + s - 6, nu = nuH(w)), nuH = 1, 1, 4, 4.
// mode 0: forward filter g = w[cout][cin];  mode 1: data-gradient filter g[ky][kx] = w[k][n][2-ky][2-kx]
__device__ __forceinline__ void wino4_pack_one(const float *__restrict__ w, float *__restrict__ u, int Cin_real, int K,
                                               int Nn, int mode, long idx, long total) {
    const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const long rest = idx >> 8;
    const int g = (int)(rest % (K / 8)), cbk = (int)(rest / (K / 8));
    const int k = g * 8 + (lane >> 5) * 4 + j, nn = cbk * 32 + (lane & 31);
    double f[3][3];
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
            f[a][b] = (double)v;
        }
    const double a_ = 0.75, b_ = 1.5, a2 = a_ * a_, b2 = b_ * b_;
    const double na = 2.0 * a2 * (a2 - b2), nb = 2.0 * b2 * (b2 - a2);
    const double G[6][3] = {{1.0 / (a2 * b2), 0.0, 0.0}, {1.0 / na, a_ / na, a2 / na}, {1.0 / na, -a_ / na, a2 / na},
                            {1.0 / nb, b_ / nb, b2 / nb}, {1.0 / nb, -b_ / nb, b2 / nb}, {0.0, 0.0, 1.0}};
    double tt[6][3];                                      // t = G f
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
#pragma unroll
        for (int b = 0; b < 3; ++b) tt[xi][b] = G[xi][0] * f[0][b] + G[xi][1] * f[1][b] + G[xi][2] * f[2][b];
#pragma unroll
    for (int wv = 0; wv < 4; ++wv)
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int nu = s < 6 ? (wv == 0 ? 0 : wv == 1 ? 2 : wv == 2 ? 3 : 5) : (wv < 2 ? 1 : 4);
            const int xi = s < 6 ? s : 3 * (wv & 1) + s - 6;
            const double v = tt[xi][0] * G[nu][0] + tt[xi][1] * G[nu][1] + tt[xi][2] * G[nu][2];
            u[(size_t)(wv * 9 + s) * (size_t)total + idx] = (float)v;
        }
}

__global__ __launch_bounds__(256) void wino4_pack_kernel(const float *__restrict__ w, float *__restrict__ u, int Cin_real,
                                                         int K, int Nn, int mode) {
    const long total = (long)(Nn / 32) * (K / 8) * 256;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) wino4_pack_one(w, u, Cin_real, K, Nn, mode, idx, total);
}

// table: [n][8] int64 = {w, u_fwd or 0, u_dgrad or 0, Cout, Cin_real, Cin, unused, unused};  grid (ceil(largest total / 256), n)
__global__ __launch_bounds__(256) void wino4_pack_many_kernel(const long long *__restrict__ table) {
    const long long *d = table + 8 * blockIdx.y;
    const float *w = reinterpret_cast<const float *>(d[0]);
    float *uf = reinterpret_cast<float *>(d[1]), *ud = reinterpret_cast<float *>(d[2]);
    const int Cout = (int)d[3], Cin_real = (int)d[4], Cin = (int)d[5];
    const long total = (long)(Cout / 32) * (Cin / 8) * 256;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    if (uf) wino4_pack_one(w, uf, Cin_real, Cin, Cout, 0, idx, total);
    if (ud) wino4_pack_one(w, ud, Cin_real, Cout, Cin, 1, idx, total);
}

}  // namespace w4
}  // namespace adyolo

using namespace adyolo;

static inline int wino4_tc(int W) { return W >= 32 ? 8 : 4; }

extern "C" int adyolo_wino4_tiles(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return ADYOLO_EINVAL;
    const int tc = wino4_tc(W), tr = 32 / tc;
    return N * cdiv(H, 4 * tr) * cdiv(W, 4 * tc);
}

extern "C" int adyolo_wino4_pack_w(const float *w, float *u_fwd, float *u_dgrad, int Cout, int Cin_real, int Cin,
                                   void *stream) {
    ADYOLO_REQUIRE(w && (u_fwd || u_dgrad) && Cout > 0 && Cin_real > 0 && Cin >= Cin_real, ADYOLO_EINVAL,
                   "wino4_pack_w: bad arguments");
    ADYOLO_REQUIRE(Cout % 32 == 0 && Cin % 32 == 0, ADYOLO_ENOSUP,
                   "wino4_pack_w: Cin=%d and Cout=%d must be multiples of 32", Cin, Cout);
    const long total = (long)(Cout / 32) * (Cin / 8) * 256;
    if (u_fwd)
        hipLaunchKernelGGL(w4::wino4_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, u_fwd,
                           Cin_real, Cin, Cout, 0);
    if (u_dgrad)
        hipLaunchKernelGGL(w4::wino4_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, u_dgrad,
                           Cin_real, Cout, Cin, 1);
    return check_launch("wino4_pack_w");
}

extern "C" int adyolo_wino4_pack_many(const int64_t *table, int n, int max_cout, int max_cin, void *stream) {
    ADYOLO_REQUIRE(table && n > 0 && max_cout > 0 && max_cin > 0 && max_cout % 32 == 0 && max_cin % 32 == 0, ADYOLO_EINVAL,
                   "wino4_pack_many: bad arguments");
    const long total = (long)(max_cout / 32) * (max_cin / 8) * 256;
    hipLaunchKernelGGL(w4::wino4_pack_many_kernel, dim3(cdiv(total, 256), n), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const long long *>(table));
    return check_launch("wino4_pack_many");
}

extern "C" int adyolo_wino4_fwd(const float *x, const float *u, const float *bias, const float *addend,
                                const float *addend_mask, const float *in_scale, const float *in_shift, float *y,
                                float *stats, const float *stat_aux, const float *stat_mean, const float *stat_invstd,
                                const float *stat_mask, int N, int H, int W, int Cin, int Cout, int relu, int mask_bits,
                                void *stream) {
    ADYOLO_REQUIRE(x && u && y && N > 0 && H > 0 && W > 0, ADYOLO_EINVAL, "wino4_fwd: bad arguments");
    ADYOLO_REQUIRE(!(mask_bits & ~3) && (!mask_bits || ((long)H * W * (Cout / 4)) % 64 == 0), ADYOLO_ENOSUP,
                   "wino4_fwd: mask bits need H*W*Cout/4 %% 64 == 0");
    ADYOLO_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0 && Cin > 0 && Cout > 0 && Cin <= WMAXC, ADYOLO_ENOSUP,
                   "wino4_fwd: Cin=%d (<= 512) must be a multiple of 32 and Cout=%d of 64", Cin, Cout);
    ADYOLO_REQUIRE((size_t)(H + 2) * W * Cin * 4 < ((size_t)1 << 31), ADYOLO_ENOSUP, "wino4_fwd: one sample must stay below 2 GiB");
    ADYOLO_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (!addend_mask || addend), ADYOLO_EINVAL,
                   "wino4_fwd: in_scale/in_shift come together; addend_mask needs addend");
    ADYOLO_REQUIRE(!stat_aux || (stats && stat_mean && stat_invstd), ADYOLO_EINVAL,
                   "wino4_fwd: stat_aux needs stats, stat_mean and stat_invstd");
    ADYOLO_REQUIRE(!stat_mask || stats, ADYOLO_EINVAL, "wino4_fwd: stat_mask needs stats");
    const int tc = wino4_tc(W), tr = 32 / tc;
    const int patchesW = cdiv(W, 4 * tc), patchesH = cdiv(H, 4 * tr);
    const int nsp = N * patchesH * patchesW;
    const int ncb = Cout / 64;
    int xcd_div = 0, blocks = nsp * ncb;
    if (ncb <= 8 && 8 % ncb == 0) {
        xcd_div = 8 / ncb;
        blocks = cdiv(nsp, xcd_div) * 8;
    }
    hipStream_t st = as_stream(stream);
#define ADYOLO_WINO4_FWD(TC_)                                                                                          \
    hipLaunchKernelGGL((w4::wino4_fwd_kernel<TC_>), dim3((unsigned)blocks), dim3(256), 0, st, x, u, bias, addend,          \
                       addend_mask, in_scale, in_shift, y, stats, stat_aux, stat_mean, stat_invstd, stat_mask, H, W, Cin, \
                       Cout, patchesW, patchesH, nsp, ncb, xcd_div, relu, mask_bits)
    if (tc == 8) ADYOLO_WINO4_FWD(8); else ADYOLO_WINO4_FWD(4);
#undef ADYOLO_WINO4_FWD
    return check_launch("wino4_fwd");
}
