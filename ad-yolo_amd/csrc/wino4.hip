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
#include "wino4_common.hpp"
#include "wino4p_launch.hpp"

namespace adyolo {
namespace w4 {

template <int TC, bool AFF>
__global__ __launch_bounds__(256, 1) void wino4_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ u, const float *__restrict__ bias,
    const float *__restrict__ addend, const float *__restrict__ addend_mask, const float *__restrict__ in_scale,
    const float *__restrict__ in_shift, float *__restrict__ y, float *__restrict__ stats,
    const float *__restrict__ stat_aux, const float *__restrict__ stat_mean, const float *__restrict__ stat_invstd,
    const float *__restrict__ stat_mask, int H, int W, int Cin, int Cout, int patchesW, int patchesH, int nsp, int ncb,
    int xcd_div, int relu, int mask_bits) {
    using C = Cfg<TC>;
    constexpr int PS = C::PS, CBUF = C::CBUF, RS = C::RS, PR = C::PR, CBP = C::CBP;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS + 2 * WMAXC];     // (one array: cdna_hip_programming.md 5, trap 4a)
    float *aff = lds + C::LDS_FLOATS;                     // producer BatchNorm scale | shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    if (AFF)
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

    // ---- GEMM-side constants.  Wave w: full column nuF (xi = 0..5 -> acc 0..5) and half column nuH (acc 6..8 = xi 0, 1, 2 for
    // hh = 0, xi 5, 3, 4 for hh = 1)
    const int nuF = wave == 0 ? 0 : wave == 1 ? 2 : wave == 2 ? 3 : 5;
    const int nuH = wave < 2 ? 1 : 4;
    const int hh = wave & 1;
    const float K2 = hh ? A2 : B2, KP = hh ? PB : PA;
    constexpr int ROW4 = 4 * TC * 16;
    // Everything below that is a per-lane constant of the whole kernel is computed ONCE and kept in registers (the 512-register
    // budget has room): on the fp32 MFMA a VALU instruction is not hidden under the matrix work, it takes ~5 cycles of the same
    // pipe (tools/micro/mfma32_coissue.hip), so address arithmetic in the loop is paid in full.
    // read offsets (bytes) of patch rows 4 tr + j (j = 0..3) of quad lh in the planes of nuF / nuH; rows 4, 5 = rows 0, 1 of the
    // next block, whose rotation is one more: row 4 tr + 4 sits at row 1's offset + 4 rows, row 4 tr + 5 at row 2's + 4 rows
    int oF[4], oH[3], oZ[3];
    {
        const int tr = li >> C::LOG_TC, tc = li & (TC - 1);
        int o_[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o_[j] = ((lh * PS + (4 * tr + ((tr + j) & 3)) * TC + tc) * 16);
        const int planeF = nuF * 2 * PS * 16, planeH = nuH * 2 * PS * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) oF[j] = planeF + o_[j];
#pragma unroll
        for (int j = 0; j < 3; ++j) oH[j] = planeH + o_[j + 1];
        // rows hh, hh + 2, hh + 4 of the half column (the single term's operands)
        oZ[0] = planeH + (hh ? o_[1] : o_[0]);
        oZ[1] = planeH + (hh ? o_[3] : o_[2]);
        oZ[2] = planeH + (hh ? o_[2] : o_[1]) + ROW4;
    }

    // ---- staging.  Item = (patch row y, tile column stc, channel quad sq4 of the 16-channel pair): six pixels -> six nu.
    // Full rounds: thread tid takes row (tid >> (2 + LOG_TC)) [+ RS in round 1]; leftover rows 2 RS ..: lanes of ONE wave.
    // Byte offsets of the six pixels inside the sample: out-of-image COLUMNS get 0x80000000 (added to any row offset of a sample
    // smaller than 2 GiB - 2 rows it stays beyond num_records: the buffer load returns 0); out-of-image ROWS need nothing: a
    // negative offset or one beyond the sample fails the range check by itself.
    const int rowb = W * Cin * 4, pixb = Cin * 4;
    const unsigned nrec = (unsigned)H * (unsigned)rowb;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x + (size_t)n * H * W * Cin), 0, (int)nrec, 0x00020000);
    int offF[6], offL[6], wbF0, wbF1, wbL;
    {
        auto item = [&](int i, int ybase, int (&off)[6], int &wb) {
            const int sq4 = i & 3, stc = (i >> 2) & (TC - 1), yy = ybase + (i >> (2 + C::LOG_TC));
            const int gy = ty0 + yy - 1, gx0 = tx0 + 4 * stc - 1;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                off[j] = (yy < PR && (unsigned)(gx0 + j) < (unsigned)W) ? gy * rowb + (gx0 + j) * pixb + sq4 * 16 : (int)0x80000000;
            const int q = yy >> 2;
            const int brow = 4 * q + (((yy & 3) + q) & 3);       // storage row: rotation inside 4-row blocks
            // buffer (sq4 >> 1) of the pair, plane (nu, sq4 & 1); -1: no such row
            wb = yy < PR ? (sq4 >> 1) * (CBUF * 4) + ((sq4 & 1) * PS + brow * TC + stc) * 16 : -1;
        };
        int dummy[6];
        item(tid, 0, offF, wbF0);
        item(tid, RS, dummy, wbF1);
        item(lane, 2 * RS, offL, wbL);
    }
    // round: 0, 1 (full rounds), 2 (leftover rows; `on`: this wave's turn, else every lane is sent out of range)
    auto st_load = [&](float4 (&p)[6], int round, int pr, bool on) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int vo = round == 0 ? offF[j] : round == 1 ? offF[j] + RS * rowb : (on ? offL[j] : (int)0x80000000);
            // whole-vector bit cast (a bit cast of ONE element makes hipcc narrow the load to a dword)
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, pr * 64, 0));
            p[j] = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    // second half of a staging round, in two parts so that the caller can place them under different MFMA steps:
    // st_xform: (producer affine on the in-image pixels,) six-point transform, in place;  st_write: six ds_write_b128
    auto st_xform = [&](float4 (&p)[6], int round, int pr) {
        if (AFF) {
            // x' = scale * x + shift on in-image pixels; out-of-image pixels were read as 0 and must stay 0
            const int sq4 = (round == 2 ? lane : tid) & 3;
            const float4 isc = *reinterpret_cast<const float4 *>(&aff[pr * 16 + sq4 * 4]);
            const float4 ish = *reinterpret_cast<const float4 *>(&aff[WMAXC + pr * 16 + sq4 * 4]);
            // (under an EXEC mask made by the range comparison instead of a compare and four selects per pixel: wino4_common.hpp)
            int vo[6];
            f32x4 q[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                vo[j] = round == 0 ? offF[j] : round == 1 ? offF[j] + RS * rowb : offL[j];
                q[j] = f32x4{p[j].x, p[j].y, p[j].z, p[j].w};
            }
            aff6_inrange(q, f32x4{isc.x, isc.y, isc.z, isc.w}, f32x4{ish.x, ish.y, ish.z, ish.w}, vo, nrec);
#pragma unroll
            for (int j = 0; j < 6; ++j) p[j] = make_float4(q[j][0], q[j][1], q[j][2], q[j][3]);
        }
        bt6(p, p);
    };
    auto st_write = [&](const float4 (&tt)[6], int round, float *Cn) {
        const int wb = round == 0 ? wbF0 : round == 1 ? wbF1 : wbL;
        if (round < 2 || wb >= 0) {
            char *dst = reinterpret_cast<char *>(Cn) + wb;
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<float4 *>(dst + j * 2 * PS * 16) = tt[j];
        }
    };
    auto st_store = [&](float4 (&p)[6], int round, float *Cn, int pr) {
        st_xform(p, round, pr);
        st_write(p, round, Cn);
    };

    const int nkg = Cin / 8, npairs = Cin / 16;
    const size_t ustride_pos = (size_t)(Cout / 32) * nkg * 256;               // floats per transform position
    // B fragments through a buffer descriptor: the lane's 16 bytes are the VGPR offset (one register for every load), the
    // fragment's position the SGPR offset -- no vector instruction per load (a global_load costs a v_or or a 64-bit vector add)
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(u), 0, (int)(36 * ustride_pos * 4), 0x00020000);
    const int ulane = lane * 16;
    const int uwave = (int)((((size_t)(wave * 9) * (Cout / 32) + (size_t)cb * 2) * nkg * 256) * 4);
    // B fragment of use uu = 2 s + nt of group kg
    auto bload = [&](int uu, int kg) {
        const int s = uu >> 1, nt = uu & 1;
        const int so = uwave + (int)((s * ustride_pos + ((size_t)nt * nkg + kg) * 256) * 4);
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, so, 0));
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    // register ring of BR fragments (BR = 9, 12 or 18 divides the 36 uses of a pair: step t uses slot t % BR in every pair)
    constexpr int BR = W4_BRING;
    float4 bq[BR];
#pragma unroll
    for (int uu = 0; uu < BR; ++uu) bq[uu] = bload(uu % 18, uu / 18 < nkg ? uu / 18 : nkg - 1);

    __syncthreads();                                      // affine table visible
    {                                                     // pair 0: all staging rounds in flight together
        float4 p0[6], p1[6];
        st_load(p0, 0, 0, true);
        st_load(p1, 1, 0, true);
        st_store(p0, 0, lds, 0);
        st_load(p0, 2, 0, wave == 0);                     // leftover rows
        st_store(p1, 1, lds, 0);
        if (wave == 0) st_store(p0, 2, lds, 0);
    }
    // two register sets for the pixels in flight: every staging load is requested >= 15 steps (3840 matrix cycles) before its use
    float4 pvA[6], pvB[6];
    st_load(pvA, 0, npairs > 1 ? 1 : 0, true);            // rounds 0 and 1 of pair 1
    st_load(pvB, 1, npairs > 1 ? 1 : 0, true);
    __syncthreads();

    f32x16 acc[9][2];
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][nt][r] = 0.f;

    // A fragments of one 8-channel group: 6 + 7 ds_read_b128, 48 + 24 VALU, in four separately placed parts.  bF / bH / bZ:
    // the pair's base addresses (read offsets + the pair's buffer), made once per pair; the pair's second group = + CBUF * 4
    float4 cF[6], cP[4], cZ[3];
    float4 a[9];                                          // ONE set: a fragment is overwritten for the next group once its MFMAs are issued
    int bF[4], bH[3], bZ[3];
    const char *ldsb = reinterpret_cast<const char *>(lds);
    auto a_reads_full = [&](int g) {
        cF[0] = *reinterpret_cast<const float4 *>(ldsb + bF[0] + g * (CBUF * 4));
        cF[1] = *reinterpret_cast<const float4 *>(ldsb + bF[1] + g * (CBUF * 4));
        cF[2] = *reinterpret_cast<const float4 *>(ldsb + bF[2] + g * (CBUF * 4));
        cF[3] = *reinterpret_cast<const float4 *>(ldsb + bF[3] + g * (CBUF * 4));
        cF[4] = *reinterpret_cast<const float4 *>(ldsb + bF[1] + g * (CBUF * 4) + ROW4);
        cF[5] = *reinterpret_cast<const float4 *>(ldsb + bF[2] + g * (CBUF * 4) + ROW4);
    };
    auto a_reads_half = [&](int g) {
        cP[0] = *reinterpret_cast<const float4 *>(ldsb + bH[0] + g * (CBUF * 4));
        cP[1] = *reinterpret_cast<const float4 *>(ldsb + bH[1] + g * (CBUF * 4));
        cP[2] = *reinterpret_cast<const float4 *>(ldsb + bH[2] + g * (CBUF * 4));
        cP[3] = *reinterpret_cast<const float4 *>(ldsb + bH[0] + g * (CBUF * 4) + ROW4);
        cZ[0] = *reinterpret_cast<const float4 *>(ldsb + bZ[0] + g * (CBUF * 4));
        cZ[1] = *reinterpret_cast<const float4 *>(ldsb + bZ[1] + g * (CBUF * 4));
        cZ[2] = *reinterpret_cast<const float4 *>(ldsb + bZ[2] + g * (CBUF * 4));
    };
    auto a_xform_full = [&]() {
        float4 tF[6];
        bt6(cF, tF);
#pragma unroll
        for (int s = 0; s < 6; ++s) a[s] = tF[s];
    };
    auto a_xform_half = [&]() {
        float4 tH[3];
        bt3(cP, cZ, tH, K2, KP);
#pragma unroll
        for (int s = 0; s < 3; ++s) a[6 + s] = tH[s];
    };

    // The pair loop is a hand-placed software pipeline of 36 steps (step = one accumulator tile's four MFMAs = 256 matrix
    // cycles), pinned with __builtin_amdgcn_sched_barrier(0): left alone, the scheduler sinks every load to just above its
    // use (first version: each B refill followed by its own vmcnt wait).  Per step: [side work + 4 MFMAs] | the step's loads.
    //   B ring: the fragment of step t + BR is requested right after step t (BR = 9: 2304 matrix cycles ahead);
    //   A fragments: fragments 0..5 (6..8) of the pair's second group are read at step 8 (16) and transformed under step 12
    //   (18), i.e. right after the first group's MFMAs on those registers are issued;
    //   staging of the next pair (two register sets A, B): round 0 (A) transformed under step 4, written under 6; round 1 (B)
    //   transformed under 12, written under 14; leftover rows requested at 8 (A), written (one wave) under 22; rounds 0 / 1 of
    //   the pair after requested at 25 (A) / 16 (B);
    //   the pair's ONE barrier sits after step 24, not at the end: every write of the next pair's image is done by then, and
    //   the buffers they went to were last read at step 16 of the pair before -- so the next pair's first A fragments are
    //   read at steps 28 / 32 and transformed under step 30 (full column; the half column follows under step 2 of the next
    //   pair), and the first MFMA of a pair never waits for an LDS round trip and a transform.
    auto pair_bases = [&](int pr) {
        const int pboff = (pr & 1) * (2 * CBUF * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) bF[j] = oF[j] + pboff;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            bH[j] = oH[j] + pboff;
            bZ[j] = oZ[j] + pboff;
        }
    };
    pair_bases(0);
    a_reads_full(0);
    a_xform_full();
    a_reads_half(0);
    for (int pr = 0; pr < npairs; ++pr) {
        float *Cn = lds + ((pr + 1) & 1) * (2 * CBUF);
        const int prn = pr + 1 < npairs ? pr + 1 : npairs - 1;                 // pair being staged (clamped)
        const int prn2 = pr + 2 < npairs ? pr + 2 : npairs - 1;
        const bool lwave = wave == (pr & 3);                                   // this wave stages the leftover rows of the pair
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kg = 2 * pr + half;
            const int kgn = kg + 1 < nkg ? kg + 1 : nkg - 1;
#pragma unroll
            for (int s = 0; s < 9; ++s) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int step = half * 18 + 2 * s + nt;
                    const int uu = 2 * s + nt, slot = step % BR;
                    // ---- side work of the step (VALU / LDS writes), free to interleave with its MFMAs
                    if (!(W4_WHATIF & 4) || pr == 0) {
                        if (step == 2) a_xform_half();
                        if (step == 12) a_xform_full();
                        if (step == 18) a_xform_half();
                        if (step == 30) a_xform_full();                        // the next pair's first group
                    }
                    if (!(W4_WHATIF & 1)) {
                        if (!(W4_WHATIF & 64)) {
                            if (step == 4) st_xform(pvA, 0, prn);
                            if (step == 12) st_xform(pvB, 1, prn);
                        }
                        if (!(W4_WHATIF & 128)) {
                            if (step == 6) st_write(pvA, 0, Cn);
                            if (step == 14) st_write(pvB, 1, Cn);
                        }
                        if (!(W4_WHATIF & 256))
                            if (step == 22 && lwave) st_store(pvA, 2, Cn, prn);
                    }
                    if (W4_WHATIF & 16) {
                        asm volatile("" : "+v"(a[s].x), "+v"(a[s].y), "+v"(a[s].z), "+v"(a[s].w));
                        asm volatile("" : "+v"(bq[slot].x), "+v"(bq[slot].y), "+v"(bq[slot].z), "+v"(bq[slot].w));
                    } else if (s < 8) {
                        // MFMAs are pure: instruction selection may emit them anywhere their operands allow, also on the far
                        // side of a sched_barrier (seen: the refills of twelve steps ahead of the first MFMA).  Two empty asm
                        // statements -- one defining an operand, one using the result -- tie them to their step.
                        asm volatile("" : "+v"(a[s].x));
                        acc[s][nt] = mfma32(a[s].x, bq[slot].x, acc[s][nt]);
                        acc[s][nt] = mfma32(a[s].y, bq[slot].y, acc[s][nt]);
                        acc[s][nt] = mfma32(a[s].z, bq[slot].z, acc[s][nt]);
                        acc[s][nt] = mfma32(a[s].w, bq[slot].w, acc[s][nt]);
                        asm volatile("" : "+a"(acc[s][nt]));
                    } else {
                        mfma32x4_vgpr(acc[s][nt], a[s], bq[slot]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- loads of the step
                    if (!(W4_WHATIF & 2)) {
                        // the use BR steps ahead (36 uses per pair): same group, the pair's other group, or the next pair's
                        const int v = step + BR;
                        const int kgv = 2 * pr + v / 18;
                        bq[slot] = bload(v % 18, kgv < nkg ? kgv : nkg - 1);
                    }
                    if (step == 24) {
                        __syncthreads();                                       // the next pair's image is complete
                        pair_bases(pr + 1);
                    }
                    if (!(W4_WHATIF & 4) || pr == 0) {
                        if (step == 8) a_reads_full(1);
                        if (step == 16) a_reads_half(1);
                        if (step == 28) a_reads_full(0);                       // (bases: the next pair's already)
                        if (step == 32) a_reads_half(0);
                    }
                    if (!(W4_WHATIF & 1)) {
                        if (!(W4_WHATIF & 32)) {
                            if (step == 25) st_load(pvA, 0, prn2, true);
                            if (step == 16) st_load(pvB, 1, prn2, true);
                        }
                        // the leftover rows: every wave issues the loads (all lanes out of range unless it is its turn), so that
                        // the compiler's vmcnt bookkeeping is the same on both sides of the branch under step 22
                        if (!(W4_WHATIF & (32 | 256)))
                            if (step == 8) st_load(pvA, 2, prn, lwave);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    __syncthreads();                                      // (the epilogue reuses the LDS image)

    // ---- epilogue.  xi-sum of A^T . A in registers: Q[p] (p = output row inside the tile) of the full column and the partial
    // one of the half column; nu-sum through LDS, TWO output rows p per round (2 x 72 KB): slot 2 w = full column of wave w,
    // 2 w + 1 = its half column
    if (W4_WHATIF & 8) {
        float sacc = 0.f;
#pragma unroll
        for (int s = 0; s < 9; ++s) sacc += acc[s][0][0] + acc[s][1][3];
        y[(size_t)blockIdx.x * 256 + tid] = sacc;
        return;
    }
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
    const bool rl = relu != 0;                            // ReLU branch-free: maximum + wave-uniform select (a maximum against a
                                                          // -inf floor, round 4, turned NaN results into -inf)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)n * H * W * Cout, 0, H * W * Cout * 4, 0x00020000);
    const float hc0 = hh ? 0.f : 1.f, hc3 = hh ? 1.f : 0.f, hk1 = hh ? PB : PA, hk2 = hh ? B2 : A2, hk3 = hh ? B3 : A3;
    // addresses / validity of the thread's pixels: tiles m0, m0 + 16 (it), output row p of the tile, columns b = 0..3
    const int etc0 = m0 & (TC - 1), etr0 = m0 >> C::LOG_TC;
    constexpr int DTR = 16 >> C::LOG_TC;                  // tile-row distance of tiles m0 and m0 + 16 (same tile column)
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        if (pp) __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mfma_row(r, lane);
                float q0, q1, q2, q3;
                at4(acc[0][nt][r], acc[1][nt][r], acc[2][nt][r], acc[3][nt][r], acc[4][nt][r], acc[5][nt][r], q0, q1, q2, q3);
                // the half column, branch-free: (acc 6, 7, 8) = xi (0, 1, 2) or (5, 3, 4); wave-uniform constants pick the terms
                const float hs = acc[7][nt][r] + acc[8][nt][r], hd = acc[7][nt][r] - acc[8][nt][r];
                const float h0 = fmaf(hc0, acc[6][nt][r], hs), h1 = hk1 * hd, h2 = hk2 * hs, h3 = fmaf(hk3, hd, hc3 * acc[6][nt][r]);
                float *d = &Pb[((wave * 2 + 0) * 32 + m) * CBP + nt * 32 + li];
                if (W4_WHATIF & 512) {
                    if (r == 0 && nt == 0) d[0] = acc[0][0][0] + acc[8][1][1];
                    continue;
                }
                d[0] = pp == 0 ? q0 : q2;
                d[32 * CBP] = pp == 0 ? h0 : h2;
                d[C::EXCH] = pp == 0 ? q1 : q3;
                d[C::EXCH + 32 * CBP] = pp == 0 ? h1 : h3;
            }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int p = 2 * pp + e;
            // One body per combination of (statistics, addend, addend mask), chosen by wave-uniform branches ONCE per output row:
            // inside the body there is no branch (a uniform branch per pixel and operand cost more than the arithmetic: the first
            // version of this epilogue took 12 us per workgroup), the fused operands of the row's 8 pixels are requested together
            // (one workgroup per CU: nobody else covers their latency), out-of-image pixels are stored to an out-of-range offset
            // of the output's buffer descriptor and counted with weight 0.
            auto body = [&](auto ST_, auto AD_, auto MK_) {
                constexpr bool ST = decltype(ST_)::value, AD = decltype(AD_)::value, MK = decltype(MK_)::value;
                int off[2][4];                                // byte offset of the pixel's channel quad inside the sample
                bool ok[2][4];
                float4 ad[2][4], ax[2][4];
                unsigned amk[2][4], smk[2][4];                // 4-bit keep masks of the addend / of the statistics
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int gy = ty0 + 4 * (etr0 + it * DTR) + p;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int gx = tx0 + 4 * etc0 + b;
                        ok[it][b] = gy < H && gx < W;
                        off[it][b] = (__mul24(__mul24(min(gy, H - 1), W) + min(gx, W - 1), Cout) + co) * 4;   // (full-rate 24-bit multiplies)
                    }
                }
                // Fused operands through buffer descriptors of the sample (round 4): a 32-bit offset per load instead of a 64-bit
                // multiply-add (quarter rate) per pixel and operand.  ReLU-mask bits: the float4 with index i inside the tensor owns bit
                // (i & 63) of the four 64-bit words at word (i >> 6) * 4; a sample starts on a word boundary (checked by the host:
                // H W Cout / 4 % 64 == 0), so inside the sample's descriptor the words sit at byte (q >> 6) * 32 for quad q of the sample
                // and the bit is fetched with 32-bit field extracts (the 64-bit variable shifts of mask_bits4 are two instructions each).
                const size_t sbase = (size_t)n * H * W * Cout;    // floats
                const int sbytes = H * W * Cout * 4;
                auto rsrc_of = [&](const float *ptr, int bytes) {
                    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ptr), 0, bytes, 0x00020000);
                };
                auto load4 = [&](const __amdgpu_buffer_rsrc_t &rs, int o_) {
                    const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o_, 0, 0));
                    return make_float4(t.x, t.y, t.z, t.w);
                };
                auto keep4 = [&](const float *mptr, bool bits, int o_) {
                    if (bits) {
                        const __amdgpu_buffer_rsrc_t rs = rsrc_of(mptr + (sbase >> 8) * 8, sbytes >> 5);      // (sbase / 4 / 64) * 4 words of 8 bytes
                        const int q = o_ >> 4;                                                            // float4 index inside the sample
                        const int wb = (q >> 6) * 32 + ((q >> 5) & 1) * 4, sh = q & 31;                        // byte of the low / high dword, bit
                        const u32x4_t lo = __builtin_amdgcn_raw_buffer_load_b128(rs, (q >> 6) * 32, 0, 0);   // words x, y
                        const u32x4_t hi = __builtin_amdgcn_raw_buffer_load_b128(rs, (q >> 6) * 32 + 16, 0, 0);   // words z, w
                        (void)wb;
                        const bool up = (q & 32) != 0;
                        const unsigned wx = up ? lo[1] : lo[0], wy = up ? lo[3] : lo[2], wz = up ? hi[1] : hi[0], ww = up ? hi[3] : hi[2];
                        return ((wx >> sh) & 1u) | (((wy >> sh) & 1u) << 1) | (((wz >> sh) & 1u) << 2) | (((ww >> sh) & 1u) << 3);
                    }
                    const float4 mk = load4(rsrc_of(mptr + sbase, sbytes), o_);
                    return (mk.x > 0.f ? 1u : 0u) | (mk.y > 0.f ? 2u : 0u) | (mk.z > 0.f ? 4u : 0u) | (mk.w > 0.f ? 8u : 0u);
                };
                if (AD) {
                    const __amdgpu_buffer_rsrc_t ars = rsrc_of(addend + sbase, sbytes);
#pragma unroll
                    for (int it = 0; it < 2; ++it)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            ad[it][b] = load4(ars, off[it][b]);
                    if (MK) {
                        const bool bits = (mask_bits & 1) != 0;
#pragma unroll
                        for (int it = 0; it < 2; ++it)
#pragma unroll
                            for (int b = 0; b < 4; ++b) amk[it][b] = keep4(addend_mask, bits, off[it][b]);
                    }
                }
                const bool has_smk = ST && stat_mask != nullptr, has_aux = ST && stat_aux != nullptr;
                if (has_smk) {
                    const bool bits = (mask_bits & 2) != 0;
#pragma unroll
                    for (int it = 0; it < 2; ++it)
#pragma unroll
                        for (int b = 0; b < 4; ++b) smk[it][b] = keep4(stat_mask, bits, off[it][b]);
                }
                if (has_aux) {
                    const __amdgpu_buffer_rsrc_t xrs_ = rsrc_of(stat_aux + sbase, sbytes);
#pragma unroll
                    for (int it = 0; it < 2; ++it)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            ax[it][b] = load4(xrs_, off[it][b]);
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int m_ = m0 + it * 16;
                    float4 S[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        S[k] = *reinterpret_cast<const float4 *>(&Pb[e * C::EXCH + (k * 32 + m_) * CBP + c4 * 4]);
                    // Q[p][nu]: nu0 = S0, nu1 = S1 + S3, nu2 = S2, nu3 = S4, nu4 = S5 + S7, nu5 = S6
                    const float4 n1 = f4_add(S[1], S[3]), n4 = f4_add(S[5], S[7]);
                    float4 Y[4];
                    at4(S[0].x, n1.x, S[2].x, S[4].x, n4.x, S[6].x, Y[0].x, Y[1].x, Y[2].x, Y[3].x);
                    at4(S[0].y, n1.y, S[2].y, S[4].y, n4.y, S[6].y, Y[0].y, Y[1].y, Y[2].y, Y[3].y);
                    at4(S[0].z, n1.z, S[2].z, S[4].z, n4.z, S[6].z, Y[0].z, Y[1].z, Y[2].z, Y[3].z);
                    at4(S[0].w, n1.w, S[2].w, S[4].w, n4.w, S[6].w, Y[0].w, Y[1].w, Y[2].w, Y[3].w);
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        float4 v = f4_add(Y[b], bv);
                        if (AD) {
                            float4 a_ = ad[it][b];
                            if (MK) {
                                const unsigned k = amk[it][b];
                                a_ = make_float4((k & 1u) ? a_.x : 0.f, (k & 2u) ? a_.y : 0.f, (k & 4u) ? a_.z : 0.f, (k & 8u) ? a_.w : 0.f);
                            }
                            v = f4_add(v, a_);
                        }
                        v = make_float4(rl ? fmaxf(v.x, 0.f) : v.x, rl ? fmaxf(v.y, 0.f) : v.y, rl ? fmaxf(v.z, 0.f) : v.z, rl ? fmaxf(v.w, 0.f) : v.w);
                        if (!(W4_WHATIF & 1024))
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), yrs,
                                                                   ok[it][b] ? off[it][b] : (int)0x80000000, 0, 0);
                        if (ST) {
                            // out-of-image pixels and masked-out components count 0
                            unsigned k = ok[it][b] ? 15u : 0u;
                            if (has_smk) k &= smk[it][b];
                            v = make_float4((k & 1u) ? v.x : 0.f, (k & 2u) ? v.y : 0.f, (k & 4u) ? v.z : 0.f, (k & 8u) ? v.w : 0.f);
                            ssum = f4_add(ssum, v);
                            float4 w_ = v;
                            if (has_aux) {
                                const float4 x_ = ax[it][b];
                                w_ = make_float4((x_.x - smean.x) * sinv.x, (x_.y - smean.y) * sinv.y, (x_.z - smean.z) * sinv.z,
                                                 (x_.w - smean.w) * sinv.w);
                            }
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
//   s < 6: (xi = s, nu = nuF(w)), nuF = 0, 2, 3, 5;   s >= 6: nu = nuH(w) = 1, 1, 4, 4 and xi = 0, 1, 2 (w even) or 5, 3, 4 (w odd).
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
            const int xi = s < 6 ? s : (wv & 1) ? (s == 6 ? 5 : s - 4) : s - 6;
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
static int g_wino4_last_form = 0;          // 1: the last adyolo_wino4_fwd launched the one-patch kernel, 2: the persistent one

extern "C" int adyolo_wino4_last_form(void) { return g_wino4_last_form; }

// The two environment switches adyolo_wino4_fwd consults, read ONCE (first launch) and again only by adyolo_reload_switches()
// (ops.reload_thresholds(): the host code and the library move together; round 5 ADVICE -- they used to be three getenv calls
// in every launch).  bit 0: persistent kernel, bit 1: narrow patches; -1: not read yet.
static int g_w4_switches = -1;
static int w4_read_switches() {
    const char *pe = getenv("ADYOLO_W4_PERSIST"), *ne = getenv("ADYOLO_W4_NARROW");
    g_w4_switches = ((pe && pe[0] == '0') ? 0 : 1) | ((ne && ne[0] == '0') ? 0 : 2);
    return g_w4_switches;
}
static inline int w4_switches() { return g_w4_switches >= 0 ? g_w4_switches : w4_read_switches(); }
extern "C" int adyolo_reload_switches(void) { return w4_read_switches(); }

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
    ADYOLO_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0 && Cin > 0 && Cout > 0 && Cin <= WMAXC, ADYOLO_ENOSUP,
                   "wino4_fwd: Cin=%d (<= 512) and Cout=%d must be multiples of 32", Cin, Cout);
    const int nb = Cout % 64 == 0 ? 2 : 1;             // 32-channel output blocks per workgroup (1: persistent kernel only)
    // (31-bit byte offsets inside a sample's buffer descriptors, input and output side; 24-bit pixel indices: __mul24)
    ADYOLO_REQUIRE((size_t)(H + 2) * W * Cin * 4 < ((size_t)1 << 31) && (size_t)H * W * Cout * 4 < ((size_t)1 << 31) &&
                       (long)H * W < (1L << 23),
                   ADYOLO_ENOSUP, "wino4_fwd: one sample (input and output) must stay below 2 GiB and 2^23 pixels");
    ADYOLO_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (!addend_mask || addend), ADYOLO_EINVAL,
                   "wino4_fwd: in_scale/in_shift come together; addend_mask needs addend");
    ADYOLO_REQUIRE(!stat_aux || (stats && stat_mean && stat_invstd), ADYOLO_EINVAL,
                   "wino4_fwd: stat_aux needs stats, stat_mean and stat_invstd");
    ADYOLO_REQUIRE(!stat_mask || stats, ADYOLO_EINVAL, "wino4_fwd: stat_mask needs stats");
    const int ncb = Cout / (32 * nb);
    // Narrow maps (W <= 8: the middle stages of the ResNet-Conformer, 800 frames x 4 or 8 bins): plain launches take patches ONE
    // or TWO tiles wide (128 x 4 / 64 x 8 pixels) on the persistent kernel instead of padding a 16-pixel-wide patch 4 or 2 times
    // over (ADYOLO_W4_NARROW=0: the 16-wide patch)
    const int sw = w4_switches();
    const bool narrow = W <= 8 && !stats && !addend && !addend_mask && !stat_aux && !stat_mask && !bias && !in_scale && nb == 2 &&
                        ncb <= 8 && 8 % ncb == 0 && (sw & 3) == 3;
    const int tc = narrow ? (W <= 4 ? 1 : 2) : wino4_tc(W), tr = 32 / tc;
    const int patchesW = cdiv(W, 4 * tc), patchesH = cdiv(H, 4 * tr);
    const int nsp = N * patchesH * patchesW;
    int xcd_div = 0, blocks = nsp * ncb;
    if (ncb <= 8 && 8 % ncb == 0) {
        xcd_div = 8 / ncb;
        blocks = cdiv(nsp, xcd_div) * 8;
    }
    hipStream_t st = as_stream(stream);
    // Persistent form (round 5, wino4p.hpp): one workgroup per CU walks the patches of its XCD slot.  Needs the XCD dealing
    // (Cout / 64 in {1, 2, 4, 8}), no bias, masks given as bits, and one of the operand combinations it is instantiated for (the
    // ones the SE-ResNet block launches); everything else, and ADYOLO_W4_PERSIST=0, takes the one-patch-per-workgroup kernel below.
    const int epi = (stats ? 1 : 0) | (addend ? 2 : 0) | (addend_mask ? 4 : 0) | (stat_aux ? 8 : 0) | (stat_mask ? 16 : 0);
    const bool bits_ok = (!addend_mask || (mask_bits & 1)) && (!stat_mask || (mask_bits & 2));
    if (xcd_div > 0 && bits_ok && !bias && (sw & 1) && (epi == 0 || epi == 1 || epi == 2 || epi == 9 || epi == 15 || epi == 27 || epi == 31)) {
        static int ncus = 0;
        if (ncus == 0) {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8)
                v = 256;
            ncus = v;
        }
        const int njs = cdiv(nsp, xcd_div);
        const int slots = njs < ncus / 8 ? njs : ncus / 8;
        w4::W4Launch a = {x, u, bias, addend, addend_mask, in_scale, in_shift, y, stats, stat_aux, stat_mean, stat_invstd, stat_mask,
                          H, W, Cin, Cout, patchesW, patchesH, nsp, ncb, xcd_div, relu, mask_bits, tc, slots * 8, nb, st};
        g_wino4_last_form = 2;
        switch (epi) {
            case 0: w4::launch_wino4p<0>(a); break;
            case 1: w4::launch_wino4p<1>(a); break;
            case 2: w4::launch_wino4p<2>(a); break;
            case 9: w4::launch_wino4p<9>(a); break;
            case 15: w4::launch_wino4p<15>(a); break;
            case 27: w4::launch_wino4p<27>(a); break;
            default: w4::launch_wino4p<31>(a); break;
        }
        return check_launch("wino4_fwd (persistent)");
    }
    ADYOLO_REQUIRE(nb == 2, ADYOLO_ENOSUP,
                   "wino4_fwd: Cout=%d (a multiple of 32 but not of 64) needs the persistent kernel: no bias, masks as bits, Cout / 32 in "
                   "{1, 2, 4, 8}, operand combination %d not built", Cout, epi);
    g_wino4_last_form = 1;
#define ADYOLO_WINO4_FWD(TC_, AFF_)                                                                                    \
    hipLaunchKernelGGL((w4::wino4_fwd_kernel<TC_, AFF_>), dim3((unsigned)blocks), dim3(256), 0, st, x, u, bias, addend,    \
                       addend_mask, in_scale, in_shift, y, stats, stat_aux, stat_mean, stat_invstd, stat_mask, H, W, Cin, \
                       Cout, patchesW, patchesH, nsp, ncb, xcd_div, relu, mask_bits)
    if (tc == 8) {
        if (in_scale) ADYOLO_WINO4_FWD(8, true); else ADYOLO_WINO4_FWD(8, false);
    } else {
        if (in_scale) ADYOLO_WINO4_FWD(4, true); else ADYOLO_WINO4_FWD(4, false);
    }
#undef ADYOLO_WINO4_FWD
    return check_launch("wino4_fwd");
}
