// K2w4p: the PERSISTENT form of the F(4x4,3x3) forward / data-gradient kernel (see wino4.hip for the algorithm and the
// one-patch-per-workgroup form; nn.Conv2d at /root/reference/src/models/backbones/resnet.py:16,18).  Round 5; the default
// wherever it is instantiated (wino4p_launch.hpp), ADYOLO_W4_PERSIST=0 selects the one-patch form.
// One workgroup per CU walks its share of the patches of ONE 64-channel block (the U slice, the B ring and the XCD's L2
// contents stay what they are); with one workgroup per CU nothing else hides a workgroup's prologue and epilogue, so
//   * the software pipeline of the pair loop simply continues across patches: during the LAST pair of a patch the staging
//     half of the pipeline loads / transforms / writes pair 0 of the NEXT patch (the one-patch form staged a clamped copy of
//     the last pair there), and pair 1 of the next patch and its first B fragments are requested from inside the epilogue --
//     the global-memory latency, the W-transform and the LDS round trip of a patch's first pair, and the launch of a
//     workgroup, are paid once per workgroup instead of once per patch;
//   * the epilogue no longer transforms in the accumulator-owning wave: the RAW accumulators go to LDS straight from the
//     AccVGPRs (ds_write_b32 takes an AGPR data operand: no v_accvgpr_read, no VALU at all on the writer side; the registers
//     are zeroed for the next patch while the stores drain), four rounds of [36 positions][8 accumulator rows][64 lanes]
//     (73.7 KB, beside the 56 KB image of the next patch's pair 0), and the reader does the whole separable output transform
//     A^T M A for its (tile, channel quad, output-row parity): rows 0 / 2 of A^T need only the sums m1 + m2, m3 + m4, rows 1 / 3
//     only the differences, so splitting a tile's work by row parity between two threads duplicates no arithmetic.  A thread
//     reads 30 float4 (lanes of a 16-lane ds_read_b128 group cover one 256-byte row: conflict-free), does 240 transform
//     instructions and finishes 8 pixels x 4 channels per round.  Old epilogue: 576 accumulator reads + ~1700 VALU + 256
//     ds_write_b32 per lane on the writer side alone.
#pragma once
#include "wino4_common.hpp"
#include "wino4p_launch.hpp"

#ifndef W4P_TIMING
#define W4P_TIMING 0      // 1: one workgroup writes s_memtime stamps of its first patches to the stat_mean pointer (results stay valid
                          // for plain launches; tools/wino4/persist_timing.py)
#endif
#ifndef W4P_HOIST
#define W4P_HOIST 1       // 0: the next patch's first requests from inside the last epilogue round (the first form), 1: before the epilogue's
                          // first store unless an addend AND the BatchNorm-input statistics are fused (no registers for it there), 2: always
#endif
#ifndef W4P_ST_AUX
#define W4P_ST_AUX 2      // cache-policy bits of the output stores: 2 = non-temporal (the 0.3-1.3 GB outputs are read by the NEXT kernel:
                          // no cache holds them; per launch -1 %, step -0.65 ms with the consumers' lost hits: profiles/r05_nt_ab.txt)
#endif
#ifndef W4P_X_AUX
#define W4P_X_AUX 0       // ... of the staged input's loads (A/B switch)
#endif
#ifndef W4P_OP_AUX
#define W4P_OP_AUX 2      // ... of the fused operands' loads (addend, statistics input: each read once): non-temporal since round 6 --
                          // they no longer push the staged input's half-read lines out of L2 (-5.5 % reads at stage 1, data-gradient
                          // launches -1.3 to -2.5 % at stages 1-3; profiles/r06_w4p_opaux_ab.txt; no gain when round 5 tried it)
#endif
#ifndef W4P_STAGGER
#define W4P_STAGGER 0     // experiment (round 6): workgroups with an odd slot start this many cycles late, so that the epilogues of the
                          // CUs of an XCD (32 KB store bursts per round) do not coincide (profiles/r06_store_burst.txt)
#endif
#ifndef W4P_AFF_EXEC
#define W4P_AFF_EXEC 1    // 1: the producer's BatchNorm affine is applied under an EXEC mask made by the range comparison (aff6_inrange),
                          // 0: compare + four selects of the shift per pixel (until round 6)
#endif
#ifndef W4P_BRES
#define W4P_BRES 1        // 1: 32 -> 32 layers take the resident-U form of the kernel (BRES below), 0: the B ring as everywhere else
#endif
#ifndef W4P_MKDEDUP
#define W4P_MKDEDUP 1     // 1: the epilogue's ReLU-mask bit loads shared between the pixels of a row segment where Cout allows (32 / 64 / 128)
#endif
#ifndef W4P_REQ_FENCE
#define W4P_REQ_FENCE 0   // 1: a compiler barrier behind the step-16 requests of the per-pair bodies (BRES, FULL), which otherwise drift down
                          // to ~step 23 -- measured neutral (+-0.5 %, profiles/r06_w4p_leftpin_ab.txt): off
#endif
#ifndef W4P_SWAP1
#define W4P_SWAP1 1       // 1: BRES stages round 1 before round 0 in a patch's second pair (see pair_body)
#endif
#ifndef W4P_LEFT_PIN
#define W4P_LEFT_PIN 1    // 1: the leftover-row data get an (empty) unconditional use in front of `if (lwave)`: see pair_body, step 22
#endif
#ifndef W4P_FULL
#define W4P_FULL 1        // 1: 32 -> 32 layers with operand sets 15 / 27 / 31 request both halves of the next patch's input lines together
#endif
#ifndef W4P_BRES_ACC
#define W4P_BRES_ACC 3    // channel groups (of the four) of the resident U kept in AccVGPRs
#endif
#ifndef W4P_STAT_EXEC
#define W4P_STAT_EXEC 1   // 1: the epilogue's BatchNorm-statistics sums skip out-of-image pixels through EXEC (stat_acc_inimage)
                          // except with both kinds of mask bits (operand set 31: measured 0.3-1.3 % slower there, set 27 4-5 % faster,
                          // the others 0-1.5 % faster: profiles/r06_w4p_exec_masks.txt), 0: four selects per pixel (until round 6)
#endif
#ifndef W4P_KILLDUP
#define W4P_KILLDUP 1     // 1 (2: and its B fragments past the end -- measured no better, profiles/r06_w4p_killdup_ab.txt): the last pair's requests for pair 1 of the NEXT patch (issued for the schedule's sake, never used, issued again
                          // from the epilogue) fetch nothing; 0: they are real loads (until round 6: "by then they are L2 hits" -- at stage 1
                          // they were not: profiles/r06_w4p_fetch_excess.txt)
#endif
#ifndef W4P_WHATIF
#define W4P_WHATIF 0      // timing-only builds (results invalid): bit 0 no reader half of the epilogue rounds, 1 no writer half, 2 no
                          // output stores, 3 no xi pass (LDS reads), 4 no MFMAs, 5 no nu pass, 6 no B refills in the pair loop, 7 no staging
                          // loads in it (profiles/r05_w4p_whatif_loads.txt), 8 no ReLU-mask bit loads in the epilogue
#endif

namespace adyolo {
namespace w4 {

template <int TC>
struct CfgP {
    using C = Cfg<TC>;
    static constexpr int XOFF = 2 * C::CBUF;              // floats: the exchange region starts at image buffer 2
    static constexpr int XCH = 36 * 8 * 64;               // floats per exchange round
    static constexpr int LDS_FLOATS = (XOFF + XCH > 4 * C::CBUF) ? XOFF + XCH : 4 * C::CBUF;
};

// Byte offsets of the pointer arguments the epilogue reads back from the kernel-argument segment (``karg`` below: an opaque
// load instead of a value that would stay live across the pair loop).  They are CHECKED against the kernel's real signature by
// the static_asserts after the kernel (KargLayout walks the parameter types with their alignments): adding, removing or
// reordering a parameter without moving these fails to compile (round 5, ADVICE).
enum : int {
    KA_ADDEND = 24, KA_ADDEND_MASK = 32, KA_Y = 56, KA_STATS = 64, KA_STAT_AUX = 72, KA_STAT_MEAN = 80, KA_STAT_INVSTD = 88,
    KA_STAT_MASK = 96
};
template <typename F>
struct KargLayout;
template <typename... A>
struct KargLayout<void (*)(A...)> {
    static constexpr int count = (int)sizeof...(A);
    static constexpr size_t offset(int idx) {
        constexpr size_t sz[] = {sizeof(A)...}, al[] = {alignof(A)...};
        size_t off = 0;
        for (int i = 0; i <= idx; ++i) {
            off = (off + al[i] - 1) / al[i] * al[i];
            if (i < idx) off += sz[i];
        }
        return off;
    }
};

// NB: 32-channel output blocks per workgroup -- 2 (Cout % 64 == 0) or 1 (32-channel layers: half the MFMAs per A fragment and per
// staged pixel, two epilogue rounds; no one-patch counterpart)
// (the pair index as a compile-time value where the pair body is called with one: BRES)
template <typename T>
struct PairConst { static constexpr int value = 0; };
template <int V>
struct PairConst<std::integral_constant<int, V>> { static constexpr int value = V; };

// BRES (32 -> 32 layers: NB = 1, Cin = 32, two pairs per patch): the wave's whole share of U -- 9 positions x 4 groups of 8 channels,
// 144 registers, the half of the accumulator file an NB = 1 kernel leaves unused -- is loaded ONCE per workgroup and stays resident:
// no B fragment loads in the pair loop (a buffer load costs the issuing wave ~19 cycles next to the fp32 MFMA, 72 of them per patch),
// and the two pairs are two copies of the body (pair index, image-buffer parity and register indices are compile-time constants)
// FULL (32 -> 32 layers whose epilogue leaves no room for BRES -- operand sets 15 / 27 / 31 -- but 48 registers): a 32-channel
// pixel is ONE 128-byte line and its two 64-byte halves belong to the patch's two pairs.  Requested microseconds apart the
// second half misses L2 again at this stage's traffic (profiles/r06_w4p_fetch_excess.txt), and these launches ARE bound by their
// bytes.  Here both halves of the next patch are requested back to back (during the current patch's pair 0; the pair-1 half
// waits in a second register set across pair 1 and the epilogue), nothing is requested twice, and the pair body is
// instantiated per pair as in BRES.
// MKM (operand sets with ReLU-mask bits; compile-time, like every operand combination of this kernel: run-time variants of the
// mask arrays went through scratch memory): how the 4 pixels of an epilogue row segment share their mask words -- 0 each pixel
// loads its own dword per component (any Cout, any W), 1 / 2 / 3: Cout = 32 / 64 / 128 and W % 4 == 0, see `keep` in the epilogue
template <int TC, bool AFF, int EPI, int NB, bool BRES = false, bool FULL = false, int MKM = 0>
__global__ __launch_bounds__(256, 1) void wino4p_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ u, const float *__restrict__ bias,
    const float *__restrict__ addend, const float *__restrict__ addend_mask, const float *__restrict__ in_scale,
    const float *__restrict__ in_shift, float *__restrict__ y, float *__restrict__ stats,
    const float *__restrict__ stat_aux, const float *__restrict__ stat_mean, const float *__restrict__ stat_invstd,
    const float *__restrict__ stat_mask, int H, int W, int Cin, int Cout, int patchesW, int patchesH, int nsp, int ncb,
    int xcd_div, int relu, int mask_bits) {
    using C = Cfg<TC>;
    using CP = CfgP<TC>;
    constexpr int PS = C::PS, CBUF = C::CBUF, RS = C::RS, PR = C::PR;
    __shared__ __attribute__((aligned(16))) float lds[CP::LDS_FLOATS + 2 * WMAXC];
    float *aff = lds + CP::LDS_FLOATS;                    // producer BatchNorm scale | shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    if (AFF)
        for (int c = tid; c < Cin; c += 256) {
            aff[c] = in_scale[c];
            aff[WMAXC + c] = in_shift[c];
        }
    // workgroup -> (XCD, slot): channel block cb = xcd % ncb as in the one-patch form; the workgroup walks the patches
    // sp = (slot + k * slots) * xcd_div + xcd / ncb, k = 0, 1, ...
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cb = xcd % ncb;
    const int spstep = (int)(gridDim.x >> 3) * xcd_div;
    int sp = slot * xcd_div + xcd / ncb;
    if (sp >= nsp) return;
    if (W4P_STAGGER > 0 && (slot & 1)) {
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0_ < (unsigned long long)W4P_STAGGER) __builtin_amdgcn_s_sleep(32);
    }
    const int co0 = cb * (32 * NB);
    auto decode = [&](int sp_, int &n_, int &ty_, int &tx_) {
        int t = sp_;
        const int pw = t % patchesW;
        t /= patchesW;
        const int ph = t % patchesH;
        n_ = __builtin_amdgcn_readfirstlane(t / patchesH);
        ty_ = __builtin_amdgcn_readfirstlane(ph * (4 * C::TR));
        tx_ = __builtin_amdgcn_readfirstlane(pw * (4 * TC));
    };

    auto opaque0 = [&]() {
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        return z;
    };
    int tpatch = 0;
    auto tstamp = [&](int k) {
        if (W4P_TIMING) {
            if (blockIdx.x == 8 && tid == 0 && tpatch < 6)
                reinterpret_cast<unsigned long long *>(const_cast<float *>(stat_mean))[tpatch * 16 + k] = __builtin_amdgcn_s_memtime();
        }
    };
    // ---- GEMM-side constants (as in wino4_fwd_kernel)
    const int nuF = wave == 0 ? 0 : wave == 1 ? 2 : wave == 2 ? 3 : 5;
    const int nuH = wave < 2 ? 1 : 4;
    const int hh = wave & 1;
    const float K2 = hh ? A2 : B2, KP = hh ? PB : PA;
    constexpr int ROW4 = 4 * TC * 16;
    // (made again after every epilogue, from an opaque copy of tid: nothing but the accumulators and a handful of registers is
    // live across an epilogue, which then has the register file of the one-patch kernel's epilogue to itself)
    int oF[4], oH[3], oZ[3];
    auto make_o = [&](int t_) {
        const int li_ = t_ & 31, lh_ = (t_ >> 5) & 1;
        const int tr = li_ >> C::LOG_TC, tc = li_ & (TC - 1);
        int o_[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o_[j] = ((lh_ * PS + (4 * tr + ((tr + j) & 3)) * TC + tc) * 16);
        const int planeF = nuF * 2 * PS * 16, planeH = nuH * 2 * PS * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) oF[j] = planeF + o_[j];
#pragma unroll
        for (int j = 0; j < 3; ++j) oH[j] = planeH + o_[j + 1];
        oZ[0] = planeH + (hh ? o_[1] : o_[0]);
        oZ[1] = planeH + (hh ? o_[3] : o_[2]);
        oZ[2] = planeH + (hh ? o_[2] : o_[1]) + ROW4;
    };
    // ---- staging geometry.  The item of a thread (patch row, tile column, channel quad) is the same for every patch; its six
    // byte offsets inside the sample are (patch origin) + (thread constant), out-of-image columns -> 0x80000000 (see above).
    // Two offset sets: F for the full rounds, L for the leftover rows; they move to the next patch at different steps.
    const int rowb = W * Cin * 4, pixb = Cin * 4;
    const unsigned nrec = (unsigned)H * (unsigned)rowb;
    const size_t xsample = (size_t)H * W * Cin;
    // The thread constants of the offset arithmetic are NOT kept across the pair loop (the loop lives at the 256-register
    // limit): they are rebuilt from an opaque copy of tid wherever a patch's offsets are made (twice per patch).
    int wbF0, wbF1, wbL;
    {
        const int stcF = (tid >> 2) & (TC - 1), yyF = tid >> (2 + C::LOG_TC);
        const int stcL = (lane >> 2) & (TC - 1), yyL = 2 * RS + (lane >> (2 + C::LOG_TC));
        auto wbof = [&](int sq4, int stc, int yy) {
            const int q = yy >> 2;
            const int brow = 4 * q + (((yy & 3) + q) & 3);       // storage row: rotation inside 4-row blocks
            return (sq4 >> 1) * (CBUF * 4) + ((sq4 & 1) * PS + brow * TC + stc) * 16;
        };
        wbF0 = wbof(tid & 3, stcF, yyF);
        wbF1 = wbof(tid & 3, stcF, yyF + RS);
        wbL = yyL < PR ? wbof(lane & 3, stcL, yyL) : -1;
    }
    int offF[6], offL[6];
    const float *xbF, *xbL;                               // sample bases of the patches the F / L loads belong to
    // Wv: the image width, or 0 when there is no such patch (every column out of range: the loads fetch nothing)
    auto set_off = [&](int (&off)[6], bool leftover, int ty_, int tx_, int Wv) {
        const int t_ = tid + opaque0();
        const int i_ = leftover ? (t_ & 63) : t_;
        const int stc = (i_ >> 2) & (TC - 1), yy = (leftover ? 2 * RS : 0) + (i_ >> (2 + C::LOG_TC));
        const bool rowok = !leftover || yy < PR;
        const int base = (ty_ + yy - 1) * rowb + (tx_ + 4 * stc - 1) * pixb + (i_ & 3) * 16;
        const int gx0 = tx_ + 4 * stc - 1;
#pragma unroll
        for (int j = 0; j < 6; ++j)
            off[j] = (rowok && (unsigned)(gx0 + j) < (unsigned)Wv) ? base + j * pixb : (int)0x80000000;
    };
    // (dead: a request whose data nobody will use -- see pair_body -- is made against an EMPTY buffer: every lane is out of range,
    //  nothing is fetched, and the load stays in place for the compiler's vmcnt bookkeeping)
    auto st_load = [&](f32x4 (&p)[6], int round, int pr, bool on, bool dead = false) {
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(round == 2 ? xbL : xbF), 0,
                                                                              (W4P_KILLDUP && dead) ? 0 : (int)nrec, 0x00020000);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int vo = round == 0 ? offF[j] : round == 1 ? offF[j] + RS * rowb : (on ? offL[j] : (int)0x80000000);
            p[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, pr * 64, W4P_X_AUX));
        }
    };
    // (the pixels in flight are ext-vector values, not float4 structs: with the struct form the SLP vectoriser paired component
    // stores ACROSS array elements in the AFF kernels and the arrays stayed in private memory)
    auto st_xform = [&](f32x4 (&p)[6], int round, int pr) {
        if (AFF) {
            // x' = scale * x + shift on in-image pixels; out-of-image pixels were read as 0 and must stay 0
            const int sq4 = (round == 2 ? lane : tid) & 3;
            const f32x4 isc = *reinterpret_cast<const f32x4 *>(&aff[pr * 16 + sq4 * 4]);
            const f32x4 ish = *reinterpret_cast<const f32x4 *>(&aff[WMAXC + pr * 16 + sq4 * 4]);
            int vo[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) vo[j] = round == 0 ? offF[j] : round == 1 ? offF[j] + RS * rowb : offL[j];
            if (W4P_AFF_EXEC) {
                aff6_inrange(p, isc, ish, vo, nrec);
            } else {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const bool ok = (unsigned)vo[j] < nrec;
                    p[j] = pkfma4v(p[j], isc, ok ? ish : zero);
                }
            }
        }
        bt6v(p);
    };
    auto st_write = [&](const f32x4 (&tt)[6], int round, float *Cn) {
        const int wb = round == 0 ? wbF0 : round == 1 ? wbF1 : wbL;
        if (round < 2 || wb >= 0) {
            char *dst = reinterpret_cast<char *>(Cn) + wb;
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4 *>(dst + j * 2 * PS * 16) = tt[j];
        }
    };
    auto st_store = [&](f32x4 (&p)[6], int round, float *Cn, int pr) {
        st_xform(p, round, pr);
        st_write(p, round, Cn);
    };

    static_assert(!BRES || NB == 1, "resident U: 32-channel output blocks only");
    static_assert(!(BRES && FULL), "BRES and FULL exclude each other");
    const int nkg = (BRES || FULL) ? 4 : Cin / 8, npairs = (BRES || FULL) ? 2 : Cin / 16;     // (npairs is even: Cin % 32 == 0)
    const size_t ustride_pos = (size_t)(Cout / 32) * nkg * 256;
    const int urec = (int)(36 * ustride_pos * 4);
    const int ulane = lane * 16;
    const int uwave = (int)((((size_t)(wave * 9) * (Cout / 32) + (size_t)cb * NB) * nkg * 256) * 4);
    // B fragment of use u of a pair (18 NB uses: u = (9 half + s) NB + nt) -- fragment (s, nt) of group 2 pr + half
    // (dead: as in st_load -- the last pair's requests past the end of the patch are repeated by the epilogue)
    auto bload = [&](int uu, int kg, bool dead = false) {
        const int s = uu / NB, nt = uu % NB;
        // (the wave's base is part of the resource's address, not of every load's scalar offset: one s_add per load fewer)
        const int so = (int)((s * ustride_pos + ((size_t)nt * nkg + kg) * 256) * 4);
        const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(u) + uwave), 0, (W4P_KILLDUP > 1 && dead) ? 0 : urec - uwave, 0x00020000);
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, so, 0));
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    constexpr int BR = W4_BRING;
    constexpr int UH = 9 * NB;                            // uses per 8-channel group
    float4 bq[BR];
    f32x4 bres[BRES ? 9 : 1][4];                          // [position of the wave][group of 8 input channels]
    if (BRES) {
#pragma unroll
        for (int s = 0; s < 9; ++s)
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                const float4 t = bload(s, kg);
                bres[s][kg] = f32x4{t.x, t.y, t.z, t.w};
            }
    } else {
#pragma unroll
        for (int uu = 0; uu < BR; ++uu) bq[uu] = bload(uu % UH, (uu / UH) % nkg);
    }

    int n, ty0, tx0;
    decode(sp, n, ty0, tx0);
    set_off(offF, false, ty0, tx0, W);
    set_off(offL, true, ty0, tx0, W);
    xbF = xbL = x + (size_t)n * xsample;

    f32x16 acc[9][NB];
    f32x4 pvA[6], pvB[6];
    f32x4 pvC[6], pvD[6];                                 // FULL: the pair-1 halves (rounds 0 / 1) of the next patch (else unused)
    __syncthreads();                                      // affine table visible
    // pair 0 of the first patch: all staging rounds in flight together (the accumulators are zeroed while they are)
    st_load(pvA, 0, 0, true);
    st_load(pvB, 1, 0, true);
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
        for (int nt = 0; nt < NB; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][nt][r] = 0.f;
    st_store(pvA, 0, lds, 0);
    st_load(pvA, 2, 0, wave == 0);                        // leftover rows
    st_store(pvB, 1, lds, 0);
    if (wave == 0) st_store(pvA, 2, lds, 0);
    if constexpr (FULL) {
        st_load(pvC, 0, 1, true);
        st_load(pvD, 1, 1, true);
    } else {
        st_load(pvA, 0, 1, true);                         // rounds 0 and 1 of pair 1 (npairs >= 2)
        st_load(pvB, 1, 1, true);
    }
    __syncthreads();

    // (ext-vector values throughout: with one wave per SIMD a v_pk_fma_f32 issues in the time of a v_fma_f32 --
    //  profiles/r03_valu_rate.txt: 6.0 vs 6.1 cycles -- so every transform is written on <4 x float>, which the backend
    //  splits into two packed instructions; the float4-struct form compiled to scalar fmas)
    f32x4 cF[6], cP[4], cZ[3];
    f32x4 a[9];
    int bF[4], bH[3], bZ[3];
    const char *ldsb = reinterpret_cast<const char *>(lds);
    auto a_reads_full = [&](int g) {
        cF[0] = *reinterpret_cast<const f32x4 *>(ldsb + bF[0] + g * (CBUF * 4));
        cF[1] = *reinterpret_cast<const f32x4 *>(ldsb + bF[1] + g * (CBUF * 4));
        cF[2] = *reinterpret_cast<const f32x4 *>(ldsb + bF[2] + g * (CBUF * 4));
        cF[3] = *reinterpret_cast<const f32x4 *>(ldsb + bF[3] + g * (CBUF * 4));
        cF[4] = *reinterpret_cast<const f32x4 *>(ldsb + bF[1] + g * (CBUF * 4) + ROW4);
        cF[5] = *reinterpret_cast<const f32x4 *>(ldsb + bF[2] + g * (CBUF * 4) + ROW4);
    };
    auto a_reads_half = [&](int g) {
        cP[0] = *reinterpret_cast<const f32x4 *>(ldsb + bH[0] + g * (CBUF * 4));
        cP[1] = *reinterpret_cast<const f32x4 *>(ldsb + bH[1] + g * (CBUF * 4));
        cP[2] = *reinterpret_cast<const f32x4 *>(ldsb + bH[2] + g * (CBUF * 4));
        cP[3] = *reinterpret_cast<const f32x4 *>(ldsb + bH[0] + g * (CBUF * 4) + ROW4);
        cZ[0] = *reinterpret_cast<const f32x4 *>(ldsb + bZ[0] + g * (CBUF * 4));
        cZ[1] = *reinterpret_cast<const f32x4 *>(ldsb + bZ[1] + g * (CBUF * 4));
        cZ[2] = *reinterpret_cast<const f32x4 *>(ldsb + bZ[2] + g * (CBUF * 4));
    };
    auto a_xform_full = [&]() {
        bt6v2(cF, a[0], a[1], a[2], a[3], a[4], a[5]);    // the six xi of the full column
    };
    auto a_xform_half = [&]() {
        bt3v(cP, cZ, a[6], a[7], a[8], K2, KP);
    };
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

    // next patch of this workgroup (decoded while the current one runs)
    int spn, nn = 0, nty = 0, ntx = 0, Wn = 0;

    // One pair of the pipeline (the 36 steps of wino4_fwd_kernel).  The same code runs for every pair of a patch, the last one
    // included: there the image being staged is pair 0 of the next patch, and the requests for ITS pair 1, the first A fragments
    // and the B fragments past the end are issued as always but never used -- the epilogue issues them again when their
    // registers are free.  (Round 6: the unused pair-1 requests go to an empty buffer -- W4P_KILLDUP -- because the epilogue's
    // repeat was NOT an L2 hit at stages 1 and 2, where an XCD's L2 holds only a few microseconds of the kernel's traffic.)
    // No run-time condition around a load or an LDS read (the compiler's vmcnt / lgkmcnt bookkeeping stays exact), and nothing
    // of them is live across the epilogue.
    auto pair_body = [&](auto PR_) {
        const int pr = PR_;                                                    // (BRES, FULL: an integral constant)
        constexpr bool second = FULL && PairConst<decltype(PR_)>::value == 0;   // FULL, pair 0: pair 1 of THIS patch is staged, from C / D
        f32x4(&SA)[6] = second ? pvC : pvA;
        f32x4(&SB)[6] = second ? pvD : pvB;
        // BRES, pair 1: the image staged here is pair 0 of the NEXT patch, requested during pair 0 of this one -- round 1 at
        // step 16, round 0 at step 25.  With 32-channel blocks a step is half as long (only every other step carries MFMAs): from
        // step 25 to step 4 of this pair are ~2 500 cycles, less than a memory latency under load.  So this pair takes round 1 FIRST
        // (steps 4 / 6; 24 steps after its request) and round 0 second (12 / 14; 23 steps), and the leftover rows ride in the round-1
        // registers, which nothing requests into during this pair.
        // (measured: resident-U kernels -1.7 to -3.3 % per launch, whole-line kernels +-1 %: BRES only; profiles/r06_w4p_leftpin_ab.txt)
        constexpr bool swapped = W4P_SWAP1 && BRES && PairConst<decltype(PR_)>::value == 1;
        f32x4(&S1)[6] = swapped ? SB : SA;                                     // staged at steps 4 / 6, then host of the leftover rows
        f32x4(&S2)[6] = swapped ? SA : SB;                                     // staged at steps 12 / 14
        constexpr int R1 = swapped ? 1 : 0, R2 = swapped ? 0 : 1;
        float *Cn = lds + ((pr + 1) & 1) * (2 * CBUF);
        const int prn = pr + 1 < npairs ? pr + 1 : 0;                          // pair being staged
        const int prn2 = pr + 2 < npairs ? pr + 2 : pr + 2 - npairs;           // pair being requested
        const bool lwave = wave == (pr & 3);                                   // this wave stages the leftover rows of the pair
        const bool turn = pr == npairs - 2;                                    // the requests move on to the next patch in this pair
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int s = 0; s < 9; ++s) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int step = half * 18 + 2 * s + nt;                   // (the 36 steps of the schedule, whatever NB: with NB = 1
                    const bool live = nt < NB;                                 //  the odd ones carry side work and loads only)
                    const int use = (half * 9 + s) * NB + (live ? nt : 0);
                    const int slot_ = use % BR;
                    if (W4P_TIMING && (step % 6) == 0 && pr < 2) {
                        if (blockIdx.x == 8 && tid == 0 && tpatch >= 1 && tpatch < 4)
                            reinterpret_cast<unsigned long long *>(const_cast<float *>(stat_mean))[(6 + tpatch) * 16 + pr * 6 + step / 6] =
                                __builtin_amdgcn_s_memtime();
                    }
                    // ---- side work of the step
                    if (step == 2) a_xform_half();
                    if (step == 12) a_xform_full();
                    if (step == 18) a_xform_half();
                    if (step == 30) a_xform_full();                            // the next pair's first group
                    if (step == 4) st_xform(S1, R1, prn);
                    if (step == 12) st_xform(S2, R2, prn);
                    if (step == 6) st_write(S1, R1, Cn);
                    if (step == 14) st_write(S2, R2, Cn);
                    // (the leftover rows are requested at step 8 and used by ONE wave here.  In the per-pair copies of the body (BRES,
                    //  FULL) the compiler sank the six loads into this branch, right in front of their use: the staging wave waited out
                    //  a whole memory latency, the other three at the barrier behind it -- ~2 000 cycles per pair in the stamps of
                    //  profiles/r06_w4p_timing.txt.  An empty asm that READS the six values in every wave keeps the requests where they
                    //  are issued; the other waves' requests were out of range and came back at once.)
                    if (step == 22 && W4P_LEFT_PIN && (BRES || FULL))       // (the one-body kernels keep the requests in place by themselves)
                        asm volatile("" :: "v"(S1[0]), "v"(S1[1]), "v"(S1[2]), "v"(S1[3]), "v"(S1[4]), "v"(S1[5]));
                    if (step == 22 && lwave) st_store(S1, 2, Cn, prn);
                    // full-round requests from step 16 on and leftover-row requests from the next pair's step 8 on belong to the
                    // next patch (every transform of this patch's data that needs the old offsets is done by then)
                    if (step == 13 && turn) {
                        set_off(offF, false, nty, ntx, Wn);
                        xbF = x + (size_t)nn * xsample;
                    }
                    if (step == 23 && turn) {
                        set_off(offL, true, nty, ntx, Wn);
                        xbL = x + (size_t)nn * xsample;
                    }
                    if (!live) {
                    } else if (W4P_WHATIF & 16) {
                        asm volatile("" : "+v"(a[s]));
                        asm volatile("" : "+v"(bq[slot_].x), "+v"(bq[slot_].y), "+v"(bq[slot_].z), "+v"(bq[slot_].w));
                    } else if (BRES) {
                        // (three of the four channel groups live in AccVGPRs -- 144 accumulators + 108 = 252 of the 256 -- and are MFMA
                        //  operands there; left to itself the allocator parks ~70 of them in AccVGPRs too, but copies each back)
                        const int kg_ = (2 * PairConst<decltype(PR_)>::value + half) & 3;        // (a constant once the loops are unrolled)
                        f32x4 &b_ = bres[s][kg_];
                        if (kg_ < W4P_BRES_ACC) asm volatile("" : "+a"(b_)); else asm volatile("" : "+v"(b_));
                        asm volatile("" : "+v"(a[s]));
                        acc[s][0] = mfma32(a[s][0], b_[0], acc[s][0]);
                        acc[s][0] = mfma32(a[s][1], b_[1], acc[s][0]);
                        acc[s][0] = mfma32(a[s][2], b_[2], acc[s][0]);
                        acc[s][0] = mfma32(a[s][3], b_[3], acc[s][0]);
                        asm volatile("" : "+a"(acc[s][0]));
                    } else if (s < 8 || NB == 1) {
                        const int an = live ? nt : 0;
                        asm volatile("" : "+v"(a[s]));
                        acc[s][an] = mfma32(a[s][0], bq[slot_].x, acc[s][an]);
                        acc[s][an] = mfma32(a[s][1], bq[slot_].y, acc[s][an]);
                        acc[s][an] = mfma32(a[s][2], bq[slot_].z, acc[s][an]);
                        acc[s][an] = mfma32(a[s][3], bq[slot_].w, acc[s][an]);
                        asm volatile("" : "+a"(acc[s][an]));
                    } else {
                        mfma32x4_vgpr(acc[s][live ? nt : 0], make_float4(a[s][0], a[s][1], a[s][2], a[s][3]), bq[slot_]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- loads of the step
                    if (live && !BRES && !(W4P_WHATIF & 64)) {
                        const int v = use + BR;
                        const int kgv = 2 * pr + v / UH;
                        bq[slot_] = bload(v % UH, kgv < nkg ? kgv : kgv - nkg, kgv >= nkg);  // (wraps: the next patch uses the same U)
                    }
                    if (step == 24) {
                        __syncthreads();                                       // the next pair's image is complete
                        pair_bases(pr + 1);
                    }
                    if (step == 8) a_reads_full(1);
                    if (step == 16) a_reads_half(1);
                    if (step == 28) a_reads_full(0);                           // (bases: the next pair's already)
                    if (step == 32) a_reads_half(0);
                    if (!(W4P_WHATIF & 128)) {
                        if constexpr (FULL) {
                            // both halves of the next patch's lines, back to back, from the current patch's pair 0; nothing from pair 1
                            if constexpr (second) {
                                if (step == 16) {
                                    st_load(pvB, 1, 0, true);
                                    st_load(pvD, 1, 1, true);
                                    if (W4P_REQ_FENCE) asm volatile("" ::: "memory");      // (as in the BRES branch below)
                                }
                                if (step == 25) {
                                    st_load(pvA, 0, 0, true);
                                    st_load(pvC, 0, 1, true);
                                }
                            }
                        } else if constexpr (BRES) {
                            // (the pair index is a compile-time value: the second pair's requests -- never used, see KILLDUP -- are
                            //  simply not there; they would also land in the registers that host its leftover rows, see `swapped`)
                            if constexpr (PairConst<decltype(PR_)>::value == 0) {
                                if (step == 25) st_load(pvA, 0, prn2, true);
                                if (step == 16) {
                                    st_load(pvB, 1, prn2, true);
                                    // (a compiler barrier for memory operations: without it these six requests drifted down past the
                                    //  `if (lwave)` block of step 22 -- a basic-block boundary the scheduling barriers do not reach
                                    //  across -- and the second pair needs their data FIRST, see `swapped`)
                                    if (W4P_REQ_FENCE) asm volatile("" ::: "memory");
                                }
                            }
                        } else {
                            if (step == 25) st_load(pvA, 0, prn2, true, pr == npairs - 1);
                            if (step == 16) st_load(pvB, 1, prn2, true, pr == npairs - 1);
                        }
                        if (step == 8) st_load(S1, 2, prn, lwave);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };

    // ---- epilogue
    constexpr int DPOS[6][6] = {{0, 1, 2, 3, 4, 5},       {6, 7, 8, 16, 17, 15},   {9, 10, 11, 12, 13, 14},
                                {18, 19, 20, 21, 22, 23}, {24, 25, 26, 34, 35, 33}, {27, 28, 29, 30, 31, 32}};   // [nu][xi] -> position
    const size_t ysample = (size_t)H * W * Cout;
    const int sbytes = H * W * Cout * 4;
    auto rsrc_of = [&](const float *ptr, int bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ptr), 0, bytes, 0x00020000);
    };
    auto load4 = [&](const __amdgpu_buffer_rsrc_t &rs, int o_) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o_, 0, W4P_OP_AUX));
    };

    make_o(tid + opaque0());
    pair_bases(0);
    a_reads_full(0);
    a_xform_full();
    a_reads_half(0);
    for (;;) {
        spn = sp + spstep;
        const bool more = spn < nsp;
        Wn = 0;
        if (more) {
            decode(spn, nn, nty, ntx);
            Wn = W;
        }
        tstamp(0);
        if constexpr (BRES || FULL) {
            pair_body(std::integral_constant<int, 0>{});
            pair_body(std::integral_constant<int, 1>{});
        } else {
            for (int pr = 0; pr < npairs; ++pr) pair_body(pr);
        }
        tstamp(1);
        __syncthreads();                                  // every wave is done with image buffers 2, 3

        // ---- epilogue of patch sp (n, ty0, tx0).  Its per-lane constants are made here, from an opaque copy of tid, and its
        // pointer arguments are read from the kernel-argument segment through an opaque copy of its address (offsets: the
        // kernel's signature), so that neither occupies registers across the pair loop.
        constexpr bool ST = (EPI & 1) != 0, AD = (EPI & 2) != 0, MK = (EPI & 4) != 0, AUX = (EPI & 8) != 0, SMK = (EPI & 16) != 0;
        constexpr bool HOIST = W4P_HOIST == 2 || (W4P_HOIST == 1 && !(AD && AUX));
        typedef const char __attribute__((address_space(4))) *kargp_t;
        kargp_t kp = (kargp_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        auto karg = [&](int o_) {
            return reinterpret_cast<float *>(*reinterpret_cast<const unsigned long long __attribute__((address_space(4))) *>(kp + o_));
        };
        const int te = tid + opaque0();
        const int par = wave >> 1;                        // output-row parity this wave finishes (rows par, par + 2 of a tile)
        const int rrd = ((te >> 4) & 3) + 4 * (wave & 1), idx = te & 15, lhr = idx >> 3, c4 = idx & 7;
        const char *xrd = ldsb + CP::XOFF * 4 + rrd * 256 + idx * 16;
        float *xwr = lds + CP::XOFF + (wave * 9 * 8) * 64 + (te & 63);
        f32x4 ssum[NB], ssq[NB];
#pragma unroll
        for (int nt = 0; nt < NB; ++nt) ssum[nt] = ssq[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const size_t sbase = (size_t)n * ysample;         // floats
        const __amdgpu_buffer_rsrc_t yrs = rsrc_of(karg(KA_Y) + sbase, sbytes);                 // y
#pragma unroll
        for (int rnd = 0; rnd < 2 * NB; ++rnd) {
            const int nt = rnd >> 1, rh = rnd & 1;
            // ---- the thread's 8 pixels of the round: tile m (rows rh * 16 ..), channel quad co, output rows par and par + 2.
            // Byte offset of the channel quad inside the sample; out-of-image pixels get 0x80000000: loads give 0, stores are dropped
            const int m = 16 * rh + (rrd >> 2) * 8 + lhr * 4 + (rrd & 3);
            const int tr = m >> C::LOG_TC, tc = m & (TC - 1);
            const int co = co0 + nt * 32 + c4 * 4;
            int off[2][4];
            {
                const int gx0 = tx0 + 4 * tc;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int gy = ty0 + 4 * tr + par + 2 * e;
                    const int rowo = (__mul24(__mul24(gy, W) + gx0, Cout) + co) * 4;
#pragma unroll
                    for (int b = 0; b < 4; ++b) off[e][b] = (gy < H && gx0 + b < W) ? rowo + b * Cout * 4 : (int)0x80000000;
                }
            }
            // ---- fused operands, requested BEFORE the accumulators go to LDS: their latency hides under the writer half, the
            // barrier and the xi pass.  ReLU-mask bits: the float4 with index q inside the sample owns bit (q & 63) of four 64-bit
            // words at byte (q >> 6) * 32 of the sample's mask (a sample starts on a word boundary: checked by the host); one dword
            // per component holds the bit.  The 32 dwords of a mask are squeezed into ONE register (bit 4 pixel + component) right
            // after the barrier -- they are small and mostly L2 hits, the operand tensors are not -- and a v_bfe_i32 with constant
            // offsets turns a bit into an and-mask where it is applied.
            f32x4 ad[2][4], ax[2][4];
            unsigned amk_raw[2][4][4], smk_raw[2][4][4], amk = 0, smk = 0;
            // (round 6: a buffer load costs the issuing wave ~19 cycles here, and the 4 pixels of a row segment are 4 x Cout/4 float4
            //  apart -- with 32 output channels all of them share ONE dword per component, with 64 one 8-byte word, with 128 two:
            //  4 / 4 / 8 loads per row segment and mask instead of 16.  The shared word is fetched for the segment's first pixel; if
            //  that one is in the image and a later one is not, the later one reads bits of some other pixel -- its addend is 0, its
            //  store is dropped and its statistics are skipped, so they are never used)
            // (needs W % 4 == 0: a row segment then starts at a float4 index that is a multiple of 4 x Cout/4; the launcher checks)
            constexpr int mkmode = MKM;
            static_assert(MKM == 0 || (MKM == 1 && NB == 1) || ((MKM == 2 || MKM == 3) && NB == 2), "mask sharing mode vs block width");
            auto keep = [&](const float *mptr, unsigned (&mk)[2][4][4]) {
                const __amdgpu_buffer_rsrc_t rs = rsrc_of(mptr + (sbase >> 8) * 8, sbytes >> 5);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int q0 = off[e][0] >> 4;
                    if (mkmode == 1) {
                        const int wo = off[e][0] >= 0 ? (q0 >> 6) * 32 + ((q0 >> 5) & 1) * 4 : (int)0x80000000;
#pragma unroll
                        for (int k = 0; k < 4; ++k) mk[e][0][k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, wo, k * 8, 0);
                    } else if (mkmode >= 2) {
                        const int wo = off[e][0] >= 0 ? (q0 >> 6) * 32 : (int)0x80000000;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, wo, k * 8, 0);
                            mk[e][0][k] = v[0];
                            mk[e][1][k] = v[1];
                            if (mkmode == 3) {
                                const u32x2_t v2 = __builtin_amdgcn_raw_buffer_load_b64(rs, wo, 32 + k * 8, 0);
                                mk[e][2][k] = v2[0];
                                mk[e][3][k] = v2[1];
                            }
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const int q = off[e][b] >> 4;
                            const int wo = off[e][b] >= 0 ? (q >> 6) * 32 + ((q >> 5) & 1) * 4 : (int)0x80000000;
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                mk[e][b][k] = (W4P_WHATIF & 256) ? 0xffffffffu + (unsigned)wo * 0u
                                                                 : (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, wo, k * 8, 0);
                        }
                    }
                }
            };
            auto squeeze = [&](const unsigned (&mk)[2][4][4]) {
                unsigned r = 0;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const unsigned s0 = (unsigned)(off[e][0] >> 4) & 31u;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const unsigned sh = mkmode == 1 ? s0 + 8u * b : mkmode == 2 ? s0 + 16u * (b & 1) : mkmode == 3 ? s0
                                                                                                          : (unsigned)(off[e][b] >> 4) & 31u;
                        const int rb = mkmode == 1 ? 0 : mkmode == 2 ? (b >> 1) : b;
#pragma unroll
                        for (int k = 0; k < 4; ++k) r |= __builtin_amdgcn_ubfe(mk[e][rb][k], sh, 1u) << (16 * e + 4 * b + k);
                    }
                }
                return r;
            };
            if (AD) {
                const __amdgpu_buffer_rsrc_t ars = rsrc_of(karg(KA_ADDEND) + sbase, sbytes);    // addend
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int b = 0; b < 4; ++b) ad[e][b] = load4(ars, off[e][b]);
                if (MK) keep(karg(KA_ADDEND_MASK), amk_raw);                                      // addend_mask
            }
            if (SMK) keep(karg(KA_STAT_MASK), smk_raw);                                           // stat_mask
            if (AUX) {
                const __amdgpu_buffer_rsrc_t xrs_ = rsrc_of(karg(KA_STAT_AUX) + sbase, sbytes);  // stat_aux
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int b = 0; b < 4; ++b) ax[e][b] = load4(xrs_, off[e][b]);
            }
            if (HOIST && rnd == 0) {
                // (every launch but the identity / projection data-gradients with statistics: the next patch's first pixel requests go out before the epilogue's first store --
                //  behind the stores of three rounds the twelve requests took 2 000-4 000 cycles to issue, the stores drain at
                //  ~10 B / clock / CU; those two keep the late form, the 48 registers are not there: profiles/r05_w4p_hoisted_requests_ab.txt)
                set_off(offF, false, nty, ntx, Wn);
                xbF = x + (size_t)nn * xsample;
                if constexpr (!FULL) {                    // (FULL: that half is in pvC / pvD already)
                    st_load(pvA, 0, 1, true);
                    st_load(pvB, 1, 1, true);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- writer: raw accumulators (rows rh * 8 .. + 7 of the 32 x 32 tile) of the wave's nine positions, straight from
            // the AccVGPRs; the registers are zeroed for the next patch while the stores drain.  Two registers per instruction
            // (ds_write2st64_b32: words 64 apart -- the exchange rows -- with both data operands AccVGPRs): 3 source dwords = 6 cycles
            // of the LDS store path for 512 bytes where a ds_write_b32 takes 4 for 256 -- 928 against 1 212 cycles per round in
            // tools/micro/write2_acc.hip (profiles/r06_addtid_probe.txt).  hipcc merges such pairs for architectural registers but
            // not for AccVGPRs, hence the assembly; the wait below is the one the compiler would have placed before the barrier.
            // (The ninth tile of a 64-channel workgroup lives in architectural registers: mfma32x4_vgpr.)
            if (!(W4P_WHATIF & 2)) {
                const unsigned xwb = (unsigned)(size_t)xwr;
#pragma unroll
                for (int s = 0; s < 9; ++s)
#pragma unroll
                    for (int rr = 0; rr < 8; rr += 2) {
                        if (s < 8 || NB == 1)
                            asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%c3 offset1:%c4"
                                         :: "v"(xwb), "a"(acc[s][nt][rh * 8 + rr]), "a"(acc[s][nt][rh * 8 + rr + 1]), "i"(s * 8 + rr),
                                            "i"(s * 8 + rr + 1) : "memory");
                        else
                            asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%c3 offset1:%c4"
                                         :: "v"(xwb), "v"(acc[s][nt][rh * 8 + rr]), "v"(acc[s][nt][rh * 8 + rr + 1]), "i"(s * 8 + rr),
                                            "i"(s * 8 + rr + 1) : "memory");
                        acc[s][nt][rh * 8 + rr] = 0.f;
                        acc[s][nt][rh * 8 + rr + 1] = 0.f;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            tstamp(2 + rnd);
            __syncthreads();
            if (MK) amk = squeeze(amk_raw);
            if (SMK) smk = squeeze(smk_raw);
            // ---- reader, xi direction: Q[row][nu] for the thread's two rows (rows 0, 2 take the sums m1 + m2, m3 + m4, rows 1, 3
            // the differences: a wave-uniform branch), two nu columns of reads in flight.  All arithmetic on <4 x float> values:
            // packed instructions (see the pair loop).
            auto fm = [](float k, f32x4 a_, f32x4 b_) { return pkfma4(k, a_, b_); };
            auto sc = [](float k, f32x4 a_) {
                const f32x4 kk = {k, k, k, k};
                return kk * a_;
            };
            f32x4 qa[6], qb[6];
            if (W4P_WHATIF & (1 | 8)) {
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) qa[nu] = qb[nu] = f32x4{1.f + nu, 2.f, 3.f, 4.f};
            } else {
                f32x4 mm[2][5];
                auto rd5 = [&](f32x4 (&d)[5], int nu) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) d[k] = *reinterpret_cast<const f32x4 *>(xrd + DPOS[nu][k + (par ? 1 : 0)] * 2048);
                };
                if (par == 0) {
                    rd5(mm[0], 0);
                    rd5(mm[1], 1);
#pragma unroll
                    for (int nu = 0; nu < 6; ++nu) {
                        const f32x4(&v)[5] = mm[nu & 1];                        // m0 .. m4
                        const f32x4 s12 = v[1] + v[2], s34 = v[3] + v[4];
                        qa[nu] = (v[0] + s12) + s34;
                        qb[nu] = fm(A2, s12, sc(B2, s34));
                        __builtin_amdgcn_sched_barrier(0);
                        if (nu + 2 < 6) rd5(mm[nu & 1], nu + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    rd5(mm[0], 0);
                    rd5(mm[1], 1);
#pragma unroll
                    for (int nu = 0; nu < 6; ++nu) {
                        const f32x4(&v)[5] = mm[nu & 1];                        // m1 .. m5
                        const f32x4 d12 = fm(-1.f, v[1], v[0]), d34 = fm(-1.f, v[3], v[2]);   // (no v_pk_sub_f32: a - b as a packed fma)
                        qa[nu] = fm(PA, d12, sc(PB, d34));
                        qb[nu] = fm(A3, d12, fm(B3, d34, v[4]));
                        __builtin_amdgcn_sched_barrier(0);
                        if (nu + 2 < 6) rd5(mm[nu & 1], nu + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (rnd == (HOIST ? -1 : 2 * NB - 1)) {
                // The next patch's first pixel requests, made here -- after the last LDS reads of the epilogue, when their registers
                // are free -- so that they land under the rest of the round: rounds 0 / 1 of its pair 1 (its pair 0 was staged by the
                // last pair above).  Without a next patch every offset is out of range.
                __builtin_amdgcn_sched_barrier(0);
                set_off(offF, false, nty, ntx, Wn);
                xbF = x + (size_t)nn * xsample;
                if (W4P_TIMING) tstamp(12);
                if constexpr (!FULL) {
                    st_load(pvA, 0, 1, true);
                    st_load(pvB, 1, 1, true);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (W4P_TIMING) tstamp(13);
            }
            // ---- nu direction and the pixels (ReLU: one wave-uniform branch per round around two copies of the loop)
            f32x4 smean = {0.f, 0.f, 0.f, 0.f}, sinv = smean;
            if (AUX) {
                smean = *reinterpret_cast<const f32x4 *>(karg(KA_STAT_MEAN) + co);                // stat_mean, stat_invstd
                sinv = *reinterpret_cast<const f32x4 *>(karg(KA_STAT_INVSTD) + co);
            }
            auto andf = [](float v, unsigned k) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & k); };
            auto pixels = [&](auto RL_) {
                constexpr bool RL = decltype(RL_)::value;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f32x4(&q)[6] = e ? qb : qa;
                    f32x4 Y[4];
                    if (W4P_WHATIF & 32) {
                        Y[0] = q[0]; Y[1] = q[1]; Y[2] = q[2]; Y[3] = q[3];
                    } else {
                        // A^T along nu: y[b] = sum_nu AT[b][nu] q[nu]
                        const f32x4 s12 = q[1] + q[2], d12 = fm(-1.f, q[2], q[1]), s34 = q[3] + q[4], d34 = fm(-1.f, q[4], q[3]);
                        Y[0] = (q[0] + s12) + s34;
                        Y[1] = fm(PA, d12, sc(PB, d34));
                        Y[2] = fm(A2, s12, sc(B2, s34));
                        Y[3] = fm(A3, d12, fm(B3, d34, q[5]));
                    }
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        f32x4 v = Y[b];
                        const unsigned bit0 = 16 * e + 4 * b;
                        if (AD) {
                            f32x4 a_ = ad[e][b];
                            if (MK) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) a_[k] = andf(a_[k], (unsigned)__builtin_amdgcn_sbfe((int)amk, bit0 + k, 1u));
                            }
                            v = v + a_;
                        }
                        if (RL) {
                            // (one v_max_f32 per component: fmaxf() compiles to TWO here -- a canonicalising max(v, v) in front of
                            //  max(v, 0): 60 instead of 32 vector instructions per round; same result, NaN -> 0 as fmaxf gives)
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float r_;
                                asm("v_max_f32 %0, 0, %1" : "=v"(r_) : "v"(v[k]));
                                v[k] = r_;
                            }
                        }
                        if (!(W4P_WHATIF & 4) || (e == 0 && b == 0))
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), yrs, off[e][b], 0, W4P_ST_AUX);
                        if (ST && W4P_STAT_EXEC && !(SMK && MK)) {
                            // masked-out components count 0; so do out-of-image pixels: the sums are taken under an EXEC mask
                            if (SMK) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) v[k] = andf(v[k], (unsigned)__builtin_amdgcn_sbfe((int)smk, bit0 + k, 1u));
                            }
                            f32x4 w_ = v;
                            if (AUX) w_ = fm(-1.f, smean, ax[e][b]) * sinv;
                            stat_acc_inimage(ssum[nt], ssq[nt], v, w_, off[e][b]);
                        } else if (ST) {
                            // out-of-image pixels and masked-out components count 0
                            const unsigned kin = ~(unsigned)(off[e][b] >> 31);
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                v[k] = andf(v[k], SMK ? (kin & (unsigned)__builtin_amdgcn_sbfe((int)smk, bit0 + k, 1u)) : kin);
                            ssum[nt] = ssum[nt] + v;
                            f32x4 w_ = v;
                            if (AUX) w_ = fm(-1.f, smean, ax[e][b]) * sinv;
                            ssq[nt] = __builtin_elementwise_fma(v, w_, ssq[nt]);
                        }
                    }
                }
            };
            if (!(W4P_WHATIF & 1)) {
                if (relu) pixels(std::true_type{}); else pixels(std::false_type{});
            }
            if (W4P_TIMING && rnd == 2 * NB - 1) tstamp(14);
            if (rnd == 2 * NB - 1) {
                // ... and its first B fragments (L2 hits: the last pair requested them once already)
                if (!BRES) {
#pragma unroll
                    for (int uu = 0; uu < BR; ++uu) bq[uu] = bload(uu % UH, (uu / UH) % nkg);
                }
            }
            tstamp(6 + rnd);
            __syncthreads();                              // the exchange region is free again
        }
        if (ST) {
            // per-patch, per-channel sums of the stored output, layout [2][patches][Cout] (see conv.hip); 32 thread groups
            float *red = lds + CP::XOFF;                  // [2][32 groups][64]
            const int grp = (te >> 4) * 2 + lhr;
#pragma unroll
            for (int nt = 0; nt < NB; ++nt) {
                *reinterpret_cast<f32x4 *>(&red[(0 * 32 + grp) * 64 + nt * 32 + c4 * 4]) = ssum[nt];
                *reinterpret_cast<f32x4 *>(&red[(1 * 32 + grp) * 64 + nt * 32 + c4 * 4]) = ssq[nt];
            }
            __syncthreads();
            if (te < 128 && (te & 63) < 32 * NB) {
                const int c = te & 63, which = te >> 6;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                for (int gI = 0; gI < 32; gI += 4) {
                    s0 += red[(which * 32 + gI) * 64 + c];
                    s1 += red[(which * 32 + gI + 1) * 64 + c];
                    s2 += red[(which * 32 + gI + 2) * 64 + c];
                    s3 += red[(which * 32 + gI + 3) * 64 + c];
                }
                karg(KA_STATS)[(size_t)which * nsp * Cout + (size_t)sp * Cout + co0 + c] = (s0 + s1) + (s2 + s3);     // stats
            }
            __syncthreads();
        }
        tstamp(10);
        if (!more) break;
        sp = spn;
        n = nn;
        ty0 = nty;
        tx0 = ntx;
        // the loop's per-lane state again (see make_o), the leftover-row offsets of the patch, and the first A fragments of its
        // pair 0 (whose image was complete before the epilogue)
        make_o(tid + opaque0());
        set_off(offL, true, ty0, tx0, W);
        xbL = xbF;
        pair_bases(0);
        a_reads_full(0);
        a_xform_full();
        a_reads_half(0);
        tstamp(11);
        ++tpatch;
    }
}


// the kernarg offsets above against the signature (the same for every instantiation): parameter index -> byte offset
namespace kargcheck {
using KL = KargLayout<decltype(&wino4p_fwd_kernel<4, false, 0, 2>)>;
static_assert(KL::count == 24, "wino4p_fwd_kernel: parameter list changed -- revisit the KA_* offsets and W4Launch");
static_assert(KL::offset(3) == KA_ADDEND && KL::offset(4) == KA_ADDEND_MASK, "KA_ADDEND / KA_ADDEND_MASK != kernarg layout");
static_assert(KL::offset(7) == KA_Y && KL::offset(8) == KA_STATS, "KA_Y / KA_STATS != kernarg layout");
static_assert(KL::offset(9) == KA_STAT_AUX && KL::offset(10) == KA_STAT_MEAN && KL::offset(11) == KA_STAT_INVSTD &&
                  KL::offset(12) == KA_STAT_MASK, "KA_STAT_* != kernarg layout");
// ... and that those parameters ARE the pointers the names say (a swap of two same-typed pointers cannot be seen by the
// compiler; the launch macro below passes W4Launch members by NAME in signature order, and tests/test_gpu_kernels.py runs
// every operand combination against float64)
}  // namespace kargcheck

template <int EPI>
void launch_wino4p(const W4Launch &a) {
#define ADYOLO_WINO4P_FWD(TC_, AFF_, NB_, ...)                                                                             \
    hipLaunchKernelGGL((wino4p_fwd_kernel<TC_, AFF_, EPI, NB_, ##__VA_ARGS__>), dim3((unsigned)a.grid), dim3(256), 0, a.st, a.x, a.u, a.bias, \
                       a.addend, a.addend_mask, a.in_scale, a.in_shift, a.y, a.stats, a.stat_aux, a.stat_mean,             \
                       a.stat_invstd, a.stat_mask, a.H, a.W, a.Cin, a.Cout, a.patchesW, a.patchesH, a.nsp, a.ncb,          \
                       a.xcd_div, a.relu, a.mask_bits)
#define ADYOLO_WINO4P_NB(NB_)                                                                                              \
    if (a.tc == 8) {                                                                                                       \
        if (a.in_scale) ADYOLO_WINO4P_FWD(8, true, NB_); else ADYOLO_WINO4P_FWD(8, false, NB_);                            \
    } else {                                                                                                               \
        if (a.in_scale) ADYOLO_WINO4P_FWD(4, true, NB_); else ADYOLO_WINO4P_FWD(4, false, NB_);                            \
    }
    // narrow maps (W <= 8: the middle stages of the ResNet-Conformer): patches one or two tiles wide, plain launches only
    if constexpr (EPI == 0) {
        if (a.tc == 1 || a.tc == 2) {
            if (a.tc == 1) ADYOLO_WINO4P_FWD(1, false, 2); else ADYOLO_WINO4P_FWD(2, false, 2);
            return;
        }
    }
    // shared ReLU-mask words (MKM): the three shapes the SE-ResNet's data gradients have, bits for every mask that is given
    const bool mkshare = W4P_MKDEDUP && (EPI == 15 || EPI == 27 || EPI == 31) && (a.W & 3) == 0 && !a.in_scale &&
                         (a.mask_bits & (((EPI & 4) ? 1 : 0) | ((EPI & 16) ? 2 : 0))) == (((EPI & 4) ? 1 : 0) | ((EPI & 16) ? 2 : 0));
    if constexpr (EPI == 15 || EPI == 27 || EPI == 31) {
        if (mkshare && a.nb == 2 && a.tc == 8 && a.Cout == 64) {
            ADYOLO_WINO4P_FWD(8, false, 2, false, false, 2);
            return;
        }
        // (Cout = 128, two 8-byte words per component and row segment -- MKM 3 -- measured 0.5-2.9 % SLOWER than the dwords and is
        //  not instantiated: profiles/r06_w4p_mkdedup_ab.txt)
        if (mkshare && a.nb == 1 && a.tc == 8 && a.Cout == 32 && a.Cin == 32 && W4P_FULL) {
            ADYOLO_WINO4P_FWD(8, false, 1, false, true, 1);
            return;
        }
    }
    if (a.nb == 2) {
        ADYOLO_WINO4P_NB(2)
    } else {
        // 32 -> 32 layers: the resident-U form where the epilogue leaves it the registers (operand sets 15 / 27 / 31 do not: 20-46
        // spilled registers and 6-14 % slower, profiles/r06_w4p_bres_ab.txt)
        if constexpr (W4P_FULL && (EPI == 15 || EPI == 27 || EPI == 31)) {
            // the same layers where resident U does not fit: whole input lines at once.  (Data gradients carry no producer affine;
            //  a launch that does takes the B-ring form below, which is the one the affine variants are checked in.)
            if (a.Cin == 32 && !a.in_scale) {
                if (a.tc == 8) ADYOLO_WINO4P_FWD(8, false, 1, false, true); else ADYOLO_WINO4P_FWD(4, false, 1, false, true);
                return;
            }
        }
        if constexpr (W4P_BRES && (EPI == 0 || EPI == 1 || EPI == 2 || EPI == 9)) {
            if (a.Cin == 32) {
                if (a.tc == 8) {
                    if (a.in_scale) ADYOLO_WINO4P_FWD(8, true, 1, true); else ADYOLO_WINO4P_FWD(8, false, 1, true);
                } else {
                    if (a.in_scale) ADYOLO_WINO4P_FWD(4, true, 1, true); else ADYOLO_WINO4P_FWD(4, false, 1, true);
                }
                return;
            }
        }
        ADYOLO_WINO4P_NB(1)
    }
#undef ADYOLO_WINO4P_NB
#undef ADYOLO_WINO4P_FWD
}

}  // namespace w4
}  // namespace adyolo
