// EXPERIMENT (round 3, not built): K1 with BOTH packed signals in one thread -- real parts of the two signals in one 64-bit
// register pair, imaginary parts in another, so that every real operation of a butterfly is one v_pk_*_f32 instruction without
// swizzles or moves; passes 1-2 on 120 threads, pass 3 on 100; one 16-byte LDS element per position; 16-byte audio loads.
// Static count: 469 packed + ~35 scalar arithmetic instructions per thread and frame on HALF the threads (round 2: 309 packed +
// 159 scalar + 242 v_mov on all of them) -- the FFT's issue slots are cut by ~45 %.  Measured on MI355X (tools/feat_ab.sh,
// B = 64 x 60 s): 1.74 / 1.63 ms (nhwc8 / nchw7) at three workgroups per CU (168 VGPRs) against 1.78 / 1.62 ms for the round-2
// kernel; 2.35 ms at four workgroups per CU (128 VGPRs: 82 spilled).  Conclusion: K1 is NOT bound by the FFT's instruction
// issue -- halving it moved nothing -- but by the per-frame chain of barriers, LDS round trips and table loads with four
// frames in flight per CU.  K1_SWAP=1 (odd workgroups run the FFT passes on waves 2-3, in case waves map to SIMDs by index and
// two SIMDs carried all the FFT work): 1.77 / 1.66 ms -- no change either.  Build it as an A/B library with  copy this file over ad-yolo_amd/csrc/features.hip, bash tools/build_variant.sh k1v4 features.hip -DK1_OCC=3, restore
// K1: 4-channel STFT (n_fft = win = 1200, hop 600, periodic Hann, reflect-centred) -> log-mel (4 ch)
// + mel-scale FOA intensity vector (3 ch) -> z-score.  Replaces the NumPy float64 / librosa path of
// /root/reference/src/datasets.py:252-292 (librosa.core.stft :255, mel products :264/:275,
// power_to_db :265, scaler :289-290) and the tensorise step :158-160.
//
// One workgroup walks FR consecutive frames of one (virtual) clip.  Per frame the four real channels are packed
// as two complex signals (W + iY, Z + iX), each transformed by a 1200-point mixed-radix FFT (10 x 10 x 12 as
// in-register (5x2),(5x2),(4x3) composite butterflies: 240/240/200 butterflies per frame on 256 lanes), decimation in
// frequency and IN PLACE in one 22.4 KB LDS buffer (see the kernel's comment), twiddles from a table built in double on the
// host (L1-resident), the first pass fed straight from global memory with the Hann window folded in; the transforms are
// untangled into the four 601-bin spectra, turned into the 7 per-bin quantities
// (|W|^2,|Y|^2,|Z|^2,|X|^2, Iy/E, Iz/E, Ix/E), which overwrite the transform buffer, and contracted with the sparse
// (1165 non-zero) triangular mel filters (per-piece partial sums in LDS, combined per filter in a fixed order:
// bit-reproducible).  34.7 KB of LDS and 122 VGPRs: four workgroups (16 waves) per CU.
// A "virtual clip" is n_samples samples starting at clip_offset[b] of the audio buffer (NULL: clip b of a [B][n][4]
// batch): the 20 s / 1 s-stride training chunks of a recording (/root/reference/src/preprocess.py:13-84) are computed
// from the recording in place, each with its own reflect padding and its own top_db reference.
// power_to_db's top_db=80 clip is relative to the maximum over the whole clip and channel, so the
// un-clipped log-mel is written first with a float atomic-max per (clip, channel); adyolo_feat_finish
// applies the clip and the z-score of the four log-mel channels.
#include "common.hpp"

namespace adyolo {

constexpr int FN = 1200, FBINS = 601, FHOP = 600, NMEL = 64, FR = 8;

// Two complex values side by side -- the two packed signals of a frame (0: W + iY, 1: Z + iX) -- with the real parts in
// one 64-bit register pair and the imaginary parts in another: every real operation of a butterfly is then ONE packed
// instruction (v_pk_add / v_pk_mul / v_pk_fma_f32) working on both signals, with no lane swizzles and no register moves
// (round 2 ran one signal per thread and left the packing to the SLP vectoriser: 309 packed + 159 scalar arithmetic
// instructions plus 242 v_mov to marshal complex values into aligned pairs, per thread and frame).
typedef float v2f __attribute__((ext_vector_type(2)));
struct c2 {
    v2f re, im;
};
__device__ __forceinline__ c2 cadd(c2 a, c2 b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ c2 csub(c2 a, c2 b) { return {a.re - b.re, a.im - b.im}; }
// a * (wr + i wi), the same factor for both signals
__device__ __forceinline__ c2 cmulw(c2 a, float wr, float wi) { return {a.re * wr - a.im * wi, a.re * wi + a.im * wr}; }
// a + (-i) e * s  and  a - (-i) e * s      ((-i) e = (e.im, -e.re))
__device__ __forceinline__ c2 add_mi(c2 a, c2 e, float s) { return {a.re + e.im * s, a.im - e.re * s}; }
__device__ __forceinline__ c2 sub_mi(c2 a, c2 e, float s) { return {a.re - e.im * s, a.im + e.re * s}; }

template <int R>
__device__ __forceinline__ void butterfly(c2 *v);
template <>
__device__ __forceinline__ void butterfly<4>(c2 *v) {
    const c2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), e = csub(v[1], v[3]);
    v[0] = cadd(a, c);
    v[2] = csub(a, c);
    v[1] = {b.re + e.im, b.im - e.re};
    v[3] = {b.re - e.im, b.im + e.re};
}
template <>
__device__ __forceinline__ void butterfly<3>(c2 *v) {
    const c2 t = cadd(v[1], v[2]), e = csub(v[1], v[2]);
    const c2 m = {v[0].re - t.re * 0.5f, v[0].im - t.im * 0.5f};
    v[0] = cadd(v[0], t);
    v[1] = add_mi(m, e, 0.86602540378443864676f);
    v[2] = sub_mi(m, e, 0.86602540378443864676f);
}
template <>
__device__ __forceinline__ void butterfly<5>(c2 *v) {
    const float c1 = 0.30901699437494742410f, c2_ = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const c2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const c2 a1 = {v[0].re + t1.re * c1 + t2.re * c2_, v[0].im + t1.im * c1 + t2.im * c2_};
    const c2 a2 = {v[0].re + t1.re * c2_ + t2.re * c1, v[0].im + t1.im * c2_ + t2.im * c1};
    const c2 u1 = {t3.re * s1 + t4.re * s2, t3.im * s1 + t4.im * s2};
    const c2 u2 = {t3.re * s2 - t4.re * s1, t3.im * s2 - t4.im * s1};
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = {a1.re + u1.im, a1.im - u1.re};
    v[4] = {a1.re - u1.im, a1.im + u1.re};
    v[2] = {a2.re + u2.im, a2.im - u2.re};
    v[3] = {a2.re - u2.im, a2.im + u2.re};
}
template <>
__device__ __forceinline__ void butterfly<2>(c2 *v) {
    const c2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}

// exp(-2 pi i m / R) tables for the in-register composite butterflies (folded to immediates after unrolling)
__device__ constexpr float RC10[10] = {1.f, 0.809016994f, 0.309016994f, -0.309016994f, -0.809016994f, -1.f, -0.809016994f, -0.309016994f, 0.309016994f, 0.809016994f};
__device__ constexpr float RS10[10] = {0.f, -0.587785252f, -0.951056516f, -0.951056516f, -0.587785252f, 0.f, 0.587785252f, 0.951056516f, 0.951056516f, 0.587785252f};
__device__ constexpr float RC12[12] = {1.f, 0.866025404f, 0.5f, 0.f, -0.5f, -0.866025404f, -1.f, -0.866025404f, -0.5f, 0.f, 0.5f, 0.866025404f};
__device__ constexpr float RS12[12] = {0.f, -0.5f, -0.866025404f, -1.f, -0.866025404f, -0.5f, 0.f, 0.5f, 0.866025404f, 1.f, 0.866025404f, 0.5f};
template <int R>
__device__ __forceinline__ float rootc(int m);
template <int R>
__device__ __forceinline__ float roots(int m);
template <>
__device__ __forceinline__ float rootc<10>(int m) { return RC10[m]; }
template <>
__device__ __forceinline__ float roots<10>(int m) { return RS10[m]; }
template <>
__device__ __forceinline__ float rootc<12>(int m) { return RC12[m]; }
template <>
__device__ __forceinline__ float roots<12>(int m) { return RS12[m]; }

// radix R1*R2 butterfly entirely in registers (Cooley-Tukey: n = n2 + R2 n1, k = k1 + R1 k2)
template <int R1, int R2>
__device__ __forceinline__ void butterfly_composite(c2 *v) {
    constexpr int R = R1 * R2;
    c2 y[R2][R1];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
        c2 t[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) t[n1] = v[n2 + R2 * n1];
        butterfly<R1>(t);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1)
            y[n2][k1] = (n2 * k1 == 0) ? t[k1] : cmulw(t[k1], rootc<R>((n2 * k1) % R), roots<R>((n2 * k1) % R));
    }
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        c2 t[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) t[n2] = y[n2][k1];
        butterfly<R2>(t);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = t[k2];
    }
}
template <>
__device__ __forceinline__ void butterfly<10>(c2 *v) { butterfly_composite<5, 2>(v); }
template <>
__device__ __forceinline__ void butterfly<12>(c2 *v) { butterfly_composite<4, 3>(v); }

__device__ __forceinline__ void atomic_max_float(float *addr, float val) {
    if (val >= 0.f) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(val));
    else atomicMin(reinterpret_cast<unsigned *>(addr), __float_as_uint(val));
}

constexpr int MAX_MELW = 1200, SPS8 = 8, MAX_CHUNKS = 224;     // LDS total 33 KB: four workgroups per CU
// LDS image of the transform: ONE 16-byte element {re0, re1, im0, im1} per position (both signals), position
// p = k1*120 + k2*12 + k3 at element p + (p/12) = k1*130 + k2*13 + k3: rows of 12 padded to 13 elements (52 dwords), which
// puts the eight lanes of a ds_read_b128 group of the twelve-point pass on eight distinct bank quads
constexpr int FROW = 13, FSIG = 10 * 10 * FROW;          // 1300 elements = 20.8 KB
__device__ __forceinline__ int fpos(int k) {             // where bin k ends up after the three in-place passes
    const int k1 = k % 10, q = k / 10;
    return k1 * (10 * FROW) + (q % 10) * FROW + q / 10;
}
__device__ __forceinline__ c2 lds_get(const float4 *p) {
    const float4 q = *p;
    return {v2f{q.x, q.y}, v2f{q.z, q.w}};
}
__device__ __forceinline__ void lds_put(float4 *p, c2 a) { *p = make_float4(a.re.x, a.re.y, a.im.x, a.im.y); }

// The transform is decimation in frequency, IN PLACE (every thread writes back to the LDS elements it has just read, so one
// barrier per pass and one 20.8 KB buffer): n = n1*120 + n2*12 + n3, pass 1 = ten-point DFTs over n1
// (x W_1200^{(n mod 120) k1}), pass 2 = ten-point DFTs over n2 (x W_120^{n3 k2}), pass 3 = twelve-point DFTs over n3;
// bin k = k1 + 10 k2 + 100 k3 then sits at position k1*120 + k2*12 + k3.  Pass 1 takes its input straight from global
// memory (lane = n mod 120: one 16-byte load = the four channels of a sample, consecutive lanes = consecutive samples), the
// Hann window is folded in as 0.5 - 0.5 cos(2 pi (t + 120 n1) / 1200) = 0.5 - 0.5 (cos a cos b - sin a sin b), a fixed per
// thread.  Passes 1-2 occupy 120 threads (two waves), pass 3 100; the untangling, the mel contraction and the output all 256.
#ifndef K1_OCC
#define K1_OCC 3               // workgroups per CU the register budget is set for (4: 128 VGPRs, 82 spilled)
#endif
#ifndef K1_SWAP
#define K1_SWAP 0              // 1: odd workgroups run the FFT passes on their waves 2-3 instead of 0-1 (SIMD balance)
#endif
__global__ __launch_bounds__(256, K1_OCC) void feat_stft_mel_kernel(
    const float *__restrict__ audio, const long *__restrict__ clip_offset, const float *__restrict__ twiddle,
    const int *__restrict__ chunk_mel, const int *__restrict__ chunk_start, const int *__restrict__ chunk_len,
    const int *__restrict__ chunk_off, const float *__restrict__ mel_w, int n_chunks, int n_melw,
    const float *__restrict__ sc_mean, const float *__restrict__ sc_rstd, float *__restrict__ out,
    float *__restrict__ chan_max, int n_samples, int T, int layout) {
    const char *__restrict__ twb = reinterpret_cast<const char *>(twiddle);       // uniform base + 32-bit offsets
    __shared__ __attribute__((aligned(16))) float4 buf[FSIG];       // doubles as the [601][8] per-bin feature table
    __shared__ float melw[MAX_MELW];
    __shared__ __attribute__((aligned(16))) float melpart[MAX_CHUNKS * 8];    // per-piece partial sums, combined per filter in piece order (deterministic)
    __shared__ int mel_first[NMEL + 1];
    __shared__ float cmax[4][4];
    float *spec = reinterpret_cast<float *>(buf);
    // (waves of a workgroup go to the SIMDs by wave index: with every workgroup's FFT passes on waves 0-1, two of the four SIMDs
    //  of a CU would carry all the FFT work -- K1_SWAP moves them to waves 2-3 in every other workgroup)
    const int tid = K1_SWAP ? (int)((threadIdx.x + 128u * ((blockIdx.x + blockIdx.y) & 1u)) & 255u) : (int)threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * FR;
    for (int i = tid; i < n_melw; i += 256) melw[i] = mel_w[i];
    // pieces are stored filter after filter and every filter has at least one: first piece of every filter
    for (int ck = tid; ck < n_chunks; ck += 256) {
        const int m = chunk_mel[ck];
        if (ck == 0 || chunk_mel[ck - 1] != m) mel_first[m] = ck;
    }
    if (tid == 0) mel_first[NMEL] = n_chunks;
    float lmax = -INFINITY;                       // lanes with (tid & 7) < 4 track channel tid & 7 (layout-independent)
    // a virtual clip = n_samples samples starting at clip_offset[b] (chunks of a longer recording) or clip b of the batch
    const char *audb = reinterpret_cast<const char *>(audio) + 16 * (clip_offset ? (size_t)clip_offset[b] : (size_t)b * n_samples);
    const bool p1 = tid < 120;                    // passes 1-2: thread = residue n mod 120, both signals
    float4 av[10];
    auto load_frame = [&](int t) {
        if (p1) {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                int s = t * FHOP - FHOP + tid + 120 * i;
                if (s < 0) s = -s;                   // np.pad(..., mode='reflect') at the start of the (virtual) clip
                av[i] = *reinterpret_cast<const float4 *>(audb + (unsigned)s * 16u);
            }
        }
    };
    if (t0 < T) load_frame(t0);
    __syncthreads();
    for (int fr = 0; fr < FR; ++fr) {
        const int t = t0 + fr;
        if (t >= T) break;
        // Everything below is a function of the thread index alone; left to itself the optimiser hoists all of it (window
        // factors, table addresses and loaded twiddles, LDS addresses of four passes: > 100 values) out of the frame
        // loop and then spills.  `ti` is made opaque once per frame so that the index arithmetic is redone per frame.
        int ti = tid;
        asm volatile("" : "+v"(ti));
        const bool p1o = ti < 120;
        // ---- pass 1 (from registers): window, ten-point DFT over n1, twiddle, store at (k1, st)
        if (p1o) {
            const float2 w0 = *reinterpret_cast<const float2 *>(twb + (unsigned)ti * 8u);
            const float wc = w0.x, ws = -w0.y;      // cos / sin of 2 pi st / 1200 (the table holds exp(-i .))
            c2 v[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const float w = 0.5f - 0.5f * (wc * RC10[i] + ws * RS10[i]);      // RS10 = -sin
                v[i].re = v2f{av[i].x * w, av[i].z * w};      // sample = (W, Y, Z, X): signal 0 = W + iY, signal 1 = Z + iX
                v[i].im = v2f{av[i].y * w, av[i].w * w};
            }
            butterfly<10>(v);
            float4 *dst = buf + ti + (ti / 12) * (FROW - 12);
            lds_put(dst, v[0]);
            // twiddle index (st * k) mod 1200, stepped; `ti` is opaque to the optimiser once per frame, otherwise the 18
            // table loads of passes 1-2 (and their addresses) are hoisted out of the frame loop into ~50 live registers
            __builtin_amdgcn_sched_barrier(0);
            int idx = 0;
#pragma unroll
            for (int k = 1; k < 10; ++k) {
                idx += ti;
                if (idx >= FN) idx -= FN;
                const float2 w = *reinterpret_cast<const float2 *>(twb + (unsigned)idx * 8u);
                lds_put(dst + k * (10 * FROW), cmulw(v[k], w.x, w.y));
                if (k % 3 == 0) __builtin_amdgcn_sched_barrier(0);      // three table loads in flight at a time
            }
        }
        __syncthreads();
        // ---- pass 2: ten-point DFT over n2 for (k1, n3), twiddle W_120^{n3 k2}
        if (p1o) {
            const int k1 = ti / 12, n3 = ti - 12 * k1;
            float4 *base = buf + k1 * (10 * FROW) + n3;
            c2 v[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) v[i] = lds_get(base + i * FROW);
            butterfly<10>(v);
            __builtin_amdgcn_sched_barrier(0);
            lds_put(base, v[0]);
#pragma unroll
            for (int k = 1; k < 10; ++k) {
                const float2 w = *reinterpret_cast<const float2 *>(twb + (unsigned)(n3 * (k * 10)) * 8u);      // n3 k 10 <= 1080 < 1200
                lds_put(base + k * FROW, cmulw(v[k], w.x, w.y));
                if (k % 3 == 0) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // ---- pass 3: twelve-point DFT over n3 for the row (k1, k2)
        if (ti < 100) {
            float4 *row = buf + ti * FROW;
            c2 v[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) v[i] = lds_get(row + i);
            butterfly<12>(v);
#pragma unroll
            for (int i = 0; i < 12; ++i) lds_put(row + i, v[i]);
        }
        __syncthreads();
        // ---- untangle the two packed transforms into W, Y, Z, X and form the 7 per-bin quantities (held in registers:
        //      the table overwrites the transform buffer).  z = element at bin k, n = conj-partner at bin N - k:
        //      S = z + conj(n) = 2 (W | Z),  D = z - conj(n) = 2 i (Y | X)
        float4 qa[3], qb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = ti + 256 * i;
            if (k < FBINS) {
                const float4 z = buf[fpos(k)], n = buf[fpos(k == 0 ? 0 : FN - k)];
                const v2f sre = v2f{z.x, z.y} + v2f{n.x, n.y}, sim = v2f{z.z, z.w} - v2f{n.z, n.w};
                const v2f dre = v2f{z.x, z.y} - v2f{n.x, n.y}, dim = v2f{z.z, z.w} + v2f{n.z, n.w};
                const v2f ps = (sre * sre + sim * sim) * 0.25f;       // |W|^2, |Z|^2
                const v2f pd = (dre * dre + dim * dim) * 0.25f;       // |Y|^2, |X|^2
                const float e = 1e-8f + (ps.x + (pd.x + ps.y + pd.y) * 0.333333343f);
                const float ie = 0.25f * __builtin_amdgcn_rcpf(e);
                // Re(conj(W) Y) = (S0.re D0.im - S0.im D0.re) / 4,  Re(conj(W) Z) = (S0.re S1.re + S0.im S1.im) / 4, X like Y
                qa[i] = make_float4(ps.x, pd.x, ps.y, pd.y);
                qb[i] = make_float4((sre.x * dim.x - sim.x * dre.x) * ie, (sre.x * sre.y + sim.x * sim.y) * ie,
                                    (sre.x * dim.y - sim.x * dre.y) * ie, 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);       // one round's temporaries at a time
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = ti + 256 * i;
            if (k < FBINS) {
                float4 *sp = reinterpret_cast<float4 *>(&spec[k * SPS8]);
                sp[0] = qa[i];
                sp[1] = qb[i];
            }
        }
        __syncthreads();
        if (fr + 1 < FR && t + 1 < T) load_frame(t + 1);      // the next frame's samples travel under the mel contraction
        // sparse mel contraction: work item = a piece of <= 8 consecutive bins of one filter, all 7 quantities at once
        // (two ds_read_b128 + one weight per bin feed 8 FMAs)
        for (int ck = ti; ck < n_chunks; ck += 256) {
            const int st_ = chunk_start[ck], ln = chunk_len[ck], of = chunk_off[ck];
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
            const float4 *sp = reinterpret_cast<const float4 *>(spec) + 2 * st_;
            // neighbouring lanes own neighbouring pieces (bin ranges 8 apart = 256 B apart in the table): every lane walks
            // its piece from a different starting bin so that a ds_read_b128 group does not pile onto one bank quad
            int j = ck & 7;
            if (j >= ln) j = 0;
            for (int i = 0; i < ln; ++i) {
                const float w = melw[of + j];
                const float4 u = sp[2 * j], v = sp[2 * j + 1];
                j = j + 1 == ln ? 0 : j + 1;
                a0.x += w * u.x; a0.y += w * u.y; a0.z += w * u.z; a0.w += w * u.w;
                a1.x += w * v.x; a1.y += w * v.y; a1.z += w * v.z; a1.w += w * v.w;
            }
            float4 *mp = reinterpret_cast<float4 *>(melpart) + 2 * ck;
            mp[0] = a0;
            mp[1] = a1;
        }
        __syncthreads();
        float *out_bt = out + (layout == 1 ? ((size_t)b * T + t) * (NMEL * 8) : (size_t)b * 7 * T * NMEL + (size_t)t * NMEL);
        for (int o = ti; o < 512; o += 256) {
            const int m = o >> 3, c = o & 7;
            float acc = 0.f;
            if (c < 7)
                for (int ck = mel_first[m]; ck < mel_first[m + 1]; ++ck) acc += melpart[ck * 8 + c];
            float v = 0.f;
            if (c < 4) {
                v = 10.0f * log10f(fmaxf(acc, 1e-10f));
                lmax = fmaxf(lmax, v);
            } else if (c < 7) {
                v = (acc - sc_mean[c * NMEL + m]) * sc_rstd[c * NMEL + m];
            }
            if (layout == 1) out_bt[o] = v;
            else if (c < 7) out_bt[(unsigned)(c * T * NMEL + m)] = v;
        }
        // (the next frame's pass 1 writes the transform buffer, which the mel loop above has finished reading at the last
        //  barrier; melpart is rewritten four barriers from here)
    }
    // per-channel maximum of the un-clipped log-mel: lanes with equal (tid & 7) hold the same channel
    for (int o = 32; o >= 8; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < 4) cmax[wave][lane] = lmax;
    __syncthreads();
    if (tid < 4) {
        const float v = fmaxf(fmaxf(cmax[0][tid], cmax[1][tid]), fmaxf(cmax[2][tid], cmax[3][tid]));
        if (v > -INFINITY) atomic_max_float(&chan_max[b * 4 + tid], v);
    }
}

__global__ __launch_bounds__(256) void feat_finish_kernel(float *__restrict__ out, const float *__restrict__ chan_max,
                                                          const float *__restrict__ sc_mean,
                                                          const float *__restrict__ sc_rstd, int T, int layout,
                                                          long per_clip) {
    // per_clip = 4 * T * 64 log-mel values of one clip
    const int b = blockIdx.y;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_clip; i += (long)gridDim.x * blockDim.x) {
        int c, m;
        size_t o;
        if (layout == 0) {
            m = (int)(i & 63);
            const long tt = (i >> 6) % T;
            c = (int)((i >> 6) / T);
            o = (((size_t)b * 7 + c) * T + tt) * NMEL + m;
        } else {
            c = (int)(i & 3);
            m = (int)((i >> 2) & 63);
            const long tt = i >> 8;
            o = (((size_t)b * T + tt) * NMEL + m) * 8 + c;
        }
        const float floor_db = chan_max[b * 4 + c] - 80.0f;
        const float v = fmaxf(out[o], floor_db);
        out[o] = (v - sc_mean[c * NMEL + m]) * sc_rstd[c * NMEL + m];
    }
}

}  // namespace adyolo

using namespace adyolo;

extern "C" int adyolo_feat_stft_mel(const float *audio, const int64_t *clip_offset, const float *twiddle,
                                    const int32_t *chunk_mel, const int32_t *chunk_start, const int32_t *chunk_len,
                                    const int32_t *chunk_off, const float *mel_w, int n_chunks, int n_mel_w,
                                    const float *scaler_mean, const float *scaler_rstd, float *out, float *chan_max,
                                    int B, int n_samples, int layout, void *stream) {
    ADYOLO_REQUIRE(audio && twiddle && chunk_mel && chunk_start && chunk_len && chunk_off && mel_w &&
                       scaler_mean && scaler_rstd && out && chan_max,
                   ADYOLO_EINVAL, "feat_stft_mel: null pointer");
    ADYOLO_REQUIRE(B > 0 && n_samples >= 1200 && n_samples % FHOP == 0 && (layout == 0 || layout == 1), ADYOLO_EINVAL,
                   "feat_stft_mel: n_samples=%d must be a multiple of 600 and >= 1200", n_samples);
    ADYOLO_REQUIRE(n_chunks > 0 && n_chunks <= MAX_CHUNKS && n_mel_w > 0 && n_mel_w <= MAX_MELW, ADYOLO_ENOSUP,
                   "feat_stft_mel: %d mel weights / %d pieces exceed the LDS tables (%d / %d)", n_mel_w, n_chunks, MAX_MELW,
                   MAX_CHUNKS);
    hipStream_t st = as_stream(stream);
    const int T = n_samples / FHOP;
    int rc0 = fill32(chan_max, 0xFF800000u, (size_t)B * 4, st);          // -inf (a kernel, not a memset node: see common.hpp)
    if (rc0) return rc0;
    hipLaunchKernelGGL(feat_stft_mel_kernel, dim3(cdiv(T, FR), B), dim3(256), 0, st, audio, reinterpret_cast<const long *>(clip_offset), twiddle, chunk_mel,
                       chunk_start, chunk_len, chunk_off, mel_w, n_chunks, n_mel_w, scaler_mean, scaler_rstd, out,
                       chan_max, n_samples, T, layout);
    return check_launch("feat_stft_mel");
}

extern "C" int adyolo_feat_finish(float *out, const float *chan_max, const float *scaler_mean,
                                  const float *scaler_rstd, int B, int T, int layout, void *stream) {
    ADYOLO_REQUIRE(out && chan_max && scaler_mean && scaler_rstd && B > 0 && T > 0 && (layout == 0 || layout == 1),
                   ADYOLO_EINVAL, "feat_finish: bad arguments");
    const long per_clip = 4L * T * NMEL;
    long g = (per_clip + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(feat_finish_kernel, dim3((unsigned)g, B), dim3(256), 0, as_stream(stream), out, chan_max,
                       scaler_mean, scaler_rstd, T, layout, per_clip);
    return check_launch("feat_finish");
}
