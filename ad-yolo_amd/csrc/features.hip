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
#include "fft1200.hpp"

namespace adyolo {

#ifndef K1_FR
#define K1_FR 8
#endif
constexpr int NMEL = 64, FR = K1_FR;

__device__ __forceinline__ void atomic_max_float(float *addr, float val) {
    if (val >= 0.f) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(val));
    else atomicMin(reinterpret_cast<unsigned *>(addr), __float_as_uint(val));
}

constexpr int MAX_MELW = 1200, SPS8 = 8, MAX_CHUNKS = 224;     // LDS total 34.7 KB, 122 VGPRs: four workgroups per CU
// The transform is decimation in frequency, IN PLACE (every thread writes back to the LDS words it has just read, so one
// barrier per pass and one 22.4 KB buffer for both signals): n = n1*120 + n2*12 + n3, pass 1 = ten-point DFTs over n1
// (x W_1200^{(n mod 120) k1}), pass 2 = ten-point DFTs over n2 (x W_120^{n3 k2}), pass 3 = twelve-point DFTs over n3;
// bin k = k1 + 10 k2 + 100 k3 then sits at position k1*120 + k2*12 + k3.  Pass 1 takes its input straight from global
// memory (lane = n mod 120: consecutive samples), the Hann window is folded in as
// 0.5 - 0.5 cos(2 pi (t + 120 n1) / 1200) = 0.5 - 0.5 (cos a cos b - sin a sin b) with a fixed per thread.
#ifndef K1_SKIP
#define K1_SKIP 0      // timing-only what-if builds (tools/build_variant.sh ... -DK1_SKIP=mask; results invalid): bit 0 global loads,
#endif                 // 1 pass 1, 2 pass 2, 3 pass 3, 4 untangling, 5 mel contraction, 6 combine + log + store, 7 all audio loads hit
                       // one cache line, 8 no window, 9 no pass-1 table loads, 10 one pass-1 LDS store instead of ten
__global__ __launch_bounds__(256, 4) void feat_stft_mel_kernel(
    const float *__restrict__ audio, const long *__restrict__ clip_offset, const float *__restrict__ twiddle,
    const int *__restrict__ chunk_mel, const int *__restrict__ chunk_start, const int *__restrict__ chunk_len,
    const int *__restrict__ chunk_off, const float *__restrict__ mel_w, int n_chunks, int n_melw,
    const float *__restrict__ sc_mean, const float *__restrict__ sc_rstd, float *__restrict__ out,
    float *__restrict__ chan_max, int n_samples, int T, int layout) {
    const float2 *__restrict__ tw = reinterpret_cast<const float2 *>(twiddle);
    __shared__ __attribute__((aligned(16))) float2 buf[2 * FSIG];   // doubles as the [601][8] per-bin feature table
    __shared__ float melw[MAX_MELW];
    __shared__ __attribute__((aligned(16))) float melpart[MAX_CHUNKS * 8];    // per-piece partial sums, combined per filter in piece order (deterministic)
    __shared__ int mel_first[NMEL + 1];
    __shared__ float cmax[4][4];
    float *spec = reinterpret_cast<float *>(buf);
    const int tid = threadIdx.x;
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
    const float2 *aud = reinterpret_cast<const float2 *>(audio) + 2 * (clip_offset ? (size_t)clip_offset[b] : (size_t)b * n_samples);
    // pass-1 role: signal sf (0: W + iY, 1: Z + iX), residue st = n mod 120
    const int sf = tid >= 120 ? 1 : 0, st = tid - 120 * sf;
    const bool p1 = tid < 240;
    // hop = 600 = 5 x 120: thread (sf, st)'s samples 5..9 of frame t ARE its samples 0..4 of frame t + 1 -- only the first
    // frame of a workgroup loads all ten, every later one shifts five registers and loads five (half the global loads)
    float2 av[10];
    auto load_frame = [&](int t, int first) {
        if (p1 && !(K1_SKIP & 1)) {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                if (i < first) continue;
                int s = t * FHOP - FHOP + st + 120 * i;
                if (s < 0) s = -s;                   // np.pad(..., mode='reflect') at the start of the (virtual) clip
                if (K1_SKIP & 128) s = st;
                av[i] = aud[2 * (size_t)s + sf];
            }
        }
    };
    if (t0 < T) load_frame(t0, 0);
    __syncthreads();
    for (int fr = 0; fr < FR; ++fr) {
        const int t = t0 + fr;
        if (t >= T) break;
        // Everything below is a function of the thread index alone; left to itself the optimiser hoists all of it (window
        // factors, table addresses and loaded twiddles, LDS addresses of four passes: > 100 values) out of the frame
        // loop and then spills.  `ti` is made opaque once per frame so that the index arithmetic is redone per frame.
        int ti = tid;
        asm volatile("" : "+v"(ti));
        const int sfo = ti >= 120 ? 1 : 0, sto = ti - 120 * sfo;
        const bool p1o = ti < 240;
        // ---- pass 1 (from registers): window, ten-point DFT over n1, twiddle, store at (k1, st)
        if (p1o && !(K1_SKIP & 2)) {
            const float wc = tw[sto].x, ws = -tw[sto].y;      // cos / sin of 2 pi st / 1200 (the table holds exp(-i .))
            float2 v[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const float w = (K1_SKIP & 256) ? 1.f : 0.5f - 0.5f * (wc * RC10[i] + ws * RS10[i]);      // RS10 = -sin
                v[i] = make_float2(av[i].x * w, av[i].y * w);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) av[i] = av[i + 5];
            butterfly<10>(v);
            float2 *dst = buf + sfo * FSIG + sto + 2 * (sto / 12);
            dst[0] = v[0];
            // twiddles W^{st k} from the lane-ordered copy of the table; `sto` is opaque to the optimiser once per frame,
            // otherwise the 18 table loads of passes 1-2 (and their addresses) are hoisted out of the frame loop into ~50 live
            // registers
            __builtin_amdgcn_sched_barrier(0);
            const float2 *t1 = tw + FTW1 + sto;
#pragma unroll
            for (int k = 1; k < 10; ++k) {
                if (!(K1_SKIP & 1024) || k == 1) dst[k * (10 * FROW)] = cmul(v[k], (K1_SKIP & 512) ? make_float2(0.6f, 0.8f) : t1[(k - 1) * 120]);
                if (k % 3 == 0) __builtin_amdgcn_sched_barrier(0);      // three table loads in flight at a time
            }
        }
        __syncthreads();
        // ---- pass 2: ten-point DFT over n2 for (k1, n3), twiddle W_120^{n3 k2}
        if (p1o && !(K1_SKIP & 4)) {
            const int k1 = sto / 12, n3 = sto - 12 * k1;
            float2 *base = buf + sfo * FSIG + k1 * (10 * FROW) + n3;
            float2 v[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) v[i] = base[i * FROW];
            butterfly<10>(v);
            __builtin_amdgcn_sched_barrier(0);
            base[0] = v[0];
#pragma unroll
            for (int k = 1; k < 10; ++k) {
                base[k * FROW] = cmul(v[k], tw[FTW2 + (k - 1) * 12 + n3]);
                if (k % 3 == 0) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // ---- pass 3: twelve-point DFT over n3 for the row (k1, k2)
        if (ti < 200 && !(K1_SKIP & 8)) {
            const int f = ti >= 100 ? 1 : 0, r = ti - 100 * f;
            float4 *row = reinterpret_cast<float4 *>(buf + f * FSIG + r * FROW);
            float2 v[12];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4 q = row[i];
                v[2 * i] = make_float2(q.x, q.y);
                v[2 * i + 1] = make_float2(q.z, q.w);
            }
            butterfly<12>(v);
#pragma unroll
            for (int i = 0; i < 6; ++i) row[i] = make_float4(v[2 * i].x, v[2 * i].y, v[2 * i + 1].x, v[2 * i + 1].y);
        }
        __syncthreads();
        // ---- untangle the two packed transforms into W, Y, Z, X and form the 7 per-bin quantities (held in registers:
        //      the table overwrites the transform buffer)
        float4 qa[3], qb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = tid + 256 * i;              // (tid, not ti: these six LDS addresses are worth keeping in registers)
            if (k < FBINS && !(K1_SKIP & 16)) {
                const int pk = fpos(k), pn = fpos(k == 0 ? 0 : FN - k);
                const float2 z1 = buf[pk], z2 = buf[FSIG + pk];
                const float2 n1 = make_float2(buf[pn].x, -buf[pn].y), n2 = make_float2(buf[FSIG + pn].x, -buf[FSIG + pn].y);
                const float2 W = cscale(cadd(z1, n1), 0.5f), Y = cscale(cmi(csub(z1, n1)), 0.5f);
                const float2 Z = cscale(cadd(z2, n2), 0.5f), X = cscale(cmi(csub(z2, n2)), 0.5f);
                const float pw = W.x * W.x + W.y * W.y, py = Y.x * Y.x + Y.y * Y.y;
                const float pz = Z.x * Z.x + Z.y * Z.y, px = X.x * X.x + X.y * X.y;
                const float e = 1e-8f + (pw + (py + pz + px) / 3.0f);
                const float ie = 1.0f / e;
                qa[i] = make_float4(pw, py, pz, px);
                qb[i] = make_float4((W.x * Y.x + W.y * Y.y) * ie, (W.x * Z.x + W.y * Z.y) * ie, (W.x * X.x + W.y * X.y) * ie, 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);       // one round's temporaries at a time (96-register budget)
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
        if (fr + 1 < FR && t + 1 < T) load_frame(t + 1, 5);      // the next frame's samples travel under the mel contraction
        // sparse mel contraction: work item = a piece of <= 8 consecutive bins of one filter, all 7 quantities at once
        // (two ds_read_b128 + one weight per bin feed 8 FMAs)
        for (int ck = ti; ck < ((K1_SKIP & 32) ? 0 : n_chunks); ck += 256) {
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
        for (int o = ti; o < ((K1_SKIP & 64) ? 0 : 512); o += 256) {
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
            if (layout == 1) out[(((size_t)b * T + t) * NMEL + m) * 8 + c] = v;
            else if (c < 7) out[(((size_t)b * 7 + c) * T + t) * NMEL + m] = v;
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
