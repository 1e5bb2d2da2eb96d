// K1: 4-channel STFT (n_fft = win = 1200, hop 600, periodic Hann, reflect-centred) -> log-mel (4 ch)
// + mel-scale FOA intensity vector (3 ch) -> z-score.  Replaces the NumPy float64 / librosa path of
// /root/reference/src/datasets.py:252-292 (librosa.core.stft :255, mel products :264/:275,
// power_to_db :265, scaler :289-290) and the tensorise step :158-160.
//
// One workgroup walks FR consecutive frames of one clip.  Per frame the four real channels are packed
// as two complex signals (W + iY, Z + iX), each transformed by a 1200-point mixed-radix Stockham FFT
// (10 x 10 x 12 as in-register (5x2),(5x2),(4x3) composite butterflies: 240/240/200 butterflies per frame on 256
// lanes; auto-sorting, ping-pong in LDS, twiddles from an LDS-resident table built in double
// on the host), untangled into the four 601-bin spectra, turned into the 7 per-bin quantities
// (|W|^2,|Y|^2,|Z|^2,|X|^2, Iy/E, Iz/E, Ix/E) in LDS, and contracted with the sparse (1165 non-zero)
// triangular mel filters (per-piece partial sums in LDS, combined per filter in a fixed order: bit-reproducible).
// Audio samples are read once as float4 (all four channels of a sample).
// power_to_db's top_db=80 clip is relative to the maximum over the whole clip and channel, so the
// un-clipped log-mel is written first with a float atomic-max per (clip, channel); adyolo_feat_finish
// applies the clip and the z-score of the four log-mel channels.
#include "common.hpp"

namespace adyolo {

constexpr int FN = 1200, FBINS = 601, FHOP = 600, NMEL = 64, FR = 8;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmi(float2 a) { return make_float2(a.y, -a.x); }      // -i * a
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

template <int R>
__device__ __forceinline__ void butterfly(float2 *v);
template <>
__device__ __forceinline__ void butterfly<4>(float2 *v) {
    const float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = cmi(csub(v[1], v[3]));
    v[0] = cadd(a, c); v[1] = cadd(b, d); v[2] = csub(a, c); v[3] = csub(b, d);
}
template <>
__device__ __forceinline__ void butterfly<3>(float2 *v) {
    const float2 t = cadd(v[1], v[2]);
    const float2 q = cscale(cmi(csub(v[1], v[2])), 0.86602540378443864676f);
    const float2 m = csub(v[0], cscale(t, 0.5f));
    v[0] = cadd(v[0], t); v[1] = cadd(m, q); v[2] = csub(m, q);
}
template <>
__device__ __forceinline__ void butterfly<5>(float2 *v) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const float2 a1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
    const float2 a2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
    const float2 b1 = cmi(make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
    const float2 b2 = cmi(make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = cadd(a1, b1); v[4] = csub(a1, b1);
    v[2] = cadd(a2, b2); v[3] = csub(a2, b2);
}

template <>
__device__ __forceinline__ void butterfly<2>(float2 *v) {
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}

// exp(-2 pi i m / R) tables for the in-register composite butterflies (folded to immediates after unrolling)
__device__ constexpr float RC10[10] = {1.f, 0.809016994f, 0.309016994f, -0.309016994f, -0.809016994f, -1.f, -0.809016994f, -0.309016994f, 0.309016994f, 0.809016994f};
__device__ constexpr float RS10[10] = {0.f, -0.587785252f, -0.951056516f, -0.951056516f, -0.587785252f, 0.f, 0.587785252f, 0.951056516f, 0.951056516f, 0.587785252f};
__device__ constexpr float RC12[12] = {1.f, 0.866025404f, 0.5f, 0.f, -0.5f, -0.866025404f, -1.f, -0.866025404f, -0.5f, 0.f, 0.5f, 0.866025404f};
__device__ constexpr float RS12[12] = {0.f, -0.5f, -0.866025404f, -1.f, -0.866025404f, -0.5f, 0.f, 0.5f, 0.866025404f, 1.f, 0.866025404f, 0.5f};
template <int R>
__device__ __forceinline__ float2 root(int m);
template <>
__device__ __forceinline__ float2 root<10>(int m) { return make_float2(RC10[m], RS10[m]); }
template <>
__device__ __forceinline__ float2 root<12>(int m) { return make_float2(RC12[m], RS12[m]); }

// radix R1*R2 butterfly entirely in registers (Cooley-Tukey: n = n2 + R2 n1, k = k1 + R1 k2)
template <int R1, int R2>
__device__ __forceinline__ void butterfly_composite(float2 *v) {
    constexpr int R = R1 * R2;
    float2 y[R2][R1];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
        float2 t[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) t[n1] = v[n2 + R2 * n1];
        butterfly<R1>(t);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) y[n2][k1] = (n2 * k1 == 0) ? t[k1] : cmul(t[k1], root<R>((n2 * k1) % R));
    }
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        float2 t[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) t[n2] = y[n2][k1];
        butterfly<R2>(t);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = t[k2];
    }
}
template <>
__device__ __forceinline__ void butterfly<10>(float2 *v) { butterfly_composite<5, 2>(v); }
template <>
__device__ __forceinline__ void butterfly<12>(float2 *v) { butterfly_composite<4, 3>(v); }

// one Stockham stage over both packed signals (in/out: [2][FN]); one radix-R butterfly per thread and pass
template <int R, int Ns>
__device__ __forceinline__ void fft_stage(const float2 *__restrict__ in, float2 *__restrict__ out,
                                          const float2 *__restrict__ tw, int tid) {
    constexpr int T = FN / R;
    constexpr int tstep = FN / (Ns * R);
    for (int j2 = tid; j2 < 2 * T; j2 += 256) {
        const int f = j2 >= T ? 1 : 0;
        const int j = j2 - f * T;
        const int k = j % Ns;
        const float2 *src = in + f * FN + j;
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = src[r * T];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) v[r] = cmul(v[r], tw[r * k * tstep]);
        }
        butterfly<R>(v);
        float2 *dst = out + f * FN + (j / Ns) * Ns * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) dst[r * Ns] = v[r];
    }
}

__device__ __forceinline__ void atomic_max_float(float *addr, float val) {
    if (val >= 0.f) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(val));
    else atomicMin(reinterpret_cast<unsigned *>(addr), __float_as_uint(val));
}

constexpr int MAX_MELW = 1280, SPS8 = 8, MAX_CHUNKS = 160;

__global__ __launch_bounds__(256, 3) void feat_stft_mel_kernel(
    const float *__restrict__ audio, const float *__restrict__ twiddle, const float *__restrict__ window,
    const int *__restrict__ chunk_mel, const int *__restrict__ chunk_start, const int *__restrict__ chunk_len,
    const int *__restrict__ chunk_off, const float *__restrict__ mel_w, int n_chunks, int n_melw,
    const float *__restrict__ sc_mean, const float *__restrict__ sc_rstd, float *__restrict__ out,
    float *__restrict__ chan_max, int n_samples, int T, int layout) {
    // twiddles and window stay in global memory (14 KB, L1/L2 resident): 46 KB of LDS -> three workgroups per CU
    const float2 *__restrict__ tw = reinterpret_cast<const float2 *>(twiddle);
    const float *__restrict__ win = window;
    __shared__ float2 bufA[2 * FN + 4];          // doubles as the [601][8] per-bin feature table after the last FFT stage
    __shared__ float2 bufB[2 * FN];
    __shared__ float melw[MAX_MELW];
    __shared__ float melpart[MAX_CHUNKS * 8];    // per-piece partial sums, combined per filter in piece order (deterministic)
    __shared__ int mel_first[NMEL + 1];
    __shared__ float cmax[4][4];
    float *spec = reinterpret_cast<float *>(bufA);
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * FR;
    for (int i = tid; i < n_melw; i += 256) melw[i] = mel_w[i];
    // pieces are stored filter after filter: first piece of every filter (filters without a piece get an empty range)
    for (int m = tid; m <= NMEL; m += 256) {
        int f = n_chunks;
        for (int ck = n_chunks - 1; ck >= 0; --ck)
            if (chunk_mel[ck] >= m) f = ck;
        mel_first[m] = f;
    }
    float lmax = -INFINITY;                       // lanes with (tid & 7) < 4 track channel tid & 7 (layout-independent)
    const float4 *aud = reinterpret_cast<const float4 *>(audio) + (size_t)b * n_samples;
    // audio of the next frame is prefetched into registers while the current frame is transformed
    float4 av[5];
    auto load_frame = [&](int t) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int n = tid + i * 256;
            if (n < FN) {
                int s = t * FHOP - FHOP + n;
                if (s < 0) s = -s;                   // np.pad(..., mode='reflect')
                av[i] = aud[s];
            }
        }
    };
    if (t0 < T) load_frame(t0);
    __syncthreads();
    for (int fr = 0; fr < FR; ++fr) {
        const int t = t0 + fr;
        if (t >= T) break;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int n = tid + i * 256;
            if (n < FN) {
                const float w = win[n];
                bufA[n] = make_float2(av[i].x * w, av[i].y * w);
                bufA[FN + n] = make_float2(av[i].z * w, av[i].w * w);
            }
        }
        __syncthreads();
        if (fr + 1 < FR && t + 1 < T) load_frame(t + 1);
        fft_stage<10, 1>(bufA, bufB, tw, tid);
        __syncthreads();
        fft_stage<10, 10>(bufB, bufA, tw, tid);
        __syncthreads();
        fft_stage<12, 100>(bufA, bufB, tw, tid);
        __syncthreads();
        for (int k = tid; k < FBINS; k += 256) {
            const int kn = k == 0 ? 0 : FN - k;
            const float2 z1 = bufB[k], z2 = bufB[FN + k];
            const float2 n1 = make_float2(bufB[kn].x, -bufB[kn].y), n2 = make_float2(bufB[FN + kn].x, -bufB[FN + kn].y);
            const float2 W = cscale(cadd(z1, n1), 0.5f), Y = cscale(cmi(csub(z1, n1)), 0.5f);
            const float2 Z = cscale(cadd(z2, n2), 0.5f), X = cscale(cmi(csub(z2, n2)), 0.5f);
            const float pw = W.x * W.x + W.y * W.y, py = Y.x * Y.x + Y.y * Y.y;
            const float pz = Z.x * Z.x + Z.y * Z.y, px = X.x * X.x + X.y * X.y;
            const float e = 1e-8f + (pw + (py + pz + px) / 3.0f);
            const float ie = 1.0f / e;
            float4 *sp = reinterpret_cast<float4 *>(&spec[k * SPS8]);
            sp[0] = make_float4(pw, py, pz, px);
            sp[1] = make_float4((W.x * Y.x + W.y * Y.y) * ie, (W.x * Z.x + W.y * Z.y) * ie, (W.x * X.x + W.y * X.y) * ie, 0.f);
        }
        __syncthreads();
        // sparse mel contraction: work item = (chunk of <= 16 consecutive bins of one filter, feature channel)
        for (int it = tid; it < n_chunks * 7; it += 256) {
            const int ck = it / 7, c = it - ck * 7;
            const int st = chunk_start[ck], ln = chunk_len[ck], of = chunk_off[ck];
            float s = 0.f;
#pragma unroll 4
            for (int i = 0; i < ln; ++i) s += melw[of + i] * spec[(st + i) * SPS8 + c];
            melpart[ck * 8 + c] = s;
        }
        __syncthreads();
        for (int o = tid; o < 512; o += 256) {
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
        __syncthreads();
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

extern "C" int adyolo_feat_stft_mel(const float *audio, const float *twiddle, const float *window,
                                    const int32_t *chunk_mel, const int32_t *chunk_start, const int32_t *chunk_len,
                                    const int32_t *chunk_off, const float *mel_w, int n_chunks, int n_mel_w,
                                    const float *scaler_mean, const float *scaler_rstd, float *out, float *chan_max,
                                    int B, int n_samples, int layout, void *stream) {
    ADYOLO_REQUIRE(audio && twiddle && window && chunk_mel && chunk_start && chunk_len && chunk_off && mel_w &&
                       scaler_mean && scaler_rstd && out && chan_max,
                   ADYOLO_EINVAL, "feat_stft_mel: null pointer");
    ADYOLO_REQUIRE(B > 0 && n_samples >= 1200 && n_samples % FHOP == 0 && (layout == 0 || layout == 1), ADYOLO_EINVAL,
                   "feat_stft_mel: n_samples=%d must be a multiple of 600 and >= 1200", n_samples);
    ADYOLO_REQUIRE(n_chunks > 0 && n_chunks <= MAX_CHUNKS && n_mel_w > 0 && n_mel_w <= MAX_MELW, ADYOLO_ENOSUP,
                   "feat_stft_mel: %d mel weights / %d pieces exceed the LDS tables (%d / %d)", n_mel_w, n_chunks, MAX_MELW,
                   MAX_CHUNKS);
    hipStream_t st = as_stream(stream);
    const int T = n_samples / FHOP;
    hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(chan_max), (int)0xFF800000, (size_t)B * 4, st);
    if (e != hipSuccess) {
        set_error("feat_stft_mel: memset failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    hipLaunchKernelGGL(feat_stft_mel_kernel, dim3(cdiv(T, FR), B), dim3(256), 0, st, audio, twiddle, window, chunk_mel,
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
