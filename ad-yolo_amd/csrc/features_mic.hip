// K1m: GCC-PHAT features of 4-channel microphone-array (MIC format) audio -- the second half of BASELINE config 5's
// feature set ("DCASE2022 MIC (GCC-PHAT features)"; the first half, the four log-mel channels, is K1 of features.hip run on
// the same audio).  NOT IN THE REFERENCE: /root/reference hard-codes the FOA format (src/datasets.py:36-37,55; the
// --feature switch is commented out, src/main.py:40), so there is no reference code to restate and no golden vector:
// PARITY UNPINNED.  The definition is the one of the DCASE2022 SELD baseline that the reference's README credits for its
// metrics code (README.md:156, sharathadavanne/seld-dcase2022, cls_feature_class.py::_get_gcc), restated in
// oracle/features.py::gcc_phat:
//     for every microphone pair m < n:  R = conj(X_m) X_n;  cc = irfft(exp(i angle(R)))  (1200 lags);
//     feature[t][lag bin] = concat(cc[-32:], cc[:32])           -> (T, 64, 6), then the scaler's z-score
// with X the same STFT as K1 (n_fft = win = 1200, hop 600, periodic Hann, reflect-centred).
//
// One workgroup walks GR consecutive frames of one clip.  Per frame: (1) the two packed 1200-point transforms of K1 (same
// in-place decimation-in-frequency passes, fft1200.hpp); (2) untangling into the four spectra and the six unit-phase cross
// spectra P_p[k]; two real inverse transforms are packed into ONE complex transform -- H = P_a + i P_b extended to 1200
// bins by Hermitian symmetry gives irfft(P_a) + i irfft(P_b) -- and the inverse is run as conj(FFT(conj(H))) / N on the
// SAME forward passes, their first pass reading its input from LDS (natural order) instead of global memory; (3) two more
// rounds of passes: pair-packs (a, b) together (the two sequences of a round), then pack c; the 64 lags wanted sit at known
// positions of the transform buffer.  51 KB of LDS, three workgroups per CU.
#include "common.hpp"
#include "fft1200.hpp"

namespace adyolo {

constexpr int GR = 4;                       // frames per workgroup
constexpr int NLAG = 64;

// unit-phase of z (exp(i angle(z)); angle(0) = 0 -> 1, like numpy)
__device__ __forceinline__ float2 unit_phase(float2 z) {
    const float m = fmaxf(fabsf(z.x), fabsf(z.y));
    if (!(m > 0.f)) return make_float2(1.f, 0.f);
    const float rx = z.x / m, ry = z.y / m;
    const float inv = rsqrtf(rx * rx + ry * ry);
    return make_float2(rx * inv, ry * inv);
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// passes 2 and 3 of the in-place transform of the two sequences in buf (pass 1 has just written them); ends with a barrier
__device__ __forceinline__ void fft_passes_2_3(float2 *buf, const float2 *__restrict__ tw, int ti, int nseq) {
    const int sfo = ti >= 120 ? 1 : 0, sto = ti - 120 * sfo;
    __syncthreads();
    if (ti < 240 && sfo < nseq) {
        const int k1 = sto / 12, n3 = sto - 12 * k1;
        float2 *base = buf + sfo * FSIG + k1 * (10 * FROW) + n3;
        float2 v[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) v[i] = base[i * FROW];
        butterfly<10>(v);
        base[0] = v[0];
#pragma unroll
        for (int k = 1; k < 10; ++k) base[k * FROW] = cmul(v[k], tw[FTW2 + (k - 1) * 12 + n3]);
    }
    __syncthreads();
    if (ti < 200) {
        const int f = ti >= 100 ? 1 : 0, r = ti - 100 * f;
        if (f < nseq) {
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
    }
    __syncthreads();
}

// pass 1 on ten values per thread (sequence sfo, residue sto): ten-point DFT over n1, twiddle W_1200^{sto k1}, store at (k1, sto)
__device__ __forceinline__ void fft_pass_1(float2 *buf, const float2 *__restrict__ tw, float2 (&v)[10], int sfo, int sto) {
    butterfly<10>(v);
    float2 *dst = buf + sfo * FSIG + sto + 2 * (sto / 12);
    dst[0] = v[0];
    const float2 *t1 = tw + FTW1 + sto;
#pragma unroll
    for (int k = 1; k < 10; ++k) dst[k * (10 * FROW)] = cmul(v[k], t1[(k - 1) * 120]);
}

__global__ __launch_bounds__(256, 3) void feat_gcc_kernel(const float *__restrict__ audio, const long *__restrict__ clip_offset,
                                                          const float *__restrict__ twiddle,
                                                          const float *__restrict__ sc_mean,
                                                          const float *__restrict__ sc_rstd, float *__restrict__ out,
                                                          int n_samples, int T, int pix_stride, int ch0) {
    const float2 *__restrict__ tw = reinterpret_cast<const float2 *>(twiddle);
    __shared__ __attribute__((aligned(16))) float2 buf[2 * FSIG];          // the transform buffer (two sequences)
    __shared__ __attribute__((aligned(16))) float2 G[3][FN];               // conj(H) of the three pair-packs, natural order
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * GR;
    const float2 *aud = reinterpret_cast<const float2 *>(audio) + 2 * (clip_offset ? (size_t)clip_offset[b] : (size_t)b * n_samples);
    for (int fr = 0; fr < GR; ++fr) {
        const int t = t0 + fr;
        if (t >= T) break;
        int ti = tid;
        asm volatile("" : "+v"(ti));                  // (index arithmetic redone per frame: see features.hip)
        const int sfo = ti >= 120 ? 1 : 0, sto = ti - 120 * sfo;
        // ---- round 0: STFT of the frame, as K1 (signal 0 = mic 0 + i mic 1, signal 1 = mic 2 + i mic 3)
        if (ti < 240) {
            const float wc = tw[sto].x, ws = -tw[sto].y;
            float2 v[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                int s = t * FHOP - FHOP + sto + 120 * i;
                if (s < 0) s = -s;                   // np.pad(..., mode='reflect') at the start of the clip
                const float2 a = aud[2 * (size_t)s + sfo];
                const float w = 0.5f - 0.5f * (wc * RC10[i] + ws * RS10[i]);
                v[i] = make_float2(a.x * w, a.y * w);
            }
            fft_pass_1(buf, tw, v, sfo, sto);
        }
        fft_passes_2_3(buf, tw, ti, 2);
        // ---- the four spectra -> six unit-phase cross spectra -> conj(H) of the three pair-packs
        for (int k = ti; k < FBINS; k += 256) {
            const int pk = fpos(k), pn = fpos(k == 0 ? 0 : FN - k);
            const float2 z1 = buf[pk], z2 = buf[FSIG + pk];
            const float2 n1 = cconj(buf[pn]), n2 = cconj(buf[FSIG + pn]);
            float2 X[4];
            X[0] = cscale(cadd(z1, n1), 0.5f);
            X[1] = cscale(cmi(csub(z1, n1)), 0.5f);
            X[2] = cscale(cadd(z2, n2), 0.5f);
            X[3] = cscale(cmi(csub(z2, n2)), 0.5f);
            float2 P[6];
            int p = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = m + 1; n < 4; ++n) {
                    float2 R = cmul(cconj(X[m]), X[n]);
                    if (k == 0 || k == FN / 2) R.y = 0.f;       // real bins of real signals (irfft ignores their imaginary part)
                    P[p++] = unit_phase(R);
                }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float2 pa = P[2 * q], pb = P[2 * q + 1];
                // H[k] = pa + i pb;  H[N - k] = conj(pa) + i conj(pb);  stored: conj(H)
                G[q][k] = make_float2(pa.x - pb.y, -(pa.y + pb.x));
                if (k > 0 && k < FN / 2) G[q][FN - k] = make_float2(pa.x + pb.y, pa.y - pb.x);
            }
        }
        __syncthreads();
        // ---- two more rounds of the forward passes: packs (0, 1) as the two sequences, then pack 2
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            const int nseq = round == 0 ? 2 : 1;
            if (ti < 240 && sfo < nseq) {
                const float2 *src = G[2 * round + sfo];
                float2 v[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) v[i] = src[sto + 120 * i];
                fft_pass_1(buf, tw, v, sfo, sto);
            }
            fft_passes_2_3(buf, tw, ti, nseq);
            // irfft(pa)[n] = Re(Y[n]) / N, irfft(pb)[n] = -Im(Y[n]) / N with Y = FFT(conj(H)); lag bin m: n = m < 32 ? N - 32 + m : m - 32
            if (ti < nseq * NLAG) {
                const int s = ti >> 6, m = ti & 63;
                const int n = m < NLAG / 2 ? FN - NLAG / 2 + m : m - NLAG / 2;
                const float2 y = buf[s * FSIG + fpos(n)];
                const int c0 = 2 * (2 * round + s);                   // channel of pair-pack member a
                float *o = out + (((size_t)b * T + t) * NLAG + m) * pix_stride + ch0;
                o[c0] = (y.x * (1.0f / FN) - sc_mean[c0 * NLAG + m]) * sc_rstd[c0 * NLAG + m];
                o[c0 + 1] = (-y.y * (1.0f / FN) - sc_mean[(c0 + 1) * NLAG + m]) * sc_rstd[(c0 + 1) * NLAG + m];
            }
            __syncthreads();                        // the next round / frame overwrites the transform buffer
        }
    }
}

}  // namespace adyolo

using namespace adyolo;

extern "C" int adyolo_feat_gcc_phat(const float *audio, const int64_t *clip_offset, const float *twiddle,
                                    const float *scaler_mean, const float *scaler_rstd, float *out, int B, int n_samples,
                                    int pix_stride, int ch0, void *stream) {
    ADYOLO_REQUIRE(audio && twiddle && scaler_mean && scaler_rstd && out, ADYOLO_EINVAL, "feat_gcc_phat: null pointer");
    ADYOLO_REQUIRE(B > 0 && n_samples >= 1200 && n_samples % FHOP == 0 && pix_stride >= 6 && ch0 >= 0 && ch0 + 6 <= pix_stride,
                   ADYOLO_EINVAL, "feat_gcc_phat: n_samples=%d must be a multiple of 600 and >= 1200, 6 channels must fit the pixel",
                   n_samples);
    const int T = n_samples / FHOP;
    hipLaunchKernelGGL(feat_gcc_kernel, dim3(cdiv(T, GR), B), dim3(256), 0, as_stream(stream), audio,
                       reinterpret_cast<const long *>(clip_offset), twiddle, scaler_mean, scaler_rstd, out, n_samples, T,
                       pix_stride, ch0);
    return check_launch("feat_gcc_phat");
}
