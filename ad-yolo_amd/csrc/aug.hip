// Input-pipeline kernels around K1: PCM16 -> float conversion of staged WAV data, SpecAug
// masking on the channels-last feature tensor, and the train-set scaler statistics (mean / std / max / min per
// mel bin and channel).  References: /root/reference/src/datasets.py:101-162 (audio / 32768 + 1e-8, spec-augment call),
// /root/reference/src/utils/augmentations.py:6-33 (SpecAug), /root/reference/src/preprocess.py:86-130 (scaler).
#include <float.h>
#include "common.hpp"

namespace adyolo {

// int16 [n] -> float [n]:  x / 32768 + 1e-8   (datasets.py:105, preprocess.py:104); 8 samples per thread
__global__ __launch_bounds__(256) void pcm16_to_f32_kernel(const int16_t *__restrict__ pcm, float *__restrict__ out, long n8,
                                                           long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long b = i * 8;
        if (b + 8 <= n) {
            const int4 v = *reinterpret_cast<const int4 *>(pcm + b);          // 8 x int16
            const int w[4] = {v.x, v.y, v.z, v.w};
            float f[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f[2 * k] = (float)(short)(w[k] & 0xffff) / 32768.0f + 1e-8f;
                f[2 * k + 1] = (float)(short)(w[k] >> 16) / 32768.0f + 1e-8f;
            }
            *reinterpret_cast<float4 *>(out + b) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4 *>(out + b + 4) = make_float4(f[4], f[5], f[6], f[7]);
        } else {
            for (long j = b; j < n; ++j) out[j] = (float)pcm[j] / 32768.0f + 1e-8f;
        }
    }
}

// SpecAug on feat [B][T][F][C]: per sample b zero frames [t0,t1) and mel bins [f0,f1) (ranges in `rng` [B][4], an
// empty range = no mask); mask value 0 like torchaudio's default
__global__ __launch_bounds__(256) void mask_ranges_kernel(float *__restrict__ feat, const int *__restrict__ rng, int T,
                                                          int F, int C4) {
    const int b = blockIdx.y;
    const int t0 = rng[b * 4 + 0], t1 = rng[b * 4 + 1], f0 = rng[b * 4 + 2], f1 = rng[b * 4 + 3];
    if (t0 >= t1 && f0 >= f1) return;
    float4 *base = reinterpret_cast<float4 *>(feat) + (size_t)b * T * F * C4;
    const long total = (long)T * F * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)((i / C4) % F), t = (int)(i / ((long)C4 * F));
        if ((t >= t0 && t < t1) || (f >= f0 && f < f1)) base[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// per-column sum / sum of squares / max / min of a [rows][cols] matrix, stage 1: one partial row per workgroup
// part [4][nblk][cols]
__global__ __launch_bounds__(256) void colstats_partial_kernel(const float *__restrict__ a, float *__restrict__ part,
                                                               long rows, int cols) {
    const int nblk = gridDim.x;
    const long per = (rows + nblk - 1) / nblk;
    const long r0 = (long)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        float s = 0.f, q = 0.f, mx = -FLT_MAX, mn = FLT_MAX;
        for (long r = r0; r < r1; ++r) {
            const float v = a[r * cols + c];
            s += v;
            q += v * v;
            mx = fmaxf(mx, v);
            mn = fminf(mn, v);
        }
        part[((size_t)0 * nblk + blockIdx.x) * cols + c] = s;
        part[((size_t)1 * nblk + blockIdx.x) * cols + c] = q;
        part[((size_t)2 * nblk + blockIdx.x) * cols + c] = mx;
        part[((size_t)3 * nblk + blockIdx.x) * cols + c] = mn;
    }
}
// stage 2: out [4][cols] doubles (sum, sumsq, max, min)
__global__ __launch_bounds__(256) void colstats_final_kernel(const float *__restrict__ part, double *__restrict__ out,
                                                             int nblk, int cols) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double s = 0.0, q = 0.0, mx = -DBL_MAX, mn = DBL_MAX;
    for (int b = 0; b < nblk; ++b) {
        s += (double)part[((size_t)0 * nblk + b) * cols + c];
        q += (double)part[((size_t)1 * nblk + b) * cols + c];
        mx = fmax(mx, (double)part[((size_t)2 * nblk + b) * cols + c]);
        mn = fmin(mn, (double)part[((size_t)3 * nblk + b) * cols + c]);
    }
    out[0 * cols + c] = s;
    out[1 * cols + c] = q;
    out[2 * cols + c] = mx;
    out[3 * cols + c] = mn;
}

}  // namespace adyolo

using namespace adyolo;

extern "C" int adyolo_pcm16_to_f32(const int16_t *pcm, float *out, long n, void *stream) {
    ADYOLO_REQUIRE(pcm && out && n > 0, ADYOLO_EINVAL, "pcm16_to_f32: bad arguments");
    ADYOLO_REQUIRE(((uintptr_t)pcm & 15) == 0 && ((uintptr_t)out & 15) == 0, ADYOLO_EINVAL, "pcm16_to_f32: 16-byte aligned buffers");
    const long n8 = (n + 7) / 8;
    long g = (n8 + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(pcm16_to_f32_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), pcm, out, n8, n);
    return check_launch("pcm16_to_f32");
}

extern "C" int adyolo_mask_ranges(float *feat, const int *ranges, int B, int T, int F, int C, void *stream) {
    ADYOLO_REQUIRE(feat && ranges && B > 0 && T > 0 && F > 0 && C > 0 && C % 4 == 0, ADYOLO_EINVAL, "mask_ranges: bad arguments");
    long g = ((long)T * F * (C / 4) + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(mask_ranges_kernel, dim3((unsigned)g, B), dim3(256), 0, as_stream(stream), feat, ranges, T, F, C / 4);
    return check_launch("mask_ranges");
}

extern "C" int adyolo_colstats(const float *a, float *partial, double *out, long rows, int cols, void *stream) {
    ADYOLO_REQUIRE(a && partial && out && rows > 0 && cols > 0, ADYOLO_EINVAL, "colstats: bad arguments");
    int nblk = (int)(rows < 1024 ? rows : 1024);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(colstats_partial_kernel, dim3(nblk), dim3(256), 0, st, a, partial, rows, cols);
    int rc = check_launch("colstats_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(colstats_final_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, st, partial, out, nblk, cols);
    return check_launch("colstats_final");
}
