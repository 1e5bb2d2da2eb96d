// K11: fused Adam over one flat parameter buffer (all 6.68 M parameters live in one allocation, so one
// launch updates the whole model and one RCCL all-reduce covers all gradients).  Arithmetic follows
// torch.optim.Adam's single-tensor path (used at /root/reference/src/train.py:31,55; amsgrad off):
//   m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;  p -= lr/(1 - b1^t) * m / (sqrt(v)/sqrt(1 - b2^t) + eps)
// Also: the error string holder of the library and the NCHW -> NHWC8 entry transpose.
#include <stdarg.h>
#include <math.h>
#include "common.hpp"

namespace adyolo {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// DEV: step_size / inv_sqrt_bc2 come from device memory (`bc`, written by adam_prep_kernel from a device-side step
// counter), so that a launch recorded in a hipGraph does the right bias correction at every replay
template <bool DEV>
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, long n, float beta1,
                                                   float beta2, float eps, float wd, float step_size,
                                                   float inv_sqrt_bc2, float grad_scale,
                                                   const float *__restrict__ bc) {
    if (DEV) {
        step_size = bc[0];
        inv_sqrt_bc2 = bc[1];
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * grad_scale;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        float mi = m[i], vi = v[i];
        mi += (gi - mi) * (1.f - beta1);
        vi = vi * beta2 + (1.f - beta2) * gi * gi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = pi - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

// step counter += 1 on the device; bc = {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} in double like the host entry point
__global__ void adam_prep_kernel(unsigned long long *__restrict__ step, float *__restrict__ bc, float lr, float beta1,
                                 float beta2) {
    const unsigned long long s = *step + 1ull;
    *step = s;
    const double bc1 = 1.0 - pow((double)beta1, (double)s);
    const double bc2 = 1.0 - pow((double)beta2, (double)s);
    bc[0] = (float)((double)lr / bc1);
    bc[1] = (float)(1.0 / sqrt(bc2));
}

__global__ __launch_bounds__(256) void fill32_kernel(uint32_t *__restrict__ p, uint32_t v, size_t n) {
    const size_t n4 = n >> 2;
    uint4 *p4 = reinterpret_cast<uint4 *>(p);
    const uint4 v4 = make_uint4(v, v, v, v);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p4[i] = v4;
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[n4 * 4 + threadIdx.x] = v;
}

int fill32(void *ptr, uint32_t value, size_t n32, hipStream_t st) {
    if (n32 == 0) return 0;
    if (reinterpret_cast<uintptr_t>(ptr) & 15) {          // (never the case for the workspaces of this library)
        set_error("fill32: pointer not 16-byte aligned");
        return ADYOLO_EINVAL;
    }
    size_t g = ((n32 >> 2) + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(fill32_kernel, dim3((unsigned)g), dim3(256), 0, st, reinterpret_cast<uint32_t *>(ptr), value, n32);
    return check_launch("fill32");
}

__global__ void counter_add_kernel(unsigned long long *__restrict__ c, unsigned long long inc) { *c += inc; }

__global__ __launch_bounds__(256) void nchw_to_nhwc8_kernel(const float *__restrict__ x, float *__restrict__ y, int C,
                                                            long HW, long total) {
    // one thread per output pixel: gathers C (<= 8) planes, writes 32 contiguous bytes
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / HW, p = i - n * HW;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = c < C ? x[((size_t)n * C + c) * HW + p] : 0.f;
        float4 *o = reinterpret_cast<float4 *>(y + (size_t)i * 8);
        o[0] = make_float4(v[0], v[1], v[2], v[3]);
        o[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

}  // namespace adyolo

using namespace adyolo;

extern "C" int adyolo_abi_version(void) { return ADYOLO_ABI_VERSION; }
extern "C" const char *adyolo_last_error(void) { return g_err; }

extern "C" int adyolo_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                                void *stream) {
    ADYOLO_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, ADYOLO_EINVAL, "adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const long g = (n + 255) / 256;
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, as_stream(stream), param,
                       grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, weight_decay, step_size, inv_sqrt_bc2, grad_scale,
                       (const float *)nullptr);
    return check_launch("adam_step");
}

extern "C" int adyolo_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, uint64_t *step_dev,
                                    float *bc_dev, float grad_scale, void *stream) {
    ADYOLO_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step_dev && bc_dev, ADYOLO_EINVAL,
                   "adam_step_dev: bad arguments");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned long long *>(step_dev), bc_dev,
                       lr, beta1, beta2);
    int rc = check_launch("adam_prep");
    if (rc) return rc;
    const long g = (n + 255) / 256;
    hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, st, param, grad, exp_avg,
                       exp_avg_sq, n, beta1, beta2, eps, weight_decay, 0.f, 0.f, grad_scale, (const float *)bc_dev);
    return check_launch("adam_step_dev");
}

extern "C" int adyolo_counter_add(uint64_t *counter, uint64_t inc, void *stream) {
    ADYOLO_REQUIRE(counter, ADYOLO_EINVAL, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, as_stream(stream),
                       reinterpret_cast<unsigned long long *>(counter), (unsigned long long)inc);
    return check_launch("counter_add");
}

extern "C" int adyolo_nchw_to_nhwc8(const float *x, float *y, int B, int C, int H, int W, void *stream) {
    ADYOLO_REQUIRE(x && y && B > 0 && C > 0 && C <= 8 && H > 0 && W > 0, ADYOLO_EINVAL, "nchw_to_nhwc8: bad arguments");
    const long HW = (long)H * W, total = (long)B * HW;
    const long g = (total + 255) / 256;
    hipLaunchKernelGGL(nchw_to_nhwc8_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, as_stream(stream), x,
                       y, C, HW, total);
    return check_launch("nchw_to_nhwc8");
}

// ---- FOA rotation augmentation on raw audio (reference src/utils/augmentations.py:81-96): per clip a sign for each of
// the Y, Z, X channels and an optional X <-> Y swap;  audio [B][n][4] (W, Y, Z, X), cfg[b] = {sy, sz, sx, swap}
namespace adyolo {
__global__ __launch_bounds__(256) void foa_rotate_kernel(const float4 *__restrict__ x, float4 *__restrict__ y,
                                                         const float *__restrict__ cfg, long n_per_clip) {
    const int b = blockIdx.y;
    const float sy = cfg[b * 4 + 0], sz = cfg[b * 4 + 1], sx = cfg[b * 4 + 2];
    const bool swap = cfg[b * 4 + 3] != 0.f;
    const float4 *src = x + (size_t)b * n_per_clip;
    float4 *dst = y + (size_t)b * n_per_clip;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_per_clip; i += (long)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        const float yy = v.y * sy, zz = v.z * sz, xx = v.w * sx;
        dst[i] = swap ? make_float4(v.x, xx, zz, yy) : make_float4(v.x, yy, zz, xx);
    }
}
}  // namespace adyolo

extern "C" int adyolo_foa_rotate(const float *audio, float *out, const float *cfg, int B, long n_samples, void *stream) {
    ADYOLO_REQUIRE(audio && out && cfg && B > 0 && n_samples > 0, ADYOLO_EINVAL, "foa_rotate: bad arguments");
    long g = (n_samples + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(adyolo::foa_rotate_kernel, dim3((unsigned)g, B), dim3(256), 0, adyolo::as_stream(stream),
                       (const float4 *)audio, (float4 *)out, cfg, n_samples);
    return adyolo::check_launch("foa_rotate");
}
