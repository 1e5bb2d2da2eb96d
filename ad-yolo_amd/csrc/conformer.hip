// Kernels of the ResNet-Conformer encoder (BASELINE config 4; /root/reference/src/models/backbones/resnet_conformer.py):
//   * filter packing for the general strided convolution (adyolo_conv_gemm in gemm.hip: 7x7 s(1,2) stem :347, torchvision
//     BasicBlock 3x3 s(1,2) / 1x1 s(1,2) convolutions :353-393; the im2col / col2im buffers of round 1 are gone)
//   * MaxPool2d(3, s(1,2), p1) :350, BN->ReLU fusions of BasicBlock
//   * LayerNorm :160,211,236,262, Swish :142-150, GLU :167, depthwise dilated Conv1d :169, row softmax of the
//     attention scores :74, AvgPool1d :288-295, a*x + b*z residual mixing :98
// All tensors channels-last float32.  HBM-bound elementwise passes, float4 where the channel count allows.
#include "common.hpp"

namespace adyolo {

static inline int ew_grid_c(long n) {
    long g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// w [Cout][Cin][KH][KW] <-> wk [Cout][Kp] with k = (kh*KW+kw)*Cin + ci
__global__ void pack_wk_kernel(const float *__restrict__ w, float *__restrict__ wk, int Cout, int Cin, int KH, int KW,
                               int Kp, int to_packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Kp) return;
    const int k = idx % Kp, co = idx / Kp;
    if (k >= KH * KW * Cin) {
        if (to_packed) wk[idx] = 0.f;
        return;
    }
    const int ci = k % Cin, kk = k / Cin;
    const size_t o = ((size_t)co * Cin + ci) * KH * KW + kk;
    if (to_packed) wk[idx] = w[o];
    else const_cast<float *>(w)[o] = wk[idx];
}

// ---------------------------------------------------------------------------------------------- max-pool 3x3 s(1,2) p1
__global__ __launch_bounds__(256) void maxpool3_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                           unsigned char *__restrict__ arg, int H, int W, int C, int Wo,
                                                           long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long p = i / C;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int h = (int)(p % H);
        const long n = p / H;
        float best = -INFINITY;
        int bi = 0;
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = h - 1 + kh;
            if (hh < 0 || hh >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int ww = wo * 2 - 1 + kw;
                if (ww < 0 || ww >= W) continue;
                const float v = x[(((size_t)n * H + hh) * W + ww) * C + c];
                if (v > best) {
                    best = v;
                    bi = kh * 3 + kw;
                }
            }
        }
        y[i] = best;
        arg[i] = (unsigned char)bi;
    }
}
// Gather form (round 4): every input element sums the (at most six) windows it can be the maximum of, in a fixed order -- the
// scatter form with atomicAdd made the Conformer step differ from run to run (and from its hipGraph replay) in the last bits.
__global__ __launch_bounds__(256) void maxpool3_bwd_kernel(const float *__restrict__ dy,
                                                           const unsigned char *__restrict__ arg, float *__restrict__ dx,
                                                           int H, int W, int C, int Wo, long total_in) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_in; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long p = i / C;
        const int ww = (int)(p % W);
        p /= W;
        const int hh = (int)(p % H);
        const long n = p / H;
        float s = 0.f;
        for (int kh = 0; kh < 3; ++kh) {
            const int h = hh + 1 - kh;                     // window row whose tap kh is this input row
            if (h < 0 || h >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int t = ww + 1 - kw;                 // = 2 wo
                if (t < 0 || (t & 1)) continue;
                const int wo = t >> 1;
                if (wo >= Wo) continue;
                const size_t o = (((size_t)n * H + h) * Wo + wo) * C + c;
                if (arg[o] == kh * 3 + kw) s += dy[o];
            }
        }
        dx[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------- small elementwise ops
__global__ __launch_bounds__(256) void affine_relu_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                          const float *__restrict__ shift, float *__restrict__ y,
                                                          long n4, int c4n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int cx = (int)(i % c4n);
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        const float4 s = reinterpret_cast<const float4 *>(scale)[cx];
        const float4 t = reinterpret_cast<const float4 *>(shift)[cx];
        reinterpret_cast<float4 *>(y)[i] = make_float4(fmaxf(v.x * s.x + t.x, 0.f), fmaxf(v.y * s.y + t.y, 0.f),
                                                       fmaxf(v.z * s.z + t.z, 0.f), fmaxf(v.w * s.w + t.w, 0.f));
    }
}
// dx = dy * (y > 0)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ y,
                                                       float4 *__restrict__ dx, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 g = dy[i], v = y[i];
        dx[i] = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
    }
}
__global__ __launch_bounds__(256) void axpby_kernel(const float4 *__restrict__ x, const float4 *__restrict__ z,
                                                    float4 *__restrict__ y, float a, float b, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 u = x[i], v = z[i];
        y[i] = make_float4(a * u.x + b * v.x, a * u.y + b * v.y, a * u.z + b * v.z, a * u.w + b * v.w);
    }
}
__global__ __launch_bounds__(256) void swish_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = v * sigmoidf_(v);
    }
}
__global__ __launch_bounds__(256) void swish_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                        float *__restrict__ dx, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i], s = sigmoidf_(v);
        dx[i] = dy[i] * (s + v * s * (1.f - s));
    }
}
// GLU over the channel axis of a channels-last [R][2C] tensor: y = x[:, :C] * sigmoid(x[:, C:])
__global__ __launch_bounds__(256) void glu_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, long R, int C) {
    const long total = R * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        y[i] = x[r * 2 * C + c] * sigmoidf_(x[r * 2 * C + C + c]);
    }
}
__global__ __launch_bounds__(256) void glu_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                      float *__restrict__ dx, long R, int C) {
    const long total = R * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const float a = x[r * 2 * C + c], g = sigmoidf_(x[r * 2 * C + C + c]), d = dy[i];
        dx[r * 2 * C + c] = d * g;
        dx[r * 2 * C + C + c] = d * a * g * (1.f - g);
    }
}

// ---------------------------------------------------------------------------------------------- depthwise conv1d k=3
// x [B][T][C], w [C][3], bias [C]: y[b][t][c] = bias[c] + sum_j w[c][j] x[b][t+(j-1)d][c]
__global__ __launch_bounds__(256) void dwconv3_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, float *__restrict__ y, int T,
                                                          int C, int d, long total, int flip) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long bt = i / C;
        const int t = (int)(bt % T);
        float s = bias ? bias[c] : 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int tt = t + (j - 1) * d;
            if (tt >= 0 && tt < T) s += w[c * 3 + (flip ? 2 - j : j)] * x[i + (long)(tt - t) * C];
        }
        y[i] = s;
    }
}
// weight / bias gradient partials: partial[blk][4*C] = (dw0, dw1, dw2, db) per channel over a strip of rows
__global__ __launch_bounds__(256) void dwconv3_wgrad_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                            float *__restrict__ partial, int T, int C, int d, long rows,
                                                            long rows_per_block) {
    // threads: C <= 256 channels per pass; loop rows
    const long rbeg = (long)blockIdx.x * rows_per_block;
    const long rend = rbeg + rows_per_block < rows ? rbeg + rows_per_block : rows;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, ab = 0.f;
        for (long r = rbeg; r < rend; ++r) {
            const int t = (int)(r % T);
            const float g = dy[r * C + c];
            ab += g;
            a1 += g * x[r * C + c];
            if (t - d >= 0) a0 += g * x[(r - d) * C + c];
            if (t + d < T) a2 += g * x[(r + d) * C + c];
        }
        float *p = partial + (size_t)blockIdx.x * 4 * C;
        p[c * 3 + 0] = a0; p[c * 3 + 1] = a1; p[c * 3 + 2] = a2; p[3 * C + c] = ab;
    }
}

// ---------------------------------------------------------------------------------------------- row softmax
// P[r][:] = softmax(scale * S[r][:]) in place or out of place; L <= 4096; one wave per row
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float *__restrict__ s, float *__restrict__ p, long R,
                                                          int L, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float *src = s + (size_t)row * L;
    float *dst = p + (size_t)row * L;
    float mx = -INFINITY;
    for (int i = lane; i < L; i += 64) mx = fmaxf(mx, src[i] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < L; i += 64) {
        const float e = expf(src[i] * scale - mx);
        dst[i] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int i = lane; i < L; i += 64) dst[i] *= inv;
}
// dS = scale * P * (dP - sum(P * dP))
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float *__restrict__ dp, const float *__restrict__ p,
                                                          float *__restrict__ ds, long R, int L, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float *g = dp + (size_t)row * L, *pr = p + (size_t)row * L;
    float *o = ds + (size_t)row * L;
    float dot = 0.f;
    for (int i = lane; i < L; i += 64) dot += g[i] * pr[i];
    dot = wave_sum(dot);
    for (int i = lane; i < L; i += 64) o[i] = scale * pr[i] * (g[i] - dot);
}

// ---------------------------------------------------------------------------------------------- AvgPool1d(k) over T
// y[b][t'][c] = fac * mean_{j<k} x[b][t'*k+j][c]   (fac = 2: the reference adds two average pools, :288-294)
__global__ __launch_bounds__(256) void avgpool1d_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int T,
                                                            int C, int k, float fac, long total) {
    const int To = T / k;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long q = i / C;
        const int to = (int)(q % To);
        const long b = q / To;
        float s = 0.f;
        for (int j = 0; j < k; ++j) s += x[((size_t)b * T + (size_t)to * k + j) * C + c];
        y[i] = s * fac / (float)k;
    }
}
__global__ __launch_bounds__(256) void avgpool1d_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int T,
                                                            int C, int k, float fac, long total) {
    const int To = T / k;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long q = i / C;
        const int t = (int)(q % T);
        const long b = q / T;
        const int to = t / k;
        dx[i] = to < To ? dy[((size_t)b * To + to) * C + c] * fac / (float)k : 0.f;
    }
}

}  // namespace adyolo

using namespace adyolo;

extern "C" int adyolo_pack_wk(float *w, float *wk, int Cout, int Cin, int KH, int KW, int to_packed, void *stream) {
    ADYOLO_REQUIRE(w && wk && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, ADYOLO_EINVAL, "pack_wk: bad arguments");
    const int Kp = (KH * KW * Cin + 3) / 4 * 4;
    hipLaunchKernelGGL(pack_wk_kernel, dim3(cdiv((long)Cout * Kp, 256)), dim3(256), 0, as_stream(stream), w, wk, Cout, Cin,
                       KH, KW, Kp, to_packed);
    return check_launch("pack_wk");
}

extern "C" int adyolo_maxpool3_fwd(const float *x, float *y, unsigned char *arg, int N, int H, int W, int C,
                                   void *stream) {
    ADYOLO_REQUIRE(x && y && arg && N > 0 && H > 0 && W > 0 && C > 0, ADYOLO_EINVAL, "maxpool3_fwd: bad arguments");
    const int Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * H * Wo * C;
    hipLaunchKernelGGL(maxpool3_fwd_kernel, dim3(ew_grid_c(total)), dim3(256), 0, as_stream(stream), x, y, arg, H, W, C, Wo,
                       total);
    return check_launch("maxpool3_fwd");
}
extern "C" int adyolo_maxpool3_bwd(const float *dy, const unsigned char *arg, float *dx_zeroed, int N, int H, int W,
                                   int C, void *stream) {
    ADYOLO_REQUIRE(dy && arg && dx_zeroed && N > 0 && H > 0 && W > 0 && C > 0, ADYOLO_EINVAL, "maxpool3_bwd: bad arguments");
    const int Wo = (W + 2 - 3) / 2 + 1;
    const long total_in = (long)N * H * W * C;
    hipLaunchKernelGGL(maxpool3_bwd_kernel, dim3(ew_grid_c(total_in)), dim3(256), 0, as_stream(stream), dy, arg, dx_zeroed, H,
                       W, C, Wo, total_in);
    return check_launch("maxpool3_bwd");
}

extern "C" int adyolo_affine_relu_nhwc(const float *x, const float *scale, const float *shift, float *y, long rows, int C,
                                       void *stream) {
    ADYOLO_REQUIRE(x && scale && shift && y && rows > 0 && C > 0 && C % 4 == 0, ADYOLO_EINVAL, "affine_relu: bad arguments");
    const long n4 = rows * (C / 4);
    hipLaunchKernelGGL(affine_relu_kernel, dim3(ew_grid_c(n4)), dim3(256), 0, as_stream(stream), x, scale, shift, y, n4, C / 4);
    return check_launch("affine_relu");
}
extern "C" int adyolo_relu_bwd(const float *dy, const float *y, float *dx, long n, void *stream) {
    ADYOLO_REQUIRE(dy && y && dx && n > 0 && n % 4 == 0, ADYOLO_EINVAL, "relu_bwd: n must be a positive multiple of 4");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid_c(n / 4)), dim3(256), 0, as_stream(stream), (const float4 *)dy,
                       (const float4 *)y, (float4 *)dx, n / 4);
    return check_launch("relu_bwd");
}
extern "C" int adyolo_axpby(const float *x, const float *z, float *y, float a, float b, long n, void *stream) {
    ADYOLO_REQUIRE(x && z && y && n > 0 && n % 4 == 0, ADYOLO_EINVAL, "axpby: n must be a positive multiple of 4");
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid_c(n / 4)), dim3(256), 0, as_stream(stream), (const float4 *)x,
                       (const float4 *)z, (float4 *)y, a, b, n / 4);
    return check_launch("axpby");
}
extern "C" int adyolo_swish_fwd(const float *x, float *y, long n, void *stream) {
    ADYOLO_REQUIRE(x && y && n > 0, ADYOLO_EINVAL, "swish_fwd: bad arguments");
    hipLaunchKernelGGL(swish_fwd_kernel, dim3(ew_grid_c(n)), dim3(256), 0, as_stream(stream), x, y, n);
    return check_launch("swish_fwd");
}
extern "C" int adyolo_swish_bwd(const float *dy, const float *x, float *dx, long n, void *stream) {
    ADYOLO_REQUIRE(dy && x && dx && n > 0, ADYOLO_EINVAL, "swish_bwd: bad arguments");
    hipLaunchKernelGGL(swish_bwd_kernel, dim3(ew_grid_c(n)), dim3(256), 0, as_stream(stream), dy, x, dx, n);
    return check_launch("swish_bwd");
}
extern "C" int adyolo_glu_fwd(const float *x, float *y, long rows, int C, void *stream) {
    ADYOLO_REQUIRE(x && y && rows > 0 && C > 0, ADYOLO_EINVAL, "glu_fwd: bad arguments");
    hipLaunchKernelGGL(glu_fwd_kernel, dim3(ew_grid_c(rows * C)), dim3(256), 0, as_stream(stream), x, y, rows, C);
    return check_launch("glu_fwd");
}
extern "C" int adyolo_glu_bwd(const float *dy, const float *x, float *dx, long rows, int C, void *stream) {
    ADYOLO_REQUIRE(dy && x && dx && rows > 0 && C > 0, ADYOLO_EINVAL, "glu_bwd: bad arguments");
    hipLaunchKernelGGL(glu_bwd_kernel, dim3(ew_grid_c(rows * C)), dim3(256), 0, as_stream(stream), dy, x, dx, rows, C);
    return check_launch("glu_bwd");
}

extern "C" int adyolo_dwconv3_fwd(const float *x, const float *w, const float *bias, float *y, int B, int T, int C,
                                  int dilation, int flip, void *stream) {
    ADYOLO_REQUIRE(x && w && y && B > 0 && T > 0 && C > 0 && dilation > 0, ADYOLO_EINVAL, "dwconv3_fwd: bad arguments");
    const long total = (long)B * T * C;
    hipLaunchKernelGGL(dwconv3_fwd_kernel, dim3(ew_grid_c(total)), dim3(256), 0, as_stream(stream), x, w, bias, y, T, C,
                       dilation, total, flip);
    return check_launch("dwconv3_fwd");
}
// partial: [1024][4*C]; dw [C][3], db [C] (both overwritten)
extern "C" int adyolo_dwconv3_wgrad(const float *dy, const float *x, float *dw, float *db, float *partial,
                                    float *colsum_ws, int B, int T, int C, int dilation, void *stream) {
    ADYOLO_REQUIRE(dy && x && dw && db && partial && colsum_ws && B > 0 && T > 0 && C > 0 && C <= 1024, ADYOLO_EINVAL,
                   "dwconv3_wgrad: bad arguments");
    hipStream_t st = as_stream(stream);
    const long rows = (long)B * T;
    int nblk = (int)(rows / 64 > 1024 ? 1024 : (rows / 64 < 1 ? 1 : rows / 64));
    const long rpb = (rows + nblk - 1) / nblk;
    nblk = (int)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(dwconv3_wgrad_kernel, dim3(nblk), dim3(256), 0, st, dy, x, partial, T, C, dilation, rows, rpb);
    int rc = check_launch("dwconv3_wgrad");
    if (rc) return rc;
    rc = adyolo_colsum(partial, dw, colsum_ws, nblk, 3 * C, 4 * C, 0, stream);
    if (rc) return rc;
    return adyolo_colsum(partial + 3 * C, db, colsum_ws, nblk, C, 4 * C, 0, stream);
}

extern "C" int adyolo_softmax_fwd(const float *s, float *p, long rows, int L, float scale, void *stream) {
    ADYOLO_REQUIRE(s && p && rows > 0 && L > 0, ADYOLO_EINVAL, "softmax_fwd: bad arguments");
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), s, p, rows, L, scale);
    return check_launch("softmax_fwd");
}
extern "C" int adyolo_softmax_bwd(const float *dp, const float *p, float *ds, long rows, int L, float scale,
                                  void *stream) {
    ADYOLO_REQUIRE(dp && p && ds && rows > 0 && L > 0, ADYOLO_EINVAL, "softmax_bwd: bad arguments");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), dp, p, ds, rows, L, scale);
    return check_launch("softmax_bwd");
}

extern "C" int adyolo_avgpool1d_fwd(const float *x, float *y, int B, int T, int C, int k, float fac, void *stream) {
    ADYOLO_REQUIRE(x && y && B > 0 && T >= k && C > 0 && k > 0, ADYOLO_EINVAL, "avgpool1d_fwd: bad arguments");
    const long total = (long)B * (T / k) * C;
    hipLaunchKernelGGL(avgpool1d_fwd_kernel, dim3(ew_grid_c(total)), dim3(256), 0, as_stream(stream), x, y, T, C, k, fac, total);
    return check_launch("avgpool1d_fwd");
}
extern "C" int adyolo_avgpool1d_bwd(const float *dy, float *dx, int B, int T, int C, int k, float fac, void *stream) {
    ADYOLO_REQUIRE(dy && dx && B > 0 && T >= k && C > 0 && k > 0, ADYOLO_EINVAL, "avgpool1d_bwd: bad arguments");
    const long total = (long)B * T * C;
    hipLaunchKernelGGL(avgpool1d_bwd_kernel, dim3(ew_grid_c(total)), dim3(256), 0, as_stream(stream), dy, dx, T, C, k, fac, total);
    return check_launch("avgpool1d_bwd");
}
