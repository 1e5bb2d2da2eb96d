// K9w: 1-D Winograd F(4, 3) along the TIME axis for the 3 x 1 convolutions of the ResNet-Conformer's deep stages (round 5).
// torchvision BasicBlock's stride-1 3x3 convolutions (reference src/models/backbones/resnet_conformer.py:353-393) meet maps whose
// frequency axis has been strided down to 1 bin (512 channels) or 2 bins (256 channels, folded into 512: models/backbones/
// resnet_conformer.py::_conv3x3_s1) -- there the convolution IS a 3 x 1 one along the 800 frames, a 2-D Winograd patch would be
// 75-94 % padding, and the implicit-GEMM form (adyolo_conv_gemm) does the full 3 multiplies per output at 0.67 of the fp32 MFMA peak.
//     y = A^T [ (G g) (.) (B^T d) ]        d: 6 input rows (4 t - 1 .. 4 t + 4), g: the 3 taps, y: 4 output rows
// with the matrices of csrc/wino4.hip (interpolation points 0, +-3/4, +-3/2, infinity): 6 multiplies per 4 outputs = half the
// matrix FLOPs.  The six positions are six GEMMs [N T][Cin] x [Cin][Cout] -- one adyolo_gemm_batched launch -- between two
// elementwise passes over tensors that stay in the memory-side cache (52-79 MB at 32 x 800 x 512):
//     forward / data gradient:  V = B^T d (wino1d_in),  M[p] = V[p] U[p]^T,  y = A^T M (wino1d_out)
//     weight gradient:          E = A e (wino1d_dy),    dU[p] = E[p]^T V[p],  dw = G^T dU (wino1d_filter, mode 2)
// Per launch at 32 x 800 x 512 -> 512: forward 0.385 -> ~0.30 ms, weight gradient 0.47 -> ~0.34 ms (profiles/r05_wino1d_ab.txt).
#include "wino4_common.hpp"

namespace adyolo {
namespace w1 {

using w4::A2;
using w4::A3;
using w4::B2;
using w4::B3;
using w4::P2;
using w4::PA;
using w4::PB;
using w4::S2;

// one component of B^T d
__device__ __forceinline__ void bt6(const float (&c)[6], float (&t)[6]) { w4::bt6s(c, t); }

// x [N][H][C] -> V [6][N T][C], T = H / 4; rows outside [0, H) are zeros (the convolution's padding)
__global__ __launch_bounds__(256) void wino1d_in_kernel(const float *__restrict__ x, float *__restrict__ V, int N, int H, int C4,
                                                         long total) {
    const int T = H >> 2;
    const size_t plane = (size_t)N * T * C4;              // float4 per position
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long rt = i / C4;
        const int t = (int)(rt % T);
        const long n = rt / T;
        const float4 *src = reinterpret_cast<const float4 *>(x) + ((size_t)n * H) * C4 + c4;
        float4 d[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int h = 4 * t - 1 + r;
            d[r] = (h >= 0 && h < H) ? src[(size_t)h * C4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 o[6];
#define ADYOLO_W1_COMP(F)                                                     \
    {                                                                         \
        const float c_[6] = {d[0].F, d[1].F, d[2].F, d[3].F, d[4].F, d[5].F}; \
        float t_[6];                                                          \
        bt6(c_, t_);                                                          \
        o[0].F = t_[0]; o[1].F = t_[1]; o[2].F = t_[2]; o[3].F = t_[3]; o[4].F = t_[4]; o[5].F = t_[5]; \
    }
        ADYOLO_W1_COMP(x) ADYOLO_W1_COMP(y) ADYOLO_W1_COMP(z) ADYOLO_W1_COMP(w)
#undef ADYOLO_W1_COMP
#pragma unroll
        for (int p = 0; p < 6; ++p) reinterpret_cast<float4 *>(V)[p * plane + (size_t)rt * C4 + c4] = o[p];
    }
}

// M [6][N T][C] -> y [N][H][C]:  y = A^T m
__global__ __launch_bounds__(256) void wino1d_out_kernel(const float *__restrict__ M, float *__restrict__ y, int N, int H, int C4,
                                                          long total) {
    const int T = H >> 2;
    const size_t plane = (size_t)N * T * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long rt = i / C4;
        const int t = (int)(rt % T);
        const long n = rt / T;
        float4 m[6];
#pragma unroll
        for (int p = 0; p < 6; ++p) m[p] = reinterpret_cast<const float4 *>(M)[p * plane + (size_t)rt * C4 + c4];
        float4 o[4];
        w4::at4(m[0].x, m[1].x, m[2].x, m[3].x, m[4].x, m[5].x, o[0].x, o[1].x, o[2].x, o[3].x);
        w4::at4(m[0].y, m[1].y, m[2].y, m[3].y, m[4].y, m[5].y, o[0].y, o[1].y, o[2].y, o[3].y);
        w4::at4(m[0].z, m[1].z, m[2].z, m[3].z, m[4].z, m[5].z, o[0].z, o[1].z, o[2].z, o[3].z);
        w4::at4(m[0].w, m[1].w, m[2].w, m[3].w, m[4].w, m[5].w, o[0].w, o[1].w, o[2].w, o[3].w);
        float4 *dst = reinterpret_cast<float4 *>(y) + ((size_t)n * H + 4 * t) * C4 + c4;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(size_t)r * C4] = o[r];
    }
}

// dy [N][H][C] -> E [6][N T][C]:  E = A e  (A = (A^T)^T: 4 output rows -> 6 positions)
__global__ __launch_bounds__(256) void wino1d_dy_kernel(const float *__restrict__ dy, float *__restrict__ E, int N, int H, int C4,
                                                         long total) {
    const int T = H >> 2;
    const size_t plane = (size_t)N * T * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long rt = i / C4;
        const int t = (int)(rt % T);
        const long n = rt / T;
        const float4 *src = reinterpret_cast<const float4 *>(dy) + ((size_t)n * H + 4 * t) * C4 + c4;
        const float4 v0 = src[0], v1 = src[(size_t)C4], v2 = src[(size_t)2 * C4], v3 = src[(size_t)3 * C4];
        float4 o[6];
#define ADYOLO_W1_COMP(F)                                                              \
    {                                                                                  \
        const float ea = fmaf(A2, v2.F, v0.F), oa = fmaf(A3, v3.F, PA * v1.F);          \
        const float eb = fmaf(B2, v2.F, v0.F), ob = fmaf(B3, v3.F, PB * v1.F);          \
        o[0].F = v0.F; o[1].F = ea + oa; o[2].F = ea - oa; o[3].F = eb + ob; o[4].F = eb - ob; o[5].F = v3.F; \
    }
        ADYOLO_W1_COMP(x) ADYOLO_W1_COMP(y) ADYOLO_W1_COMP(z) ADYOLO_W1_COMP(w)
#undef ADYOLO_W1_COMP
#pragma unroll
        for (int p = 0; p < 6; ++p) reinterpret_cast<float4 *>(E)[p * plane + (size_t)rt * C4 + c4] = o[p];
    }
}

// G (6 x 3), in double as in csrc/wino4w.hip's finishing kernel
__device__ __forceinline__ void g_matrix(double (&G)[6][3]) {
    const double a_ = 0.75, b_ = 1.5, a2 = a_ * a_, b2 = b_ * b_;
    const double na = 2.0 * a2 * (a2 - b2), nb = 2.0 * b2 * (b2 - a2);
    const double g[6][3] = {{1.0 / (a2 * b2), 0.0, 0.0}, {1.0 / na, a_ / na, a2 / na}, {1.0 / na, -a_ / na, a2 / na},
                            {1.0 / nb, b_ / nb, b2 / nb}, {1.0 / nb, -b_ / nb, b2 / nb}, {0.0, 0.0, 1.0}};
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int k = 0; k < 3; ++k) G[p][k] = g[p][k];
}

// mode 0: U[p][co][ci] = sum_k G[p][k] w[co][ci][k]                 (forward filter, k-major B operand of M[p] = V[p] U[p]^T)
// mode 1: U[p][ci][co] = sum_k G[p][k] w[co][ci][2 - k]             (data gradient: flipped taps, channel roles swapped)
// mode 2: w[co][ci][k] = sum_p G[p][k] U[p][co][ci]                 (weight gradient from dU; `w` is written)
__global__ __launch_bounds__(256) void wino1d_filter_kernel(float *__restrict__ w, float *__restrict__ U, int Cout, int Cin, int mode) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int pairs = Cout * Cin;
    if (idx >= pairs) return;
    const int ci = idx % Cin, co = idx / Cin;
    double G[6][3];
    g_matrix(G);
    if (mode == 2) {
        double d[6];
#pragma unroll
        for (int p = 0; p < 6; ++p) d[p] = (double)U[(size_t)p * pairs + idx];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double s = 0.0;
#pragma unroll
            for (int p = 0; p < 6; ++p) s += G[p][k] * d[p];
            w[(size_t)idx * 3 + k] = (float)s;
        }
        return;
    }
    const double g0 = (double)w[(size_t)idx * 3 + (mode ? 2 : 0)], g1 = (double)w[(size_t)idx * 3 + 1],
                 g2 = (double)w[(size_t)idx * 3 + (mode ? 0 : 2)];
    const size_t o = mode ? (size_t)ci * Cout + co : (size_t)idx;
#pragma unroll
    for (int p = 0; p < 6; ++p) U[(size_t)p * pairs + o] = (float)(G[p][0] * g0 + G[p][1] * g1 + G[p][2] * g2);
}

static inline int w1_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace w1
}  // namespace adyolo

using namespace adyolo;

static int w1_check(const void *a, const void *b, int N, int H, int C, const char *what) {
    ADYOLO_REQUIRE(a && b && N > 0 && H > 0 && C > 0, ADYOLO_EINVAL, "%s: bad arguments", what);
    ADYOLO_REQUIRE(H % 4 == 0 && C % 4 == 0, ADYOLO_ENOSUP, "%s: H=%d must be a multiple of 4 (whole output tiles) and C=%d of 4", what, H, C);
    return 0;
}

extern "C" int adyolo_wino1d_in(const float *x, float *V, int N, int H, int C, void *stream) {
    int rc = w1_check(x, V, N, H, C, "wino1d_in");
    if (rc) return rc;
    const long total = (long)N * (H / 4) * (C / 4);
    hipLaunchKernelGGL(w1::wino1d_in_kernel, dim3(w1::w1_grid(total)), dim3(256), 0, as_stream(stream), x, V, N, H, C / 4, total);
    return check_launch("wino1d_in");
}

extern "C" int adyolo_wino1d_out(const float *M, float *y, int N, int H, int C, void *stream) {
    int rc = w1_check(M, y, N, H, C, "wino1d_out");
    if (rc) return rc;
    const long total = (long)N * (H / 4) * (C / 4);
    hipLaunchKernelGGL(w1::wino1d_out_kernel, dim3(w1::w1_grid(total)), dim3(256), 0, as_stream(stream), M, y, N, H, C / 4, total);
    return check_launch("wino1d_out");
}

extern "C" int adyolo_wino1d_dy(const float *dy, float *E, int N, int H, int C, void *stream) {
    int rc = w1_check(dy, E, N, H, C, "wino1d_dy");
    if (rc) return rc;
    const long total = (long)N * (H / 4) * (C / 4);
    hipLaunchKernelGGL(w1::wino1d_dy_kernel, dim3(w1::w1_grid(total)), dim3(256), 0, as_stream(stream), dy, E, N, H, C / 4, total);
    return check_launch("wino1d_dy");
}

extern "C" int adyolo_wino1d_filter(float *w, float *U, int Cout, int Cin, int mode, void *stream) {
    ADYOLO_REQUIRE(w && U && Cout > 0 && Cin > 0 && mode >= 0 && mode <= 2, ADYOLO_EINVAL, "wino1d_filter: bad arguments");
    hipLaunchKernelGGL(w1::wino1d_filter_kernel, dim3(cdiv(Cout * Cin, 256)), dim3(256), 0, as_stream(stream), w, U, Cout, Cin, mode);
    return check_launch("wino1d_filter");
}
