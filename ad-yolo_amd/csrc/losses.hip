// K10 and the remaining plugin losses/heads of the reference's --loss switch (src/main.py:43):
//   * head activations  (sigmoid | tanh on column ranges)      linearheads.py:44-47,65,83
//   * ACCDOA   MSE                                               loss.py:57-67
//   * SEDDOA   BCE + 1000 * (masked) MSE                         loss.py:32-54
//   * ADPIT    13-permutation min-MSE (multi-ACCDOA)             loss.py:70-153
// Each loss is ONE pass over the network output that also writes d(loss)/d(output) (HBM-bound: read output +
// target, write gradient), per-workgroup partial sums combined in double by a one-workgroup finishing kernel.
#include "common.hpp"

namespace adyolo {

constexpr int LOSSES_BLOCKS = 1024;

// y[r][c] = c < nsig ? sigmoid(x) : tanh(x)
__global__ __launch_bounds__(256) void act_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, long n,
                                                      int cols, int nsig) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols);
        const float v = x[i];
        y[i] = c < nsig ? sigmoidf_(v) : tanhf(v);
    }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                      float *__restrict__ dx, long n, int cols, int nsig) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols);
        const float v = y[i];
        dx[i] = dy[i] * (c < nsig ? v * (1.f - v) : 1.f - v * v);
    }
}

__device__ __forceinline__ void block_partial(float v, float *partial, int slot_stride, int slot) {
    __shared__ float red[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * slot_stride + slot] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
}

// SEDDOA / ACCDOA: columns [0, nsed) -> BCE (nn.BCELoss semantics), columns [nsed, cols) -> squared error,
// optionally with the output multiplied by the target activity of its class (masked MSE, loss.py:47-48).
__global__ __launch_bounds__(256) void seddoa_loss_kernel(const float *__restrict__ out, const float *__restrict__ tgt,
                                                          float *__restrict__ dout, float *__restrict__ partial,
                                                          long n, int cols, int nsed, int masked, float w_bce,
                                                          float w_mse) {
    float s_bce = 0.f, s_mse = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols);
        const float o = out[i], t = tgt[i];
        float g;
        if (c < nsed) {
            const float lp = fmaxf(logf(o), -100.f), lq = fmaxf(logf(1.f - o), -100.f);
            s_bce += -(t * lp + (1.f - t) * lq);
            g = w_bce * (o - t) / fmaxf(o * (1.f - o), 1e-12f);
        } else {
            float m = 1.f;
            if (masked) m = tgt[i - c + (c - nsed) % nsed];        // activity of this column's class
            const float d = o * m - t;
            s_mse += d * d;
            g = w_mse * 2.f * d * m;
        }
        if (dout) dout[i] = g;
    }
    block_partial(s_bce, partial, 2, 0);
    block_partial(s_mse, partial, 2, 1);
}
__global__ __launch_bounds__(256) void seddoa_final_kernel(const float *__restrict__ partial, int nblk, double n_bce,
                                                           double n_mse, float c_bce, float c_mse,
                                                           float *__restrict__ loss) {
    __shared__ double red[256];
    __shared__ double tot[2];
    const double s = block_colsum32(partial, nblk, 2, 0, 2, red);
    if ((threadIdx.x >> 5) == 0 && (threadIdx.x & 31) < 2) tot[threadIdx.x & 31] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        loss[0] = (float)((n_bce > 0 ? (double)c_bce * tot[0] / n_bce : 0.0) + (double)c_mse * tot[1] / n_mse);
}

// ADPIT: out [R][9][C] (track*3+axis major, class minor), tgt [R][6][4][C] (dummy, act|x|y|z, class).
// One lane per (row, class): the 13 candidate targets, their mean-squared errors over the 9 entries, arg-min (first
// minimum, like torch.min), loss and gradient.
__global__ __launch_bounds__(256) void adpit_loss_kernel(const float *__restrict__ out, const float *__restrict__ tgt,
                                                         float *__restrict__ dout, float *__restrict__ partial, long R,
                                                         int C, float gscale) {
    float acc = 0.f;
    const long total = R * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const float *t = tgt + (size_t)r * 24 * C + c;
        float v[6][3];                                   // act * xyz of A0, B0, B1, C0, C1, C2
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            const float act = t[(size_t)(d * 4) * C];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) v[d][ax] = act * t[(size_t)(d * 4 + 1 + ax) * C];
        }
        float o[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = out[(size_t)r * 9 * C + (size_t)k * C + c];
        // permutations: index into v for the three tracks
        const int perm[13][3] = {{0, 0, 0}, {1, 1, 2}, {1, 2, 1}, {1, 2, 2}, {2, 1, 1}, {2, 1, 2}, {2, 2, 1},
                                 {3, 4, 5}, {3, 5, 4}, {4, 3, 5}, {4, 5, 3}, {5, 3, 4}, {5, 4, 3}};
        // paddings (loss.py:106-108): pad4A = B0B0B1 + C0C1C2, pad4B = A0A0A0 + C0C1C2, pad4C = A0A0A0 + B0B0B1
        float best = 0.f;
        int besti = 0;
        float bt[9];
#pragma unroll
        for (int p = 0; p < 13; ++p) {
            float tg[9];
            float l = 0.f;
#pragma unroll
            for (int tr = 0; tr < 3; ++tr)
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const float a0 = v[0][ax];
                    const float bbb = v[perm[1][tr]][ax];
                    const float ccc = v[perm[7][tr]][ax];
                    float pad;
                    if (p == 0) pad = bbb + ccc;
                    else if (p < 7) pad = a0 + ccc;
                    else pad = a0 + bbb;
                    const float tv = v[perm[p][tr]][ax] + pad;
                    tg[tr * 3 + ax] = tv;
                    const float d = o[tr * 3 + ax] - tv;
                    l += d * d;
                }
            l *= (1.0f / 9.0f);
            if (p == 0 || l < best) {
                best = l;
                besti = p;
#pragma unroll
                for (int k = 0; k < 9; ++k) bt[k] = tg[k];
            }
        }
        (void)besti;
        acc += best;
        if (dout) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                dout[(size_t)r * 9 * C + (size_t)k * C + c] = gscale * (2.0f / 9.0f) * (o[k] - bt[k]);
        }
    }
    block_partial(acc, partial, 1, 0);
}
__global__ __launch_bounds__(256) void sum_final_kernel(const float *__restrict__ partial, int nblk, double denom,
                                                        float *__restrict__ loss) {
    __shared__ double red[256];
    const double s = block_colsum32(partial, nblk, 1, 0, 1, red);
    if (threadIdx.x == 0) loss[0] = (float)(s / denom);
}

static inline int loss_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g > LOSSES_BLOCKS ? LOSSES_BLOCKS : (g < 1 ? 1 : g));
}

}  // namespace adyolo

using namespace adyolo;

extern "C" int adyolo_act_fwd(const float *x, float *y, long rows, int cols, int n_sigmoid_cols, void *stream) {
    ADYOLO_REQUIRE(x && y && rows > 0 && cols > 0, ADYOLO_EINVAL, "act_fwd: bad arguments");
    const long n = rows * cols;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(loss_grid(n) * 4), dim3(256), 0, as_stream(stream), x, y, n, cols, n_sigmoid_cols);
    return check_launch("act_fwd");
}
extern "C" int adyolo_act_bwd(const float *dy, const float *y, float *dx, long rows, int cols, int n_sigmoid_cols,
                              void *stream) {
    ADYOLO_REQUIRE(dy && y && dx && rows > 0 && cols > 0, ADYOLO_EINVAL, "act_bwd: bad arguments");
    const long n = rows * cols;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(loss_grid(n) * 4), dim3(256), 0, as_stream(stream), dy, y, dx, n, cols,
                       n_sigmoid_cols);
    return check_launch("act_bwd");
}

// partial: workspace of 2*1024 floats.  loss = w_bce * mean BCE(first nsed columns) + w_mse * mean SE(other columns)
extern "C" int adyolo_seddoa_loss(const float *out, const float *tgt, float *loss, float *dout, float *partial,
                                  long rows, int cols, int nsed, int masked, float w_bce, float w_mse, void *stream) {
    ADYOLO_REQUIRE(out && tgt && loss && partial && rows > 0 && cols > 0 && nsed >= 0 && nsed < cols, ADYOLO_EINVAL,
                   "seddoa_loss: bad arguments");
    ADYOLO_REQUIRE(!masked || (nsed > 0 && (cols - nsed) % nsed == 0), ADYOLO_EINVAL, "seddoa_loss: masked needs cols = nsed*(1+k)");
    hipStream_t st = as_stream(stream);
    const long n = rows * cols;
    const int g = loss_grid(n);
    const double n_bce = (double)rows * nsed, n_mse = (double)rows * (cols - nsed);
    hipLaunchKernelGGL(seddoa_loss_kernel, dim3(g), dim3(256), 0, st, out, tgt, dout, partial, n, cols, nsed, masked,
                       n_bce > 0 ? (float)(w_bce / n_bce) : 0.f, (float)(w_mse / n_mse));
    int rc = check_launch("seddoa_loss");
    if (rc) return rc;
    hipLaunchKernelGGL(seddoa_final_kernel, dim3(1), dim3(256), 0, st, partial, g, n_bce, n_mse, w_bce, w_mse, loss);
    return check_launch("seddoa_final");
}

// out [rows][9][C], tgt [rows][6][4][C]; partial: 1024 floats
extern "C" int adyolo_adpit_loss(const float *out, const float *tgt, float *loss, float *dout, float *partial,
                                 long rows, int C, void *stream) {
    ADYOLO_REQUIRE(out && tgt && loss && partial && rows > 0 && C > 0, ADYOLO_EINVAL, "adpit_loss: bad arguments");
    hipStream_t st = as_stream(stream);
    const long total = rows * C;
    const int g = loss_grid(total);
    hipLaunchKernelGGL(adpit_loss_kernel, dim3(g), dim3(256), 0, st, out, tgt, dout, partial, rows, C,
                       (float)(1.0 / (double)total));
    int rc = check_launch("adpit_loss");
    if (rc) return rc;
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, st, partial, g, (double)total, loss);
    return check_launch("adpit_final");
}
