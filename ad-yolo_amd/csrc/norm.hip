// K3 / K3b / K4: BatchNorm2d (train + eval), the squeeze-excite + residual tail of SEBasicBlock, and
// AvgPool2d(2,2) -- all HBM-bound, channels-last, float4 per lane.  Replaces nn.BatchNorm2d
// (/root/reference/src/models/backbones/resnet.py:17,19,144,163), SELayer (:91-106), the residual
// add + ReLU (:45-46) and nn.AvgPool2d (:13,27-29).
// Reductions are two-stage and deterministic: per-workgroup fp32 partials (a thread sums <= a few hundred
// values), combined in double by a single small finishing kernel.  No atomics.
#include "common.hpp"

namespace adyolo {

#ifndef ADYOLO_EW_NT
#define ADYOLO_EW_NT 3      // (bit 0: loads, bit 1: stores) the streamed tensors of the elementwise / reduction passes (0.3-1.3 GB each, read or written once per pass:
                            // no cache level holds them until their next use) are read and written with the non-temporal hint:
                            // the three big passes 0.67-0.72 -> 0.73-0.78 of 8 TB/s, -2.1 ms per step (profiles/r05_nt_ab.txt)
#endif
__device__ __forceinline__ float4 ew_ld(const float4 *p) {
#if ADYOLO_EW_NT & 1
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
#else
    return *p;
#endif
}
__device__ __forceinline__ void ew_st(float4 *p, const float4 &v) {
#if ADYOLO_EW_NT & 2
    const f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(p));
#else
    *p = v;
#endif
}


// ---- generic two-value column reduction over rows of a [N][HW][C] tensor ---------------------------------
// grid (G, N); partial layout [2][N*G][C]
template <class F>
__global__ __launch_bounds__(256) void reduce2_partial_kernel(F f, float *__restrict__ partial, int HW, int C,
                                                              int G) {
    __shared__ float red[2][256 * 4];
    const int c4n = C >> 2;
    const int ry_n = 256 / c4n;
    const int tid = threadIdx.x;
    const int cx = tid % c4n, ry = tid / c4n;
    const int g = blockIdx.x, n = blockIdx.y;
    const int chunk = (HW + G - 1) / G;
    const int rbeg = g * chunk, rend = min(HW, rbeg + chunk);
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
    if (ry < ry_n) {
        for (int r = rbeg + ry; r < rend; r += ry_n) {
            float4 a, b;
            f(n, r, cx, a, b);
            u.x += a.x; u.y += a.y; u.z += a.z; u.w += a.w;
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        float *p0 = &red[0][(ry * c4n + cx) * 4];
        float *p1 = &red[1][(ry * c4n + cx) * 4];
        p0[0] = u.x; p0[1] = u.y; p0[2] = u.z; p0[3] = u.w;
        p1[0] = v.x; p1[1] = v.y; p1[2] = v.z; p1[3] = v.w;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s0 = 0.f, s1 = 0.f;
        for (int y = 0; y < ry_n; ++y) {
            s0 += red[0][y * C + c];
            s1 += red[1][y * C + c];
        }
        const size_t slot = (size_t)n * G + g;
        const size_t half = (size_t)gridDim.y * G * C;
        partial[slot * C + c] = s0;
        partial[half + slot * C + c] = s1;
    }
}

struct StatsF {          // (x, x^2)
    const float *x; int HW, C;
    __device__ void operator()(int n, int r, int cx, float4 &a, float4 &b) const {
        a = ew_ld(reinterpret_cast<const float4 *>(x + ((size_t)n * HW + r) * C + cx * 4));
        b = make_float4(a.x * a.x, a.y * a.y, a.z * a.z, a.w * a.w);
    }
};
struct BnBwdF {          // (dy, dy * xhat)
    const float *dy, *x, *mean, *invstd; int HW, C;
    __device__ void operator()(int n, int r, int cx, float4 &a, float4 &b) const {
        const size_t o = ((size_t)n * HW + r) * C + cx * 4;
        a = ew_ld(reinterpret_cast<const float4 *>(dy + o));
        const float4 xv = ew_ld(reinterpret_cast<const float4 *>(x + o));
        const float4 m = *reinterpret_cast<const float4 *>(mean + cx * 4);
        const float4 is = *reinterpret_cast<const float4 *>(invstd + cx * 4);
        b = make_float4(a.x * (xv.x - m.x) * is.x, a.y * (xv.y - m.y) * is.y, a.z * (xv.z - m.z) * is.z,
                        a.w * (xv.w - m.w) * is.w);
    }
};
struct SeBwdF {          // g = de * (e > 0): (g, g * xhat(c)); the mask comes from `mask` bits when given, else from e
    // Wp > 0: `de` is the gradient of avgpool2(e), [N][H/2][Wp/2][C] with Wp the width of e: de = 0.25 * that, spread 2 x 2
    const float *de, *e, *c, *mean, *invstd; const unsigned long long *mask; int HW, C, Wp;
    __device__ void operator()(int n, int r, int cx, float4 &a, float4 &b) const {
        const size_t o = ((size_t)n * HW + r) * C + cx * 4;
        float4 d;
        if (Wp > 0) {
            const int y = r / Wp, x = r - y * Wp;
            const float4 g4 = *reinterpret_cast<const float4 *>(de + (((size_t)n * (HW / Wp >> 1) + (y >> 1)) * (Wp >> 1) + (x >> 1)) * C + cx * 4);
            d = make_float4(0.25f * g4.x, 0.25f * g4.y, 0.25f * g4.z, 0.25f * g4.w);
        } else {
            d = ew_ld(reinterpret_cast<const float4 *>(de + o));
        }
        const float4 cv = ew_ld(reinterpret_cast<const float4 *>(c + o));
        const float4 m = *reinterpret_cast<const float4 *>(mean + cx * 4);
        const float4 is = *reinterpret_cast<const float4 *>(invstd + cx * 4);
        bool px, py, pz, pw;
        if (mask) {
            mask_bits4(mask, o >> 2, px, py, pz, pw);
        } else {
            const float4 ev = ew_ld(reinterpret_cast<const float4 *>(e + o));
            px = ev.x > 0.f; py = ev.y > 0.f; pz = ev.z > 0.f; pw = ev.w > 0.f;
        }
        a = make_float4(px ? d.x : 0.f, py ? d.y : 0.f, pz ? d.z : 0.f, pw ? d.w : 0.f);
        b = make_float4(a.x * (cv.x - m.x) * is.x, a.y * (cv.y - m.y) * is.y, a.z * (cv.z - m.z) * is.z,
                        a.w * (cv.w - m.w) * is.w);
    }
};

// stage 2: per-sample sums of the two reduced quantities (32 channels x 8 part-groups per workgroup, double)
// grid (ceil(C/32), N)
__global__ __launch_bounds__(256) void persample_reduce_kernel(const float *__restrict__ partial,
                                                               float *__restrict__ out0, float *__restrict__ out1,
                                                               int N, int G, int C) {
    __shared__ double red[512];
    const int n = blockIdx.y, c0 = blockIdx.x * 32;
    const size_t half = (size_t)N * G * C;
    const float *base = partial + (size_t)n * G * C;
    double s0, s1;                              // (both sums in one pass: ~110 of these launches per step, each two dependent reductions before; round 6)
    block_colsum32x2(base, base + half, G, (size_t)C, c0, C, red, s0, s1);
    const int c = c0 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) == 0 && c < C) {
        out0[(size_t)n * C + c] = (float)s0;
        out1[(size_t)n * C + c] = (float)s1;
    }
}

// stage 3 (BatchNorm forward): batch mean / invstd from the per-sample sums, running statistics update
__global__ void bn_stats_final_kernel(const float *__restrict__ ps0, const float *__restrict__ ps1,
                                      float *__restrict__ mean, float *__restrict__ invstd,
                                      float *__restrict__ rmean, float *__restrict__ rvar, int N, int C, double R,
                                      float momentum, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double t0 = 0.0, t1 = 0.0;
    for (int n = 0; n < N; ++n) {
        t0 += (double)ps0[(size_t)n * C + c];
        t1 += (double)ps1[(size_t)n * C + c];
    }
    const double m = t0 / R;
    double var = t1 / R - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    if (rvar) {
        const double unbiased = R > 1.0 ? var * R / (R - 1.0) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
}

// stage 3, parallel form: 32 channels x 8 sample-groups per workgroup (double), then mean / invstd, the running
// statistics update and -- when gamma / beta are given -- the affine (scale, shift) the consumers apply, in ONE launch
// grid ceil(C/32)
__global__ __launch_bounds__(256) void bn_finish_kernel(const float *__restrict__ ps0, const float *__restrict__ ps1,
                                                        float *__restrict__ mean, float *__restrict__ invstd,
                                                        float *__restrict__ rmean, float *__restrict__ rvar,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        float *__restrict__ scale, float *__restrict__ shift, int N, int C,
                                                        double R, float momentum, float eps) {
    __shared__ double red[512];
    const int c0 = blockIdx.x * 32;
    double t0, t1;
    block_colsum32x2(ps0, ps1, N, (size_t)C, c0, C, red, t0, t1);
    const int c = c0 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) != 0 || c >= C) return;
    const double m = t0 / R;
    double var = t1 / R - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = mf;
    invstd[c] = is;
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mf;
    if (rvar) {
        const double unbiased = R > 1.0 ? var * R / (R - 1.0) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
    if (scale) {
        const float sc = gamma[c] * is;
        scale[c] = sc;
        shift[c] = beta[c] - mf * sc;
    }
}

__global__ void bn_eval_stats_kernel(const float *rm, const float *rv, float *mean, float *invstd, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        mean[c] = rm[c];
        invstd[c] = 1.0f / sqrtf(rv[c] + eps);
    }
}
__global__ void bn_scale_shift_kernel(const float *gamma, const float *beta, const float *mean, const float *invstd,
                                      float *scale, float *shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float s = gamma[c] * invstd[c];
        scale[c] = s;
        shift[c] = beta[c] - mean[c] * s;
    }
}

__global__ __launch_bounds__(256) void affine_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                     const float *__restrict__ shift, float *__restrict__ y,
                                                     long n4, int c4n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int cx = (int)(i % c4n);
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        const float4 s = reinterpret_cast<const float4 *>(scale)[cx];
        const float4 t = reinterpret_cast<const float4 *>(shift)[cx];
        reinterpret_cast<float4 *>(y)[i] =
            make_float4(v.x * s.x + t.x, v.y * s.y + t.y, v.z * s.z + t.z, v.w * s.w + t.w);
    }
}

// Streaming kernels of this file (bn_bwd_apply, se_tail_fwd, se_tail_bwd_apply) share one loop shape: a thread owns the
// float4 index tid + 256 j, so when C/4 divides 256 (INV) its channel quad never changes -- the per-channel operands are
// loaded once into registers (no 64-bit modulo, no L1 traffic per element) and EW_U independent float4 loads per tensor
// are in flight before the first use.
#ifndef ADYOLO_EW_U
#define ADYOLO_EW_U 4
#endif
constexpr int EW_U = ADYOLO_EW_U;


template <bool INV>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ gamma,
    const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ sdy,
    const float *__restrict__ sdyx, float *__restrict__ dx, float *__restrict__ sum_partial, long n4, int c4n, float invR,
    int relu_mask) {
    // sum_partial (optional, [gridDim.x][C]): per-workgroup channel sums of dx -- e.g. the bias gradient of the convolution
    // that feeds this BatchNorm (the stem's: a separate column sum would re-read the 1.26 GB tensor).  A thread's channel
    // quad is fixed (the grid stride is a multiple of C/4), so it accumulates in registers.
    __shared__ float4 sred[256];
    float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 g, m, is, a, b;
    auto load_params = [&](int cx) {
        g = reinterpret_cast<const float4 *>(gamma)[cx];
        m = reinterpret_cast<const float4 *>(mean)[cx];
        is = reinterpret_cast<const float4 *>(invstd)[cx];
        a = reinterpret_cast<const float4 *>(sdy)[cx];
        b = reinterpret_cast<const float4 *>(sdyx)[cx];
    };
    auto one = [&](float4 d, float4 xv) {
        float4 o;
        o.x = g.x * is.x * (d.x - a.x * invR - (xv.x - m.x) * is.x * b.x * invR);
        o.y = g.y * is.y * (d.y - a.y * invR - (xv.y - m.y) * is.y * b.y * invR);
        o.z = g.z * is.z * (d.z - a.z * invR - (xv.z - m.z) * is.z * b.z * invR);
        o.w = g.w * is.w * (d.w - a.w * invR - (xv.w - m.w) * is.w * b.w * invR);
        if (relu_mask) {
            if (!(xv.x > 0.f)) o.x = 0.f;
            if (!(xv.y > 0.f)) o.y = 0.f;
            if (!(xv.z > 0.f)) o.z = 0.f;
            if (!(xv.w > 0.f)) o.w = 0.f;
        }
        sacc.x += o.x; sacc.y += o.y; sacc.z += o.z; sacc.w += o.w;
        return o;
    };
    const float4 *dy4 = reinterpret_cast<const float4 *>(dy), *x4 = reinterpret_cast<const float4 *>(x);
    float4 *dx4 = reinterpret_cast<float4 *>(dx);
    if (INV) {
        load_params((int)threadIdx.x % c4n);
        const long step = (long)gridDim.x * 256 * EW_U;
        for (long i0 = (long)blockIdx.x * 256 * EW_U + threadIdx.x; i0 < n4; i0 += step) {
            float4 d[EW_U], xv[EW_U];
#pragma unroll
            for (int u = 0; u < EW_U; ++u) {
                const long i = i0 + u * 256;
                if (i < n4) {
                    d[u] = ew_ld(&dy4[i]);
                    xv[u] = ew_ld(&x4[i]);
                }
            }
#pragma unroll
            for (int u = 0; u < EW_U; ++u) {
                const long i = i0 + u * 256;
                if (i < n4) ew_st(&dx4[i], one(d[u], xv[u]));
            }
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            load_params((int)(i % c4n));
            dx4[i] = one(dy4[i], x4[i]);
        }
    }
    if (sum_partial) {          // fixed order: threads with the same channel quad are tid, tid + c4n, ...
        sred[threadIdx.x] = sacc;
        __syncthreads();
        if ((int)threadIdx.x < c4n) {
            float4 t = sred[threadIdx.x];
            for (int k = threadIdx.x + c4n; k < 256; k += c4n) {
                const float4 u = sred[k];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            reinterpret_cast<float4 *>(sum_partial + (size_t)blockIdx.x * c4n * 4)[threadIdx.x] = t;
        }
    }
}
__global__ __launch_bounds__(256) void partial_colsum_kernel(const float *__restrict__ partial, float *__restrict__ out,
                                                             int nblk, int C) {
    __shared__ double red[256];
    const double sm = block_colsum32(partial, nblk, (size_t)C, blockIdx.x * 32, C, red);
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) == 0 && c < C) out[c] = (float)sm;
}
__global__ void accum2_kernel(const float *a, const float *b, float *da, float *db, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        if (da) da[c] += a[c];
        if (db) db[c] += b[c];
    }
}

// ---- SE -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se_fc_fwd_kernel(
    const float *__restrict__ ssum, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w2,
    const float *__restrict__ b2, float *__restrict__ pooled, float *__restrict__ hid, float *__restrict__ s,
    int HW, int C, int Cr) {
    __shared__ float pl[1024];
    __shared__ float hd[128];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float invHW = 1.0f / (float)HW;
    for (int c = tid; c < C; c += 256) {
        const float p = scale[c] * (ssum[(size_t)n * C + c] * invHW) + shift[c];
        pl[c] = p;
        pooled[(size_t)n * C + c] = p;
    }
    __syncthreads();
    for (int j = wave; j < Cr; j += 4) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a += w1[(size_t)j * C + c] * pl[c];
        a = wave_sum(a);
        if (lane == 0) {
            const float h = fmaxf(a + b1[j], 0.f);
            hd[j] = h;
            hid[(size_t)n * Cr + j] = h;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float a = b2[c];
        for (int j = 0; j < Cr; ++j) a += w2[(size_t)c * Cr + j] * hd[j];
        s[(size_t)n * C + c] = sigmoidf_(a);
    }
}

template <bool INV>
__global__ __launch_bounds__(256) void se_tail_fwd_kernel(
    const float *__restrict__ c, const float *__restrict__ r, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ s, const float *__restrict__ r_scale,
    const float *__restrict__ r_shift, float *__restrict__ e, unsigned long long *__restrict__ mask, long hw4, int c4n) {
    // grid (blocks, N): hw4 = HW * C/4 float4 per sample; mask (optional, needs hw4 % 64 == 0): bits of (e > 0)
    const int n = blockIdx.y;
    const size_t base = (size_t)n * hw4;
    const float4 *c4 = reinterpret_cast<const float4 *>(c) + base, *r4 = reinterpret_cast<const float4 *>(r) + base;
    float4 *e4 = reinterpret_cast<float4 *>(e) + base;
    float4 sc, sh, sv, rs = make_float4(1.f, 1.f, 1.f, 1.f), rt = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_params = [&](int cx) {
        sc = reinterpret_cast<const float4 *>(scale)[cx];
        sh = reinterpret_cast<const float4 *>(shift)[cx];
        sv = reinterpret_cast<const float4 *>(s)[(size_t)n * c4n + cx];
        if (r_scale) {           // the shortcut's own BatchNorm affine (downsample branch) applied on the fly
            rs = reinterpret_cast<const float4 *>(r_scale)[cx];
            rt = reinterpret_cast<const float4 *>(r_shift)[cx];
        }
    };
    auto one = [&](long i, float4 cv, float4 rv) {
        if (r_scale) rv = make_float4(fmaf(rv.x, rs.x, rt.x), fmaf(rv.y, rs.y, rt.y), fmaf(rv.z, rs.z, rt.z), fmaf(rv.w, rs.w, rt.w));
        float4 o;
        o.x = fmaxf((cv.x * sc.x + sh.x) * sv.x + rv.x, 0.f);
        o.y = fmaxf((cv.y * sc.y + sh.y) * sv.y + rv.y, 0.f);
        o.z = fmaxf((cv.z * sc.z + sh.z) * sv.z + rv.z, 0.f);
        o.w = fmaxf((cv.w * sc.w + sh.w) * sv.w + rv.w, 0.f);
        ew_st(&e4[i], o);
        if (mask) {      // the whole wave is here: i runs over 64-aligned groups of 64 and hw4 % 64 == 0
            const unsigned long long bx = __ballot(o.x > 0.f), by = __ballot(o.y > 0.f);
            const unsigned long long bz = __ballot(o.z > 0.f), bw = __ballot(o.w > 0.f);
            const int lane = threadIdx.x & 63;
            if (lane < 4) mask[((base + i) >> 6) * 4 + lane] = lane == 0 ? bx : (lane == 1 ? by : (lane == 2 ? bz : bw));
        }
    };
    if (INV) {
        load_params((int)threadIdx.x % c4n);
        const long step = (long)gridDim.x * 256 * EW_U;
        for (long i0 = (long)blockIdx.x * 256 * EW_U + threadIdx.x; i0 < hw4; i0 += step) {
            float4 cv[EW_U], rv[EW_U];
#pragma unroll
            for (int u = 0; u < EW_U; ++u) {
                const long i = i0 + u * 256;
                if (i < hw4) {
                    cv[u] = ew_ld(&c4[i]);
                    rv[u] = ew_ld(&r4[i]);
                }
            }
#pragma unroll
            for (int u = 0; u < EW_U; ++u) {
                const long i = i0 + u * 256;
                if (i < hw4) one(i, cv[u], rv[u]);
            }
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += (long)gridDim.x * blockDim.x) {
            load_params((int)(i % c4n));
            one(i, c4[i], r4[i]);
        }
    }
}

// The same tail at a POOLED stage boundary (the next block starts with AvgPool2d(2, 2), reference resnet.py:147-150 /
// SEBasicBlock's `pool`): e itself is only ever read by that pooling (the backward passes read its ReLU-mask bits), so this
// form writes avgpool2(e) [N][H/2][W/2][C] and the bits of e, and e never goes to HBM -- one tensor write and one tensor read
// fewer per boundary.  A thread owns one float4 of an even image row AND the one below it; its horizontal neighbour is c4n lanes
// away (__shfl_xor; 2 c4n divides 64 and a row has a multiple of 64 float4s, so a wave covers whole pixel pairs of one row
// and 64-aligned mask words of both rows); the even-pixel lanes store 0.25 * (((a + b) + c) + d), avgpool2_fwd_kernel's order.
__global__ __launch_bounds__(256) void se_tail_fwd_pool_kernel(
    const float *__restrict__ c, const float *__restrict__ r, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ s, const float *__restrict__ r_scale,
    const float *__restrict__ r_shift, float *__restrict__ pooled, unsigned long long *__restrict__ mask, int H, int W,
    int c4n, int c4shift) {
    const int n = blockIdx.y;
    const int W4 = W * c4n, items = (H >> 1) * W4;
    const size_t base = (size_t)n * H * W4;
    const float4 *c4 = reinterpret_cast<const float4 *>(c) + base, *r4 = reinterpret_cast<const float4 *>(r) + base;
    float4 *o4 = reinterpret_cast<float4 *>(pooled) + (base >> 2);
    const int cx = (int)threadIdx.x & (c4n - 1);
    const float4 sc = reinterpret_cast<const float4 *>(scale)[cx], sh = reinterpret_cast<const float4 *>(shift)[cx];
    const float4 sv = reinterpret_cast<const float4 *>(s)[(size_t)n * c4n + cx];
    float4 rs = make_float4(1.f, 1.f, 1.f, 1.f), rt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r_scale) {
        rs = reinterpret_cast<const float4 *>(r_scale)[cx];
        rt = reinterpret_cast<const float4 *>(r_shift)[cx];
    }
    auto tail = [&](float4 cv, float4 rv) {
        if (r_scale) rv = make_float4(fmaf(rv.x, rs.x, rt.x), fmaf(rv.y, rs.y, rt.y), fmaf(rv.z, rs.z, rt.z), fmaf(rv.w, rs.w, rt.w));
        float4 o;
        o.x = fmaxf((cv.x * sc.x + sh.x) * sv.x + rv.x, 0.f);
        o.y = fmaxf((cv.y * sc.y + sh.y) * sv.y + rv.y, 0.f);
        o.z = fmaxf((cv.z * sc.z + sh.z) * sv.z + rv.z, 0.f);
        o.w = fmaxf((cv.w * sc.w + sh.w) * sv.w + rv.w, 0.f);
        return o;
    };
    auto bits = [&](const float4 &o, size_t i) {          // (the whole wave is here: items and every row are multiples of 64)
        const unsigned long long bx = __ballot(o.x > 0.f), by = __ballot(o.y > 0.f);
        const unsigned long long bz = __ballot(o.z > 0.f), bw = __ballot(o.w > 0.f);
        const int lane = threadIdx.x & 63;
        if (lane < 4) mask[((base + i) >> 6) * 4 + lane] = lane == 0 ? bx : (lane == 1 ? by : (lane == 2 ? bz : bw));
    };
    auto across = [&](const float4 &v) {
        return make_float4(__shfl_xor(v.x, c4n, 64), __shfl_xor(v.y, c4n, 64), __shfl_xor(v.z, c4n, 64), __shfl_xor(v.w, c4n, 64));
    };
    constexpr int U = 2;
    const int step = (int)gridDim.x * 256 * U;
    for (int j0 = (int)blockIdx.x * 256 * U + (int)threadIdx.x; j0 < items; j0 += step) {
        float4 cv[U][2], rv[U][2];
        int src[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * 256;
            if (j < items) {
                const int yp = j / W4;
                src[u] = j + yp * W4;                     // = (2 yp) W4 + (j - yp W4)
                cv[u][0] = ew_ld(&c4[src[u]]);
                rv[u][0] = ew_ld(&r4[src[u]]);
                cv[u][1] = ew_ld(&c4[src[u] + W4]);
                rv[u][1] = ew_ld(&r4[src[u] + W4]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * 256;
            if (j < items) {
                const float4 a = tail(cv[u][0], rv[u][0]), cc = tail(cv[u][1], rv[u][1]);
                if (mask) {
                    bits(a, (size_t)src[u]);
                    bits(cc, (size_t)src[u] + W4);
                }
                const float4 b = across(a), d = across(cc);
                const int yp = j / W4, ce = j - yp * W4, px = ce >> c4shift;
                if (!(px & 1)) {
                    const float4 o = make_float4(0.25f * (a.x + b.x + cc.x + d.x), 0.25f * (a.y + b.y + cc.y + d.y),
                                                 0.25f * (a.z + b.z + cc.z + d.z), 0.25f * (a.w + b.w + cc.w + d.w));
                    o4[((size_t)yp * (W >> 1) + (px >> 1)) * c4n + cx] = o;
                }
            }
        }
    }
}

// One workgroup per sample: dpool[n][:] and this sample's contribution to every parameter gradient / batch sum,
// written to part[n][P] with P = 2*C*Cr + Cr + 3*C laid out [db2 C | dw2 C*Cr | db1 Cr | dw1 Cr*C | sdd C | sddx C] -- the
// order in which these six gradients (se.fc.2.bias, se.fc.2.weight, se.fc.0.bias, se.fc.0.weight, bn2.bias, bn2.weight) lie in
// the flat gradient buffer of dist.FlatParameters (reverse registration order), so `packed` can BE that slice of it;
// a deterministic column sum over the N rows finishes the job.
__global__ __launch_bounds__(256) void se_fc_bwd_sample_kernel(
    const float *__restrict__ sg, const float *__restrict__ sgx, const float *__restrict__ ssum,
    const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ pooled, const float *__restrict__ hid,
    const float *__restrict__ s, const float *__restrict__ w1, const float *__restrict__ w2,
    float *__restrict__ dpool, float *__restrict__ part, int HW, int C, int Cr) {
    __shared__ float dz2[1024];
    __shared__ float pl[1024];
    __shared__ float hd[128];
    __shared__ float dz1[128];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t P = (size_t)2 * C * Cr + Cr + 3 * (size_t)C;
    float *p_db2 = part + (size_t)n * P;
    float *p_dw2 = p_db2 + C;
    float *p_db1 = p_dw2 + (size_t)C * Cr;
    float *p_dw1 = p_db1 + Cr;
    float *p_sdd = p_dw1 + (size_t)Cr * C;
    float *p_sddx = p_sdd + C;
    const float invHW = 1.0f / (float)HW;
    for (int c = tid; c < C; c += 256) {
        const float sgv = sg[(size_t)n * C + c], sgxv = sgx[(size_t)n * C + c], sv = s[(size_t)n * C + c];
        const float ds = gamma[c] * sgxv + beta[c] * sgv;           // sum_hw g * d,  d = xhat*gamma + beta
        const float z2 = ds * sv * (1.f - sv);
        dz2[c] = z2;
        pl[c] = pooled[(size_t)n * C + c];
        p_db2[c] = z2;
    }
    for (int j = tid; j < Cr; j += 256) hd[j] = hid[(size_t)n * Cr + j];
    __syncthreads();
    for (int j = wave; j < Cr; j += 4) {
        float dh = 0.f;
        for (int c = lane; c < C; c += 64) dh += dz2[c] * w2[(size_t)c * Cr + j];
        dh = wave_sum(dh);
        if (lane == 0) {
            const float z1 = hd[j] > 0.f ? dh : 0.f;
            dz1[j] = z1;
            p_db1[j] = z1;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const float z2 = dz2[c], pc = pl[c];
        float dp = 0.f;
        for (int j = 0; j < Cr; ++j) {
            const float z1 = dz1[j];
            dp += z1 * w1[(size_t)j * C + c];
            p_dw1[(size_t)j * C + c] = z1 * pc;
            p_dw2[(size_t)c * Cr + j] = z2 * hd[j];
        }
        dpool[(size_t)n * C + c] = dp;
        const float sv = s[(size_t)n * C + c];
        const float sxhat = (ssum[(size_t)n * C + c] - (float)HW * mean[c]) * invstd[c];     // sum_hw xhat
        p_sdd[c] = sv * sg[(size_t)n * C + c] + dp;
        p_sddx[c] = sv * sgx[(size_t)n * C + c] + dp * invHW * sxhat;
    }
}

// POOLED (INV only): `de` is the gradient of avgpool2(e) -- [N][H/2][W/2][C], the block's output at a pooled stage boundary was
// the pooled tensor (se_tail_fwd_pool_kernel) -- and the gradient of e itself, 0.25 x that spread over 2 x 2 pixels, is formed
// here; it is also WRITTEN to de_out when given (the identity shortcut's share, the addend of conv1's data gradient): no
// avgpool2_bwd launch, and neither pass of this backward reads a full-size de
template <bool INV, bool POOLED = false>
__global__ __launch_bounds__(256) void se_tail_bwd_apply_kernel(
    const float *__restrict__ de, const float *__restrict__ e, const float *__restrict__ c,
    const float *__restrict__ gamma, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ s, const float *__restrict__ dpool, const float *__restrict__ sdd,
    const float *__restrict__ sddx, float *__restrict__ dc, float *__restrict__ dr,
    const unsigned long long *__restrict__ mask, long hw4, int c4n, float invHW, float invR, int W = 0, int c4shift = 0,
    float *__restrict__ de_out = nullptr) {
    const int n = blockIdx.y;
    const size_t base = (size_t)n * hw4;
    const float4 *de4 = reinterpret_cast<const float4 *>(de) + (POOLED ? (base >> 2) : base), *c4 = reinterpret_cast<const float4 *>(c) + base;
    float4 *deo4 = (POOLED && de_out) ? reinterpret_cast<float4 *>(de_out) + base : nullptr;
    const float4 *e4 = e ? reinterpret_cast<const float4 *>(e) + base : nullptr;
    float4 *dc4 = reinterpret_cast<float4 *>(dc) + base, *dr4 = dr ? reinterpret_cast<float4 *>(dr) + base : nullptr;
    float4 ga, m, is, sv, dp, a, b;
    auto load_params = [&](int cx) {
        ga = reinterpret_cast<const float4 *>(gamma)[cx];
        m = reinterpret_cast<const float4 *>(mean)[cx];
        is = reinterpret_cast<const float4 *>(invstd)[cx];
        sv = reinterpret_cast<const float4 *>(s)[(size_t)n * c4n + cx];
        dp = reinterpret_cast<const float4 *>(dpool)[(size_t)n * c4n + cx];
        a = reinterpret_cast<const float4 *>(sdd)[cx];
        b = reinterpret_cast<const float4 *>(sddx)[cx];
    };
    auto one = [&](long i, float4 d, float4 cv, bool px, bool py, bool pz, bool pw) {
        float4 g, o;
        g.x = px ? d.x : 0.f;
        g.y = py ? d.y : 0.f;
        g.z = pz ? d.z : 0.f;
        g.w = pw ? d.w : 0.f;
        o.x = ga.x * is.x * (g.x * sv.x + dp.x * invHW - a.x * invR - (cv.x - m.x) * is.x * b.x * invR);
        o.y = ga.y * is.y * (g.y * sv.y + dp.y * invHW - a.y * invR - (cv.y - m.y) * is.y * b.y * invR);
        o.z = ga.z * is.z * (g.z * sv.z + dp.z * invHW - a.z * invR - (cv.z - m.z) * is.z * b.z * invR);
        o.w = ga.w * is.w * (g.w * sv.w + dp.w * invHW - a.w * invR - (cv.w - m.w) * is.w * b.w * invR);
        ew_st(&dc4[i], o);
        if (dr4) dr4[i] = g;
    };
    if (INV) {
        load_params((int)threadIdx.x % c4n);
        const long step = (long)gridDim.x * 256 * EW_U;
        for (long i0 = (long)blockIdx.x * 256 * EW_U + threadIdx.x; i0 < hw4; i0 += step) {
            float4 d[EW_U], cv[EW_U];
            bool px[EW_U], py[EW_U], pz[EW_U], pw[EW_U];
#pragma unroll
            for (int u = 0; u < EW_U; ++u) {
                const long i = i0 + u * 256;
                if (i < hw4) {
                    if (POOLED) {
                        const int pix = (int)(i >> c4shift), y = pix / W, x = pix - y * W;
                        const float4 g4 = de4[(((size_t)(y >> 1) * (W >> 1) + (x >> 1)) << c4shift) + (i & (c4n - 1))];
                        d[u] = make_float4(0.25f * g4.x, 0.25f * g4.y, 0.25f * g4.z, 0.25f * g4.w);
                    } else {
                        d[u] = ew_ld(&de4[i]);
                    }
                    cv[u] = ew_ld(&c4[i]);
                    if (mask) {
                        mask_bits4(mask, base + i, px[u], py[u], pz[u], pw[u]);
                    } else {
                        const float4 ev = e4[i];
                        px[u] = ev.x > 0.f; py[u] = ev.y > 0.f; pz[u] = ev.z > 0.f; pw[u] = ev.w > 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < EW_U; ++u) {
                const long i = i0 + u * 256;
                if (i < hw4) {
                    one(i, d[u], cv[u], px[u], py[u], pz[u], pw[u]);
                    if (POOLED && deo4) ew_st(&deo4[i], d[u]);
                }
            }
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += (long)gridDim.x * blockDim.x) {
            load_params((int)(i % c4n));
            bool px, py, pz, pw;
            if (mask) {
                mask_bits4(mask, base + i, px, py, pz, pw);
            } else {
                const float4 ev = e4[i];
                px = ev.x > 0.f; py = ev.y > 0.f; pz = ev.z > 0.f; pw = ev.w > 0.f;
            }
            one(i, de4[i], c4[i], px, py, pz, pw);
        }
    }
}

// ---- AvgPool2d(2,2) -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                           int H, int W, int c4n, long total4) {
    const int Ho = H >> 1, Wo = W >> 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int cx = (int)(i % c4n);
        long p = i / c4n;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho);
        const long n = p / Ho;
        const float4 *src = reinterpret_cast<const float4 *>(x) + (((size_t)n * H + 2 * ho) * W + 2 * wo) * c4n + cx;
        const float4 a = ew_ld(src), b = ew_ld(src + c4n), c = ew_ld(src + (size_t)W * c4n), d = ew_ld(src + (size_t)W * c4n + c4n);
        reinterpret_cast<float4 *>(y)[i] = make_float4(0.25f * (a.x + b.x + c.x + d.x), 0.25f * (a.y + b.y + c.y + d.y),
                                                       0.25f * (a.z + b.z + c.z + d.z), 0.25f * (a.w + b.w + c.w + d.w));
    }
}
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx,
                                                           int H, int W, int c4n, long total4) {
    // total4 over the INPUT (H x W) grid
    const int Ho = H >> 1, Wo = W >> 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int cx = (int)(i % c4n);
        long p = i / c4n;
        const int w = (int)(p % W);
        p /= W;
        const int h = (int)(p % H);
        const long n = p / H;
        const float4 g = reinterpret_cast<const float4 *>(dy)[(((size_t)n * Ho + (h >> 1)) * Wo + (w >> 1)) * c4n + cx];
        ew_st(&reinterpret_cast<float4 *>(dx)[i], make_float4(0.25f * g.x, 0.25f * g.y, 0.25f * g.z, 0.25f * g.w));
    }
}

static inline int ew_grid(long n4) {
    long g = (n4 + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
static inline bool chan_ok(int C) { return C >= 4 && C <= 1024 && (C % 4) == 0 && (256 % (C / 4)) == 0; }
static inline int pick_G(int N, int HW) {
    int G = 1024 / N;
    if (G < 1) G = 1;
    const int need = (HW + 63) / 64;
    if (G > need) G = need;
    return G;
}

}  // namespace adyolo

using namespace adyolo;

// workspace `partial`: 4*1024*C floats = [2*1024*C stage-1 partials][1024*C per-sample sum][1024*C per-sample sum-sq]
extern "C" int adyolo_bn_stats(const float *x, float *ssum, float *mean, float *invstd, float *running_mean,
                               float *running_var, float *partial, int N, int HW, int C, float momentum, float eps,
                               void *stream) {
    ADYOLO_REQUIRE(x && mean && invstd && partial && N > 0 && HW > 0, ADYOLO_EINVAL, "bn_stats: bad arguments");
    ADYOLO_REQUIRE(chan_ok(C) && N <= 1024, ADYOLO_ENOSUP, "bn_stats: unsupported C=%d or N=%d", C, N);
    hipStream_t st = as_stream(stream);
    const int G = pick_G(N, HW);
    StatsF f{x, HW, C};
    hipLaunchKernelGGL((reduce2_partial_kernel<StatsF>), dim3(G, N), dim3(256), 0, st, f, partial, HW, C, G);
    int rc = check_launch("bn_stats_partial");
    if (rc) return rc;
    float *ps0 = ssum ? ssum : partial + (size_t)2 * 1024 * C;
    float *ps1 = partial + (size_t)3 * 1024 * C;
    hipLaunchKernelGGL(persample_reduce_kernel, dim3(cdiv(C, 32), N), dim3(256), 0, st, partial, ps0, ps1, N, G, C);
    rc = check_launch("bn_stats_persample");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finish_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, ps0, ps1, mean, invstd, running_mean,
                       running_var, (const float *)nullptr, (const float *)nullptr, (float *)nullptr, (float *)nullptr, N, C,
                       (double)N * (double)HW, momentum, eps);
    return check_launch("bn_stats_final");
}

extern "C" int adyolo_bn_stats_tiles(const float *tile_stats, float *ssum, float *mean, float *invstd,
                                     float *running_mean, float *running_var, const float *gamma, const float *beta,
                                     float *scale, float *shift, float *partial, int N, int G, int HW, int C,
                                     float momentum, float eps, void *stream) {
    ADYOLO_REQUIRE(tile_stats && mean && invstd && partial && N > 0 && G > 0 && HW > 0 && C > 0 && N <= 1024,
                   ADYOLO_EINVAL, "bn_stats_tiles: bad arguments");
    ADYOLO_REQUIRE(!scale || (gamma && beta && shift), ADYOLO_EINVAL, "bn_stats_tiles: scale needs gamma, beta and shift");
    hipStream_t st = as_stream(stream);
    float *ps0 = ssum ? ssum : partial;
    float *ps1 = partial + (size_t)1024 * C;
    hipLaunchKernelGGL(persample_reduce_kernel, dim3(cdiv(C, 32), N), dim3(256), 0, st, tile_stats, ps0, ps1, N, G, C);
    int rc = check_launch("bn_stats_tiles_persample");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finish_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, ps0, ps1, mean, invstd, running_mean,
                       running_var, gamma, beta, scale, shift, N, C, (double)N * (double)HW, momentum, eps);
    return check_launch("bn_stats_tiles_final");
}

// the two halves of adyolo_bn_stats_tiles as separate calls: under exact data parallelism the per-sample sums of all ranks
// are gathered between them, so that N ranks compute the statistics of the concatenated batch (bit-identical to one device)
extern "C" int adyolo_bn_persample(const float *tile_stats, float *ps0, float *ps1, int N, int G, int C, void *stream) {
    ADYOLO_REQUIRE(tile_stats && ps0 && ps1 && N > 0 && G > 0 && C > 0, ADYOLO_EINVAL, "bn_persample: bad arguments");
    hipLaunchKernelGGL(persample_reduce_kernel, dim3(cdiv(C, 32), N), dim3(256), 0, as_stream(stream), tile_stats, ps0, ps1,
                       N, G, C);
    return check_launch("bn_persample");
}

extern "C" int adyolo_bn_finish(const float *ps0, const float *ps1, float *mean, float *invstd, float *running_mean,
                                float *running_var, const float *gamma, const float *beta, float *scale, float *shift,
                                int N, int HW, int C, float momentum, float eps, void *stream) {
    ADYOLO_REQUIRE(ps0 && ps1 && mean && invstd && N > 0 && HW > 0 && C > 0, ADYOLO_EINVAL, "bn_finish: bad arguments");
    ADYOLO_REQUIRE(!scale || (gamma && beta && shift), ADYOLO_EINVAL, "bn_finish: scale needs gamma, beta and shift");
    hipLaunchKernelGGL(bn_finish_kernel, dim3(cdiv(C, 32)), dim3(256), 0, as_stream(stream), ps0, ps1, mean, invstd,
                       running_mean, running_var, gamma, beta, scale, shift, N, C, (double)N * (double)HW, momentum, eps);
    return check_launch("bn_finish");
}

extern "C" int adyolo_bn_eval_stats(const float *running_mean, const float *running_var, float *mean, float *invstd,
                                    int C, float eps, void *stream) {
    ADYOLO_REQUIRE(running_mean && running_var && mean && invstd && C > 0, ADYOLO_EINVAL, "bn_eval_stats: bad arguments");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), running_mean,
                       running_var, mean, invstd, C, eps);
    return check_launch("bn_eval_stats");
}

extern "C" int adyolo_bn_scale_shift(const float *gamma, const float *beta, const float *mean, const float *invstd,
                                     float *scale, float *shift, int C, void *stream) {
    ADYOLO_REQUIRE(gamma && beta && mean && invstd && scale && shift && C > 0, ADYOLO_EINVAL, "bn_scale_shift: bad arguments");
    hipLaunchKernelGGL(bn_scale_shift_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), gamma, beta, mean,
                       invstd, scale, shift, C);
    return check_launch("bn_scale_shift");
}

extern "C" int adyolo_affine_nhwc(const float *x, const float *scale, const float *shift, float *y, long rows, int C,
                                  void *stream) {
    ADYOLO_REQUIRE(x && scale && shift && y && rows > 0 && C > 0 && C % 4 == 0, ADYOLO_EINVAL, "affine_nhwc: bad arguments");
    const long n4 = rows * (C / 4);
    hipLaunchKernelGGL(affine_kernel, dim3(ew_grid(n4)), dim3(256), 0, as_stream(stream), x, scale, shift, y, n4, C / 4);
    return check_launch("affine_nhwc");
}

extern "C" int adyolo_bn_bwd_reduce(const float *dy, const float *x, const float *mean, const float *invstd,
                                    float *sdy, float *sdyx, float *partial, long rows, int C, void *stream) {
    ADYOLO_REQUIRE(dy && x && mean && invstd && sdy && sdyx && partial && rows > 0, ADYOLO_EINVAL, "bn_bwd_reduce: bad arguments");
    ADYOLO_REQUIRE(chan_ok(C) && rows < (1L << 31), ADYOLO_ENOSUP, "bn_bwd_reduce: unsupported C=%d", C);
    hipStream_t st = as_stream(stream);
    const int HW = (int)rows;
    int G = (HW + 63) / 64;
    if (G > 1024) G = 1024;
    BnBwdF f{dy, x, mean, invstd, HW, C};
    hipLaunchKernelGGL((reduce2_partial_kernel<BnBwdF>), dim3(G, 1), dim3(256), 0, st, f, partial, HW, C, G);
    int rc = check_launch("bn_bwd_reduce_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(persample_reduce_kernel, dim3(cdiv(C, 32), 1), dim3(256), 0, st, partial, sdy, sdyx, 1, G, C);
    return check_launch("bn_bwd_reduce_final");
}

extern "C" int adyolo_bn_bwd_tiles(const float *tile_stats, float *sdy, float *sdyx, float *partial, int tiles, int C,
                                   void *stream) {
    ADYOLO_REQUIRE(tile_stats && sdy && sdyx && partial && tiles > 0 && C > 0, ADYOLO_EINVAL, "bn_bwd_tiles: bad arguments");
    // two stages (a single stage would leave C/32 workgroups summing tens of thousands of patches each):
    // groups = the largest divisor of `tiles` <= 256;  partial = [2][256][C] floats
    int groups = 1;
    for (int g = 256; g > 1; --g)
        if (tiles % g == 0) {
            groups = g;
            break;
        }
    hipStream_t st = as_stream(stream);
    if (groups == 1) {
        hipLaunchKernelGGL(persample_reduce_kernel, dim3(cdiv(C, 32), 1), dim3(256), 0, st, tile_stats, sdy, sdyx, 1, tiles, C);
        return check_launch("bn_bwd_tiles");
    }
    hipLaunchKernelGGL(persample_reduce_kernel, dim3(cdiv(C, 32), groups), dim3(256), 0, st, tile_stats, partial,
                       partial + (size_t)groups * C, groups, tiles / groups, C);
    int rc = check_launch("bn_bwd_tiles_groups");
    if (rc) return rc;
    hipLaunchKernelGGL(persample_reduce_kernel, dim3(cdiv(C, 32), 1), dim3(256), 0, st, partial, sdy, sdyx, 1, groups, C);
    return check_launch("bn_bwd_tiles");
}

extern "C" int adyolo_bn_bwd_apply(const float *dy, const float *x, const float *gamma, const float *mean,
                                   const float *invstd, const float *sdy, const float *sdyx, float *dx, float *dgamma,
                                   float *dbeta, float *dx_colsum, float *colsum_partial, long rows, int C,
                                   int relu_mask, float count_scale, void *stream) {
    ADYOLO_REQUIRE(dy && x && gamma && mean && invstd && sdy && sdyx && dx && rows > 0 && C % 4 == 0 && count_scale >= 1.f,
                   ADYOLO_EINVAL, "bn_bwd_apply: bad arguments");
    const float inv_r = (float)(1.0 / ((double)rows * (double)count_scale));      // batch sums over count_scale equal micro-batches
    hipStream_t st = as_stream(stream);
    const long n4 = rows * (C / 4);
    ADYOLO_REQUIRE(!dx_colsum || (colsum_partial && 256 % (C / 4) == 0), ADYOLO_EINVAL,
                   "bn_bwd_apply: dx_colsum needs a [8192][C] partial workspace and C/4 dividing 256");
    const bool inv = 256 % (C / 4) == 0;
    const int grid = ew_grid(inv ? cdiv(n4, (long)EW_U) : n4);
    if (inv)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(grid), dim3(256), 0, st, dy, x, gamma, mean, invstd, sdy, sdyx, dx,
                           dx_colsum ? colsum_partial : (float *)nullptr, n4, C / 4, inv_r, relu_mask);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid), dim3(256), 0, st, dy, x, gamma, mean, invstd, sdy, sdyx, dx,
                           dx_colsum ? colsum_partial : (float *)nullptr, n4, C / 4, inv_r, relu_mask);
    int rc = check_launch("bn_bwd_apply");
    if (rc) return rc;
    if (dx_colsum) {
        hipLaunchKernelGGL(partial_colsum_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, colsum_partial, dx_colsum, grid, C);
        rc = check_launch("bn_bwd_apply_colsum");
        if (rc) return rc;
    }
    if (dgamma || dbeta) {
        hipLaunchKernelGGL(accum2_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, sdyx, sdy, dgamma, dbeta, C);
        rc = check_launch("bn_bwd_accum");
    }
    return rc;
}

extern "C" int adyolo_se_fc_fwd(const float *ssum, const float *scale, const float *shift, const float *w1,
                                const float *b1, const float *w2, const float *b2, float *pooled, float *hid, float *s,
                                int N, int HW, int C, int Cr, void *stream) {
    ADYOLO_REQUIRE(ssum && scale && shift && w1 && b1 && w2 && b2 && pooled && hid && s && N > 0, ADYOLO_EINVAL,
                   "se_fc_fwd: bad arguments");
    ADYOLO_REQUIRE(C <= 1024 && Cr <= 128, ADYOLO_ENOSUP, "se_fc_fwd: C=%d Cr=%d too large", C, Cr);
    hipLaunchKernelGGL(se_fc_fwd_kernel, dim3(N), dim3(256), 0, as_stream(stream), ssum, scale, shift, w1, b1, w2, b2,
                       pooled, hid, s, HW, C, Cr);
    return check_launch("se_fc_fwd");
}

extern "C" long adyolo_relu_mask_words(int N, int HW, int C) {
    const long hw4 = (long)HW * (C / 4);
    return (C % 4 == 0 && hw4 % 64 == 0) ? (long)N * hw4 / 64 * 4 : 0;
}

extern "C" int adyolo_se_tail_fwd(const float *c, const float *r, const float *scale, const float *shift,
                                  const float *s, const float *r_scale, const float *r_shift, float *e, uint64_t *mask,
                                  int N, int HW, int C, void *stream) {
    ADYOLO_REQUIRE(c && r && scale && shift && s && e && N > 0 && HW > 0 && C % 4 == 0 && (!r_scale == !r_shift),
                   ADYOLO_EINVAL, "se_tail_fwd: bad arguments");
    const long hw4 = (long)HW * (C / 4);
    ADYOLO_REQUIRE(!mask || hw4 % 64 == 0, ADYOLO_ENOSUP, "se_tail_fwd: mask bits need HW*C/4 %% 64 == 0 (HW=%d C=%d)", HW, C);
    const bool inv = 256 % (C / 4) == 0;
    int gx = ew_grid(inv ? cdiv(hw4, (long)EW_U) : hw4);
    if ((long)gx * N > 16384) gx = (int)(16384 / N > 0 ? 16384 / N : 1);
    if (inv)
        hipLaunchKernelGGL(se_tail_fwd_kernel<true>, dim3(gx, N), dim3(256), 0, as_stream(stream), c, r, scale, shift, s,
                           r_scale, r_shift, e, reinterpret_cast<unsigned long long *>(mask), hw4, C / 4);
    else
        hipLaunchKernelGGL(se_tail_fwd_kernel<false>, dim3(gx, N), dim3(256), 0, as_stream(stream), c, r, scale, shift, s,
                           r_scale, r_shift, e, reinterpret_cast<unsigned long long *>(mask), hw4, C / 4);
    return check_launch("se_tail_fwd");
}

// 1 when adyolo_se_tail_fwd_pool takes the shape (C/4 a power of two <= 32, even H and W, a row of W*C/4 float4 a multiple of 64)
extern "C" int adyolo_se_tail_fwd_pool_ok(int H, int W, int C) {
    if (C % 4 || H <= 0 || W <= 0) return 0;
    const int c4n = C / 4;
    return chan_ok(C) && (c4n & (c4n - 1)) == 0 && c4n <= 32 && H % 2 == 0 && W % 2 == 0 && ((long)W * c4n) % 64 == 0 &&
           (long)H * W * c4n < (1L << 30);
}

extern "C" int adyolo_se_tail_fwd_pool(const float *c, const float *r, const float *scale, const float *shift, const float *s,
                                       const float *r_scale, const float *r_shift, float *pooled, uint64_t *mask, int N, int H,
                                       int W, int C, void *stream) {
    ADYOLO_REQUIRE(c && r && scale && shift && s && pooled && N > 0 && (!r_scale) == (!r_shift), ADYOLO_EINVAL,
                   "se_tail_fwd_pool: bad arguments");
    ADYOLO_REQUIRE(adyolo_se_tail_fwd_pool_ok(H, W, C), ADYOLO_ENOSUP, "se_tail_fwd_pool: unsupported shape H=%d W=%d C=%d", H, W, C);
    const int c4n = C / 4;
    int c4shift = 0;
    while ((1 << c4shift) < c4n) ++c4shift;
    const long items = (long)(H / 2) * W * c4n;
    int gx = ew_grid(cdiv(items, 2L));
    if ((long)gx * N > 16384) gx = (int)(16384 / N > 0 ? 16384 / N : 1);
    hipLaunchKernelGGL(se_tail_fwd_pool_kernel, dim3(gx, N), dim3(256), 0, as_stream(stream), c, r, scale, shift, s, r_scale,
                       r_shift, pooled, reinterpret_cast<unsigned long long *>(mask), H, W, c4n, c4shift);
    return check_launch("se_tail_fwd_pool");
}

extern "C" int adyolo_se_tail_bwd_reduce(const float *de, const float *e, const uint64_t *mask, const float *c,
                                         const float *mean, const float *invstd, float *sg, float *sgx, float *partial,
                                         int N, int HW, int C, void *stream) {
    ADYOLO_REQUIRE(de && (e || mask) && c && mean && invstd && sg && sgx && partial && N > 0 && HW > 0, ADYOLO_EINVAL,
                   "se_tail_bwd_reduce: bad arguments");
    ADYOLO_REQUIRE(chan_ok(C) && N <= 1024, ADYOLO_ENOSUP, "se_tail_bwd_reduce: unsupported C=%d or N=%d", C, N);
    hipStream_t st = as_stream(stream);
    const int G = pick_G(N, HW);
    SeBwdF f{de, e, c, mean, invstd, reinterpret_cast<const unsigned long long *>(mask), HW, C, 0};
    hipLaunchKernelGGL((reduce2_partial_kernel<SeBwdF>), dim3(G, N), dim3(256), 0, st, f, partial, HW, C, G);
    int rc = check_launch("se_tail_bwd_reduce_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(persample_reduce_kernel, dim3(cdiv(C, 32), N), dim3(256), 0, st, partial, sg, sgx, N, G, C);
    return check_launch("se_tail_bwd_reduce_final");
}

// ... with the gradient of avgpool2(e) in place of de (see se_tail_bwd_apply_kernel<INV, POOLED>)
extern "C" int adyolo_se_tail_bwd_reduce_pooled(const float *dpooled, const uint64_t *mask, const float *c, const float *mean,
                                                const float *invstd, float *sg, float *sgx, float *partial, int N, int H, int W,
                                                int C, void *stream) {
    ADYOLO_REQUIRE(dpooled && mask && c && mean && invstd && sg && sgx && partial && N > 0 && H > 0 && W > 0, ADYOLO_EINVAL,
                   "se_tail_bwd_reduce_pooled: bad arguments");
    ADYOLO_REQUIRE(chan_ok(C) && N <= 1024 && adyolo_se_tail_fwd_pool_ok(H, W, C), ADYOLO_ENOSUP,
                   "se_tail_bwd_reduce_pooled: unsupported shape H=%d W=%d C=%d N=%d", H, W, C, N);
    hipStream_t st = as_stream(stream);
    const int HW = H * W;
    const int G = pick_G(N, HW);
    SeBwdF f{dpooled, nullptr, c, mean, invstd, reinterpret_cast<const unsigned long long *>(mask), HW, C, W};
    hipLaunchKernelGGL((reduce2_partial_kernel<SeBwdF>), dim3(G, N), dim3(256), 0, st, f, partial, HW, C, G);
    int rc = check_launch("se_tail_bwd_reduce_pooled_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(persample_reduce_kernel, dim3(cdiv(C, 32), N), dim3(256), 0, st, partial, sg, sgx, N, G, C);
    return check_launch("se_tail_bwd_reduce_pooled_final");
}

// the same two per-sample sums from the per-patch sums a convolution epilogue already produced (stat_mask = e,
// stat_aux = c): tile_stats [2][N*G][C] -> sg, sgx [N][C]
extern "C" int adyolo_se_tail_bwd_tiles(const float *tile_stats, float *sg, float *sgx, int N, int G, int C, void *stream) {
    ADYOLO_REQUIRE(tile_stats && sg && sgx && N > 0 && G > 0 && C > 0, ADYOLO_EINVAL, "se_tail_bwd_tiles: bad arguments");
    hipLaunchKernelGGL(persample_reduce_kernel, dim3(cdiv(C, 32), N), dim3(256), 0, as_stream(stream), tile_stats, sg, sgx,
                       N, G, C);
    return check_launch("se_tail_bwd_tiles");
}

extern "C" long adyolo_se_fc_bwd_words(int C, int Cr) { return 2L * C * Cr + Cr + 3L * C; }

extern "C" int adyolo_se_fc_bwd(const float *sg, const float *sgx, const float *ssum, const float *gamma,
                                const float *beta, const float *mean, const float *invstd, const float *pooled,
                                const float *hid, const float *s, const float *w1, const float *w2, float *dpool,
                                float *part, float *packed, float *colsum_ws, int N, int HW, int C, int Cr,
                                void *stream) {
    ADYOLO_REQUIRE(sg && sgx && ssum && gamma && beta && mean && invstd && pooled && hid && s && w1 && w2 && dpool &&
                       part && packed && colsum_ws && N > 0,
                   ADYOLO_EINVAL, "se_fc_bwd: bad arguments");
    ADYOLO_REQUIRE(C <= 1024 && Cr <= 128, ADYOLO_ENOSUP, "se_fc_bwd: C=%d (<=1024) Cr=%d (<=128)", C, Cr);
    hipLaunchKernelGGL(se_fc_bwd_sample_kernel, dim3(N), dim3(256), 0, as_stream(stream), sg, sgx, ssum, gamma, beta,
                       mean, invstd, pooled, hid, s, w1, w2, dpool, part, HW, C, Cr);
    int rc = check_launch("se_fc_bwd_sample");
    if (rc) return rc;
    const long P = adyolo_se_fc_bwd_words(C, Cr);
    return adyolo_colsum(part, packed, colsum_ws, N, (int)P, (int)P, 0, stream);
}

extern "C" int adyolo_se_tail_bwd_apply(const float *de, const float *e, const uint64_t *mask, const float *c,
                                        const float *gamma, const float *mean, const float *invstd, const float *s,
                                        const float *dpool, const float *sdd, const float *sddx, float *dc, float *dr,
                                        int N, int HW, int C, float count_scale, void *stream) {
    ADYOLO_REQUIRE(de && (e || mask) && c && gamma && mean && invstd && s && dpool && sdd && sddx && dc && N > 0 &&
                       HW > 0 && C % 4 == 0 && count_scale >= 1.f,
                   ADYOLO_EINVAL, "se_tail_bwd_apply: bad arguments");
    const float inv_r = (float)(1.0 / ((double)N * (double)HW * (double)count_scale));
    const long hw4 = (long)HW * (C / 4);
    const bool inv = 256 % (C / 4) == 0;
    int gx = ew_grid(inv ? cdiv(hw4, (long)EW_U) : hw4);
    if ((long)gx * N > 16384) gx = (int)(16384 / N > 0 ? 16384 / N : 1);
    if (inv)
        hipLaunchKernelGGL(se_tail_bwd_apply_kernel<true>, dim3(gx, N), dim3(256), 0, as_stream(stream), de, e, c, gamma,
                           mean, invstd, s, dpool, sdd, sddx, dc, dr, reinterpret_cast<const unsigned long long *>(mask),
                           hw4, C / 4, 1.0f / (float)HW, inv_r);
    else
        hipLaunchKernelGGL(se_tail_bwd_apply_kernel<false>, dim3(gx, N), dim3(256), 0, as_stream(stream), de, e, c, gamma,
                           mean, invstd, s, dpool, sdd, sddx, dc, dr, reinterpret_cast<const unsigned long long *>(mask),
                           hw4, C / 4, 1.0f / (float)HW, inv_r);
    return check_launch("se_tail_bwd_apply");
}

extern "C" int adyolo_se_tail_bwd_apply_pooled(const float *dpooled, const uint64_t *mask, const float *c, const float *gamma,
                                               const float *mean, const float *invstd, const float *s, const float *dpool,
                                               const float *sdd, const float *sddx, float *dc, float *dr, float *de_out, int N,
                                               int H, int W, int C, float count_scale, void *stream) {
    ADYOLO_REQUIRE(dpooled && mask && c && gamma && mean && invstd && s && dpool && sdd && sddx && dc && N > 0 && H > 0 && W > 0 &&
                       count_scale >= 1.f,
                   ADYOLO_EINVAL, "se_tail_bwd_apply_pooled: bad arguments");
    ADYOLO_REQUIRE(adyolo_se_tail_fwd_pool_ok(H, W, C), ADYOLO_ENOSUP, "se_tail_bwd_apply_pooled: unsupported shape H=%d W=%d C=%d",
                   H, W, C);
    const int HW = H * W, c4n = C / 4;
    int c4shift = 0;
    while ((1 << c4shift) < c4n) ++c4shift;
    const float inv_r = (float)(1.0 / ((double)N * (double)HW * (double)count_scale));
    const long hw4 = (long)HW * c4n;
    int gx = ew_grid(cdiv(hw4, (long)EW_U));
    if ((long)gx * N > 16384) gx = (int)(16384 / N > 0 ? 16384 / N : 1);
    hipLaunchKernelGGL((se_tail_bwd_apply_kernel<true, true>), dim3(gx, N), dim3(256), 0, as_stream(stream), dpooled,
                       (const float *)nullptr, c, gamma, mean, invstd, s, dpool, sdd, sddx, dc, dr,
                       reinterpret_cast<const unsigned long long *>(mask), hw4, c4n, 1.0f / (float)HW, inv_r, W, c4shift, de_out);
    return check_launch("se_tail_bwd_apply_pooled");
}

extern "C" int adyolo_avgpool2_fwd(const float *x, float *y, int N, int H, int W, int C, void *stream) {
    ADYOLO_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, ADYOLO_EINVAL,
                   "avgpool2_fwd: H, W must be even and C %% 4 == 0");
    const long total4 = (long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(ew_grid(total4)), dim3(256), 0, as_stream(stream), x, y, H, W, C / 4,
                       total4);
    return check_launch("avgpool2_fwd");
}
extern "C" int adyolo_avgpool2_bwd(const float *dy, float *dx, int N, int H, int W, int C, void *stream) {
    ADYOLO_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, ADYOLO_EINVAL,
                   "avgpool2_bwd: H, W must be even and C %% 4 == 0");
    const long total4 = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(ew_grid(total4)), dim3(256), 0, as_stream(stream), dy, dx, H, W, C / 4,
                       total4);
    return check_launch("avgpool2_bwd");
}
