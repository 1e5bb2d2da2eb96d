// K5 / K6 / K6b: self-attention pooling, the bidirectional GRU recurrence, LayerNorm+tanh, dropout mask.
// Replaces SelfAttentionPooling (/root/reference/src/models/backbones/resnet.py:109-123), nn.GRU
// (:153,195), nn.LayerNorm + tanh (:154,196-197).
//
// GRU: the recurrence is independent per (sample, direction), so one workgroup owns one (sample, direction)
// for the whole sequence: thread j of 384 keeps row j of W_hh (128 floats) in VGPRs for all T steps, the
// hidden state lives in LDS and is broadcast-read; two barriers per step.  Nothing is re-read from HBM
// inside the time loop except the pre-computed input projections gx (prefetched one step ahead).
#include "common.hpp"

namespace adyolo {

// ---------------------------------------------------------------------------------------- SAP (C == 256)
template <int F>
__global__ __launch_bounds__(256) void sap_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ b, float *__restrict__ y,
                                                      float *__restrict__ attn, int R) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float4 wv = reinterpret_cast<const float4 *>(w)[lane];
    const float4 *xr = reinterpret_cast<const float4 *>(x + (size_t)row * F * 256);
    float4 xv[F];
    float lg[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        xv[f] = xr[f * 64 + lane];
        lg[f] = xv[f].x * wv.x + xv[f].y * wv.y + xv[f].z * wv.z + xv[f].w * wv.w;
    }
#pragma unroll
    for (int f = 0; f < F; ++f) lg[f] = wave_sum(lg[f]) + b[0];
    float mx = lg[0];
#pragma unroll
    for (int f = 1; f < F; ++f) mx = fmaxf(mx, lg[f]);
    float den = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        lg[f] = expf(lg[f] - mx);
        den += lg[f];
    }
    const float inv = 1.0f / den;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const float a = lg[f] * inv;
        o.x += a * xv[f].x; o.y += a * xv[f].y; o.z += a * xv[f].z; o.w += a * xv[f].w;
        if (lane == f) attn[(size_t)row * F + f] = a;
    }
    reinterpret_cast<float4 *>(y + (size_t)row * 256)[lane] = o;
}

template <int F>
__global__ __launch_bounds__(256) void sap_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                      const float *__restrict__ w, const float *__restrict__ attn,
                                                      float *__restrict__ dx, float *__restrict__ partial, int R,
                                                      int rows_per_block) {
    // partial [gridDim.x][260]: dW (256) + db (1)
    __shared__ float red[4][260];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 wv = reinterpret_cast<const float4 *>(w)[lane];
    float4 dwacc = make_float4(0.f, 0.f, 0.f, 0.f);
    float dbacc = 0.f;
    const int rbeg = blockIdx.x * rows_per_block;
    const int rend = min(R, rbeg + rows_per_block);
    for (int row = rbeg + wave; row < rend; row += 4) {
        const float4 g = reinterpret_cast<const float4 *>(dy + (size_t)row * 256)[lane];
        const float4 *xr = reinterpret_cast<const float4 *>(x + (size_t)row * F * 256);
        float4 xv[F];
        float da[F], at[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            xv[f] = xr[f * 64 + lane];
            da[f] = xv[f].x * g.x + xv[f].y * g.y + xv[f].z * g.z + xv[f].w * g.w;
            at[f] = attn[(size_t)row * F + f];
        }
        float dot = 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            da[f] = wave_sum(da[f]);
            dot += at[f] * da[f];
        }
        float4 *dxr = reinterpret_cast<float4 *>(dx + (size_t)row * F * 256);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float dl = at[f] * (da[f] - dot);       // d loss / d logit_f
            float4 o;
            o.x = at[f] * g.x + dl * wv.x;
            o.y = at[f] * g.y + dl * wv.y;
            o.z = at[f] * g.z + dl * wv.z;
            o.w = at[f] * g.w + dl * wv.w;
            dxr[f * 64 + lane] = o;
            dwacc.x += dl * xv[f].x; dwacc.y += dl * xv[f].y; dwacc.z += dl * xv[f].z; dwacc.w += dl * xv[f].w;
            dbacc += dl;
        }
    }
    red[wave][lane * 4 + 0] = dwacc.x;
    red[wave][lane * 4 + 1] = dwacc.y;
    red[wave][lane * 4 + 2] = dwacc.z;
    red[wave][lane * 4 + 3] = dwacc.w;
    if (lane == 0) red[wave][256] = dbacc;
    __syncthreads();
    for (int c = threadIdx.x; c < 257; c += 256)
        partial[(size_t)blockIdx.x * 260 + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

__global__ __launch_bounds__(256) void sap_bwd_final_kernel(const float *__restrict__ partial, float *__restrict__ dw,
                                                            float *__restrict__ db, int nblk) {
    __shared__ double red[256];
    const double s = block_colsum32(partial, nblk, 260, blockIdx.x * 32, 257, red);
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) == 0 && c <= 256) {
        if (c < 256) dw[c] += (float)s;
        else db[0] += (float)s;
    }
}

// ---------------------------------------------------------------------------------------- GRU (H = 128)
constexpr int GH = 128;

__global__ __launch_bounds__(384) void gru_fwd_kernel(const float *__restrict__ gx, const float *__restrict__ whh,
                                                      const float *__restrict__ bhh, float *__restrict__ out,
                                                      float *__restrict__ gates, float *__restrict__ hprev, int T) {
    __shared__ __attribute__((aligned(16))) float hs[GH];
    __shared__ float gh[3 * GH];
    const int j = threadIdx.x;              // gate row 0..383
    const int b = blockIdx.x, dir = blockIdx.y;
    float wrow[GH];
    const float *wsrc = whh + ((size_t)dir * 3 * GH + j) * GH;
#pragma unroll
    for (int k = 0; k < GH; k += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(wsrc + k);
        wrow[k] = v.x; wrow[k + 1] = v.y; wrow[k + 2] = v.z; wrow[k + 3] = v.w;
    }
    const float bj = bhh[dir * 3 * GH + j];
    if (j < GH) hs[j] = 0.f;
    __syncthreads();
    const size_t gx_b = (size_t)b * T * 2 * 3 * GH;
    int t = dir ? T - 1 : 0;
    const int dt = dir ? -1 : 1;
    float gxr = 0.f, gxz = 0.f, gxn = 0.f;
    if (j < GH) {
        const float *g = gx + gx_b + ((size_t)t * 2 + dir) * 3 * GH;
        gxr = g[j]; gxz = g[GH + j]; gxn = g[2 * GH + j];
    }
    for (int step = 0; step < T; ++step, t += dt) {
        float nr = 0.f, nz = 0.f, nn = 0.f;
        if (j < GH && step + 1 < T) {           // prefetch next step's input projection
            const float *g = gx + gx_b + ((size_t)(t + dt) * 2 + dir) * 3 * GH;
            nr = g[j]; nz = g[GH + j]; nn = g[2 * GH + j];
        }
        float a = bj;
#pragma unroll
        for (int k = 0; k < GH; k += 4) {
            const float4 h4 = *reinterpret_cast<const float4 *>(&hs[k]);
            a += wrow[k] * h4.x + wrow[k + 1] * h4.y + wrow[k + 2] * h4.z + wrow[k + 3] * h4.w;
        }
        gh[j] = a;
        __syncthreads();
        if (j < GH) {
            const float hp = hs[j];
            const float r = sigmoidf_(gxr + gh[j]);
            const float z = sigmoidf_(gxz + gh[GH + j]);
            const float hn = gh[2 * GH + j];
            const float n = tanhf(gxn + r * hn);
            const float h = (1.f - z) * n + z * hp;
            const size_t bt = (size_t)b * T + t;
            out[bt * 2 * GH + dir * GH + j] = h;
            if (gates) {
                float *gp = gates + (bt * 2 + dir) * 4 * GH;
                gp[j] = r; gp[GH + j] = z; gp[2 * GH + j] = n; gp[3 * GH + j] = hn;
                hprev[(bt * 2 + dir) * GH + j] = hp;
            }
            hs[j] = h;            // only thread j reads hs[j] after the barrier above
            gxr = nr; gxz = nz; gxn = nn;
        }
        __syncthreads();          // gh fully consumed, new hs visible
    }
}

__global__ __launch_bounds__(384) void gru_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ gates,
                                                      const float *__restrict__ hprev, const float *__restrict__ whh,
                                                      float *__restrict__ dgx, float *__restrict__ dgh, int T) {
    // thread (g, k) = (tid / 128, tid % 128) keeps column k of gate block g: W_hh[g*128 + jj][k], jj = 0..127
    __shared__ __attribute__((aligned(16))) float dg[3 * GH];   // dgh of this step
    __shared__ float part[3 * GH];
    __shared__ float dhs[GH];                                   // recurrent gradient dL/dh_prev
    const int tid = threadIdx.x, g = tid / GH, k = tid - g * GH;
    const int b = blockIdx.x, dir = blockIdx.y;
    float wcol[GH];
    const float *wsrc = whh + ((size_t)dir * 3 * GH + g * GH) * GH + k;
#pragma unroll
    for (int jj = 0; jj < GH; ++jj) wcol[jj] = wsrc[(size_t)jj * GH];
    if (tid < GH) dhs[tid] = 0.f;
    __syncthreads();
    int t = dir ? 0 : T - 1;                 // reverse of the forward order
    const int dt = dir ? 1 : -1;
    for (int step = 0; step < T; ++step, t += dt) {
        const size_t bt = (size_t)b * T + t;
        float dh_direct = 0.f;
        if (tid < GH) {
            const float *gp = gates + (bt * 2 + dir) * 4 * GH;
            const float r = gp[tid], z = gp[GH + tid], n = gp[2 * GH + tid], hn = gp[3 * GH + tid];
            const float hp = hprev[(bt * 2 + dir) * GH + tid];
            const float dh = dout[bt * 2 * GH + dir * GH + tid] + dhs[tid];
            const float dn = dh * (1.f - z);
            const float dz = dh * (hp - n);
            dh_direct = dh * z;
            const float dn_pre = dn * (1.f - n * n);
            const float dz_pre = dz * z * (1.f - z);
            const float dr_pre = dn_pre * hn * r * (1.f - r);
            float *ox = dgx + (bt * 2 + dir) * 3 * GH;
            float *oh = dgh + (bt * 2 + dir) * 3 * GH;
            ox[tid] = dr_pre; ox[GH + tid] = dz_pre; ox[2 * GH + tid] = dn_pre;
            const float dhn = dn_pre * r;
            oh[tid] = dr_pre; oh[GH + tid] = dz_pre; oh[2 * GH + tid] = dhn;
            dg[tid] = dr_pre; dg[GH + tid] = dz_pre; dg[2 * GH + tid] = dhn;
        }
        __syncthreads();
        float a = 0.f;
#pragma unroll
        for (int jj = 0; jj < GH; jj += 4) {
            const float4 d4 = *reinterpret_cast<const float4 *>(&dg[g * GH + jj]);
            a += wcol[jj] * d4.x + wcol[jj + 1] * d4.y + wcol[jj + 2] * d4.z + wcol[jj + 3] * d4.w;
        }
        part[tid] = a;
        __syncthreads();
        if (tid < GH) dhs[tid] = dh_direct + part[tid] + part[GH + tid] + part[2 * GH + tid];   // own element only
    }
}

// ---------------------------------------------------------------------------------- LayerNorm + tanh (C = 256)
template <bool TANH>
__global__ __launch_bounds__(256) void ln_tanh_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ y,
                                                          long R, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float4 v = reinterpret_cast<const float4 *>(x + (size_t)row * 256)[lane];
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.f);
    const float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
    const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.0f / 256.f);
    const float is = 1.0f / sqrtf(var + eps);
    const float4 g = reinterpret_cast<const float4 *>(gamma)[lane];
    const float4 bt = reinterpret_cast<const float4 *>(beta)[lane];
    float4 o;
    o.x = d.x * is * g.x + bt.x;
    o.y = d.y * is * g.y + bt.y;
    o.z = d.z * is * g.z + bt.z;
    o.w = d.w * is * g.w + bt.w;
    if (TANH) o = make_float4(tanhf(o.x), tanhf(o.y), tanhf(o.z), tanhf(o.w));
    reinterpret_cast<float4 *>(y + (size_t)row * 256)[lane] = o;
}

template <bool TANH>
__global__ __launch_bounds__(256) void ln_tanh_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                          const float *__restrict__ y, const float *__restrict__ gamma,
                                                          float *__restrict__ dx, float *__restrict__ partial, long R,
                                                          int rows_per_block, float eps) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 g = reinterpret_cast<const float4 *>(gamma)[lane];
    float4 dga = make_float4(0.f, 0.f, 0.f, 0.f), dba = dga;
    const long rbeg = (long)blockIdx.x * rows_per_block;
    const long rend = rbeg + rows_per_block < R ? rbeg + rows_per_block : R;
    for (long row = rbeg + wave; row < rend; row += 4) {
        const float4 v = reinterpret_cast<const float4 *>(x + (size_t)row * 256)[lane];
        const float4 go = reinterpret_cast<const float4 *>(dy + (size_t)row * 256)[lane];
        float4 yo = make_float4(0.f, 0.f, 0.f, 0.f);
        if (TANH) yo = reinterpret_cast<const float4 *>(y + (size_t)row * 256)[lane];
        const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.f);
        const float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
        const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.0f / 256.f);
        const float is = 1.0f / sqrtf(var + eps);
        const float4 xh = make_float4(d.x * is, d.y * is, d.z * is, d.w * is);
        const float4 dp = make_float4(go.x * (1.f - yo.x * yo.x), go.y * (1.f - yo.y * yo.y),
                                      go.z * (1.f - yo.z * yo.z), go.w * (1.f - yo.w * yo.w));
        dga.x += dp.x * xh.x; dga.y += dp.y * xh.y; dga.z += dp.z * xh.z; dga.w += dp.w * xh.w;
        dba.x += dp.x; dba.y += dp.y; dba.z += dp.z; dba.w += dp.w;
        const float4 dxh = make_float4(dp.x * g.x, dp.y * g.y, dp.z * g.z, dp.w * g.w);
        const float m1 = wave_sum(dxh.x + dxh.y + dxh.z + dxh.w) * (1.0f / 256.f);
        const float m2 = wave_sum(dxh.x * xh.x + dxh.y * xh.y + dxh.z * xh.z + dxh.w * xh.w) * (1.0f / 256.f);
        float4 o;
        o.x = is * (dxh.x - m1 - xh.x * m2);
        o.y = is * (dxh.y - m1 - xh.y * m2);
        o.z = is * (dxh.z - m1 - xh.z * m2);
        o.w = is * (dxh.w - m1 - xh.w * m2);
        reinterpret_cast<float4 *>(dx + (size_t)row * 256)[lane] = o;
    }
    float *p = red[wave];
    p[lane * 4 + 0] = dga.x; p[lane * 4 + 1] = dga.y; p[lane * 4 + 2] = dga.z; p[lane * 4 + 3] = dga.w;
    p[256 + lane * 4 + 0] = dba.x; p[256 + lane * 4 + 1] = dba.y; p[256 + lane * 4 + 2] = dba.z;
    p[256 + lane * 4 + 3] = dba.w;
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256)
        partial[(size_t)blockIdx.x * 512 + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}
__global__ __launch_bounds__(256) void ln_bwd_final_kernel(const float *__restrict__ partial,
                                                           float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                           int nblk) {
    __shared__ double red[256];
    const double s = block_colsum32(partial, nblk, 512, blockIdx.x * 32, 512, red);
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) == 0) {
        if (c < 256) dgamma[c] += (float)s;
        else dbeta[c - 256] += (float)s;
    }
}

// ---------------------------------------------------------------------------------------- dropout mask
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void dropout_mask_kernel(float *__restrict__ mask, long n, float p, float keep_scale, uint64_t seed,
                                    uint64_t offset) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const uint64_t h = splitmix64(splitmix64(seed) ^ (offset + (uint64_t)i));
        const float u = (float)(h >> 40) * (1.0f / 16777216.0f);     // 24 random bits -> [0,1)
        mask[i] = u >= p ? keep_scale : 0.f;
    }
}

}  // namespace adyolo

using namespace adyolo;

extern "C" int adyolo_sap_fwd(const float *x, const float *w, const float *b, float *y, float *attn, int R, int F,
                              int C, void *stream) {
    ADYOLO_REQUIRE(x && w && b && y && attn && R > 0, ADYOLO_EINVAL, "sap_fwd: bad arguments");
    ADYOLO_REQUIRE(C == 256 && (F == 16 || F == 8 || F == 4), ADYOLO_ENOSUP, "sap_fwd: needs C=256 and F in {4,8,16} (C=%d F=%d)", C, F);
    hipStream_t st = as_stream(stream);
    dim3 grid(cdiv(R, 4));
    if (F == 16) hipLaunchKernelGGL((sap_fwd_kernel<16>), grid, dim3(256), 0, st, x, w, b, y, attn, R);
    else if (F == 8) hipLaunchKernelGGL((sap_fwd_kernel<8>), grid, dim3(256), 0, st, x, w, b, y, attn, R);
    else hipLaunchKernelGGL((sap_fwd_kernel<4>), grid, dim3(256), 0, st, x, w, b, y, attn, R);
    return check_launch("sap_fwd");
}

extern "C" int adyolo_sap_bwd(const float *dy, const float *x, const float *w, const float *attn, float *dx,
                              float *dw, float *db, float *partial, int R, int F, int C, void *stream) {
    ADYOLO_REQUIRE(dy && x && w && attn && dx && dw && db && partial && R > 0, ADYOLO_EINVAL, "sap_bwd: bad arguments");
    ADYOLO_REQUIRE(C == 256 && (F == 16 || F == 8 || F == 4), ADYOLO_ENOSUP, "sap_bwd: needs C=256 and F in {4,8,16}");
    hipStream_t st = as_stream(stream);
    int nblk = cdiv(R, 4);
    if (nblk > 1024) nblk = 1024;
    const int rpb = cdiv(R, nblk);
    nblk = cdiv(R, rpb);
    if (F == 16) hipLaunchKernelGGL((sap_bwd_kernel<16>), dim3(nblk), dim3(256), 0, st, dy, x, w, attn, dx, partial, R, rpb);
    else if (F == 8) hipLaunchKernelGGL((sap_bwd_kernel<8>), dim3(nblk), dim3(256), 0, st, dy, x, w, attn, dx, partial, R, rpb);
    else hipLaunchKernelGGL((sap_bwd_kernel<4>), dim3(nblk), dim3(256), 0, st, dy, x, w, attn, dx, partial, R, rpb);
    int rc = check_launch("sap_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(sap_bwd_final_kernel, dim3(9), dim3(256), 0, st, partial, dw, db, nblk);
    return check_launch("sap_bwd_final");
}

extern "C" int adyolo_gru_fwd(const float *gx, const float *whh, const float *bhh, float *out, float *gates,
                              float *hprev, int B, int T, void *stream) {
    ADYOLO_REQUIRE(gx && whh && bhh && out && B > 0 && T > 0 && ((gates == nullptr) == (hprev == nullptr)),
                   ADYOLO_EINVAL, "gru_fwd: bad arguments");
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(B, 2), dim3(384), 0, as_stream(stream), gx, whh, bhh, out, gates, hprev, T);
    return check_launch("gru_fwd");
}
extern "C" int adyolo_gru_bwd(const float *dout, const float *gates, const float *hprev, const float *whh,
                              float *dgx, float *dgh, int B, int T, void *stream) {
    ADYOLO_REQUIRE(dout && gates && hprev && whh && dgx && dgh && B > 0 && T > 0, ADYOLO_EINVAL, "gru_bwd: bad arguments");
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(B, 2), dim3(384), 0, as_stream(stream), dout, gates, hprev, whh, dgx, dgh, T);
    return check_launch("gru_bwd");
}

extern "C" int adyolo_ln_tanh_fwd(const float *x, const float *gamma, const float *beta, float *y, long R, int C,
                                  float eps, void *stream) {
    ADYOLO_REQUIRE(x && gamma && beta && y && R > 0, ADYOLO_EINVAL, "ln_tanh_fwd: bad arguments");
    ADYOLO_REQUIRE(C == 256, ADYOLO_ENOSUP, "ln_tanh_fwd: C must be 256 (got %d)", C);
    hipLaunchKernelGGL(ln_tanh_fwd_kernel<true>, dim3(cdiv(R, 4)), dim3(256), 0, as_stream(stream), x, gamma, beta, y, R, eps);
    return check_launch("ln_tanh_fwd");
}
extern "C" int adyolo_ln_tanh_bwd(const float *dy, const float *x, const float *y, const float *gamma, float *dx,
                                  float *dgamma, float *dbeta, float *partial, long R, int C, float eps, void *stream) {
    ADYOLO_REQUIRE(dy && x && y && gamma && dx && dgamma && dbeta && partial && R > 0, ADYOLO_EINVAL, "ln_tanh_bwd: bad arguments");
    ADYOLO_REQUIRE(C == 256, ADYOLO_ENOSUP, "ln_tanh_bwd: C must be 256 (got %d)", C);
    hipStream_t st = as_stream(stream);
    int nblk = cdiv(R, 4);
    if (nblk > 1024) nblk = 1024;
    const int rpb = cdiv(R, nblk);
    nblk = cdiv(R, rpb);
    hipLaunchKernelGGL(ln_tanh_bwd_kernel<true>, dim3(nblk), dim3(256), 0, st, dy, x, y, gamma, dx, partial, R, rpb, eps);
    int rc = check_launch("ln_tanh_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(ln_bwd_final_kernel, dim3(16), dim3(256), 0, st, partial, dgamma, dbeta, nblk);
    return check_launch("ln_bwd_final");
}

// plain LayerNorm(256) (Conformer: resnet_conformer.py:160,211,236,262,290)
extern "C" int adyolo_ln_fwd(const float *x, const float *gamma, const float *beta, float *y, long R, int C, float eps,
                             void *stream) {
    ADYOLO_REQUIRE(x && gamma && beta && y && R > 0, ADYOLO_EINVAL, "ln_fwd: bad arguments");
    ADYOLO_REQUIRE(C == 256, ADYOLO_ENOSUP, "ln_fwd: C must be 256 (got %d)", C);
    hipLaunchKernelGGL(ln_tanh_fwd_kernel<false>, dim3(cdiv(R, 4)), dim3(256), 0, as_stream(stream), x, gamma, beta, y, R, eps);
    return check_launch("ln_fwd");
}
extern "C" int adyolo_ln_bwd(const float *dy, const float *x, const float *gamma, float *dx, float *dgamma,
                             float *dbeta, float *partial, long R, int C, float eps, void *stream) {
    ADYOLO_REQUIRE(dy && x && gamma && dx && dgamma && dbeta && partial && R > 0, ADYOLO_EINVAL, "ln_bwd: bad arguments");
    ADYOLO_REQUIRE(C == 256, ADYOLO_ENOSUP, "ln_bwd: C must be 256 (got %d)", C);
    hipStream_t st = as_stream(stream);
    int nblk = cdiv(R, 4);
    if (nblk > 1024) nblk = 1024;
    const int rpb = cdiv(R, nblk);
    nblk = cdiv(R, rpb);
    hipLaunchKernelGGL(ln_tanh_bwd_kernel<false>, dim3(nblk), dim3(256), 0, st, dy, x, (const float *)nullptr, gamma, dx,
                       partial, R, rpb, eps);
    int rc = check_launch("ln_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(ln_bwd_final_kernel, dim3(16), dim3(256), 0, st, partial, dgamma, dbeta, nblk);
    return check_launch("ln_bwd_final");
}

extern "C" int adyolo_dropout_mask(float *mask, long n, float p, uint64_t seed, uint64_t offset, void *stream) {
    ADYOLO_REQUIRE(mask && n > 0 && p >= 0.f && p < 1.f, ADYOLO_EINVAL, "dropout_mask: bad arguments");
    const long g = (n + 255) / 256;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, as_stream(stream), mask,
                       n, p, 1.0f / (1.0f - p), seed, offset);
    return check_launch("dropout_mask");
}
