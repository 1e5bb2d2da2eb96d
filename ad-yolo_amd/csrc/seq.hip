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

// One workgroup = one (sample, direction), 768 threads.
//  * The recurrent matrix-vector product is tiled 4 rows x 16 columns per thread (64 weights in registers): a thread reads
//    16 of the 128 state values per step (4 ds_read_b128), four independent 16-term chains, and the 8 partial sums of a row
//    are joined by three lane exchanges.
//  * NO global memory instruction sits in the per-step loop.  gfx950 has one counter (vmcnt) for loads and stores and the
//    compiler must drain it before it reuses a store's data register, so a step that stores its results waits for the
//    previous step's stores to be acknowledged: ~1 us per step, whatever the arithmetic (measured: 940-1020 ns per step for
//    1, 8, 32 or 64 samples, with 384 or 768 threads, with or without a vmcnt-free barrier).  Steps therefore run in chunks
//    of 16 on LDS rings: the chunk's inputs are fetched into registers one chunk ahead and parked in LDS, its outputs are
//    collected in LDS and flushed with coalesced stores at the chunk boundary, where one latency is paid per 16 steps.
constexpr int GTH = 768;
constexpr int GSL = 20;            // floats per 16-value slice in LDS (80 B: the 8 slices of a read hit distinct bank quads)
constexpr int GCH = 16;            // steps per chunk

// sum over the 8 lanes of an aligned lane octet, result in every lane: three DPP adds (quad xor 1, quad xor 2, mirror of the
// half row) instead of three ds_bpermute round trips through the LDS pipe
__device__ __forceinline__ float octet_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}
__device__ __forceinline__ float fsig(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.0f - 2.0f * __frcp_rn(__expf(2.0f * x) + 1.0f); }

__global__ __launch_bounds__(GTH) void gru_fwd_kernel(const float *__restrict__ gx, const float *__restrict__ whh,
                                                      const float *__restrict__ bhh, float *__restrict__ out,
                                                      float *__restrict__ gates, float *__restrict__ hprev, int T) {
    __shared__ __attribute__((aligned(16))) float hs[8 * GSL];      // h[k] at (k >> 4) * GSL + (k & 15)
    __shared__ __attribute__((aligned(16))) float gh[3 * GH];
    __shared__ float xring[GCH][3 * GH];                            // input projections of the chunk: [r | z | n] per step
    __shared__ float oring[GCH][6 * GH];                            // results of the chunk: [h | r | z | n | hn | hprev] per step
    const int tid = threadIdx.x;
    const int rg = tid >> 3, ks = tid & 7;       // rows 4 rg .. 4 rg + 3, columns 16 ks .. 16 ks + 15
    const int b = blockIdx.x, dir = blockIdx.y;
    float w[4][16];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float *wsrc = whh + ((size_t)dir * 3 * GH + 4 * rg + r) * GH + 16 * ks;
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(wsrc + k);
            w[r][k] = v.x; w[r][k + 1] = v.y; w[r][k + 2] = v.z; w[r][k + 3] = v.w;
        }
    }
    const float4 bj = *reinterpret_cast<const float4 *>(bhh + dir * 3 * GH + 4 * rg);
    if (tid < 8 * GSL) hs[tid] = 0.f;
    const size_t gx_b = (size_t)b * T * 2 * 3 * GH;
    // chunk input: element e = i * 768 + tid (i < 8) -> step e / 384 of the chunk, column e % 384
    float xin[GCH / 2];
    auto fetch_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < GCH / 2; ++i) {
            const int e = i * GTH + tid, sidx = e / (3 * GH), col = e - sidx * (3 * GH);
            const int step = min(c * GCH + sidx, T - 1);
            const int tt = dir ? T - 1 - step : step;
            xin[i] = gx[gx_b + ((size_t)tt * 2 + dir) * 3 * GH + col];
        }
    };
    auto park_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < GCH / 2; ++i) {
            const int e = i * GTH + tid, sidx = e / (3 * GH), col = e - sidx * (3 * GH);
            xring[sidx][col] = xin[i];
        }
    };
    fetch_chunk(0);
    park_chunk();
    __syncthreads();
    const float4 *hh = reinterpret_cast<const float4 *>(hs + ks * GSL);
    const int hidx = (tid >> 4) * GSL + (tid & 15);      // where state element `tid` lives (tid < 128)
    const int nchunks = (T + GCH - 1) / GCH;
    for (int c = 0; c < nchunks; ++c) {
        const int len = min(GCH, T - c * GCH);
        if (c + 1 < nchunks) fetch_chunk(c + 1);          // lands during the 16 steps below
        for (int sidx = 0; sidx < len; ++sidx) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 h4 = hh[q];
                a0 = fmaf(w[0][4 * q + 3], h4.w, fmaf(w[0][4 * q + 2], h4.z, fmaf(w[0][4 * q + 1], h4.y, fmaf(w[0][4 * q], h4.x, a0))));
                a1 = fmaf(w[1][4 * q + 3], h4.w, fmaf(w[1][4 * q + 2], h4.z, fmaf(w[1][4 * q + 1], h4.y, fmaf(w[1][4 * q], h4.x, a1))));
                a2 = fmaf(w[2][4 * q + 3], h4.w, fmaf(w[2][4 * q + 2], h4.z, fmaf(w[2][4 * q + 1], h4.y, fmaf(w[2][4 * q], h4.x, a2))));
                a3 = fmaf(w[3][4 * q + 3], h4.w, fmaf(w[3][4 * q + 2], h4.z, fmaf(w[3][4 * q + 1], h4.y, fmaf(w[3][4 * q], h4.x, a3))));
            }
            a0 = octet_sum(a0);
            a1 = octet_sum(a1);
            a2 = octet_sum(a2);
            a3 = octet_sum(a3);
            if (ks == 0) *reinterpret_cast<float4 *>(&gh[4 * rg]) = make_float4(a0 + bj.x, a1 + bj.y, a2 + bj.z, a3 + bj.w);
            __syncthreads();
            if (tid < GH) {
                const float hp = hs[hidx];
                const float r = fsig(xring[sidx][tid] + gh[tid]);
                const float z = fsig(xring[sidx][GH + tid] + gh[GH + tid]);
                const float hn = gh[2 * GH + tid];
                const float n = ftanh(xring[sidx][2 * GH + tid] + r * hn);
                const float h = (1.f - z) * n + z * hp;
                float *o = oring[sidx];
                o[tid] = h; o[GH + tid] = r; o[2 * GH + tid] = z; o[3 * GH + tid] = n; o[4 * GH + tid] = hn; o[5 * GH + tid] = hp;
                hs[hidx] = h;         // only this thread reads its element after the barrier above
            }
            __syncthreads();          // gh fully consumed, new hs visible
        }
        // chunk boundary: flush the results (thread f owns element f of every step), park the next chunk's inputs
        for (int sidx = 0; sidx < len; ++sidx) {
            const int step = c * GCH + sidx;
            const size_t bt = (size_t)b * T + (dir ? T - 1 - step : step);
            const float v = oring[sidx][tid];
            if (tid < GH) out[bt * 2 * GH + dir * GH + tid] = v;
            else if (gates) {
                if (tid < 5 * GH) gates[(bt * 2 + dir) * 4 * GH + (tid - GH)] = v;
                else hprev[(bt * 2 + dir) * GH + (tid - 5 * GH)] = v;
            }
        }
        if (c + 1 < nchunks) park_chunk();
        __syncthreads();
    }
}

// Backward: dh_prev[k] = sum_j W_hh[j][k] dgh[j] (+ the direct path).  Thread (js, kg) = (tid / 32, tid % 32) owns the 16 rows
// j = 16 js .. +15 of the 4 columns k = 4 kg .. +3 (64 weights in registers); lanes 0-31 of a wave share js, so the reads of
// the gate-gradient slice are broadcasts; the 24 partial sums of a column go through LDS.  Chunked like the forward pass.
__global__ __launch_bounds__(GTH) void gru_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ gates,
                                                      const float *__restrict__ hprev, const float *__restrict__ whh,
                                                      float *__restrict__ dgx, float *__restrict__ dgh, int T) {
    __shared__ __attribute__((aligned(16))) float dg[3 * GH];   // dgh of this step
    __shared__ __attribute__((aligned(16))) float part[24 * GH];
    __shared__ float dhs[GH];                                   // recurrent gradient dL/dh_prev
    __shared__ float iring[GCH][6 * GH];                        // inputs of the chunk: [r | z | n | hn | hprev | dout] per step
    __shared__ float oring[GCH][6 * GH];                        // results: [dgx r z n | dgh r z n] per step
    const int tid = threadIdx.x, js = tid >> 5, kg = tid & 31;
    const int b = blockIdx.x, dir = blockIdx.y;
    float w[16][4];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const float4 v = *reinterpret_cast<const float4 *>(whh + ((size_t)dir * 3 * GH + 16 * js + jj) * GH + 4 * kg);
        w[jj][0] = v.x; w[jj][1] = v.y; w[jj][2] = v.z; w[jj][3] = v.w;
    }
    if (tid < GH) dhs[tid] = 0.f;
    // step -> time: the reverse of the forward order
    float xin[GCH];
    auto fetch_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < GCH; ++i) {          // thread f owns element f of step i of the chunk
            const int step = min(c * GCH + i, T - 1);
            const size_t bt = (size_t)b * T + (dir ? step : T - 1 - step);
            float v;
            if (tid < 4 * GH) v = gates[(bt * 2 + dir) * 4 * GH + tid];
            else if (tid < 5 * GH) v = hprev[(bt * 2 + dir) * GH + (tid - 4 * GH)];
            else v = dout[bt * 2 * GH + dir * GH + (tid - 5 * GH)];
            xin[i] = v;
        }
    };
    auto park_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < GCH; ++i) iring[i][tid] = xin[i];
    };
    fetch_chunk(0);
    park_chunk();
    __syncthreads();
    const float4 *dgs = reinterpret_cast<const float4 *>(dg + 16 * js);
    const int nchunks = (T + GCH - 1) / GCH;
    for (int c = 0; c < nchunks; ++c) {
        const int len = min(GCH, T - c * GCH);
        if (c + 1 < nchunks) fetch_chunk(c + 1);
        for (int sidx = 0; sidx < len; ++sidx) {
            float dh_direct = 0.f;
            if (tid < GH) {
                const float *in = iring[sidx];
                const float r = in[tid], z = in[GH + tid], n = in[2 * GH + tid], hn = in[3 * GH + tid], hp = in[4 * GH + tid];
                const float dh = in[5 * GH + tid] + dhs[tid];
                const float dn = dh * (1.f - z);
                const float dz = dh * (hp - n);
                dh_direct = dh * z;
                const float dn_pre = dn * (1.f - n * n);
                const float dz_pre = dz * z * (1.f - z);
                const float dr_pre = dn_pre * hn * r * (1.f - r);
                const float dhn = dn_pre * r;
                float *o = oring[sidx];
                o[tid] = dr_pre; o[GH + tid] = dz_pre; o[2 * GH + tid] = dn_pre;
                o[3 * GH + tid] = dr_pre; o[4 * GH + tid] = dz_pre; o[5 * GH + tid] = dhn;
                dg[tid] = dr_pre; dg[GH + tid] = dz_pre; dg[2 * GH + tid] = dhn;
            }
            __syncthreads();
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 d4 = dgs[q];
                a0 = fmaf(w[4 * q + 3][0], d4.w, fmaf(w[4 * q + 2][0], d4.z, fmaf(w[4 * q + 1][0], d4.y, fmaf(w[4 * q][0], d4.x, a0))));
                a1 = fmaf(w[4 * q + 3][1], d4.w, fmaf(w[4 * q + 2][1], d4.z, fmaf(w[4 * q + 1][1], d4.y, fmaf(w[4 * q][1], d4.x, a1))));
                a2 = fmaf(w[4 * q + 3][2], d4.w, fmaf(w[4 * q + 2][2], d4.z, fmaf(w[4 * q + 1][2], d4.y, fmaf(w[4 * q][2], d4.x, a2))));
                a3 = fmaf(w[4 * q + 3][3], d4.w, fmaf(w[4 * q + 2][3], d4.z, fmaf(w[4 * q + 1][3], d4.y, fmaf(w[4 * q][3], d4.x, a3))));
            }
            *reinterpret_cast<float4 *>(&part[js * GH + 4 * kg]) = make_float4(a0, a1, a2, a3);
            __syncthreads();
            if (tid < GH) {           // own element only; fixed order
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    s0 += part[i * GH + tid];
                    s1 += part[(8 + i) * GH + tid];
                    s2 += part[(16 + i) * GH + tid];
                }
                dhs[tid] = dh_direct + (s0 + s1 + s2);
            }
        }
        __syncthreads();              // the chunk's last results are in the ring
        for (int sidx = 0; sidx < len; ++sidx) {
            const int step = c * GCH + sidx;
            const size_t bt = (size_t)b * T + (dir ? step : T - 1 - step);
            const float v = oring[sidx][tid];
            if (tid < 3 * GH) dgx[(bt * 2 + dir) * 3 * GH + tid] = v;
            else dgh[(bt * 2 + dir) * 3 * GH + (tid - 3 * GH)] = v;
        }
        if (c + 1 < nchunks) park_chunk();
        __syncthreads();
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

// y = x * mask with the SAME mask values dropout_mask_kernel writes for (seed, offset) -- generated on the fly, so neither
// the forward nor the backward pass (dx = dy * mask: the same call on dy) moves a mask tensor
// offset_dev (may be null): a device-side running offset added to `offset` -- a launch recorded in a hipGraph then draws a
// fresh part of the stream at every replay (the host advances the counter with adyolo_counter_add at the end of a step)
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float4 *__restrict__ x, float4 *__restrict__ y, long n4,
                                                            float p, float keep_scale, uint64_t seed, uint64_t offset,
                                                            const uint64_t *__restrict__ offset_dev) {
    if (offset_dev) offset += *offset_dev;
    const uint64_t s0 = splitmix64(seed);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        const uint64_t b = offset + (uint64_t)i * 4;
        const float u0 = (float)(splitmix64(s0 ^ b) >> 40) * (1.0f / 16777216.0f);
        const float u1 = (float)(splitmix64(s0 ^ (b + 1)) >> 40) * (1.0f / 16777216.0f);
        const float u2 = (float)(splitmix64(s0 ^ (b + 2)) >> 40) * (1.0f / 16777216.0f);
        const float u3 = (float)(splitmix64(s0 ^ (b + 3)) >> 40) * (1.0f / 16777216.0f);
        y[i] = make_float4(v.x * (u0 >= p ? keep_scale : 0.f), v.y * (u1 >= p ? keep_scale : 0.f),
                           v.z * (u2 >= p ? keep_scale : 0.f), v.w * (u3 >= p ? keep_scale : 0.f));
    }
}

// out = a * dropout(x) + b * z with the mask of dropout_apply_kernel (z may be null: out = a * dropout(x), the gradient form):
// the residual mix of a Conformer sub-module whose last operation is a Dropout (resnet_conformer.py:98 on top of :209, :280)
// in one pass instead of two.  The rounding sequence is that of dropout_apply followed by axpby.
__global__ __launch_bounds__(256) void dropout_axpby_kernel(const float4 *__restrict__ x, const float4 *__restrict__ z,
                                                            float4 *__restrict__ y, long n4, float p, float keep_scale,
                                                            uint64_t seed, uint64_t offset,
                                                            const uint64_t *__restrict__ offset_dev, float a, float b) {
    if (offset_dev) offset += *offset_dev;
    const uint64_t s0 = splitmix64(seed);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        const uint64_t q = offset + (uint64_t)i * 4;
        const float u0 = (float)(splitmix64(s0 ^ q) >> 40) * (1.0f / 16777216.0f);
        const float u1 = (float)(splitmix64(s0 ^ (q + 1)) >> 40) * (1.0f / 16777216.0f);
        const float u2 = (float)(splitmix64(s0 ^ (q + 2)) >> 40) * (1.0f / 16777216.0f);
        const float u3 = (float)(splitmix64(s0 ^ (q + 3)) >> 40) * (1.0f / 16777216.0f);
        const float d0 = v.x * (u0 >= p ? keep_scale : 0.f), d1 = v.y * (u1 >= p ? keep_scale : 0.f);
        const float d2 = v.z * (u2 >= p ? keep_scale : 0.f), d3 = v.w * (u3 >= p ? keep_scale : 0.f);
        if (z) {
            const float4 w = z[i];
            y[i] = make_float4(a * d0 + b * w.x, a * d1 + b * w.y, a * d2 + b * w.z, a * d3 + b * w.w);
        } else {
            y[i] = make_float4(a * d0, a * d1, a * d2, a * d3);
        }
    }
}

}  // namespace adyolo

using namespace adyolo;

extern "C" int adyolo_dropout_axpby(const float *x, const float *z, float *y, long n, float p, uint64_t seed, uint64_t offset,
                                    const uint64_t *offset_dev, float a, float b, void *stream) {
    ADYOLO_REQUIRE(x && y && n > 0 && n % 4 == 0 && p >= 0.f && p < 1.f, ADYOLO_EINVAL,
                   "dropout_axpby: n must be a positive multiple of 4, 0 <= p < 1");
    const long n4 = n / 4, g = (n4 + 255) / 256;
    hipLaunchKernelGGL(dropout_axpby_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(z), reinterpret_cast<float4 *>(y),
                       n4, p, 1.0f / (1.0f - p), seed, offset, offset_dev, a, b);
    return check_launch("dropout_axpby");
}

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
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(B, 2), dim3(GTH), 0, as_stream(stream), gx, whh, bhh, out, gates, hprev, T);
    return check_launch("gru_fwd");
}
extern "C" int adyolo_gru_bwd(const float *dout, const float *gates, const float *hprev, const float *whh,
                              float *dgx, float *dgh, int B, int T, void *stream) {
    ADYOLO_REQUIRE(dout && gates && hprev && whh && dgx && dgh && B > 0 && T > 0, ADYOLO_EINVAL, "gru_bwd: bad arguments");
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(B, 2), dim3(GTH), 0, as_stream(stream), dout, gates, hprev, whh, dgx, dgh, T);
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

extern "C" int adyolo_dropout_apply_dev(const float *x, float *y, long n, float p, uint64_t seed, uint64_t offset,
                                        const uint64_t *offset_dev, void *stream) {
    ADYOLO_REQUIRE(x && y && n > 0 && n % 4 == 0 && p >= 0.f && p < 1.f, ADYOLO_EINVAL,
                   "dropout_apply: n must be a positive multiple of 4, 0 <= p < 1");
    const long n4 = n / 4, g = (n4 + 255) / 256;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(y), n4, p, 1.0f / (1.0f - p), seed,
                       offset, offset_dev);
    return check_launch("dropout_apply");
}

extern "C" int adyolo_dropout_apply(const float *x, float *y, long n, float p, uint64_t seed, uint64_t offset, void *stream) {
    return adyolo_dropout_apply_dev(x, y, n, p, seed, offset, nullptr, stream);
}

extern "C" int adyolo_dropout_mask(float *mask, long n, float p, uint64_t seed, uint64_t offset, void *stream) {
    ADYOLO_REQUIRE(mask && n > 0 && p >= 0.f && p < 1.f, ADYOLO_EINVAL, "dropout_mask: bad arguments");
    const long g = (n + 255) / 256;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, as_stream(stream), mask,
                       n, p, 1.0f / (1.0f - p), seed, offset);
    return check_launch("dropout_mask");
}
