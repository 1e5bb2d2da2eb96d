// K9a: multi-head self-attention core, flash style, exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32).
// Replaces the score / softmax / dropout / context products of MultiHeadAttention.forward
// (/root/reference/src/models/backbones/resnet_conformer.py:57-85: energy = einsum(q, k) * d^-1/2, softmax over the keys,
// dropout(p = 0.2) on the attention weights, einsum(att, v)) without ever writing the (B, heads, T, T) scores to HBM
// (92 MB per sample and layer at the evaluation length T = 2400).
//
// Layout: q, k, v, ctx are [B][T][E], head h in columns h*64 .. h*64+63 (head dimension 64).
// Forward: a workgroup (4 waves) owns 128 query rows of one (batch, head); wave w owns 32 of them.  Keys are walked in
// blocks of 32 (K and V tiles double-buffered in LDS, next tile prefetched through registers).  Per block and wave:
//   S^T (32 keys x 32 queries) = K_tile Q^T        32 MFMAs: A = K from LDS (8 x ds_read_b128, rows padded to 68 floats:
//                                                  conflict-free), B = Q held in 32 registers for the whole kernel
//                                                  (pre-scaled by d^-1/2 log2 e); the query sits on the lane
//   online softmax                                 a lane holds 16 keys of ONE query: row maximum / sum are in-register
//                                                  reductions + one exchange between the lane halves
//   O^T (64 x 32 queries) += V_tile^T P^T          32 MFMAs: the S^T accumulator registers ARE the B operand (step j
//                                                  contracts the two keys register j stands for in the two lane halves),
//                                                  A = V[that key][dv] read from LDS by the same rule
// so the probabilities never leave the registers.  The per-row log-sum-exp (base 2) is kept for the backward pass.
// Dropout: keep / drop of (b, h, query, key) is a stateless hash of its linear index and a 32-bit seed (attn_keep), the
// same function in forward, backward and adyolo_attn_dropout_mask (which materialises the mask for tests).
//
// Backward (two launches, deterministic, no atomics): P is recomputed from Q, K and the log-sum-exp.
//   attn_bwd_dkv_kernel: a workgroup owns 128 keys, walks the queries:  dV += P_d^T dO,  dK += dS^T Q
//   attn_bwd_dq_kernel : a workgroup owns 128 queries, walks the keys:  dQ += dS K
// with dP = dO V^T, dS = P (dP_d - delta) d^-1/2, delta[q] = sum_dv dO O (attn_delta_kernel), P_d = P mask / (1 - p).
#include "common.hpp"

namespace adyolo {

constexpr int AD = 64;              // head dimension
constexpr int AKS = 68;             // LDS row stride (floats): 16-lane ds_read_b128 groups along d hit 16 distinct bank quads

__device__ __forceinline__ unsigned attn_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// keep (true) / drop decision of attention weight (row = (b*H + h)*T + q, key) under dropout probability thr / 2^32
__device__ __forceinline__ bool attn_keep(unsigned row, unsigned key, unsigned T, unsigned seed, unsigned thr) {
    return attn_hash((row * T + key) * 0x9E3779B1u + seed) >= thr;
}

// key index (within a 32-key block) that accumulator register j of a 32x32 tile stands for in lane half hi
__device__ __forceinline__ int acc_row(int j, int hi) { return (j & 3) + 8 * (j >> 2) + 4 * hi; }

struct AttnTiles {                  // two double-buffered 32-row tiles (34.8 KB)
    float k[2][32 * AKS];
    float v[2][32 * AKS];
};

// ------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                          const float *__restrict__ v, float *__restrict__ ctx,
                                                          float *__restrict__ lse2, int T, int H, float scale_log2e,
                                                          unsigned seed, const unsigned *__restrict__ seed_dev, unsigned drop_thr, float keep_scale) {
    if (seed_dev) seed = *seed_dev;                       // (uniform: the seed a recorded step derives on the device, adyolo_seed32_dev)
    __shared__ __attribute__((aligned(16))) AttnTiles tl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, col = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z, E = H * AD;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const size_t base = (size_t)b * T * E + (size_t)h * AD;
    // Q fragment: Q[q0 + col][32 hi + i], pre-scaled
    float qf[32];
    {
        const int qr = min(q0 + col, T - 1);
        const float4 *src = reinterpret_cast<const float4 *>(q + base + (size_t)qr * E + 32 * hi);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 t = src[i];
            qf[4 * i] = t.x * scale_log2e; qf[4 * i + 1] = t.y * scale_log2e;
            qf[4 * i + 2] = t.z * scale_log2e; qf[4 * i + 3] = t.w * scale_log2e;
        }
    }
    f32x16 o0 = {0}, o1 = {0};
    float m_run = -INFINITY, l_run = 0.f;
    const int nkb = (T + 31) / 32;
    // tile loads: thread -> (row = tid / 8, 8 floats at column (tid % 8) * 8) of K and of V
    const int lr = tid >> 3, lc = (tid & 7) * 8;
    float4 pk0, pk1, pv0, pv1;
    auto load_tile = [&](int kb) {
        const int kr = min(kb * 32 + lr, T - 1);
        const float4 *ks = reinterpret_cast<const float4 *>(k + base + (size_t)kr * E + lc);
        const float4 *vs = reinterpret_cast<const float4 *>(v + base + (size_t)kr * E + lc);
        pk0 = ks[0]; pk1 = ks[1]; pv0 = vs[0]; pv1 = vs[1];
    };
    load_tile(0);
    const unsigned row_id = (unsigned)((b * H + h) * T + min(q0 + col, T - 1));
    for (int kb = 0; kb < nkb; ++kb) {
        float *kt = tl.k[kb & 1], *vt = tl.v[kb & 1];
        *reinterpret_cast<float4 *>(&kt[lr * AKS + lc]) = pk0;
        *reinterpret_cast<float4 *>(&kt[lr * AKS + lc + 4]) = pk1;
        *reinterpret_cast<float4 *>(&vt[lr * AKS + lc]) = pv0;
        *reinterpret_cast<float4 *>(&vt[lr * AKS + lc + 4]) = pv1;
        __syncthreads();
        if (kb + 1 < nkb) load_tile(kb + 1);
        // S^T = K Q^T
        f32x16 st = {0};
        const float4 *krow = reinterpret_cast<const float4 *>(&kt[col * AKS + 32 * hi]);
__builtin_amdgcn_s_setprio(2);      // matrix work first among the waves sharing this SIMD (-2 % per launch)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 a = krow[i];
            st = mfma32(a.x, qf[4 * i], st);
            st = mfma32(a.y, qf[4 * i + 1], st);
            st = mfma32(a.z, qf[4 * i + 2], st);
            st = mfma32(a.w, qf[4 * i + 3], st);
        }
__builtin_amdgcn_s_setprio(0);
        // online softmax (base 2): this lane's 16 keys of query `col`, the other half of the keys sits in lane ^ 32
        const int key0 = kb * 32;
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (key0 + acc_row(j, hi) >= T) st[j] = -INFINITY;
            mx = fmaxf(mx, st[j]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - m_new);          // first block: exp2(-inf) = 0
        float rs = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float p = exp2f(st[j] - m_new);
            rs += p;
            st[j] = p;
        }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            o0[j] *= alpha;
            o1[j] *= alpha;
        }
        if (drop_thr) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                st[j] = attn_keep(row_id, (unsigned)(key0 + acc_row(j, hi)), (unsigned)T, seed, drop_thr) ? st[j] * keep_scale : 0.f;
        }
        // O^T += V^T P^T: step j contracts key acc_row(j, 0) (lane half 0) and acc_row(j, 1) (lane half 1)
__builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float *vr = &vt[acc_row(j, hi) * AKS + col];
            o0 = mfma32(vr[0], st[j], o0);
            o1 = mfma32(vr[32], st[j], o1);
        }
__builtin_amdgcn_s_setprio(0);
    }
    // epilogue: ctx[q][h*64 + dv] = O^T[dv][q] / l;  register r of o<dvb> holds dv = dvb*32 + acc_row(r, hi)
    if (q0 + col < T) {
        const float inv = 1.0f / l_run;
        float *dst = ctx + base + (size_t)(q0 + col) * E;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4 *>(dst + 8 * g + 4 * hi) =
                make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *reinterpret_cast<float4 *>(dst + 32 + 8 * g + 4 * hi) =
                make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
        if (lse2 && hi == 0) lse2[(size_t)(b * H + h) * T + q0 + col] = m_run + log2f(l_run);
    }
}

// delta[b][h][q] = sum_dv dO[q][dv] * O[q][dv]   (one wave per (b, q): 4 heads x 64 columns = 256 floats per row)
__global__ __launch_bounds__(256) void attn_delta_kernel(const float *__restrict__ dctx, const float *__restrict__ ctx,
                                                         float *__restrict__ delta, long rows, int T, int H) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63, E = H * AD;
    const long b = r / T, t = r - b * T;
    for (int h = 0; h < H; ++h) {
        const float s = wave_sum(dctx[r * E + h * AD + lane] * ctx[r * E + h * AD + lane]);
        if (lane == 0) delta[(b * H + h) * T + t] = s;
    }
}

// ------------------------------------------------------------------------------------------------------ backward: dK, dV
// A workgroup owns 128 keys of one (batch, head), wave w 32 of them; queries are walked in blocks of 32 (Q and dO tiles in
// LDS).  Per block and wave, with the KEY on the lane (S = Q K^T as a 32 queries x 32 keys tile):
//   S  = Q_tile K^T    (A = Q from LDS, B = K held in registers)      P = exp2(S - lse)
//   dP = dO_tile V^T   (A = dO from LDS, B = V held in registers)     dS = P (dP_d - delta) scale
//   dV^T (64 x 32 keys) += dO^T P_d      A = dO[query][dv] from LDS by the register rule, B = P_d registers
//   dK^T (64 x 32 keys) += Q^T dS        A = Q[query][d] likewise, B = dS registers
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ dctx, const float *__restrict__ lse2, const float *__restrict__ delta,
    float *__restrict__ dk, float *__restrict__ dv, int T, int H, float scale, float scale_log2e, unsigned seed,
    const unsigned *__restrict__ seed_dev, unsigned drop_thr, float keep_scale) {
    if (seed_dev) seed = *seed_dev;
    __shared__ __attribute__((aligned(16))) AttnTiles tl;          // .k = Q tile, .v = dO tile
    __shared__ float row_lse[2][32], row_delta[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, col = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z, E = H * AD;
    const int k0 = blockIdx.x * 128 + wave * 32;
    const size_t base = (size_t)b * T * E + (size_t)h * AD;
    float kf[32], vf[32];                                   // K[k0 + col][32 hi + i] (pre-scaled by scale log2 e), V likewise (raw)
    {
        const int kr = min(k0 + col, T - 1);
        const float4 *ks = reinterpret_cast<const float4 *>(k + base + (size_t)kr * E + 32 * hi);
        const float4 *vs = reinterpret_cast<const float4 *>(v + base + (size_t)kr * E + 32 * hi);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 a = ks[i], c = vs[i];
            kf[4 * i] = a.x * scale_log2e; kf[4 * i + 1] = a.y * scale_log2e; kf[4 * i + 2] = a.z * scale_log2e; kf[4 * i + 3] = a.w * scale_log2e;
            vf[4 * i] = c.x; vf[4 * i + 1] = c.y; vf[4 * i + 2] = c.z; vf[4 * i + 3] = c.w;
        }
    }
    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    const int nqb = (T + 31) / 32;
    const int lr = tid >> 3, lc = (tid & 7) * 8;
    float4 pq0, pq1, pd0, pd1;
    float pl = 0.f, pdl = 0.f;
    auto load_tile = [&](int qb) {
        const int qr = min(qb * 32 + lr, T - 1);
        const float4 *qs = reinterpret_cast<const float4 *>(q + base + (size_t)qr * E + lc);
        const float4 *ds = reinterpret_cast<const float4 *>(dctx + base + (size_t)qr * E + lc);
        pq0 = qs[0]; pq1 = qs[1]; pd0 = ds[0]; pd1 = ds[1];
        if (tid < 32) {
            const int r = min(qb * 32 + tid, T - 1);
            pl = lse2[(size_t)(b * H + h) * T + r];
            pdl = delta[(size_t)(b * H + h) * T + r];
        }
    };
    load_tile(0);
    const unsigned key_id = (unsigned)min(k0 + col, T - 1);
    const unsigned row_base = (unsigned)((b * H + h) * T);
    for (int qb = 0; qb < nqb; ++qb) {
        float *qt = tl.k[qb & 1], *dt = tl.v[qb & 1];
        *reinterpret_cast<float4 *>(&qt[lr * AKS + lc]) = pq0;
        *reinterpret_cast<float4 *>(&qt[lr * AKS + lc + 4]) = pq1;
        *reinterpret_cast<float4 *>(&dt[lr * AKS + lc]) = pd0;
        *reinterpret_cast<float4 *>(&dt[lr * AKS + lc + 4]) = pd1;
        if (tid < 32) {
            row_lse[qb & 1][tid] = pl;
            row_delta[qb & 1][tid] = pdl;
        }
        __syncthreads();
        if (qb + 1 < nqb) load_tile(qb + 1);
        // S (queries x keys) and dP (queries x keys): key on the lane; A rows = queries from LDS
        f32x16 s = {0}, dp = {0};
        const float4 *qrow = reinterpret_cast<const float4 *>(&qt[col * AKS + 32 * hi]);
        const float4 *drow = reinterpret_cast<const float4 *>(&dt[col * AKS + 32 * hi]);
__builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 a = qrow[i], d4 = drow[i];
            s = mfma32(a.x, kf[4 * i], s);
            s = mfma32(a.y, kf[4 * i + 1], s);
            s = mfma32(a.z, kf[4 * i + 2], s);
            s = mfma32(a.w, kf[4 * i + 3], s);
            dp = mfma32(d4.x, vf[4 * i], dp);
            dp = mfma32(d4.y, vf[4 * i + 1], dp);
            dp = mfma32(d4.z, vf[4 * i + 2], dp);
            dp = mfma32(d4.w, vf[4 * i + 3], dp);
        }
__builtin_amdgcn_s_setprio(0);
        // register j: query qb*32 + acc_row(j, hi), key k0 + col
        const int q00 = qb * 32;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int qi = acc_row(j, hi);
            const bool valid = q00 + qi < T && k0 + col < T;
            float p = valid ? exp2f(s[j] - row_lse[qb & 1][qi]) : 0.f;
            float keep = 1.f;
            if (drop_thr) keep = attn_keep(row_base + (unsigned)min(q00 + qi, T - 1), key_id, (unsigned)T, seed, drop_thr) ? keep_scale : 0.f;
            const float pd = p * keep;
            const float ds = p * (dp[j] * keep - row_delta[qb & 1][qi]) * scale;
            s[j] = pd;          // P_d  -> dV
            dp[j] = ds;         // dS   -> dK
        }
        // dV^T += dO^T P_d ; dK^T += Q^T dS : step j contracts query acc_row(j, 0) / acc_row(j, 1)
__builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int qi = acc_row(j, hi);
            const float *dr = &dt[qi * AKS + col];
            const float *qr = &qt[qi * AKS + col];
            dv0 = mfma32(dr[0], s[j], dv0);
            dv1 = mfma32(dr[32], s[j], dv1);
            dk0 = mfma32(qr[0], dp[j], dk0);
            dk1 = mfma32(qr[32], dp[j], dk1);
        }
__builtin_amdgcn_s_setprio(0);
    }
    if (k0 + col < T) {
        float *dkd = dk + base + (size_t)(k0 + col) * E, *dvd = dv + base + (size_t)(k0 + col) * E;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4 *>(dkd + 8 * g + 4 * hi) = make_float4(dk0[4 * g], dk0[4 * g + 1], dk0[4 * g + 2], dk0[4 * g + 3]);
            *reinterpret_cast<float4 *>(dkd + 32 + 8 * g + 4 * hi) = make_float4(dk1[4 * g], dk1[4 * g + 1], dk1[4 * g + 2], dk1[4 * g + 3]);
            *reinterpret_cast<float4 *>(dvd + 8 * g + 4 * hi) = make_float4(dv0[4 * g], dv0[4 * g + 1], dv0[4 * g + 2], dv0[4 * g + 3]);
            *reinterpret_cast<float4 *>(dvd + 32 + 8 * g + 4 * hi) = make_float4(dv1[4 * g], dv1[4 * g + 1], dv1[4 * g + 2], dv1[4 * g + 3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------ backward: dQ
// A workgroup owns 128 queries, wave w 32 of them (query on the lane, as in the forward pass); keys are walked in blocks:
//   S^T = K_tile Q^T, dP^T = V_tile dO^T   (B = Q / dO held in registers, pre-loaded once)
//   dS^T = P^T (dP_d^T - delta) scale      dQ^T (64 x 32 queries) += K^T dS^T (A = K[key][d] by the register rule)
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ dctx, const float *__restrict__ lse2, const float *__restrict__ delta,
    float *__restrict__ dq, int T, int H, float scale, float scale_log2e, unsigned seed, const unsigned *__restrict__ seed_dev,
    unsigned drop_thr, float keep_scale) {
    if (seed_dev) seed = *seed_dev;
    __shared__ __attribute__((aligned(16))) float kt_[2][32 * AKS];
    __shared__ __attribute__((aligned(16))) float vt_[2][32 * AKS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, col = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z, E = H * AD;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const size_t base = (size_t)b * T * E + (size_t)h * AD;
    float qf[32], df[32];
    const int qr_ = min(q0 + col, T - 1);
    {
        const float4 *qs = reinterpret_cast<const float4 *>(q + base + (size_t)qr_ * E + 32 * hi);
        const float4 *ds = reinterpret_cast<const float4 *>(dctx + base + (size_t)qr_ * E + 32 * hi);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 a = qs[i], c = ds[i];
            qf[4 * i] = a.x * scale_log2e; qf[4 * i + 1] = a.y * scale_log2e; qf[4 * i + 2] = a.z * scale_log2e; qf[4 * i + 3] = a.w * scale_log2e;
            df[4 * i] = c.x; df[4 * i + 1] = c.y; df[4 * i + 2] = c.z; df[4 * i + 3] = c.w;
        }
    }
    const float my_lse = lse2[(size_t)(b * H + h) * T + qr_], my_delta = delta[(size_t)(b * H + h) * T + qr_];
    f32x16 dq0 = {0}, dq1 = {0};
    const int nkb = (T + 31) / 32;
    const int lr = tid >> 3, lc = (tid & 7) * 8;
    float4 pk0, pk1, pv0, pv1;
    auto load_tile = [&](int kb) {
        const int kr = min(kb * 32 + lr, T - 1);
        const float4 *ks = reinterpret_cast<const float4 *>(k + base + (size_t)kr * E + lc);
        const float4 *vs = reinterpret_cast<const float4 *>(v + base + (size_t)kr * E + lc);
        pk0 = ks[0]; pk1 = ks[1]; pv0 = vs[0]; pv1 = vs[1];
    };
    load_tile(0);
    const unsigned row_id = (unsigned)((b * H + h) * T + qr_);
    for (int kb = 0; kb < nkb; ++kb) {
        float *kt = kt_[kb & 1], *vt = vt_[kb & 1];
        *reinterpret_cast<float4 *>(&kt[lr * AKS + lc]) = pk0;
        *reinterpret_cast<float4 *>(&kt[lr * AKS + lc + 4]) = pk1;
        *reinterpret_cast<float4 *>(&vt[lr * AKS + lc]) = pv0;
        *reinterpret_cast<float4 *>(&vt[lr * AKS + lc + 4]) = pv1;
        __syncthreads();
        if (kb + 1 < nkb) load_tile(kb + 1);
        f32x16 st = {0}, dpt = {0};
        const float4 *krow = reinterpret_cast<const float4 *>(&kt[col * AKS + 32 * hi]);
        const float4 *vrow = reinterpret_cast<const float4 *>(&vt[col * AKS + 32 * hi]);
__builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 a = krow[i], c = vrow[i];
            st = mfma32(a.x, qf[4 * i], st);
            st = mfma32(a.y, qf[4 * i + 1], st);
            st = mfma32(a.z, qf[4 * i + 2], st);
            st = mfma32(a.w, qf[4 * i + 3], st);
            dpt = mfma32(c.x, df[4 * i], dpt);
            dpt = mfma32(c.y, df[4 * i + 1], dpt);
            dpt = mfma32(c.z, df[4 * i + 2], dpt);
            dpt = mfma32(c.w, df[4 * i + 3], dpt);
        }
__builtin_amdgcn_s_setprio(0);
        const int key0 = kb * 32;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = key0 + acc_row(j, hi);
            const float p = key < T ? exp2f(st[j] - my_lse) : 0.f;
            float keep = 1.f;
            if (drop_thr) keep = attn_keep(row_id, (unsigned)min(key, T - 1), (unsigned)T, seed, drop_thr) ? keep_scale : 0.f;
            st[j] = p * (dpt[j] * keep - my_delta) * scale;            // dS^T
        }
__builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float *kr = &kt[acc_row(j, hi) * AKS + col];
            dq0 = mfma32(kr[0], st[j], dq0);
            dq1 = mfma32(kr[32], st[j], dq1);
        }
__builtin_amdgcn_s_setprio(0);
    }
    if (q0 + col < T) {
        float *dst = dq + base + (size_t)(q0 + col) * E;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4 *>(dst + 8 * g + 4 * hi) = make_float4(dq0[4 * g], dq0[4 * g + 1], dq0[4 * g + 2], dq0[4 * g + 3]);
            *reinterpret_cast<float4 *>(dst + 32 + 8 * g + 4 * hi) = make_float4(dq1[4 * g], dq1[4 * g + 1], dq1[4 * g + 2], dq1[4 * g + 3]);
        }
    }
}

__global__ __launch_bounds__(256) void attn_dropout_mask_kernel(float *__restrict__ mask, long n, unsigned T, unsigned seed,
                                                                unsigned thr, float keep_scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned row = (unsigned)(i / T), key = (unsigned)(i - (long)row * T);
        mask[i] = attn_keep(row, key, T, seed, thr) ? keep_scale : 0.f;
    }
}

static inline unsigned drop_threshold(float p) {
    if (p <= 0.f) return 0u;
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 4294967295u : (unsigned)t;
}

}  // namespace adyolo

using namespace adyolo;

// The 32-bit seed rng.DropoutStream.seed32 computes on the host, derived on the device from (seed, offset + *offset_dev): a recorded
// step (graph.StepGraphs) replays it with the stream's current offset, and the attention kernels read it through `seed_dev`.
__global__ void seed32_dev_kernel(unsigned long long seed, unsigned long long offset, const long long *__restrict__ offset_dev,
                                  unsigned *__restrict__ out) {
    const unsigned long long off = offset + (offset_dev ? (unsigned long long)*offset_dev : 0ull);
    unsigned long long x = seed ^ (off * 0x9E3779B97F4A7C15ull);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    out[0] = (unsigned)((x ^ (x >> 31)) & 0xFFFFFFFFull);
}

extern "C" int adyolo_seed32_dev(uint64_t seed, uint64_t offset, const int64_t *offset_dev, uint32_t *out, void *stream) {
    ADYOLO_REQUIRE(out, ADYOLO_EINVAL, "seed32_dev: bad arguments");
    hipLaunchKernelGGL(seed32_dev_kernel, dim3(1), dim3(1), 0, as_stream(stream), (unsigned long long)seed, (unsigned long long)offset,
                       reinterpret_cast<const long long *>(offset_dev), out);
    return check_launch("seed32_dev");
}

extern "C" int adyolo_attn_fwd(const float *q, const float *k, const float *v, float *ctx, float *lse2, int B, int T, int H,
                               int D, float scale, float dropout_p, uint32_t seed, const uint32_t *seed_dev, void *stream) {
    ADYOLO_REQUIRE(q && k && v && ctx && B > 0 && T > 0 && H > 0, ADYOLO_EINVAL, "attn_fwd: bad arguments");
    ADYOLO_REQUIRE(D == AD && dropout_p >= 0.f && dropout_p < 1.f, ADYOLO_ENOSUP, "attn_fwd: head dimension %d (needs 64)", D);
    ADYOLO_REQUIRE((double)B * H * T * T < 4294967296.0, ADYOLO_ENOSUP, "attn_fwd: B*H*T*T exceeds the 32-bit dropout index");
    const unsigned thr = drop_threshold(dropout_p);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(T, 128), H, B), dim3(256), 0, as_stream(stream), q, k, v, ctx, lse2, T, H,
                       scale * 1.4426950408889634f, seed, seed_dev, thr, 1.0f / (1.0f - dropout_p));
    return check_launch("attn_fwd");
}

extern "C" int adyolo_attn_bwd(const float *q, const float *k, const float *v, const float *ctx, const float *dctx,
                               const float *lse2, float *delta, float *dq, float *dk, float *dv, int B, int T, int H, int D,
                               float scale, float dropout_p, uint32_t seed, const uint32_t *seed_dev, void *stream) {
    ADYOLO_REQUIRE(q && k && v && ctx && dctx && lse2 && delta && dq && dk && dv && B > 0 && T > 0 && H > 0, ADYOLO_EINVAL,
                   "attn_bwd: bad arguments");
    ADYOLO_REQUIRE(D == AD && dropout_p >= 0.f && dropout_p < 1.f, ADYOLO_ENOSUP, "attn_bwd: head dimension %d (needs 64)", D);
    hipStream_t st = as_stream(stream);
    const unsigned thr = drop_threshold(dropout_p);
    const float ks = 1.0f / (1.0f - dropout_p), sl = scale * 1.4426950408889634f;
    const long rows = (long)B * T;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, dctx, ctx, delta, rows, T, H);
    int rc = check_launch("attn_delta");
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(cdiv(T, 128), H, B), dim3(256), 0, st, q, k, v, dctx, lse2, delta, dk, dv, T,
                       H, scale, sl, seed, seed_dev, thr, ks);
    rc = check_launch("attn_bwd_dkv");
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(cdiv(T, 128), H, B), dim3(256), 0, st, q, k, v, dctx, lse2, delta, dq, T, H,
                       scale, sl, seed, seed_dev, thr, ks);
    return check_launch("attn_bwd_dq");
}

extern "C" int adyolo_attn_dropout_mask(float *mask, int B, int T, int H, float dropout_p, uint32_t seed, void *stream) {
    ADYOLO_REQUIRE(mask && B > 0 && T > 0 && H > 0 && dropout_p >= 0.f && dropout_p < 1.f, ADYOLO_EINVAL,
                   "attn_dropout_mask: bad arguments");
    const long n = (long)B * H * T * T;
    long g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), mask, n, (unsigned)T, seed,
                       drop_threshold(dropout_p), 1.0f / (1.0f - dropout_p));
    return check_launch("attn_dropout_mask");
}
