// K8: AD-YOLO angular-distance responsibility-assignment loss, forward + backward, no host syncs.
// Replaces ADYOLOloss.__call__ (/root/reference/src/models/loss.py:189-251): ~40 ATen kernels, three
// CPU-resident label tensors and six .item() syncs per call become three launches:
//   1. assign : one lane per target row -> decode its cell's anchors, great-circle distance D[m][a],
//               arg-min, threshold masks; marks anchors (bit sets), counts distinct positives per
//               threshold with the value returned by atomicOr, accumulates the angular term and its
//               (un-normalised) gradient on the u,v logits (fixed-point atomics: order-independent).
//   2. main   : ONE pass over the logits, one lane per logit (coalesced): BCE terms of the three
//               thresholds and dlogit written in the same pass (HBM traffic = read logits + write dlogits
//               + 6 words per anchor of assignment state).
//   3. final  : combines the per-workgroup partial sums in double into the (1,) loss.
#include "common.hpp"

namespace adyolo {

constexpr int LOSS_HDR = 64;        // counters: [0..2] Npos_i, [3] Npairs, [4] arrival counter of the assign workgroups
constexpr int LOSS_MAIN_BLOCKS = 2048, LOSS_ASSIGN_BLOCKS = 1024;

struct LossGeom {
    int B, T, Gaz, Gel, A, C, M;
    float thr[3];
    float gain_ang, gain_obj, gain_nonobj, gain_cls;
    float grid_az, grid_el, span;          // span = 0.5 + g_overlap
};

__device__ __forceinline__ float deg2rad_(float d) { return d * 0.017453292519943295f; }
__device__ __forceinline__ float rad2deg_(float r) { return r * 57.29577951308232f; }

// 32.32 fixed point for the order-independent accumulation of the angular gradient (|value| < 2^31, step 2.3e-10)
__device__ __forceinline__ unsigned long long to_fixed(float v) {
    const double c = fmin(fmax((double)v, -2147483000.0), 2147483000.0);
    return (unsigned long long)__double2ll_rn(c * 4294967296.0);
}
__device__ __forceinline__ float from_fixed(unsigned long long q) {
    return (float)((double)(long long)q * (1.0 / 4294967296.0));
}

// 8 lanes per target row: lane a of the octet owns anchor a (one decode + great-circle distance per lane instead of a
// serial loop over the anchors: 8x the parallelism of a one-lane-per-row kernel, which was latency-bound at 137 k rows);
// the arg-min is three shuffle steps inside the octet (ties -> the lowest anchor index, like the reference's argmin);
// distinct-positive counts are kept per wave (ballot + popcount) and leave once per workgroup.
__global__ __launch_bounds__(256) void loss_assign_kernel(const float *__restrict__ logit,
                                                          const float *__restrict__ target, LossGeom g,
                                                          unsigned *__restrict__ hdr, unsigned *__restrict__ pos_bits,
                                                          unsigned *__restrict__ cls_bits,
                                                          unsigned long long *__restrict__ ang_grad,
                                                          float *__restrict__ ang_partial, float *__restrict__ dist,
                                                          long NA) {
    __shared__ float red_sum[4];
    __shared__ int red_cnt[4][4];
    __shared__ bool is_last;
    const int a = threadIdx.x & 7;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float blk_sum = 0.f;                      // this thread's share of sum(D / 180) over responsible pairs
    int cnt0 = 0, cnt1 = 0, cnt2 = 0, cntp = 0;     // wave-uniform: distinct positives per threshold, responsible pairs
    const long nlanes = (long)g.M * 8;
    for (long gt = (long)blockIdx.x * blockDim.x + threadIdx.x; gt - threadIdx.x < nlanes; gt += (long)gridDim.x * blockDim.x) {
    const int m = (int)(gt >> 3);
    float my_sum = 0.f;
    int my_pairs = 0;
    float D = INFINITY, gu = 0.f, gv = 0.f;
    long cell = 0;
    int cl = 0;
    bool valid = false;
    if (m < g.M) {
        const float *tr = target + (size_t)m * 7;
        const int b = (int)tr[0], t = (int)tr[1], gi = (int)tr[2], gj = (int)tr[3];
        cl = (int)tr[4];
        const float U = tr[5], V = tr[6];
        if (b >= 0 && b < g.B && t >= 0 && t < g.T && gi >= 0 && gi < g.Gaz && gj >= 0 && gj < g.Gel && cl >= 0 &&
            cl < g.C) {
            valid = true;
            cell = (((long)b * g.T + t) * g.Gaz + gi) * g.Gel + gj;
            if (a < g.A) {
                const int CH = g.C + 3;
                const float off_u = gi * g.grid_az - 180.f + 0.5f * g.grid_az;
                const float off_v = gj * g.grid_el - 90.f + 0.5f * g.grid_el;
                const float u2 = deg2rad_(U), v2 = deg2rad_(V);
                const float sv2 = sinf(v2), cv2 = cosf(v2);
                const float *lp = logit + ((size_t)cell * g.A + a) * CH + g.C + 1;
                const float tu = tanhf(lp[0]), tv = tanhf(lp[1]);
                float ud = tu * g.span * g.grid_az + off_u;
                const float vraw = tv * g.span * g.grid_el + off_v;
                const float vd = fminf(fmaxf(vraw, -90.f), 90.f);
                if (ud >= 180.f) ud -= 360.f;
                if (ud < -180.f) ud += 360.f;
                const float u1 = deg2rad_(ud), v1 = deg2rad_(vd);
                const float sv1 = sinf(v1), cv1 = cosf(v1);
                const float du = u1 - u2, adu = fabsf(du);
                const float cs = sv1 * sv2 + cv1 * cv2 * cosf(adu);
                const float lo = -1.f + 1e-7f, hi = 1.f - 1e-7f;
                const float cc = fminf(fmaxf(cs, lo), hi);
                D = rad2deg_(acosf(cc));
                float dD_dcs = 0.f;
                if (cs >= lo && cs <= hi) dD_dcs = -57.29577951308232f / sqrtf(1.f - cc * cc);
                const float sgn = du > 0.f ? 1.f : (du < 0.f ? -1.f : 0.f);
                const float dcs_du1 = -cv1 * cv2 * sinf(adu) * sgn;
                const float dcs_dv1 = cv1 * sv2 - sv1 * cv2 * cosf(adu);
                const float dU_dl = (1.f - tu * tu) * g.span * g.grid_az;
                const float dV_dl = (vraw >= -90.f && vraw <= 90.f) ? (1.f - tv * tv) * g.span * g.grid_el : 0.f;
                gu = dD_dcs * dcs_du1 * 0.017453292519943295f * dU_dl;
                gv = dD_dcs * dcs_dv1 * 0.017453292519943295f * dV_dl;
                if (dist) dist[(size_t)m * g.A + a] = D;
            }
        }
    }
    // first index of the minimum over the octet (all 64 lanes take part in the shuffles)
    float dmin = D;
    int amin = a;
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
        const float od = __shfl_xor(dmin, o, 64);
        const int oa = __shfl_xor(amin, o, 64);
        if (od < dmin || (od == dmin && oa < amin)) {
            dmin = od;
            amin = oa;
        }
    }
    unsigned bits = 0, fresh = 0;
    if (valid && a < g.A) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (D < g.thr[i] || a == amin) bits |= 1u << i;
    }
    if (bits) {
        const long anchor = cell * g.A + a;
        const unsigned old = atomicOr(&pos_bits[anchor], bits);
        fresh = bits & ~old;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (bits & (1u << i)) atomicOr(&cls_bits[(size_t)i * NA + anchor], 1u << cl);
        if (bits & 1u) {          // angular term uses the first threshold only (loss.py:241-243)
            my_sum = D / 180.f;
            my_pairs = 1;
            // several targets can share an anchor: their gradients are summed in 32.32 fixed point, so the
            // result does not depend on the order the atomics land in (bit-reproducible training steps)
            atomicAdd(&ang_grad[anchor * 2 + 0], to_fixed(gu / 180.f));
            atomicAdd(&ang_grad[anchor * 2 + 1], to_fixed(gv / 180.f));
        }
    }
    // distinct positives / pairs of this wave (same-address atomics are serialised by the memory system at ~12 ns each: one
    // per wave and counter cost 0.65 ms at 137 k rows -- the counts are kept in registers and leave once per workgroup)
    cnt0 += __popcll(__ballot(fresh & 1u));
    cnt1 += __popcll(__ballot((fresh >> 1) & 1u));
    cnt2 += __popcll(__ballot((fresh >> 2) & 1u));
    cntp += __popcll(__ballot(my_pairs != 0));
    blk_sum += my_sum;
    }
    blk_sum = wave_sum(blk_sum);
    if (lane == 0) {
        red_sum[wave] = blk_sum;
        red_cnt[wave][0] = cnt0;
        red_cnt[wave][1] = cnt1;
        red_cnt[wave][2] = cnt2;
        red_cnt[wave][3] = cntp;
    }
    __syncthreads();
    // per-workgroup results go to plain arrays; the LAST workgroup to arrive (one atomic per workgroup) adds the counts up
    // in a fixed order and publishes them in the header
    unsigned *cnt_partial = reinterpret_cast<unsigned *>(ang_partial + LOSS_ASSIGN_BLOCKS);
    if (threadIdx.x < 4)
        cnt_partial[blockIdx.x * 4 + threadIdx.x] = (unsigned)(red_cnt[0][threadIdx.x] + red_cnt[1][threadIdx.x] +
                                                                red_cnt[2][threadIdx.x] + red_cnt[3][threadIdx.x]);
    if (threadIdx.x == 0) ang_partial[blockIdx.x] = red_sum[0] + red_sum[1] + red_sum[2] + red_sum[3];
    __syncthreads();
    if (threadIdx.x == 0) {        // ONE device-scope fence per workgroup (a fence costs 5-20 ns per wave on this multi-XCD part;
        __threadfence();           // release is cumulative over the barrier: it also covers the stores of threads 1-3)
        is_last = atomicAdd(&hdr[4], 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (is_last) {
        __threadfence();
        unsigned t4[4] = {0u, 0u, 0u, 0u};
        for (unsigned bk = threadIdx.x; bk < gridDim.x; bk += blockDim.x) {
#pragma unroll
            for (int i = 0; i < 4; ++i)        // (device-scope loads: the other workgroups' stores, not a stale L1 line)
                t4[i] += __hip_atomic_load(cnt_partial + 4 * bk + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                  // integer sums: any order gives the same result
            unsigned v = t4[i];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) red_cnt[wave][i] = (int)v;
        }
        __syncthreads();
        if (threadIdx.x < 4)
            hdr[threadIdx.x] = (unsigned)(red_cnt[0][threadIdx.x] + red_cnt[1][threadIdx.x] + red_cnt[2][threadIdx.x] +
                                          red_cnt[3][threadIdx.x]);
    }
}

__device__ __forceinline__ float bce_grad_logit(float s, float y) {
    // nn.BCELoss backward (s - y) / max(s (1 - s), 1e-12), chained with sigmoid' = s (1 - s)
    const float q = s * (1.f - s);
    return (s - y) / fmaxf(q, 1e-12f) * q;
}

// One pass over the logits, tile by tile (LM_TILE anchors = LM_TILE * (C + 3) consecutive floats): the tile is copied into
// LDS with coalesced 16-byte loads, one lane then owns one ANCHOR (row stride odd: conflict-free) -- a negative anchor costs
// one sigmoid / log pair (its class and angle gradients are zero), a positive one the C class terms and the angular
// gradient -- and the gradients leave through the same LDS tile with coalesced 16-byte stores.
constexpr int LM_TILE = 256;
template <bool PAD>
__global__ __launch_bounds__(256) void loss_main_kernel(const float *__restrict__ logit, LossGeom g,
                                                        const unsigned *__restrict__ hdr,
                                                        const unsigned *__restrict__ pos_bits,
                                                        const unsigned *__restrict__ cls_bits,
                                                        const unsigned long long *__restrict__ ang_grad,
                                                        float *__restrict__ dlogit,
                                                        float *__restrict__ partial, long NA, long NA_total,
                                                        float grad_scale) {
    // NA_total: anchors of the whole (data-parallel) batch the header's counts refer to (== NA on one device)
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [LM_TILE][CHP]
    __shared__ float red[4][9];
    const int CH = g.C + 3;
    const int CHP = PAD ? CH + 1 : CH;              // PAD: CH is even -> odd row stride
    const int tid = threadIdx.x;
    float npos[3], nneg[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        npos[i] = (float)hdr[i];
        nneg[i] = (float)(NA_total - (long)hdr[i]);
    }
    const float npairs = (float)hdr[3];
    float wpos[3], wneg[3], wcls[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        wpos[i] = g.gain_obj / (3.f * npos[i]);
        wneg[i] = g.gain_nonobj / (3.f * nneg[i]);
        wcls[i] = g.gain_cls / (3.f * npos[i] * (float)g.C);
    }
    const float wang = g.gain_ang / npairs;
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    const long ntiles = (NA + LM_TILE - 1) / LM_TILE;
    for (long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long a0 = tl * LM_TILE;
        const int na = (int)((NA - a0) < LM_TILE ? (NA - a0) : LM_TILE);
        const int nel = na * CH, n4 = nel >> 2;
        const float *src = logit + (size_t)a0 * CH;         // LM_TILE * CH * 4 bytes is a multiple of 16
        for (int i = tid; i < n4; i += 256) {
            const float4 v = reinterpret_cast<const float4 *>(src)[i];
            if (PAD) {
                const unsigned e = 4u * i, r = e / (unsigned)CH, c = e - r * (unsigned)CH;
                float *d = tile + r * CHP + c;               // CH even and e % 4 == 0: c is even, c + 3 may cross the row end
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned ck = c + k;
                    (ck < (unsigned)CH ? d + k : tile + (r + 1) * CHP + (ck - CH))[0] = vv[k];
                }
            } else {
                reinterpret_cast<float4 *>(tile)[i] = v;
            }
        }
        for (int e = 4 * n4 + tid; e < nel; e += 256) {
            const int r = e / CH;
            tile[r * CHP + (e - r * CH)] = src[e];
        }
        __syncthreads();
        if (tid < na) {
            const long anchor = a0 + tid;
            float *x = tile + tid * CHP;
            const unsigned pb = pos_bits[anchor];
            {
                const float s = sigmoidf_(x[0]);
                const float lp = -fmaxf(logf(s), -100.f);           // -log(s), clamped like nn.BCELoss
                const float lq = -fmaxf(logf(1.f - s), -100.f);     // -log(1-s)
                const float g1 = bce_grad_logit(s, 1.f), g0 = bce_grad_logit(s, 0.f);
                float grad = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (pb & (1u << i)) {
                        acc[i] += lp;
                        grad += wpos[i] * g1;
                    } else {
                        acc[3 + i] += lq;
                        grad += wneg[i] * g0;
                    }
                }
                x[0] = grad * grad_scale;
            }
            if (pb) {
                unsigned cb[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) cb[i] = (pb & (1u << i)) ? cls_bits[(size_t)i * NA + anchor] : 0u;
                for (int ch = 1; ch <= g.C; ++ch) {
                    const float s = sigmoidf_(x[ch]);
                    const float lp = -fmaxf(logf(s), -100.f);
                    const float lq = -fmaxf(logf(1.f - s), -100.f);
                    const float g1 = bce_grad_logit(s, 1.f), g0 = bce_grad_logit(s, 0.f);
                    float grad = 0.f;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        if (pb & (1u << i)) {
                            const unsigned yb = (cb[i] >> (ch - 1)) & 1u;
                            acc[6 + i] += yb ? lp : lq;
                            grad += wcls[i] * (yb ? g1 : g0);
                        }
                    }
                    x[ch] = grad * grad_scale;
                }
                // (only anchors responsible at the first threshold ever received an angular gradient)
                const ulonglong2 ag = (pb & 1u) ? *reinterpret_cast<const ulonglong2 *>(ang_grad + anchor * 2)
                                                : make_ulonglong2(0ull, 0ull);
                x[g.C + 1] = wang * from_fixed(ag.x) * grad_scale;
                x[g.C + 2] = wang * from_fixed(ag.y) * grad_scale;
            } else {
                for (int ch = 1; ch < CH; ++ch) x[ch] = 0.f;
            }
        }
        __syncthreads();
        if (dlogit) {
            float *dst = dlogit + (size_t)a0 * CH;
            for (int i = tid; i < n4; i += 256) {
                float4 v;
                if (PAD) {
                    const unsigned e = 4u * i, r = e / (unsigned)CH, c = e - r * (unsigned)CH;
                    float vv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned ck = c + k;
                        vv[k] = (ck < (unsigned)CH ? tile + r * CHP + ck : tile + (r + 1) * CHP + (ck - CH))[0];
                    }
                    v = make_float4(vv[0], vv[1], vv[2], vv[3]);
                } else {
                    v = reinterpret_cast<const float4 *>(tile)[i];
                }
                reinterpret_cast<float4 *>(dst)[i] = v;
            }
            for (int e = 4 * n4 + tid; e < nel; e += 256) {
                const int r = e / CH;
                dst[e] = tile[r * CHP + (e - r * CH)];
            }
        }
        __syncthreads();
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (tid < 9)
        partial[(size_t)blockIdx.x * 9 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

__global__ __launch_bounds__(256) void loss_final_kernel(const float *__restrict__ partial, int nblk,
                                                         const float *__restrict__ ang_partial, int nang,
                                                         const unsigned *__restrict__ hdr, LossGeom g, long NA,
                                                         float *__restrict__ loss) {
    __shared__ double red[256];
    __shared__ double tot[10];
    const double s = block_colsum32(partial, nblk, 9, 0, 9, red);        // 9 BCE sums
    if ((threadIdx.x >> 5) == 0 && (threadIdx.x & 31) < 9) tot[threadIdx.x & 31] = s;
    double a = 0.0;
    for (int b = threadIdx.x; b < nang; b += 256) a += (double)ang_partial[b];
    red[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double ang = 0.0;
        for (int k = 0; k < 256; ++k) ang += red[k];
        double total = (double)g.gain_ang * ang / (double)hdr[3];
        for (int i = 0; i < 3; ++i) {
            const double np_ = (double)hdr[i], nn_ = (double)(NA - (long)hdr[i]);   // (NA here = the batch total)
            total += ((double)g.gain_obj * tot[i] / np_ + (double)g.gain_nonobj * tot[3 + i] / nn_ +
                      (double)g.gain_cls * tot[6 + i] / (np_ * (double)g.C)) / 3.0;
        }
        loss[0] = (float)total;
    }
}


}  // namespace adyolo

using namespace adyolo;

// workspace (32-bit words): [hdr 64][pos_bits NA][cls_bits 3*NA][ang_grad 2*NA x int64][ang_partial 1024][cnt_partial 4*1024][partial 9*2048]
extern "C" long adyolo_loss_workspace_words(int BT, int G, int A, int M) {
    const long NA = (long)BT * G * A;
    return LOSS_HDR + 8 * NA + 5L * LOSS_ASSIGN_BLOCKS + 9L * LOSS_MAIN_BLOCKS + 64;
}

// phases: 1 = workspace reset + assignment (fills the header's counts), 2 = the pass over the logits + the final sum.  One
// device runs both back to back (adyolo_loss_fwd_bwd).  Under EXACT data parallelism the header's four counts (distinct
// positives per threshold, responsible pairs) are all-reduced between the phases and phase 2 gets the anchor count of the
// whole batch (na_total): every term is then normalised like the single-device loss on the concatenated batch
// (loss.py:236-243), and the per-rank loss values add up to it.
extern "C" int adyolo_loss_phase(const float *logit, const float *target, float *ws, float *loss, float *dlogit,
                                 float *dist, int B, int T, int Gaz, int Gel, int A, int C, int M,
                                 const float *thr_host, const float *gains_host, float grid_az, float grid_el,
                                 float g_overlap, float grad_scale, int phases, long na_total, void *stream) {
    ADYOLO_REQUIRE(logit && target && ws && loss && thr_host && gains_host, ADYOLO_EINVAL, "loss: null pointer");
    ADYOLO_REQUIRE(B > 0 && T > 0 && Gaz > 0 && Gel > 0 && A > 0 && A <= 8 && C > 0 && C <= 32, ADYOLO_ENOSUP,
                   "loss: unsupported geometry A=%d (<=8) C=%d (<=32)", A, C);
    ADYOLO_REQUIRE(M > 0, ADYOLO_EINVAL, "loss: M == 0 (the reference fails on an empty target too, loss.py:224)");
    hipStream_t st = as_stream(stream);
    const long NA = (long)B * T * Gaz * Gel * A;
    ADYOLO_REQUIRE((phases & ~3) == 0 && phases != 0 && (na_total == 0 || na_total >= NA), ADYOLO_EINVAL, "loss: bad phases / na_total");
    if (na_total == 0) na_total = NA;
    LossGeom g;
    g.B = B; g.T = T; g.Gaz = Gaz; g.Gel = Gel; g.A = A; g.C = C; g.M = M;
    for (int i = 0; i < 3; ++i) g.thr[i] = thr_host[i];
    g.gain_ang = gains_host[0]; g.gain_obj = gains_host[1]; g.gain_nonobj = gains_host[2]; g.gain_cls = gains_host[3];
    g.grid_az = grid_az; g.grid_el = grid_el; g.span = 0.5f + g_overlap;

    unsigned *hdr = reinterpret_cast<unsigned *>(ws);
    unsigned *pos_bits = hdr + LOSS_HDR;
    unsigned *cls_bits = pos_bits + NA;
    unsigned long long *ang_grad = reinterpret_cast<unsigned long long *>(cls_bits + 3 * NA);   // 8-byte aligned: 64 + 4 NA words
    float *ang_partial = reinterpret_cast<float *>(ang_grad + 2 * NA);
    int nang = cdiv(M, 32);                 // 8 lanes per target row: 32 rows per workgroup and round
    if (nang > LOSS_ASSIGN_BLOCKS) nang = LOSS_ASSIGN_BLOCKS;
    float *partial = ang_partial + 5 * LOSS_ASSIGN_BLOCKS;
    int rc = 0;
    if (phases & 1) {
        rc = fill32(ws, 0u, (size_t)(LOSS_HDR + 8 * NA), st);             // (a kernel, not a memset node: see common.hpp)
        if (rc) return rc;
        hipLaunchKernelGGL(loss_assign_kernel, dim3(nang), dim3(256), 0, st, logit, target, g, hdr, pos_bits, cls_bits,
                           ang_grad, ang_partial, dist, NA);
        rc = check_launch("loss_assign");
        if (rc) return rc;
    }
    if (phases & 2) {
        long nb = (NA + LM_TILE - 1) / LM_TILE;
        if (nb > LOSS_MAIN_BLOCKS) nb = LOSS_MAIN_BLOCKS;
        const int CH = C + 3;
        if (CH & 1)
            hipLaunchKernelGGL(loss_main_kernel<false>, dim3((unsigned)nb), dim3(256), (size_t)LM_TILE * CH * 4, st, logit, g,
                               hdr, pos_bits, cls_bits, ang_grad, dlogit, partial, NA, na_total, grad_scale);
        else
            hipLaunchKernelGGL(loss_main_kernel<true>, dim3((unsigned)nb), dim3(256), (size_t)(LM_TILE + 1) * (CH + 1) * 4, st,
                               logit, g, hdr, pos_bits, cls_bits, ang_grad, dlogit, partial, NA, na_total, grad_scale);
        rc = check_launch("loss_main");
        if (rc) return rc;
        hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, partial, (int)nb, ang_partial, nang, hdr, g,
                           na_total, loss);
        rc = check_launch("loss_final");
    }
    return rc;
}

extern "C" int adyolo_loss_fwd_bwd(const float *logit, const float *target, float *ws, float *loss, float *dlogit,
                                   float *dist, int B, int T, int Gaz, int Gel, int A, int C, int M,
                                   const float *thr_host, const float *gains_host, float grid_az, float grid_el,
                                   float g_overlap, float grad_scale, void *stream) {
    return adyolo_loss_phase(logit, target, ws, loss, dlogit, dist, B, T, Gaz, Gel, A, C, M, thr_host, gains_host, grid_az,
                             grid_el, g_overlap, grad_scale, 3, 0, stream);
}

// ---- K8b: AD-YOLO decode for inference (reference datasets.py:752-771, LabelPostProcessor.get_yolo_output) ----
// out[anchor][0] = sigmoid(obj); out[anchor][1..C] = sigmoid(cls) * sigmoid(obj) (class-confidence score);
// out[anchor][C+1] = U (deg, wrapped to [-180,180)), out[anchor][C+2] = V (deg, clamped to [-90, 90 - 1e-7])
namespace adyolo {
__global__ __launch_bounds__(256) void yolo_decode_kernel(const float *__restrict__ logit, float *__restrict__ out,
                                                          long n_anchor, int Gaz, int Gel, int A, int C, float grid_az,
                                                          float grid_el, float span) {
    const int CH = C + 3;
    for (long a = (long)blockIdx.x * blockDim.x + threadIdx.x; a < n_anchor; a += (long)gridDim.x * blockDim.x) {
        const long cell = a / A;
        const int gj = (int)(cell % Gel), gi = (int)((cell / Gel) % Gaz);
        const float *lp = logit + (size_t)a * CH;
        float *op = out + (size_t)a * CH;
        const float conf = sigmoidf_(lp[0]);
        op[0] = conf;
        for (int c = 1; c <= C; ++c) op[c] = sigmoidf_(lp[c]) * conf;
        float u = tanhf(lp[C + 1]) * span * grid_az + (gi * grid_az - 180.f + 0.5f * grid_az);
        float v = tanhf(lp[C + 2]) * span * grid_el + (gj * grid_el - 90.f + 0.5f * grid_el);
        v = fminf(fmaxf(v, -90.f), 90.f - 1e-7f);
        if (u >= 180.f) u -= 360.f;
        if (u < -180.f) u += 360.f;
        op[C + 1] = u;
        op[C + 2] = v;
    }
}
}  // namespace adyolo

extern "C" int adyolo_yolo_decode(const float *logit, float *out, long n_frames, int Gaz, int Gel, int A, int C,
                                  float grid_az, float grid_el, float g_overlap, void *stream) {
    ADYOLO_REQUIRE(logit && out && n_frames > 0 && Gaz > 0 && Gel > 0 && A > 0 && C > 0, ADYOLO_EINVAL, "yolo_decode: bad arguments");
    const long n_anchor = n_frames * Gaz * Gel * A;
    long g = (n_anchor + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(adyolo::yolo_decode_kernel, dim3((unsigned)g), dim3(256), 0, adyolo::as_stream(stream), logit, out,
                       n_anchor, Gaz, Gel, A, C, grid_az, grid_el, 0.5f + g_overlap);
    return adyolo::check_launch("yolo_decode");
}
