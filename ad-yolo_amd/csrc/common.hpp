// Shared helpers for the gfx950 kernels of libadyolo_hip.so (CDNA4 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/adyolo_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace adyolo {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

#define ADYOLO_REQUIRE(cond, code, ...)      \
    do {                                     \
        if (!(cond)) {                       \
            adyolo::set_error(__VA_ARGS__);  \
            return (code);                   \
        }                                    \
    } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Workspace initialisation as a KERNEL (defined in optim.hip), not hipMemsetAsync: a memset recorded into a hipGraph becomes a
// memset node, and on this stack a replayed memset node was seen to land out of order with the kernels around it (the
// per-clip maxima of K1 were reset late in about half of the replays of one evaluation graph).  n32 = number of 32-bit words.
int fill32(void *ptr, uint32_t value, size_t n32, hipStream_t st);

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32); lane l holds A[l&31][l>>5], B[l>>5][l&31];
// D register r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ReLU mask of a tensor as bits: the float4 with global index i (4 consecutive channels) owns bit (i & 63) of the four
// 64-bit words mask[(i >> 6) * 4 + k], k = component -- one wave-wide ballot per component when a wave covers 64
// consecutive, 64-aligned float4s.  Reading the mask moves 1/32 of the bytes of reading the tensor itself.
__device__ __forceinline__ void mask_bits4(const unsigned long long *__restrict__ mask, size_t i, bool &x, bool &y,
                                           bool &z, bool &w) {
    const ulonglong2 lo = *reinterpret_cast<const ulonglong2 *>(mask + (i >> 6) * 4);
    const ulonglong2 hi = *reinterpret_cast<const ulonglong2 *>(mask + (i >> 6) * 4 + 2);
    const int b = (int)(i & 63);
    x = (lo.x >> b) & 1ull; y = (lo.y >> b) & 1ull; z = (hi.x >> b) & 1ull; w = (hi.y >> b) & 1ull;
}

// one element (flat index o of the tensor) of the same bit mask
__device__ __forceinline__ bool mask_bit1(const unsigned long long *__restrict__ mask, size_t o) {
    const size_t i = o >> 2;
    return (mask[(i >> 6) * 4 + (o & 3)] >> (i & 63)) & 1ull;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Sum `nparts` partial rows of a [nparts][ld] fp32 array for 32 consecutive columns starting at c0, in double.
// Call from a 256-thread workgroup (32 columns x 8 part-groups); returns the column total in threads with
// (threadIdx.x >> 5) == 0 (valid for c0 + (threadIdx.x & 31) < C).  `red` = 256 doubles of LDS.
__device__ __forceinline__ double block_colsum32(const float *__restrict__ partial, int nparts, size_t ld, int c0,
                                                 int C, double *red) {
    const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
    const int c = c0 + cx;
    double s = 0.0;
    if (c < C) {
        // four independent chains: the loads of a step are in flight together (a single chain is one L2 latency per row)
        double s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int p = py;
        for (; p + 24 < nparts; p += 32) {
            const float a0 = partial[(size_t)p * ld + c], a1 = partial[(size_t)(p + 8) * ld + c];
            const float a2 = partial[(size_t)(p + 16) * ld + c], a3 = partial[(size_t)(p + 24) * ld + c];
            s += (double)a0;
            s1 += (double)a1;
            s2 += (double)a2;
            s3 += (double)a3;
        }
        for (; p < nparts; p += 8) s += (double)partial[(size_t)p * ld + c];
        s = (s + s1) + (s2 + s3);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (py == 0) {
#pragma unroll
        for (int k = 1; k < 8; ++k) s += red[k * 32 + cx];
    }
    __syncthreads();
    return s;
}

// The same for TWO arrays at once (the sum and the sum of squares / the two BatchNorm-backward sums: persample_reduce_kernel):
// both arrays' loads are in flight together and the workgroup meets at two barriers instead of four.  Each total is summed in
// exactly the order block_colsum32 uses (bit-identical results).  `red` = 512 doubles of LDS.
__device__ __forceinline__ void block_colsum32x2(const float *__restrict__ pa, const float *__restrict__ pb, int nparts, size_t ld,
                                                 int c0, int C, double *red, double &ra, double &rb) {
    const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
    const int c = c0 + cx;
    double s = 0.0, t = 0.0;
    if (c < C) {
        double s1 = 0.0, s2 = 0.0, s3 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        int p = py;
        for (; p + 24 < nparts; p += 32) {
            const float a0 = pa[(size_t)p * ld + c], a1 = pa[(size_t)(p + 8) * ld + c];
            const float a2 = pa[(size_t)(p + 16) * ld + c], a3 = pa[(size_t)(p + 24) * ld + c];
            const float b0 = pb[(size_t)p * ld + c], b1 = pb[(size_t)(p + 8) * ld + c];
            const float b2 = pb[(size_t)(p + 16) * ld + c], b3 = pb[(size_t)(p + 24) * ld + c];
            s += (double)a0; s1 += (double)a1; s2 += (double)a2; s3 += (double)a3;
            t += (double)b0; t1 += (double)b1; t2 += (double)b2; t3 += (double)b3;
        }
        for (; p < nparts; p += 8) {
            s += (double)pa[(size_t)p * ld + c];
            t += (double)pb[(size_t)p * ld + c];
        }
        s = (s + s1) + (s2 + s3);
        t = (t + t1) + (t2 + t3);
    }
    red[threadIdx.x] = s;
    red[256 + threadIdx.x] = t;
    __syncthreads();
    if (py == 0) {
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            s += red[k * 32 + cx];
            t += red[256 + k * 32 + cx];
        }
    }
    __syncthreads();
    ra = s;
    rb = t;
}

}  // namespace adyolo
