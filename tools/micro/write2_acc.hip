// ds_write2st64_b32 with AccVGPR data operands (round 6 probe): (1) does it store the AccVGPRs (ds_write_addtid_b32 does not)?
// (2) its rate for the epilogue's writer pattern -- 72 accumulator registers per wave and round, 4 waves per CU, word (s * 8 + rr) * 64
// + lane of the wave's region -- against 72 ds_write_b32 from AccVGPRs (what the compiler emits for the C++ form: it merges VGPR
// pairs into ds_write2st64_b32 but not AccVGPR pairs).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/write2_acc.hip -o tools/micro/write2_acc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int XOFF = 13280, REG = 36 * 8 * 64;

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *stamps, int rounds) {
    __shared__ __attribute__((aligned(16))) float lds[XOFF + REG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 acc[5];
    // fill the accumulators through the matrix pipe (so that they ARE AccVGPRs): D = 1 * B with B[k][j] = value -> 2 * value
#pragma unroll
    for (int s = 0; s < 5; ++s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, (float)(s * 64 + (lane & 31)), acc[s], 0, 0, 0);
        asm volatile("" : "+a"(acc[s]));
    }
    float *xwr = lds + XOFF + wave * (9 * 8 * 64) + lane;
    const unsigned xb = (unsigned)(size_t)xwr;
    unsigned long long tsum = 0;
    for (int r = 0; r < rounds; ++r) {
        __builtin_amdgcn_s_barrier();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 72; ++q) {
                asm volatile("ds_write_b32 %0, %1 offset:%c2" :: "v"(xb), "a"(acc[q / 16][q % 16]), "i"(q * 256) : "memory");
            }
        } else {
#pragma unroll
            for (int q = 0; q < 72; q += 2) {
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%c3 offset1:%c4"
                             :: "v"(xb), "a"(acc[q / 16][q % 16]), "a"(acc[(q + 1) / 16][(q + 1) % 16]), "i"(q), "i"(q + 1) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        tsum += __builtin_amdgcn_s_memtime() - t0;
    }
    if (tid == 0) stamps[blockIdx.x] = tsum / rounds;
    __syncthreads();
    for (int i = tid; i < REG; i += 256) out[(size_t)blockIdx.x * REG + i] = lds[XOFF + i];
#pragma unroll
    for (int s = 0; s < 5; ++s) asm volatile("" :: "a"(acc[s]));
}

int main() {
    const int grid = 256, rounds = 32;
    float *o0, *o1;
    unsigned long long *st;
    hipMalloc(&o0, (size_t)grid * REG * 4);
    hipMalloc(&o1, (size_t)grid * REG * 4);
    hipMalloc(&st, grid * 8);
    std::vector<unsigned long long> hs(grid);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 2; ++mode) {
            if (mode) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, o1, st, rounds);
            else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, o0, st, rounds);
            hipDeviceSynchronize();
            hipMemcpy(hs.data(), st, grid * 8, hipMemcpyDeviceToHost);
            std::sort(hs.begin(), hs.end());
            printf("%-34s: %5llu cycles per 72-register round per wave (median of %d workgroups) = %.1f B/clk/CU\n",
                   mode ? "36 x ds_write2st64_b32 from AccVGPRs" : "72 x ds_write_b32 from AccVGPRs", hs[grid / 2], grid, 4.0 * 72 * 256 / (double)hs[grid / 2]);
        }
    std::vector<float> a((size_t)grid * REG), b(a.size());
    hipMemcpy(a.data(), o0, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), o1, b.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, badref = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        bad += a[i] != b[i];
        const size_t w = i % REG;
        const int q = (int)((w % 4608) / 64), lane = (int)(w % 64);
        const float want = 2.0f * (float)((q / 16) * 64 + (lane & 31));         // every register of accumulator s holds 2 (64 s + column)
        badref += a[i] != want;
    }
    printf("write2st64 image differs from the ds_write_b32 image in %zu of %zu words; ds_write_b32 image differs from its definition in %zu  %s\n", bad, a.size(), badref,
           bad == 0 && badref == 0 ? "OK" : "FAIL");
    return bad != 0 || badref != 0;
}
