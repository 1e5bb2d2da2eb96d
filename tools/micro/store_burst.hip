// Store-burst probe for gfx950 (round 6): is the ~10 B / clock / CU at which the persistent F(4x4) kernel's epilogue stores drain
// ("Forward kernel: the epilogue is bounded by its stores") a limit of ONE CU's store path or the chip's write
// bandwidth shared by 256 CUs that all store at the same time (the workgroups of a persistent launch start together and walk
// patches of equal cost: their epilogues coincide)?
// One 256-thread workgroup per CU (the LDS request keeps a second one away) stores `rounds` bursts of 32 KB -- the persistent
// kernel's epilogue round: 8 x buffer_store_dwordx4 per thread, 16-byte pieces 4 pixels x 256 B apart like the kernel's -- with
// `gap` cycles of s_sleep between bursts (the pair loop stands in as idle time), on grids of 256 / 128 / 64 / 32 / 8 workgroups.
// Reported per grid: cycles per burst (issue of the 8 stores + s_waitcnt vmcnt(0)) and bytes per clock per CU, median over
// workgroups and bursts; wall time per launch next to it.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_burst.hip -o tools/micro/store_burst
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef unsigned int u32x4_t __attribute__((__vector_size__(16)));

template <int AUX>
__global__ __launch_bounds__(256, 1) void burst_kernel(float *out, unsigned long long *stamps, int rounds, int gap, size_t wg_bytes,
                                                       int stagger) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    if (tid == 0) lds[0] = 0.f;
    char *base = reinterpret_cast<char *>(out) + (size_t)blockIdx.x * wg_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)wg_bytes, 0x00020000);
    u32x4_t v = {(unsigned)tid, 1u, 2u, 3u};
    // lane -> (row of 8 pixels, channel quad): 16 consecutive lanes cover one 256-byte pixel, a wave 4 pixels; the 8 stores of a
    // thread go to 8 different pixel rows 16 KB apart in the workgroup's slab (a patch's rows are W * Cout * 4 bytes apart)
    const int off0 = (tid >> 4) * 256 + (tid & 15) * 16;
    if (stagger && (blockIdx.x & 8)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)stagger) __builtin_amdgcn_s_sleep(32);
    }
    for (int r = 0; r < rounds; ++r) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const int ro = (r & 15) * (32 * 1024);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, off0 + k * 4096 + ro, 0, AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (tid == 0) stamps[(size_t)blockIdx.x * rounds + r] = t1 - t0;
        // idle gap (the pair loop): s_sleep in 64-cycle units
        const unsigned long long g0 = __builtin_amdgcn_s_memtime();
        while ((int)(__builtin_amdgcn_s_memtime() - g0) < gap) __builtin_amdgcn_s_sleep(16);
        v[1] += 1u;
    }
}

int main(int argc, char **argv) {
    const int rounds = 64;
    const size_t wg_bytes = 16 * 32 * 1024;              // 512 KB slab per workgroup, rewritten every 16 bursts
    float *out;
    unsigned long long *stamps, *host;
    hipMalloc(&out, 256 * wg_bytes);
    hipMalloc(&stamps, 256 * rounds * sizeof(unsigned long long));
    host = (unsigned long long *)malloc(256 * rounds * sizeof(unsigned long long));
    hipEvent_t s, e;
    hipEventCreate(&s);
    hipEventCreate(&e);
    const int lds_bytes = 100 * 1024;
    hipFuncSetAttribute((const void *)burst_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipFuncSetAttribute((const void *)burst_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    printf("burst = 32 KB per workgroup (8 x dwordx4 per thread), %d bursts per workgroup; cycles = issue + s_waitcnt vmcnt(0)\n", rounds);
    for (int aux = 0; aux <= 2; aux += 2)
        for (int gap : {0, 3000, 12000})
            for (int stagger : {0, 1})
                for (int grid : {256, 128, 64, 32, 8}) {
                    if (stagger && (grid != 256 || gap == 0)) continue;
                    const int st = stagger ? (gap + 3300) / 2 : 0;
                    auto launch = [&]() {
                        if (aux) hipLaunchKernelGGL(burst_kernel<2>, dim3(grid), dim3(256), lds_bytes, 0, out, stamps, rounds, gap, wg_bytes, st);
                        else hipLaunchKernelGGL(burst_kernel<0>, dim3(grid), dim3(256), lds_bytes, 0, out, stamps, rounds, gap, wg_bytes, st);
                    };
                    launch();
                    hipDeviceSynchronize();
                    hipEventRecord(s);
                    launch();
                    hipEventRecord(e);
                    hipEventSynchronize(e);
                    float ms;
                    hipEventElapsedTime(&ms, s, e);
                    hipMemcpy(host, stamps, (size_t)grid * rounds * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                    std::vector<double> v;
                    for (int b = 0; b < grid; ++b)
                        for (int r = 8; r < rounds; ++r) v.push_back((double)host[(size_t)b * rounds + r]);
                    std::sort(v.begin(), v.end());
                    const double med = v[v.size() / 2], p10 = v[v.size() / 10], p90 = v[v.size() * 9 / 10];
                    printf("aux %d gap %5d stagger %5d grid %3d: burst %6.0f cycles median (p10 %6.0f, p90 %6.0f) = %5.1f B/clk/CU; launch %.3f ms "
                           "= %.2f TB/s over the launch\n", aux, gap, st, grid, med, p10, p90, 32768.0 / med, ms,
                           (double)grid * rounds * 32768.0 / (ms * 1e-3) / 1e12);
                    fflush(stdout);
                }
    return 0;
}
