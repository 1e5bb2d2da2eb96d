// Three-stream bandwidth probe (round 6): c[i] = s * a[i] + b[i] on float4 with non-temporal loads / stores -- the shape of the
// step's three big elementwise passes (bn_bwd_apply, se_tail_fwd, se_tail_bwd_apply: two tensors read, one written).  Question:
// profiles/r05_stage_rooflines.txt has those passes at 80-86 % of 8 TB/s on 0.31 GB tensors and 72-75 % on 1.26 GB ones --
// is that the size (footprint of a launch), the grid, or the workgroup -> address mapping?
//   mode 0: the product's mapping (grid <= 8192 workgroups, each iteration 4 x 256 float4 = 16 KB contiguous per tensor)
//   mode 1: the same with 2048 / 4096 / 16384 workgroups
//   mode 2: one launch per half / quarter of the tensors (the footprint of a launch shrinks)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/stream3.hip -o tools/micro/stream3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 ld_nt(const f4 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_nt(f4 *p, f4 v) { __builtin_nontemporal_store(v, p); }

template <int U, bool NT>
__global__ __launch_bounds__(256) void k3(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ c, long n4, float s) {
    const long step = (long)gridDim.x * 256 * U;
    for (long i0 = (long)blockIdx.x * 256 * U + threadIdx.x; i0 < n4; i0 += step) {
        f4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * 256;
            if (i < n4) {
                x[u] = NT ? ld_nt(a + i) : a[i];
                y[u] = NT ? ld_nt(b + i) : b[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * 256;
            if (i < n4) {
                const f4 v = x[u] * s + y[u];
                if (NT) st_nt(c + i, v); else c[i] = v;
            }
        }
    }
}

// one contiguous piece of U x 4 KB per tensor per workgroup, NO loop: the dispatcher starts workgroups in index order as slots free
// up, so the addresses in flight are a contiguous window that moves through the tensors
template <int U>
__global__ __launch_bounds__(256) void k3_once(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ c, long n4, float s) {
    const long i0 = (long)blockIdx.x * 256 * U + threadIdx.x;
    f4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = i0 + u * 256;
        if (i < n4) {
            x[u] = ld_nt(a + i);
            y[u] = ld_nt(b + i);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = i0 + u * 256;
        if (i < n4) st_nt(c + i, x[u] * s + y[u]);
    }
}
// the same piece size, but CH pieces per workgroup one after the other (contiguous 'CH x U x 4 KB' per workgroup)
template <int U>
__global__ __launch_bounds__(256) void k3_chunk(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ c, long n4, float s, int ch) {
    for (int q = 0; q < ch; ++q) {
        const long i0 = ((long)blockIdx.x * ch + q) * 256 * U + threadIdx.x;
        f4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * 256;
            if (i < n4) {
                x[u] = ld_nt(a + i);
                y[u] = ld_nt(b + i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * 256;
            if (i < n4) st_nt(c + i, x[u] * s + y[u]);
        }
    }
}

int main() {
    const long maxf = 1258291200L / 4 * 2;       // floats: up to 2.5 GB per tensor
    float *a, *b, *c;
    hipMalloc(&a, maxf * 4);
    hipMalloc(&b, maxf * 4);
    hipMalloc(&c, maxf * 4);
    hipMemset(a, 0, maxf * 4);
    hipMemset(b, 0, maxf * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](long n4, int grid, int parts, bool nt, int U) {
        auto once = [&]() {
            for (int p = 0; p < parts; ++p) {
                const long o = n4 / parts * p, m = n4 / parts;
                const f4 *pa = (const f4 *)a + o, *pb = (const f4 *)b + o;
                f4 *pc = (f4 *)c + o;
                int g = grid > 0 ? grid : (int)std::min<long>(8192, (m / U + 255) / 256);
                if (U == 4) { if (nt) hipLaunchKernelGGL((k3<4, true>), dim3(g), dim3(256), 0, 0, pa, pb, pc, m, 1.5f); else hipLaunchKernelGGL((k3<4, false>), dim3(g), dim3(256), 0, 0, pa, pb, pc, m, 1.5f); }
                else { if (nt) hipLaunchKernelGGL((k3<8, true>), dim3(g), dim3(256), 0, 0, pa, pb, pc, m, 1.5f); else hipLaunchKernelGGL((k3<8, false>), dim3(g), dim3(256), 0, 0, pa, pb, pc, m, 1.5f); }
            }
        };
        once();
        hipDeviceSynchronize();
        std::vector<float> t;
        for (int r = 0; r < 7; ++r) {
            hipEventRecord(e0);
            once();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        const double gb = 3.0 * n4 * 16 / 1e9;
        printf("tensor %.2f GB  grid %5d  launches %d  nt %d  U %d: median %.3f ms  %.0f GB/s  (%.1f %% of 8 TB/s)\n", n4 * 16 / 1e9, grid, parts,
               (int)nt, U, t[3], gb / (t[3] * 1e-3), gb / (t[3] * 1e-3) / 80.0);
        fflush(stdout);
    };
    auto run2 = [&](long n4, int U, int ch) {
        auto once = [&]() {
            const long per = 256L * U * ch;
            const int g = (int)((n4 + per - 1) / per);
            const f4 *pa = (const f4 *)a, *pb = (const f4 *)b;
            f4 *pc = (f4 *)c;
            if (ch == 1) {
                if (U == 4) hipLaunchKernelGGL((k3_once<4>), dim3(g), dim3(256), 0, 0, pa, pb, pc, n4, 1.5f);
                else if (U == 8) hipLaunchKernelGGL((k3_once<8>), dim3(g), dim3(256), 0, 0, pa, pb, pc, n4, 1.5f);
                else hipLaunchKernelGGL((k3_once<16>), dim3(g), dim3(256), 0, 0, pa, pb, pc, n4, 1.5f);
            } else {
                if (U == 4) hipLaunchKernelGGL((k3_chunk<4>), dim3(g), dim3(256), 0, 0, pa, pb, pc, n4, 1.5f, ch);
                else hipLaunchKernelGGL((k3_chunk<8>), dim3(g), dim3(256), 0, 0, pa, pb, pc, n4, 1.5f, ch);
            }
        };
        once();
        hipDeviceSynchronize();
        std::vector<float> t;
        for (int r = 0; r < 7; ++r) {
            hipEventRecord(e0);
            once();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        const double gb = 3.0 * n4 * 16 / 1e9;
        printf("tensor %.2f GB  one piece of %d x 4 KB x %d per workgroup, no stride loop: median %.3f ms  %.0f GB/s  (%.1f %% of 8 TB/s)\n", n4 * 16 / 1e9, U, ch,
               t[3], gb / (t[3] * 1e-3), gb / (t[3] * 1e-3) / 80.0);
        fflush(stdout);
    };
    for (long bytes : {314572800L, 629145600L, 1258291200L, 2516582400L}) {
        const long n4 = bytes / 16;
        run2(n4, 4, 1);
        run2(n4, 8, 1);
        run2(n4, 16, 1);
        run2(n4, 4, 4);
        run2(n4, 4, 16);
        run2(n4, 8, 8);
        run(n4, 0, 1, true, 4);
        run(n4, 0, 1, false, 4);
        run(n4, 2048, 1, true, 4);
        run(n4, 4096, 1, true, 4);
        run(n4, 16384, 1, true, 4);
        run(n4, 0, 1, true, 8);
        run(n4, 0, 2, true, 4);
        run(n4, 0, 4, true, 4);
    }
    return 0;
}
