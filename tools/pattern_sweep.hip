// pattern_sweep.hip -- which block -> address mapping does HBM like?  Copy kernels whose tile
// (the unit one block owns between two table phases in k_rows_direct) is either CONTIGUOUS
// (nch chunks back to back: the round-1 layout) or STRIDED (chunk i of block b = (t*nch+i)*G + b,
// so concurrently running blocks touch neighbouring chunks, like a plain grid-stride copy).
// Build: hipcc --offload-arch=gfx950 -O3 tools/pattern_sweep.hip -o tools/pattern_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool STRIDED, bool BARRIER>
__global__ void __launch_bounds__(256) k_tile_copy(const vf4 *__restrict__ x, vf4 *__restrict__ y, long nchunks_total,
                                                   int gpc, int nch)
{
    extern __shared__ float dummy[];
    const int tid = threadIdx.x, G = gridDim.x, b = blockIdx.x;
    constexpr int U = 4, BS = 256;
    for (long t = 0;; ++t) {
        const long c0 = STRIDED ? (t * nch) * G + b : (t * G + b) * nch;
        if (c0 >= nchunks_total) break;
        const long cstep = STRIDED ? G : 1;
        long left = (nchunks_total - c0 + cstep - 1) / cstep;
        const int n_i = left < nch ? (int)left : nch;
        if (BARRIER) { __syncthreads(); if (tid == 0) dummy[0] = (float)t; __syncthreads(); }
        const int total = n_i * gpc;
        for (int g0 = tid; g0 < total; g0 += BS * U) {
            vf4 v[U];
            long off[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int g = g0 + u * BS;
                const int i = g / gpc, q = g - i * gpc;
                off[u] = (c0 + i * cstep) * gpc + q;
                if (g < total) v[u] = __builtin_nontemporal_load(x + off[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (g0 + u * BS < total) __builtin_nontemporal_store(v[u], y + off[u]);
        }
    }
}

// one 16 KiB piece per block; MAP 0: piece = blockIdx (neighbouring blocks = different XCDs = neighbouring pieces),
// MAP 1: XCD-partitioned (blocks are dealt round-robin to the 8 XCDs: XCD k streams the k-th eighth of the tensor)
template <int MAP>
__global__ void __launch_bounds__(256) k_piece_copy(const vf4 *__restrict__ x, vf4 *__restrict__ y, long npieces)
{
    const long b = blockIdx.x;
    const long per = npieces / 8;
    const long piece = MAP == 0 ? b : ((b & 7) * per + (b >> 3));
    if (piece >= npieces) return;
    const vf4 *xp = x + piece * 1024;
    vf4 *yp = y + piece * 1024;
    vf4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(xp + u * 256 + threadIdx.x);
#pragma unroll
    for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], yp + u * 256 + threadIdx.x);
}

template <typename F>
double time_ms(F launch, int iters = 10)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main()
{
    const long n = 308281344L;   // the bench tensor: 2^21 x 147 fp32
    float *x, *y; CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4));
    CK(hipMemset(x, 1, n * 4)); CK(hipMemset(y, 0, n * 4));
    const long nvec = n / 4;
    {
        const long np = (nvec / 1024) & ~7L;
        for (int rep = 0; rep < 3; ++rep) {
            double m0 = time_ms([&] { hipLaunchKernelGGL((k_piece_copy<0>), dim3(np), dim3(256), 0, 0, (const vf4 *)x, (vf4 *)y, np); });
            double m1 = time_ms([&] { hipLaunchKernelGGL((k_piece_copy<1>), dim3(np), dim3(256), 0, 0, (const vf4 *)x, (vf4 *)y, np); });
            printf("one piece per block (%ld blocks): linear %6.3f TB/s   XCD-partitioned %6.3f TB/s\n", np, 2.0 * np * 16384 / m0 / 1e9,
                   2.0 * np * 16384 / m1 / 1e9);
        }
    }
    if (getenv("PIECES_ONLY")) return 0;
    struct Cfg { int gpc, nch; };
    const Cfg cfgs[] = {{1029, 8}, {1029, 4}, {1024, 8}, {1024, 4}, {1024, 1}, {441, 9}, {735, 4}, {512, 8}, {256, 16}, {2048, 4}, {4096, 2}};
    const int ldss[] = {0, 36 * 1024};
    const int grids[] = {1024, 2048, 4096};
    for (int lds : ldss) for (int grid : grids) for (const Cfg &c : cfgs) {
        const long nct = nvec / c.gpc;
        double m1 = time_ms([&] { hipLaunchKernelGGL((k_tile_copy<false, true>), dim3(grid), dim3(256), lds, 0, (const vf4 *)x, (vf4 *)y, nct, c.gpc, c.nch); });
        double m2 = time_ms([&] { hipLaunchKernelGGL((k_tile_copy<true, true>), dim3(grid), dim3(256), lds, 0, (const vf4 *)x, (vf4 *)y, nct, c.gpc, c.nch); });
        double m3 = time_ms([&] { hipLaunchKernelGGL((k_tile_copy<true, false>), dim3(grid), dim3(256), lds, 0, (const vf4 *)x, (vf4 *)y, nct, c.gpc, c.nch); });
        const double bytes = 2.0 * nct * c.gpc * 16;
        printf("lds=%5d grid=%5d gpc=%4d nch=%2d (tile %6.1f KB)  contiguous %6.3f  strided %6.3f  strided-nobarrier %6.3f TB/s\n", lds, grid, c.gpc, c.nch,
               c.gpc * c.nch * 16 / 1024.0, bytes / m1 / 1e9, bytes / m2 / 1e9, bytes / m3 / 1e9);
    }
    return 0;
}
