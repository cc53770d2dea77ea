// stream_sweep.hip -- standalone sweep of streaming-kernel launch shapes on one MI355X.
// Build: hipcc --offload-arch=gfx950 -O3 tools/stream_sweep.hip -o tools/stream_sweep
// Finds the copy (read+write) and read-only ceilings that K1 / K3 are measured against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float vf4 __attribute__((ext_vector_type(4)));
#define float4 vf4
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int U, int NT>
__global__ void __launch_bounds__(1024) k_copy(const float4 *__restrict__ x, float4 *__restrict__ y, long nvec)
{
    const int tid = threadIdx.x, B = blockDim.x;
    const long step = (long)gridDim.x * (B * U);
    for (long base = (long)blockIdx.x * (B * U); base < nvec; base += step) {
        float4 v[U];
        if (base + B * U <= nvec) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (NT & 1) v[u] = __builtin_nontemporal_load(x + base + u * B + tid);
                else v[u] = x[base + u * B + tid];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (NT & 2) __builtin_nontemporal_store(v[u], y + base + u * B + tid);
                else y[base + u * B + tid] = v[u];
            }
        } else {
            for (int u = 0; u < U; ++u) { long i = base + u * B + tid; if (i < nvec) y[i] = x[i]; }
        }
    }
}

// blocked variant: each block owns one contiguous slab (nvec / grid), walks it
template <int U, int NT>
__global__ void __launch_bounds__(1024) k_copy_slab(const float4 *__restrict__ x, float4 *__restrict__ y, long nvec)
{
    const int tid = threadIdx.x, B = blockDim.x;
    const long per = (nvec + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per;
    long hi = lo + per; if (hi > nvec) hi = nvec;
    for (long base = lo; base < hi; base += B * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long i = base + u * B + tid;
            if (i < hi) {
                float4 v;
                if (NT & 1) v = __builtin_nontemporal_load(x + i); else v = x[i];
                if (NT & 2) __builtin_nontemporal_store(v, y + i); else y[i] = v;
            }
        }
    }
}

template <int U, int NT>
__global__ void __launch_bounds__(1024) k_read(const float4 *__restrict__ x, float *__restrict__ out, long nvec)
{
    const int tid = threadIdx.x, B = blockDim.x;
    const long step = (long)gridDim.x * (B * U);
    float acc = 0.f;
    for (long base = (long)blockIdx.x * (B * U); base < nvec; base += step) {
        if (base + B * U <= nvec) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (NT & 1) v[u] = __builtin_nontemporal_load(x + base + u * B + tid);
                else v[u] = x[base + u * B + tid];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc = fmaxf(acc, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
        }
    }
    if (acc == 123456.f) out[0] = acc;
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
    const long n = 1L << 28;  // 1 GiB of fp32 in, 1 GiB out
    float *x, *y; CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4));
    CK(hipMemset(x, 1, n * 4)); CK(hipMemset(y, 0, n * 4));
    const long nvec = n / 4;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz memclk %d kHz bus %d\n", p.name, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.memoryBusWidth);
    {
        double ms = time_ms([&] { CK(hipMemcpyAsync(y, x, n * 4, hipMemcpyDeviceToDevice, 0)); });
        printf("hipMemcpyDtoD                              %8.1f us %6.3f TB/s\n", ms * 1e3, 2.0 * n * 4 / ms / 1e9);
    }
    const int blocksizes[] = {256, 512, 1024};
    const int bpc[] = {2, 4, 8, 16, 32};
#define RUNCOPY(U, NT, NAME, KERN) \
    for (int bs : blocksizes) for (int b : bpc) { \
        long grid = (long)256 * b * 256 / bs; if (grid < 256) continue; \
        long need = (nvec + (long)bs * U - 1) / ((long)bs * U); if (grid > need) grid = need; \
        double ms = time_ms([&] { hipLaunchKernelGGL((KERN<U, NT>), dim3(grid), dim3(bs), 0, 0, (const float4 *)x, (float4 *)y, nvec); }); \
        printf("%-10s U=%d NT=%d bs=%4d grid=%6ld  %8.1f us %6.3f TB/s\n", NAME, U, NT, bs, grid, ms * 1e3, 2.0 * n * 4 / ms / 1e9); }
    RUNCOPY(1, 0, "copy", k_copy) RUNCOPY(2, 0, "copy", k_copy) RUNCOPY(4, 0, "copy", k_copy) RUNCOPY(8, 0, "copy", k_copy)
    RUNCOPY(4, 1, "copy", k_copy) RUNCOPY(4, 2, "copy", k_copy) RUNCOPY(4, 3, "copy", k_copy) RUNCOPY(8, 3, "copy", k_copy) RUNCOPY(2, 3, "copy", k_copy)
    RUNCOPY(4, 0, "slab", k_copy_slab) RUNCOPY(4, 3, "slab", k_copy_slab)
    {   // one tile per block, no loop (grid = nvec / (bs*U))
        for (int bs : blocksizes) {
            long grid = nvec / ((long)bs * 4);
            double ms = time_ms([&] { hipLaunchKernelGGL((k_copy<4, 0>), dim3(grid), dim3(bs), 0, 0, (const float4 *)x, (float4 *)y, nvec); });
            printf("copy-flat  U=4 NT=0 bs=%4d grid=%6ld  %8.1f us %6.3f TB/s\n", bs, grid, ms * 1e3, 2.0 * n * 4 / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL((k_copy<4, 3>), dim3(grid), dim3(bs), 0, 0, (const float4 *)x, (float4 *)y, nvec); });
            printf("copy-flat  U=4 NT=3 bs=%4d grid=%6ld  %8.1f us %6.3f TB/s\n", bs, grid, ms * 1e3, 2.0 * n * 4 / ms / 1e9);
        }
    }
#define RUNREAD(U, NT) \
    for (int bs : blocksizes) for (int b : bpc) { \
        long grid = (long)256 * b * 256 / bs; if (grid < 256) continue; \
        double ms = time_ms([&] { hipLaunchKernelGGL((k_read<U, NT>), dim3(grid), dim3(bs), 0, 0, (const float4 *)x, y, nvec); }); \
        printf("read       U=%d NT=%d bs=%4d grid=%6ld  %8.1f us %6.3f TB/s\n", U, NT, bs, grid, ms * 1e3, 1.0 * n * 4 / ms / 1e9); }
    RUNREAD(4, 0) RUNREAD(8, 0) RUNREAD(8, 1)
    return 0;
}
