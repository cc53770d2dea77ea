// cache_bits_sweep.hip -- does any gfx950 cache-policy bit combination (sc0 / sc1 / nt) on the streaming loads and
// stores beat the compiler's "nontemporal" (= nt) for a read+write stream?  One 16 KiB piece per block.
// Build: hipcc --offload-arch=gfx950 -O3 tools/cache_bits_sweep.hip -o tools/cache_bits_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define DEF_KERNEL(NAME, LDBITS, STBITS)                                                                   \
    __global__ void __launch_bounds__(256) NAME(const vf4 *__restrict__ x, vf4 *__restrict__ y, long npieces) \
    {                                                                                                      \
        const long piece = blockIdx.x;                                                                     \
        if (piece >= npieces) return;                                                                      \
        const vf4 *xp = x + piece * 1024 + threadIdx.x;                                                    \
        vf4 *yp = y + piece * 1024 + threadIdx.x;                                                          \
        vf4 v0, v1, v2, v3;                                                                                \
        asm volatile("global_load_dwordx4 %0, %1, off " LDBITS : "=v"(v0) : "v"(xp));                      \
        asm volatile("global_load_dwordx4 %0, %1, off " LDBITS : "=v"(v1) : "v"(xp + 256));          \
        asm volatile("global_load_dwordx4 %0, %1, off " LDBITS : "=v"(v2) : "v"(xp + 512));          \
        asm volatile("global_load_dwordx4 %0, %1, off " LDBITS : "=v"(v3) : "v"(xp + 768));         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
        asm volatile("global_store_dwordx4 %0, %1, off " STBITS :: "v"(yp), "v"(v0) : "memory");           \
        asm volatile("global_store_dwordx4 %0, %1, off " STBITS :: "v"(yp + 256), "v"(v1) : "memory"); \
        asm volatile("global_store_dwordx4 %0, %1, off " STBITS :: "v"(yp + 512), "v"(v2) : "memory"); \
        asm volatile("global_store_dwordx4 %0, %1, off " STBITS :: "v"(yp + 768), "v"(v3) : "memory"); \
    }

#define BITS0 ""
#define BITS1 "nt"
#define BITS2 "sc0"
#define BITS3 "sc1"
#define BITS4 "sc0 sc1"
#define BITS5 "sc0 nt"
#define BITS6 "sc1 nt"
#define BITS7 "sc0 sc1 nt"
#define ROW(L) DEF_KERNEL(k_##L##_0, BITS##L, BITS0) DEF_KERNEL(k_##L##_1, BITS##L, BITS1) DEF_KERNEL(k_##L##_2, BITS##L, BITS2) \
    DEF_KERNEL(k_##L##_3, BITS##L, BITS3) DEF_KERNEL(k_##L##_4, BITS##L, BITS4) DEF_KERNEL(k_##L##_5, BITS##L, BITS5) \
    DEF_KERNEL(k_##L##_6, BITS##L, BITS6) DEF_KERNEL(k_##L##_7, BITS##L, BITS7)
ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7)

typedef void (*kern_t)(const vf4 *, vf4 *, long);
#define PTRS(L) k_##L##_0, k_##L##_1, k_##L##_2, k_##L##_3, k_##L##_4, k_##L##_5, k_##L##_6, k_##L##_7
static kern_t kerns[64] = {PTRS(0), PTRS(1), PTRS(2), PTRS(3), PTRS(4), PTRS(5), PTRS(6), PTRS(7)};
static const char *names[8] = {"-", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};

int main()
{
    const long n = 1L << 28;
    float *x, *y; CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4));
    CK(hipMemset(x, 1, n * 4)); CK(hipMemset(y, 0, n * 4));
    const long np = n / 4096;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 30; ++w) hipLaunchKernelGGL(kerns[9], dim3(np), dim3(256), 0, 0, (const vf4 *)x, (vf4 *)y, np);
    for (int rep = 0; rep < 2; ++rep)
        for (int l = 0; l < 8; ++l) {
            printf("load %-11s:", names[l]);
            for (int s = 0; s < 8; ++s) {
                std::vector<float> ts;
                for (int i = 0; i < 7; ++i) {
                    CK(hipEventRecord(a)); hipLaunchKernelGGL(kerns[l * 8 + s], dim3(np), dim3(256), 0, 0, (const vf4 *)x, (vf4 *)y, np);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
                }
                std::sort(ts.begin(), ts.end());
                printf(" %5.2f", 2.0 * n * 4 / ts[3] / 1e9);
            }
            printf("   TB/s (stores: - nt sc0 sc1 sc0sc1 sc0nt sc1nt sc0sc1nt)\n");
        }
    return 0;
}
