// sweep of rocPRIM onesweep configurations for 1e9 64-bit words sorted on bits [30, 64)
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)
__global__ void fill(uint64_t *k, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; k[i] = x ^ (x >> 31);
    }
}
using namespace rocprim;
template <unsigned HB, unsigned HI, unsigned RB, unsigned BS, unsigned IPT, block_radix_rank_algorithm A> using cfg =
    radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<HB, HI>, kernel_config<BS, IPT>, RB, A>>;
template <class Config> void run_keys(uint64_t *ka, uint64_t *kb, size_t n, const char *name)
{
    size_t tmpb = 0;
    CK((radix_sort_keys<Config>(nullptr, tmpb, ka, kb, n, 30, 64, 0)));
    void *tmp; CK(hipMalloc(&tmp, tmpb));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(a));
        CK((radix_sort_keys<Config>(tmp, tmpb, ka, kb, n, 30, 64, 0)));
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%-34s : %.2f ms\n", name, best);
    CK(hipFree(tmp));
}
#define RUN(HB, HI, RB, BS, IPT, A) run_keys<cfg<HB, HI, RB, BS, IPT, block_radix_rank_algorithm::A>>(ka, kb, n, #RB " bits, hist " #HB "x" #HI ", sort " #BS "x" #IPT " " #A)
int main(int argc, char **argv)
{
    size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : (1ull << 28);
    uint64_t *ka, *kb;
    if (hipMalloc(&ka, n * 8) != hipSuccess || hipMalloc(&kb, n * 8) != hipSuccess) return 1;
    fill<<<4096, 256>>>(ka, n);
    (void)hipDeviceSynchronize();
    RUN(512, 32, 9, 1024, 8, match);
    RUN(512, 32, 9, 1024, 6, match);
    RUN(512, 32, 9, 1024, 10, match);
    RUN(512, 32, 9, 512, 8, match);
    RUN(512, 32, 9, 512, 12, match);
    RUN(512, 32, 9, 256, 16, match);
    RUN(512, 32, 9, 256, 24, match);
    RUN(256, 16, 9, 1024, 8, match);
    RUN(1024, 16, 9, 1024, 8, match);
    RUN(512, 32, 8, 1024, 8, match);
    RUN(512, 32, 8, 512, 16, match);
    RUN(512, 32, 7, 1024, 8, match);
    return 0;
}
