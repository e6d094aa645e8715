// micro-benchmark: rocPRIM onesweep radix sort of u64 items by a bit range (keys only) vs (u64, u32) pairs
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill(uint64_t *k, uint32_t *v, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
        k[i] = x; v[i] = (uint32_t)i;
    }
}
using namespace rocprim;
template <unsigned RB, unsigned BS, unsigned IPT> using cfg =
    radix_sort_config<default_config, default_config,
                      radix_sort_onesweep_config<kernel_config<512, 32>, kernel_config<BS, IPT>, RB, block_radix_rank_algorithm::match>>;

template <class Config> void run_keys(uint64_t *ka, uint64_t *kb, size_t n, unsigned lo, unsigned hi, const char *name)
{
    size_t tmpb = 0;
    CK((radix_sort_keys<Config>(nullptr, tmpb, ka, kb, n, lo, hi, 0)));
    void *tmp; CK(hipMalloc(&tmp, tmpb));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(a));
        CK((radix_sort_keys<Config>(tmp, tmpb, ka, kb, n, lo, hi, 0)));
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("keys  %-22s n=%zu bits[%u,%u) : %.2f ms\n", name, n, lo, hi, best);
    CK(hipFree(tmp));
}
template <class Config> void run_pairs(uint64_t *ka, uint64_t *kb, uint32_t *va, uint32_t *vb, size_t n, unsigned lo, unsigned hi, const char *name)
{
    size_t tmpb = 0;
    CK((radix_sort_pairs<Config>(nullptr, tmpb, ka, kb, va, vb, n, lo, hi, 0)));
    void *tmp; CK(hipMalloc(&tmp, tmpb));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(a));
        CK((radix_sort_pairs<Config>(tmp, tmpb, ka, kb, va, vb, n, lo, hi, 0)));
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("pairs %-22s n=%zu bits[%u,%u) : %.2f ms\n", name, n, lo, hi, best);
    CK(hipFree(tmp));
}

int main(int argc, char **argv)
{
    size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : (1ull << 28);
    uint64_t *ka, *kb; uint32_t *va, *vb;
    CK(hipMalloc(&ka, n * 8)); CK(hipMalloc(&kb, n * 8)); CK(hipMalloc(&va, n * 4)); CK(hipMalloc(&vb, n * 4));
    fill<<<4096, 256>>>(ka, va, n);
    CK(hipDeviceSynchronize());
    run_keys<cfg<9, 1024, 8>>(ka, kb, n, 30, 64, "rb9 1024x8");
    run_keys<cfg<9, 1024, 12>>(ka, kb, n, 30, 64, "rb9 1024x12");
    run_keys<cfg<9, 512, 16>>(ka, kb, n, 30, 64, "rb9 512x16");
    run_keys<cfg<8, 1024, 12>>(ka, kb, n, 30, 64, "rb8 1024x12");
    run_keys<default_config>(ka, kb, n, 30, 64, "default");
    run_keys<cfg<9, 1024, 8>>(ka, kb, n, 28, 64, "rb9 1024x8");
    run_pairs<cfg<9, 1024, 8>>(ka, kb, va, vb, n, 0, 36, "rb9 1024x8");
    run_pairs<cfg<9, 1024, 8>>(ka, kb, va, vb, n, 0, 54, "rb9 1024x8");
    return 0;
}
