// micro-benchmark: rocPRIM onesweep radix sort of (u64 key, u32 value) pairs with different digit widths
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill(uint64_t *k, uint32_t *v, size_t n, int bits)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
        k[i] = bits >= 64 ? x : (x & ((1ull << bits) - 1)); v[i] = (uint32_t)i;
    }
}

template <class Config> float run(uint64_t *ka, uint64_t *kb, uint32_t *va, uint32_t *vb, size_t n, int bits, const char *name)
{
    size_t tmpb = 0;
    CK((rocprim::radix_sort_pairs<Config>(nullptr, tmpb, ka, kb, va, vb, n, 0u, (unsigned)bits, 0)));
    void *tmp; CK(hipMalloc(&tmp, tmpb));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(a));
        CK((rocprim::radix_sort_pairs<Config>(tmp, tmpb, ka, kb, va, vb, n, 0u, (unsigned)bits, 0)));
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%-28s n=%zu bits=%d : %.2f ms (tmp %zu MB)\n", name, n, bits, best, tmpb >> 20);
    CK(hipFree(tmp));
    return best;
}

using namespace rocprim;
template <unsigned RB, unsigned BS, unsigned IPT> using cfg =
    radix_sort_config<default_config, default_config,
                      radix_sort_onesweep_config<kernel_config<512, 32>, kernel_config<BS, IPT>, RB, block_radix_rank_algorithm::match>>;

int main(int argc, char **argv)
{
    size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : (1ull << 28);
    int bits = argc > 2 ? atoi(argv[2]) : 63;
    uint64_t *ka, *kb; uint32_t *va, *vb;
    CK(hipMalloc(&ka, n * 8)); CK(hipMalloc(&kb, n * 8)); CK(hipMalloc(&va, n * 4)); CK(hipMalloc(&vb, n * 4));
    fill<<<4096, 256>>>(ka, va, n, bits);
    CK(hipDeviceSynchronize());
    run<default_config>(ka, kb, va, vb, n, bits, "default");
    run<cfg<8, 512, 12>>(ka, kb, va, vb, n, bits, "rb8 512x12 match");
    run<cfg<8, 1024, 8>>(ka, kb, va, vb, n, bits, "rb8 1024x8 match");
    run<cfg<9, 512, 12>>(ka, kb, va, vb, n, bits, "rb9 512x12 match");
    run<cfg<9, 1024, 8>>(ka, kb, va, vb, n, bits, "rb9 1024x8 match");
    run<cfg<10, 512, 12>>(ka, kb, va, vb, n, bits, "rb10 512x12 match");
    run<cfg<10, 1024, 8>>(ka, kb, va, vb, n, bits, "rb10 1024x8 match");
    return 0;
}
