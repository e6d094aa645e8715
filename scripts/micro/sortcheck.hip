// checks rocPRIM radix_sort_keys on a bit range [lo, hi) of u64 words for small and large inputs
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace rocprim;
using cfg = radix_sort_config<default_config, default_config,
                              radix_sort_onesweep_config<kernel_config<512, 32>, kernel_config<1024, 8>, 9, block_radix_rank_algorithm::match>>;
template <class C> int check(size_t n, unsigned lo, unsigned hi, const char *name)
{
    std::vector<uint64_t> h(n), o(n);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = x; }
    uint64_t *a, *b; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
    CK(hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice));
    size_t tb = 0; CK((radix_sort_keys<C>(nullptr, tb, a, b, n, lo, hi, 0)));
    void *tmp; CK(hipMalloc(&tmp, tb ? tb : 8));
    CK((radix_sort_keys<C>(tmp, tb, a, b, n, lo, hi, 0)));
    CK(hipMemcpy(o.data(), b, n * 8, hipMemcpyDeviceToHost));
    const uint64_t mask = hi - lo >= 64 ? ~0ull : ((1ull << (hi - lo)) - 1);
    size_t bad = 0;
    for (size_t i = 1; i < n; i++) if (((o[i - 1] >> lo) & mask) > ((o[i] >> lo) & mask)) bad++;
    printf("%-8s n=%zu bits[%u,%u): %zu inversions\n", name, n, lo, hi, bad);
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(tmp));
    return bad != 0;
}
__global__ void k_fill(uint64_t *a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; a[i] = x ^ (x >> 31);
    }
}
__global__ void k_inv(const uint64_t *o, size_t n, unsigned lo, uint64_t mask, unsigned long long *bad)
{
    for (size_t i = 1 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (((o[i - 1] >> lo) & mask) > ((o[i] >> lo) & mask)) atomicAdd(bad, 1ull);
}
template <class C> int check_big(size_t n, unsigned lo, unsigned hi, const char *name)
{
    uint64_t *a, *b; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
    k_fill<<<4096, 256>>>(a, n);
    size_t tb = 0; CK((radix_sort_keys<C>(nullptr, tb, a, b, n, lo, hi, 0)));
    void *tmp; CK(hipMalloc(&tmp, tb ? tb : 8));
    CK((radix_sort_keys<C>(tmp, tb, a, b, n, lo, hi, 0)));
    unsigned long long *d_bad, bad = 0; CK(hipMalloc(&d_bad, 8)); CK(hipMemset(d_bad, 0, 8));
    const uint64_t mask = hi - lo >= 64 ? ~0ull : ((1ull << (hi - lo)) - 1);
    k_inv<<<4096, 256>>>(b, n, lo, mask, d_bad);
    CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    printf("%-8s n=%zu bits[%u,%u): %llu inversions (device check)\n", name, n, lo, hi, bad);
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(tmp)); CK(hipFree(d_bad));
    return bad != 0;
}
int main()
{
    for (size_t n : {(size_t)1 << 23, (size_t)9000000, (size_t)250000000, (size_t)1000001001}) {
        check_big<cfg>(n, 30, 64, "tuned"); check_big<cfg>(n, 24, 64, "tuned");
    }
    int r = 0;
    for (size_t n : {100ul, 3000ul, 10508ul, 100000ul, 3000000ul}) {
        r |= check<cfg>(n, 14, 41, "tuned"); r |= check<cfg>(n, 14, 64, "tuned"); r |= check<cfg>(n, 14, 63, "tuned");
        r |= check<default_config>(n, 14, 64, "default");
    }
    return r;
}
