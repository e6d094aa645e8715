// text_cmp.h -- device helpers shared by the index kernels: 8-byte text comparison and the
// "same MSA column" test used for the run hints.
#pragma once
#include <stdint.h>

#define FBG_LCP_MASK 0x7fffffffu   // low 31 bits: LCP value; bit 31: SA neighbour may be coloured in the same column

__device__ __forceinline__ uint64_t fbg_load8(const uint8_t *__restrict__ T, uint64_t p)
{
    // unaligned 8-byte read assembled from two aligned words (text buffer is zero padded by 64 bytes)
    const uint64_t *w = reinterpret_cast<const uint64_t *>(T) + (p >> 3);
    const unsigned s = (unsigned)(p & 7) * 8;
    uint64_t lo = w[0];
    if (s == 0) return lo;
    uint64_t hi = w[1];
    return (lo >> s) | (hi << (64 - s));
}

// length of the common prefix of suffixes p and q, given that the first h symbols already match;
// terminates at the unique 0 sentinel at the latest
__device__ __forceinline__ uint32_t fbg_extend_match(const uint8_t *__restrict__ T, uint64_t p, uint64_t q, uint32_t h)
{
    uint64_t x = fbg_load8(T, p + h) ^ fbg_load8(T, q + h);
    while (x == 0) {
        h += 8;
        x = fbg_load8(T, p + h) ^ fbg_load8(T, q + h);
    }
    return h + (uint32_t)(__ffsll((unsigned long long)x) - 1) / 8;
}

__device__ __forceinline__ uint32_t fbg_clamp_lcp(uint32_t h) { return h > FBG_LCP_MASK ? FBG_LCP_MASK : h; }
