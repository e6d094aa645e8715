// msd_keys.h -- key building shared by the MSD sorts (msd_sort.hip, msd_sort_pairs.hip): the symbol codes of a tile
// of text into LDS, then the keys of a thread's 8 consecutive positions.  Same keys as k_pack (suffix_sort.hip).
#pragma once
#include <stdint.h>

#define MSD_SEP 0x80                           // FBG_SEP of suffix_sort.hip
#define MSD_ITEMS 8

// The symbol codes of text positions base .. base + TILE + 64 go to LDS in two steps, so that a kernel walking several
// tiles can have the next tile's loads in flight: msd_fetch_raw (8 text bytes per thread, threads 0..7 also the 64
// bytes of lookahead; T is padded beyond N, base is a multiple of 8) and msd_store_tile (code table cd in LDS).
template <int TILE>
__device__ __forceinline__ void msd_fetch_raw(const uint8_t *__restrict__ T, uint64_t N, uint64_t base, uint64_t &raw, uint64_t &raw2)
{
    const uint64_t p = base + threadIdx.x * 8;
    raw = p < N + 56 ? *reinterpret_cast<const uint64_t *>(T + p) : 0ull;
    raw2 = 0;
    if (threadIdx.x < 8) {
        const uint64_t q = base + TILE + threadIdx.x * 8;
        raw2 = q < N + 56 ? *reinterpret_cast<const uint64_t *>(T + q) : 0ull;
    }
}
// FULL: the tile and its lookahead lie inside the text (base + TILE + 64 <= N): no end-of-text checks
template <int TILE, bool FULL = false>
__device__ __forceinline__ void msd_store_tile(uint8_t *tile, const uint8_t *cd, uint64_t N, uint64_t base, uint64_t raw, uint64_t raw2)
{
    const int k8 = threadIdx.x * 8;
    const uint64_t p = base + k8;
    uint64_t codes = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t c = (FULL || p + j < N) ? cd[(raw >> (8 * j)) & 255u] : (uint32_t)MSD_SEP;
        codes |= (uint64_t)c << (8 * j);
    }
    *reinterpret_cast<uint64_t *>(tile + k8) = codes;
    if (threadIdx.x < 8) {
        const int kk = TILE + threadIdx.x * 8;
        const uint64_t q = base + kk;
        uint64_t codes2 = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t c = (FULL || q + j < N) ? cd[(raw2 >> (8 * j)) & 255u] : (uint32_t)MSD_SEP;
            codes2 |= (uint64_t)c << (8 * j);
        }
        *reinterpret_cast<uint64_t *>(tile + kk) = codes2;
    }
}
template <int TILE, bool FULL = false>
__device__ __forceinline__ void msd_load_tile(uint8_t *tile, const uint8_t *cd, const uint8_t *__restrict__ T, uint64_t N, uint64_t base)
{
    uint64_t raw, raw2;
    msd_fetch_raw<TILE>(T, N, base, raw, raw2);
    msd_store_tile<TILE, FULL>(tile, cd, N, base, raw, raw2);
}

// w[i] = key of position t0 + i of the tile, i < MSD_ITEMS (t0 = 8 * threadIdx.x)
__device__ __forceinline__ void msd_build_keys(const uint8_t *tile, int t0, int b, int K, uint64_t *w)
{
    bool slow = true;
    if (b == 2 && K + MSD_ITEMS - 1 <= 32) {
        // 2-bit symbols: the thread's 32 symbols packed into one word (two multiplies per 8 bytes gather the low
        // two bits of every byte), every key a shift of it
        uint64_t P = 0, any = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint64_t x = *reinterpret_cast<const uint64_t *>(tile + t0 + 8 * q);
            any |= x;
            const uint32_t lo = (uint32_t)x & 0x03030303u, hi = (uint32_t)(x >> 32) & 0x03030303u;
            const uint32_t g = (((lo * 0x40100401u) >> 24) << 8) | ((hi * 0x40100401u) >> 24);    // 8 symbols, first one on top
            P |= (uint64_t)g << (48 - 16 * q);
        }
        slow = (any & 0x8080808080808080ull) != 0;              // a separator among the 32 symbols
#pragma unroll
        for (int i = 0; i < MSD_ITEMS; i++) w[i] = (P << (2 * i)) >> (64 - 2 * K);
    }
    if (b != 2 || K + MSD_ITEMS - 1 > 32) {
        // general alphabet: rolling, one LDS byte per position (k_pack)
        const uint64_t mask = (K * b) >= 64 ? ~0ull : ((1ull << (K * b)) - 1);
        uint64_t key = 0;
        uint32_t seen = 0;
        for (int k = 0; k < K; k++) { const uint32_t c = tile[t0 + k]; seen |= c; key = (key << b) | c; }
#pragma unroll
        for (int i = 0; i < MSD_ITEMS; i++) {
            w[i] = key;
            const uint32_t c = tile[t0 + K + i];
            seen |= c;
            key = ((key << b) | c) & mask;
        }
        slow = (seen & MSD_SEP) != 0;
    }
    if (slow) {
        // a row ends nearby: symbol by symbol; a separator and all behind it count as 0 (k_pack)
#pragma unroll
        for (int i = 0; i < MSD_ITEMS; i++) {
            uint64_t kk = 0;
            bool dead = false;
            for (int k = 0; k < K; k++) {
                const uint32_t c = tile[t0 + i + k];
                dead = dead || (c & MSD_SEP);
                kk = (kk << b) | (dead ? 0u : c);
            }
            w[i] = kk;
        }
    }
}
