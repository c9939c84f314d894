// MurmurHash3 x64-128 (first half), seed 0 -- the key function of the k-mer
// index: KmersMap::hash_kmer, core/src/domain/dtos/kmers_map.rs:157-159
// (`mur3::murmurhash3_x64_128(kmer.as_bytes(), 0).0`, crate mur3 0.1.0).
// Shared by the host-side encoder/generator and the HIP kernels.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CLS_HD __host__ __device__ __forceinline__
#else
#define CLS_HD inline
#endif

namespace cls {

constexpr uint64_t MUR_C1 = 0x87c37b91114253d5ULL;
constexpr uint64_t MUR_C2 = 0x4cf5ad432745937fULL;

CLS_HD uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

CLS_HD uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

struct Mur3 {
    uint64_t h1 = 0, h2 = 0;
    CLS_HD void block(uint64_t k1, uint64_t k2) {
        k1 *= MUR_C1; k1 = rotl64(k1, 31); k1 *= MUR_C2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= MUR_C2; k2 = rotl64(k2, 33); k2 *= MUR_C1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    // tail words: bytes 0..7 in k1, 8..14 in k2 (little endian, zero padded)
    CLS_HD uint64_t finish(uint64_t k1, uint64_t k2, uint32_t tail_len, uint64_t total_len) {
        if (tail_len > 8) { k2 *= MUR_C2; k2 = rotl64(k2, 33); k2 *= MUR_C1; h2 ^= k2; }
        if (tail_len > 0) { k1 *= MUR_C1; k1 = rotl64(k1, 31); k1 *= MUR_C2; h1 ^= k1; }
        h1 ^= total_len; h2 ^= total_len;
        h1 += h2; h2 += h1;
        h1 = fmix64(h1); h2 = fmix64(h2);
        h1 += h2;
        return h1;
    }
};

// Generic byte-fetch form: get(i) returns byte i of the message.
template <class Get>
CLS_HD uint64_t murmur3_h1(Get get, uint32_t len) {
    Mur3 m;
    uint32_t nblocks = len >> 4;
    for (uint32_t b = 0; b < nblocks; ++b) {
        uint64_t k1 = 0, k2 = 0;
        for (int i = 0; i < 8; ++i) k1 |= (uint64_t)get(16 * b + i) << (8 * i);
        for (int i = 0; i < 8; ++i) k2 |= (uint64_t)get(16 * b + 8 + i) << (8 * i);
        m.block(k1, k2);
    }
    uint32_t t = len & 15, base = nblocks << 4;
    uint64_t k1 = 0, k2 = 0;
    for (uint32_t i = 0; i < t && i < 8; ++i) k1 |= (uint64_t)get(base + i) << (8 * i);
    for (uint32_t i = 8; i < t; ++i) k2 |= (uint64_t)get(base + i) << (8 * (i - 8));
    return m.finish(k1, k2, t, len);
}

inline uint64_t murmur3_h1_bytes(const char* p, uint32_t len) {
    return murmur3_h1([p](uint32_t i) { return (uint8_t)p[i]; }, len);
}

}  // namespace cls
