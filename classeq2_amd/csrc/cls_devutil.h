// Device-side helpers shared by the kernel files (cls_kernels.hip, cls_tile.hip): wave-level primitives, scalar-unit
// loads of tree rows, offset loads.  gfx950 / wave64 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cls_device.h"
#include "cls_murmur.h"

namespace cls {
namespace {

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, true);   // quad_perm:[1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, true);   // quad_perm:[2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2,3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// A DNode read through the constant address space: wave-uniform address -> s_load_dwordx8
// (scalar cache, no vector-memory instruction).  {pre, size, first_child, n_nonleaf, id lo, id hi, split, flags}
struct snode_t { uint32_t s[8]; };
__device__ __forceinline__ snode_t load_node(const DNode* nodes, uint32_t row) {
    typedef __attribute__((address_space(4))) const uint32_t as4_u32;
    typedef __attribute__((address_space(4))) const char as4_char;
    // (a 32-bit byte offset off the table's base: the node table stays far below 4 GiB, and the scalar load takes base + offset)
    as4_u32* p = (as4_u32*)((as4_char*)(uintptr_t)nodes + (uint32_t)(__builtin_amdgcn_readfirstlane(row) * (uint32_t)sizeof(DNode)));
    snode_t r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.s[i] = p[i];
    return r;
}

// both children of a binary clade (consecutive rows): one 64-byte scalar load
struct snode_pair_t { uint32_t s[16]; };
__device__ __forceinline__ snode_pair_t load_node_pair(const DNode* nodes, uint32_t row) {
    typedef __attribute__((address_space(4))) const uint32_t as4_u32;
    as4_u32* p = (as4_u32*)(uintptr_t)(nodes + __builtin_amdgcn_readfirstlane(row));
    snode_pair_t r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r.s[i] = p[i];
    return r;
}

// Indexed load off a wave-uniform base.  ADDR32: the byte offset is known to fit 32 bits (arrays below
// 4 GiB), which lets the compiler use the SGPR-base + 32-bit-VGPR-offset form instead of building a 64-bit
// address pair per access (each pair costs an extra VGPR holding the zero high half).
template <class T, bool ADDR32>
__device__ __forceinline__ T ldx(const void* base, uint32_t index) {
    if constexpr (ADDR32) return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (uint32_t)(index * (uint32_t)sizeof(T)));
    else return reinterpret_cast<const T*>(base)[index];
}

// 8-byte half of split record x (cls_device.h: TipRec): half 2x leads into the left part, 2x + 1 into the right one.
// Record indices reach 2^32 - 1 (cls_db.cpp rejects more), so without ADDR32 the half index is formed in 64 bits.
template <bool ADDR32>
__device__ __forceinline__ uint2 ld_half(const uint32_t* half, uint32_t x, uint32_t right) {
    if constexpr (ADDR32) return ldx<uint2, true>(half, 2 * x + right);
    else return reinterpret_cast<const uint2*>(half)[2ull * x + right];
}

// MurmurHash3_x64_128(p[0..len), seed 0).0 with the message fetched eight bytes at a time (gfx950 reads unaligned
// 64-bit words from LDS in one ds_read_b64).  Reads up to 15 bytes past the message, inside the caller's LDS buffer.
__device__ __forceinline__ uint64_t lds_u64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ uint64_t murmur3_h1_lds(const uint8_t* p, uint32_t len) {
    Mur3 m;
    const uint32_t nblocks = len >> 4;
#pragma unroll 1
    for (uint32_t b = 0; b < nblocks; ++b) m.block(lds_u64(p + 16 * b), lds_u64(p + 16 * b + 8));
    const uint32_t t = len & 15u;
    const uint8_t* tail = p + 16 * nblocks;
    uint64_t k1 = 0, k2 = 0;
    if (t > 0) { k1 = lds_u64(tail); if (t < 8) k1 &= (1ull << (8 * t)) - 1; }
    if (t > 8) { k2 = lds_u64(tail + 8) & ((1ull << (8 * (t - 8))) - 1); }
    return m.finish(k1, k2, t, len);
}

// 2-bit code (A0 C1 T2 G3, first character in the low bits) of the first m <= 8 characters at p (upper-case ACGT in LDS, eight
// readable bytes): the index into DbDev::mz_bucket
__device__ __forceinline__ uint32_t lds_prefix_code(const uint8_t* p, uint32_t m) {
    uint64_t x = (lds_u64(p) >> 1) & 0x0303030303030303ull;
    x = (x | (x >> 6)) & 0x000F000F000F000Full;
    x = (x | (x >> 12)) & 0x000000FF000000FFull;
    x = (x | (x >> 24)) & 0xFFFFull;
    return (uint32_t)x & ((1u << (2 * m)) - 1u);
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

}  // namespace
}  // namespace cls
