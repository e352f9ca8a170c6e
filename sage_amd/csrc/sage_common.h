// Shared host/device definitions for libsage_hip.so (gfx950 only).
#pragma once
#include <stdint.h>

#define SAGE_MAX_KEY_LEN 32
#define SAGE_ID_BITS 20
#define SAGE_ID_MASK 0xFFFFFu
#define SAGE_MAX_ID (SAGE_ID_MASK - 1)  // 0xFFFFF itself is reserved so that no live slot equals SAGE_SLOT_EMPTY
#define SAGE_SLOT_EMPTY 0xFFFFFFFFu
#define SAGE_SET_EMPTY 0xFFFFFFFFFFFFFFFFull

#if defined(__HIPCC__)
#define SAGE_HD __host__ __device__ __forceinline__
#else
#define SAGE_HD static inline
#endif

// FNV-1a over the key bytes followed by a 32-bit finaliser.  Prefix hashes of one text position are produced
// incrementally: h(j) = fnv_step(h(j-1), byte[j-1]).
#define SAGE_FNV_OFFSET 0x811C9DC5u
SAGE_HD uint32_t sage_fnv_step(uint32_t h, uint32_t b) { return (h ^ b) * 16777619u; }
SAGE_HD uint32_t sage_mix32(uint32_t x) {   // one multiply: v_mul_lo_u32 is quarter rate on gfx950
    x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 13;
    return x;
}
// A vocab-table slot: id[0:20) | (len-1)[20:25) | tag[25:32).  A bucket is 4 slots (one 16-byte load), filled in
// order; a bucket whose last slot is empty terminates a probe sequence (load factor <= 0.25, see sage_set_vocab).
// The 7-bit tag never takes the value 0x7F, so the top 12 bits of a live slot are never 0xFFF: an empty slot
// (0xFFFFFFFF) cannot pass a tag+length comparison and needs no separate test in the matcher's inner loop.
SAGE_HD uint32_t sage_slot_tag(uint32_t f) { const uint32_t t = f >> 25; return t < 0x7Eu ? t : 0x7Eu; }
SAGE_HD uint32_t sage_slot_make(uint32_t id, uint32_t len, uint32_t f) {
    return id | ((len - 1u) << 20) | (sage_slot_tag(f) << 25);
}
SAGE_HD uint32_t sage_slot_want(uint32_t len, uint32_t f) { return (sage_slot_tag(f) << 5) | (len - 1u); }

// LDS presence filter of the matcher: 2^20 bits (128 KiB) as 32,768 words; a key sets two bits inside ONE word
// (blocked Bloom filter: a membership test is a single LDS read).
#define SAGE_BLOOM_BITS (1u << 20)
SAGE_HD uint32_t sage_bloom_word(uint32_t f) { return (f >> 7) & 0x7FFFu; }
SAGE_HD uint32_t sage_bloom_mask(uint32_t f) { return (1u << (f & 31u)) | (1u << ((f >> 22) & 31u)); }

// Prefix-length table of the matcher: slot = mixed hash of a key's first 4 bytes, value = OR of (1 << (len-1)) over the
// keys of >= 4 bytes that land there.  Collisions only add bits, so it never rules out a real match.
#ifndef SAGE_PMASK_SLOTS
#define SAGE_PMASK_SLOTS (1u << 18)
#endif
#ifndef SAGE_PMASK8_SLOTS
#define SAGE_PMASK8_SLOTS (1u << 17)
#endif

SAGE_HD uint64_t sage_mix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

// Hashes of a 64-bit (a, lo, hi) key record for the distinct count: 32-bit arithmetic only (a 64-bit multiply is six
// instructions on the GPU, and the partition and dedupe kernels are bound by instruction issue).  hash_a: bits 0-5 the
// coarse bucket, 6-13 the fine bucket, 14-31 the set pass; hash_b: the slot in an LDS set.  They only spread keys --
// every set compares whole keys.
// The two hashes of a 60-bit key (a << 40 | lo << 20 | hi).  32-bit multiplies run at a quarter of the vector rate on
// gfx950 and the partition kernel hashes every key three times, so the key is taken as three 24-bit pieces and mixed with
// 24-bit multiplies (full rate); a fold and one more multiply bring every key bit into the low bits that pick a bucket
// (a) or a slot (b).  Bucket loads on 1.4 M keys with Zipf ids: Poisson (std 26.5 at a mean of 696, as with the 32-bit
// multiplies); b within a bucket of a: uncorrelated.
SAGE_HD uint32_t sage_mul24(uint32_t a, uint32_t b) {
#ifdef __HIP_DEVICE_COMPILE__
    return __umul24(a, b);
#else
    return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#endif
}
SAGE_HD uint32_t sage_key_hash_a(uint64_t k) {
    const uint32_t x = (uint32_t)k, y = (uint32_t)(k >> 32);
    const uint32_t mid = (x >> 24) | (y << 8);
    uint32_t h = sage_mul24(x, 0x9E3779u) + sage_mul24(mid, 0x85EBCBu) + sage_mul24(y >> 16, 0xC2B2AFu);
    h ^= h >> 16; h = sage_mul24(h, 0x2C1B3Du); h ^= h >> 13;
    return h;
}
SAGE_HD uint32_t sage_key_hash_b(uint64_t k) {
    const uint32_t x = (uint32_t)k, y = (uint32_t)(k >> 32);
    const uint32_t mid = (x >> 24) | (y << 8);
    uint32_t h = sage_mul24(x, 0x7FEB35u) + sage_mul24(mid, 0x27D4EBu) + sage_mul24(y >> 16, 0x165667u);
    h ^= h >> 15; h = sage_mul24(h, 0x846CA7u); h ^= h >> 12;
    return h;
}

// A match-table entry per corpus byte: x = id1 | len1<<20 | ends<<31 (longest key at this byte, never crossing the
// line end; ends = it reaches the line end exactly), y = id2 | len2<<20 (second longest; len2 == 0 iff len1 == 1).
#define SAGE_M_ID(w) ((w) & SAGE_ID_MASK)
#define SAGE_M_LEN(w) (((w) >> 20) & 0x3Fu)
#define SAGE_M_ENDS(w) ((w) >> 31)

// Token word: id | min(index in line, 3) << 20 | min(tokens after it in the line, 3) << 22
#define SAGE_T_ID(w) ((w) & SAGE_ID_MASK)
#define SAGE_T_LOC3(w) (((w) >> 20) & 3u)
#define SAGE_T_AFTER3(w) (((w) >> 22) & 3u)
#define SAGE_T_LEN(w) (((w) >> 24) & 0x3Fu)    /* width of the token in bytes */

// Occurrence descriptor: 16 u32 per base token.  word 0 = header, words 1.. = the ablated window
// [left pads | x1..xk | right pads] (at most SAGE_DESC_MAXWIN ids, otherwise OVER is set and the slow path runs).
// Distinct count (sage_kernels3.h): a workgroup stages the keys of a tile of occurrences in LDS; an occurrence with more
// position pairs than this is not part of any tile.  Inline descriptors keep their base span at or below
// SAGE_INLINE_MAX_M tokens so that an inline occurrence can never be a giant.
#define SAGE_PART_GIANT 2048u
#define SAGE_INLINE_MAX_M 255u
#define SAGE_DESC_WORDS 16
#define SAGE_DESC_MAXWIN 15
#define SAGE_DH_VALID 0x80000000u
#define SAGE_DH_OVER 0x40000000u
#define SAGE_DH_NEVER 0x20000000u
#define SAGE_DH_GIANT 0x10000000u   // more than SAGE_PART_GIANT position pairs: the distinct count's slow path
#define SAGE_DH_K(h) ((h) & 0xFFFu)
#define SAGE_DH_M(h) (((h) >> 12) & 0xFFFu)
#define SAGE_DH_PL(h) (((h) >> 24) & 3u)
#define SAGE_DH_PR(h) (((h) >> 26) & 3u)
