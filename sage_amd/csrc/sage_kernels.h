// Device kernels of libsage_hip.so -- hand-written for gfx950 (MI355X): 64-wide wavefronts, 160 KB LDS per CU,
// HBM-bound integer/byte work plus a gather-bound f64 scoring kernel.  No MFMA on purpose: the only arithmetic
// is a D=50 dot product per (target, context) pair.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "sage_common.h"

// A workgroup barrier for kernels whose threads exchange data through LDS only.  __syncthreads() also orders GLOBAL memory:
// its release fence makes every wave wait for all of its outstanding vector-memory operations (`s_waitcnt vmcnt(0)`), so a
// prefetch issued before a barrier is waited for AT the barrier and overlaps nothing behind it.  Here only LDS traffic is
// waited for; global loads stay in flight across the barrier (their results are per-thread registers, which no other
// thread reads).  Not for code that passes data between threads through global memory.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct VocabDev {
    const uint4* buckets;     // nbuckets * 4 slots
    uint32_t bucket_mask;     // nbuckets - 1
    const uint64_t* keys;     // [word][id] zero-padded key bytes as little-endian 8-byte words, one plane per word:
                              // keys of <= 8 bytes (most matches) touch plane 0 only, 16 keys per 128-B line
    uint32_t key_plane;       // ids per plane
    const uint32_t* byte_id;  // id of each single-byte key
    const uint32_t* direct2;  // [65536] id of each 2-byte key (index = its two bytes, little-endian), SAGE_SLOT_EMPTY if none
    const uint32_t* bloom;    // SAGE_BLOOM_BITS-bit presence filter over (length, bytes), two bits per key
    const void* pmask;        // [SAGE_PMASK_SLOTS] lengths (bit j-1) of the keys whose first 4 bytes hash to the slot;
                              // uint16 entries when max_len <= 16 (half the cache footprint), else uint32
    const void* pmask8;       // [SAGE_PMASK8_SLOTS] the same for the first 8 bytes (keys of >= 8 bytes)
    int max_len;
    const uint4* ebk;         // embedded-key table of the queued matcher: one 64-byte bucket = 4 slots as
                              // {u64 first 8 key bytes [4]; u32 id | (len-1) << 20 [4]; u32 key bytes 8..11 [4]}, keys of >= 3 bytes.
                              // A probe is ONE cache-line request: keys of <= 12 bytes are verified from the bucket itself
    uint32_t enb;             // buckets of ebk (any number; bucket = mulhi(hash, enb))
    const uint32_t* parent;   // [id] the longest PROPER prefix of key id that is itself a key, as id | len << 20 (every key of
                              // 2+ bytes has one: its first byte).  The second-longest match at a text position is the
                              // parent of the longest one -- every shorter match is a prefix of the longest -- so the
                              // queued matcher stops probing at its first hit (k_vocab_parents builds the table)
};

enum { FLAG_SET_OVERFLOW = 0, FLAG_ROW_OOB = 1, FLAG_EMPTY_LINE = 2, FLAG_COUNT = 8 };
enum { STAT_OCC = 0, STAT_ROWS = 1, STAT_PAIRS = 2, STAT_DISTINCT = 3, STAT_INEXACT = 4, STAT_WROWS = 5 /* ids of the ablated windows: the rows k_dots gathers */,
       STAT_LONG = 6 /* occurrences whose window went to the id pool */, STAT_ABPAIRS = 7 /* position pairs of the ablated windows */,
       STAT_RECORDS = 8 /* key records the dedupe kernel read: must equal STAT_PAIRS, every record scattered and read once */,
       STAT_WROWS_INEXACT = 9 /* window ids of the occurrences on the inexact list (scored by k_dots_mixed) */,
       STAT_MARKED_ROWS = 10 /* of those, the rows k_dots_mixed gathered from the float64 matrix (marked in the bitmap) */,
       STAT_COUNT = 12 };

// ----------------------------------------------------------------------------------------------------------
// rem[p] = min(255, line_end(p) - p): one byte per corpus byte, computed once per corpus (set_corpus).
// ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_line_remaining(const int64_t* __restrict__ line_off, int64_t n_lines,
                                                        uint8_t* __restrict__ rem, int64_t nbytes) {
    // grid-stride: a launch may not have 2^32 threads or more, and a corpus may have more bytes than that
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nbytes; p += (int64_t)gridDim.x * 256) {
        // upper_bound(line_off, p) over line_off[1..n_lines]; lanes of a wave walk nearly the same path (L1 hits)
        int64_t lo = 0, hi = n_lines;  // find smallest l with line_off[l+1] > p
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (line_off[mid + 1] > p) hi = mid; else lo = mid + 1;
        }
        int64_t r = line_off[lo + 1] - p;
        rem[p] = (uint8_t)(r > 255 ? 255 : r);
    }
}

// ----------------------------------------------------------------------------------------------------------
// K1  match: for EVERY corpus byte p, the longest and second-longest vocabulary key that is a prefix of the
// text at p and does not cross the line end (model.py:90-92 with the slice saturation of sent[i:i+j]).
// One thread per byte; the block's text window is staged through LDS once; the (ML-1) candidate lengths are
// hashed incrementally (FNV-1a), all their 16-byte bucket loads are issued before the first is consumed.
// The table is 4 B/slot so 262,144 keys occupy 2 MiB: resident in each XCD's 4 MiB L2.
// ----------------------------------------------------------------------------------------------------------
template <int NW>  // key words of 8 bytes: handles max_len <= 8*NW
__global__ __launch_bounds__(256) void k_match(const uint8_t* __restrict__ corpus, const uint8_t* __restrict__ rem,
                                               int64_t pos0, uint32_t count, VocabDev v, uint32_t* __restrict__ M) {
    constexpr int ML = 8 * NW;
    constexpr int NDW = (256 + ML + 8) / 4 + 1;
    __shared__ uint32_t text[NDW];
    const uint32_t tid = threadIdx.x;
    const int64_t blk0 = pos0 + (int64_t)blockIdx.x * 256;
    const int64_t a0 = blk0 & ~(int64_t)3;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(corpus + a0);
    for (uint32_t i = tid; i < NDW; i += 256) text[i] = src[i];
    __syncthreads();
    const uint32_t rel = blockIdx.x * 256u + tid;
    if (rel >= count) return;
    const uint32_t o = (uint32_t)(blk0 - a0) + tid;
    const uint32_t wi = o >> 2, sh = (o & 3u) * 8u;
    uint32_t T[2 * NW];
    {
        uint32_t d[2 * NW + 1];
#pragma unroll
        for (int i = 0; i <= 2 * NW; ++i) d[i] = text[wi + i];
#pragma unroll
        for (int i = 0; i < 2 * NW; ++i) T[i] = __funnelshift_r(d[i], d[i + 1], sh);
    }
    const int lim = min((int)rem[blk0 + tid], v.max_len);

    // phase 1: prefix hashes, bucket loads in flight
    uint32_t want[ML + 1];
    uint32_t bidx[ML + 1];
    uint4 bk[ML + 1];
    uint32_t h = sage_fnv_step(SAGE_FNV_OFFSET, T[0] & 0xFFu);
#pragma unroll
    for (int j = 2; j <= ML; ++j) {
        const uint32_t b = (T[(j - 1) >> 2] >> (8 * ((j - 1) & 3))) & 0xFFu;
        h = sage_fnv_step(h, b);
        const uint32_t f = sage_mix32(h);
        want[j] = sage_slot_want(j, f);
        bidx[j] = f & v.bucket_mask;
        if (j <= lim) bk[j] = v.buckets[bidx[j]];
        else bk[j] = make_uint4(SAGE_SLOT_EMPTY, SAGE_SLOT_EMPTY, SAGE_SLOT_EMPTY, SAGE_SLOT_EMPTY);
    }
    uint64_t W[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) W[w] = (uint64_t)T[2 * w] | ((uint64_t)T[2 * w + 1] << 32);

    uint32_t id1 = v.byte_id[T[0] & 0xFFu], len1 = 1, id2 = 0, len2 = 0;
    // phase 2: ascending lengths, so the last two hits are the longest two
#pragma unroll
    for (int j = 2; j <= ML; ++j) {
        if (j > lim) continue;
        uint4 s = bk[j];
        uint32_t b = bidx[j];
        bool hit = false;
        uint32_t hid = 0;
        for (;;) {
            const uint32_t slot[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!hit && slot[q] != SAGE_SLOT_EMPTY && (slot[q] >> 20) == want[j]) {
                    const uint32_t cand = slot[q] & SAGE_ID_MASK;
                    bool same = true;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const int nb = j - 8 * w;
                        if (nb > 0) {
                            const uint64_t mask = nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull);
                            same = same && ((W[w] & mask) == v.keys[(size_t)w * v.key_plane + cand]);
                        }
                    }
                    if (same) { hit = true; hid = cand; }
                }
            }
            if (hit || s.w == SAGE_SLOT_EMPTY) break;
            b = (b + 1u) & v.bucket_mask;
            s = v.buckets[b];
        }
        if (hit) { id2 = id1; len2 = len1; id1 = hid; len1 = (uint32_t)j; }
    }
    const uint32_t ends1 = len1 == (uint32_t)rem[blk0 + tid] ? 0x80000000u : 0u;
    (void)id2; (void)len2;      // the second match is the parent of the first (k_vocab_parents): not stored
    M[rel] = id1 | (len1 << 20) | ends1;
}

// ----------------------------------------------------------------------------------------------------------
// K0  vocabulary tables, built on the GPU from the packed keys (model.py:28-51 keeps a dict; this is its device form).
// One thread per key: key planes, single-byte and 2-byte direct tables, the LDS filter's bits, the two prefix-length
// tables, the slot table (plain / column-probe matchers) and the embedded-key table (queued matcher).  Slots are
// claimed with compare-and-swap in slot order, so "last slot empty" still means "nothing spilled to the next bucket";
// which key sits in which slot depends on the run, what a lookup returns does not (keys are unique).
// ----------------------------------------------------------------------------------------------------------
struct VocabBuild {
    const uint8_t* key_bytes; const int64_t* key_off; const int32_t* key_ids; uint32_t n_keys; uint32_t max_len; int nw;
    uint32_t* slots; uint32_t bucket_mask; uint64_t* keys; uint32_t key_plane;
    uint32_t* byte_id; uint32_t* direct2; uint32_t* bloom; uint32_t* pmask; uint32_t* pmask8; int pm16;
    uint32_t* etab; uint32_t enb;
};
__device__ __forceinline__ void pmask_or(uint32_t* tab, uint32_t idx, uint32_t bits, int pm16) {
    if (pm16) atomicOr(&tab[idx >> 1], (bits & 0xFFFFu) << ((idx & 1u) * 16u));
    else atomicOr(&tab[idx], bits);
}
__global__ __launch_bounds__(256) void k_vocab_build(VocabBuild B) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= B.n_keys) return;
    const uint32_t id = B.key_ids ? (uint32_t)B.key_ids[i] : i;
    const uint8_t* k = B.key_bytes + B.key_off[i];
    const uint32_t len = (uint32_t)(B.key_off[i + 1] - B.key_off[i]);
    if (len > B.max_len) return;      // can never match (model.py:90 tries lengths max_len..1 only): in no table
    uint64_t words[4] = {0ull, 0ull, 0ull, 0ull};
    for (uint32_t j = 0; j < len; ++j) words[j >> 3] |= (uint64_t)k[j] << (8u * (j & 7u));
    for (int w = 0; w < B.nw; ++w) B.keys[(size_t)w * B.key_plane + id] = words[w];
    if (len == 1u) { B.byte_id[k[0]] = id; return; }
    if (len == 2u) B.direct2[(uint32_t)k[0] | ((uint32_t)k[1] << 8)] = id;     // (and in the slot table for k_match)
    uint32_t h = SAGE_FNV_OFFSET;
    for (uint32_t j = 0; j < len; ++j) {
        h = sage_fnv_step(h, k[j]);
        if (j == 3u) pmask_or(B.pmask, sage_mix32(h) & (SAGE_PMASK_SLOTS - 1u), 1u << (len - 1u), B.pm16);    // len >= 4 here
        if (j == 7u) pmask_or(B.pmask8, sage_mix32(h) & (SAGE_PMASK8_SLOTS - 1u), 1u << (len - 1u), B.pm16);  // len >= 8 here
    }
    const uint32_t f = sage_mix32(h);
    atomicOr(&B.bloom[sage_bloom_word(f)], sage_bloom_mask(f));
    {
        const uint32_t slot = sage_slot_make(id, len, f);
        uint32_t b = f & B.bucket_mask;
        for (bool done = false; !done; b = (b + 1u) & B.bucket_mask)
            for (int q = 0; q < 4 && !done; ++q) done = atomicCAS(&B.slots[(size_t)b * 4 + q], SAGE_SLOT_EMPTY, slot) == SAGE_SLOT_EMPTY;
    }
    if (len >= 3u) {
        const uint32_t meta = id | ((len - 1u) << 20);
        uint32_t b = __umulhi(f, B.enb);
        for (bool done = false; !done; b = b + 1u == B.enb ? 0u : b + 1u) {
            uint32_t* s = B.etab + (size_t)b * 16;
            for (int q = 0; q < 4 && !done; ++q)
                if (atomicCAS(&s[8 + q], SAGE_SLOT_EMPTY, meta) == SAGE_SLOT_EMPTY) {
                    s[2 * q] = (uint32_t)words[0];
                    s[2 * q + 1] = (uint32_t)(words[0] >> 32);
                    s[12 + q] = (uint32_t)words[1];
                    done = true;
                }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------
// K1 (filtered form): the same match, with a 128 KiB presence filter (2 bits per key) resident in LDS.  k_match is
// bound by the L2 request rate of its 15 bucket probes per byte (measured: ~20 L2 requests per byte); the filter
// turns most misses into two LDS reads (measured: ~5 L2 requests per byte).  Candidate lengths are probed in chunks
// of 8 from the longest down: all bucket loads of a chunk are in flight together, then all key loads, so a wave
// pays ~2 memory latencies per chunk instead of one per length; a chunk is skipped once two matches are known.
// One persistent 1024-thread block per CU keeps the filter in LDS while it sweeps tiles of 1024 corpus bytes.
// ----------------------------------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ bool key_equal(const uint64_t (&W)[NW], const uint64_t* __restrict__ key, int j) {
    bool same = true;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int nb = j - 8 * w;
        if (nb > 0) {
            const uint64_t mask = nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull);
            same = same && ((W[w] & mask) == key[w]);
        }
    }
    return same;
}
// the same against the key planes in memory
template <int NW>
__device__ __forceinline__ bool key_equal_mem(const VocabDev& v, const uint64_t (&W)[NW], uint32_t id, int j) {
    uint64_t K[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) K[w] = (j - 8 * w > 0) ? v.keys[(size_t)w * v.key_plane + id] : 0ull;
    return key_equal<NW>(W, K, j);
}

// exact serial probe of one length (any bucket chain length, any number of tag collisions)
template <int NW>
__device__ __forceinline__ bool probe_exact(const VocabDev& v, uint32_t b, uint32_t want, const uint64_t (&W)[NW], int j,
                                            uint32_t& hid) {
    for (;;) {
        const uint4 s = v.buckets[b];
        const uint32_t slot[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (slot[q] != SAGE_SLOT_EMPTY && (slot[q] >> 20) == want) {
                const uint32_t cnd = slot[q] & SAGE_ID_MASK;
                if (key_equal_mem<NW>(v, W, cnd, j)) { hid = cnd; return true; }
            }
        }
        if (s.w == SAGE_SLOT_EMPTY) return false;
        b = (b + 1u) & v.bucket_mask;
    }
}

// The two longest matches found so far as packed words id | len << 20 (the match table's own format), m1 >= m2, 0 = none.
// A match of any length folds in with two instructions, whatever order the lengths come in: the word's high bits are
// the length, so the largest word is the longest match and the median of (m1, m2, new) the second longest.
struct Top2 {
    uint32_t m1, m2;
    __device__ __forceinline__ void fold(uint32_t r) {      // r = id | len << 20, or 0 for "no match at this length"
        const uint32_t second = max(m2, min(m1, r));     // the median of the three, given m1 >= m2
        m1 = max(m1, r);
        m2 = second;
    }
};

// Probes the candidate lengths JLO..JHI (at most 8) with all loads of a phase in flight together.  Branch-free:
// a lane that has no candidate at a length loads bucket 0 / key 0 instead (one shared cache line), which costs
// far less than the exec-mask bookkeeping of a divergent branch per length.  Lengths are folded into `top`
// from the longest down.  want_of(f, j) = (tag << 25) | ((j-1) << 20): a slot matches when slot ^ want < 2^20.
// fh[j] = finalised hash of the j-byte prefix; bucket index and wanted tag/length word are derived where a length is
// actually probed (they used to be computed for all 15 lengths up front)
__device__ __forceinline__ uint32_t want_of(uint32_t f, uint32_t j) { return (sage_slot_tag(f) << 25) | ((j - 1u) << 20); }

template <int NW, int JLO, int JHI, int ML>
__device__ __forceinline__ void probe_chunk(const VocabDev& v, uint32_t cand, const uint32_t (&fh)[ML + 1],
                                            const uint64_t (&W)[NW], Top2& top) {
    constexpr int N = JHI - JLO + 1;
    uint4 bk[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        // 0 or ~0 from the candidate bit without going through a condition register
        const uint32_t onm = (uint32_t)(((int32_t)(cand << (32 - (JLO + i)))) >> 31);
        bk[i] = v.buckets[fh[JLO + i] & v.bucket_mask & onm];
    }
    uint32_t cid[N];
    bool have[N], full[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t w = want_of(fh[JLO + i], JLO + i);
        const bool m0 = (bk[i].x ^ w) < (1u << 20), m1 = (bk[i].y ^ w) < (1u << 20);
        const bool m2 = (bk[i].z ^ w) < (1u << 20), m3 = (bk[i].w ^ w) < (1u << 20);
        uint32_t c = bk[i].x;
        c = m1 ? bk[i].y : c;
        c = m2 ? bk[i].z : c;
        c = m3 ? bk[i].w : c;
        cid[i] = c & SAGE_ID_MASK;
        have[i] = (m0 | m1 | m2 | m3) && ((cand >> (JLO + i - 1)) & 1u);
        full[i] = bk[i].w != SAGE_SLOT_EMPTY;
    }
    uint64_t K[N][NW];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t* key = v.keys + (have[i] ? cid[i] : 0u);
#pragma unroll
        for (int w = 0; w < NW; ++w)
            if (JLO + i - 8 * w > 0) K[i][w] = key[(size_t)w * v.key_plane];
    }
    uint32_t hitm = 0, slow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool eq = key_equal<NW>(W, K[i], JLO + i);
        const bool on = (cand >> (JLO + i - 1)) & 1u;
        hitm |= (have[i] && eq) ? (1u << i) : 0u;
        // rare: a tag match that fails the key compare, or a full bucket without a match -> exact serial probe
        slow |= (on && !(have[i] && eq) && (have[i] || full[i])) ? (1u << i) : 0u;
    }
    if (slow) {
#pragma unroll
        for (int i = 0; i < N; ++i)
            if ((slow >> i) & 1u) {
                uint32_t id;
                if (probe_exact<NW>(v, fh[JLO + i] & v.bucket_mask, want_of(fh[JLO + i], JLO + i) >> 20, W, JLO + i, id)) { hitm |= 1u << i; cid[i] = id; }
            }
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) top.fold(((hitm >> i) & 1u) ? (cid[i] | ((uint32_t)(JLO + i) << 20)) : 0u);
}

// Lengths above 8 are sparse once the prefix tables have spoken (a few lanes of a wave have a candidate there, and
// rarely more than two), so they are not probed column by column: every lane walks ITS OWN candidate lengths from the
// longest down, one per round, until no lane has any left.  Two or three rounds replace eight columns.
template <int N>
__device__ __forceinline__ uint32_t pick_of(const uint32_t* a, uint32_t idx) {   // a[idx], idx < N, registers only
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = a[i];
#pragma unroll
    for (int sft = 1; sft < N; sft <<= 1) {
        const bool b = (idx & (uint32_t)sft) != 0u;
#pragma unroll
        for (int k = 0; k + sft < N; k += 2 * sft) t[k] = b ? t[k + sft] : t[k];
    }
    return t[0];
}

template <int NW, int JLO, int JHI, int ML>
__device__ __forceinline__ void probe_sparse(const VocabDev& v, uint32_t cand, const uint32_t (&fh)[ML + 1],
                                             const uint64_t (&W)[NW], Top2& top) {
    constexpr int N = JHI - JLO + 1;
    constexpr int PW = 2;   // candidates a lane probes per round: both bucket loads, then both key loads, in flight together
    uint32_t c = (cand >> (JLO - 1)) & ((N >= 32) ? 0xFFFFFFFFu : ((1u << N) - 1u));   // bit i <-> length JLO + i
    while (__any(c != 0u && top.m2 == 0u)) {
        bool active[PW], have[PW];
        uint32_t j[PW], w[PW], b[PW], id[PW];
        uint4 bk[PW];
        uint64_t K[PW][NW];
#pragma unroll
        for (int r = 0; r < PW; ++r) {
            active[r] = c != 0u && top.m2 == 0u;
            const uint32_t i = active[r] ? 31u - (uint32_t)__clz((int)c) : 0u;
            c = active[r] ? (c & ~(1u << i)) : c;
            j[r] = (uint32_t)JLO + i;
            const uint32_t f = pick_of<N>(&fh[JLO], i);
            w[r] = want_of(f, j[r]);
            b[r] = f & v.bucket_mask;
            bk[r] = v.buckets[active[r] ? b[r] : 0u];
        }
#pragma unroll
        for (int r = 0; r < PW; ++r) {
            const bool m0 = (bk[r].x ^ w[r]) < (1u << 20), m1 = (bk[r].y ^ w[r]) < (1u << 20);
            const bool m2 = (bk[r].z ^ w[r]) < (1u << 20), m3 = (bk[r].w ^ w[r]) < (1u << 20);
            uint32_t x = bk[r].x;
            x = m1 ? bk[r].y : x;
            x = m2 ? bk[r].z : x;
            x = m3 ? bk[r].w : x;
            id[r] = x & SAGE_ID_MASK;
            have[r] = active[r] && (m0 | m1 | m2 | m3);
#pragma unroll
            for (int q = 0; q < NW; ++q)
                if (8 * q < JHI) K[r][q] = v.keys[(size_t)q * v.key_plane + (have[r] && (int)j[r] > 8 * q ? id[r] : 0u)];
        }
#pragma unroll
        for (int r = 0; r < PW; ++r) {   // r = 0 is the longer length
            bool eq = have[r];
#pragma unroll
            for (int q = 0; q < NW; ++q)
                if (8 * q < JHI) {
                    if (JLO > 8 && JHI <= 16) {   // word 0 is whole, word 1 holds j - 8 = 1..8 bytes: no clamping
                        if (q == 0) eq = eq && W[0] == K[r][0];
                        else eq = eq && ((W[q] ^ K[r][q]) << ((16u - j[r]) * 8u)) == 0ull;
                    } else {
                        const int nb = (int)j[r] - 8 * q;              // bytes of the key in word q
                        const uint32_t sh = 64u - 8u * (uint32_t)min(max(nb, 0), 8);
                        eq = eq && (nb <= 0 || ((W[q] ^ K[r][q]) << (sh & 63u)) == 0ull);
                    }
                }
            // rare: a tag match that fails the key compare, or a full bucket without a match -> exact serial probe
            if (active[r] && !eq && (have[r] || bk[r].w != SAGE_SLOT_EMPTY))
                eq = probe_exact<NW>(v, b[r], w[r] >> 20, W, (int)j[r], id[r]);
            top.fold((active[r] && eq) ? (id[r] | (j[r] << 20)) : 0u);
        }
    }
}

// exact serial probe of one length in the embedded-key table (any chain length): every slot of the right length whose
// first 8 bytes match is verified against the key planes
template <int NW>
__device__ __forceinline__ bool probe_exact_e(const VocabDev& v, uint32_t b, const uint64_t (&W)[NW], uint32_t j, uint32_t& hid) {
    const uint32_t sh0 = 64u - 8u * min(j, 8u);
    for (;;) {
        const uint4* bp = v.ebk + (size_t)b * 4u;
        const uint4 k01 = bp[0], k23 = bp[1], meta = bp[2];
        const uint64_t key[4] = {(uint64_t)k01.x | ((uint64_t)k01.y << 32), (uint64_t)k01.z | ((uint64_t)k01.w << 32),
                                 (uint64_t)k23.x | ((uint64_t)k23.y << 32), (uint64_t)k23.z | ((uint64_t)k23.w << 32)};
        const uint32_t mt[4] = {meta.x, meta.y, meta.z, meta.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if ((mt[t] >> 20) == j - 1u && ((key[t] ^ W[0]) << (sh0 & 63u)) == 0ull) {
                const uint32_t cnd = mt[t] & SAGE_ID_MASK;
                if (j <= 8u || key_equal_mem<NW>(v, W, cnd, (int)j)) { hid = cnd; return true; }
            }
        }
        if (meta.w == SAGE_SLOT_EMPTY) return false;
        b = b + 1u == v.enb ? 0u : b + 1u;
    }
}

// Queued probe (the default): column-by-column probing spends a probe slot per lane and length whether or not the lane
// has a candidate there (about 10 slots per position for 3-4 candidates).  Here the (position, length) candidates of a
// length class are compacted into a per-wave LDS queue, longest length first (a ballot per length gives every lane its
// entry index without a scan), and the wave then probes 64 REAL candidates per round: lane e takes entry e, fetches
// the text words of the entry's position from the owner lane's registers (ds_bpermute), probes, and leaves the result
// in the entry; the owners fold their results afterwards.  A class larger than the queue is done in windows.
#ifdef MATCH_STATS
__device__ unsigned long long g_match_dbg[16];
#endif
#define MQ_CAP 224u      // entries per wave: 16 waves x 224 x 8 B = 28 KiB beside the 128 KiB filter
template <int NW, int JLO, int JHI, int ML>
__device__ __forceinline__ void probe_queued(const VocabDev& v, uint32_t cand, const uint32_t (&fh)[ML + 1],
                                             const uint64_t (&W)[NW], Top2& top, uint2* __restrict__ q, uint32_t lane) {
    constexpr int N = JHI - JLO + 1;
    constexpr int KW = (JHI + 7) / 8;          // key words a length of this class can reach
    const uint32_t c = top.m1 == 0u ? cand : 0u;   // a lane with a match is done: its second match is the parent of the first
    uint32_t pre[N];                            // queue index of this lane's entry for length JHI - i
    uint32_t total = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int j = JHI - i;
        const unsigned long long m = __ballot((c >> (j - 1)) & 1u);
        pre[i] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        total += (uint32_t)__popcll(m);
    }
#ifdef MATCH_STATS
    if (lane == 0 && total) atomicAdd(&g_match_dbg[2 * (JLO >= 13 ? 0 : JLO >= 9 ? 1 : JLO >= 7 ? 2 : JLO >= 5 ? 3 : 4)], (unsigned long long)total);
#endif
    for (uint32_t w0 = 0; w0 < total; w0 += MQ_CAP) {        // wave-uniform
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = JHI - i;
            if (((c >> (j - 1)) & 1u) && pre[i] - w0 < MQ_CAP) q[pre[i] - w0] = make_uint2(fh[j], lane | ((uint32_t)j << 6));
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t n = min(total - w0, MQ_CAP);
        for (uint32_t e0 = 0; e0 < n; e0 += 64u) {
            const uint32_t e = e0 + lane;
            const bool valid = e < n;
            const uint2 ent = q[valid ? e : 0u];
            const uint32_t f = ent.x, src = ent.y & 63u, j = ent.y >> 6;
            const uint32_t b = valid ? __umulhi(f, v.enb) : 0u;
            const uint4* bp = v.ebk + (size_t)b * 4u;
            const uint4 k01 = bp[0], k23 = bp[1], meta = bp[2];
            uint4 ext = make_uint4(0u, 0u, 0u, 0u);
            if (JHI > 8) ext = bp[3];
            uint64_t Ws[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                if (k < KW) {
                    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)W[k], (int)src, 64);
                    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(W[k] >> 32), (int)src, 64);
                    Ws[k] = (uint64_t)lo | ((uint64_t)hi << 32);
                } else Ws[k] = 0ull;
            }
            // slot t matches when its length is j and its stored bytes (8, or 12 for the long class) equal the text's.
            // All of the bucket is loaded at once (3-4 requests to ONE line): fetching only the slot whose length fits,
            // after a look at the lengths, makes the second load wait for the first and measured 22 % slower.
            const uint32_t lenw = j - 1u;
            const uint32_t sh0 = JLO > 8 ? 0u : 64u - 8u * min(j, 8u);
            const uint64_t key[4] = {(uint64_t)k01.x | ((uint64_t)k01.y << 32), (uint64_t)k01.z | ((uint64_t)k01.w << 32),
                                     (uint64_t)k23.x | ((uint64_t)k23.y << 32), (uint64_t)k23.z | ((uint64_t)k23.w << 32)};
            const uint32_t mt[4] = {meta.x, meta.y, meta.z, meta.w};
            const uint32_t ex[4] = {ext.x, ext.y, ext.z, ext.w};
            uint32_t she = 0, w1 = 0;
            if (JHI > 8) { she = 32u - 8u * min(max(j, 9u) - 8u, 4u); w1 = (uint32_t)Ws[NW > 1 ? 1 : 0]; }
            bool hit = false;
            uint32_t id = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bool m = (mt[t] >> 20) == lenw && ((key[t] ^ Ws[0]) << (sh0 & 63u)) == 0ull;
                if (JHI > 8) m = m && (j <= 8u || ((ex[t] ^ w1) << she) == 0u);
                id = m ? (mt[t] & SAGE_ID_MASK) : id;
                hit = hit || m;
            }
            bool eq = valid && hit;
            // keys of more than 12 bytes: the rest of the key from the planes (a second request, rare)
            if (JHI > 12) {
                if (eq && j > 12u) {
#pragma unroll
                    for (int k = 1; k < NW; ++k) {
                        const int nb = (int)j - 8 * k;
                        if (nb > 0) {
                            const uint64_t kk = v.keys[(size_t)k * v.key_plane + id];
                            const uint32_t sh = 64u - 8u * (uint32_t)min(nb, 8);
                            eq = eq && ((Ws[k] ^ kk) << (sh & 63u)) == 0ull;
                        }
                    }
                    hit = true;   // (a failed verification falls to the exact probe below)
                }
            }
            // rare: the bucket is full (the key may live in the next one), or a 12-byte prefix matched a different key
            if (valid && !eq && (hit || meta.w != SAGE_SLOT_EMPTY)) eq = probe_exact_e<NW>(v, b, Ws, j, id);
            if (valid) q[e].x = eq ? (id | (j << 20)) : 0u;
#ifdef MATCH_STATS
            { const unsigned long long hb = __ballot(valid && eq);
              if (lane == 0 && hb) atomicAdd(&g_match_dbg[2 * (JLO >= 13 ? 0 : JLO >= 9 ? 1 : JLO >= 7 ? 2 : JLO >= 5 ? 3 : 4) + 1], (unsigned long long)__popcll(hb)); }
#endif
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = JHI - i;
            if (((c >> (j - 1)) & 1u) && pre[i] - w0 < MQ_CAP) top.fold(q[pre[i] - w0].x);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// K0b  parent[id] = the longest proper prefix of key id that is a key (id | len << 20); one thread per key, after
// k_vocab_build has completed the tables it probes.  Prefixes of 3+ bytes are looked up in the embedded-key table from
// the longest down, 2-byte prefixes in the direct table, the first byte always exists.
template <int NW>
__global__ __launch_bounds__(256) void k_vocab_parents(VocabBuild B, VocabDev v, uint32_t* __restrict__ parent) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= B.n_keys) return;
    const uint32_t id = B.key_ids ? (uint32_t)B.key_ids[i] : i;
    const uint8_t* k = B.key_bytes + B.key_off[i];
    const uint32_t len = (uint32_t)(B.key_off[i + 1] - B.key_off[i]);
    if (len > B.max_len) { parent[id] = 0u; return; }      // in no table, never matched
    if (len == 1u) { parent[id] = 0u; return; }            // "no second match" (len2 == 0 iff len1 == 1)
    uint64_t W[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) W[w] = 0ull;
    uint32_t fh[8 * NW + 1];
    uint32_t h = SAGE_FNV_OFFSET;
    for (uint32_t j = 0; j < len; ++j) {
        W[j >> 3] |= (uint64_t)k[j] << (8u * (j & 7u));
        h = sage_fnv_step(h, k[j]);
        fh[j + 1] = sage_mix32(h);
    }
    uint32_t out = 0u;
    for (uint32_t L = len - 1u; L >= 3u && out == 0u; --L) {
        uint64_t P[NW];        // the prefix's words (bytes past L zeroed: the probe compares whole words of long keys)
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int nb = (int)L - 8 * w;
            P[w] = nb <= 0 ? 0ull : nb >= 8 ? W[w] : (W[w] & ((1ull << (8 * nb)) - 1ull));
        }
        uint32_t hid;
        if (probe_exact_e<NW>(v, __umulhi(fh[L], v.enb), P, L, hid)) out = hid | (L << 20);
    }
    if (out == 0u && len >= 3u) {
        const uint32_t d2 = v.direct2[(uint32_t)k[0] | ((uint32_t)k[1] << 8)];
        if (d2 != SAGE_SLOT_EMPTY) out = d2 | (2u << 20);
    }
    if (out == 0u) out = v.byte_id[k[0]] | (1u << 20);
    parent[id] = out;
}

// ----------------------------------------------------------------------------------------------------------
// A vocabulary that only LOST keys (SaGe's prune, SaGeVocabBuilder.py:159-192) needs no new matching.  Every key that
// matches at a text position is a prefix of the longest one there, so after the prune the longest match is the nearest
// surviving ancestor-or-self of the old one in the parent chain, and the second longest is that key's parent in the new
// vocabulary.  k_build_remap walks the OLD parent chain once per old id; k_remap_matches is then one streamed pass over
// the match table: two lookups in tables of <= 1 MiB per corpus byte, no hashing, no probes.
//   remap[old id] = new id | L << 20, L = 0: the key itself survives (length and "ends the line" bit stay),
//                   L > 0: its nearest surviving ancestor, of L bytes;  0xFFFFFFFF: none (cannot happen for a matched key)
// ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_build_remap(const int32_t* __restrict__ old_to_new, const uint32_t* __restrict__ parent_old,
                                                     uint32_t n_old, uint32_t* __restrict__ remap) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_old) return;
    uint32_t x = i, L = 0u;
    int32_t to = old_to_new[x];
    while (to < 0) {
        const uint32_t p = parent_old[x];
        if (p == 0u) { remap[i] = 0xFFFFFFFFu; return; }      // a single byte or an unmatched key without a survivor
        x = SAGE_M_ID(p); L = SAGE_M_LEN(p);
        to = old_to_new[x];
    }
    remap[i] = (uint32_t)to | (L << 20);
}

__device__ __forceinline__ uint32_t remap_one(uint32_t m1, const uint32_t* __restrict__ remap, uint32_t n_old, bool& bad) {
    const uint32_t id = SAGE_M_ID(m1);
    const uint32_t r = id < n_old ? remap[id] : 0xFFFFFFFFu;
    if (r == 0xFFFFFFFFu) { bad = true; return 1u << 20; }     // (flagged: the table is not used; the id stays in bounds)
    const uint32_t nid = SAGE_M_ID(r), L = SAGE_M_LEN(r);
    return L == 0u ? (nid | (m1 & 0xFFF00000u)) : (nid | (L << 20));      // a shorter match cannot end the line
}
// four positions per thread and iteration: one 16-byte load, the four table lookups in flight together, one 16-byte store
__global__ __launch_bounds__(256) void k_remap_matches(uint32_t* __restrict__ M, uint64_t n_pos, const uint32_t* __restrict__ remap,
                                                       uint32_t n_old, uint32_t* __restrict__ flags) {
    bool bad = false;
    const uint64_t n4 = n_pos / 4u;
    uint4* M4 = reinterpret_cast<uint4*>(M);
    for (uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x; q < n4; q += (uint64_t)gridDim.x * 256u) {
        const uint4 a = M4[q];
        M4[q] = make_uint4(remap_one(a.x, remap, n_old, bad), remap_one(a.y, remap, n_old, bad),
                           remap_one(a.z, remap, n_old, bad), remap_one(a.w, remap, n_old, bad));
    }
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)(n_pos - 4u * n4)) {
        const uint64_t p = 4u * n4 + threadIdx.x;
        M[p] = remap_one(M[p], remap, n_old, bad);
    }
    if (bad) flags[FLAG_SET_OVERFLOW] = 1;
}

// NW = 8-byte key words compared, ML = longest length hashed and probed (max_len <= ML <= 8 * NW): the instantiation
// for max_len 17 hashes 16 prefix lengths, not the 23 a 24-byte window would allow.
template <int NW, int ML = 8 * NW, bool QUEUED = true>
__global__ __launch_bounds__(1024) void k_match_lds(const uint8_t* __restrict__ corpus, const uint8_t* __restrict__ rem,
                                                    int64_t pos0, uint32_t count, VocabDev v, uint32_t* __restrict__ M) {
    static_assert(ML <= 8 * NW && ML > 8 * (NW - 1), "ML must need exactly NW key words");
    constexpr int WDW = (64 + ML + 8) / 4 + 1;   // dwords of text one wave stages for its 64 positions (<= 27)
    __shared__ uint32_t bloom[SAGE_BLOOM_BITS / 32];
    __shared__ uint32_t text_all[16 * 32];
    __shared__ uint2 mq_all[QUEUED ? 16 * MQ_CAP : 1];
    const uint32_t tid = threadIdx.x;
    {
        const uint4* src4 = reinterpret_cast<const uint4*>(v.bloom);
        uint4* dst4 = reinterpret_cast<uint4*>(bloom);
        for (uint32_t i = tid; i < SAGE_BLOOM_BITS / 128; i += blockDim.x) dst4[i] = src4[i];
    }
    __syncthreads();
    // From here on the 16 waves of the block are independent: each sweeps its own 64-byte tiles and stages its text
    // window in its own LDS strip, so no block barrier puts the waves' memory phases in lockstep.  The next tile's
    // text and line-remainder bytes are fetched before the current tile is processed.
    const uint32_t lane = tid & 63u;
    uint32_t* text = text_all + (tid >> 6) * 32u;
    const uint32_t n_tiles = (count + 63u) / 64u;
    const uint32_t stride = gridDim.x * (blockDim.x >> 6);
    uint32_t tile = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
    uint32_t tw_n = 0, rm_n = 0;
    auto fetch = [&](uint32_t tl) {
        const int64_t b0 = pos0 + (int64_t)tl * 64;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(corpus + (b0 & ~(int64_t)3));
        if (lane < (uint32_t)WDW) tw_n = __builtin_nontemporal_load(src + lane);
        rm_n = (tl * 64u + lane < count) ? (uint32_t)__builtin_nontemporal_load(rem + b0 + lane) : 0u;
    };
    if (tile < n_tiles) fetch(tile);
    for (; tile < n_tiles; tile += stride) {
        const int64_t blk0 = pos0 + (int64_t)tile * 64;
        const uint32_t tw = tw_n, rm = rm_n;
        if (tile + stride < n_tiles) fetch(tile + stride);
        __builtin_amdgcn_wave_barrier();
        if (lane < (uint32_t)WDW) text[lane] = tw;
        __builtin_amdgcn_wave_barrier();
        const uint32_t rel = tile * 64u + lane;
        // a lane past the end has rm = 0, hence no candidates; in the queued form it stays to serve cross-lane reads
        if (!QUEUED && rel >= count) continue;
        const uint32_t o = (uint32_t)(blk0 & 3) + lane;
        const uint32_t wi = o >> 2, sh = (o & 3u) * 8u;
        uint32_t T[2 * NW];
        {
            uint32_t d[2 * NW + 1];
#pragma unroll
            for (int i = 0; i <= 2 * NW; ++i) d[i] = text[wi + i];
#pragma unroll
            for (int i = 0; i < 2 * NW; ++i) T[i] = __funnelshift_r(d[i], d[i + 1], sh);
        }
        const uint32_t lim = min(rm, (uint32_t)v.max_len);
        uint32_t d2 = SAGE_SLOT_EMPTY;
        uint32_t fh[ML + 1];
        uint32_t miss = 0;   // bit (j-1) set when the filter rules length j out
        using PM = typename std::conditional<(NW <= 2), uint16_t, uint32_t>::type;
        uint32_t pm = 0xFFFFFFFFu;   // lengths >= 4 that some key with these first 4 bytes has (superset: slots are shared)
        uint32_t pm8 = 0xFFFFFFFFu;  // lengths >= 8 that some key with these first 8 bytes has; lengths < 8 pass
        uint32_t h = sage_fnv_step(SAGE_FNV_OFFSET, T[0] & 0xFFu);
#pragma unroll
        for (int j = 2; j <= ML; ++j) {
            const uint32_t b = (T[(j - 1) >> 2] >> (8 * ((j - 1) & 3))) & 0xFFu;
            h = sage_fnv_step(h, b);
            const uint32_t f = sage_mix32(h);
            if (j == 4) pm = reinterpret_cast<const PM*>(v.pmask)[f & (SAGE_PMASK_SLOTS - 1u)];   // issued early, used last
            fh[j] = f;
            const uint32_t mk = sage_bloom_mask(f);
            const uint32_t lack = (bloom[sage_bloom_word(f)] & mk) ^ mk;      // 0 when both bits are present
            miss |= min(lack, 1u) << (j - 1);
        }
        // candidate lengths: pass the filter, 2 <= j <= lim
        // Requests to the vector L1 are what this kernel is short of, so the two lookups that most lanes can do without are
        // made late and only by the lanes that need them (-5 %): the 8-byte prefix table by lanes that still have a
        // candidate of 8+ bytes, the 2-byte table (below) by lanes with fewer than two longer matches.
        if (ML > 8) {
            const uint32_t c0 = ~miss & (pm | 0x6u) & ((lim >= 32u ? 0xFFFFFFFFu : ((1u << lim) - 1u)) & ~1u) & (ML >= 32 ? 0xFFFFFFFFu : ((1u << ML) - 1u));
            if (c0 >> 7) pm8 = reinterpret_cast<const PM*>(v.pmask8)[fh[8] & (SAGE_PMASK8_SLOTS - 1u)] | 0x7Fu;
        }
        const uint32_t cand = ~miss & (pm | 0x6u) & pm8 & ((lim >= 32u ? 0xFFFFFFFFu : ((1u << lim) - 1u)) & ~1u) & (ML >= 32 ? 0xFFFFFFFFu : ((1u << ML) - 1u));
        uint64_t W[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) W[w] = (uint64_t)T[2 * w] | ((uint64_t)T[2 * w + 1] << 32);
        Top2 top{0u, 0u};
        // chunks from the longest lengths down.  A chunk runs if ANY lane of the wave still needs it (wave-uniform
        // branch); lanes that do not take part are masked through their candidate bits, not through EXEC.
        if constexpr (QUEUED) {
            uint2* q = mq_all + (tid >> 6) * MQ_CAP;
if constexpr (ML >= 9) {
                // Five length classes, longest first.  A lane stops at its first hit, so finer classes mean fewer wasted
                // probes and coarser ones fewer dependent rounds per tile; both cost.  Measured with the first-hit rule, ms per
                // 152 MB: 2 classes (7+, 3..6) 7.21, 3 (9+, 5..8, 3..4) 6.02, 4 (11+, 7..10, 5..6, 3..4) 5.93, these 5 5.74,
                // 7 (the short lengths one by one) 6.23, 9 6.77.
                if constexpr (ML >= 13) { if (__any((cand >> 12) != 0u)) probe_queued<NW, 13, ML, ML>(v, cand, fh, W, top, q, lane); }
                if (__any(((cand >> 8) & 0xFu) != 0u)) probe_queued<NW, 9, (ML < 12 ? ML : 12), ML>(v, cand, fh, W, top, q, lane);
            }
            probe_queued<NW, 7, 8, ML>(v, cand, fh, W, top, q, lane);
            probe_queued<NW, 5, 6, ML>(v, cand, fh, W, top, q, lane);
            probe_queued<NW, 3, 4, ML>(v, cand, fh, W, top, q, lane);
        } else {
            if constexpr (ML >= 9) probe_sparse<NW, 9, ML, ML>(v, cand, fh, W, top);
            const uint32_t c1 = top.m2 == 0u ? cand : 0u;
            if (__any((c1 & 0xF0u) != 0u)) probe_chunk<NW, 5, 8, ML>(v, c1, fh, W, top);
            const uint32_t c2 = top.m2 == 0u ? cand : 0u;
            if (__any((c2 & 0x0Cu) != 0u)) probe_chunk<NW, 3, 4, ML>(v, c2, fh, W, top);
        }
        if constexpr (QUEUED) {
            // The queued classes stop at a lane's first hit: the longest match is all the table keeps (the second longest
            // match at a position is the parent of the longest, which its one reader looks up in k_vocab_parents' table).
            if (top.m1 == 0u) {
                // no key of 3+ bytes: 2-byte keys need no hash (a 256 KiB table indexed by the two bytes), and the single
                // byte always matches -- the longest match when nothing else hit, the second when the 2-byte key did
                if (lim >= 2u) d2 = v.direct2[T[0] & 0xFFFFu];
                top.fold((lim >= 2u && d2 != SAGE_SLOT_EMPTY) ? (d2 | (2u << 20)) : 0u);
                top.fold(v.byte_id[T[0] & 0xFFu] | (1u << 20));
            }
        } else {
            // 2-byte keys need no hash: a 256 KiB table indexed by the two bytes
            if (top.m2 == 0u && lim >= 2u) d2 = v.direct2[T[0] & 0xFFFFu];
            top.fold((lim >= 2u && d2 != SAGE_SLOT_EMPTY) ? (d2 | (2u << 20)) : 0u);
            // the single byte always matches: it is the longest match when nothing else hit, the second when one did
            if (top.m2 == 0u) top.fold(v.byte_id[T[0] & 0xFFu] | (1u << 20));
        }
        // (more than one match may have been folded in -- a chunk probes all its lengths -- m1 is the longest in any case)
        // streamed once: keep it out of the way of the table and key planes in L2
        const uint32_t ends1 = SAGE_M_LEN(top.m1) == rm ? 0x80000000u : 0u;
        if (rel < count) __builtin_nontemporal_store(top.m1 | ends1, M + rel);
    }
}

// ----------------------------------------------------------------------------------------------------------
// K2a  base chain: one thread per line follows p -> p + len1(p) (model.py:89-99).  Pass 1 counts tokens, pass 2
// (after an exclusive scan of the counts) writes the compact token arrays.
// ----------------------------------------------------------------------------------------------------------
// A walk is a chain of dependent loads (~0.3 us each), so a launch lasts as long as its longest line: 1,600 tokens in a
// 9.5 KB line of the bench corpus outlast the other 300,000 lines of 500 bytes (1.61 ms for the two walks; 1.21 with no
// line above 2 KB), and a 1 MB line of real text would take tens of milliseconds.  Lines of more than SAGE_LONG_LINE bytes
// are therefore listed once per corpus (k_long_lines) and walked by the FIRST blocks of each walk launch, a WAVE per
// line: it stages 512 entries of the match table in LDS at a time (coalesced loads) and one lane follows the chain there,
// at LDS latency, beside the line-per-thread blocks, which skip the listed lines.  n_long = 0 switches the scheme off.
#define SAGE_LONG_LINE 2048
#define SAGE_LONG_CHUNK 512u
__global__ __launch_bounds__(256) void k_long_lines(const int64_t* __restrict__ line_off, int64_t n_lines,
                                                    uint32_t* __restrict__ list /* or nullptr: count only */, uint32_t* __restrict__ count) {
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= n_lines) return;
    if (line_off[l + 1] - line_off[l] > SAGE_LONG_LINE) {
        const uint32_t at = atomicAdd(count, 1u);
        if (list) list[at] = (uint32_t)l;
    }
}
// blocks of a walk launch that take listed lines (4 waves = 4 lines each)
static inline __host__ __device__ uint32_t long_blocks(uint32_t n_long) { return (n_long + 3u) / 4u; }
// the listed line of this wave, relative to line0; false: none in this batch
__device__ __forceinline__ bool long_line_of_wave(int64_t line0, uint32_t n_lines, const uint32_t* __restrict__ long_list,
                                                  uint32_t n_long, uint32_t& l) {
    const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (i >= n_long) return false;
    const int64_t g = (int64_t)long_list[i];
    if (g < line0 || g >= line0 + (int64_t)n_lines) return false;
    l = (uint32_t)(g - line0);
    return true;
}
// One chunk of a listed line's chain: the wave stages M[p, p + SAGE_LONG_CHUNK) (clipped to the line's end e) and lane 0
// follows the chain while it stays inside; `each(m, p)` sees every token's entry.  Returns the position reached.
template <class F>
__device__ __forceinline__ uint32_t long_chunk(const uint32_t* __restrict__ M, uint32_t* buf, uint32_t p, uint32_t e, F&& each) {
    const uint32_t lane = threadIdx.x & 63u, base = p;
    __builtin_amdgcn_wave_barrier();                   // (lane 0 is done with the previous chunk)
    for (uint32_t j = lane; j < SAGE_LONG_CHUNK && base + j < e; j += 64u) buf[j] = M[base + j];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0u) {
        while (p < e && p - base < SAGE_LONG_CHUNK) {
            const uint32_t m = buf[p - base];
            each(m, p);
            p += SAGE_M_LEN(m);
        }
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
}

__global__ __launch_bounds__(256) void k_walk_count(const int64_t* __restrict__ line_off, int64_t line0,
                                                    uint32_t n_lines, int64_t pos0, const uint32_t* __restrict__ M,
                                                    uint32_t* __restrict__ ntok, uint32_t* __restrict__ nocc,
                                                    uint32_t* __restrict__ flags,
                                                    const uint32_t* __restrict__ long_list, uint32_t n_long) {
    __shared__ uint32_t buf_all[4 * SAGE_LONG_CHUNK];
    const uint32_t nlb = long_blocks(n_long);
    if (blockIdx.x < nlb) {      // a listed line per wave
        uint32_t ll;
        if (!long_line_of_wave(line0, n_lines, long_list, n_long, ll)) return;
        uint32_t* buf = buf_all + (threadIdx.x >> 6) * SAGE_LONG_CHUNK;
        uint32_t q = (uint32_t)(line_off[line0 + ll] - pos0);
        const uint32_t qe = (uint32_t)(line_off[line0 + ll + 1] - pos0);
        uint32_t cc = 0, mm = 0;
        while (q < qe) q = long_chunk(M, buf, q, qe, [&](uint32_t m, uint32_t) { ++cc; mm += SAGE_M_LEN(m) > 1u; });
        if ((threadIdx.x & 63u) == 0u) { ntok[ll] = cc; nocc[ll] = mm; }
        return;
    }
    uint32_t l = (blockIdx.x - nlb) * 256u + threadIdx.x;
    if (l >= n_lines) return;
    if (n_long != 0u && line_off[line0 + l + 1] - line_off[line0 + l] > SAGE_LONG_LINE) return;     // a listed line
    uint32_t p = (uint32_t)(line_off[line0 + l] - pos0), e = (uint32_t)(line_off[line0 + l + 1] - pos0);
    if (p == e) flags[FLAG_EMPTY_LINE] = 1;
    uint32_t c = 0, mb = 0;
    while (p < e) {
        const uint32_t len = SAGE_M_LEN(M[p]);
        p += len; ++c; mb += len > 1u;
    }
    ntok[l] = c;
    nocc[l] = mb;   // tokens wider than one byte = ablated occurrences (model.py:170)
}

// A lane that writes its own run of consecutive 32-bit words gathers four and stores 16 aligned bytes at a time:
// 4-byte stores from 64 lanes to 64 different lines reach HBM as partial-line writes (8x the bytes, measured).
struct GroupedWriter {
    uint32_t* dst;
    uint32_t first;      // first index this lane writes
    uint32_t buf[4];
    __device__ __forceinline__ void init(uint32_t* d, uint32_t f) { dst = d; first = f; buf[0] = buf[1] = buf[2] = buf[3] = 0; }
    __device__ __forceinline__ void put(uint32_t i, uint32_t v) {
        const uint32_t k4 = i & 3u;
        buf[0] = k4 == 0u ? v : buf[0]; buf[1] = k4 == 1u ? v : buf[1];
        buf[2] = k4 == 2u ? v : buf[2]; buf[3] = k4 == 3u ? v : buf[3];
        if (k4 == 3u) {
            if (i - first >= 3u) *reinterpret_cast<uint4*>(dst + (i - 3u)) = make_uint4(buf[0], buf[1], buf[2], buf[3]);
            else for (uint32_t q = first; q <= i; ++q) dst[q] = pick(q);     // the run's first, partial group
        }
    }
    __device__ __forceinline__ uint32_t pick(uint32_t q) const {
        const uint32_t k4 = q & 3u;
        return k4 == 0u ? buf[0] : k4 == 1u ? buf[1] : k4 == 2u ? buf[2] : buf[3];
    }
    __device__ __forceinline__ void finish(uint32_t end) {   // end = one past the last index written
        if ((end & 3u) != 0u && end > first) {
            const uint32_t g0 = max(end & ~3u, first);
            for (uint32_t q = g0; q < end; ++q) dst[q] = pick(q);
        }
    }
};

template <bool EXPORT>
__global__ __launch_bounds__(256) void k_walk_write(const int64_t* __restrict__ line_off, int64_t line0,
                                                    uint32_t n_lines, int64_t pos0, const uint32_t* __restrict__ M,
                                                    const uint32_t* __restrict__ tok_off,
                                                    const uint32_t* __restrict__ occ_off,
                                                    uint32_t* __restrict__ tok_w0, uint32_t* __restrict__ occ_pos,
                                                    uint32_t* __restrict__ occ_tok,
                                                    int32_t* __restrict__ ex_ids, int32_t* __restrict__ ex_starts,
                                                    int32_t* __restrict__ ex_widths,
                                                    const uint32_t* __restrict__ long_list, uint32_t n_long) {
    __shared__ uint32_t buf_all[4 * SAGE_LONG_CHUNK];
    const uint32_t nlb = long_blocks(n_long);
    if (blockIdx.x < nlb) {      // a listed line per wave: the same words, written one by one by the lane that follows the chain
        uint32_t ll;
        if (!long_line_of_wave(line0, n_lines, long_list, n_long, ll)) return;
        uint32_t* buf = buf_all + (threadIdx.x >> 6) * SAGE_LONG_CHUNK;
        const uint32_t q_begin = (uint32_t)(line_off[line0 + ll] - pos0);
        uint32_t q = q_begin;
        const uint32_t qe = (uint32_t)(line_off[line0 + ll + 1] - pos0);
        uint32_t tt = tok_off[ll], cc = 0;
        const uint32_t ntt = tok_off[ll + 1] - tt;
        uint32_t oo = EXPORT ? 0u : occ_off[ll];
        while (q < qe)
            q = long_chunk(M, buf, q, qe, [&](uint32_t m, uint32_t at) {
                if (EXPORT) {
                    ex_ids[tt] = (int32_t)SAGE_M_ID(m);
                    ex_starts[tt] = (int32_t)(at - q_begin);
                    ex_widths[tt] = (int32_t)SAGE_M_LEN(m);
                } else {
                    const uint32_t after = ntt - 1u - cc;
                    tok_w0[tt] = SAGE_M_ID(m) | ((cc < 3u ? cc : 3u) << 20) | ((after < 3u ? after : 3u) << 22) | (SAGE_M_LEN(m) << 24);
                    if (SAGE_M_LEN(m) > 1u) { occ_tok[oo] = tt; occ_pos[oo] = at; ++oo; }
                }
                ++tt; ++cc;
            });
        return;
    }
    uint32_t l = (blockIdx.x - nlb) * 256u + threadIdx.x;
    if (l >= n_lines) return;
    if (n_long != 0u && line_off[line0 + l + 1] - line_off[line0 + l] > SAGE_LONG_LINE) return;     // a listed line
    const uint32_t p_begin = (uint32_t)(line_off[line0 + l] - pos0);
    uint32_t p = p_begin, e = (uint32_t)(line_off[line0 + l + 1] - pos0);
    uint32_t t = tok_off[l], c = 0;
    const uint32_t nt = tok_off[l + 1] - t;
    uint32_t o = EXPORT ? 0u : occ_off[l];
    GroupedWriter gw_tok, gw_occ, gw_pos;
    if (!EXPORT) { gw_tok.init(tok_w0, t); gw_occ.init(occ_tok, o); gw_pos.init(occ_pos, o); }
    while (p < e) {
        const uint32_t m = M[p];
        if (EXPORT) {
            ex_ids[t] = (int32_t)SAGE_M_ID(m);
            ex_starts[t] = (int32_t)(p - p_begin);
            ex_widths[t] = (int32_t)SAGE_M_LEN(m);
        } else {
            const uint32_t after = nt - 1u - c;
            gw_tok.put(t, SAGE_M_ID(m) | ((c < 3u ? c : 3u) << 20) | ((after < 3u ? after : 3u) << 22) | (SAGE_M_LEN(m) << 24));
            if (SAGE_M_LEN(m) > 1u) { gw_occ.put(o, t); gw_pos.put(o, p); ++o; }   // compact list of the tokens that get ablated
        }
        p += SAGE_M_LEN(m);
        ++t; ++c;
    }
    if (!EXPORT) { gw_tok.finish(t); gw_occ.finish(o); gw_pos.finish(o); }
}

// ----------------------------------------------------------------------------------------------------------
// Exclusive scan of u32 (tokens per line -> token offset of each line), three small kernels.
// ----------------------------------------------------------------------------------------------------------
#define SCAN_ITEMS 8
#define SCAN_TILE (256 * SCAN_ITEMS)

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* total, uint32_t* lds /*[8]*/) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += n;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t s = lds[w];
        if ((uint32_t)w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(256) void k_scan_tile_sums(const uint32_t* __restrict__ in, uint32_t n,
                                                        uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t lds[8];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) s += in[base + i];
    uint32_t tot;
    block_exclusive_scan_256(s, &tot, lds);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_scan_top(uint32_t* __restrict__ tile_sums, uint32_t n_tiles,
                                                  uint64_t* __restrict__ grand_total) {
    __shared__ uint32_t lds[8];
    __shared__ unsigned long long wide[4];
    // The grand total is summed in 64 bits on its own: the host decides from it whether the 32-bit offsets below can be
    // trusted at all (a batch with 2^32 or more items wraps them, and must be cut smaller before anything reads them).
    unsigned long long mine = 0;
    for (uint32_t i = threadIdx.x; i < n_tiles; i += 256) mine += tile_sums[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63u) == 0) wide[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) *grand_total = wide[0] + wide[1] + wide[2] + wide[3];
    uint32_t carry = 0;
    for (uint32_t b = 0; b < n_tiles; b += 256) {
        const uint32_t i = b + threadIdx.x;
        uint32_t v = i < n_tiles ? tile_sums[i] : 0u, tot;
        uint32_t ex = block_exclusive_scan_256(v, &tot, lds);
        if (i < n_tiles) tile_sums[i] = carry + ex;
        carry += tot;
    }
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* __restrict__ in, uint32_t n,
                                                    const uint32_t* __restrict__ tile_sums,
                                                    uint32_t* __restrict__ out /* n+1 */) {
    __shared__ uint32_t lds[8];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = base + i < n ? in[base + i] : 0u; s += v[i]; }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan_256(s, &tot, lds) + tile_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
        if (base + i + 1 == n) out[n] = ex;
    }
}

__global__ void k_gather_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t n,
                             uint32_t* __restrict__ dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ void k_widen_counts(const uint32_t* __restrict__ src, uint32_t n, int64_t* __restrict__ dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (int64_t)src[i];
}

// What the ablation walk (walk_occurrence, sage_kernels2.h) reads: the match table M and the compact token arrays.
struct WalkArrays {
    const uint32_t* M;        // longest match per position: id | len << 20 | (ends the line) << 31
    uint32_t n_pos;           // positions of the batch (entries of M)
    const uint32_t* tok_w0;   // base tokens: id | min(index in line, 3) << 20 | min(tokens after it, 3) << 22 | width << 24
    const uint32_t* occ_pos;  // byte position of every ablated occurrence
    const uint32_t* parent;   // [ids] longest proper prefix of a key that is a key, as id | len << 20 (k_vocab_parents)
};

// ----------------------------------------------------------------------------------------------------------
// K2b  counts: one thread per base token.  ablated_sizes histogram (model.py:208-209), occurrences per id (the
// keys of the loss dict), and the distinct (ablated, target, context) keys per reference segment that sum to
// total_triples (utils.py:237,245).  do_triples emits both orders of every position pair, so the set stores
// the unordered key (a, min, max) and a new key weighs 2 (1 when target == context).
// v1 of the distinct count: one open-addressed u64 set per segment in HBM, claimed with 64-bit CAS.
// ----------------------------------------------------------------------------------------------------------
struct SegPlan {
    const uint32_t* seg_tok_off;  // [n_seg + 1] token offset of each segment inside the batch
    const uint64_t* set_base;     // [n_seg] first slot of the segment's set
    const uint64_t* set_mask;     // [n_seg] capacity - 1 (capacity is a power of two)
    uint32_t n_seg;
};

__device__ __forceinline__ uint32_t find_segment(const SegPlan& S, uint32_t t) {
    uint32_t lo = 0, hi = S.n_seg;  // largest s with seg_tok_off[s] <= t
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (S.seg_tok_off[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ uint32_t set_insert(unsigned long long* __restrict__ set, uint64_t base, uint64_t mask,
                                               uint64_t key, uint32_t* __restrict__ flags) {
    uint64_t s = sage_mix64(key) & mask;
    for (uint64_t probes = 0; probes <= mask; ++probes) {
        unsigned long long* slot = set + base + s;
        unsigned long long cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return 0;
        if (cur == SAGE_SET_EMPTY) {
            cur = atomicCAS(slot, (unsigned long long)SAGE_SET_EMPTY, (unsigned long long)key);
            if (cur == SAGE_SET_EMPTY) return 1;
            if (cur == key) return 0;
        }
        s = (s + 1) & mask;
    }
    flags[FLAG_SET_OVERFLOW] = 1;
    return 0;
}

// ----------------------------------------------------------------------------------------------------------
// log(sigmoid(x)) in float64 (utils.py:136)
// ----------------------------------------------------------------------------------------------------------
// The polynomial coefficients are read from constant memory: uniform loads, so they sit in SGPRs and feed v_fma_f64 as
// its scalar operand.  As literals every Horner step costs two v_mov_b32 for the coefficient on top of its fma
// (108 VALU instructions per evaluation, 39 of them moves; 70 this way), and k_logsig's dot loop is VALU-bound.
__constant__ double LS_C[24] = {
    1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
    1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0,                                                  // exp: r^13 .. r^3
    1.0 / 23.0, 1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0,
    1.0 / 3.0,                                                                                        // atanh: s^22 .. s^2
    -6.93147180369123816490e-01, -1.90821492927058770002e-10};                                        // -ln2, high and low
__device__ __forceinline__ double log_sigmoid(double x) {
    // log(expit(x)) of utils.py:136 in its overflow-free form min(x,0) - log1p(exp(-|x|)), with both functions written
    // out (the library's exp and log1p carry special-case paths and ~2x the instructions; k_logsig and k_pair_cache are
    // bound by VALU issue).  Relative error <= 7e-16 against libm on 2e7 points over |x| < 800 and around 0
    // (tools/check_log_sigmoid.cpp: the same arithmetic compiled for the host).  NaN stays NaN, -inf gives -inf.
    double t = -fabs(x);
    t = t < -750.0 ? -750.0 : t;                                   // exp underflows to 0 below (NaN passes through)
    // u = exp(t) = 2^n * exp(r), |r| <= ln2/2
    const double n = rint(t * 1.4426950408889634);
    double r = fma(n, LS_C[22], t);
    r = fma(n, LS_C[23], r);
    double p = LS_C[0];
#pragma unroll
    for (int i = 1; i <= 10; ++i) p = fma(p, r, LS_C[i]);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double u = ldexp(p, (int)n);
    // log1p(u), 0 <= u <= 1: z = 1 + u with its rounding error e kept; z in (sqrt2, 2] is halved; log(m) = 2 atanh(s)
    const double z = 1.0 + u;
    const double e = u - (z - 1.0);
    const bool big = z > 1.4142135623730951;
    const double m = big ? 0.5 * z : z;
    const double f = m - 1.0, d = m + 1.0;
    double y = __builtin_amdgcn_rcp(d);                             // two Newton steps take any 24-bit start to 2^-96
    y = y * fma(-d, y, 2.0);
    y = y * fma(-d, y, 2.0);
    double s = f * y;
    s = fma(fma(-s, d, f), y, s);
    const double s2 = s * s;
    double q = LS_C[11];
#pragma unroll
    for (int i = 12; i <= 21; ++i) q = fma(q, s2, LS_C[i]);
    double lg = fma(2.0 * s * s2, q, 2.0 * s);
    lg = lg + (big ? 6.93147180559945286227e-01 : 0.0);
    lg = lg + (big ? 0.5 * e : e);
    return fmin(x, 0.0) - lg;
}

