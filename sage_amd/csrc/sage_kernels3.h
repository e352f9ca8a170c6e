// Distinct (ablated, target, context) keys per count segment (utils.py:237,245 `len(triples)`): key generation from the
// occurrence descriptors, the per-segment HBM sets (count mode 1) and the default partition + LDS dedupe (further down).
#pragma once
#include "sage_kernels2.h"

// Calls f(lo, hi) for every position pair of the occurrence's two windows (model.py:115-136 via descriptors).
// d = the 16 descriptor words in registers; long windows stream their ids from the pool.
// POOLED_TOO = false: the caller has already set pooled occurrences aside (k_key_partition_long handles them).
template <bool POOLED_TOO, class F>
__device__ __forceinline__ void keys_from_desc(const uint32_t (&d)[16], const uint32_t* __restrict__ pool,
                                               const uint32_t* __restrict__ tok_w0, uint32_t t, F&& f) {
    const Win w = win_decode(d[0], d[2], d[3]);
    const uint32_t pl = w.pl, k = w.k, m = w.m, pr = w.pr, L = w.L;
    if (!POOLED_TOO && w.pooled) return;
    if (!w.pooled) {
#pragma unroll
        for (int j = 1; j < SAGE_DESC_MAXWIN; ++j) {
            if ((uint32_t)j < L) {
                const bool c = (uint32_t)j >= pl && (uint32_t)j < pl + k;
                const bool c1 = (uint32_t)(j - 1) >= pl && (uint32_t)(j - 1) < pl + k;
                if (c || c1) f(min(d[1 + j], d[j]), max(d[1 + j], d[j]));
                if (j >= 2) {
                    const bool c2 = (uint32_t)(j - 2) >= pl && (uint32_t)(j - 2) < pl + k;
                    if (c || c2) f(min(d[1 + j], d[j - 1]), max(d[1 + j], d[j - 1]));
                }
            }
        }
    } else if (POOLED_TOO) {
        const uint32_t* ids = pool + d[1];
        uint32_t p1 = 0, p2 = 0;
        for (uint32_t j = 0; j < L; ++j) {
            const uint32_t id = ids[j];
            const bool c = j >= pl && j < pl + k;
            if (j >= 1) { const bool c1 = (j - 1) >= pl && (j - 1) < pl + k; if (c || c1) f(min(id, p1), max(id, p1)); }
            if (j >= 2) { const bool c2 = (j - 2) >= pl && (j - 2) < pl + k; if (c || c2) f(min(id, p2), max(id, p2)); }
            p2 = p1; p1 = id;
        }
    }
    const uint32_t w0 = t - pl, w1 = t + m + pr;  // base window tokens [w0, w1), centre [t, t+m)
    uint32_t p1 = 0, p2 = 0;
    for (uint32_t i = w0; i < w1; ++i) {
        const uint32_t id = SAGE_T_ID(tok_w0[i]);
        const bool c = i >= t && i < t + m;
        if (i >= w0 + 1) { const bool c1 = (i - 1) >= t && (i - 1) < t + m; if (c || c1) f(min(id, p1), max(id, p1)); }
        if (i >= w0 + 2) { const bool c2 = (i - 2) >= t && (i - 2) < t + m; if (c || c2) f(min(id, p2), max(id, p2)); }
        p2 = p1; p1 = id;
    }
}

// The inline occurrence as the partition kernel holds it between its two phases: descriptor words and the first
// BASE_PF ids of the base window [t - pl, t + m + pr), fetched in ONE batch of loads (the loop form pays one memory
// latency per window token).
#define BASE_PF 8
struct InlineOcc {
    uint32_t d[16];
    uint32_t base[BASE_PF];
    uint32_t t, a;
};
__device__ __forceinline__ void load_inline(const uint4* __restrict__ desc, const uint32_t* __restrict__ occ_tok,
                                            const uint32_t* __restrict__ tok_w0, uint32_t o, InlineOcc& x);
template <class F>
__device__ __forceinline__ void keys_from_inline(const InlineOcc& x, const uint32_t* __restrict__ tok_w0, F&& f) {
    const Win w = win_decode(x.d[0], x.d[2], x.d[3]);
    const uint32_t pl = w.pl, k = w.k, m = w.m, pr = w.pr, L = w.L;
#pragma unroll
    for (int j = 1; j < SAGE_DESC_MAXWIN; ++j) {
        if ((uint32_t)j < L) {
            const bool c = (uint32_t)j >= pl && (uint32_t)j < pl + k;
            const bool c1 = (uint32_t)(j - 1) >= pl && (uint32_t)(j - 1) < pl + k;
            if (c || c1) f(min(x.d[1 + j], x.d[j]), max(x.d[1 + j], x.d[j]));
            if (j >= 2) {
                const bool c2 = (uint32_t)(j - 2) >= pl && (uint32_t)(j - 2) < pl + k;
                if (c || c2) f(min(x.d[1 + j], x.d[j - 1]), max(x.d[1 + j], x.d[j - 1]));
            }
        }
    }
    const uint32_t nb = pl + m + pr;      // base window tokens, centre [pl, pl + m)
    uint32_t p1 = 0, p2 = 0;
#pragma unroll
    for (int i = 0; i < BASE_PF; ++i) {
        if ((uint32_t)i < nb) {
            const uint32_t id = x.base[i];
            const bool c = (uint32_t)i >= pl && (uint32_t)i < pl + m;
            if (i >= 1) { const bool c1 = (uint32_t)(i - 1) >= pl && (uint32_t)(i - 1) < pl + m; if (c || c1) f(min(id, p1), max(id, p1)); }
            if (i >= 2) { const bool c2 = (uint32_t)(i - 2) >= pl && (uint32_t)(i - 2) < pl + m; if (c || c2) f(min(id, p2), max(id, p2)); }
            p2 = p1; p1 = id;
        }
    }
    for (uint32_t i = BASE_PF; i < nb; ++i) {        // a base span of more than 4 tokens: rare
        const uint32_t id = SAGE_T_ID(tok_w0[x.t - pl + i]);
        const bool c = i >= pl && i < pl + m;
        { const bool c1 = (i - 1) >= pl && (i - 1) < pl + m; if (c || c1) f(min(id, p1), max(id, p1)); }
        { const bool c2 = (i - 2) >= pl && (i - 2) < pl + m; if (c || c2) f(min(id, p2), max(id, p2)); }
        p2 = p1; p1 = id;
    }
}

__device__ __forceinline__ void load_desc(const uint4* __restrict__ desc, uint32_t o, uint32_t (&d)[16]) {
    const uint4 a = desc[(size_t)o * 4], b = desc[(size_t)o * 4 + 1], c = desc[(size_t)o * 4 + 2], e = desc[(size_t)o * 4 + 3];
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
    d[8] = c.x; d[9] = c.y; d[10] = c.z; d[11] = c.w; d[12] = e.x; d[13] = e.y; d[14] = e.z; d[15] = e.w;
}

__device__ __forceinline__ void load_inline(const uint4* __restrict__ desc, const uint32_t* __restrict__ occ_tok,
                                            const uint32_t* __restrict__ tok_w0, uint32_t o, InlineOcc& x) {
    load_desc(desc, o, x.d);
    x.t = occ_tok[o];
    const uint32_t pl = SAGE_DH_PL(x.d[0]);
    const uint32_t nb = (x.d[0] & SAGE_DH_OVER) ? 0u : pl + SAGE_DH_M(x.d[0]) + SAGE_DH_PR(x.d[0]);
#pragma unroll
    for (int i = 0; i < BASE_PF; ++i) x.base[i] = (uint32_t)i < nb ? SAGE_T_ID(tok_w0[x.t - pl + (uint32_t)i]) : 0u;
    // (pl <= 2; a select chain, not x.base[pl]: one dynamic index would put the whole struct into scratch memory)
    x.a = (x.d[0] & SAGE_DH_OVER) ? SAGE_T_ID(tok_w0[x.t]) : (pl == 0u ? x.base[0] : pl == 1u ? x.base[1] : x.base[2]);
}

__device__ __forceinline__ uint64_t make_key(uint32_t a, uint32_t lo, uint32_t hi) {
    return ((uint64_t)a << 40) | ((uint64_t)lo << 20) | hi;
}

// ---- v1 back end (count mode 1): per-segment u64 open-addressed sets in HBM, one CAS per key ---------------
__global__ __launch_bounds__(256) void k_count_sets(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                                    const uint32_t* __restrict__ occ_tok,
                                                    const uint32_t* __restrict__ tok_w0, uint32_t n_occ, SegPlan S,
                                                    unsigned long long* __restrict__ set,
                                                    unsigned long long* __restrict__ stats, uint32_t* __restrict__ flags) {
    unsigned long long n_distinct = 0;
    for (uint32_t o = blockIdx.x * 256u + threadIdx.x; o < n_occ; o += gridDim.x * 256u) {
        uint32_t d[16];
        load_desc(desc, o, d);
        const uint32_t t = occ_tok[o];
        const uint32_t seg = find_segment(S, t);
        const uint64_t base = S.set_base[seg], mask = S.set_mask[seg];
        const uint32_t a = SAGE_T_ID(tok_w0[t]);
        keys_from_desc<true>(d, pool, tok_w0, t, [&](uint32_t lo, uint32_t hi) {
            const uint32_t nw = set_insert(set, base, mask, make_key(a, lo, hi), flags);
            n_distinct += nw ? (lo == hi ? 1u : 2u) : 0u;
        });
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) n_distinct += __shfl_down(n_distinct, dd, 64);
    if ((threadIdx.x & 63u) == 0 && n_distinct) atomicAdd(&stats[STAT_DISTINCT], n_distinct);
}

// =================================================================================================================
// Distinct count, default back end: ONE partition level.  The key stream goes straight into buckets small enough for one
// LDS set pass (<= PART_NBK_MAX buckets per count segment, ~8k records each), so the dedupe kernel reads every record
// exactly once and writes nothing.  What makes a 2,048-way scatter write whole sectors is the size of the stage: a
// workgroup gathers the keys of a TILE of occurrences (up to PART_STAGE = 16,384 records, 128 KiB of LDS), groups
// them by bucket on chip and writes each bucket's run (8 records on average) with consecutive lanes.
//   k_tile_plan        per count segment: first occurrence, number of tiles (host-free: the grid reads the total)
//   k_key_partition3   persistent 1024-thread workgroups, one tile at a time:
//                        count keys per bucket (LDS) -> scan -> reserve each bucket's range with one coalesced atomic
//                        per bucket -> regenerate keys into their sorted place in the stage -> copy the stage out
//   k_key_giant        occurrences with more than PART_GIANT position pairs (a line of thousands of tokens that never
//                      rejoins): one workgroup each, one global atomic per key (they are rare)
//   k_key_dedupe3      one workgroup per bucket: a 16,384-slot LDS set; the next bucket's records are loaded while
//                      the current ones are inserted
// Tiles are cut at multiples of PART_FILL position pairs inside a segment; every non-giant occurrence has at most
// PART_GIANT pairs, so a tile never holds more than PART_FILL + PART_GIANT = PART_STAGE records.
// =================================================================================================================
#define PART_THREADS 1024u
#define PART_STAGE 16384u
#define PART_GIANT SAGE_PART_GIANT
#define PART_FILL (PART_STAGE - PART_GIANT)
#define PART_NBK_MAX 2048u
#define PART_LONGS 512u          // pooled occurrences a tile lists (a tile with more scans its headers instead)
#define PART_INVALID 0xFFFFFFFFu
#define SET3_SLOTS 16384u        // 128 KiB LDS set
#define SET3_FILL 13312u         // records offered to one set pass (load <= 0.81; 10,240 made the 12k-record buckets of
                                 // long-line corpora two passes, each re-reading all of them: count 18.8 -> 16.9 ms per 1M Turkish-style lines)

struct PartPlan {
    const uint32_t* pair_off;        // [n_occ + 1] exclusive scan of the per-occurrence pair counts (giants count 0)
    const uint32_t* seg_first_occ;   // [n_seg + 1]
    const uint32_t* seg_tile_off;    // [n_seg + 1] exclusive prefix of the tiles per segment
    const uint32_t* seg_bucket_base; // [n_seg + 1] index of the segment's first bucket
    const uint32_t* seg_nbk_mask;    // [n_seg] buckets of the segment - 1 (a power of two <= PART_NBK_MAX)
    const unsigned long long* bucket_off;   // [n_buckets + 1] first record of every bucket's region
    uint32_t n_seg;
};

// Per count segment: its first occurrence (occurrences are in token order) and its tile count.
__global__ __launch_bounds__(256) void k_tile_plan(const uint32_t* __restrict__ occ_tok, uint32_t n_occ,
                                                   const uint32_t* __restrict__ seg_tok_off, uint32_t n_seg,
                                                   const uint32_t* __restrict__ pair_off,
                                                   uint32_t* __restrict__ seg_first_occ, uint32_t* __restrict__ seg_tile_off) {
    for (uint32_t s = threadIdx.x; s <= n_seg; s += 256u) {
        const uint32_t t0 = seg_tok_off[s];
        uint32_t lo = 0, hi = n_occ;           // first occurrence whose token is >= t0
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (occ_tok[mid] < t0) lo = mid + 1; else hi = mid;
        }
        seg_first_occ[s] = s == n_seg ? n_occ : lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t s = 0; s < n_seg; ++s) {
            seg_tile_off[s] = acc;
            const uint32_t pairs = pair_off[seg_first_occ[s + 1]] - pair_off[seg_first_occ[s]];
            acc += (pairs + PART_FILL - 1u) / PART_FILL;
        }
        seg_tile_off[n_seg] = acc;
    }
}

// One thread per tile: its segment and occurrence range (binary searches that would otherwise stall a whole workgroup).
__global__ __launch_bounds__(256) void k_tile_table(PartPlan P, uint32_t max_tiles, uint4* __restrict__ tile_tab) {
    const uint32_t n_tiles = min(P.seg_tile_off[P.n_seg], max_tiles);
    for (uint32_t tile = blockIdx.x * 256u + threadIdx.x; tile < n_tiles; tile += gridDim.x * 256u) {
        uint32_t lo = 0, hi = P.n_seg;     // segment of the tile: largest s with seg_tile_off[s] <= tile
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (P.seg_tile_off[mid] <= tile) lo = mid; else hi = mid;
        }
        const uint32_t s = lo, j = tile - P.seg_tile_off[s];
        const uint32_t o_a = P.seg_first_occ[s], o_b = P.seg_first_occ[s + 1];
        const uint32_t p_lo = P.pair_off[o_a] + j * PART_FILL, p_hi = p_lo + PART_FILL;
        // occurrences of the tile: those whose first pair lies in [p_lo, p_hi) (a giant counts no pairs, falls to
        // whichever tile its offset lands in and is skipped there)
        uint32_t a = o_a, b = o_b;
        while (a < b) { const uint32_t mid = (a + b) >> 1; if (P.pair_off[mid] < p_lo) a = mid + 1; else b = mid; }
        const uint32_t first = a;
        b = o_b;
        while (a < b) { const uint32_t mid = (a + b) >> 1; if (P.pair_off[mid] < p_hi) a = mid + 1; else b = mid; }
        tile_tab[tile] = make_uint4(s, first, a, 0u);
    }
}

// keys of one pooled occurrence, positions j = first, first + stride, ... (a wave, or a whole workgroup)
template <class F>
__device__ __forceinline__ void keys_of_pooled(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                               const uint32_t* __restrict__ tok_w0, uint32_t o, uint32_t t, uint32_t first,
                                               uint32_t stride, F&& f) {
    const uint4 h = desc[(size_t)o * 4];
    const Win w = win_decode(h.x, h.z, h.w);
    const uint32_t pl = w.pl, k = w.k, m = w.m, pr = w.pr, L = w.L;
    const uint32_t* ids = pool + h.y;
    for (uint32_t j = first; j < L; j += stride) {
        const uint32_t id = ids[j];
        const bool c = j >= pl && j < pl + k;
        if (j >= 1u) { const uint32_t p1 = ids[j - 1]; const bool c1 = (j - 1) >= pl && (j - 1) < pl + k; if (c || c1) f(min(id, p1), max(id, p1)); }
        if (j >= 2u) { const uint32_t p2 = ids[j - 2]; const bool c2 = (j - 2) >= pl && (j - 2) < pl + k; if (c || c2) f(min(id, p2), max(id, p2)); }
    }
    const uint32_t w0 = t - pl, w1 = t + m + pr;  // base window tokens [w0, w1), centre [t, t+m)
    for (uint32_t i = w0 + first; i < w1; i += stride) {
        const uint32_t id = SAGE_T_ID(tok_w0[i]);
        const bool c = i >= t && i < t + m;
        if (i >= w0 + 1) { const uint32_t p1 = SAGE_T_ID(tok_w0[i - 1]); const bool c1 = (i - 1) >= t && (i - 1) < t + m; if (c || c1) f(min(id, p1), max(id, p1)); }
        if (i >= w0 + 2) { const uint32_t p2 = SAGE_T_ID(tok_w0[i - 2]); const bool c2 = (i - 2) >= t && (i - 2) < t + m; if (c || c2) f(min(id, p2), max(id, p2)); }
    }
}

// MODE 0: count only (the exact-size pass: adds every tile's bucket counts to `cursor`).  MODE 1: scatter.
template <int MODE>
__global__ __launch_bounds__(PART_THREADS) void k_key_partition3(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                                                 const uint32_t* __restrict__ occ_tok,
                                                                 const uint32_t* __restrict__ tok_w0, PartPlan P,
                                                                 const uint4* __restrict__ tile_tab, uint32_t max_tiles,
                                                                 uint32_t* __restrict__ cursor /* [n_buckets] records placed so far */,
                                                                 unsigned long long* __restrict__ rec, uint32_t* __restrict__ flags,
                                                                 uint2* __restrict__ tile_dbg /* debug: (records, pooled) per tile, or nullptr */) {
    extern __shared__ unsigned long long part_lds[];
    unsigned long long* stage = part_lds;                                     // [PART_STAGE]
    uint32_t* cnt = reinterpret_cast<uint32_t*>(stage + PART_STAGE);          // [PART_NBK_MAX]
    uint32_t* cstart = cnt + PART_NBK_MAX;                                    // [PART_NBK_MAX] (cnt + cstart: 64-bit deltas at copy-out)
    uint32_t* gbase = cstart + PART_NBK_MAX;                                  // [PART_NBK_MAX]
    uint32_t* longs = gbase + PART_NBK_MAX;                                   // [PART_LONGS]
    uint32_t* misc = longs + PART_LONGS;                                      // [8]: tile scalars, wave sums
    const uint32_t tid = threadIdx.x;
    const uint32_t n_tiles = min(P.seg_tile_off[P.n_seg], max_tiles);
    PH_DECL;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        PH_START();
        if (tid == 0) misc[3] = 0u;     // long occurrences listed
        for (uint32_t i = tid; i < PART_NBK_MAX; i += PART_THREADS) cnt[i] = 0u;
        __syncthreads();
        const uint4 tt = tile_tab[tile];
        const uint32_t s = tt.x, o_lo = tt.y, o_hi = tt.z;
        const uint32_t mask = P.seg_nbk_mask[s];
        const uint32_t bbase = P.seg_bucket_base[s];
        // ---- phase A: keys per bucket; pooled occurrences are set aside for the wave-per-occurrence loop.  The usual tile
        // has fewer occurrences than the workgroup has threads: each thread then keeps its occurrence in registers for
        // phase B instead of loading it again.
        const bool one_round = o_hi - o_lo <= PART_THREADS;
        InlineOcc X;
        bool have = false;
        PH_MARK(0);
        for (uint32_t o = o_lo + tid; o < o_hi; o += PART_THREADS) {
            load_inline(desc, occ_tok, tok_w0, o, X);
            PH_MARK(1);
            if (X.d[0] & SAGE_DH_GIANT) continue;                             // k_key_giant's
            if (X.d[0] & SAGE_DH_OVER) {
                // listed for the wave-per-occurrence loops.  A pooled occurrence has >= 27 position pairs, so a tile can hold
                // up to (PART_STAGE - 1) / 27 = 606 of them: more than the list.  Entries past its end are NOT dropped
                // (which ones would depend on the order the atomics land in, i.e. differ between the counting and the
                // scatter pass): a tile whose list ran full finds its pooled occurrences by scanning the headers instead.
                const uint32_t at = atomicAdd(&misc[3], 1u);
                if (at < PART_LONGS) longs[at] = o;
                continue;
            }
            have = true;
            const uint32_t a = X.a;
            keys_from_inline(X, tok_w0, [&](uint32_t lo, uint32_t hi) {
                atomicAdd(&cnt[sage_key_hash_a(make_key(a, lo, hi)) & mask], 1u);
            });
        }
        PH_MARK(2);
        __syncthreads();
        PH_MARK(3);
        // pooled occurrences (~1.5 % of them): a wave each, so that their dependent loads (header -> pool ids) overlap
        const uint32_t n_long = misc[3];
        // every pooled occurrence of the tile, a wave each: from the list, or -- when the list ran full -- from a scan of
        // the tile's headers (same occurrences either way, and the same in both passes)
        auto for_pooled = [&](auto&& body) {
            if (n_long <= PART_LONGS) {
                for (uint32_t q = tid >> 6; q < n_long; q += PART_THREADS / 64u) body(longs[q]);
            } else {
                for (uint32_t o = o_lo + (tid >> 6); o < o_hi; o += PART_THREADS / 64u) {
                    const uint32_t h = desc[(size_t)o * 4].x;
                    if ((h & SAGE_DH_OVER) && !(h & SAGE_DH_GIANT)) body(o);
                }
            }
        };
        for_pooled([&](uint32_t o) {
            const uint32_t t = occ_tok[o];
            const uint32_t a = SAGE_T_ID(tok_w0[t]);
            keys_of_pooled(desc, pool, tok_w0, o, t, tid & 63u, 64u, [&](uint32_t lo, uint32_t hi) {
                atomicAdd(&cnt[sage_key_hash_a(make_key(a, lo, hi)) & mask], 1u);
            });
        });
        __syncthreads();
        PH_MARK(4);
        // ---- exclusive scan of the bucket counts (2 per thread), bucket ranges reserved with coalesced atomics
        {
            const uint32_t c0 = cnt[2 * tid], c1 = cnt[2 * tid + 1];
            const uint32_t v = c0 + c1;
            uint32_t incl = v;
            const uint32_t lane = tid & 63u, wave = tid >> 6;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const uint32_t nn = __shfl_up(incl, dd, 64);
                if (lane >= (uint32_t)dd) incl += nn;
            }
            uint32_t* wsum = misc + 8;      // [16] wave totals
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t wbase = 0;
            for (uint32_t ww = 0; ww < wave; ++ww) wbase += wsum[ww];
            const uint32_t ex = wbase + incl - v;
            cstart[2 * tid] = ex;
            cstart[2 * tid + 1] = ex + c0;
            if (tid == PART_THREADS - 1u) {
                misc[4] = ex + v;      // records of the tile
                // a tile can hold at most PART_STAGE records by construction (k_tile_table); one that claims more means
                // the per-occurrence pair counts and the key generators disagree: the step must not trust this count
                if (ex + v > PART_STAGE) flags[FLAG_PART_MISMATCH] = 1;
                if (tile_dbg) {        // debug: the counting pass records, the scatter pass compares
                    if (MODE == 0) tile_dbg[tile] = make_uint2(ex + v, n_long);
                    else if (tile_dbg[tile].x != ex + v || tile_dbg[tile].y != n_long) {
                        flags[FLAG_PART_MISMATCH] = 1;
                        printf("k_key_partition3: tile %u (segment %u, occurrences %u..%u) counted %u records / %u pooled, scatters %u / %u\n",
                               tile, s, o_lo, o_hi, tile_dbg[tile].x, tile_dbg[tile].y, ex + v, n_long);
                    }
                }
            }
        }
        __syncthreads();
        for (uint32_t b = tid; b <= mask; b += PART_THREADS) {
            const uint32_t n = cnt[b];
            uint32_t old = 0;
            if (n) old = atomicAdd(&cursor[bbase + b], n);
            if (MODE == 1) {
                const unsigned long long cap = P.bucket_off[bbase + b + 1] - P.bucket_off[bbase + b];
                if (n && (unsigned long long)old + n > cap) { flags[FLAG_BUCKET_OVERFLOW] = 1; old = PART_INVALID; }
                gbase[b] = old;
                cnt[b] = 0u;
            }
        }
        if (MODE == 0) { __syncthreads(); continue; }
        __syncthreads();
        PH_MARK(5);
        // ---- phase B: the same keys again, each into its place in the stage
        auto place = [&](const InlineOcc& x) {
            const uint32_t a = x.a;
            keys_from_inline(x, tok_w0, [&](uint32_t lo, uint32_t hi) {
                const uint64_t key = make_key(a, lo, hi);
                const uint32_t b = sage_key_hash_a(key) & mask;
                const uint32_t at = cstart[b] + atomicAdd(&cnt[b], 1u);
                if (at < PART_STAGE) stage[at] = key;
            });
        };
        if (one_round) {
            if (have) place(X);
        } else {
            for (uint32_t o = o_lo + tid; o < o_hi; o += PART_THREADS) {
                load_inline(desc, occ_tok, tok_w0, o, X);
                if (X.d[0] & SAGE_DH_OVER) continue;                          // pooled (loop below) or giant
                place(X);
            }
        }
        PH_MARK(6);
        for_pooled([&](uint32_t o) {
            const uint32_t t = occ_tok[o];
            const uint32_t a = SAGE_T_ID(tok_w0[t]);
            keys_of_pooled(desc, pool, tok_w0, o, t, tid & 63u, 64u, [&](uint32_t lo, uint32_t hi) {
                const uint64_t key = make_key(a, lo, hi);
                const uint32_t b = sage_key_hash_a(key) & mask;
                const uint32_t at = cstart[b] + atomicAdd(&cnt[b], 1u);
                if (at < PART_STAGE) stage[at] = key;
            });
        });
        __syncthreads();
        PH_MARK(7);
        // ---- copy-out: consecutive lanes write one bucket's run.  dst = rec + delta[b] + e with delta[b] = the bucket's
        // reserved start minus its start in the stage, kept as 64-bit words over the space of cnt and gbase
        {
            unsigned long long dl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint32_t b = 2u * tid + (uint32_t)q;
                const uint32_t gb = gbase[b];
                dl[q] = (b <= mask && gb != PART_INVALID) ? P.bucket_off[bbase + b] + gb - cstart[b] : ~0ull;
            }
            __syncthreads();
            unsigned long long* delta = reinterpret_cast<unsigned long long*>(cnt);     // [PART_NBK_MAX]: cnt + cstart space
            delta[2u * tid] = dl[0];
            delta[2u * tid + 1u] = dl[1];
        }
        __syncthreads();
        const uint32_t n_rec = min(misc[4], PART_STAGE);
        const unsigned long long* delta = reinterpret_cast<const unsigned long long*>(cnt);
        for (uint32_t e = tid; e < n_rec; e += PART_THREADS) {
            const uint64_t key = stage[e];
            const unsigned long long dlt = delta[sage_key_hash_a(key) & mask];
            if (dlt != ~0ull) rec[dlt + e] = key;
        }
        __syncthreads();
        PH_MARK(8);
#ifdef PHASE_CLOCKS
        ph_d[9] += 1;
#endif
    }
    if (MODE == 1) PH_FLUSH(32, 10);
}

// Giant occurrences: one workgroup each, one global atomic per key.
template <int MODE>
__global__ __launch_bounds__(256) void k_key_giant(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                                   const uint32_t* __restrict__ occ_tok, const uint32_t* __restrict__ tok_w0,
                                                   const uint32_t* __restrict__ giant_list, const uint32_t* __restrict__ giant_count,
                                                   SegPlan S, PartPlan P, uint32_t* __restrict__ cursor,
                                                   unsigned long long* __restrict__ rec, uint32_t* __restrict__ flags) {
    const uint32_t n = *giant_count;
    for (uint32_t g = blockIdx.x; g < n; g += gridDim.x) {
        const uint32_t o = giant_list[g], t = occ_tok[o];
        const uint32_t s = find_segment(S, t);
        const uint32_t mask = P.seg_nbk_mask[s], bbase = P.seg_bucket_base[s];
        const uint32_t a = SAGE_T_ID(tok_w0[t]);
        keys_of_pooled(desc, pool, tok_w0, o, t, threadIdx.x, 256u, [&](uint32_t lo, uint32_t hi) {
            const uint64_t key = make_key(a, lo, hi);
            const uint32_t b = bbase + (sage_key_hash_a(key) & mask);
            const uint32_t at = atomicAdd(&cursor[b], 1u);
            if (MODE == 1) {
                if ((unsigned long long)at < P.bucket_off[b + 1] - P.bucket_off[b]) rec[P.bucket_off[b] + at] = key;
                else flags[FLAG_BUCKET_OVERFLOW] = 1;
            }
        });
    }
}

// Double hashing (an odd step from the key's own hash), not linear probing: a bucket's set fills to half its slots (nearly
// all records of a segment are distinct keys), and an insertion that lands in a cluster walks the cluster while its wave
// waits (12.3 -> 9.7 ms for the kernel).  Measured alternatives, all slower: the first probe of a thread's eight records in flight together
// (10.9 ms), every record of a thread probing in the same round, lanes that move on to their next record as soon as a
// probe ends, a per-wave queue that probes the stragglers densely, a set split into 32-bit claim words + stored high
// words, two half-full passes per bucket (11.9 ms).  Without the insertions the kernel takes 4.5 ms.
__device__ __forceinline__ uint32_t lds_set3_insert(unsigned long long* set, uint64_t key, uint32_t* flags) {
    const uint32_t hb = sage_key_hash_b(key);
    uint32_t s = hb & (SET3_SLOTS - 1u);
    const uint32_t step = (hb >> 14) | 1u;               // odd: visits every slot of the power-of-two table
    for (uint32_t probes = 0; probes < SET3_SLOTS; ++probes) {
        const unsigned long long old = atomicCAS(&set[s], (unsigned long long)SAGE_SET_EMPTY, (unsigned long long)key);
        if (old == SAGE_SET_EMPTY) return ((uint32_t)(key >> 20) & SAGE_ID_MASK) == ((uint32_t)key & SAGE_ID_MASK) ? 1u : 2u;
        if (old == key) return 0u;
        s = (s + step) & (SET3_SLOTS - 1u);
    }
    flags[FLAG_SET_OVERFLOW] = 1;
    return 0u;
}

#define DD3_UNR 8   // records a thread holds of one bucket (8 x 1024 = the usual bucket in one round)
__global__ __launch_bounds__(PART_THREADS) void k_key_dedupe3(const unsigned long long* __restrict__ bucket_off,
                                                              const uint32_t* __restrict__ cursor, uint32_t n_buckets,
                                                              const unsigned long long* __restrict__ rec,
                                                              unsigned long long* __restrict__ stats, uint32_t* __restrict__ flags) {
    extern __shared__ unsigned long long set3[];
    unsigned long long n_distinct = 0, n_read = 0;
    const uint32_t tid = threadIdx.x;
    // The workgroup's work is ONE stream of chunks of 8,192 records: (bucket, pass, chunk), a bucket of up to SET3_FILL
    // records being one pass of one or two chunks, a larger one several passes over all of its chunks.  While a chunk is
    // inserted the next chunk's records are on their way (and the sizes of the next two buckets), whatever it belongs to:
    // one loop body, no tail loop.  Three things keep those loads in flight: they are UNCONDITIONAL (a thread without a
    // record at that place reads record 0 of the array and ignores it) -- with a branch around each load the compiler
    // cannot count how many are outstanding and waits for all of them, `vmcnt(0)`, before it touches the current chunk;
    // the barriers are lds_barrier(), not __syncthreads(), whose fence waits for global loads too; and no rarely taken
    // loop with loads of its own shares the body (its registers made the compiler drain at the loop head).
    constexpr uint32_t CHUNK = PART_THREADS * DD3_UNR;
    auto meta = [&](uint32_t b, uint32_t& cnt, unsigned long long& off) {
        cnt = 0; off = 0;
        if (b >= n_buckets) return;
        off = bucket_off[b];
        const unsigned long long cap = bucket_off[b + 1] - off;
        cnt = (uint32_t)min((unsigned long long)cursor[b], cap);     // (an overflowed bucket is clamped; the step is redone)
    };
    uint64_t nx[DD3_UNR];
    auto fetch = [&](uint32_t cnt, unsigned long long off, uint32_t chunk) {
        const unsigned long long* in = rec + off;
#pragma unroll
        for (int u = 0; u < DD3_UNR; ++u) {
            const uint32_t i = chunk * CHUNK + tid + PART_THREADS * u;
            const unsigned long long* src = i < cnt ? in + i : rec;
            nx[u] = *src;
        }
    };
    // current item (bucket b0 of cnt0 records at off0, pass p of `passes`, chunk), the two buckets after it
    uint32_t b0 = blockIdx.x, cnt0, cnt1, cnt2, p = 0, chunk = 0;
    unsigned long long off0, off1, off2;
    meta(b0, cnt0, off0);
    meta(b0 + gridDim.x, cnt1, off1);
    meta(b0 + 2u * gridDim.x, cnt2, off2);
    auto next_bucket = [&]() {       // (uniform) shifts the bucket window by one, skipping empty buckets
        do {
            b0 += gridDim.x;
            cnt0 = cnt1; off0 = off1; cnt1 = cnt2; off1 = off2;
            meta(b0 + 2u * gridDim.x, cnt2, off2);
        } while (b0 < n_buckets && cnt0 == 0u);
        p = 0; chunk = 0;
    };
    if (b0 < n_buckets && cnt0 == 0u) next_bucket();
    if (b0 < n_buckets) fetch(cnt0, off0, 0u);
    while (b0 < n_buckets) {
        uint64_t cur[DD3_UNR];
#pragma unroll
        for (int u = 0; u < DD3_UNR; ++u) cur[u] = nx[u];
        // this item
        const uint32_t cnt = cnt0, passes = (cnt0 + SET3_FILL - 1u) / SET3_FILL, my_p = p, my_chunk = chunk;
        // the next item, and its loads
        if ((chunk + 1u) * CHUNK < cnt0) ++chunk;
        else if (p + 1u < passes) { ++p; chunk = 0; }
        else next_bucket();
        if (b0 < n_buckets) fetch(cnt0, off0, chunk);
        if (my_chunk == 0u) {            // a new (bucket, pass): every thread is done with the set's previous contents
            if (tid == 0 && my_p == 0u) n_read += cnt;
            lds_barrier();
            for (uint32_t i = tid; i < SET3_SLOTS; i += PART_THREADS) set3[i] = SAGE_SET_EMPTY;
            lds_barrier();
        }
#pragma unroll
        for (int u = 0; u < DD3_UNR; ++u)
            if (my_chunk * CHUNK + tid + PART_THREADS * u < cnt && (passes == 1u || (sage_key_hash_a(cur[u]) >> 14) % passes == my_p))
                n_distinct += lds_set3_insert(set3, cur[u], flags);
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) n_distinct += __shfl_down(n_distinct, dd, 64);
    if ((tid & 63u) == 0 && n_distinct) atomicAdd(&stats[STAT_DISTINCT], n_distinct);
    if (tid == 0 && n_read) atomicAdd(&stats[STAT_RECORDS], n_read);
}
