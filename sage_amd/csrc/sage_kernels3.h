// Distinct (ablated, target, context) keys per count segment, v2: hash-partition the key stream into buckets that
// fit an LDS set, then deduplicate each bucket on chip.  Everything is a coalesced stream over HBM; no global
// atomics per key (v1 = one random 64-bit CAS per key into an HBM-resident set).
//
//   k_key_hist     keys regenerated from descriptors -> counts per (segment, coarse bucket)           [read desc]
//   (host prefix sum of the 64 x n_seg counters)
//   k_key_scatter  keys regenerated again -> written into their coarse bucket's region                 [read desc, write rec]
//   k_key_dedupe   one block per coarse bucket: LDS histogram -> fine buckets (second buffer) -> LDS set per fine
//                  bucket, counts new keys (weight 2, or 1 when target == context)                     [read rec x3, write rec]
#pragma once
#include "sage_kernels2.h"

#define NB1 64           // coarse buckets per count segment
#define FINE_TARGET 1536 // records per fine bucket aimed at
#define SET_SLOTS 4096   // LDS set capacity (u64 slots, 32 KiB)
#define SET_FILL 2048    // at most this many records are offered to one set pass (load <= 0.5)
#define NB2_MAX 256
#define PART_TILE 256u  // occurrences per k_key_partition tile (one workgroup)
#define UNR 8          // keys a thread keeps in flight in the streaming loops of k_key_dedupe

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

__device__ __forceinline__ void load_desc(const uint4* __restrict__ desc, uint32_t o, uint32_t (&d)[16]) {
    const uint4 a = desc[(size_t)o * 4], b = desc[(size_t)o * 4 + 1], c = desc[(size_t)o * 4 + 2], e = desc[(size_t)o * 4 + 3];
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
    d[8] = c.x; d[9] = c.y; d[10] = c.z; d[11] = c.w; d[12] = e.x; d[13] = e.y; d[14] = e.z; d[15] = e.w;
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

// ---- passes 1 and 2: counts per (segment, coarse bucket), then the scatter ---------------------------------------
// MODE 0: count into `counters`.  MODE 1: scatter into rec using `counters` as cursors (they start at each bucket's
// base offset).  A tile of 256 consecutive occurrences touches at most two segments in practice; those go through
// LDS counters and one global atomic per (tile, bucket); anything else takes one global atomic per key.
template <int MODE>
__global__ __launch_bounds__(PART_TILE) void k_key_partition(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                                       const uint32_t* __restrict__ occ_tok,
                                                       const uint32_t* __restrict__ tok_w0, uint32_t n_occ, SegPlan S,
                                                       unsigned long long* __restrict__ counters,
                                                       unsigned long long* __restrict__ rec,
                                                       const unsigned long long* __restrict__ limit, uint32_t* __restrict__ flags) {
    // limit != nullptr (MODE 1 without a counting pass): bucket regions were sized from an estimate; a reservation that
    // would run past limit[bucket] raises FLAG_BUCKET_OVERFLOW and writes nothing (the step is redone with exact sizes)
    __shared__ unsigned int cnt[2][NB1];
    __shared__ unsigned long long gbase[2][NB1];
    const uint32_t n_round = (n_occ + PART_TILE - 1u) / PART_TILE * PART_TILE;
    for (uint32_t base = blockIdx.x * PART_TILE; base < n_round; base += gridDim.x * PART_TILE) {
        const uint32_t o = base + threadIdx.x;
        if (threadIdx.x < 2 * NB1) cnt[threadIdx.x / NB1][threadIdx.x % NB1] = 0;
        const uint32_t s0 = find_segment(S, occ_tok[base]);
        uint32_t d[16];
        const bool live = o < n_occ;
        uint32_t seg = 0, a = 0, t = 0;
        if (live) {
            load_desc(desc, o, d);
            t = occ_tok[o];
            seg = find_segment(S, t);
            a = SAGE_T_ID(tok_w0[t]);
        }
        const bool local = live && (seg - s0) < 2u;
        __syncthreads();
        if (live) {
            keys_from_desc<false>(d, pool, tok_w0, t, [&](uint32_t lo, uint32_t hi) {
                const uint64_t key = make_key(a, lo, hi);
                const uint32_t b1 = sage_key_hash_a(key) & (NB1 - 1);
                if (local) atomicAdd(&cnt[seg - s0][b1], 1u);
                else if (MODE == 0) atomicAdd(&counters[(size_t)seg * NB1 + b1], 1ull);
                else {
                    const unsigned long long at = atomicAdd(&counters[(size_t)seg * NB1 + b1], 1ull);
                    if (!limit || at < limit[(size_t)seg * NB1 + b1]) rec[at] = key;
                    else flags[FLAG_BUCKET_OVERFLOW] = 1;
                }
            });
        }
        __syncthreads();
        if (threadIdx.x < 2 * NB1) {
            const uint32_t ls = threadIdx.x / NB1, b = threadIdx.x % NB1;
            const uint32_t n = cnt[ls][b];
            if (n && (s0 + ls) < S.n_seg) {
                unsigned long long old = atomicAdd(&counters[(size_t)(s0 + ls) * NB1 + b], (unsigned long long)n);
                if (MODE == 1 && limit && old + n > limit[(size_t)(s0 + ls) * NB1 + b]) { flags[FLAG_BUCKET_OVERFLOW] = 1; old = ~0ull; }
                if (MODE == 1) gbase[ls][b] = old;
            }
            if (MODE == 1) cnt[ls][b] = 0;
        }
        if (MODE == 1) {
            __syncthreads();
            if (local) {
                keys_from_desc<false>(d, pool, tok_w0, t, [&](uint32_t lo, uint32_t hi) {
                    const uint64_t key = make_key(a, lo, hi);
                    const uint32_t b1 = sage_key_hash_a(key) & (NB1 - 1);
                    const uint32_t r = atomicAdd(&cnt[seg - s0][b1], 1u);
                    const unsigned long long gb = gbase[seg - s0][b1];
                    if (gb != ~0ull) rec[gb + r] = key;
                });
            }
        }
        __syncthreads();
    }
}

// ---- passes 1 and 2 for the long (pooled) occurrences: a WAVE per occurrence, lanes stride over the window --------
// A wave takes LONG_CHUNK consecutive list entries, splits them into runs of one segment (nearly always a single
// run: the list is in rough corpus order), counts the run's keys per bucket in its own LDS counters, reserves the
// bucket ranges with one global atomic per non-empty bucket, and (MODE 1) walks the run again to write the keys.
#define LONG_CHUNK 16
template <class F>
__device__ __forceinline__ void keys_of_long(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                             const uint32_t* __restrict__ tok_w0, uint32_t o, uint32_t t, uint32_t lane, F&& f) {
    const uint4 h = desc[(size_t)o * 4];
    const Win w = win_decode(h.x, h.z, h.w);
    const uint32_t pl = w.pl, k = w.k, m = w.m, pr = w.pr, L = w.L;
    const uint32_t* ids = pool + h.y;
    for (uint32_t j = lane; j < L; j += 64u) {
        const uint32_t id = ids[j];
        const bool c = j >= pl && j < pl + k;
        if (j >= 1u) { const uint32_t p1 = ids[j - 1]; const bool c1 = (j - 1) >= pl && (j - 1) < pl + k; if (c || c1) f(min(id, p1), max(id, p1)); }
        if (j >= 2u) { const uint32_t p2 = ids[j - 2]; const bool c2 = (j - 2) >= pl && (j - 2) < pl + k; if (c || c2) f(min(id, p2), max(id, p2)); }
    }
    const uint32_t w0 = t - pl, w1 = t + m + pr;  // base window tokens [w0, w1), centre [t, t+m)
    for (uint32_t i = w0 + lane; i < w1; i += 64u) {
        const uint32_t id = SAGE_T_ID(tok_w0[i]);
        const bool c = i >= t && i < t + m;
        if (i >= w0 + 1) { const uint32_t p1 = SAGE_T_ID(tok_w0[i - 1]); const bool c1 = (i - 1) >= t && (i - 1) < t + m; if (c || c1) f(min(id, p1), max(id, p1)); }
        if (i >= w0 + 2) { const uint32_t p2 = SAGE_T_ID(tok_w0[i - 2]); const bool c2 = (i - 2) >= t && (i - 2) < t + m; if (c || c2) f(min(id, p2), max(id, p2)); }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_key_partition_long(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                                            const uint32_t* __restrict__ occ_tok,
                                                            const uint32_t* __restrict__ tok_w0,
                                                            const uint32_t* __restrict__ long_list,
                                                            const uint32_t* __restrict__ long_count, SegPlan S,
                                                            unsigned long long* __restrict__ counters,
                                                            unsigned long long* __restrict__ rec,
                                                            const unsigned long long* __restrict__ limit, uint32_t* __restrict__ flags) {
    static_assert(NB1 == 64, "one lane per coarse bucket");
    __shared__ unsigned int cnt[4][NB1];
    __shared__ unsigned long long gbase[4][NB1];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n = *long_count;
    const uint32_t n_chunks = (n + LONG_CHUNK - 1) / LONG_CHUNK;
    for (uint32_t chunk = blockIdx.x * 4u + wave; chunk < n_chunks; chunk += gridDim.x * 4u) {
        const uint32_t e0 = chunk * LONG_CHUNK, e1 = min(n, e0 + LONG_CHUNK);
        // lane i < LONG_CHUNK holds entry e0 + i: occurrence, token, segment
        uint32_t my_o = 0, my_t = 0, my_seg = 0xFFFFFFFFu;
        if (e0 + lane < e1) {
            my_o = long_list[e0 + lane];
            my_t = occ_tok[my_o];
            my_seg = find_segment(S, my_t);
        }
        uint32_t e = e0;
        while (e < e1) {
            const uint32_t seg = __shfl(my_seg, (int)(e - e0), 64);
            uint32_t f = e + 1;
            while (f < e1 && __shfl(my_seg, (int)(f - e0), 64) == seg) ++f;
            cnt[wave][lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t g = e; g < f; ++g) {
                const uint32_t o = __shfl(my_o, (int)(g - e0), 64), t = __shfl(my_t, (int)(g - e0), 64);
                const uint32_t a = SAGE_T_ID(tok_w0[t]);
                keys_of_long(desc, pool, tok_w0, o, t, lane, [&](uint32_t lo, uint32_t hi) {
                    atomicAdd(&cnt[wave][sage_key_hash_a(make_key(a, lo, hi)) & (NB1 - 1)], 1u);
                });
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                const uint32_t nb = cnt[wave][lane];
                unsigned long long old = 0;
                if (nb) old = atomicAdd(&counters[(size_t)seg * NB1 + lane], (unsigned long long)nb);
                if (MODE == 1 && limit && nb && old + nb > limit[(size_t)seg * NB1 + lane]) { flags[FLAG_BUCKET_OVERFLOW] = 1; old = ~0ull; }
                if (MODE == 1) { gbase[wave][lane] = old; cnt[wave][lane] = 0; }
            }
            if (MODE == 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t g = e; g < f; ++g) {
                    const uint32_t o = __shfl(my_o, (int)(g - e0), 64), t = __shfl(my_t, (int)(g - e0), 64);
                    const uint32_t a = SAGE_T_ID(tok_w0[t]);
                    keys_of_long(desc, pool, tok_w0, o, t, lane, [&](uint32_t lo, uint32_t hi) {
                        const uint64_t key = make_key(a, lo, hi);
                        const uint32_t b1 = sage_key_hash_a(key) & (NB1 - 1);
                        const uint32_t r = atomicAdd(&cnt[wave][b1], 1u);
                        const unsigned long long gb = gbase[wave][b1];
                        if (gb != ~0ull) rec[gb + r] = key;
                    });
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            e = f;
        }
    }
}

// ---- pass 3: one block per coarse bucket ---------------------------------------------------------------------
__device__ __forceinline__ uint32_t lds_set_insert(unsigned long long* set, uint64_t key, uint32_t* flags) {
    uint32_t s = sage_key_hash_b(key) & (SET_SLOTS - 1);
    for (uint32_t probes = 0; probes < SET_SLOTS; ++probes) {
        const unsigned long long old = atomicCAS(&set[s], (unsigned long long)SAGE_SET_EMPTY, (unsigned long long)key);
        if (old == SAGE_SET_EMPTY) return ((uint32_t)(key >> 20) & SAGE_ID_MASK) == ((uint32_t)key & SAGE_ID_MASK) ? 1u : 2u;
        if (old == key) return 0u;
        s = (s + 1) & (SET_SLOTS - 1);
    }
    flags[FLAG_SET_OVERFLOW] = 1;
    return 0u;
}

__global__ __launch_bounds__(256) void k_key_dedupe(const unsigned long long* __restrict__ bucket_off /* start of every bucket */,
                                                    const unsigned long long* __restrict__ bucket_end /* cursors after the scatter */,
                                                    uint32_t n_buckets, const unsigned long long* __restrict__ rec_in,
                                                    unsigned long long* __restrict__ rec_tmp,
                                                    unsigned long long* __restrict__ stats, uint32_t* __restrict__ flags) {
    __shared__ unsigned long long set[SET_SLOTS];
    __shared__ unsigned int hist2[NB2_MAX];
    __shared__ unsigned int offs2[NB2_MAX + 1];
    __shared__ unsigned int wsum[4];
    __shared__ unsigned int chist[NB2_MAX], cstart[NB2_MAX], gpos[NB2_MAX];
    unsigned long long n_distinct = 0;
    for (uint32_t bk = blockIdx.x; bk < n_buckets; bk += gridDim.x) {
        const unsigned long long o0 = bucket_off[bk];
        const uint32_t cnt = (uint32_t)min(bucket_end[bk] - o0, bucket_off[bk + 1] - o0);   // (an overflowed bucket is clamped)
        if (cnt == 0) continue;
        const unsigned long long* in = rec_in + o0;
        unsigned long long* tmp = rec_tmp + o0;
        uint32_t nb2 = 1;
        while (nb2 < NB2_MAX && (uint64_t)nb2 * FINE_TARGET < cnt) nb2 <<= 1;
        const unsigned long long* src = in;
        __syncthreads();
        if (nb2 > 1) {
            // fine partition inside the block: histogram, exclusive scan, scatter into tmp
            for (uint32_t i = threadIdx.x; i < nb2; i += 256) hist2[i] = 0;
            __syncthreads();
            for (uint32_t i0 = threadIdx.x; i0 < cnt; i0 += 256 * UNR) {
                uint64_t kk[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) kk[u] = (i0 + 256u * u) < cnt ? in[i0 + 256u * u] : 0ull;
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    if ((i0 + 256u * u) < cnt) atomicAdd(&hist2[(sage_key_hash_a(kk[u]) >> 6) & (nb2 - 1)], 1u);
            }
            __syncthreads();
            {   // exclusive scan of up to 256 counters by the 256 threads
                const uint32_t v = threadIdx.x < nb2 ? hist2[threadIdx.x] : 0u;
                uint32_t incl = v;
                const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const uint32_t nn = __shfl_up(incl, dd, 64);
                    if (lane >= (uint32_t)dd) incl += nn;
                }
                if (lane == 63) wsum[wave] = incl;
                __syncthreads();
                uint32_t wbase = 0;
                for (uint32_t ww = 0; ww < wave; ++ww) wbase += wsum[ww];
                if (threadIdx.x < nb2) { offs2[threadIdx.x] = wbase + incl - v; hist2[threadIdx.x] = wbase + incl - v; }
                if (threadIdx.x == 0) offs2[nb2] = cnt;
            }
            __syncthreads();
            // Scatter through LDS, 2,048 records at a time: the chunk is grouped by fine bucket on chip (the dedupe set's
            // memory is free at this point) and each bucket's run leaves as consecutive 8-byte stores, so HBM sees whole
            // sectors instead of 256 interleaved streams of partial-line writes.
            unsigned long long* stage = set;                               // 2,048 x 8 B
            unsigned short* stage_fb = reinterpret_cast<unsigned short*>(set + 256 * UNR);   // 2,048 x 2 B
            for (uint32_t c0 = 0; c0 < cnt; c0 += 256 * UNR) {
                uint64_t kk[UNR];
                uint32_t fbk[UNR], rk[UNR];
                if (threadIdx.x < nb2) chist[threadIdx.x] = 0;
                __syncthreads();
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const uint32_t i = c0 + threadIdx.x + 256u * u;
                    kk[u] = i < cnt ? in[i] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const uint32_t i = c0 + threadIdx.x + 256u * u;
                    fbk[u] = (sage_key_hash_a(kk[u]) >> 6) & (nb2 - 1);
                    rk[u] = i < cnt ? atomicAdd(&chist[fbk[u]], 1u) : 0u;
                }
                __syncthreads();
                {   // exclusive scan of the chunk histogram; global base of every run
                    const uint32_t vv = threadIdx.x < nb2 ? chist[threadIdx.x] : 0u;
                    uint32_t incl = vv;
                    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
                    for (int dd = 1; dd < 64; dd <<= 1) {
                        const uint32_t nn = __shfl_up(incl, dd, 64);
                        if (lane >= (uint32_t)dd) incl += nn;
                    }
                    if (lane == 63) wsum[wave] = incl;
                    __syncthreads();
                    uint32_t wbase = 0;
                    for (uint32_t ww = 0; ww < wave; ++ww) wbase += wsum[ww];
                    if (threadIdx.x < nb2) {
                        cstart[threadIdx.x] = wbase + incl - vv;
                        gpos[threadIdx.x] = hist2[threadIdx.x];
                        hist2[threadIdx.x] += vv;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const uint32_t i = c0 + threadIdx.x + 256u * u;
                    if (i < cnt) {
                        const uint32_t at = cstart[fbk[u]] + rk[u];
                        stage[at] = kk[u];
                        stage_fb[at] = (unsigned short)fbk[u];
                    }
                }
                __syncthreads();
                const uint32_t nchunk = min(256u * UNR, cnt - c0);
                for (uint32_t e = threadIdx.x; e < nchunk; e += 256u) {
                    const uint32_t fb = stage_fb[e];
                    tmp[gpos[fb] + (e - cstart[fb])] = stage[e];
                }
                __syncthreads();
            }
            __threadfence_block();
            __syncthreads();
            src = tmp;
        } else {
            if (threadIdx.x == 0) { offs2[0] = 0; offs2[1] = cnt; }
            __syncthreads();
        }
        for (uint32_t fb = 0; fb < nb2; ++fb) {
            const uint32_t f0 = offs2[fb], fn = offs2[fb + 1] - f0;
            if (fn == 0) continue;
            const uint32_t passes = (fn + SET_FILL - 1) / SET_FILL;
            for (uint32_t p = 0; p < passes; ++p) {
                for (uint32_t i = threadIdx.x; i < SET_SLOTS; i += 256) set[i] = SAGE_SET_EMPTY;
                __syncthreads();
                for (uint32_t i0 = threadIdx.x; i0 < fn; i0 += 256 * UNR) {
                    uint64_t kk[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) kk[u] = (i0 + 256u * u) < fn ? src[f0 + i0 + 256u * u] : 0ull;
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
                        if ((i0 + 256u * u) < fn) {
                            if (passes == 1 || (sage_key_hash_a(kk[u]) >> 14) % passes == p)
                                n_distinct += lds_set_insert(set, kk[u], flags);
                        }
                }
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) n_distinct += __shfl_down(n_distinct, dd, 64);
    if ((threadIdx.x & 63u) == 0 && n_distinct) atomicAdd(&stats[STAT_DISTINCT], n_distinct);
}
