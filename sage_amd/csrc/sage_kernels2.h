// Second-generation ablation kernels.  The occurrence walk runs ONCE per step (k_describe) and leaves a 64-byte
// descriptor per ablated occurrence; the distinct-key count and the f64 score then stream descriptors with no
// dependent pointer chase.  Scoring is split so that the gather never waits on transcendental code:
//   k_pair_cache  base-window pair terms, one row gather per base token
//   k_dots        ablated-window pair dot products: a barrier-free, continuously prefetched row-gather stream
//   k_logsig      log-sigmoid of every dot + base-window range sums -> loss[a]
#pragma once
#include "sage_kernels.h"

#define DESC_LDS_STRIDE 17  // 16 words + 1 pad: conflict-free per-thread row writes
enum { FLAG_POOL_FULL = 3 };

// ----------------------------------------------------------------------------------------------------------
// Per-block accumulation cache in LDS for the vocabulary-sized result vectors.  Token frequencies are Zipfian: a few
// ids receive millions of adds per step, and same-address global atomics serialise at the memory side.  Each block
// claims direct-mapped slots for the ids it meets (first come, first served: hot ids claim early), adds into LDS,
// and flushes once at the end; ids that lose a slot fall through to a global atomic.
// ----------------------------------------------------------------------------------------------------------
#define ACC_SLOTS 2048
#define ACC_EMPTY 0xFFFFFFFFu
__device__ __forceinline__ uint32_t acc_slot(uint32_t id) { return (id * 0x9E3779B1u) >> (32 - 11); }

__device__ __forceinline__ void acc_add_f64(uint32_t* cid, double* cval, double* __restrict__ global, uint32_t id, double v) {
    const uint32_t s = acc_slot(id);
    uint32_t cur = cid[s];
    if (cur == ACC_EMPTY) cur = atomicCAS(&cid[s], ACC_EMPTY, id), cur = (cur == ACC_EMPTY) ? id : cur;
    if (cur == id) unsafeAtomicAdd(&cval[s], v);
    else unsafeAtomicAdd(&global[id], v);
}
__device__ __forceinline__ void acc_add_u32(uint32_t* cid, uint32_t* cval, unsigned long long* __restrict__ global, uint32_t id) {
    const uint32_t s = acc_slot(id);
    uint32_t cur = cid[s];
    if (cur == ACC_EMPTY) cur = atomicCAS(&cid[s], ACC_EMPTY, id), cur = (cur == ACC_EMPTY) ? id : cur;
    if (cur == id) atomicAdd(&cval[s], 1u);
    else atomicAdd(&global[id], 1ull);
}

// ----------------------------------------------------------------------------------------------------------
// Descriptor of one occurrence (16 u32).  Inline form: word 0 = VALID | k | m<<12 | pl<<24 | pr<<26 | NEVER, words
// 1..15 = the ablated window [left pads | x1..xk | right pads].  Pooled form (window longer than 15 ids, or a base
// span longer than 4095 tokens): word 0 = VALID | OVER | pl | pr | NEVER, word 1 = offset of the window in the id
// pool, word 2 = k, word 3 = m.
// ----------------------------------------------------------------------------------------------------------
struct Win {
    uint32_t k, m, pl, pr, L;
    bool pooled;
};
__device__ __forceinline__ Win win_decode(uint32_t h, uint32_t w2, uint32_t w3) {
    Win w;
    w.pooled = (h & SAGE_DH_OVER) != 0u;
    w.k = w.pooled ? w2 : SAGE_DH_K(h);
    w.m = w.pooled ? w3 : SAGE_DH_M(h);
    w.pl = SAGE_DH_PL(h);
    w.pr = SAGE_DH_PR(h);
    w.L = w.pl + w.k + w.pr;
    return w;
}
// number of position pairs of the ablated window (model.py:129-136): element j pairs with j-1 and j-2 when either
// end lies in the centre [pl, pl+k)
__device__ __forceinline__ uint32_t win_ablated_pairs(uint32_t pl, uint32_t k, uint32_t pr) {
    const uint32_t L = pl + k + pr;
    // all distance-1 and distance-2 pairs, minus those that join two pads on the same side
    uint32_t n = (L - 1u) + (L >= 2u ? L - 2u : 0u);
    n -= (pl == 2u ? 1u : 0u) + (pr == 2u ? 1u : 0u);   // (l2,l1) and (r1,r2) at distance 1; no pad-pad pair at distance 2
    return n;                                            // (a left and a right pad are never within 2 of each other: k >= 2)
}

#ifndef SAGE_NI
#define SAGE_NI 2   // window ids in flight in front of the row ring of k_dots
#endif

// ----------------------------------------------------------------------------------------------------------
// k_describe: one thread per ablated occurrence walks it (model.py:168-232) once.
//   -> descriptor, ablated_sizes histogram, occurrences per id, position pairs per count segment, number of
//      ablated-window pairs per occurrence (offsets of its dots), step statistics.
// ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_describe(WalkArrays A, const uint32_t* __restrict__ occ_tok, uint32_t n_occ,
                                                  SegPlan S, uint4* __restrict__ desc, uint32_t* __restrict__ npairs_ab,
                                                  uint32_t* __restrict__ pool, uint32_t* __restrict__ pool_cursor,
                                                  uint32_t pool_cap, unsigned long long* __restrict__ seg_pairs,
                                                  unsigned long long* __restrict__ occ,
                                                  unsigned long long* __restrict__ hist, uint32_t hist_len,
                                                  unsigned long long* __restrict__ stats, uint32_t* __restrict__ flags,
                                                  const uint32_t* __restrict__ inexact, uint32_t* __restrict__ inex_list,
                                                  uint32_t* __restrict__ inex_count) {
    __shared__ uint32_t rows[256 * DESC_LDS_STRIDE];
    __shared__ unsigned int lhist[32];
    __shared__ uint32_t occ_id[ACC_SLOTS];
    __shared__ uint32_t occ_n[ACC_SLOTS];
    if (threadIdx.x < 32) lhist[threadIdx.x] = 0;
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256) { occ_id[i] = ACC_EMPTY; occ_n[i] = 0; }
    __syncthreads();
    unsigned long long n_rows = 0, n_pairs = 0;
    const uint32_t n_round = (n_occ + 255u) & ~255u;
    uint32_t* my = rows + threadIdx.x * DESC_LDS_STRIDE;
    for (uint32_t base = blockIdx.x * 256u; base < n_round; base += gridDim.x * 256u) {
        const uint32_t o = base + threadIdx.x;
        const bool live = o < n_occ;
        const uint32_t t = live ? occ_tok[o] : 0u;
        const uint32_t seg = live ? find_segment(S, t) : 0xFFFFFFFFu;
        unsigned long long my_pairs = 0;
        uint32_t inex = 0;   // a row of the ablated window is flagged in the `inexact` bitmap (float32 fast path only)
        if (live) {
            Walker w;
            w.init(A, t);
            uint32_t id, L = 0, m = 0, ab_pairs = 0;
            bool c, c1 = false, c2 = false, base_window = false;
            int have = 0, fresh;
            while (w.next(A, id, c, fresh)) {
                if (fresh) { have = 0; base_window = fresh < 0; }
                uint32_t np = 0;
                if (have >= 1 && (c || c1)) ++np;
                if (have >= 2 && (c || c2)) ++np;
                if (!base_window) {
                    ++n_rows;
                    if (L < SAGE_DESC_MAXWIN) my[1 + L] = id;
                    if (inexact) inex |= (inexact[id >> 5] >> (id & 31u)) & 1u;
                    ++L;
                    ab_pairs += np;
                } else if (c) {
                    ++n_rows; ++m;
                }
                my_pairs += np;
                c2 = c1; c1 = c; ++have;
            }
            const bool pooled = L > SAGE_DESC_MAXWIN || m > 0xFFFu;
            uint32_t hdr = SAGE_DH_VALID | (w.never ? SAGE_DH_NEVER : 0u) | ((uint32_t)w.pl << 24) | ((uint32_t)w.pr << 26) |
                           (inex ? SAGE_DH_INEXACT : 0u);
            if (!pooled) {
                hdr |= w.k | (m << 12);
            } else {
                hdr |= SAGE_DH_OVER;
                const uint32_t at = atomicAdd(pool_cursor, L);
                my[1] = at; my[2] = w.k; my[3] = m;
                if (at + L <= pool_cap && at + L >= at) {
                    Walker w2;       // second walk of a long window: ids straight into the pool
                    w2.init(A, t);
                    uint32_t j = 0;
                    while (w2.next(A, id, c, fresh)) {
                        if (fresh < 0) break;
                        pool[at + j++] = id;
                    }
                } else {
                    flags[FLAG_POOL_FULL] = 1;
                }
            }
            my[0] = hdr;
            npairs_ab[o] = ab_pairs;
            acc_add_u32(occ_id, occ_n, occ, w.a);
            const uint32_t kk = w.k < hist_len ? w.k : hist_len - 1u;
            if (kk < 32u) atomicAdd(&lhist[kk], 1u); else atomicAdd(&hist[kk], 1ull);
        }
        n_pairs += my_pairs;
        if (inexact) {   // flagged occurrences go on a list for the float64 fix-up pass (one atomic per wave)
            const unsigned long long bal = __ballot(inex != 0u);
            if (bal) {
                const uint32_t lane = threadIdx.x & 63u;
                uint32_t at = 0;
                if (lane == (uint32_t)__ffsll((long long)bal) - 1u) {
                    at = atomicAdd(inex_count, (uint32_t)__popcll(bal));
                    atomicAdd(&stats[STAT_INEXACT], (unsigned long long)__popcll(bal));
                }
                at = __shfl(at, __ffsll((long long)bal) - 1, 64);
                if (inex) inex_list[at + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = o;
            }
        }
        {   // pairs per segment: one atomic per wave when the wave sits inside one segment (the usual case)
            const uint32_t seg0 = __shfl(seg, 0, 64);
            if (__all(seg == seg0 || seg == 0xFFFFFFFFu) && seg0 != 0xFFFFFFFFu) {
                unsigned long long v = my_pairs;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
                if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&seg_pairs[seg0], v);
            } else if (my_pairs) {
                atomicAdd(&seg_pairs[seg], my_pairs);
            }
        }
        __syncthreads();
        // coalesced write-out of the block's 256 descriptors (4 x 16 B each)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t q = threadIdx.x + 256u * r;      // uint4 index inside the block's tile
            const uint32_t row = q >> 2, col = (q & 3u) * 4u;
            if (base + row < n_occ) {
                const uint32_t* src = rows + row * DESC_LDS_STRIDE + col;
                desc[(size_t)(base + row) * 4 + (q & 3u)] = make_uint4(src[0], src[1], src[2], src[3]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 32 && threadIdx.x < hist_len && lhist[threadIdx.x])
        atomicAdd(&hist[threadIdx.x], (unsigned long long)lhist[threadIdx.x]);
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256)
        if (occ_n[i]) atomicAdd(&occ[occ_id[i]], (unsigned long long)occ_n[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        n_rows += __shfl_down(n_rows, d, 64);
        n_pairs += __shfl_down(n_pairs, d, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (n_rows) atomicAdd(&stats[STAT_ROWS], n_rows);
        if (n_pairs) atomicAdd(&stats[STAT_PAIRS], n_pairs);
    }
}

// Row slice of one lane: EPL doubles as EPL/2 16-byte loads; the G lanes of a group cover one 128-byte line per
// load instruction (element index 2*gl + 2*G*i + {0,1}).
template <int G, int EPL>
__device__ __forceinline__ void load_row_slice(const double* __restrict__ emb, uint32_t id, uint32_t dp, uint32_t gl,
                                               double (&r)[EPL]) {
    const double2* row = reinterpret_cast<const double2*>(emb + (size_t)id * dp) + gl;
#pragma unroll
    for (int i = 0; i < EPL / 2; ++i) {
        const double2 v = row[(size_t)G * i];
        r[2 * i] = v.x; r[2 * i + 1] = v.y;
    }
}
// float32 storage (lossless for word2vec vectors, which are float32 upcast to float64: embeddings.py:81): the slice
// stays float32 in registers while in flight -- half the registers and half the bytes of the float64 form.
template <int G, int EPL>
__device__ __forceinline__ void load_row_slice(const float* __restrict__ emb, uint32_t id, uint32_t dp, uint32_t gl,
                                               float (&r)[EPL]) {
    static_assert(EPL % 4 == 0, "float slices are whole 16-byte loads");
    const float4* row = reinterpret_cast<const float4*>(emb + (size_t)id * dp) + gl;
#pragma unroll
    for (int i = 0; i < EPL / 4; ++i) {
        const float4 v = row[(size_t)G * i];
        r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
    }
}
// Sum over the G lanes of a group, result in every lane.  The first levels are DPP moves inside a 16-lane row
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror): VALU only, no trip through the LDS crossbar.
__device__ __forceinline__ double dpp_add(double v, const int ctrl_sel) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int plo, phi;
    switch (ctrl_sel) {
    case 0: plo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false); phi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false); break;
    case 1: plo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false); phi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false); break;
    case 2: plo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, false); phi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, false); break;
    default: plo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, false); phi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, false); break;
    }
    return v + __hiloint2double(phi, plo);
}
template <int G>
__device__ __forceinline__ double group_sum(double d) {
    d = dpp_add(d, 0);
    d = dpp_add(d, 1);
    if (G >= 8) d = dpp_add(d, 2);
    if (G >= 16) d = dpp_add(d, 3);
#pragma unroll
    for (int m = 16; m < G; m <<= 1) d += __shfl_xor(d, m, G);
    return d;
}
// float64 accumulation in both forms; the product of two float32 values is exact in float64
template <int G, int EPL, typename T>
__device__ __forceinline__ double group_dot(const T (&a)[EPL], const T (&b)[EPL]) {
    double d = 0.0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) d = fma((double)a[e], (double)b[e], d);
    return group_sum<G>(d);
}
// OR of a predicate over the G lanes of a group
template <int G>
__device__ __forceinline__ bool group_any(bool p) {
    const unsigned long long bal = __ballot(p);
    if (G >= 64) return bal != 0ull;
    const uint32_t base = (threadIdx.x & 63u) & ~(uint32_t)(G - 1);
    return ((bal >> base) & ((1ull << (G & 63)) - 1ull)) != 0ull;
}

// ----------------------------------------------------------------------------------------------------------
// k_pair_cache: G[i] = ( logsig(E[id_i].E[id_{i+1}]), logsig(E[id_i].E[id_{i+2}]) ) for every base token i whose
// partner lies in the same line.  The base window of ANY occurrence is a short range of G (model.py:231-232 +
// do_triples), so base rows are gathered once per token, not once per occurrence.
// A group of G lanes owns 4 consecutive tokens: 6 rows, 8 dots, one log-sigmoid per lane.
// ----------------------------------------------------------------------------------------------------------
template <int G, int EPL, typename T, bool FIX>
__global__ __launch_bounds__(256) void k_pair_cache(const uint32_t* __restrict__ tok_w0, uint32_t n_tok,
                                                    const T* __restrict__ emb, uint32_t emb_rows, uint32_t dp,
                                                    double2* __restrict__ Gc, uint32_t* __restrict__ flags,
                                                    const uint32_t* __restrict__ inexact) {
    // FIX: the float64 fix-up pass of the float32 fast path -- only quads that touch a flagged row are recomputed
    const uint32_t gl = threadIdx.x % G;
    const uint32_t groups_per_block = 256 / G;
    const uint32_t n_quads = (n_tok + 3u) / 4u;
    for (uint32_t q0 = blockIdx.x * groups_per_block; q0 < n_quads; q0 += gridDim.x * groups_per_block) {
        const uint32_t quad = q0 + threadIdx.x / G;
        const uint32_t i0 = quad * 4u;
        T r[6][EPL];
        uint32_t after[4], ids[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const uint32_t i = i0 + j;
            const uint32_t w = i < n_tok ? tok_w0[i] : 0u;
            if (j < 4) after[j] = i < n_tok ? SAGE_T_AFTER3(w) : 0u;
            uint32_t id = SAGE_T_ID(w);
            if (id >= emb_rows) { if (i < n_tok) flags[FLAG_ROW_OOB] = 1; id = 0; }
            ids[j] = id;
        }
        if (FIX) {
            uint32_t mine_id = ids[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) mine_id = ((int)gl == j) ? ids[j] : mine_id;
            const bool flagged = gl < 6u && ((inexact[mine_id >> 5] >> (mine_id & 31u)) & 1u);
            if (!group_any<G>(flagged)) continue;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) load_row_slice<G, EPL>(emb, ids[j], dp, gl, r[j]);
        double mine = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int s = 1; s <= 2; ++s) {
                const double d = group_dot<G, EPL, T>(r[j], r[j + s]);
                if ((int)gl == 2 * j + (s - 1)) mine = d;
            }
        }
        // lane gl holds dot (token j = gl/2, distance s = gl%2 + 1)
        if (gl < 8) {
            const uint32_t j = gl >> 1, s = (gl & 1u) + 1u;
            const bool valid = (i0 + j) < n_tok && after[j] >= s;
            const double v = valid ? log_sigmoid(mine) : 0.0;
            if ((i0 + j) < n_tok) reinterpret_cast<double*>(Gc)[(size_t)(i0 + j) * 2 + (s - 1)] = v;
        }
    }
}

// ----------------------------------------------------------------------------------------------------------
// k_dots: for every occurrence, the dot products E[w_j].E[w_j-1], E[w_j].E[w_j-2] of its ablated window that
// do_triples counts (model.py:129-136), written back to back at dot_off[o].  Gather-bound: G lanes own one
// occurrence and walk a continuous stream of rows across occurrence boundaries -- row j+2 and j+1 are always in
// flight while row j is reduced, the next occurrence's header is loaded one occurrence ahead, and there is no
// barrier, no LDS and no transcendental code in this kernel, so a CU keeps >100 KB of row gathers outstanding.
// ----------------------------------------------------------------------------------------------------------
template <int G, int EPL, typename T, int NS>
__global__ __launch_bounds__(256) void k_dots(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                              uint32_t n_occ, const uint32_t* __restrict__ dot_off,
                                              const T* __restrict__ emb, uint32_t emb_rows, uint32_t dp,
                                              double* __restrict__ dots, uint32_t* __restrict__ flags,
                                              const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count) {
    // list != nullptr: only the *list_count occurrences on the list are processed (float64 fix-up pass of the float32
    // fast path)
    const uint32_t gl = threadIdx.x % G;
    const uint32_t n_groups = gridDim.x * (256u / G);
    const uint32_t gid = blockIdx.x * (256u / G) + threadIdx.x / G;
    const uint32_t n = list ? *list_count : n_occ;
    // Each element travelling through the pipeline carries: its row slice and a meta word
    //   bit0: dot with the previous element, bit1: dot with the one before, bit2: first element of an occurrence,
    //   bit3: element exists.  `out` is only read at a first element.
    // Ring of NS row slices; roles rotate, data never moves: the current row, the two before it, and NF = NS - 3 rows
    // whose loads are in flight (fully unrolled: every index below is a compile-time constant).  In front of the rows
    // runs a queue of NI window ids whose loads are in flight, so the dependent chain id -> row address -> row never
    // stalls a step: an element enters as an id load, NI steps later its row load is issued, NF steps later it is
    // reduced against its two predecessors.
    constexpr int NF = NS - 3, NI = SAGE_NI, NQ = NF + NI;
    static_assert(NF >= 1, "ring too short");
    T B[NS][EPL];
    uint32_t qid[NI];           // ids in flight, oldest first
    uint32_t qm[NQ], qo[NQ];    // meta / output offset of every element in flight (rows first, then ids), oldest first
#pragma unroll
    for (int i = 0; i < NQ; ++i) { qm[i] = 0; qo[i] = 0; }
#pragma unroll
    for (int i = 0; i < NI; ++i) qid[i] = 0;
    // id side: fi = position (in the list, or plain index) of the next occurrence to open, foi = its index,
    // nh = its header words (prefetched one occurrence ahead)
    uint32_t fi = gid, foi = 0, fj = 0, fL = 0, fpl = 0, fk = 0;
    const uint32_t* fids = nullptr;
    uint4 nh = make_uint4(0, 0, 0, 0);
    if (fi < n) { foi = list ? list[fi] : fi; nh = desc[(size_t)foi * 4]; }
    auto fetch_id = [&](uint32_t& id, uint32_t& meta, uint32_t& out) {
        meta = 0; id = 0;
        if (fj == fL) {   // open the next occurrence from the prefetched header, prefetch the one after
            if (fi >= n) return;
            const Win w = win_decode(nh.x, nh.z, nh.w);
            fL = w.L; fpl = w.pl; fk = w.k; fj = 0;
            fids = w.pooled ? pool + nh.y : reinterpret_cast<const uint32_t*>(desc + (size_t)foi * 4) + 1;
            out = dot_off[foi];
            fi += n_groups;
            if (fi < n) { foi = list ? list[fi] : fi; nh = desc[(size_t)foi * 4]; }
        }
        id = fids[fj];
        const bool c = fj >= fpl && fj < fpl + fk;
        const bool c1 = (fj - 1u) >= fpl && (fj - 1u) < fpl + fk;   // fj - 1 wraps for fj == 0: false
        const bool c2 = (fj - 2u) >= fpl && (fj - 2u) < fpl + fk;
        meta = 8u | (fj == 0u ? 4u : 0u) | ((fj >= 1u && (c || c1)) ? 1u : 0u) | ((fj >= 2u && (c || c2)) ? 2u : 0u);
        ++fj;
    };
    uint32_t wr = 0;   // write cursor into dots[]
#pragma unroll
    for (int i = 0; i < NI; ++i) fetch_id(qid[i], qm[NF + i], qo[NF + i]);
    for (;;) {
        uint32_t pending = 0;
#pragma unroll
        for (int i = 0; i < NQ; ++i) pending |= qm[i];
        if (!__any(pending & 8u)) break;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            // slot s becomes current (its load was issued NF steps ago); the slot freed by the last step takes the row
            // of the oldest id; a new id load joins the queue
            const uint32_t m0 = qm[0], out0 = qo[0];
            uint32_t rid = qid[0];
            const bool rlive = (qm[NF] & 8u) != 0u;
#pragma unroll
            for (int i = 0; i + 1 < NQ; ++i) { qm[i] = qm[i + 1]; qo[i] = qo[i + 1]; }
#pragma unroll
            for (int i = 0; i + 1 < NI; ++i) qid[i] = qid[i + 1];
            fetch_id(qid[NI - 1], qm[NQ - 1], qo[NQ - 1]);
            if (rid >= emb_rows) { if (rlive) flags[FLAG_ROW_OOB] = 1; rid = 0; }
            load_row_slice<G, EPL>(emb, rid, dp, gl, B[(s + NF) % NS]);
            if (m0 & 8u) {
                if (m0 & 4u) wr = out0;
                if (m0 & 1u) {
                    const double d_ = group_dot<G, EPL, T>(B[s], B[(s + NS - 1) % NS]);
                    if (gl == 0) dots[wr] = d_;
                    ++wr;
                }
                if (m0 & 2u) {
                    const double d_ = group_dot<G, EPL, T>(B[s], B[(s + NS - 2) % NS]);
                    if (gl == 0) dots[wr] = d_;
                    ++wr;
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------
// k_logsig: loss[a] += -2 * ( sum_{ablated pairs} logsig(dot)  -  sum_{base pairs} G )   (utils.py:125-138,255; every
// position pair stands for both (t,c) orders).  Window lengths are heavy-tailed, so the dots are NOT walked per
// occurrence: a workgroup takes 256 consecutive occurrences, cuts their contiguous run of dots into 256 equal slices
// (one per thread, whatever occurrence they belong to), and every thread adds its partial sums into per-occurrence
// LDS accumulators, following the occurrence boundaries as it goes.  A second phase (one thread per occurrence) adds
// the base window as a range sum over the pair cache and accumulates per ablated id.
// ----------------------------------------------------------------------------------------------------------
#define LOGSIG_TILE 256
__global__ __launch_bounds__(256) void k_logsig(const uint4* __restrict__ desc, const uint32_t* __restrict__ occ_tok,
                                                const uint32_t* __restrict__ tok_w0, uint32_t n_occ,
                                                const uint32_t* __restrict__ dot_off, const double* __restrict__ dots,
                                                const double2* __restrict__ Gc, double* __restrict__ loss) {
    __shared__ uint32_t acc_id[ACC_SLOTS];
    __shared__ double acc_val[ACC_SLOTS];
    __shared__ uint32_t off[LOGSIG_TILE + 1];
    __shared__ double osum[LOGSIG_TILE];
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256) { acc_id[i] = ACC_EMPTY; acc_val[i] = 0.0; }
    const double* Gd = reinterpret_cast<const double*>(Gc);
    for (uint32_t o0 = blockIdx.x * LOGSIG_TILE; o0 < n_occ; o0 += gridDim.x * LOGSIG_TILE) {
        const uint32_t nt = min((uint32_t)LOGSIG_TILE, n_occ - o0);   // occurrences of this tile
        __syncthreads();
        if (threadIdx.x < nt) off[threadIdx.x] = dot_off[o0 + threadIdx.x];
        if (threadIdx.x == 0) off[nt] = dot_off[o0 + nt];   // nt can be 256: there is no thread 256
        osum[threadIdx.x] = 0.0;
        // phase-2 operands are independent of phase 1: issue their loads first
        const uint32_t o = o0 + threadIdx.x;
        uint4 h = make_uint4(0, 0, 0, 0);
        uint32_t t = 0;
        if (threadIdx.x < nt) { h = desc[(size_t)o * 4]; t = occ_tok[o]; }
        __syncthreads();
        const uint32_t d0 = off[0], d1 = off[nt];
        const uint32_t per = (d1 - d0 + 255u) / 256u;
        uint32_t i = d0 + threadIdx.x * per;
        const uint32_t iend = min(d1, i + per);
        if (i < iend) {
            // owner of dot i: largest j with off[j] <= i
            uint32_t lo = 0, hi = nt;
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (off[mid] <= i) lo = mid; else hi = mid;
            }
            uint32_t own = lo, nxt = off[own + 1];
            double acc = 0.0;
            for (; i < iend; ++i) {
                if (i >= nxt) {
                    unsafeAtomicAdd(&osum[own], acc);
                    acc = 0.0;
                    do { ++own; nxt = off[own + 1]; } while (i >= nxt);
                }
                acc += log_sigmoid(dots[i]);
            }
            unsafeAtomicAdd(&osum[own], acc);
        }
        __syncthreads();
        if (threadIdx.x < nt) {
            const Win w = win_decode(h.x, h.z, h.w);
            const uint32_t a = SAGE_T_ID(tok_w0[t]);
            // base window: pair (i,i+1) counts iff i or i+1 is in the centre [t, t+m): rel = i-t in [-1, m-1] clipped to
            // the window [t-pl, t+m+pr-1]; pair (i,i+2) iff rel in [0,m-1] or rel+2 in [0,m-1] (not contiguous for
            // m == 1: rel = -1 joins two pads).
            const int im = (int)w.m, ipl = (int)w.pl, ipr = (int)w.pr;
            const int lo1 = ipl >= 1 ? -1 : 0, hi1 = im - 2 + (ipr >= 1 ? 1 : 0);
            const int lo2 = -ipl, hi2 = im + ipr - 3;
            double sb = 0.0;
            for (int rel = lo1; rel <= hi1; ++rel) sb += Gd[((size_t)t + rel) * 2];
            for (int rel = lo2; rel <= hi2; ++rel)
                if ((rel >= 0 && rel <= im - 1) || (rel + 2 >= 0 && rel + 2 <= im - 1)) sb += Gd[((size_t)t + rel) * 2 + 1];
            acc_add_f64(acc_id, acc_val, loss, a, -2.0 * (osum[threadIdx.x] - sb));
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256)
        if (acc_id[i] != ACC_EMPTY) unsafeAtomicAdd(&loss[acc_id[i]], acc_val[i]);
}
