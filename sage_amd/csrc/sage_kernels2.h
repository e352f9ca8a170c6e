// Second-generation ablation kernels: the occurrence walk runs ONCE per step (k_describe) and leaves a 64-byte
// descriptor per base token; the distinct-key count and the f64 score then stream descriptors with no dependent
// pointer chase, and the base-window terms come from a per-token pair cache (k_pair_cache).
#pragma once
#include "sage_kernels.h"

#define DESC_LDS_STRIDE 17  // 16 words + 1 pad: conflict-free per-thread row writes

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
// k_describe: one thread per base token walks its occurrence (model.py:168-232) once.
//   -> descriptor (header + ablated window ids), ablated_sizes histogram, occurrences per id, position pairs per
//      count segment (sizes the distinct-key structures), step statistics, list of over-long windows.
// ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_describe(WalkArrays A, uint32_t n_tok, SegPlan S, uint4* __restrict__ desc,
                                                  unsigned long long* __restrict__ seg_pairs,
                                                  unsigned long long* __restrict__ occ,
                                                  unsigned long long* __restrict__ hist, uint32_t hist_len,
                                                  unsigned long long* __restrict__ stats,
                                                  uint32_t* __restrict__ over_list, uint32_t* __restrict__ over_count) {
    __shared__ uint32_t rows[256 * DESC_LDS_STRIDE];
    __shared__ unsigned int lhist[32];
    __shared__ uint32_t occ_id[ACC_SLOTS];
    __shared__ uint32_t occ_n[ACC_SLOTS];
    if (threadIdx.x < 32) lhist[threadIdx.x] = 0;
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256) { occ_id[i] = ACC_EMPTY; occ_n[i] = 0; }
    __syncthreads();
    unsigned long long n_occ = 0, n_rows = 0, n_pairs = 0;
    const uint32_t n_round = (n_tok + 255u) & ~255u;
    uint32_t* my = rows + threadIdx.x * DESC_LDS_STRIDE;
    for (uint32_t base = blockIdx.x * 256u; base < n_round; base += gridDim.x * 256u) {
        const uint32_t t = base + threadIdx.x;
        Walker w;
        const bool live = t < n_tok && w.init(A, t);
        const uint32_t seg = t < n_tok ? find_segment(S, t) : 0xFFFFFFFFu;
        unsigned long long my_pairs = 0;
        uint32_t hdr = 0;
        if (live) {
            uint32_t id, L = 0, m = 0;
            bool c, c1 = false, c2 = false, base_window = false;
            int have = 0, fresh;
            while (w.next(A, id, c, fresh)) {
                if (fresh) { have = 0; base_window = fresh < 0; }
                if (!base_window) {
                    ++n_rows;
                    if (L < SAGE_DESC_MAXWIN) my[1 + L] = id;
                    ++L;
                } else if (c) {
                    ++n_rows; ++m;
                }
                if (have >= 1 && (c || c1)) ++my_pairs;
                if (have >= 2 && (c || c2)) ++my_pairs;
                c2 = c1; c1 = c; ++have;
            }
            const bool over = L > SAGE_DESC_MAXWIN || w.k > 0xFFFu || m > 0xFFFu;
            hdr = SAGE_DH_VALID | (over ? SAGE_DH_OVER : 0u) | (w.never ? SAGE_DH_NEVER : 0u) |
                  (over ? 0u : (w.k | (m << 12))) | ((uint32_t)w.pl << 24) | ((uint32_t)w.pr << 26);
            if (over) over_list[atomicAdd(over_count, 1u)] = t;
            ++n_occ;
            acc_add_u32(occ_id, occ_n, occ, w.a);
            const uint32_t kk = w.k < hist_len ? w.k : hist_len - 1u;
            if (kk < 32u) atomicAdd(&lhist[kk], 1u); else atomicAdd(&hist[kk], 1ull);
        }
        my[0] = hdr;
        n_pairs += my_pairs;
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
        // coalesced write-out of the block's 256 descriptors (4 x 16 B per token)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t q = threadIdx.x + 256u * r;      // uint4 index inside the block's tile
            const uint32_t row = q >> 2, col = (q & 3u) * 4u;
            if (base + row < n_tok) {
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
        n_occ += __shfl_down(n_occ, d, 64);
        n_rows += __shfl_down(n_rows, d, 64);
        n_pairs += __shfl_down(n_pairs, d, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (n_occ) atomicAdd(&stats[STAT_OCC], n_occ);
        if (n_rows) atomicAdd(&stats[STAT_ROWS], n_rows);
        if (n_pairs) atomicAdd(&stats[STAT_PAIRS], n_pairs);
    }
}

// ----------------------------------------------------------------------------------------------------------
// k_pair_cache: G[i] = ( logsig(E[id_i].E[id_{i+1}]), logsig(E[id_i].E[id_{i+2}]) ) for every base token i whose
// partner lies in the same line.  The base window of ANY occurrence is a short contiguous range of G
// (model.py:231-232 + do_triples), so base rows are gathered once per token, not once per occurrence.
// A group of G lanes owns 4 consecutive tokens: 6 rows, 8 dots, one log-sigmoid per lane.
// ----------------------------------------------------------------------------------------------------------
template <int G, int EPL>
__global__ __launch_bounds__(256) void k_pair_cache(const uint32_t* __restrict__ tok_w0, uint32_t n_tok,
                                                    const double* __restrict__ emb, uint32_t emb_rows, uint32_t dp,
                                                    double2* __restrict__ Gc, uint32_t* __restrict__ flags) {
    const uint32_t gl = threadIdx.x % G;
    const uint32_t groups_per_block = 256 / G;
    const uint32_t n_quads = (n_tok + 3u) / 4u;
    for (uint32_t q0 = blockIdx.x * groups_per_block; q0 < n_quads; q0 += gridDim.x * groups_per_block) {
        const uint32_t quad = q0 + threadIdx.x / G;
        const uint32_t i0 = quad * 4u;
        double r[6][EPL];
        uint32_t after[4];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const uint32_t i = i0 + j;
            uint32_t w = i < n_tok ? tok_w0[i] : 0u;
            if (j < 4) after[j] = i < n_tok ? SAGE_T_AFTER3(w) : 0u;
            uint32_t id = SAGE_T_ID(w);
            if (id >= emb_rows) { if (i < n_tok) flags[FLAG_ROW_OOB] = 1; id = 0; }
            const double* row = emb + (size_t)id * dp;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const uint32_t idx = gl + (uint32_t)G * e;
                r[j][e] = (i < n_tok && idx < dp) ? row[idx] : 0.0;
            }
        }
        double mine = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int s = 1; s <= 2; ++s) {
                double d = 0.0;
#pragma unroll
                for (int e = 0; e < EPL; ++e) d = fma(r[j][e], r[j + s][e], d);
#pragma unroll
                for (int m = G / 2; m >= 1; m >>= 1) d += __shfl_xor(d, m, G);
                if ((int)gl == 2 * j + (s - 1)) mine = d;
            }
        }
        // lane gl holds dot (token j = gl/2, distance s = gl%2 + 1); G == 8 exactly covers the 8 dots
        if (G == 8 || gl < 8) {
            const uint32_t j = gl >> 1, s = (gl & 1u) + 1u;
            const bool valid = (i0 + j) < n_tok && after[j] >= s;
            const double v = valid ? log_sigmoid(mine) : 0.0;
            if ((i0 + j) < n_tok) reinterpret_cast<double*>(Gc)[(size_t)(i0 + j) * 2 + (s - 1)] = v;
        }
    }
}

// ----------------------------------------------------------------------------------------------------------
// k_score_desc: loss[a] += -2 * ( sum over ablated-window pairs of logsig(E.E)  -  sum over base-window pairs )
// (utils.py:125-138,255; every position pair stands for both (t,c) orders).  A block stages 256 descriptors in
// LDS; G lanes own one occurrence; the next row is in flight while the current one is being reduced; pending dots
// are spread over the G lanes so each log-sigmoid is evaluated once; the base window is a range sum over Gc.
// ----------------------------------------------------------------------------------------------------------
template <int G, int EPL>
__global__ __launch_bounds__(256, 4) void k_score_desc(const uint4* __restrict__ desc, const uint32_t* __restrict__ tok_w0,
                                                       uint32_t n_tok, const double* __restrict__ emb, uint32_t emb_rows,
                                                       uint32_t dp, const double2* __restrict__ Gc,
                                                       double* __restrict__ loss, uint32_t* __restrict__ flags) {
    constexpr uint32_t NG = 256 / G;                 // occurrences per round
    constexpr uint32_t DSLOTS = 2 * SAGE_DESC_MAXWIN; // >= 2L-3 pair dots of one window
    __shared__ uint4 tile[256 * 4];
    __shared__ double dots[2][NG][DSLOTS];
    __shared__ double bsum[2][NG];
    __shared__ uint32_t ndots[2][NG];
    __shared__ uint32_t acc_id[ACC_SLOTS];
    __shared__ double acc_val[ACC_SLOTS];
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256) { acc_id[i] = ACC_EMPTY; acc_val[i] = 0.0; }
    const uint32_t gl = threadIdx.x % G;
    const uint32_t grp = threadIdx.x / G;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(tile);
    const double* Gd = reinterpret_cast<const double*>(Gc);
    const uint32_t n_round = (n_tok + 255u) & ~255u;
    for (uint32_t base = blockIdx.x * 256u; base < n_round; base += gridDim.x * 256u) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t q = threadIdx.x + 256u * r;
            tile[q] = (base + (q >> 2)) < n_tok ? desc[(size_t)base * 4 + q] : make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        for (uint32_t round = 0; round < 256u / NG; ++round) {
            const uint32_t buf = round & 1u;
            // ---- phase 1 (this round): gather the ablated window's rows, pair dots -> LDS ----
            {
                const uint32_t row = round * NG + grp;
                const uint32_t t = base + row;
                const uint32_t* d = words + row * 16u;
                const uint32_t hdr = d[0];
                const bool active = (hdr & SAGE_DH_VALID) && !(hdr & SAGE_DH_OVER);
                const uint32_t k = SAGE_DH_K(hdr), m = SAGE_DH_M(hdr), pl = SAGE_DH_PL(hdr), pr = SAGE_DH_PR(hdr);
                const uint32_t L = active ? pl + k + pr : 0u;
                double sb = 0.0;
                if (active) {
                    // base window = range sums over the pair cache.  pair (i,i+1) counts iff i or i+1 is in the centre
                    // [t, t+m): rel = i-t in [-1, m-1] clipped to the window; pair (i,i+2) iff rel in [0,m-1] or
                    // rel+2 in [0,m-1] (not contiguous for m == 1: rel = -1 joins two pads).
                    const int im = (int)m, ipl = (int)pl, ipr = (int)pr;
                    const int lo1 = ipl >= 1 ? -1 : 0, hi1 = im - 2 + (ipr >= 1 ? 1 : 0);
                    const int lo2 = -ipl, hi2 = im + ipr - 3;
                    for (int rel = lo1 + (int)gl; rel <= hi1; rel += G) sb += Gd[((size_t)t + rel) * 2];
                    for (int rel = lo2 + (int)gl; rel <= hi2; rel += G)
                        if ((rel >= 0 && rel <= im - 1) || (rel + 2 >= 0 && rel + 2 <= im - 1))
                            sb += Gd[((size_t)t + rel) * 2 + 1];
                }
#pragma unroll
                for (int mm = G / 2; mm >= 1; mm >>= 1) sb += __shfl_xor(sb, mm, G);
                // Register ring of 6 row slices: rows j-2, j-1, j are in use while j+1..j+3 are in flight, so three
                // dependent-latency periods overlap per group.  Fully unrolled: every ring index is static.
                double ring[6][EPL];
                uint32_t nd = 0;
#define SAGE_FETCH(slot, jj)                                                          \
    {                                                                                 \
        uint32_t id_ = d[1 + (jj)];                                                   \
        if (id_ >= emb_rows) { flags[FLAG_ROW_OOB] = 1; id_ = 0; }                    \
        const double* rowp_ = emb + (size_t)id_ * dp;                                 \
        _Pragma("unroll") for (int e = 0; e < EPL; ++e) {                             \
            const uint32_t idx_ = gl + (uint32_t)G * e;                               \
            ring[slot][e] = idx_ < dp ? rowp_[idx_] : 0.0;                            \
        }                                                                             \
    }
                if (0 < L) SAGE_FETCH(0, 0)
                if (1 < L) SAGE_FETCH(1, 1)
                if (2 < L) SAGE_FETCH(2, 2)
#pragma unroll
                for (int j = 0; j < SAGE_DESC_MAXWIN; ++j) {
                    if ((uint32_t)j < L) {
                        if ((uint32_t)(j + 3) < L) SAGE_FETCH((j + 3) % 6, j + 3)
                        const bool c = (uint32_t)j >= pl && (uint32_t)j < pl + k;
                        if (j >= 1) {
                            const bool c1 = (uint32_t)(j - 1) >= pl && (uint32_t)(j - 1) < pl + k;
                            if (c || c1) {
                                double dd = 0.0;
#pragma unroll
                                for (int e = 0; e < EPL; ++e) dd = fma(ring[j % 6][e], ring[(j + 5) % 6][e], dd);
#pragma unroll
                                for (int mm = G / 2; mm >= 1; mm >>= 1) dd += __shfl_xor(dd, mm, G);
                                if (gl == 0) dots[buf][grp][nd] = dd;
                                ++nd;
                            }
                        }
                        if (j >= 2) {
                            const bool c2 = (uint32_t)(j - 2) >= pl && (uint32_t)(j - 2) < pl + k;
                            if (c || c2) {
                                double dd = 0.0;
#pragma unroll
                                for (int e = 0; e < EPL; ++e) dd = fma(ring[j % 6][e], ring[(j + 4) % 6][e], dd);
#pragma unroll
                                for (int mm = G / 2; mm >= 1; mm >>= 1) dd += __shfl_xor(dd, mm, G);
                                if (gl == 0) dots[buf][grp][nd] = dd;
                                ++nd;
                            }
                        }
                    }
                }
#undef SAGE_FETCH
                if (gl == 0) { ndots[buf][grp] = nd; bsum[buf][grp] = sb; }
            }
            __syncthreads();
            // ---- phase 2: log-sigmoid of every staged dot, one per lane; the row registers are dead here.  The two
            // LDS buffers alternate, so the next round's phase 1 may start without a second barrier. ----
            {
                {
                    const uint32_t nd = ndots[buf][grp];
                    double acc = 0.0;
                    for (uint32_t sidx = gl; sidx < nd; sidx += G) acc += log_sigmoid(dots[buf][grp][sidx]);
#pragma unroll
                    for (int mm = G / 2; mm >= 1; mm >>= 1) acc += __shfl_xor(acc, mm, G);
                    const uint32_t t = base + round * NG + grp;
                    if (gl == 0 && nd) acc_add_f64(acc_id, acc_val, loss, SAGE_T_ID(tok_w0[t]), -2.0 * (acc - bsum[buf][grp]));
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256)
        if (acc_id[i] != ACC_EMPTY) unsafeAtomicAdd(&loss[acc_id[i]], acc_val[i]);
}

// ----------------------------------------------------------------------------------------------------------
// k_count_desc: distinct (ablated, target, context) keys per count segment from descriptors (no walk).
// v1 back end: per-segment u64 open-addressed sets (see set_insert).
// ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_count_desc(const uint4* __restrict__ desc, const uint32_t* __restrict__ tok_w0,
                                                    uint32_t n_tok, SegPlan S, unsigned long long* __restrict__ set,
                                                    unsigned long long* __restrict__ stats, uint32_t* __restrict__ flags) {
    unsigned long long n_distinct = 0;
    const uint32_t n_round = (n_tok + 63u) & ~63u;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n_round; t += gridDim.x * 256u) {
        if (t >= n_tok) continue;
        uint32_t d[16];
        {
            const uint4 a = desc[(size_t)t * 4], b = desc[(size_t)t * 4 + 1], c = desc[(size_t)t * 4 + 2], e = desc[(size_t)t * 4 + 3];
            d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
            d[8] = c.x; d[9] = c.y; d[10] = c.z; d[11] = c.w; d[12] = e.x; d[13] = e.y; d[14] = e.z; d[15] = e.w;
        }
        const uint32_t hdr = d[0];
        if (!(hdr & SAGE_DH_VALID) || (hdr & SAGE_DH_OVER)) continue;
        const uint32_t k = SAGE_DH_K(hdr), m = SAGE_DH_M(hdr), pl = SAGE_DH_PL(hdr), pr = SAGE_DH_PR(hdr);
        const uint32_t L = pl + k + pr;
        const uint32_t seg = find_segment(S, t);
        const uint64_t base = S.set_base[seg], mask = S.set_mask[seg];
        const uint64_t akey = (uint64_t)SAGE_T_ID(tok_w0[t]) << 40;
        auto put = [&](uint32_t x, uint32_t y) {
            const uint32_t lo = min(x, y), hi = max(x, y);
            const uint32_t nw = set_insert(set, base, mask, akey | ((uint64_t)lo << 20) | hi, flags);
            n_distinct += nw ? (lo == hi ? 1u : 2u) : 0u;
        };
        // ablated window
#pragma unroll
        for (int j = 1; j < SAGE_DESC_MAXWIN; ++j) {
            if ((uint32_t)j < L) {
                const bool c = (uint32_t)j >= pl && (uint32_t)j < pl + k;
                const bool c1 = (uint32_t)(j - 1) >= pl && (uint32_t)(j - 1) < pl + k;
                if (c || c1) put(d[1 + j], d[j]);
                if (j >= 2) {
                    const bool c2 = (uint32_t)(j - 2) >= pl && (uint32_t)(j - 2) < pl + k;
                    if (c || c2) put(d[1 + j], d[j - 1]);
                }
            }
        }
        // base window: tokens t-pl .. t+m+pr-1, centre [t, t+m)
        const uint32_t w0 = t - pl, w1 = t + m + pr;  // exclusive end
        uint32_t p1 = 0, p2 = 0;
        for (uint32_t i = w0; i < w1; ++i) {
            const uint32_t id = SAGE_T_ID(tok_w0[i]);
            const bool c = i >= t && i < t + m;
            if (i >= w0 + 1) { const bool c1 = (i - 1) >= t && (i - 1) < t + m; if (c || c1) put(id, p1); }
            if (i >= w0 + 2) { const bool c2 = (i - 2) >= t && (i - 2) < t + m; if (c || c2) put(id, p2); }
            p2 = p1; p1 = id;
        }
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) n_distinct += __shfl_down(n_distinct, dd, 64);
    if ((threadIdx.x & 63u) == 0 && n_distinct) atomicAdd(&stats[STAT_DISTINCT], n_distinct);
}
