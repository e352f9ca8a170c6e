// Second-generation ablation kernels.  The occurrence walk runs ONCE per step (k_describe) and leaves a 64-byte
// descriptor per ablated occurrence; the distinct-key count and the f64 score then stream descriptors with no
// dependent pointer chase.  Scoring is split so that the gather never waits on transcendental code:
//   k_pair_cache  base-window pair terms, one row gather per base token
//   k_dots        ablated-window pair dot products: a barrier-free, continuously prefetched row-gather stream
//   k_logsig      log-sigmoid of every dot + base-window range sums -> loss[a]
#pragma once
#include "sage_kernels.h"

#define DESC_LDS_STRIDE 17  // 16 words + 1 pad: conflict-free per-thread row writes
#define SEG_LDS 128         // count segments of a batch whose pair totals k_describe accumulates in LDS
#define DESC_ACC_BITS 9    // k_describe's occurrence-count cache: 1,024 slots (LDS budget: 4 workgroups per CU)
#define DESC_ACC (1 << DESC_ACC_BITS)
#define DESC_SPAN 2048      // corpus positions of the match table one k_describe tile stages in LDS
enum { FLAG_POOL_FULL = 3, FLAG_BUCKET_OVERFLOW = 4,
       FLAG_NAN_PAIR = 5,     // not an error: a base token's float32 row is NaN-marked, the float64 pair pass has work
       FLAG_PART_MISMATCH = 6 };   // the distinct count's passes disagree with the per-occurrence pair counts (never seen; checked)

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
template <int BITS = 11>
__device__ __forceinline__ void acc_add_u32(uint32_t* cid, uint32_t* cval, unsigned long long* __restrict__ global, uint32_t id) {
    const uint32_t s = (id * 0x9E3779B1u) >> (32 - BITS);
    uint32_t cur = cid[s];
    if (cur == ACC_EMPTY) cur = atomicCAS(&cid[s], ACC_EMPTY, id), cur = (cur == ACC_EMPTY) ? id : cur;
    if (cur == id) atomicAdd(&cval[s], 1u);
    else atomicAdd(&global[id], 1ull);
}

#ifdef PHASE_CLOCKS
// debug build: shader-clock stamps at the phase boundaries of k_describe<false> ([0..8]) and k_logsig<false> ([16..23]), k_key_partition3<1> ([32..40])
__device__ unsigned long long g_phase_dbg[48];
#define PH_DECL unsigned long long ph_t = 0, ph_d[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PH_START() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ph_t = __builtin_amdgcn_s_memtime(); } while (0)
#define PH_MARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_d[i] += n_ - ph_t; ph_t = n_; } while (0)
#define PH_FLUSH(base, n) do { if (threadIdx.x == 0) for (int i_ = 0; i_ < (n); ++i_) atomicAdd(&g_phase_dbg[(base) + i_], ph_d[i_]); } while (0)
#else
#define PH_DECL
#define PH_START()
#define PH_MARK(i)
#define PH_FLUSH(base, n)
#endif

// ----------------------------------------------------------------------------------------------------------
// Deterministic mode (sage_set_deterministic): every log-sigmoid term is turned into a 128-bit fixed-point number
// (units of 2^-60) BEFORE anything is added, and all sums -- per thread, per occurrence, per workgroup, per ablated id
// -- are integer additions, which commute exactly: the loss vector no longer depends on the order in which atomics
// land, on the batch size or on how a workgroup's dots are cut into slices.  Terms that are not finite (a caller's NaN
// or infinite embedding value) bypass it and go straight into the float64 vector, where NaN + x = NaN in any order.
// ----------------------------------------------------------------------------------------------------------
struct Fix128 {
    unsigned long long lo;
    long long hi;
};
#define FIX_LIMIT 5.0e17    // |x| below 2^59: the fixed-point form holds it (and millions of such terms) without overflow
__device__ __forceinline__ Fix128 fix_from(double x) {      // x finite, |x| < FIX_LIMIT; truncates below 2^-60
    const double q = floor(x * 0.0625);
    const double rem = fma(-16.0, q, x);                    // exact, in [0, 16)
    return Fix128{(unsigned long long)(rem * 1152921504606846976.0), (long long)q};
}
__device__ __forceinline__ void fix_add(Fix128& a, const Fix128 b) {
    a.lo += b.lo;
    a.hi += b.hi + (a.lo < b.lo ? 1 : 0);
}
__device__ __forceinline__ Fix128 fix_neg(const Fix128 a) {
    return Fix128{~a.lo + 1ull, ~a.hi + (a.lo == 0ull ? 1 : 0)};
}
__device__ __forceinline__ Fix128 fix_twice(const Fix128 a) {
    return Fix128{a.lo << 1, (long long)(((unsigned long long)a.hi << 1) | (a.lo >> 63))};
}
// the same addition on two 64-bit words in LDS or global memory: the carry out of the low word is added exactly once
__device__ __forceinline__ void fix_atomic_add(unsigned long long* lo, unsigned long long* hi, const Fix128 v) {
    const unsigned long long old = atomicAdd(lo, v.lo);
    const unsigned long long carry = (old + v.lo) < old ? 1ull : 0ull;
    const unsigned long long add_hi = (unsigned long long)v.hi + carry;
    if (add_hi) atomicAdd(hi, add_hi);
}
__device__ __forceinline__ void acc_add_fix(uint32_t* cid, unsigned long long* clo, unsigned long long* chi,
                                            unsigned long long* __restrict__ global /* [id][2] */, uint32_t id, const Fix128 v) {
    const uint32_t s = acc_slot(id);
    uint32_t cur = cid[s];
    if (cur == ACC_EMPTY) cur = atomicCAS(&cid[s], ACC_EMPTY, id), cur = (cur == ACC_EMPTY) ? id : cur;
    if (cur == id) fix_atomic_add(&clo[s], &chi[s], v);
    else fix_atomic_add(&global[2 * (size_t)id], &global[2 * (size_t)id + 1], v);
}
// loss[i] += the fixed-point sum of id i (end of a deterministic step)
__global__ void k_fix_finish(const unsigned long long* __restrict__ fix, uint32_t n, double* __restrict__ loss) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long lo = fix[2 * (size_t)i];
    const long long hi = (long long)fix[2 * (size_t)i + 1];
    loss[i] += (double)hi * 16.0 + (double)lo * (1.0 / 1152921504606846976.0);
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
#ifndef SAGE_NI
#define SAGE_NI 2   // window ids in flight in front of the row ring of k_dots
#endif

// ----------------------------------------------------------------------------------------------------------
// The ablation walk of one occurrence (model.py:168-232), written as straight phases so that the lanes of a wave run
// the same code and differ only in trip counts (an element-at-a-time state machine makes every lane pay for every
// state any lane is in):
//   left pads (model.py:218-219)  ->  first piece = second-longest match at the token start (model.py:176-182)
//   ->  greedy pieces until one ends on a base-token start or at the line end (model.py:185-202; the line end is the
//       `beyond = n` case: no right pads, base span to the end of the line)
//   ->  right pads (model.py:220-223); the base span [loc, beyond) is counted along the way.
// emit(id) receives the ablated window in order; returning false from emit abandons the walk (window too long).
// ----------------------------------------------------------------------------------------------------------
struct Ablation {
    uint32_t a;          // ablated id
    uint32_t k, m;       // pieces, base tokens in the span
    uint32_t pl, pr;     // pads actually present
    bool never;          // the walk ran into the line end
    bool complete;       // false when emit() stopped it
};

// ldx(p) returns M[p].x; the kernels pass a loader that serves a staged span from LDS.
// The occurrence's own words (token word, match entry, the two tokens before it) are passed in: k_describe loads them
// while its span of M is still on the way instead of after it.
struct OccWords { uint32_t w0, prev1, prev2; uint2 m0; };
__device__ __forceinline__ OccWords load_occ_words(const WalkArrays& A, uint32_t t, uint32_t p0) {
    OccWords w;
    w.w0 = A.tok_w0[t];
    // the token's own match entry from its token word, its second-longest match = the parent of the longest: a lookup in a
    // 1 MiB table instead of a second 4-byte column of the match table
    (void)p0;
    w.m0 = make_uint2(SAGE_T_ID(w.w0) | (SAGE_T_LEN(w.w0) << 20) | (SAGE_T_AFTER3(w.w0) == 0u ? 0x80000000u : 0u),
                      A.parent[SAGE_T_ID(w.w0)]);
    w.prev1 = t >= 1u ? A.tok_w0[t - 1] : 0u;     // used only when the token has that many predecessors in its line
    w.prev2 = t >= 2u ? A.tok_w0[t - 2] : 0u;
    return w;
}
template <class LoadX, class Emit>
__device__ __forceinline__ Ablation walk_occurrence(const WalkArrays& A, const OccWords& ow, uint32_t p0, LoadX&& ldx, Emit&& emit) {
    Ablation r;
    const uint32_t w0 = ow.w0;
    const uint2 m0 = ow.m0;
    r.a = SAGE_T_ID(w0);
    const uint32_t loc3 = SAGE_T_LOC3(w0);
    r.pl = loc3 < 2u ? loc3 : 2u;                              // padleft = min(pad, loc), model.py:218
    r.k = 0; r.m = 1; r.pr = 0; r.never = false; r.complete = false;
    if (r.pl == 2u && !emit(SAGE_T_ID(ow.prev2))) return r;
    if (r.pl >= 1u && !emit(SAGE_T_ID(ow.prev1))) return r;
    // first piece: longest key of at most width-1 bytes = the second-longest match
    if (!emit(SAGE_M_ID(m0.y))) return r;
    r.k = 1;
    uint32_t q = p0 + SAGE_M_LEN(m0.y);
    uint32_t bcur = p0 + SAGE_M_LEN(m0.x);                     // start of the next base token not yet in the span
    bool blast = SAGE_M_ENDS(m0.x) != 0u;                      // the last counted base token ends the line
    bool rejoined = false;
    for (;;) {
        const uint32_t mq = ldx(q);
        if (!emit(SAGE_M_ID(mq))) return r;
        ++r.k;
        if (SAGE_M_ENDS(mq)) { r.never = true; break; }        // beyond = n (model.py:194-196)
        q += SAGE_M_LEN(mq);
        while (bcur < q) {                                     // on_base: is q a base-token start? (model.py:163-165,200)
            const uint32_t mb = ldx(bcur);
            blast = SAGE_M_ENDS(mb) != 0u;
            bcur += SAGE_M_LEN(mb); ++r.m;
        }
        if (bcur == q) { rejoined = true; break; }
    }
    if (rejoined) {
        const uint32_t m1 = ldx(q);
        if (!emit(SAGE_M_ID(m1))) return r;
        if (SAGE_M_ENDS(m1)) {
            r.pr = 1;                                          // padright = min(pad, ntok - beyond)
        } else {
            r.pr = 2;
            if (!emit(SAGE_M_ID(ldx(q + SAGE_M_LEN(m1))))) return r;
        }
    } else {
        while (!blast) {                                       // base span runs to the end of the line
            const uint32_t mb = ldx(bcur);
            blast = SAGE_M_ENDS(mb) != 0u;
            bcur += SAGE_M_LEN(mb); ++r.m;
        }
    }
    r.complete = true;
    return r;
}

template <class LoadX, class Emit>
__device__ __forceinline__ Ablation walk_occurrence(const WalkArrays& A, uint32_t t, uint32_t p0, LoadX&& ldx, Emit&& emit) {
    return walk_occurrence(A, load_occ_words(A, t, p0), p0, ldx, emit);
}

// Position pairs at distance 1 or 2 of a window [pl pads | c centre elements | pr pads] with at least one end in
// the centre (model.py:129-136), c >= 1.
__device__ __forceinline__ uint32_t window_pairs(uint32_t pl_, uint32_t c_, uint32_t pr_) {
    const int pl = (int)pl_, c = (int)c_, pr = (int)pr_;
    const int d1 = (c - 1) + (pl >= 1 ? 1 : 0) + (pr >= 1 ? 1 : 0);
    const int fwd = max(0, min(c, c + pr - 2));                                   // j in the centre, partner j+2
    const int bwd = max(0, min(pl - 1, pl + c - 3) - max(0, pl - 2) + 1);         // j a left pad, j+2 in the centre
    return (uint32_t)(d1 + fwd + bwd);
}

// ----------------------------------------------------------------------------------------------------------
// k_describe: one thread per ablated occurrence walks it once.
//   -> descriptor, ablated id, ablated_sizes histogram, occurrences per id, position pairs per count segment and per
//      occurrence (the distinct count's tiles), step statistics; with float32 row storage also the split of the
//      occurrences into those whose window touches a row the float32 copy cannot hold (scored from the float64 matrix)
//      and the rest (scored from the float32 copy): two lists, so that neither score kernel meets the other's work.
//   LONG = false  every occurrence; a window that needs more than the 15 inline ids (or a base span above
//                 SAGE_INLINE_MAX_M) is abandoned and goes on the long list;
//   LONG = true   one thread per long-list entry: counting walk, pool allocation, second walk that writes the ids.
// All list appends are aggregated per 256-thread tile: one global atomic per tile and list (a global atomic per wave on
// one hot counter costs ~8 ns each and serialises).
// ----------------------------------------------------------------------------------------------------------
struct DescribeArgs {
    uint4* desc;
    uint32_t* occ_a;              // [n_occ] ablated id | pl << 20 | pr << 22 | min(m, 127) << 24 | (touches an inexact row) << 31:
                                  // all k_logsig needs of an occurrence (m = 127: look at the descriptor)
    uint32_t* npairs_ab;          // [n_occ] position pairs of the ablated window (-> offsets of its dots)
    uint32_t* npairs_all;         // [n_occ] ablated + base position pairs; 0 for a giant
    uint32_t* giant_list; uint32_t* giant_count;
    uint32_t* pool; uint32_t* pool_cursor; uint32_t pool_cap;
    unsigned long long* seg_pairs;
    unsigned long long* occ;
    unsigned long long* hist; uint32_t hist_len;
    unsigned long long* stats;
    uint32_t* flags;
    uint32_t* long_list; uint32_t* long_count;
    const uint32_t* inexact;      // bitmap of the rows the float32 copy cannot hold; nullptr = one list-free score pass
    uint32_t* exact_list; uint32_t* exact_count;
    uint32_t* inex_list; uint32_t* inex_count;
};
#define SAGE_OCC_INEXACT 0x80000000u
#define SAGE_OCC_M_SEE_DESC 127u

template <bool LONG>
__global__ __launch_bounds__(256) void k_describe(WalkArrays A, const uint32_t* __restrict__ occ_tok, uint32_t n_occ,
                                                  SegPlan S, DescribeArgs D) {
    __shared__ uint32_t rows[LONG ? 1 : 256 * DESC_LDS_STRIDE];
    __shared__ uint32_t wcnt[3][5];
    __shared__ unsigned long long lseg[SEG_LDS];
    __shared__ uint32_t span[LONG ? 1 : DESC_SPAN];   // M[P0 .. P0 + DESC_SPAN).x of the tile: the walks read it from LDS
    __shared__ unsigned int lhist[32];
    __shared__ uint32_t occ_id[DESC_ACC];
    __shared__ uint32_t occ_n[DESC_ACC];
    if (threadIdx.x < 32) lhist[threadIdx.x] = 0;
    if (threadIdx.x < SEG_LDS) lseg[threadIdx.x] = 0ull;
    for (uint32_t i = threadIdx.x; i < DESC_ACC; i += 256) { occ_id[i] = ACC_EMPTY; occ_n[i] = 0; }
    __syncthreads();
    unsigned long long n_rows = 0, n_pairs = 0, n_wrows = 0, n_ab = 0, n_wrows_bad = 0;
    const uint32_t n_items = LONG ? *D.long_count : n_occ;
    const uint32_t n_round = (n_items + 255u) & ~255u;
    uint32_t* my = rows + (LONG ? 0 : threadIdx.x * DESC_LDS_STRIDE);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t* __restrict__ bitmap = D.inexact;
    PH_DECL;
    // A tile's loads are three dependent levels: (A) its occurrences' token index and position and the start of its span,
    // (B) each occurrence's token word and the two token words before it, (C) the parent of its token (= the second-longest
    // match) and the span of M.  Level A runs two tiles ahead and level B one tile ahead, so every level's addresses arrived
    // a whole tile ago and a tile starts with ONE batch of loads.  All of them are unconditional (clamped indices; the
    // predicate is applied where the value is used): a branch around a load makes the compiler wait for everything in
    // flight ("Reading the waits").
    const uint32_t tile_stride = gridDim.x * 256u;
    uint32_t a_t = 0, a_p0 = 0, a_P0 = 0;                                   // level A of tile + 2 strides
    uint32_t b_t = 0, b_p0 = 0, b_P0 = 0, b_w0 = 0, b_prev1 = 0, b_prev2 = 0;   // levels A, B of tile + 1 stride
    const uint32_t last_item = n_items ? n_items - 1u : 0u;
    auto level_a = [&](uint32_t bb) {
        a_t = occ_tok[min(bb + threadIdx.x, last_item)];
        a_p0 = A.occ_pos[min(bb + threadIdx.x, last_item)];
        a_P0 = A.occ_pos[min(bb, last_item)];
    };
    auto level_b = [&]() {       // promotes level A's tile and issues its level B
        b_t = a_t; b_p0 = a_p0; b_P0 = a_P0;
        b_w0 = A.tok_w0[b_t];
        b_prev1 = A.tok_w0[b_t >= 1u ? b_t - 1u : 0u];       // used only when the token has that many predecessors in its line
        b_prev2 = A.tok_w0[b_t >= 2u ? b_t - 2u : 0u];
    };
    if (!LONG && n_items) {
        level_a(blockIdx.x * 256u);
        level_b();
        level_a(blockIdx.x * 256u + tile_stride);
    }
    for (uint32_t base = blockIdx.x * 256u; base < n_round; base += gridDim.x * 256u) {
        PH_START();
        const uint32_t item = base + threadIdx.x;
        const bool live = item < n_items;
        const uint32_t o = !live ? 0u : (LONG ? D.long_list[item] : item);
        uint32_t P0 = 0, t_pre = 0, p0_pre = 0;
        OccWords ow{0u, 0u, 0u, make_uint2(0u, 0u)};
        if (!LONG) {
            // 256 consecutive occurrences cover ~1.5 KB of text; their walks stay within a few hundred bytes of it.
            // Every dependent step of a walk is then an LDS read instead of an L2 round trip.
            t_pre = b_t;
            p0_pre = b_p0;
            P0 = b_P0;
            ow.w0 = b_w0; ow.prev1 = b_prev1; ow.prev2 = b_prev2;
            // level C of this tile first (it is what the tile waits for), then the other tiles' levels
            uint32_t sp[DESC_SPAN / 256u];
            const uint32_t last_pos = A.n_pos ? A.n_pos - 1u : 0u;
#pragma unroll
            for (uint32_t q = 0; q < DESC_SPAN / 256u; ++q) sp[q] = A.M[min(P0 + threadIdx.x + 256u * q, last_pos)];
            const uint32_t par = A.parent[SAGE_T_ID(ow.w0)];
            level_b();
            level_a(base + 2u * tile_stride);
            ow.m0 = make_uint2(SAGE_T_ID(ow.w0) | (SAGE_T_LEN(ow.w0) << 20) | (SAGE_T_AFTER3(ow.w0) == 0u ? 0x80000000u : 0u), par);
#pragma unroll
            for (uint32_t q = 0; q < DESC_SPAN / 256u; ++q)
                span[threadIdx.x + 256u * q] = (P0 + threadIdx.x + 256u * q) < A.n_pos ? sp[q] : 0u;
            lds_barrier();       // (not __syncthreads(): the other tiles' loads stay in flight)
        }
        if (!LONG) PH_MARK(0);
        auto ldx = [&](uint32_t p) -> uint32_t {
            const uint32_t d = p - P0;
            if (!LONG && d < (uint32_t)DESC_SPAN) return span[d];
            uint32_t v = A.M[p];
            if (!LONG) asm volatile("" : "+v"(v));      // consumed inside the rare branch: the wait for it stays there
            return v;
        };
        const uint32_t t = LONG ? (live ? occ_tok[o] : 0u) : t_pre;
        const uint32_t p0 = LONG ? (live ? A.occ_pos[o] : 0u) : p0_pre;
        const uint32_t seg = live ? find_segment(S, t) : 0xFFFFFFFFu;
        unsigned long long my_pairs = 0;
        bool is_long = false, done = false, bad = false;
        if (!LONG) PH_MARK(1);
        if (live) {
            uint32_t L = 0;
            Ablation w;
            if (!LONG) {
                w = walk_occurrence(A, ow, p0, ldx, [&](uint32_t id) {
                    if (L == SAGE_DESC_MAXWIN) return false;     // does not fit: leave it to the long pass
                    my[1 + L] = id;
                    ++L;
                    return true;
                });
                is_long = !w.complete || w.m > SAGE_INLINE_MAX_M;
                PH_MARK(2);
            } else {
                w = walk_occurrence(A, t, p0, ldx, [&](uint32_t) { ++L; return true; });
            }
            if (is_long) {
                my[0] = SAGE_DH_VALID | SAGE_DH_OVER;  // placeholder, rewritten by the long pass
            } else {
                done = true;
                uint32_t hdr = SAGE_DH_VALID | (w.never ? SAGE_DH_NEVER : 0u) | (w.pl << 24) | (w.pr << 26);
                if (!LONG) {
                    my[0] = hdr | w.k | (w.m << 12);
                    if (bitmap) {    // independent loads of a 32 KB table, after the walk and all in flight together
                        uint32_t acc = 0;
#pragma unroll
                        for (uint32_t j = 0; j < SAGE_DESC_MAXWIN; ++j) {
                            const uint32_t id = my[1 + j];
                            const uint32_t bits = bitmap[j < L ? id >> 5 : 0u];       // (slots past L hold stale words)
                            acc |= j < L ? bits >> (id & 31u) : 0u;
                        }
                        bad = (acc & 1u) != 0u;
                    }
                    PH_MARK(3);
                } else {
                    uint32_t at = atomicAdd(D.pool_cursor, L);
                    const bool fits = at + L <= D.pool_cap && at + L >= at;
                    // no room: the step is redone with a larger pool; until then the window points at the pool's first
                    // L ids (in bounds: pool_cap >= the longest possible window), so every reader stays in bounds
                    if (!fits) at = 0;
                    if (fits) {
                        uint32_t j = 0;          // second walk of a long window: ids straight into the pool
                        walk_occurrence(A, t, p0, ldx, [&](uint32_t id) {
                            D.pool[at + j++] = id;
                            if (bitmap) bad = bad || ((bitmap[id >> 5] >> (id & 31u)) & 1u);
                            return true;
                        });
                    } else {
                        D.flags[FLAG_POOL_FULL] = 1;
                    }
                    hdr |= SAGE_DH_OVER;
                    if ((unsigned long long)window_pairs(w.pl, w.k, w.pr) + window_pairs(w.pl, w.m, w.pr) > (unsigned long long)SAGE_PART_GIANT)
                        hdr |= SAGE_DH_GIANT;
                    uint4* dst = D.desc + (size_t)o * 4;
                    dst[0] = make_uint4(hdr, at, w.k, w.m);
                    dst[1] = make_uint4(0, 0, 0, 0); dst[2] = make_uint4(0, 0, 0, 0); dst[3] = make_uint4(0, 0, 0, 0);
                }
                const uint32_t ab_pairs = window_pairs(w.pl, w.k, w.pr);
                my_pairs = (unsigned long long)ab_pairs + window_pairs(w.pl, w.m, w.pr);
                n_rows += L + w.m;
                n_wrows += L;
                n_wrows_bad += bad ? L : 0u;
                n_ab += ab_pairs;
                D.occ_a[o] = w.a | (w.pl << 20) | (w.pr << 22) | (min(w.m, SAGE_OCC_M_SEE_DESC) << 24) | (bad ? SAGE_OCC_INEXACT : 0u);
                D.npairs_ab[o] = ab_pairs;
                const bool giant = LONG && my_pairs > (unsigned long long)SAGE_PART_GIANT;
                D.npairs_all[o] = giant ? 0u : (uint32_t)my_pairs;
                if (giant) D.giant_list[atomicAdd(D.giant_count, 1u)] = o;      // rare: a line of thousands of tokens
                acc_add_u32<DESC_ACC_BITS>(occ_id, occ_n, D.occ, w.a);
                const uint32_t kk = w.k < D.hist_len ? w.k : D.hist_len - 1u;
                if (kk < 32u) atomicAdd(&lhist[kk], 1u); else atomicAdd(&D.hist[kk], 1ull);
            }
        }
        n_pairs += my_pairs;
        if (!LONG) PH_MARK(4);
        {   // list appends of the tile: abandoned walks -> long list; described occurrences -> exact / inexact lists
            const bool f[3] = {is_long, done && bitmap != nullptr && !bad, done && bitmap != nullptr && bad};
            uint32_t* const lists[3] = {D.long_list, D.exact_list, D.inex_list};
            uint32_t* const counts[3] = {D.long_count, D.exact_count, D.inex_count};
            unsigned long long bal[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                bal[q] = __ballot(f[q]);
                if (lane == 0) wcnt[q][wave] = (uint32_t)__popcll(bal[q]);
            }
            __syncthreads();
            if (threadIdx.x < 3) {
                const uint32_t q = threadIdx.x;
                const uint32_t tot = wcnt[q][0] + wcnt[q][1] + wcnt[q][2] + wcnt[q][3];
                const bool used = q == 0 ? !LONG : bitmap != nullptr;
                wcnt[q][4] = (tot && used) ? atomicAdd(counts[q], tot) : 0u;
            }
            __syncthreads();
            if (!LONG) PH_MARK(5);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (f[q] && (q != 0 || !LONG)) {
                    uint32_t at = wcnt[q][4];
                    for (uint32_t ww = 0; ww < wave; ++ww) at += wcnt[q][ww];
                    lists[q][at + (uint32_t)__popcll(bal[q] & ((1ull << lane) - 1ull))] = o;
                }
        }
        if (my_pairs) {   // pairs per segment: LDS for the first SEG_LDS segments of the batch, flushed once per workgroup
            if (seg < (uint32_t)SEG_LDS) atomicAdd(&lseg[seg], my_pairs);
            else atomicAdd(&D.seg_pairs[seg], my_pairs);
        }
        __syncthreads();
        if (!LONG) PH_MARK(6);
        if (!LONG) {
            // coalesced write-out of the block's 256 descriptors (4 x 16 B each)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t q = threadIdx.x + 256u * r;      // uint4 index inside the block's tile
                const uint32_t row = q >> 2, col = (q & 3u) * 4u;
                if (base + row < n_occ) {
                    const uint32_t* src = rows + row * DESC_LDS_STRIDE + col;
                    D.desc[(size_t)(base + row) * 4 + (q & 3u)] = make_uint4(src[0], src[1], src[2], src[3]);
                }
            }
            __syncthreads();
            PH_MARK(7);
#ifdef PHASE_CLOCKS
            ph_d[8] += 1;
#endif
        }
    }
    __syncthreads();
    if (!LONG) PH_FLUSH(0, 9);
    if (threadIdx.x < 32 && threadIdx.x < D.hist_len && lhist[threadIdx.x])
        atomicAdd(&D.hist[threadIdx.x], (unsigned long long)lhist[threadIdx.x]);
    for (uint32_t i = threadIdx.x; i < DESC_ACC; i += 256)
        if (occ_n[i]) atomicAdd(&D.occ[occ_id[i]], (unsigned long long)occ_n[i]);
    if (threadIdx.x < SEG_LDS && threadIdx.x < S.n_seg && lseg[threadIdx.x]) atomicAdd(&D.seg_pairs[threadIdx.x], lseg[threadIdx.x]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        n_rows += __shfl_down(n_rows, d, 64);
        n_pairs += __shfl_down(n_pairs, d, 64);
        n_wrows += __shfl_down(n_wrows, d, 64);
        n_ab += __shfl_down(n_ab, d, 64);
        n_wrows_bad += __shfl_down(n_wrows_bad, d, 64);
    }
    if (lane == 0) {
        if (n_rows) atomicAdd(&D.stats[STAT_ROWS], n_rows);
        if (n_pairs) atomicAdd(&D.stats[STAT_PAIRS], n_pairs);
        if (n_wrows) atomicAdd(&D.stats[STAT_WROWS], n_wrows);
        if (n_ab) atomicAdd(&D.stats[STAT_ABPAIRS], n_ab);
        if (n_wrows_bad) atomicAdd(&D.stats[STAT_WROWS_INEXACT], n_wrows_bad);
    }
    if (LONG && blockIdx.x == 0 && threadIdx.x == 0 && n_items) atomicAdd(&D.stats[STAT_LONG], (unsigned long long)n_items);
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
// A float32 row loaded with the float64 slice's element mapping (lane gl, load i: elements 2*gl + 2*G*i + {0,1}) into
// the bit storage of the first EPL/2 doubles of the slice; widen_slice() turns it into EPL doubles in place.  For
// k_dots<double, MIXED>: the rows of an inexact occurrence's window that the float32 copy does hold.
template <int G, int EPL>
__device__ __forceinline__ void load_row_slice_f32_bits(const float* __restrict__ emb, uint32_t id, uint32_t dp, uint32_t gl,
                                                        double (&r)[EPL]) {
    const uint2* row = reinterpret_cast<const uint2*>(emb + (size_t)id * dp) + gl;
#pragma unroll
    for (int i = 0; i < EPL / 2; ++i) {
        const uint2 v = row[(size_t)G * i];
        r[i] = __hiloint2double((int)v.y, (int)v.x);
    }
}
template <int EPL>
__device__ __forceinline__ void widen_slice(double (&r)[EPL]) {
#pragma unroll
    for (int i = EPL / 2 - 1; i >= 0; --i) {
        const float lo = __int_as_float(__double2loint(r[i])), hi = __int_as_float(__double2hiint(r[i]));
        r[2 * i] = (double)lo;
        r[2 * i + 1] = (double)hi;
    }
}
// (bound_ctrl: every lane of these patterns has a source lane, and without it the compiler zeroes the destination
// register before every DPP move)
// Sum over the G lanes of a group, result in every lane.  The first levels are DPP moves inside a 16-lane row
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror): VALU only, no trip through the LDS crossbar.
__device__ __forceinline__ double dpp_add(double v, const int ctrl_sel) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int plo, phi;
    switch (ctrl_sel) {
    case 0: plo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); phi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;
    case 1: plo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); phi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;
    case 2: plo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); phi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break;
    default: plo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); phi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break;
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
        if (FIX && flags[FLAG_NAN_PAIR] == 0u) return;   // the float32 pass met no marked row: nothing to recompute
        if (FIX) {
            uint32_t mine_id = ids[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) mine_id = ((int)gl == j) ? ids[j] : mine_id;
            const bool flagged = gl < 6u && ((inexact[mine_id >> 5] >> (mine_id & 31u)) & 1u);
            if (!group_any<G>(flagged)) continue;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) load_row_slice<G, EPL>(emb, ids[j], dp, gl, r[j]);
        if constexpr (!FIX && std::is_same<T, float>::value) {
            // a NaN-marked row among the base tokens (lane 0 of the group holds element 0 of every row): tell the float64
            // pass it has work.  A plain idempotent store -- base tokens are tokens the corpus produces, so with
            // word2vec-trained embeddings this never fires and the float64 pass returns at once.
            bool bad = false;
#pragma unroll
            for (int j = 0; j < 6; ++j) bad = bad || (i0 + (uint32_t)j < n_tok && r[j][0] != r[j][0]);   // (padding reads row 0)
            if (bad && gl == 0) flags[FLAG_NAN_PAIR] = 1u;
        }
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
// flight while row j is reduced, the next occurrence's words come from a branch-free two-stage prefetch (below), and
// there is no barrier in the loop, no LDS beyond one counter and no transcendental code in this kernel, so a CU keeps
// >100 KB of row gathers outstanding -- provided nothing in the loop waits for ALL vector memory (`vmcnt(0)`).
// ----------------------------------------------------------------------------------------------------------
// MIXED (T = double, the float64 pass over the inexact list): only the rows marked in `bitmap` come from the float64
// matrix; the other rows of such a window (all but one of its ~7, on average) come from the float32 copy `emb32` --
// half the bytes -- and are widened when they become current.  The bitmap word of an id is fetched one step before
// its row load is issued, so the choice costs no exposed latency (one more id in flight than the plain kernel).
template <int G, int EPL, typename T, int NS, bool MIXED = false>
__device__ __forceinline__ void dots_body(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                              uint32_t n_occ, const uint32_t* __restrict__ dot_off,
                                              const T* __restrict__ emb, uint32_t emb_rows, uint32_t dp,
                                              double* __restrict__ dots, uint32_t* __restrict__ flags,
                                              const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count,
                                              const float* __restrict__ emb32 = nullptr,
                                              const uint32_t* __restrict__ bitmap = nullptr,
                                              unsigned long long* __restrict__ stats = nullptr) {
    static_assert(!MIXED || std::is_same<T, double>::value, "MIXED is a variant of the float64 kernel");
    uint32_t n_marked = 0;      // MIXED: rows this group took from the float64 matrix (a statistic for the byte model)
    // list != nullptr: only the *list_count occurrences on the list are processed (float64 fix-up pass of the float32
    // fast path)
    const uint32_t gl = threadIdx.x % G;
    const uint32_t n = list ? *list_count : n_occ;
    // Each element travelling through the pipeline carries: its row slice and a meta word
    //   bit0: dot with the previous element, bit1: dot with the one before, bit2: first element of an occurrence,
    //   bit3: element exists.  `out` is only read at a first element.
    // Ring of NS row slices; roles rotate, data never moves: the current row, the two before it, and NF = NS - 3 rows
    // whose loads are in flight (fully unrolled: every index below is a compile-time constant).  In front of the rows
    // runs a queue of NI window ids whose loads are in flight, so the dependent chain id -> row address -> row never
    // stalls a step: an element enters as an id load, NI steps later its row load is issued, NF steps later it is
    // reduced against its two predecessors.
    constexpr int NF = NS - 3, NI = SAGE_NI + (MIXED ? 1 : 0), NQ = NF + NI;
    static_assert(NF >= 1, "ring too short");
    T B[NS][EPL];
    // float32 storage: a row is widened to float64 once, when it becomes current, and kept that way for the two steps
    // it serves as a predecessor (3 live rows; widening inside the dots costs 32 conversions per step instead of 8)
    constexpr bool WIDEN = !std::is_same<T, double>::value;
    double C[WIDEN ? NS : 1][EPL];
    uint32_t qid[NI];           // ids in flight, oldest first
    uint32_t qm[NQ], qo[NQ];    // meta / output offset of every element in flight (rows first, then ids), oldest first
#pragma unroll
    for (int i = 0; i < NQ; ++i) { qm[i] = 0; qo[i] = 0; }
#pragma unroll
    for (int i = 0; i < NI; ++i) qid[i] = 0;
    // id side: fi = position (in the list, or plain index) of the next occurrence to open
    // Occurrences are handed out in windows of 32 consecutive ones, the windows round-robin over the workgroups (the
    // chip as a whole reads one moving window of descriptors, offsets and dots), and inside its window a workgroup's
    // groups draw the next occurrence from an LDS counter as they finish: 22.9 -> 21.8 ms against a fixed round-robin
    // over all groups of the grid, whose groups drift apart as window lengths differ, and 289 -> 198 ms at D = 300, where
    // a re-read row is 1.2 KB (windows of 32 rather than one per group there: another 1.7 %).  Larger windows change
    // nothing; a contiguous run per group (32k separate streams of descriptors, offsets and dots) took 40 ms and a
    // contiguous run per workgroup 36 ms.
    __shared__ uint32_t next_occ;
    if (threadIdx.x == 0) next_occ = 0u;
    __syncthreads();
    constexpr uint32_t DOTS_WIN = 256u / G < 32u ? 32u : 256u / G;
    const uint32_t fend = n;
    // The open occurrence: element index fj of fL; which elements take a dot with their predecessor / the one before it is two
    // RANGES of fj (an element or its partner must lie in the centre [pl, pl + k), and k >= 2 always: an ablation has a
    // first piece and at least one more): [max(pl, 1), pl + k] and [2, pl + k + 1].
    uint32_t fi = 0, fj = 0, fL = 0, flo1 = 1, fspan1 = 0, fem1 = 0;
    auto next_index = [&]() {
        uint32_t tk = 0;
        if constexpr (MIXED) {
            // One LDS atomic per wave for all its groups that open an occurrence in this step, and no shuffle: every lane of
            // an opening group (the group's lanes branch together) counts the LAST lanes of the opening groups below its
            // own, so the lanes of a group derive the same ticket.  The shuffle of the other branch needs a lane-address
            // register that lives through the whole loop; this kernel, held to 128 registers, spilled it and paid a scratch
            // reload -- a full `vmcnt(0)` drain -- at every opening.  (In the float32 kernel, which has the register, the
            // shuffle form is the faster one: 18.9 against 19.3 ms.)
            const unsigned long long m = __ballot(gl == (uint32_t)(G - 1));
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            uint32_t base = 0;
            if (rank == 0u && gl == (uint32_t)(G - 1)) base = atomicAdd(&next_occ, (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
            tk = base + rank;
        } else {
            if (gl == 0u) tk = atomicAdd(&next_occ, 1u);
            tk = __shfl(tk, (int)((threadIdx.x & 63u) & ~(uint32_t)(G - 1)), 64);       // (the group's lanes branch together)
        }
        // (32-bit: a batch holds fewer than 2^32 bytes, hence fewer than 2^31 occurrences of 2+ bytes, and a group draws at
        // most one ticket past the end of the list)
        const uint32_t at = (tk / DOTS_WIN * gridDim.x + blockIdx.x) * DOTS_WIN + tk % DOTS_WIN;
        fi = at < n ? at : n;
    };
    next_index();
    const uint32_t* fids = nullptr;
    // The NEXT occurrence's words -- list entry, descriptor head, dot offset -- are a chain of two dependent loads.  They
    // are fetched by a two-stage pipeline that runs ONCE PER STEP, unconditionally and outside every branch: stage 1 loads
    // the list entry of position fi, stage 2 (a step later) the descriptor head and dot offset of that entry.  Between two
    // openings fi does not change, so the steps re-read the same cached words; what this buys is that the opening branch
    // contains no load at all.  (With the loads inside that branch the compiler had to wait for ALL outstanding vector
    // memory -- `s_waitcnt vmcnt(0)`, row gathers included -- twice per opening plus once per loop trip, because a value
    // loaded on one side of a divergent branch is copied at the join: 1.6 full drains per step of a wave.)
    uint32_t a_foi = 0;                                  // stage 1: list entry of position fi (in flight for one step)
    uint32_t b_foi = 0, b_hx = 0, b_out = 0;             // stage 2: entry, descriptor word 0 and dot offset
    uint32_t age = 0;                                    // pipeline steps since fi last changed; >= 2: stage 2 belongs to fi
    const uint32_t* desc_w = reinterpret_cast<const uint32_t*>(desc);
    auto prefetch_next = [&]() {
        b_foi = a_foi < n_occ ? a_foi : 0u;              // (stage 1 of a position past the end read entry 0)
        b_hx = desc_w[(size_t)b_foi * 16];
        b_out = dot_off[b_foi];
        const uint32_t p = fi < fend ? fi : 0u;
        a_foi = list ? list[p] : p;
        age += age < 2u ? 1u : 0u;
    };
    auto fetch_id = [&](uint32_t& id, uint32_t& meta, uint32_t& out) {
        meta = 0; id = 0;
        if (fj == fL) {   // open the next occurrence from the prefetched words
            if (fi >= fend) return;
            uint32_t o = b_foi, hx = b_hx, ov = b_out;
            if (age < 2u) {   // not there yet: a group's first occurrence (the pipeline needs two steps)
                o = list ? list[fi] : fi;
                hx = desc_w[(size_t)o * 16];
                ov = dot_off[o];
                // consumed here, inside the branch, so that the wait for these loads stays inside it
                asm volatile("" : "+v"(hx), "+v"(ov));
            }
            uint32_t hy = 0, hz = 0;
            if (hx & SAGE_DH_OVER) {   // a pooled window (1.5 % of the occurrences): its pool offset and length, loaded here
                hy = desc_w[(size_t)o * 16 + 1];
                hz = desc_w[(size_t)o * 16 + 2];
                asm volatile("" : "+v"(hy), "+v"(hz));
            }
            const Win w = win_decode(hx, hz, 0u);
            fL = w.L; fj = 0;
            flo1 = w.pl == 2u ? 2u : 1u;
            fspan1 = w.pl + w.k - flo1;
            fem1 = w.pl + w.k - 1u;
            fids = w.pooled ? pool + hy : desc_w + (size_t)o * 16 + 1;
            out = ov;
            next_index();
            age = 0;
        }
        id = fids[fj];
        meta = (fj == 0u ? 12u : 8u) | ((fj - flo1) <= fspan1 ? 1u : 0u) | ((fj - 2u) <= fem1 ? 2u : 0u);     // (unsigned compares)
        ++fj;
    };
    // dots leave eight at a time: lane i of the group keeps the i-th pending dot, one 64-byte store per eight dots
    // (or per occurrence tail) instead of one 8-byte store per dot
    uint32_t wr = 0, pend = 0;   // write cursor into dots[] (first pending dot), pending dots
    double mydot = 0.0;
    if (n == 0u) return;
#pragma unroll
    for (int i = 0; i < NI; ++i) { fetch_id(qid[i], qm[NF + i], qo[NF + i]); prefetch_next(); }
    uint32_t bw = 0;       // MIXED: bitmap word of the id whose row load is issued next
    if constexpr (MIXED) bw = bitmap[(qid[0] < emb_rows ? qid[0] : 0u) >> 5];
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
            prefetch_next();
            if (rid >= emb_rows) { if (rlive) flags[FLAG_ROW_OOB] = 1; rid = 0; }
            if constexpr (MIXED) {
                const bool from32 = ((bw >> (rid & 31u)) & 1u) == 0u;
                bw = bitmap[(qid[0] < emb_rows ? qid[0] : 0u) >> 5];           // for the next step's row
                if (from32) {
                    load_row_slice_f32_bits<G, EPL>(emb32, rid, dp, gl, B[(s + NF) % NS]);
                    qm[NF - 1] |= 16u;                                       // bit4: the slice holds float32 bits
                } else {
                    load_row_slice<G, EPL>(emb, rid, dp, gl, B[(s + NF) % NS]);
                    n_marked += (rlive && gl == 0u) ? 1u : 0u;
                }
            } else {
                load_row_slice<G, EPL>(emb, rid, dp, gl, B[(s + NF) % NS]);
            }
            if constexpr (MIXED) { if (m0 & 16u) widen_slice<EPL>(B[s]); }
            if constexpr (WIDEN) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) C[s][e] = (double)B[s][e];
            }
            if (m0 & 8u) {
                if (m0 & 4u) {   // a new occurrence: its dots go elsewhere, so the previous one's tail leaves now
                    if (gl < pend) dots[wr + gl] = mydot;
                    wr = out0; pend = 0;
                }
                if (m0 & 1u) {
                    double d_;
                    if constexpr (WIDEN) d_ = group_dot<G, EPL, double>(C[s], C[(s + NS - 1) % NS]);
                    else d_ = group_dot<G, EPL, T>(B[s], B[(s + NS - 1) % NS]);
                    if (gl == pend) mydot = d_;
                    if (++pend == 8u) { if (gl < 8u) dots[wr + gl] = mydot; wr += 8u; pend = 0; }
                }
                if (m0 & 2u) {
                    double d_;
                    if constexpr (WIDEN) d_ = group_dot<G, EPL, double>(C[s], C[(s + NS - 2) % NS]);
                    else d_ = group_dot<G, EPL, T>(B[s], B[(s + NS - 2) % NS]);
                    if (gl == pend) mydot = d_;
                    if (++pend == 8u) { if (gl < 8u) dots[wr + gl] = mydot; wr += 8u; pend = 0; }
                }
            }
        }
    }
    if (gl < pend) dots[wr + gl] = mydot;
    if constexpr (MIXED) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) n_marked += __shfl_down(n_marked, d, 64);
        if ((threadIdx.x & 63u) == 0u && n_marked && stats) atomicAdd(&stats[STAT_MARKED_ROWS], (unsigned long long)n_marked);
    }
}

template <int G, int EPL, typename T, int NS>
__global__ __launch_bounds__(256) void k_dots(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                              uint32_t n_occ, const uint32_t* __restrict__ dot_off,
                                              const T* __restrict__ emb, uint32_t emb_rows, uint32_t dp,
                                              double* __restrict__ dots, uint32_t* __restrict__ flags,
                                              const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count) {
    dots_body<G, EPL, T, NS, false>(desc, pool, n_occ, dot_off, emb, emb_rows, dp, dots, flags, list, list_count, nullptr, nullptr);
}
// (held to 128 registers = 4 waves per SIMD: the compiler's own choice is 143 and 3 waves, which loses more than the
// bytes saved; every float64 instantiation has EPL <= 8 and fits without spilling -- the float32 kernel for D = 300,
// EPL = 12, does not, which is why the plain kernel carries no such attribute)
template <int G, int EPL, int NS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_dots_mixed(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool, uint32_t n_occ,
                  const uint32_t* __restrict__ dot_off, const double* __restrict__ emb, uint32_t emb_rows, uint32_t dp,
                  double* __restrict__ dots, uint32_t* __restrict__ flags, const uint32_t* __restrict__ list,
                  const uint32_t* __restrict__ list_count, const float* __restrict__ emb32, const uint32_t* __restrict__ bitmap,
                  unsigned long long* __restrict__ stats) {
    dots_body<G, EPL, double, NS, true>(desc, pool, n_occ, dot_off, emb, emb_rows, dp, dots, flags, list, list_count, emb32, bitmap, stats);
}

// ----------------------------------------------------------------------------------------------------------
// Measurement only (sage_measure_unique_rows): distinct embedding rows per window of `win` consecutive occurrences of a
// list -- the rows a gather kernel with that much reuse in flight MUST fetch (k_dots hands its occurrences out in windows
// of 32).  One wave per window; a per-wave LDS set with generation tags, so nothing is cleared between windows.  A
// window with more ids than the set holds counts the excess as distinct (an upper bound; such windows are rare).
// ----------------------------------------------------------------------------------------------------------
#define UR_SLOTS 4096u
__global__ __launch_bounds__(256) void k_unique_rows(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                                     const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count,
                                                     uint32_t n_occ, uint32_t win, unsigned long long* __restrict__ out /*[2]: distinct, ids*/) {
    __shared__ uint32_t set_all[4 * UR_SLOTS];
    uint32_t* set = set_all + (threadIdx.x >> 6) * UR_SLOTS;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = lane; i < UR_SLOTS; i += 64u) set[i] = 0u;
    const uint32_t n = list ? *list_count : n_occ;
    const uint32_t n_win = (n + win - 1u) / win;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    unsigned long long distinct = 0, ids_seen = 0;
    uint32_t gen = 0;
    for (uint32_t w = wave; w < n_win; w += n_waves) {
        gen = (gen + 1u) & 0xFFFu;
        if (gen == 0u) { for (uint32_t i = lane; i < UR_SLOTS; i += 64u) set[i] = 0u; gen = 1u; }
        __builtin_amdgcn_wave_barrier();
        uint32_t filled = 0;
        for (uint32_t i = w * win; i < min(n, (w + 1u) * win); ++i) {
            const uint32_t o = list ? list[i] : i;
            const uint4 h = desc[(size_t)o * 4];
            const Win ww = win_decode(h.x, h.z, h.w);
            const uint32_t* ids = ww.pooled ? pool + h.y : reinterpret_cast<const uint32_t*>(desc + (size_t)o * 4) + 1;
            for (uint32_t j = lane; j < ww.L; j += 64u) {
                const uint32_t id = ids[j] & SAGE_ID_MASK;
                ++ids_seen;
                if (filled > UR_SLOTS / 2u) { ++distinct; continue; }        // (wave-uniform: set nearly full)
                const uint32_t want = id | (gen << 20);
                uint32_t s = (id * 0x9E3779B1u) >> 20;                       // 12 bits
                for (;;) {
                    const uint32_t cur = set[s];
                    if (cur == want) break;
                    if ((cur >> 20) != gen) {
                        const uint32_t old = atomicCAS(&set[s], cur, want);
                        if (old == cur) { ++distinct; break; }
                        if (old == want) break;
                        continue;                                           // someone else took the slot: look again
                    }
                    s = (s + 1u) & (UR_SLOTS - 1u);
                }
            }
            filled += ww.L;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { distinct += __shfl_down(distinct, d, 64); ids_seen += __shfl_down(ids_seen, d, 64); }
    if (lane == 0u) { atomicAdd(&out[0], distinct); atomicAdd(&out[1], ids_seen); }
}

// ----------------------------------------------------------------------------------------------------------
// k_logsig: loss[a] += -2 * ( sum_{ablated pairs} logsig(dot)  -  sum_{base pairs} G )   (utils.py:125-138,255; every
// position pair stands for both (t,c) orders).  Window lengths are heavy-tailed, so the dots are NOT walked per
// occurrence: a workgroup takes 256 consecutive occurrences, cuts their contiguous run of dots into 256 equal slices
// (one per thread, whatever occurrence they belong to), and every thread adds its partial sums into per-occurrence
// LDS accumulators, following the occurrence boundaries as it goes.  A second phase (one thread per occurrence) adds
// the base window -- two index ranges of the pair cache -- and accumulates per ablated id.
// Base spans are heavy-tailed too (a span that never rejoins runs to the end of its line), and one lane summing 100
// entries holds its wave for 100 dependent reads: phase clocks put 35 % of the kernel there.  The tile's occurrences
// are consecutive tokens, so the pair-cache entries of tokens t0-2 .. t0+509 (t0 = the tile's first token) are
// loaded a tile ahead, two per thread, turned into prefix sums in LDS, and a range becomes two reads.  The prefix
// sums of a tile (1,024 entries, each below LOGSIG_PS_LIMIT = 64 in magnitude, i.e. dots above -64) stay below 2^16,
// so a range read as a difference of two of them is off by a few ulp(2^16) = 1.5e-11: < 2e-10 ABSOLUTE per occurrence in
// the worst case, ~1e-12 on word2vec-sized dots (an absolute bound, not a relative one: DESIGN.md, Oracle and parity);
// a tile that holds a larger or non-finite entry, a span that leaves the staged range, and deterministic mode (which adds fixed-point
// terms, not float64 sums) add the entries one by one instead.
// ----------------------------------------------------------------------------------------------------------
#define LOGSIG_TILE 256
#define LOGSIG_PF 8      // dots of a thread's slice loaded ahead (a tile of 256 occurrences has ~8 dots per thread)
#define LOGSIG_WIN 6     // DET: pair-cache entries t-2 .. t+3 loaded ahead (centres of up to 4 base tokens)
#define LOGSIG_GS 512    // pair-cache entries of a tile staged as prefix sums
#define LOGSIG_PS_LIMIT 64.0
template <bool DET>
__global__ __launch_bounds__(256) void k_logsig(const uint4* __restrict__ desc, const uint32_t* __restrict__ occ_tok,
                                                const uint32_t* __restrict__ occ_a, uint32_t n_occ,
                                                const uint32_t* __restrict__ dot_off, const double* __restrict__ dots,
                                                const double2* __restrict__ Gc, uint32_t n_tok, double* __restrict__ loss,
                                                unsigned long long* __restrict__ loss_fix /* DET: [vsize][2] fixed-point sums */,
                                                const uint32_t* __restrict__ inex_count, unsigned long long* __restrict__ stats) {
    // Every dot is in place when this runs: with float32 row storage k_describe has split the occurrences, the ones
    // whose rows the float32 copy holds exactly got their dots from it, the others from the float64 matrix.  A NaN dot
    // is therefore the caller's own NaN (or inf * 0) and goes into the loss as the reference's would.
    if (inex_count && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[STAT_INEXACT], (unsigned long long)*inex_count);
    __shared__ uint32_t acc_id[ACC_SLOTS];
    __shared__ unsigned long long acc_w0[ACC_SLOTS];               // float64 sums, or the low words of the fixed-point sums
    __shared__ unsigned long long acc_w1[DET ? ACC_SLOTS : 1];     // DET: the high words
    __shared__ uint32_t off[LOGSIG_TILE + 1];
    __shared__ unsigned long long osum0[LOGSIG_TILE];
    __shared__ unsigned long long osum1[DET ? LOGSIG_TILE : 1];
    __shared__ double2 Ps[DET ? 1 : LOGSIG_GS + 1];                // Ps[j] = sum of the staged entries before entry j
    __shared__ double2 wtot[4];
    __shared__ uint32_t tile_bad[2];                               // by tile parity: the prefix sums of this tile are not to be used
    double* acc_val = reinterpret_cast<double*>(acc_w0);
    double* osum = reinterpret_cast<double*>(osum0);
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256) { acc_id[i] = ACC_EMPTY; acc_w0[i] = 0ull; if (DET) acc_w1[i] = 0ull; }
    if (threadIdx.x < 2) tile_bad[threadIdx.x] = 0u;
    const double* Gd = reinterpret_cast<const double*>(Gc);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t stride = gridDim.x * LOGSIG_TILE;
    // DET: a term that is not finite cannot be held in fixed point; it goes into the float64 vector of its ablated id
    auto stray = [&](uint32_t o, double term) { unsafeAtomicAdd(&loss[occ_a[o] & SAGE_ID_MASK], term); };
    // the per-occurrence words of the NEXT tile are loaded while this one is processed (a first level of dependent loads
    // off every tile's critical path)
    uint32_t nx_off = 0, nx_end = 0, nx_a = 0, nx_t = 0;
    auto prefetch = [&](uint32_t o0) {
        if (o0 >= n_occ) return;
        const uint32_t nt = min((uint32_t)LOGSIG_TILE, n_occ - o0);
        if (threadIdx.x < nt) { nx_off = dot_off[o0 + threadIdx.x]; nx_a = occ_a[o0 + threadIdx.x]; nx_t = occ_tok[o0 + threadIdx.x]; }
        if (threadIdx.x == 0) nx_end = dot_off[o0 + nt];
    };
    // !DET: so are its pair-cache entries, which takes the first token of the tile after next
    double2 ng0 = make_double2(0.0, 0.0), ng1 = ng0;
    uint32_t t0_staged = 0, t0_after = 0;     // first token of the tile whose entries are in ng0/ng1; of the tile after it
    auto first_token = [&](uint32_t o0) -> uint32_t { return o0 < n_occ ? occ_tok[o0] : 0u; };
    auto stage = [&](uint32_t t0) {
        const long long i0 = (long long)t0 - 2 + 2 * (long long)threadIdx.x;
        ng0 = (i0 >= 0 && i0 < (long long)n_tok) ? Gc[i0] : make_double2(0.0, 0.0);
        ng1 = (i0 + 1 >= 0 && i0 + 1 < (long long)n_tok) ? Gc[i0 + 1] : make_double2(0.0, 0.0);
        t0_staged = t0;
    };
    prefetch(blockIdx.x * LOGSIG_TILE);
    if (!DET) {
        stage(first_token(blockIdx.x * LOGSIG_TILE));
        t0_after = first_token(blockIdx.x * LOGSIG_TILE + stride);
    }
    PH_DECL;
    uint32_t par = 0;
    for (uint32_t o0 = blockIdx.x * LOGSIG_TILE; o0 < n_occ; o0 += stride, par ^= 1u) {
        PH_START();
        const uint32_t nt = min((uint32_t)LOGSIG_TILE, n_occ - o0);   // occurrences of this tile
        __syncthreads();
        if (threadIdx.x < nt) off[threadIdx.x] = nx_off;
        if (threadIdx.x == 0) off[nt] = nx_end;             // nt can be 256: there is no thread 256
        osum0[threadIdx.x] = 0ull;
        if (DET) osum1[threadIdx.x] = 0ull;
        const uint32_t o = o0 + threadIdx.x;
        uint32_t t = 0, a_w0 = 0;
        if (threadIdx.x < nt) { a_w0 = nx_a; t = nx_t; }
        const double2 g0 = ng0, g1 = ng1;
        const uint32_t t0 = t0_staged;
        prefetch(o0 + stride);
        double2 excl = make_double2(0.0, 0.0), run = excl;
        if (!DET) {
            if (o0 + stride < n_occ) stage(t0_after);
            t0_after = first_token(o0 + 2u * stride);
            // prefix sums of this tile's entries: thread totals scanned per wave, wave totals through LDS
            const bool big = !(fabs(g0.x) < LOGSIG_PS_LIMIT) || !(fabs(g0.y) < LOGSIG_PS_LIMIT) ||
                             !(fabs(g1.x) < LOGSIG_PS_LIMIT) || !(fabs(g1.y) < LOGSIG_PS_LIMIT);       // NaN counts as big
            if (big) tile_bad[par] = 1u;
            if (threadIdx.x == 0) tile_bad[par ^ 1u] = 0u;       // last read before this tile's first barrier
            run = make_double2(g0.x + g1.x, g0.y + g1.y);
            const double2 own = run;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const double ux = __shfl_up(run.x, d, 64), uy = __shfl_up(run.y, d, 64);
                if (lane >= (uint32_t)d) { run.x += ux; run.y += uy; }
            }
            if (lane == 63u) wtot[wave] = run;
            excl = make_double2(run.x - own.x, run.y - own.y);
        }
        __syncthreads();
        PH_MARK(0);
        if (!DET) {
            for (uint32_t w = 0; w < wave; ++w) { excl.x += wtot[w].x; excl.y += wtot[w].y; }
            if (threadIdx.x == 0) Ps[0] = make_double2(0.0, 0.0);
            Ps[2u * threadIdx.x + 1u] = make_double2(excl.x + g0.x, excl.y + g0.y);
            Ps[2u * threadIdx.x + 2u] = make_double2(excl.x + g0.x + g1.x, excl.y + g0.y + g1.y);     // read after the next barrier
        }
        // pads and base-span length ride in the ablated-id word: the 64-byte descriptor is read only for a span of
        // 127 tokens or more (a strided 16-byte read per occurrence otherwise: 4x its bytes in lines)
        Win w;
        w.pl = (a_w0 >> 20) & 3u; w.pr = (a_w0 >> 22) & 3u; w.m = (a_w0 >> 24) & 127u; w.k = 0; w.L = 0; w.pooled = false;
        if (w.m == SAGE_OCC_M_SEE_DESC) { const uint4 h = desc[(size_t)o * 4]; w = win_decode(h.x, h.z, h.w); }
        // base window: pair (i,i+1) counts iff i or i+1 is in the centre [t, t+m): rel = i-t in [-1, m-1] clipped to
        // the window [t-pl, t+m+pr-1]; pair (i,i+2) iff rel in [0,m-1] or rel+2 in [0,m-1] (not contiguous for
        // m == 1: rel = -1 joins two pads).
        const int im = (int)w.m, ipl = (int)w.pl, ipr = (int)w.pr;
        const int lo1 = ipl >= 1 ? -1 : 0, hi1 = im - 2 + (ipr >= 1 ? 1 : 0);
        const int lo2 = -ipl, hi2 = im + ipr - 3;
        const bool live = threadIdx.x < nt;
        const uint32_t d0 = off[0], d1 = off[nt];
        const uint32_t per = (d1 - d0 + 255u) / 256u;
        uint32_t i = d0 + threadIdx.x * per;
        const uint32_t iend = min(d1, i + per);
        // base-window sum of this thread's occurrence: a float64 sum, or (DET) a fixed-point sum of its terms
        double sb = 0.0;
        Fix128 sbf{0ull, 0};
        auto base_term = [&](double g) {
            if (!DET) { sb += g; return; }
            if (fabs(g) < FIX_LIMIT) fix_add(sbf, fix_from(g)); else stray(o, 2.0 * g);
        };
        auto base_terms_one_by_one = [&]() {
            for (int rel = lo1; rel <= hi1; ++rel) base_term(Gd[((size_t)t + rel) * 2]);
            for (int rel = lo2; rel <= hi2; ++rel)
                if ((rel >= 0 && rel <= im - 1) || (rel + 2 >= 0 && rel + 2 <= im - 1)) base_term(Gd[((size_t)t + rel) * 2 + 1]);
        };
        double v[LOGSIG_PF];
        if constexpr (DET) {
            // loads of both phases issued together: the pair-cache window of this thread's occurrence and the first
            // LOGSIG_PF dots of this thread's slice -- one exposed latency per tile
            const bool small = im <= LOGSIG_WIN - 2;     // rel in [-2, LOGSIG_WIN - 3] covers every term
            double2 W[LOGSIG_WIN];
            if (live && small) {
#pragma unroll
                for (int q = 0; q < LOGSIG_WIN; ++q) {
                    const int rel = q - 2;
                    W[q] = (rel >= min(lo1, lo2) && rel <= max(hi1, hi2)) ? Gc[(size_t)t + rel] : make_double2(0.0, 0.0);
                }
            }
#pragma unroll
            for (int q = 0; q < LOGSIG_PF; ++q) v[q] = (i + q < iend) ? dots[i + q] : 0.0;
            if (live) {
                if (small) {
#pragma unroll
                    for (int q = 1; q < LOGSIG_WIN; ++q) { const int rel = q - 2; if (rel >= lo1 && rel <= hi1) base_term(W[q].x); }
#pragma unroll
                    for (int q = 0; q < LOGSIG_WIN; ++q) {
                        const int rel = q - 2;
                        const bool in = rel >= lo2 && rel <= hi2 && ((rel >= 0 && rel <= im - 1) || (rel + 2 >= 0 && rel + 2 <= im - 1));
                        if (in) base_term(W[q].y);
                    }
                } else {
                    base_terms_one_by_one();
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < LOGSIG_PF; ++q) v[q] = (i + q < iend) ? dots[i + q] : 0.0;
        }
        PH_MARK(1);
        PH_MARK(2);
        if (i < iend) {
            // owner of dot i: largest j with off[j] <= i
            uint32_t lo = 0, hi = nt;
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (off[mid] <= i) lo = mid; else hi = mid;
            }
            uint32_t own = lo, nxt = off[own + 1];
            PH_MARK(3);
            double acc = 0.0;
            Fix128 accf{0ull, 0};
            auto hand_over = [&]() {
                if (!DET) unsafeAtomicAdd(&osum[own], acc);
                else if (accf.lo | (unsigned long long)accf.hi) fix_atomic_add(&osum0[own], &osum1[own], accf);
                acc = 0.0; accf = Fix128{0ull, 0};
            };
            auto one_dot = [&](double x) {
                if (i >= nxt) {
                    hand_over();
                    do { ++own; nxt = off[own + 1]; } while (i >= nxt);
                }
                const double ls = log_sigmoid(x);
                if (!DET) acc += ls;
                else if (fabs(ls) < FIX_LIMIT) fix_add(accf, fix_from(ls));
                else stray(o0 + own, -2.0 * ls);
            };
            // the dots loaded ahead, each from its own register (picking v[q] by a run-time q costs 21 selects per dot), then
            // the rest of a long slice from memory
#pragma unroll
            for (int q = 0; q < LOGSIG_PF; ++q)
                if (i < iend) { one_dot(v[q]); ++i; }
            for (; i < iend; ++i) one_dot(dots[i]);
            hand_over();
        }
        PH_MARK(4);
        __syncthreads();
        PH_MARK(5);
        if (live) {
            if (!DET) {
                // the base window from the tile's prefix sums: Ps index of rel = 0 is t - t0 + 2
                const uint32_t dj = t - t0;                                     // (wraps to a large value should t < t0)
                const int top = max(max(hi1, hi2), 0);
                const bool staged = tile_bad[par] == 0u && dj < (uint32_t)LOGSIG_GS && (long long)dj + 2 + top + 1 <= (long long)LOGSIG_GS;
                if (staged) {
                    const int j0 = (int)dj + 2;
                    sb = Ps[j0 + hi1 + 1].x - Ps[j0 + lo1].x;                    // (an empty range has hi1 = lo1 - 1)
                    if (im >= 2) {
                        sb += Ps[j0 + hi2 + 1].y - Ps[j0 + lo2].y;               // rel in [lo2, hi2], all of it counted
                    } else {                                                    // m == 1: rel = -2 and rel = 0 only
                        if (ipl == 2) sb += Ps[j0 - 1].y - Ps[j0 - 2].y;
                        if (ipr == 2) sb += Ps[j0 + 1].y - Ps[j0].y;
                    }
                } else {
                    base_terms_one_by_one();
                }
                acc_add_f64(acc_id, acc_val, loss, SAGE_T_ID(a_w0), -2.0 * (osum[threadIdx.x] - sb));
            } else {      // -2 * (ablated - base) = 2 * (base - ablated), all in fixed point
                Fix128 tot = sbf;
                fix_add(tot, fix_neg(Fix128{osum0[threadIdx.x], (long long)osum1[threadIdx.x]}));
                acc_add_fix(acc_id, acc_w0, acc_w1, loss_fix, SAGE_T_ID(a_w0), fix_twice(tot));
            }
        }
        PH_MARK(6);
#ifdef PHASE_CLOCKS
        ph_d[7] += 1;
#endif
    }
    if (!DET) PH_FLUSH(16, 8);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256)
        if (acc_id[i] != ACC_EMPTY) {
            if (!DET) unsafeAtomicAdd(&loss[acc_id[i]], acc_val[i]);
            else fix_atomic_add(&loss_fix[2 * (size_t)acc_id[i]], &loss_fix[2 * (size_t)acc_id[i] + 1], Fix128{acc_w0[i], (long long)acc_w1[i]});
        }
}

// ----------------------------------------------------------------------------------------------------------
// k_embed_layout: the matrix as train_embeddings leaves it (contiguous [rows, dim] float64, uploaded as is) into the
// device layout: float64 rows padded to whole 128-byte lines, the float32 copy, and the bitmap of the rows the copy
// cannot hold exactly (embeddings.py:81 fills rows from float32 word2vec vectors, embeddings.py:86 the rest from a
// float64 uniform draw).  A wave per row; a row with any element that does not survive float64 -> float32 -> float64
// (NaN included) is marked and its float32 row set to NaN.  (The host did this element by element and uploaded the rows
// with a pitched copy: 0.96 s per call at 262,144 x 50, six times per README schedule.)
// ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_embed_layout(const double* __restrict__ src, uint32_t rows, uint32_t dim, uint32_t dp,
                                                      double* __restrict__ emb, float* __restrict__ e32 /* or nullptr */,
                                                      uint32_t* __restrict__ bitmap /* zeroed */,
                                                      unsigned long long* __restrict__ n_inexact /* zeroed */) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (r >= rows) return;
    bool exact = true;
    for (uint32_t d = lane; d < dp; d += 64u) {
        const double v = d < dim ? src[(size_t)r * dim + d] : 0.0;
        emb[(size_t)r * dp + d] = v;
        exact = exact && (double)(float)v == v;        // false for NaN: such rows keep their float64 form
    }
    const bool row_exact = __ballot(!exact) == 0ull;
    if (lane == 0u && !row_exact) atomicAdd(n_inexact, 1ull);
    if (e32 == nullptr) return;                        // float64-only storage: the count alone
    const float qnan = __builtin_nanf("");
    for (uint32_t d = lane; d < dp; d += 64u) {
        const float f = d < dim ? (float)src[(size_t)r * dim + d] : 0.0f;
        e32[(size_t)r * dp + d] = (row_exact || d >= dim) ? f : qnan;      // a marked row is NaN in the copy (padding stays 0)
    }
    if (lane == 0u && !row_exact) atomicOr(&bitmap[r >> 5], 1u << (r & 31u));
}
