// k_dots2: the dot products of the ablated windows (the job of k_dots, sage_kernels2.h; model.py:129-136 + utils.py:125-138)
// with the bookkeeping done ACROSS the lanes of a group instead of once per element step.
//
// Why: k_dots was measured to be bound by instruction issue, not by its gathers -- with the row loads removed (a timing
// ablation) it ran as long as with them, and its ~165 vector instructions per element step, of which 42 are arithmetic,
// times 4 cycles, times the step count, IS its run time.  The other ~120 were the per-step upkeep of a stream of
// elements: queue shifts, the meta word, "open the next occurrence" inside a divergent branch that 69 % of a wave's steps
// enter because its eight groups open at different steps.
//
// Here a group of G lanes still owns one stream of elements (the windows of its occurrences, back to back), but it plans
// P elements at a time, one per lane: lane e of the group works out WHICH element the e-th next one is (occurrence,
// index in its window, whether it takes a dot with its predecessor / the one before, where those dots go), loads its id,
// and the P element steps that follow contain nothing but: fetch the row offset of the element NF steps ahead from its
// lane (`ds_bpermute`, the LDS crossbar -- not a vector ALU instruction), issue its row load, widen the current row, two
// dots, and a select that leaves dot t in lane t.  After P steps every lane stores its element's dots.  No branch in the
// loop but its exit; every load unconditional (so every `s_waitcnt vmcnt` is a counted one).
//
// The occurrences a group will need are kept the same way: a ring of 8 upcoming occurrences, one per lane (head word,
// pool offset, dot offset, length), topped up every period through two in-flight stages (list entry -> head + offset), so
// the dependent loads of "the next occurrence" are a period or two old when they are needed.  Occurrences are handed out
// as k_dots does (windows of 32 consecutive ones round-robin over the workgroups, tickets from an LDS counter inside).
//
// A period that cannot be filled (the ring has not been loaded yet: the first two periods of a kernel, or a run of
// two-id windows longer than the ring) runs with invalid elements AFTER the last element of an occurrence, never inside
// one (a dot needs the two rows before it in the ring).
#pragma once

#define DOTS2_HUGE 0x10000000u
// Occurrences a group keeps asked for, the one it is in included (<= 8, the ring).  A ticket drawn early is an occurrence
// processed long after its neighbours in the text, which other groups draw later -- and neighbours share window rows
// that the L2 holds for a few microseconds only.  Measured per 300k lines (D = 50 / D = 300): 8 -> 4.48 / 21.1 ms, 6 -> 4.38,
// 4 -> 4.26 / 20.4, 3 -> 4.16 / 20.2, 2 -> 4.09 / 21.4: the periods that run short of elements with a small value (the
// top-up takes two periods) cost less than the re-fetched rows -- the kernel waits for its gathers, not for its lanes.
#ifndef DOTS2_AHEAD
#define DOTS2_AHEAD 3
#endif

template <int G, int EPL, typename T>
__device__ __forceinline__ void load_row_at(const char* base, uint32_t off, T (&r)[EPL]) {
    if constexpr (std::is_same<T, float>::value) {
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(base + off + (uint32_t)(G * 16 * i));
            r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < EPL / 2; ++i) {
            const double2 v = *reinterpret_cast<const double2*>(base + off + (uint32_t)(G * 16 * i));
            r[2 * i] = v.x; r[2 * i + 1] = v.y;
        }
    }
}

// The two dots of an element step, a[].b[] and a[].c[], as two interleaved dependence chains (each dot in the order of
// group_dot: the same sums as k_dots; a wave issues in order, and a chain of eight dependent float64 multiply-adds
// followed by the next dot's eight leaves the pipeline half empty).
template <int G, int EPL>
__device__ __forceinline__ void group_dot_pair(const double (&a)[EPL], const double (&b)[EPL], const double (&c)[EPL], double& d1, double& d2) {
    double x = 0.0, y = 0.0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { x = fma(a[e], b[e], x); y = fma(a[e], c[e], y); }
    x = dpp_add(x, 0); y = dpp_add(y, 0);
    x = dpp_add(x, 1); y = dpp_add(y, 1);
    if (G >= 8) { x = dpp_add(x, 2); y = dpp_add(y, 2); }
    if (G >= 16) { x = dpp_add(x, 3); y = dpp_add(y, 3); }
#pragma unroll
    for (int m = 16; m < G; m <<= 1) { x += __shfl_xor(x, m, G); y += __shfl_xor(y, m, G); }
    d1 = x; d2 = y;
}

// MIXED (T = double, the pass over the inexact list): a row marked in `bitmap` comes from the float64 matrix, any other
// from the float32 copy -- half the bytes -- as in k_dots_mixed.  Here the choice costs no branch: a lane owns EPL
// CONSECUTIVE elements of a row (32 bytes of a float32 row, 64 of a float64 row, at twice the offset), every step issues
// the same 16-byte loads from a per-lane base (the second half of a float32 slice re-reads the first: a cache hit), and the
// widening picks per lane between "convert the first EPL words" and "the words are the doubles".  The mark of an element
// is looked up one period after its id has arrived (one more batch in flight: Eq).
template <int G, int EPL, typename T, int P, bool LIST, bool MIXED>
__device__ __forceinline__ void dots2_body(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                           uint32_t n_occ, const uint32_t* __restrict__ dot_off,
                                           const T* emb /* not __restrict__: see the row loads */, uint32_t emb_rows, uint32_t dp,
                                           double* dots, uint32_t* __restrict__ flags,
                                           const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count,
                                           const float* emb32, const uint32_t* __restrict__ bitmap,
                                           unsigned long long* __restrict__ stats) {
    static_assert(G >= 8 && P >= 4 && P <= 8, "a group plans P <= 8 elements and keeps 8 occurrences, one per lane");
    static_assert(!MIXED || (std::is_same<T, double>::value && EPL % 4 == 0), "MIXED: float64 kernel, slices of whole 16-byte loads");
    constexpr int NF = P - 3;              // rows in flight in front of the current one and its two predecessors
    constexpr int K = 1 + P / 2;           // occurrences a period of P elements can touch (windows have >= 2 ids)
    constexpr bool WIDEN = MIXED || !std::is_same<T, double>::value;
    constexpr int NW = MIXED ? 2 * EPL : 1;      // MIXED: words of a raw slice (a float64 slice; a float32 slice fills half)
    constexpr uint32_t IDLE_END = MIXED ? 4u : 3u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gl = threadIdx.x % G;
    const uint32_t n = LIST ? *list_count : n_occ;
    if (n == 0u) return;
    __shared__ uint32_t next_occ;
    if (threadIdx.x == 0) next_occ = 0u;
    __syncthreads();
    const uint32_t next_occ_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&next_occ;
    constexpr uint32_t WIN = 64u;     // consecutive occurrences a workgroup draws from (k_dots: 32; here 64 runs 0.5-2 % faster, 128 no better)
    const uint32_t gbase4 = (lane & ~(uint32_t)(G - 1)) << 2;
    auto bc = [&](uint32_t v, uint32_t idx) -> uint32_t {      // lane `idx` of the group, to every lane of the group
        return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(gbase4 + (idx << 2)), (int)v);
    };
    const char* ebase = reinterpret_cast<const char*>(emb);
    const uint32_t row_bytes = dp * (uint32_t)sizeof(T);
    const uint32_t slice = gl * 16u;
    const uint32_t* desc_w = reinterpret_cast<const uint32_t*>(desc);

    // ring of upcoming occurrences, entry (seq & 7) in lane (seq & 7) of the group
    uint32_t F_o = 0, F_hx = 0, F_hy = 0, F_hz = 0, F_out = 0, F_L = DOTS2_HUGE;
    // top-up stages: A = list entry in flight, B = head and dot offset in flight (bit0: this lane takes, bit1: a real occurrence)
    uint32_t A_o = 0, A_fl = 0, A_end = 0;
    uint32_t B_o = 0, B_fl = 0, B_end = 0, B_out = 0;
    uint4 B_h = make_uint4(0, 0, 0, 0);
    // the group's place in its stream (the same in all its lanes)
    uint32_t seq_cur = 0, fj = 0, seq_req = 0, seq_loaded = 0, idle = 0;
    // element batches: period r (E0), period r+1 (E1), planned with the id in flight (Ep); word = id | flags << 24
    // (bit24: dot with the previous element, bit25: with the one before, bit26: element exists), dst = where its dots go
    uint32_t E0 = 0, E0_dst = 0, E1 = 0, E1_dst = 0, Ep_id = 0, Ep_fl = 0, Ep_dst = 0;
    uint32_t Eq = 0, Eq_dst = 0, Eq_bw = 0;      // MIXED: the batch between Ep and E1 (ids there, bitmap words in flight)
    uint32_t oob = 0, wn = 0, n_marked = 0;
    double k1 = 0.0, k2 = 0.0;
    T B[MIXED ? 1 : P][EPL];
    uint32_t R[MIXED ? P : 1][NW], Mk[MIXED ? P : 1];
    double C[WIDEN ? P : 1][EPL];
#pragma unroll
    for (int s = 0; s < (MIXED ? 1 : P); ++s)
#pragma unroll
        for (int e = 0; e < EPL; ++e) B[s][e] = (T)0;
#pragma unroll
    for (int s = 0; s < (MIXED ? P : 1); ++s) {
        Mk[s] = 0;
#pragma unroll
        for (int e = 0; e < NW; ++e) R[s][e] = 0;
    }
#pragma unroll
    for (int s = 0; s < (WIDEN ? P : 1); ++s)
#pragma unroll
        for (int e = 0; e < EPL; ++e) C[s][e] = 0.0;

    for (;;) {
        // (1) the dots of the period that has just run leave: lane t holds those of its element t
        if (E0 & (1u << 24)) dots[E0_dst] = k1;
        if (E0 & (1u << 25)) dots[E0_dst + ((E0 >> 24) & 1u)] = k2;
        // (2) (3) the batches move up; the planned one gets its ids
        E0 = E1; E0_dst = E1_dst;
        {
            uint32_t id = Ep_id;
            if (id >= emb_rows) { oob |= (Ep_fl >> 2) & 1u; id = 0u; }
            if constexpr (MIXED) {
                const uint32_t marked = (Eq_bw >> (Eq & 31u)) & 1u;          // (bit 27 of the word: the row is a float64 row)
                n_marked += marked & (Eq >> 26);
                E1 = Eq | (marked << 27);
                E1_dst = Eq_dst;
                Eq = id | (Ep_fl << 24);
                Eq_dst = Ep_dst;
                Eq_bw = bitmap[id >> 5];
            } else {
                E1 = id | (Ep_fl << 24);
                E1_dst = Ep_dst;
            }
        }
        // (4) stage B has landed: into the ring
        {
            const bool take = (B_fl & 1u) != 0u;
            const bool pooled = (B_h.x & SAGE_DH_OVER) != 0u;
            const uint32_t k = pooled ? B_h.z : SAGE_DH_K(B_h.x);
            const uint32_t L = (B_fl & 2u) ? SAGE_DH_PL(B_h.x) + k + SAGE_DH_PR(B_h.x) : DOTS2_HUGE;
            F_o = take ? B_o : F_o;
            F_hx = take ? B_h.x : F_hx;
            F_hy = take ? B_h.y : F_hy;
            F_hz = take ? k : F_hz;          // (the window's k, pooled or not)
            F_out = take ? B_out : F_out;
            F_L = take ? L : F_L;
            seq_loaded = B_end;
        }
        // (5) plan the P elements of the period after next
        {
            const uint32_t avail = seq_loaded - seq_cur;
            uint32_t Lk[K], ck[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t v = bc(F_L, (seq_cur + (uint32_t)k) & 7u);
                Lk[k] = (uint32_t)k < avail ? v : DOTS2_HUGE;
            }
            ck[0] = Lk[0] - fj;
#pragma unroll
            for (int k = 1; k < K; ++k) ck[k] = ck[k - 1] + Lk[k];
            idle = (avail != 0u && Lk[0] >= DOTS2_HUGE) ? idle + 1u : 0u;     // the list has ended for this group
            uint32_t q = 0;
#pragma unroll
            for (int k = 0; k + 1 < K; ++k) q += gl >= ck[k] ? 1u : 0u;
            uint32_t start = 0, Lq = Lk[0];
#pragma unroll
            for (int k = 1; k < K; ++k) { start = q == (uint32_t)k ? ck[k - 1] : start; Lq = q == (uint32_t)k ? Lk[k] : Lq; }
            uint32_t j = gl - start + (q == 0u ? fj : 0u);
            const bool valid = gl < (uint32_t)P && Lq < DOTS2_HUGE;
            const uint32_t idx = (seq_cur + q) & 7u;
            const uint32_t hx = bc(F_hx, idx), hy = bc(F_hy, idx), kk = bc(F_hz, idx), out = bc(F_out, idx);
            uint32_t o = bc(F_o, idx);
            // where the group stands after these P elements (from lane P-1)
            const uint32_t wl = bc(q | (valid ? 256u : 0u), P - 1), jl = bc(j, P - 1), Ll = bc(Lq, P - 1);
            {
                const uint32_t ql = wl & 255u;
                const bool okl = (wl >> 8) != 0u, wrap = okl && jl + 1u == Ll;
                seq_cur += ql + (wrap ? 1u : 0u);
                fj = (okl && !wrap) ? jl + 1u : 0u;
            }
            const uint32_t pl = SAGE_DH_PL(hx);
            const uint32_t flo1 = pl == 2u ? 2u : 1u;
            const uint32_t top = pl + kk;                                  // last element that takes a dot with its predecessor
            const bool has1 = valid && (j - flo1) <= (top - flo1);         // (unsigned compares: j in [flo1, pl + k])
            const bool has2 = valid && (j - 2u) <= (top - 1u);             // j in [2, pl + k + 1]
            const uint32_t before = (uint32_t)max((int)(j - flo1), 0) + (uint32_t)max((int)(j - 2u), 0);
            Ep_dst = out + before;
            Ep_fl = (has1 ? 1u : 0u) | (has2 ? 2u : 0u) | (valid ? 4u : 0u);
            bool pooled = (hx & SAGE_DH_OVER) != 0u;
            if (!valid) { pooled = false; o = 0u; j = 0u; }
            const uint32_t* ip = pooled ? pool + (size_t)hy + j : desc_w + (size_t)o * 16 + 1 + j;
            Ep_id = *ip;
        }
        // (6) stage A has landed: its heads and dot offsets take off
        {
            B_fl = A_fl; B_end = A_end;
            B_o = (A_fl & 2u) ? A_o : 0u;
            B_h = desc[(size_t)B_o * 4];
            B_out = dot_off[B_o];
        }
        // (7) ask for as many occurrences as the ring has room for
        {
            const uint32_t n_free = max((int)(seq_cur + (uint32_t)DOTS2_AHEAD - seq_req), 0);
            const uint32_t rel = (gl - seq_req) & 7u;
            const bool take = gl < 8u && rel < n_free;
            uint32_t base = 0;
            // (the instruction itself: as `atomicAdd` the compiler combines the eight adds of a wave in a loop over its lanes)
            if (gl == 0u && n_free != 0u)
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(base) : "v"(next_occ_lds), "v"(n_free) : "memory");
            base = bc(base, 0u);
            const uint32_t tk = base + rel;
            const uint32_t at = (tk / WIN * gridDim.x + blockIdx.x) * WIN + tk % WIN;
            const bool real = take && at < n;
            if constexpr (LIST) A_o = list[real ? at : 0u];
            else A_o = at;
            A_fl = (take ? 1u : 0u) | (real ? 2u : 0u);
            seq_req += n_free;
            A_end = seq_req;
        }

        // the P element steps of this period
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < P; ++t) {
            {
                const int tn = (t + NF) % P;
                // the word of the element whose row takes off now was asked for a step ago (below); the wait is written
                // here because the compiler does not know of the instruction
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wn));
                if constexpr (MIXED) {
                    const uint32_t m = (wn >> 27) & 1u;
                    Mk[tn] = m;
                    const uint32_t off = (__umul24(wn, dp * 4u) + gl * (uint32_t)(EPL * 4)) << m;    // the float64 row and slice: twice the offset
                    const char* a0 = (m ? ebase : reinterpret_cast<const char*>(emb32)) + off;
                    const char* a1 = a0 + (m ? (uint32_t)(EPL * 4) : 0u);
#pragma unroll
                    for (int i = 0; i < EPL / 4; ++i) {
                        const uint4 v = *reinterpret_cast<const uint4*>(a0 + 16 * i);
                        R[tn][4 * i] = v.x; R[tn][4 * i + 1] = v.y; R[tn][4 * i + 2] = v.z; R[tn][4 * i + 3] = v.w;
                    }
#pragma unroll
                    for (int i = 0; i < EPL / 4; ++i) {
                        const uint4 v = *reinterpret_cast<const uint4*>(a1 + 16 * i);
                        R[tn][EPL + 4 * i] = v.x; R[tn][EPL + 4 * i + 1] = v.y; R[tn][EPL + 4 * i + 2] = v.z; R[tn][EPL + 4 * i + 3] = v.w;
                    }
                } else {
                    const uint32_t off = __umul24(wn, row_bytes) + slice;     // (the low 24 bits of the word: the id)
                    load_row_at<G, EPL, T>(ebase, off, B[tn]);
                }
            }
            // (straight-line code: without a compiler-level memory barrier the instruction selector's own ordering sinks
            // each row load to just above its first use, NF steps later, to save registers -- and the gather latency is
            // exposed; `sched_barrier` alone binds only the machine scheduler, which runs after that)
            asm volatile("" ::: "memory");
            {   // ... and the next step's is asked for: its trip through the LDS crossbar runs beside this step's arithmetic
                // (issued by the compiler it sits right in front of its use, and the wave, which issues in order, waits it out)
                if (t + 1 < P) {
                    if (t + 1 + NF < P) asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(wn) : "v"(gbase4), "v"(E0), "n"(((t + 1 + NF) % P) * 4));
                    else asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(wn) : "v"(gbase4), "v"(E1), "n"(((t + 1 + NF) % P) * 4));
                } else {
                    asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(wn) : "v"(gbase4), "v"(E1), "n"(NF * 4));   // step 0 of the next period
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const int tp = (t + P - 1) % P, tpp = (t + P - 2) % P;
            // the current row passes through an (empty) asm statement: asm statements keep their order, so the arithmetic of
            // step t, which hangs on this one, stays below the load issue of step t (pinned by the barrier above) -- otherwise
            // the steps of a period, one basic block, are interleaved at will and rows are consumed right after their loads
            double d1, d2;
            if constexpr (MIXED) {
#pragma unroll
                for (int e = 0; e < NW; ++e) asm volatile("" : "+v"(R[t][e]));
                const bool m = Mk[t] != 0u;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const double from32 = (double)__uint_as_float(R[t][e]);
                    const double from64 = __hiloint2double((int)R[t][2 * e + 1], (int)R[t][2 * e]);
                    C[t][e] = m ? from64 : from32;
                }
                group_dot_pair<G, EPL>(C[t], C[tp], C[tpp], d1, d2);
            } else if constexpr (WIDEN) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) asm volatile("" : "+v"(B[t][e]));
#pragma unroll
                for (int e = 0; e < EPL; ++e) C[t][e] = (double)B[t][e];
                group_dot_pair<G, EPL>(C[t], C[tp], C[tpp], d1, d2);
            } else {
#pragma unroll
                for (int e = 0; e < EPL; ++e) asm volatile("" : "+v"(B[t][e]));
                group_dot_pair<G, EPL>(B[t], B[tp], B[tpp], d1, d2);
            }
            k1 = gl == (uint32_t)t ? d1 : k1;
            k2 = gl == (uint32_t)t ? d2 : k2;
            __builtin_amdgcn_sched_barrier(0);
        }
        // (8) three periods (MIXED: four) with nothing planned for any group of the wave: everything planned has run and its dots have
        // left at (1).  (Tested HERE, after the steps, so that the loop has one way round: with a way past the steps the
        // counted waits of the next pass are computed for that way -- hardly any loads in flight -- and drain the rows.)
        if (__all(idle >= IDLE_END)) break;
    }
    if (oob) flags[FLAG_ROW_OOB] = 1;
    if constexpr (MIXED) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) n_marked += __shfl_down(n_marked, d, 64);
        if ((threadIdx.x & 63u) == 0u && n_marked && stats) atomicAdd(&stats[STAT_MARKED_ROWS], (unsigned long long)n_marked);
    }
}

template <int G, int EPL, typename T, int P, bool LIST>
__global__ __launch_bounds__(256) void k_dots2(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                               uint32_t n_occ, const uint32_t* __restrict__ dot_off,
                                               const T* emb, uint32_t emb_rows, uint32_t dp, double* dots, uint32_t* __restrict__ flags,
                                               const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count) {
    dots2_body<G, EPL, T, P, LIST, false>(desc, pool, n_occ, dot_off, emb, emb_rows, dp, dots, flags, list, list_count, nullptr, nullptr, nullptr);
}
template <int G, int EPL, int P>
__global__ __launch_bounds__(256) void k_dots2_mixed(const uint4* __restrict__ desc, const uint32_t* __restrict__ pool,
                                                     uint32_t n_occ, const uint32_t* __restrict__ dot_off,
                                                     const double* emb, uint32_t emb_rows, uint32_t dp, double* dots,
                                                     uint32_t* __restrict__ flags, const uint32_t* __restrict__ list,
                                                     const uint32_t* __restrict__ list_count, const float* emb32,
                                                     const uint32_t* __restrict__ bitmap, unsigned long long* __restrict__ stats) {
    dots2_body<G, EPL, double, P, true, true>(desc, pool, n_occ, dot_off, emb, emb_rows, dp, dots, flags, list, list_count, emb32, bitmap, stats);
}
