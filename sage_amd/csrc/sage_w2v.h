// word2vec retraining on the GPU: skip-gram with negative sampling over the token
// stream the matcher already produces, replacing the host call gensim.models.Word2Vec(corpus_file=..., vector_size=D,
// negative=N, alpha=ALPHA, window=window_size, min_count=min_count, sg=1) of embeddings.py:68-75.
//
// gensim is not vendored in the reference (pyproject pins gensim==4.3.2); what is restated here is its published
// algorithm (word2vec_inner.pyx, w2v_fast_sentence_sg_neg / train_batch_sg):
//   * words below min_count are dropped; frequent words are subsampled with
//       p_keep(w) = min(1, (sqrt(c_w / (sample * T)) + 1) * (sample * T) / c_w),   T = retained corpus words
//   * for every kept position i of a sentence: b ~ U{0..window-1}; every j in [i-window+b, i+window-b], j != i, trains
//     the pair (input = word j, predicted = word i)
//   * a pair: target 0 = the predicted word (label 1), then `negative` draws from the unigram^0.75 distribution
//     (label 0, a draw equal to the predicted word is skipped); f = syn0[input] . syn1neg[target]; |f| >= 6 is skipped;
//     g = (label - sigmoid(f)) * alpha; syn1neg[target] += g * syn0[input]; syn0[input] += sum_targets g * syn1neg[target]
//   * alpha falls linearly from alpha to min_alpha over epochs * corpus words; syn0 ~ U(-1,1)/D, syn1neg = 0
// Training is lock-free ("Hogwild"), as gensim's is with workers > 1, so results are not reproducible bit for bit on
// either side: PARITY UNPINNED.  Tests check the statistics of the result, not its values.
//
// Mapping: a group of 16 lanes owns one LINE (sentence) and walks it like a gensim worker; each lane holds Dp/16
// elements of a row; dots are reduced with DPP inside the 16-lane row.  Random numbers are stateless hashes of
// (seed, epoch, token index, purpose), identical in the 16 lanes of a group.
#pragma once
#include "sage_kernels2.h"

#define W2V_G 16            // lanes per line
#define W2V_BUF 512         // kept tokens of a line staged per pass (gensim chunks sentences at 10,000 words)
#define W2V_MAX_EXP 6.0f

__device__ __forceinline__ uint32_t w2v_rand(uint64_t seed, uint32_t epoch, uint64_t index, uint32_t purpose) {
    return (uint32_t)(sage_mix64(seed ^ (index * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)epoch << 56) ^ ((uint64_t)purpose << 40)) >> 32);
}

// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ float row_sum_f32(float v) {
    int x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));  // row_half_mirror
    x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// token counts of a batch (ids of tok_w0) into counts[]
__global__ __launch_bounds__(256) void k_token_hist(const uint32_t* __restrict__ tok_w0, uint32_t n_tok,
                                                    unsigned long long* __restrict__ counts) {
    __shared__ uint32_t cid[ACC_SLOTS];
    __shared__ uint32_t cn[ACC_SLOTS];
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256) { cid[i] = ACC_EMPTY; cn[i] = 0; }
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_tok; i += gridDim.x * 256u)
        acc_add_u32(cid, cn, counts, SAGE_T_ID(tok_w0[i]));
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256)
        if (cn[i]) atomicAdd(&counts[cid[i]], (unsigned long long)cn[i]);
}

struct W2VArgs {
    const uint32_t* tok_w0;      // token ids of the batch
    const uint32_t* tok_off;     // [n_lines + 1] token offset of every line of the batch
    uint32_t n_lines;
    uint64_t tokens_before;      // corpus tokens in earlier batches (learning-rate progress, random streams)
    uint64_t total_tokens;       // corpus tokens
    const uint32_t* keep_u32;    // [V] keep threshold: token kept iff rand32 <= keep_u32[id]; 0 with `dropped` = never
    const uint8_t* dropped;      // [V] 1 = below min_count
    const float* alias_prob;     // [V] alias table of the unigram^0.75 distribution
    const uint32_t* alias_idx;   // [V]
    uint32_t vocab;              // V (ids 0..V-1)
    float* syn0;                 // [V][dp]
    float* syn1;                 // [V][dp]
    uint32_t dp;
    uint32_t window, negative, epoch, epochs;
    float alpha, min_alpha;
    uint64_t seed;
};

template <int EPL>   // elements per lane = dp / 16
__global__ __launch_bounds__(256) void k_w2v_train(W2VArgs a) {
    __shared__ uint32_t kept[256 / W2V_G][W2V_BUF];
    const uint32_t grp = threadIdx.x / W2V_G, gl = threadIdx.x % W2V_G;
    const uint32_t n_groups = gridDim.x * (256u / W2V_G);
    uint32_t* buf = kept[grp];
    for (uint32_t line = blockIdx.x * (256u / W2V_G) + grp; line < a.n_lines; line += n_groups) {
        const uint32_t t0 = a.tok_off[line], t1 = a.tok_off[line + 1];
        uint32_t t = t0;
        while (t < t1) {
            // ---- stage up to W2V_BUF kept tokens of the line (min_count filter + subsampling), 16 tokens per round
            uint32_t n = 0;
            while (t < t1 && n + W2V_G <= W2V_BUF) {
                const uint32_t ti = t + gl;
                bool keep = false;
                uint32_t id = 0;
                if (ti < t1) {
                    id = SAGE_T_ID(a.tok_w0[ti]);
                    keep = !a.dropped[id] && w2v_rand(a.seed, a.epoch, a.tokens_before + ti, 1) <= a.keep_u32[id];
                }
                const unsigned long long bal = __ballot(keep);
                const uint32_t base_lane = (threadIdx.x & 63u) & ~(uint32_t)(W2V_G - 1);
                const uint32_t gmask = (uint32_t)(bal >> base_lane) & 0xFFFFu;
                if (keep) buf[n + __popc(gmask & ((1u << gl) - 1u))] = id;
                n += __popc(gmask);
                t += W2V_G;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint64_t pos_base = a.tokens_before + t0 + (uint64_t)(t - t0);   // any per-chunk distinct stream offset
            const double progress = ((double)a.epoch * (double)a.total_tokens + (double)(a.tokens_before + t0)) /
                                    ((double)a.epochs * (double)a.total_tokens);
            const float alpha = fmaxf(a.min_alpha, a.alpha - (a.alpha - a.min_alpha) * (float)progress);
            // ---- train the chunk
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t center = buf[i];
                const uint32_t b = w2v_rand(a.seed, a.epoch, pos_base + i, 2) % a.window;
                const uint32_t j0 = i + b >= a.window ? i + b - a.window : 0u;
                const uint32_t j1 = min(n, i + a.window + 1u - b);
                for (uint32_t j = j0; j < j1; ++j) {
                    if (j == i) continue;
                    const uint32_t w2 = buf[j];
                    float* xrow = a.syn0 + (size_t)w2 * a.dp + gl * EPL;
                    float x[EPL], e[EPL];
#pragma unroll
                    for (int q = 0; q < EPL; ++q) { x[q] = xrow[q]; e[q] = 0.0f; }
                    for (uint32_t d = 0; d <= a.negative; ++d) {
                        uint32_t target = center;
                        float label = 1.0f;
                        if (d > 0) {
                            const uint32_t r1 = w2v_rand(a.seed, a.epoch, (pos_base + i) * 64u + (j - j0), 16u + d);
                            const uint32_t r2 = w2v_rand(a.seed, a.epoch, (pos_base + i) * 64u + (j - j0), 48u + d);
                            const uint32_t slot = (uint32_t)(((uint64_t)r1 * a.vocab) >> 32);
                            target = ((float)(r2 >> 8) * (1.0f / 16777216.0f)) < a.alias_prob[slot] ? slot : a.alias_idx[slot];
                            label = 0.0f;
                            if (target == center) continue;
                        }
                        float* yrow = a.syn1 + (size_t)target * a.dp + gl * EPL;
                        float y[EPL];
                        float f = 0.0f;
#pragma unroll
                        for (int q = 0; q < EPL; ++q) { y[q] = yrow[q]; f = fmaf(x[q], y[q], f); }
                        f = row_sum_f32(f);
                        if (f >= W2V_MAX_EXP || f <= -W2V_MAX_EXP) continue;
                        const float g = (label - 1.0f / (1.0f + __expf(-f))) * alpha;
#pragma unroll
                        for (int q = 0; q < EPL; ++q) { e[q] = fmaf(g, y[q], e[q]); yrow[q] = fmaf(g, x[q], y[q]); }
                    }
#pragma unroll
                    for (int q = 0; q < EPL; ++q) xrow[q] = x[q] + e[q];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}
