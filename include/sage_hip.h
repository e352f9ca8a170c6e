/*
 * sage_hip.h -- C-ABI of libsage_hip.so: the MI355X (gfx950) implementation of SaGe 2.0's per-iteration
 * vocabulary-ablation hot path.
 *
 * The reference has NO native boundary for this path: it is a plain Python call,
 *     run_sage_parallel(embeddings, partial_corpus, sage_model, workers_number)
 *         -> (total_tokens, total_triples, sage_losses, ablated_sizes)
 * (/root/reference/src/sage_tokenizer/utils.py:141-190, sole caller SaGeVocabBuilder.py:100-103).  The
 * entry points below are what a ctypes binding for that call binds; the binding itself
 * is sage_amd/_lib.py and INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: every function returns 0 on success and a negative SAGE_E_* code on failure;
 * sage_last_error(ctx) then holds a message.  All pointers are HOST pointers to caller-owned memory that is
 * only borrowed for the duration of the call, unless the name says `_device`.  One context belongs to one
 * GPU and one host thread at a time.  Nothing here takes or returns torch types.
 */
#ifndef SAGE_HIP_H
#define SAGE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sage_ctx sage_ctx;

enum {
    SAGE_OK = 0,
    SAGE_E_ARG = -1,         /* bad argument (null pointer, unsorted offsets, pad != 2, ...) */
    SAGE_E_MISSING_BYTE = -2,/* a single-byte key is absent: model.py:251-255 raises Exception("missing byte") */
    SAGE_E_HIP = -3,         /* a HIP runtime call failed */
    SAGE_E_STATE = -4,       /* vocab / corpus / embeddings not set yet */
    SAGE_E_LIMIT = -5,       /* a documented limit was exceeded (max_len above 32, id >= 2^20-1, a batch of 4 GiB, ...) */
    SAGE_E_ROW = -6,         /* a token id indexes past the embedding matrix (numpy IndexError in utils.py:133) */
    SAGE_E_EMPTY_LINE = -7   /* an empty line: model.py:148 cannot unpack zip(*[]) */
};

/* Library version, "major.minor.patch". */
const char* sage_version(void);

/* Creates a context bound to HIP device `device` (one context per GPU / per process rank). */
int sage_create(int device, sage_ctx** out_ctx);
void sage_destroy(sage_ctx* ctx);
const char* sage_last_error(const sage_ctx* ctx);

/*
 * Replaces SaGeTokenizer.set_vocabulary / set_bytes_vocab / verify_all_single_byte_exist_in_vocab
 * (model.py:28-51, 251-255).  Key i is key_bytes[key_offsets[i] .. key_offsets[i+1]) and gets id key_ids[i]
 * (key_ids == NULL: id = i, the reference's "position in the list").  Keys must be distinct (the Python
 * wrapper resolves repeats the way the dict does).  All 256 single bytes must be present.  A key longer than
 * max_len keeps its id but can never match (model.py:90), as in the reference, which only warns about it
 * (SaGeVocabBuilder.py:52-54).  Limits: ids < 2^20 - 1, max_len 1..32.
 */
int sage_set_vocab(sage_ctx* ctx, const uint8_t* key_bytes, const int64_t* key_offsets, const int32_t* key_ids,
                   int64_t n_keys, int32_t max_len);

/*
 * The same for a vocabulary that only LOST keys since the last sage_set_vocab / sage_set_vocab_pruned -- what
 * SaGeVocabBuilder does after every prune (SaGeVocabBuilder.py:159-192: `set_vocabulary(list(target_vocab.keys()))`, ids
 * re-compacted).  old_to_new[old id] = the key's new id, or -1 if it was pruned, for every id of the previous id space
 * (n_old = its size).  The new tables are built as by sage_set_vocab; in addition, when the match table of the whole
 * resident corpus is still valid (a step over all lines has run since the corpus and the previous vocabulary were set),
 * it is updated in place instead of recomputed: every key matching at a text position is a prefix of the longest one
 * there, so each match moves to its nearest surviving ancestor-or-self -- one streamed pass over the table, two small
 * table lookups per corpus byte, no hashing and no probes.  The steps that follow then launch no matcher.  Results are
 * identical to sage_set_vocab with the same keys (tests: tokenization and every integer of the step, prune after prune).
 * max_len must not change.  Falls back to plain sage_set_vocab behaviour when no valid match table is resident.
 */
int sage_set_vocab_pruned(sage_ctx* ctx, const uint8_t* key_bytes, const int64_t* key_offsets, const int32_t* key_ids,
                          int64_t n_keys, int32_t max_len, const int32_t* old_to_new, int64_t n_old);

/*
 * Uploads the partial corpus once per run: line i is bytes[line_offsets[i] .. line_offsets[i+1]) exactly as
 * `bytes(d, 'utf-8')` of utils.py:216 (trailing "\n" included).  Replaces the List[str] pickled to every
 * worker (utils.py:156).
 */
int sage_set_corpus(sage_ctx* ctx, const uint8_t* bytes, const int64_t* line_offsets, int64_t n_lines);

/*
 * Uploads the [rows, dim] float64 embedding matrix of embeddings.py:77-89, row = token id.  Called once per
 * embeddings_schedule step (SaGeVocabBuilder.py:89-91); `rows` may exceed the current vocabulary size: after a
 * prune the re-compacted ids keep indexing the rows of the old matrix, which is the reference's behaviour.
 */
int sage_set_embeddings(sage_ctx* ctx, const double* emb, int64_t rows, int32_t dim);

/*
 * Replaces SaGeTokenizer.tokenize (model.py:81-101) for lines [line_begin, line_end).
 * out_counts[i] = number of tokens of line line_begin+i.  If out_ids/out_starts/out_widths are non-NULL they
 * receive the (id, start, width) triples of all lines back to back (capacity `capacity` triples; the call
 * fails with SAGE_E_ARG if that is too small).  *out_total = total number of tokens.
 */
int sage_tokenize(sage_ctx* ctx, int64_t line_begin, int64_t line_end, int32_t* out_ids, int32_t* out_starts,
                  int32_t* out_widths, int64_t capacity, int64_t* out_counts, int64_t* out_total);

/*
 * Allocation-free tokenization.  NO COUNTERPART IN THE REFERENCE: a variant of sage_tokenize for callers that cannot
 * allocate device memory at the point of the call -- tight serving loops that tokenize request batches with the
 * resident vocabulary, harnesses that forbid hipMalloc, code that runs between the captures of a HIP graph (the call
 * itself synchronises the stream and copies two small results to the host, so it cannot be CAPTURED).  It allocates and frees no
 * device memory: its scratch (text copy with its zero pad, line remainders, the match table, per-line counters)
 * lives in a static device array of ~6.6 MiB that every GPU the library runs on holds until the process ends.
 * Inputs and outputs are the CALLER's DEVICE buffers: d_text[n_bytes] holds the lines back to back, d_line_offsets
 * [n_lines + 1] their byte offsets (int64, from 0 to n_bytes); d_ids/d_starts/d_widths[capacity] (all three or none)
 * and d_counts[n_lines] (optional) receive what sage_tokenize returns; *out_total (host) = number of tokens.  The
 * vocabulary is the one sage_set_vocab uploaded; no corpus needs to be resident and a resident one is left alone.
 * Limits: n_bytes <= SAGE_NOALLOC_MAX_BYTES and n_lines <= SAGE_NOALLOC_MAX_LINES per call (SAGE_E_LIMIT beyond).
 * All calls on one GPU share that GPU's scratch array: calls that overlap in time on one GPU (two contexts, two
 * threads) race -- serialise them.  The call returns after the context's stream has been synchronised.
 */
#define SAGE_NOALLOC_MAX_BYTES (1 << 20)
#define SAGE_NOALLOC_MAX_LINES (1 << 16)
int sage_tokenize_noalloc(sage_ctx* ctx, const uint8_t* d_text, int64_t n_bytes, const int64_t* d_line_offsets,
                          int64_t n_lines, int32_t* d_ids, int32_t* d_starts, int32_t* d_widths, int64_t capacity,
                          int64_t* d_counts, int64_t* out_total);

/*
 * One ablation step: replaces run_sage_parallel -> sage_per_chunk -> fast_sage/do_triples/compute_losses
 * (utils.py:141-257, model.py:115-248) over the lines [segment_line_bounds[0], segment_line_bounds[n_segments]).
 * segment_line_bounds has n_segments+1 increasing line indices; each [b[s], b[s+1]) is one range over which
 * the reference keeps one `triples` dict (a worker shard of utils.py:111-113 cut at the flush rule of
 * utils.py:220), i.e. the scope of the distinct-key count that sums to total_triples.
 *   out_loss[vsize]   f64: already negated like utils.py:255; 0 for ids that never occur
 *   out_occ[vsize]    i64: number of ablated occurrences of each id (> 0 <=> the id is a key of sage_losses)
 *   out_hist[hist_len] i64: ablated_sizes histogram, out_hist[k] = count of ablations into k tokens
 *                       (k >= hist_len is accumulated into out_hist[hist_len-1])
 * vsize must be > the largest id.  pad must be 2 (model.py:138; never overridden at utils.py:216).
 */
int sage_ablation_step(sage_ctx* ctx, const int64_t* segment_line_bounds, int64_t n_segments, int32_t pad,
                       double* out_loss, int64_t* out_occ, int64_t vsize, int64_t* out_total_tokens,
                       int64_t* out_total_triples, int64_t* out_hist, int64_t hist_len);

/*
 * Same step, results left in caller-provided DEVICE buffers (e.g. torch tensors that go straight into an
 * RCCL all-reduce): d_loss f64[vsize], d_occ i64[vsize], d_hist i64[hist_len], d_totals i64[2] =
 * {total_tokens, total_triples}.  Buffers are overwritten, not accumulated.  The call returns after the
 * context's stream has been synchronised.
 */
int sage_ablation_step_device(sage_ctx* ctx, const int64_t* segment_line_bounds, int64_t n_segments, int32_t pad,
                              double* d_loss, int64_t* d_occ, int64_t vsize, int64_t* d_totals,
                              int64_t* d_hist, int64_t hist_len);

/*
 * Measurement hooks (no counterpart in the reference).  With profiling on, every kernel launch of the next
 * steps is bracketed by HIP events on the context's stream; sage_get_kernel_times returns, per kernel class,
 * the number of launches and the summed milliseconds since the last sage_reset_kernel_times.
 */
enum { SAGE_K_MATCH = 0, SAGE_K_WALK = 1, SAGE_K_SCAN = 2, SAGE_K_COUNT = 3, SAGE_K_SCORE = 4, SAGE_K_MISC = 5,
       SAGE_K_DESCRIBE = 6, SAGE_K_PAIRS = 7, SAGE_K_LOGSIG = 8, SAGE_K_FIX = 9 /* float64 recomputation passes of the
       float32 row storage */, SAGE_K_PARTITION = 10 /* key scatter of the distinct count; SAGE_K_COUNT = its dedupe */,
       SAGE_K_NCLASSES = 11 };
int sage_set_profiling(sage_ctx* ctx, int enabled);
int sage_reset_kernel_times(sage_ctx* ctx);
int sage_get_kernel_times(sage_ctx* ctx, double* out_ms /*[SAGE_K_NCLASSES]*/, int64_t* out_launches /*[SAGE_K_NCLASSES]*/);

/* Tunables: scratch is sized for batches of at most `bytes` corpus bytes (default 512 MiB; a batch whose
   position pairs would overflow the 32-bit offsets makes the step halve it and start over). */
int sage_set_batch_bytes(sage_ctx* ctx, int64_t bytes);

/* Matcher variant: 0 = persistent blocks with the 128 KiB presence filter in LDS, candidates compacted through an LDS
   queue and probed in the embedded-key table (default); 1 = plain (every length probed in the slot table + key planes);
   2 = the LDS filter with column-by-column probes of the slot table + key planes.  Identical results; 1 and 2 are the
   cross-checks of the tests (different tables, different probe code). */
int sage_set_match_mode(sage_ctx* ctx, int mode);

/* Forgets the resident match table: the next step (or sage_tokenize) runs the matcher over the lines it covers, as the
 * first step after sage_set_vocab / sage_set_corpus does.  A table completed by a step over all lines otherwise serves
 * every later call until the vocabulary or the corpus changes.  bench.py calls this before every timed step, so that its
 * `value` is the COLD step of BASELINE.json configs[1] (matcher included); the carried-over case is reported separately. */
int sage_invalidate_matches(sage_ctx* ctx);

/* Distinct-key counting back end: 0 = hash partition + LDS dedupe (default), 1 = per-segment sets in HBM (the
 * fallback the library takes by itself if an LDS set ever overflows).  Results are identical. */
int sage_set_count_mode(sage_ctx* ctx, int mode);

/* Storage of the embedding rows on the scoring path (compute_losses, utils.py:125-138; arithmetic is float64 in
 * both modes).  0 (default): rows whose values are all float32-representable -- every row train_embeddings fills
 * from word2vec (embeddings.py:81) -- are gathered from a float32 copy (half the bytes; products of two float32
 * are exact in float64), and every pair that touches another row (the float64 uniform rows of embeddings.py:86) is
 * recomputed from the float64 matrix; if more than half of the rows are not float32-representable, or a step finds
 * that more than 30 % of the occurrences touch such a row, the float64 matrix is used for everything.  1: float64
 * storage only.  2: the float32 copy whenever at least one row is float32-representable, whatever the share (a test
 * switch: it forces the mixed path).  Takes effect at the next sage_set_embeddings, which must follow. */
int sage_set_score_mode(sage_ctx* ctx, int mode);

/* Reproducible float sums (no counterpart in the reference, whose own merge order is the workers' completion order,
 * utils.py:159-183).  By default the per-occurrence loss terms are added with float64 atomics: integers are exact, the
 * losses of two runs agree to ~1e-12 relative.  With enabled != 0 every log-sigmoid term is converted to 128-bit fixed
 * point (units of 2^-60) before anything is added and all sums are integer sums: out_loss is then bit-identical from run
 * to run and for any batch size (sage_set_batch_bytes), at ~1 ms per step on the 1M-line workload.  It differs from the
 * default by the truncation of each term to 2^-60 (<< 1e-5 relative on the per-token loss). */
int sage_set_deterministic(sage_ctx* ctx, int enabled);

/* word2vec retraining on the GPU.  Replaces the host call
 *   gensim.models.Word2Vec(corpus_file=..., vector_size=D, negative=N, alpha=ALPHA, window=window_size,
 *                          min_count=min_count, sg=1, workers=workers)        (embeddings.py:68-75)
 * and the tokenization loop that feeds it (embeddings.py:61-65): skip-gram with negative sampling over the resident
 * corpus tokenized with the resident vocabulary, gensim 4.3.2's published algorithm (subsampling `sample`, reduced
 * windows, unigram^0.75 negatives, linear learning-rate decay over `epochs`; gensim's defaults are min_alpha=1e-4,
 * epochs=5, sample=1e-3).  out_vectors[id*dim + d] receives the float32 input vectors widened to float64 (what
 * model.wv[token] holds) and out_counts[id] the token count of id in the tokenized corpus; ids whose count is below
 * min_count are not trained and the caller gives them the uniform row of embeddings.py:86.  Lock-free training, like
 * gensim's with workers > 1: results are not reproducible bit for bit (parity unpinned). */
int sage_train_embeddings(sage_ctx* ctx, int32_t dim, int32_t window, int32_t negative, double alpha, double min_alpha,
                          int32_t epochs, double sample, int32_t min_count, uint64_t seed, int64_t vsize,
                          double* out_vectors /*[vsize*dim]*/, int64_t* out_counts /*[vsize]*/);

/* Workload statistics of the last step, for the roofline's algorithmic bytes:
 * out[0]=tokens, out[1]=corpus bytes, out[2]=ablated occurrences, out[3]=embedding rows touched (ablated windows +
 * base spans), out[4]=signed pair records, out[5]=unordered position pairs (= records / 2: what the distinct count
 * partitions), out[6]=bytes per stored embedding element on the main scoring pass (4 or 8), out[7]=occurrences
 * recomputed from the float64 matrix, out[8]=ids of the ablated windows (the rows the window-dot kernel gathers),
 * out[9]=occurrences whose window is kept in the id pool, out[10]=embedding dimension, out[11]=padded row width in
 * elements, out[12]=position pairs of the ablated windows (dot products evaluated), out[13]=steps this context redid with the
 * per-segment HBM sets because the partitioned distinct count failed one of its own checks (bucket sizes against the
 * counting pass, records per tile, records read against position pairs; expected 0 -- the tests hold it there),
 * out[14]=1 if total_triples of the last step came from the HBM sets (count mode 1 or a fallback) else 0,
 * out[15]=bit 0: the key buckets were sized by a counting pass (heavily repeated keys); bit 1: the step launched no matcher
 * (its match table was carried over a prune by sage_set_vocab_pruned); out[16]=window ids of the occurrences scored by the
 * float64/mixed kernel (a subset of out[8]); out[17]=of those, rows gathered from the float64 matrix; out[18..23]=0. */
#define SAGE_STEP_STATS 24
int sage_get_step_stats(sage_ctx* ctx, int64_t* out /*[SAGE_STEP_STATS]*/);

/* Measurement hook for bench.py's roofline (no counterpart in the reference): distinct embedding rows per window of
 * `window` consecutive occurrences, over the LAST batch of the last step (its descriptors are still resident) -- the rows a
 * gather kernel that keeps `window` occurrences in flight must fetch, i.e. its compulsory traffic.  out[0], out[1] =
 * distinct rows and window ids of the occurrences on the exact list (all occurrences when the step had no float32/float64
 * split), out[2], out[3] = the same for the inexact list, out[4] = occurrences of that batch, out[5] = batches of the step. */
int sage_measure_unique_rows(sage_ctx* ctx, int32_t window, int64_t* out /*[6]*/);

/* Test hook (no counterpart in the reference): runs the step's own exclusive scan over n copies of `value` and returns
 * the 64-bit grand total and the last (32-bit, wrapping) offset.  The step decides from the 64-bit total whether a batch's
 * position pairs fit 32-bit offsets (else it halves the batch and starts over) before anything reads the offsets; the
 * tests feed it totals past 2^32. */
int sage_debug_scan_total(sage_ctx* ctx, uint32_t value, int64_t n, uint64_t* out_total, uint32_t* out_last_offset);

#ifdef __cplusplus
}
#endif
#endif /* SAGE_HIP_H */
