// libsage_hip.so -- host side of the C-ABI declared in include/sage_hip.h.  gfx950 (MI355X) only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <utility>
#include <atomic>
#include <vector>

#include "../../include/sage_hip.h"
#include "sage_common.h"
#include "sage_kernels.h"
#include "sage_kernels2.h"
#include "sage_kernels3.h"
#include "sage_kernels4.h"
#include "sage_w2v.h"

#define SAGE_VERSION_STR "0.1.0"

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct EventPair { hipEvent_t a, b; int cls; };

}  // namespace

// ---- scratch of the allocation-free entry point (sage_tokenize_noalloc) ---------------------------------------------
// One file-scope device array per GPU the library runs on, held until the process ends.  Its NAME is a host-side
// stand-in, not a device address: host code only ever uses the per-device pointer hipGetSymbolAddress returns
// (scratch_for_device below), as kernel argument and for memsets and copies alike.
constexpr size_t NA_ALIGN = 256;
constexpr size_t na_up(size_t x) { return (x + NA_ALIGN - 1) / NA_ALIGN * NA_ALIGN; }
constexpr size_t NA_TEXT = 0;                                                          // text + 1 KiB zero pad
constexpr size_t NA_REM = NA_TEXT + na_up(SAGE_NOALLOC_MAX_BYTES + 1024);              // bytes to the line end
constexpr size_t NA_M = NA_REM + na_up(SAGE_NOALLOC_MAX_BYTES + 1024);                 // match table, 4 B per byte
constexpr size_t NA_NTOK = NA_M + na_up((size_t)SAGE_NOALLOC_MAX_BYTES * 4);
constexpr size_t NA_NOCC = NA_NTOK + na_up((size_t)SAGE_NOALLOC_MAX_LINES * 4);
constexpr size_t NA_TOKOFF = NA_NOCC + na_up((size_t)SAGE_NOALLOC_MAX_LINES * 4);
constexpr size_t NA_OCCOFF = NA_TOKOFF + na_up(((size_t)SAGE_NOALLOC_MAX_LINES + 1) * 4);
constexpr size_t NA_TILES = NA_OCCOFF + na_up(((size_t)SAGE_NOALLOC_MAX_LINES + 1) * 4);
constexpr size_t NA_TOTAL = NA_TILES + na_up(((size_t)SAGE_NOALLOC_MAX_LINES / SCAN_TILE + 2) * 4);
constexpr size_t NA_FLAGS = NA_TOTAL + na_up(16);
constexpr size_t NA_BYTES = NA_FLAGS + na_up(FLAG_COUNT * 4);                           // ~6.6 MiB
static __device__ __align__(256) unsigned char scratch[NA_BYTES];

struct sage_ctx {
    int device = 0;
    unsigned char* na_scratch = nullptr;   // this device's copy of `scratch` (hipGetSymbolAddress at sage_create)
    int n_cu = 256;
    hipStream_t stream = nullptr;
    std::string err;

    // vocabulary
    DevBuf buckets, keys, byte_id, direct2, bloom, pmask, pmask8, ebk, parent;
    DevBuf parent_old, old_to_new, remap;   // sage_set_vocab_pruned: the previous vocabulary's parents, its id map, the match remap
    // The match table of the WHOLE resident corpus (4 B per byte), kept across steps when it fits m_resident_max: a
    // vocabulary that only lost keys (what SaGe does between steps) then needs no matching at all -- every match moves
    // to its nearest surviving prefix, a table lookup per corpus byte (sage_set_vocab_pruned).
    bool m_whole = false;      // M covers the whole corpus (indexed by absolute position) rather than one batch
    bool m_valid = false;      // ... and holds the matches of the current vocabulary for every position
    int64_t m_resident_max = 96ll << 30;
    int64_t steps_without_matcher = 0;   // steps that found M valid and launched no matcher
    DevBuf vk_bytes, vk_off, vk_ids;   // packed keys as uploaded (input of k_vocab_build)
    uint32_t enb = 0;
    int match_mode = 0;  // 0 = LDS-filtered persistent matcher with queued probes, 1 = plain matcher, 2 = LDS-filtered with column probes
    uint32_t bucket_mask = 0, key_plane = 0;
    int max_len = 0;
    int64_t max_id = -1;
    bool have_vocab = false;

    // corpus
    DevBuf corpus, rem, line_off;
    std::vector<int64_t> h_line_off;
    int64_t max_line_bytes = 0;
    int64_t n_lines = 0, nbytes = 0;
    bool have_corpus = false;

    // embeddings
    DevBuf emb;           // float64 rows (always kept: the exact values)
    DevBuf emb32;         // float32 rows, when most rows are float32-representable (word2vec output is)
    DevBuf lline_list, lline_count;  // lines of more than SAGE_LONG_LINE bytes: walked a wave each (k_long_lines, once per corpus)
    uint32_t n_lline = 0;            // how many (0 also with SAGE_LONG_WALK=0)
    DevBuf emb_stage, emb_cnt;   // sage_set_embeddings: the caller's contiguous matrix as uploaded; the count of inexact rows
    DevBuf inexact;       // bitmap of the rows that are not: their pairs are recomputed from `emb`
    DevBuf inex_list, inex_count;
    DevBuf w2v_counts, w2v_keep, w2v_dropped, w2v_aprob, w2v_aidx, w2v_syn0, w2v_syn1;   // sage_train_embeddings
    DevBuf long_list, long_count;   // occurrences whose window does not fit the inline descriptor (k_describe<true>)
    int64_t emb_rows = 0, n_inexact = 0;
    int dim = 0, dp = 0;
    bool have_emb = false;
    double f64_switch = -1.0;   // share of occurrences on the float64 pass above which later steps use the float64 matrix alone (SAGE_F64_SWITCH; < 0: by kernel)
    bool long_walk = true;           // lines above SAGE_LONG_LINE bytes walked a wave each (SAGE_LONG_WALK=0: off)
    uint32_t match_threads = 1024;   // experiment: SAGE_MATCH_THREADS
    bool dots2 = true;     // k_dots2 (sage_kernels4.h) for the score kernel; SAGE_DOTS2=0 selects k_dots
    bool use_f32 = false;
    int score_mode = 0;   // 0 = float32 storage when lossless for most rows, 1 = always float64 storage

    // scratch (grow-only; nothing is allocated in the steady state)
    DevBuf M, ntok, tokoff, tile_sums, grand_total, tok_w0, occ_pos, set, seg_line_rel, seg_tok_off, set_base, set_mask, seg_pairs, desc, gcache, bucket_off, rec_a,
        nocc, occoff, occ_tok, occ_a, exact_list, exact_count, np_ab, dot_off, dots, pool, pool_cursor,
        np_all, pair_off, seg_first_occ, seg_tile_off, seg_bucket_base, seg_nbk_mask, cursor, giant_list, giant_count, tile_tab;
    int pool_factor = 8;  // id-pool words per occurrence (grown on overflow)
    bool exact_buckets = false;   // v2 count: size the key buckets with a counting pass instead of the estimate
    int count_mode = 0;  // 0 = partition + LDS dedupe (v2), 1 = per-segment HBM sets (v1)
    bool count_fallback = false;   // an LDS set overflowed: v1 until the vocabulary or the corpus changes
    int64_t count_redos = 0;       // steps redone with the HBM sets because a check of the partitioned count failed (never expected)
    bool debug_part = false;       // SAGE_DEBUG_PART=1: the scatter pass is compared tile by tile and bucket by bucket with a counting pass
    DevBuf tile_dbg;
    DevBuf ex_ids, ex_starts, ex_widths;
    DevBuf loss, occ, hist, stats, flags;
    DevBuf loss_fix;              // deterministic mode: 128-bit fixed-point loss sums, two words per id
    // host tables on their way to the device: an asynchronous copy from pageable memory may still be reading its source
    // when the call returns, so the source lives here until the step's next stream synchronisation (a std::vector
    // local to the batch loop is unmapped at the end of its block -- seen as garbage bucket offsets)
    std::vector<std::vector<uint32_t>> keep32;
    std::vector<std::vector<uint64_t>> keep64;
    bool deterministic = false;
    uint64_t pair_limit = 0xFFFFFF00ull;   // 32-bit pair / dot offsets per batch (SAGE_TEST_PAIR_LIMIT lowers it: test hook)
    int64_t batch_bytes = 512ll << 20;   // fewer, larger batches: the line walks last as long as a batch's longest line (4.9 -> 3.9 ms from two batches to one)
    int set_factor = 16;
    uint32_t last_n_occ = 0;       // the last batch of the last step: what sage_measure_unique_rows looks at
    bool last_split = false;
    int last_batches = 0;

    // measurement
    bool profiling = false;
    std::vector<EventPair> ev_used;
    std::vector<hipEvent_t> ev_free;
    double k_ms[SAGE_K_NCLASSES] = {0};
    int64_t k_launches[SAGE_K_NCLASSES] = {0};
    int64_t step_stats[SAGE_STEP_STATS] = {0};
};

namespace {

int fail(sage_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

// Device address of this GPU's `scratch` array; looked up once per device, with that device current.
constexpr int NA_MAX_DEVICES = 64;
std::mutex g_na_mutex;
unsigned char* g_na_ptr[NA_MAX_DEVICES] = {nullptr};
hipError_t scratch_for_device(int device, unsigned char** out) {
    if (device < 0 || device >= NA_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_na_mutex);
    if (!g_na_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));   // resolves for the CURRENT device
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_na_ptr[device] = static_cast<unsigned char*>(p);
    }
    *out = g_na_ptr[device];
    return hipSuccess;
}

constexpr size_t PART_LDS_BYTES = (size_t)PART_STAGE * 8 + ((size_t)PART_NBK_MAX * 3 + PART_LONGS + 24) * 4;
constexpr size_t SET3_LDS_BYTES = (size_t)SET3_SLOTS * 8;

#define HIP_TRY(c, expr)                                                                                   \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail((c), SAGE_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

int ensure(sage_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return SAGE_OK;
    if (b.p) HIP_TRY(c, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;  // a little headroom so near-equal batches do not reallocate
    HIP_TRY(c, hipMalloc(&b.p, want));
    b.cap = want;
    return SAGE_OK;
}

void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
}

struct Prof {
    sage_ctx* c; int cls; hipEvent_t a = nullptr, b = nullptr;
    Prof(sage_ctx* c_, int cls_) : c(c_), cls(cls_) {
        if (!c->profiling) return;
        auto take = [&]() {
            hipEvent_t e;
            if (!c->ev_free.empty()) { e = c->ev_free.back(); c->ev_free.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        a = take(); b = take();
        (void)hipEventRecord(a, c->stream);
    }
    ~Prof() {
        if (!c->profiling) return;
        (void)hipEventRecord(b, c->stream);
        c->ev_used.push_back({a, b, cls});
    }
};

void collect_events(sage_ctx* c) {
    for (auto& e : c->ev_used) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->k_ms[e.cls] += ms;
            c->k_launches[e.cls] += 1;
        }
        c->ev_free.push_back(e.a);
        c->ev_free.push_back(e.b);
    }
    c->ev_used.clear();
}

uint64_t pow2ceil(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// exclusive scan of n u32 values: out has n+1 entries, *total (device u64) gets the grand total
inline uint32_t scan_tiles(uint32_t n) { return std::max(1u, cdiv(n, SCAN_TILE)); }

// the three scan kernels with the tile-sum buffer given (scan_tiles(n) words)
void scan_u32_with(sage_ctx* c, const uint32_t* in, uint32_t n, uint32_t* out, uint64_t* d_total, uint32_t* tile_sums) {
    const uint32_t n_tiles = scan_tiles(n);
    Prof p(c, SAGE_K_SCAN);
    k_scan_tile_sums<<<n_tiles, 256, 0, c->stream>>>(in, n, tile_sums);
    k_scan_top<<<1, 256, 0, c->stream>>>(tile_sums, n_tiles, d_total);
    k_scan_apply<<<n_tiles, 256, 0, c->stream>>>(in, n, tile_sums, out);
}

int scan_u32(sage_ctx* c, const uint32_t* in, uint32_t n, uint32_t* out, uint64_t* d_total) {
    int rc;
    if ((rc = ensure(c, c->tile_sums, (size_t)scan_tiles(n) * 4))) return rc;
    scan_u32_with(c, in, n, out, d_total, c->tile_sums.as<uint32_t>());
    return SAGE_OK;
}

// Buffers of the matcher + base walk: the context's grow-only allocations (match_and_count) or pieces of the static
// scratch array (sage_tokenize_noalloc).
struct MatchBufs {
    const uint8_t* corpus;      // text, readable up to 1 KiB past its last byte
    const uint8_t* rem;         // bytes to the line end per text byte
    const int64_t* line_off;    // device line offsets, indexed from line 0 of `corpus`
    uint32_t* M;                // [nb]
    uint32_t *ntok, *nocc;      // [nl]
    uint32_t *tokoff, *occoff;  // [nl + 1]
    uint32_t* tile_sums;        // [scan_tiles(nl)]
    uint64_t* grand_total;      // [2]
    uint32_t* flags;            // [FLAG_COUNT]
    const uint32_t* lline_list; // lines walked a wave each (k_long_lines), global line indices
    uint32_t n_lline;           // 0: none, every line per thread
};

// K1 + K2a(count) + scans over the nl lines from line0 (nb text bytes from pos0): fills M, ntok/tokoff and nocc/occoff;
// grand_total[0] = tokens, grand_total[1] = ablated occurrences (device).  Launches only: nothing is allocated.
void launch_match_and_count(sage_ctx* c, const MatchBufs& B, int64_t line0, uint32_t nl, int64_t pos0, uint64_t nb) {
    VocabDev v{c->buckets.as<uint4>(), c->bucket_mask, c->keys.as<uint64_t>(), c->key_plane, c->byte_id.as<uint32_t>(),
               c->direct2.as<uint32_t>(), c->bloom.as<uint32_t>(), c->pmask.p, c->pmask8.p, c->max_len,
               c->ebk.as<uint4>(), c->enb, c->parent.as<uint32_t>()};
    if (nb) {
        Prof p(c, SAGE_K_MATCH);
        const uint32_t grid = cdiv(nb, 256);
        const int nw = (c->max_len + 7) / 8;
        if (c->match_mode != 1) {
            const uint32_t pgrid = std::min<uint32_t>(cdiv(nb, 1024), (uint32_t)c->n_cu);
#define SAGE_MATCH_LDS(NW_, ML_) do { \
                if (c->match_mode == 0) k_match_lds<NW_, ML_, true><<<pgrid, c->match_threads, 0, c->stream>>>(B.corpus, B.rem, pos0, (uint32_t)nb, v, B.M); \
                else k_match_lds<NW_, ML_, false><<<pgrid, 1024, 0, c->stream>>>(B.corpus, B.rem, pos0, (uint32_t)nb, v, B.M); } while (0)
            const int ml = c->max_len;      // the smallest instantiation that covers max_len
            if (ml <= 8) SAGE_MATCH_LDS(1, 8);
            else if (ml <= 12) SAGE_MATCH_LDS(2, 12);
            else if (ml <= 16) SAGE_MATCH_LDS(2, 16);
            else if (ml <= 17) SAGE_MATCH_LDS(3, 17);
            else if (ml <= 20) SAGE_MATCH_LDS(3, 20);
            else if (ml <= 24) SAGE_MATCH_LDS(3, 24);
            else SAGE_MATCH_LDS(4, 32);
#undef SAGE_MATCH_LDS
        } else switch (nw) {
        case 1: k_match<1><<<grid, 256, 0, c->stream>>>(B.corpus, B.rem, pos0, (uint32_t)nb, v, B.M); break;
        case 2: k_match<2><<<grid, 256, 0, c->stream>>>(B.corpus, B.rem, pos0, (uint32_t)nb, v, B.M); break;
        case 3: k_match<3><<<grid, 256, 0, c->stream>>>(B.corpus, B.rem, pos0, (uint32_t)nb, v, B.M); break;
        default: k_match<4><<<grid, 256, 0, c->stream>>>(B.corpus, B.rem, pos0, (uint32_t)nb, v, B.M); break;
        }
    }
    {
        Prof p(c, SAGE_K_WALK);
        k_walk_count<<<cdiv(nl, 256) + long_blocks(B.n_lline), 256, 0, c->stream>>>(B.line_off, line0, nl, pos0, B.M, B.ntok, B.nocc, B.flags,
                                                                                     B.lline_list, B.n_lline);
    }
    scan_u32_with(c, B.ntok, nl, B.tokoff, B.grand_total, B.tile_sums);
    scan_u32_with(c, B.nocc, nl, B.occoff, B.grand_total + 1, B.tile_sums);
}

// M of the batch that starts at corpus position pos0
inline uint32_t* batch_M(const sage_ctx* c, int64_t pos0) { return c->M.as<uint32_t>() + (c->m_whole ? pos0 : 0); }

// The same over lines [line0, line1) of the resident corpus, into the context's buffers.  out_total (optional) syncs.
int match_and_count(sage_ctx* c, int64_t line0, int64_t line1, uint64_t* out_total) {
    const int64_t pos0 = c->h_line_off[line0];
    const uint64_t nb = (uint64_t)(c->h_line_off[line1] - pos0);
    const uint32_t nl = (uint32_t)(line1 - line0);
    if (nb >= 0xFFFFFF00ull) return fail(c, SAGE_E_LIMIT, "a batch must stay under 4 GiB of corpus bytes");
    int rc;
    if ((rc = ensure(c, c->M, std::max<uint64_t>(c->m_whole ? (uint64_t)c->nbytes : nb, 1) * sizeof(uint32_t) + 16))) return rc;
    if ((rc = ensure(c, c->ntok, (size_t)nl * 4))) return rc;
    if ((rc = ensure(c, c->nocc, (size_t)nl * 4))) return rc;
    if ((rc = ensure(c, c->tokoff, ((size_t)nl + 1) * 4))) return rc;
    if ((rc = ensure(c, c->occoff, ((size_t)nl + 1) * 4))) return rc;
    if ((rc = ensure(c, c->grand_total, 16))) return rc;
    if ((rc = ensure(c, c->tile_sums, (size_t)scan_tiles(nl) * 4))) return rc;
    const MatchBufs B{c->corpus.as<uint8_t>(), c->rem.as<uint8_t>(), c->line_off.as<int64_t>(), batch_M(c, pos0),
                      c->ntok.as<uint32_t>(), c->nocc.as<uint32_t>(), c->tokoff.as<uint32_t>(), c->occoff.as<uint32_t>(),
                      c->tile_sums.as<uint32_t>(), c->grand_total.as<uint64_t>(), c->flags.as<uint32_t>(),
                      c->lline_list.as<uint32_t>(), c->n_lline};
    launch_match_and_count(c, B, line0, nl, pos0, c->m_valid ? 0 : nb);      // a valid M needs no matcher, only the walks
    HIP_TRY(c, hipGetLastError());
    if (out_total) {
        uint64_t tot = 0;
        HIP_TRY(c, hipMemcpyAsync(&tot, c->grand_total.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (tot >= 0xFFFFFF00ull) return fail(c, SAGE_E_LIMIT, "a batch must stay under 2^32 tokens");
        *out_total = tot;
    }
    return SAGE_OK;
}

int check_ready(sage_ctx* c, bool need_emb) {
    if (!c) return SAGE_E_ARG;
    if (!c->have_vocab) return fail(c, SAGE_E_STATE, "sage_set_vocab has not been called");
    if (!c->have_corpus) return fail(c, SAGE_E_STATE, "sage_set_corpus has not been called");
    if (need_emb && !c->have_emb) return fail(c, SAGE_E_STATE, "sage_set_embeddings has not been called");
    return SAGE_OK;
}

// padded row width the score kernels are instantiated for
int padded_dim(int dim) {
    const int opts[] = {32, 64, 128, 256, 384, 512};
    for (int o : opts) if (dim <= o) return o;
    return 0;
}

// G x EPL = lanes per row x elements per lane
// asynchronous host-to-device copy of a small table whose source outlives the call (see sage_ctx::keep32)
static hipError_t h2d_keep(sage_ctx* c, void* dst, const std::vector<uint32_t>& v, size_t n) {
    c->keep32.push_back(v);
    return hipMemcpyAsync(dst, c->keep32.back().data(), n * 4, hipMemcpyHostToDevice, c->stream);
}
static hipError_t h2d_keep(sage_ctx* c, void* dst, const std::vector<uint64_t>& v, size_t n) {
    c->keep64.push_back(v);
    return hipMemcpyAsync(dst, c->keep64.back().data(), n * 8, hipMemcpyHostToDevice, c->stream);
}

template <int G, int EPL, typename T>
void launch_pairs(sage_ctx* c, const T* emb, const uint32_t* tok_w0, uint32_t n_tok, bool fix) {
    Prof p(c, fix ? SAGE_K_FIX : SAGE_K_PAIRS);
    const uint32_t grid = std::min<uint32_t>(cdiv(cdiv(n_tok, 4), 256 / G), 256u * 16u);
    if (fix)
        k_pair_cache<G, EPL, T, true><<<grid, 256, 0, c->stream>>>(tok_w0, n_tok, emb, (uint32_t)c->emb_rows, (uint32_t)c->dp,
                                                                   c->gcache.as<double2>(), c->flags.as<uint32_t>(),
                                                                   c->inexact.as<uint32_t>());
    else
        k_pair_cache<G, EPL, T, false><<<grid, 256, 0, c->stream>>>(tok_w0, n_tok, emb, (uint32_t)c->emb_rows, (uint32_t)c->dp,
                                                                    c->gcache.as<double2>(), c->flags.as<uint32_t>(), nullptr);
}

#ifndef DOTS2_P
#define DOTS2_P 6
#endif
// k_dots2 (sage_kernels4.h) addresses rows by 32-bit byte offsets: matrices under 4 GiB (every configuration of the survey)
template <int G, int EPL, typename T, bool LIST>
void launch_dots2(sage_ctx* c, const T* emb, uint32_t n_occ, const uint32_t* list, const uint32_t* list_count) {
    // period: float32 slices of up to 8 elements plan 8 elements at a time (5 rows in flight: 3.96 ms per 300k lines against
    // 4.12 with 6 at D = 50, and the same way at D = 20, 100, 200); slices of 12 floats (D = 300) run better with 5 (21.2
    // against 22.9 with 6: 159 registers; k_dots: 22.0); float64 slices keep 6
    constexpr int P2 = EPL > 8 ? 5 : (std::is_same<T, float>::value ? 8 : DOTS2_P);
    static std::atomic<int> cached{0};
    int per_cu = cached.load(std::memory_order_relaxed);
    if (per_cu == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_dots2<G, EPL, T, P2, LIST>, 256, 0) != hipSuccess || per_cu < 1)
            per_cu = 4;
        cached.store(per_cu, std::memory_order_relaxed);
    }
    const uint32_t grid = std::min<uint32_t>(cdiv(n_occ, 256 / G), (uint32_t)c->n_cu * (uint32_t)std::max(per_cu, 1));
    k_dots2<G, EPL, T, P2, LIST><<<grid, 256, 0, c->stream>>>(
        c->desc.as<uint4>(), c->pool.as<uint32_t>(), n_occ, c->dot_off.as<uint32_t>(), emb, (uint32_t)c->emb_rows,
        (uint32_t)c->dp, c->dots.as<double>(), c->flags.as<uint32_t>(), list, list_count);
}

template <int G, int EPL, typename T, int NS = 5>
void launch_dots(sage_ctx* c, const T* emb, uint32_t n_occ, const uint32_t* list, const uint32_t* list_count, int cls) {
    Prof p(c, cls);
    if (c->dots2 && (uint64_t)c->emb_rows * (uint64_t)c->dp * sizeof(T) < (1ull << 32) && c->emb_rows < (1ll << 24)) {
        if (list) launch_dots2<G, EPL, T, true>(c, emb, n_occ, list, list_count);
        else launch_dots2<G, EPL, T, false>(c, emb, n_occ, nullptr, nullptr);
        return;
    }
    // persistent: exactly as many workgroups as the chip holds at once (a partial second round would double the time)
    static std::atomic<int> cached{0};   // per instantiation; the same for every gfx950 device
    int per_cu = cached.load(std::memory_order_relaxed);
    if (per_cu == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_dots<G, EPL, T, NS>, 256, 0) != hipSuccess || per_cu < 1)
            per_cu = 4;
        cached.store(per_cu, std::memory_order_relaxed);
    }
    const uint32_t grid = std::min<uint32_t>(cdiv(n_occ, 256 / G), (uint32_t)c->n_cu * (uint32_t)std::max(per_cu, 1));
    k_dots<G, EPL, T, NS><<<grid, 256, 0, c->stream>>>(
        c->desc.as<uint4>(), c->pool.as<uint32_t>(), n_occ, c->dot_off.as<uint32_t>(), emb, (uint32_t)c->emb_rows,
        (uint32_t)c->dp, c->dots.as<double>(), c->flags.as<uint32_t>(), list, list_count);
}
// the float64 pass over the inexact list with float32 rows where the copy holds them (k_dots<..., MIXED>)
#ifndef DOTS2_MIXED_P
#define DOTS2_MIXED_P 6
#endif
template <int G, int EPL>
void launch_dots_mixed(sage_ctx* c, uint32_t n_occ) {
    Prof p(c, SAGE_K_FIX);
    // (rows of 384 elements: the float64 kernels use 64 lanes x 6 doubles, k_dots2_mixed 32 lanes x 12 -- whole 16-byte
    // loads of a float32 slice)
    constexpr int G2 = (G == 64 && EPL == 6) ? 32 : G, EPL2 = (G == 64 && EPL == 6) ? 12 : EPL;
    if constexpr (EPL2 % 4 == 0) {
        if (c->dots2 && (uint64_t)c->emb_rows * (uint64_t)c->dp * 8u < (1ull << 32) && c->emb_rows < (1ll << 24)) {
            static std::atomic<int> cached2{0};
            int per_cu = cached2.load(std::memory_order_relaxed);
            if (per_cu == 0) {
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_dots2_mixed<G2, EPL2, DOTS2_MIXED_P>, 256, 0) != hipSuccess || per_cu < 1)
                    per_cu = 3;
                cached2.store(per_cu, std::memory_order_relaxed);
            }
            const uint32_t grid = std::min<uint32_t>(cdiv(n_occ, 256 / G2), (uint32_t)c->n_cu * (uint32_t)std::max(per_cu, 1));
            k_dots2_mixed<G2, EPL2, DOTS2_MIXED_P><<<grid, 256, 0, c->stream>>>(
                c->desc.as<uint4>(), c->pool.as<uint32_t>(), n_occ, c->dot_off.as<uint32_t>(), c->emb.as<double>(), (uint32_t)c->emb_rows,
                (uint32_t)c->dp, c->dots.as<double>(), c->flags.as<uint32_t>(), c->inex_list.as<uint32_t>(), c->inex_count.as<uint32_t>(),
                c->emb32.as<float>(), c->inexact.as<uint32_t>(), c->stats.as<unsigned long long>());
            return;
        }
    }
    static std::atomic<int> cached{0};
    int per_cu = cached.load(std::memory_order_relaxed);
    if (per_cu == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_dots_mixed<G, EPL, 5>, 256, 0) != hipSuccess || per_cu < 1)
            per_cu = 4;
        cached.store(per_cu, std::memory_order_relaxed);
    }
    const uint32_t grid = std::min<uint32_t>(cdiv(n_occ, 256 / G), (uint32_t)c->n_cu * (uint32_t)std::max(per_cu, 1));
    k_dots_mixed<G, EPL, 5><<<grid, 256, 0, c->stream>>>(
        c->desc.as<uint4>(), c->pool.as<uint32_t>(), n_occ, c->dot_off.as<uint32_t>(), c->emb.as<double>(), (uint32_t)c->emb_rows,
        (uint32_t)c->dp, c->dots.as<double>(), c->flags.as<uint32_t>(), c->inex_list.as<uint32_t>(), c->inex_count.as<uint32_t>(),
        c->emb32.as<float>(), c->inexact.as<uint32_t>(), c->stats.as<unsigned long long>());
}

// instantiation per padded row width: float32 rows are whole 16-byte loads of 4 elements, float64 rows of 2
#define SAGE_BY_WIDTH_F32(FN, ...)                       \
    switch (c->dp) {                                     \
    case 32: FN<8, 4>(__VA_ARGS__); break;               \
    case 64: FN<8, 8>(__VA_ARGS__); break;               \
    case 128: FN<16, 8>(__VA_ARGS__); break;             \
    case 256: FN<32, 8>(__VA_ARGS__); break;             \
    case 384: FN<32, 12>(__VA_ARGS__); break;            \
    default: FN<64, 8>(__VA_ARGS__); break;              \
    }
#define SAGE_BY_WIDTH_F64(FN, ...)                       \
    switch (c->dp) {                                     \
    case 32: FN<8, 4>(__VA_ARGS__); break;               \
    case 64: FN<8, 8>(__VA_ARGS__); break;               \
    case 128: FN<16, 8>(__VA_ARGS__); break;             \
    case 256: FN<32, 8>(__VA_ARGS__); break;             \
    case 384: FN<64, 6>(__VA_ARGS__); break;             \
    default: FN<64, 8>(__VA_ARGS__); break;              \
    }

// Scoring of one batch (compute_losses, utils.py:125-138).  Base windows: pair cache (k_pair_cache).  Ablated windows:
// dot products (k_dots), then log-sigmoid, base-window range sums and accumulation per ablated id (k_logsig).
// float32 row storage: k_describe has split the occurrences into those whose window rows the float32 copy holds exactly
// (their dots come from it) and the others (their dots come from the float64 matrix), so neither pass gathers rows for
// the other; token quads of the pair cache that touch such a row are recomputed from the float64 matrix (bitmap test).
void launch_score(sage_ctx* c, const uint32_t* tok_w0, uint32_t n_tok, uint32_t n_occ) {
    const double* e64 = c->emb.as<double>();
    const float* e32 = c->emb32.as<float>();
    const bool split = c->use_f32 && c->n_inexact > 0;
    if (c->use_f32) {
        SAGE_BY_WIDTH_F32(launch_pairs, c, e32, tok_w0, n_tok, false)
        if (split) { SAGE_BY_WIDTH_F64(launch_pairs, c, e64, tok_w0, n_tok, true) }
    } else {
        SAGE_BY_WIDTH_F64(launch_pairs, c, e64, tok_w0, n_tok, false)
    }
    if (!c->use_f32) {
        SAGE_BY_WIDTH_F64(launch_dots, c, e64, n_occ, nullptr, nullptr, SAGE_K_SCORE)
    } else if (!split) {
        SAGE_BY_WIDTH_F32(launch_dots, c, e32, n_occ, nullptr, nullptr, SAGE_K_SCORE)
    } else {
        SAGE_BY_WIDTH_F32(launch_dots, c, e32, n_occ, c->exact_list.as<uint32_t>(), c->exact_count.as<uint32_t>(), SAGE_K_SCORE)
        SAGE_BY_WIDTH_F64(launch_dots_mixed, c, n_occ)
    }
    {
        Prof p(c, SAGE_K_LOGSIG);
        const uint32_t grid = std::min<uint32_t>(cdiv(n_occ, LOGSIG_TILE), 256u * 8u);
        if (c->deterministic)
            k_logsig<true><<<grid, 256, 0, c->stream>>>(c->desc.as<uint4>(), c->occ_tok.as<uint32_t>(), c->occ_a.as<uint32_t>(), n_occ,
                                                        c->dot_off.as<uint32_t>(), c->dots.as<double>(), c->gcache.as<double2>(), n_tok,
                                                        c->loss.as<double>(), c->loss_fix.as<unsigned long long>(),
                                                        split ? c->inex_count.as<uint32_t>() : nullptr,
                                                        c->stats.as<unsigned long long>());
        else
            k_logsig<false><<<grid, 256, 0, c->stream>>>(c->desc.as<uint4>(), c->occ_tok.as<uint32_t>(), c->occ_a.as<uint32_t>(), n_occ,
                                                         c->dot_off.as<uint32_t>(), c->dots.as<double>(), c->gcache.as<double2>(), n_tok,
                                                         c->loss.as<double>(), nullptr,
                                                         split ? c->inex_count.as<uint32_t>() : nullptr,
                                                         c->stats.as<unsigned long long>());
    }
}

int step_impl(sage_ctx* c, const int64_t* bounds, int64_t n_seg, int32_t pad, int64_t vsize, int64_t hist_len,
              int64_t* totals /*[2] host*/) {
    int rc;
    if ((rc = check_ready(c, true))) return rc;
    if (!bounds || n_seg < 0 || vsize <= 0 || hist_len < 3) return fail(c, SAGE_E_ARG, "bad argument");
    if (pad != 2) return fail(c, SAGE_E_ARG, "only pad == 2 is implemented (model.py:138 default; utils.py:216 never overrides it)");
    if (vsize <= c->max_id) return fail(c, SAGE_E_ARG, "vsize must exceed the largest token id");
    for (int64_t s = 0; s < n_seg; ++s)
        if (bounds[s] > bounds[s + 1]) return fail(c, SAGE_E_ARG, "segment bounds must be non-decreasing");
    if (n_seg && (bounds[0] < 0 || bounds[n_seg] > c->n_lines)) return fail(c, SAGE_E_ARG, "segment bounds outside the corpus");
    HIP_TRY(c, hipSetDevice(c->device));

    if ((rc = ensure(c, c->loss, (size_t)vsize * 8))) return rc;
    if ((rc = ensure(c, c->occ, (size_t)vsize * 8))) return rc;
    if ((rc = ensure(c, c->hist, (size_t)hist_len * 8))) return rc;
    if ((rc = ensure(c, c->stats, STAT_COUNT * 8))) return rc;
    if ((rc = ensure(c, c->pool_cursor, 4))) return rc;
    if (c->deterministic && (rc = ensure(c, c->loss_fix, (size_t)vsize * 16))) return rc;

    for (int attempt = 0;; ++attempt) {
        HIP_TRY(c, hipMemsetAsync(c->loss.p, 0, (size_t)vsize * 8, c->stream));
        if (c->deterministic) HIP_TRY(c, hipMemsetAsync(c->loss_fix.p, 0, (size_t)vsize * 16, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->occ.p, 0, (size_t)vsize * 8, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->hist.p, 0, (size_t)hist_len * 8, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->stats.p, 0, STAT_COUNT * 8, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->flags.p, 0, FLAG_COUNT * 4, c->stream));
        uint64_t total_tokens = 0, total_bytes = 0, total_occ = 0;
        int n_batches = 0;

        int64_t sa = 0;
        bool shrink = false;
        while (sa < n_seg) {
            // a batch = as many whole segments as fit in batch_bytes (at least one)
            int64_t sb = sa + 1;
            while (sb < n_seg && c->h_line_off[bounds[sb + 1]] - c->h_line_off[bounds[sa]] <= c->batch_bytes) ++sb;
            const int64_t line0 = bounds[sa], line1 = bounds[sb];
            const uint32_t nsb = (uint32_t)(sb - sa);
            if (line1 == line0) { sa = sb; continue; }
            const int64_t pos0 = c->h_line_off[line0];
            const uint32_t nl = (uint32_t)(line1 - line0);
            total_bytes += (uint64_t)(c->h_line_off[line1] - pos0);

            if ((rc = match_and_count(c, line0, line1, nullptr))) return rc;

            // token offset of every segment start; totals of the batch
            std::vector<uint32_t> rel(nsb + 1);
            for (uint32_t s = 0; s <= nsb; ++s) rel[s] = (uint32_t)(bounds[sa + s] - line0);
            if ((rc = ensure(c, c->seg_line_rel, (size_t)(nsb + 1) * 4))) return rc;
            if ((rc = ensure(c, c->seg_tok_off, (size_t)(nsb + 1) * 4))) return rc;
            HIP_TRY(c, h2d_keep(c, c->seg_line_rel.p, rel, (size_t)nsb + 1));
            k_gather_u32<<<cdiv(nsb + 1, 256), 256, 0, c->stream>>>(c->tokoff.as<uint32_t>(), c->seg_line_rel.as<uint32_t>(),
                                                                    nsb + 1, c->seg_tok_off.as<uint32_t>());
            uint64_t gt[2] = {0, 0};
            HIP_TRY(c, hipMemcpyAsync(gt, c->grand_total.p, 16, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (gt[0] >= 0xFFFFFF00ull) return fail(c, SAGE_E_LIMIT, "a batch must stay under 2^32 tokens");
            const uint32_t n_tok = (uint32_t)gt[0], n_occ = (uint32_t)gt[1];
            total_tokens += n_tok;
            total_occ += n_occ;
            if (n_tok == 0) { sa = sb; continue; }

            if ((rc = ensure(c, c->tok_w0, (size_t)n_tok * 4))) return rc;
            if ((rc = ensure(c, c->occ_pos, (size_t)std::max(n_occ, 1u) * 4))) return rc;
            if ((rc = ensure(c, c->occ_tok, (size_t)std::max(n_occ, 1u) * 4))) return rc;
            {
                Prof p(c, SAGE_K_WALK);
                k_walk_write<false><<<cdiv(nl, 256) + long_blocks(c->n_lline), 256, 0, c->stream>>>(
                    c->line_off.as<int64_t>(), line0, nl, pos0, batch_M(c, pos0), c->tokoff.as<uint32_t>(),
                    c->occoff.as<uint32_t>(), c->tok_w0.as<uint32_t>(), c->occ_pos.as<uint32_t>(),
                    c->occ_tok.as<uint32_t>(), nullptr, nullptr, nullptr, c->lline_list.as<uint32_t>(), c->n_lline);
            }
            if (n_occ == 0) { HIP_TRY(c, hipGetLastError()); sa = sb; continue; }
            WalkArrays A{batch_M(c, pos0), (uint32_t)(c->h_line_off[line1] - pos0), c->tok_w0.as<uint32_t>(), c->occ_pos.as<uint32_t>(),
                         c->parent.as<uint32_t>()};
            if ((rc = ensure(c, c->set_base, (size_t)nsb * 8))) return rc;
            if ((rc = ensure(c, c->set_mask, (size_t)nsb * 8))) return rc;
            if ((rc = ensure(c, c->seg_pairs, (size_t)nsb * 8))) return rc;
            if ((rc = ensure(c, c->desc, (size_t)n_occ * SAGE_DESC_WORDS * 4))) return rc;
            if ((rc = ensure(c, c->gcache, (size_t)n_tok * 16))) return rc;
            if ((rc = ensure(c, c->occ_a, (size_t)n_occ * 4))) return rc;
            if ((rc = ensure(c, c->np_ab, (size_t)n_occ * 4))) return rc;
            if ((rc = ensure(c, c->dot_off, ((size_t)n_occ + 1) * 4))) return rc;
            // never smaller than the longest possible window (<= pads + one id per byte of a line), so that a window
            // that lost the race for pool space can still be given in-bounds (meaningless) ids: see k_describe<true>
            const uint64_t pool_cap = std::min<uint64_t>(std::max<uint64_t>((uint64_t)n_occ * c->pool_factor + 1024,
                                                                            (uint64_t)c->max_line_bytes + 16), 0xFFFFFF00ull);
            if ((rc = ensure(c, c->pool, pool_cap * 4))) return rc;
            HIP_TRY(c, hipMemsetAsync(c->seg_pairs.p, 0, (size_t)nsb * 8, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->pool_cursor.p, 0, 4, c->stream));
            const bool split = c->use_f32 && c->n_inexact > 0;
            if (split) {
                if ((rc = ensure(c, c->inex_list, (size_t)n_occ * 4))) return rc;
                if ((rc = ensure(c, c->exact_list, (size_t)n_occ * 4))) return rc;
                if ((rc = ensure(c, c->inex_count, 4))) return rc;
                if ((rc = ensure(c, c->exact_count, 4))) return rc;
                HIP_TRY(c, hipMemsetAsync(c->inex_count.p, 0, 4, c->stream));
                HIP_TRY(c, hipMemsetAsync(c->exact_count.p, 0, 4, c->stream));
            }
            SegPlan S{c->seg_tok_off.as<uint32_t>(), c->set_base.as<uint64_t>(), c->set_mask.as<uint64_t>(), nsb};
            // k_describe: exactly the workgroups the chip holds at once (5 per CU at 31 KB of LDS each): 4,096 of them ran in
            // 3.2 rounds, the last one a fifth full (7.6 -> 7.2 ms)
            const uint32_t cgrid = std::min<uint32_t>(cdiv(n_occ, 256), (uint32_t)c->n_cu * 5u);
            const uint32_t lgrid = std::min<uint32_t>(cdiv(n_occ, 256), 256u * 8u);   // long-list kernels (grid-stride)
            if ((rc = ensure(c, c->long_list, (size_t)n_occ * 4))) return rc;
            if ((rc = ensure(c, c->long_count, 4))) return rc;
            if ((rc = ensure(c, c->giant_list, (size_t)n_occ * 4))) return rc;
            if ((rc = ensure(c, c->giant_count, 4))) return rc;
            if ((rc = ensure(c, c->np_all, (size_t)n_occ * 4))) return rc;
            if ((rc = ensure(c, c->pair_off, ((size_t)n_occ + 1) * 4))) return rc;
            HIP_TRY(c, hipMemsetAsync(c->long_count.p, 0, 4, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->giant_count.p, 0, 4, c->stream));
            const DescribeArgs DA{c->desc.as<uint4>(), c->occ_a.as<uint32_t>(), c->np_ab.as<uint32_t>(), c->np_all.as<uint32_t>(),
                                  c->giant_list.as<uint32_t>(), c->giant_count.as<uint32_t>(),
                                  c->pool.as<uint32_t>(), c->pool_cursor.as<uint32_t>(), (uint32_t)pool_cap,
                                  c->seg_pairs.as<unsigned long long>(), c->occ.as<unsigned long long>(),
                                  c->hist.as<unsigned long long>(), (uint32_t)hist_len, c->stats.as<unsigned long long>(),
                                  c->flags.as<uint32_t>(), c->long_list.as<uint32_t>(), c->long_count.as<uint32_t>(),
                                  split ? c->inexact.as<uint32_t>() : nullptr,
                                  c->exact_list.as<uint32_t>(), c->exact_count.as<uint32_t>(),
                                  c->inex_list.as<uint32_t>(), c->inex_count.as<uint32_t>()};
            {
                Prof p(c, SAGE_K_DESCRIBE);
                k_describe<false><<<cgrid, 256, 0, c->stream>>>(A, c->occ_tok.as<uint32_t>(), n_occ, S, DA);
            }
            {
                Prof p(c, SAGE_K_MISC);   // the long-occurrence kernels are accounted under "misc"
                k_describe<true><<<lgrid, 256, 0, c->stream>>>(A, c->occ_tok.as<uint32_t>(), n_occ, S, DA);
            }
            // offsets of every occurrence's ablated-window dots; grand_total[1] holds the dot total
            if ((rc = scan_u32(c, c->np_ab.as<uint32_t>(), n_occ, c->dot_off.as<uint32_t>(), c->grand_total.as<uint64_t>() + 1))) return rc;
            uint64_t n_dots = 0;
            if (c->count_mode == 0 && !c->count_fallback) {
                // ---- default: one partition level into LDS-set-sized buckets, every record read once (sage_kernels3.h) ----
                if ((rc = scan_u32(c, c->np_all.as<uint32_t>(), n_occ, c->pair_off.as<uint32_t>(), c->grand_total.as<uint64_t>()))) return rc;
                if ((rc = ensure(c, c->seg_first_occ, (size_t)(nsb + 1) * 4))) return rc;
                if ((rc = ensure(c, c->seg_tile_off, (size_t)(nsb + 1) * 4))) return rc;
                if ((rc = ensure(c, c->seg_bucket_base, (size_t)(nsb + 1) * 4))) return rc;
                if ((rc = ensure(c, c->seg_nbk_mask, (size_t)nsb * 4))) return rc;
                k_tile_plan<<<1, 256, 0, c->stream>>>(c->occ_tok.as<uint32_t>(), n_occ, c->seg_tok_off.as<uint32_t>(), nsb,
                                                      c->pair_off.as<uint32_t>(), c->seg_first_occ.as<uint32_t>(),
                                                      c->seg_tile_off.as<uint32_t>());
                std::vector<uint64_t> spairs(nsb);
                uint64_t gt2[2] = {0, 0};
                HIP_TRY(c, hipMemcpyAsync(spairs.data(), c->seg_pairs.p, (size_t)nsb * 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipMemcpyAsync(gt2, c->grand_total.p, 16, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                n_dots = gt2[1];
                if (gt2[0] >= c->pair_limit) {   // 32-bit pair offsets: halve the batches and redo the step
                    if (nsb <= 1) return fail(c, SAGE_E_LIMIT, "a count segment must stay under 2^32 position pairs");
                    shrink = true;
                    break;
                }
                // buckets per segment: a power of two that keeps a bucket near 8k records (one pass of the 16k-slot LDS set)
                std::vector<uint32_t> bbase(nsb + 1), bmask(nsb);
                uint32_t nbk_total = 0;
                for (uint32_t s2 = 0; s2 < nsb; ++s2) {
                    uint32_t nbk = 1;
                    while (nbk < PART_NBK_MAX && (uint64_t)nbk * 8192u < spairs[s2]) nbk <<= 1;
                    bbase[s2] = nbk_total; bmask[s2] = nbk - 1u; nbk_total += nbk;
                }
                bbase[nsb] = nbk_total;
                uint64_t max_tiles64 = 0;       // tiles are cut per segment: an upper bound from the segments' pair totals
                for (uint32_t s2 = 0; s2 < nsb; ++s2) max_tiles64 += (spairs[s2] + PART_FILL - 1) / PART_FILL;
                const uint32_t max_tiles = (uint32_t)std::max<uint64_t>(max_tiles64, 1);
                if ((rc = ensure(c, c->tile_tab, (size_t)max_tiles * 16))) return rc;
                if ((rc = ensure(c, c->cursor, (size_t)std::max(nbk_total, 1u) * 4))) return rc;
                if ((rc = ensure(c, c->bucket_off, ((size_t)nbk_total + 1) * 8))) return rc;
                HIP_TRY(c, h2d_keep(c, c->seg_bucket_base.p, bbase, (size_t)nsb + 1));
                HIP_TRY(c, h2d_keep(c, c->seg_nbk_mask.p, bmask, (size_t)nsb));
                PartPlan PP{c->pair_off.as<uint32_t>(), c->seg_first_occ.as<uint32_t>(), c->seg_tile_off.as<uint32_t>(),
                            c->seg_bucket_base.as<uint32_t>(), c->seg_nbk_mask.as<uint32_t>(),
                            c->bucket_off.as<unsigned long long>(), nsb};
                const size_t part_lds = PART_LDS_BYTES;     // (more than 64 KiB of dynamic LDS: asked for per device in sage_create)
                const size_t set_lds = SET3_LDS_BYTES;
                k_tile_table<<<std::min<uint32_t>(cdiv(max_tiles, 256), 1024u), 256, 0, c->stream>>>(PP, max_tiles, c->tile_tab.as<uint4>());
                const uint32_t tgrid = (uint32_t)c->n_cu;      // persistent: the stage fills a CU's LDS
                const uint32_t ggrid = 64u;
                std::vector<uint64_t> boff((size_t)nbk_total + 1);
                uint64_t total = 0;
                uint2* tile_dbg = nullptr;
                if (c->debug_part) {
                    if ((rc = ensure(c, c->tile_dbg, (size_t)max_tiles * 8))) return rc;
                    tile_dbg = c->tile_dbg.as<uint2>();
                }
                std::vector<uint32_t> cnts;
                if (c->exact_buckets || c->debug_part) {
                    // counting pass: exact bucket sizes (taken after an estimated region overflowed)
                    HIP_TRY(c, hipMemsetAsync(c->cursor.p, 0, (size_t)std::max(nbk_total, 1u) * 4, c->stream));
                    {
                        Prof p(c, SAGE_K_PARTITION);
                        k_key_partition3<0><<<tgrid, PART_THREADS, part_lds, c->stream>>>(
                            c->desc.as<uint4>(), c->pool.as<uint32_t>(), c->occ_tok.as<uint32_t>(), A.tok_w0, PP,
                            c->tile_tab.as<uint4>(), max_tiles, c->cursor.as<uint32_t>(), nullptr, c->flags.as<uint32_t>(), tile_dbg);
                        k_key_giant<0><<<ggrid, 256, 0, c->stream>>>(c->desc.as<uint4>(), c->pool.as<uint32_t>(),
                                                                     c->occ_tok.as<uint32_t>(), A.tok_w0,
                                                                     c->giant_list.as<uint32_t>(), c->giant_count.as<uint32_t>(), S, PP,
                                                                     c->cursor.as<uint32_t>(), nullptr, c->flags.as<uint32_t>());
                    }
                    cnts.resize(nbk_total);
                    HIP_TRY(c, hipMemcpyAsync(cnts.data(), c->cursor.p, (size_t)nbk_total * 4, hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                }
                if (c->exact_buckets) {
                    for (uint32_t b2 = 0; b2 < nbk_total; ++b2) { boff[b2] = total; total += cnts[b2]; }
                } else {
                    // no counting pass: the hash spreads a segment's pairs evenly over its buckets, so each region is the
                    // average plus room for repeated keys (all copies of a key share a bucket); a region that still
                    // overflows makes the step redo itself with the counting pass
                    for (uint32_t s2 = 0; s2 < nsb; ++s2) {
                        const uint64_t nbk = (uint64_t)bmask[s2] + 1u;
                        const uint64_t avg = spairs[s2] / nbk;
                        const uint64_t cap = avg + avg / 2 + std::min<uint64_t>(2048, spairs[s2]);
                        for (uint64_t b2 = 0; b2 < nbk; ++b2) { boff[bbase[s2] + b2] = total; total += cap; }
                    }
                }
                boff[nbk_total] = total;
                if ((rc = ensure(c, c->rec_a, std::max<uint64_t>(total, 1) * 8))) return rc;
                HIP_TRY(c, h2d_keep(c, c->bucket_off.p, boff, (size_t)nbk_total + 1));
                HIP_TRY(c, hipMemsetAsync(c->cursor.p, 0, (size_t)std::max(nbk_total, 1u) * 4, c->stream));
                {
                    Prof p(c, SAGE_K_PARTITION);
                    k_key_partition3<1><<<tgrid, PART_THREADS, part_lds, c->stream>>>(
                        c->desc.as<uint4>(), c->pool.as<uint32_t>(), c->occ_tok.as<uint32_t>(), A.tok_w0, PP,
                        c->tile_tab.as<uint4>(), max_tiles, c->cursor.as<uint32_t>(), c->rec_a.as<unsigned long long>(),
                        c->flags.as<uint32_t>(), tile_dbg);
                }
                {
                    Prof p(c, SAGE_K_MISC);
                    k_key_giant<1><<<ggrid, 256, 0, c->stream>>>(c->desc.as<uint4>(), c->pool.as<uint32_t>(),
                                                                 c->occ_tok.as<uint32_t>(), A.tok_w0,
                                                                 c->giant_list.as<uint32_t>(), c->giant_count.as<uint32_t>(), S, PP,
                                                                 c->cursor.as<uint32_t>(), c->rec_a.as<unsigned long long>(),
                                                                 c->flags.as<uint32_t>());
                }
                if (c->debug_part) {      // every bucket's scatter cursor against the counting pass
                    std::vector<uint32_t> after(nbk_total);
                    HIP_TRY(c, hipMemcpyAsync(after.data(), c->cursor.p, (size_t)nbk_total * 4, hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    uint64_t sum_cnt = 0, sum_after = 0, sum_pairs = 0;
                    uint32_t shown = 0, differ = 0;
                    for (uint32_t s2 = 0; s2 < nsb; ++s2) sum_pairs += spairs[s2];
                    for (uint32_t b2 = 0; b2 < nbk_total; ++b2) {
                        sum_cnt += cnts[b2]; sum_after += after[b2];
                        if (cnts[b2] != after[b2]) {
                            ++differ;
                            if (shown++ < 16) std::fprintf(stderr, "[sage debug] bucket %u: counted %u, scattered %u, capacity %llu\n", b2, cnts[b2],
                                                           after[b2], (unsigned long long)(boff[b2 + 1] - boff[b2]));
                        }
                    }
                    std::fprintf(stderr, "[sage debug] batch of %u segments, %u tiles max, %u buckets: pairs %llu (scan total %llu + giants), counted %llu, scattered %llu, %u buckets differ, exact=%d\n",
                                 nsb, max_tiles, nbk_total, (unsigned long long)sum_pairs, (unsigned long long)gt2[0], (unsigned long long)sum_cnt,
                                 (unsigned long long)sum_after, differ, (int)c->exact_buckets);
                }
                {
                    Prof p(c, SAGE_K_COUNT);
                    const uint32_t dgrid = std::min<uint32_t>(std::max(nbk_total, 1u), (uint32_t)c->n_cu);
                    k_key_dedupe3<<<dgrid, PART_THREADS, set_lds, c->stream>>>(c->bucket_off.as<unsigned long long>(),
                                                                               c->cursor.as<uint32_t>(), nbk_total,
                                                                               c->rec_a.as<unsigned long long>(),
                                                                               c->stats.as<unsigned long long>(), c->flags.as<uint32_t>());
                }
            } else {
                // ---- count mode 1: one distinct-key set per segment in HBM, sized from the exact pair count: load <= 0.5 ----
                std::vector<uint64_t> sbase(nsb), smask(nsb), spairs(nsb);
                HIP_TRY(c, hipMemcpyAsync(spairs.data(), c->seg_pairs.p, (size_t)nsb * 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipMemcpyAsync(&n_dots, c->grand_total.as<uint64_t>() + 1, 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                uint64_t slots = 0;
                for (uint32_t s = 0; s < nsb; ++s) {
                    const uint64_t cap = pow2ceil(std::max<uint64_t>(64, spairs[s] * 2));
                    sbase[s] = slots; smask[s] = cap - 1; slots += cap;
                }
                if ((rc = ensure(c, c->set, slots * 8))) return rc;
                HIP_TRY(c, h2d_keep(c, c->set_base.p, sbase, (size_t)nsb));
                HIP_TRY(c, h2d_keep(c, c->set_mask.p, smask, (size_t)nsb));
                {
                    Prof p(c, SAGE_K_MISC);
                    HIP_TRY(c, hipMemsetAsync(c->set.p, 0xFF, slots * 8, c->stream));
                }
                {
                    Prof p(c, SAGE_K_COUNT);
                    k_count_sets<<<cgrid, 256, 0, c->stream>>>(c->desc.as<uint4>(), c->pool.as<uint32_t>(),
                                                               c->occ_tok.as<uint32_t>(), A.tok_w0, n_occ, S,
                                                               c->set.as<unsigned long long>(),
                                                               c->stats.as<unsigned long long>(), c->flags.as<uint32_t>());
                }
            }
            if (n_dots >= c->pair_limit) {
                if (nsb <= 1) return fail(c, SAGE_E_LIMIT, "a count segment must stay under 2^32 ablated-window pairs");
                shrink = true;
                break;
            }
            if ((rc = ensure(c, c->dots, std::max<uint64_t>(n_dots, 1) * 8))) return rc;
            launch_score(c, A.tok_w0, n_tok, n_occ);
            c->last_n_occ = n_occ; c->last_split = split; ++n_batches;
            HIP_TRY(c, hipGetLastError());
            sa = sb;
        }
        if (shrink) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            c->keep32.clear(); c->keep64.clear();
            collect_events(c);
            if (c->batch_bytes <= 4096) return fail(c, SAGE_E_LIMIT, "position pairs overflow 32-bit offsets even in the smallest batches");
            c->batch_bytes = std::max<int64_t>(c->batch_bytes / 2, 4096);     // every retry halves it: the loop ends
            continue;
        }
        if (c->deterministic)
            k_fix_finish<<<cdiv((uint64_t)vsize, 256), 256, 0, c->stream>>>(c->loss_fix.as<unsigned long long>(), (uint32_t)vsize,
                                                                           c->loss.as<double>());
        uint32_t flags[FLAG_COUNT];
        uint64_t stats[STAT_COUNT];
        HIP_TRY(c, hipMemcpyAsync(flags, c->flags.p, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(stats, c->stats.p, sizeof(stats), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->keep32.clear(); c->keep64.clear();
        collect_events(c);
        if (flags[FLAG_EMPTY_LINE])
            return fail(c, SAGE_E_EMPTY_LINE, "not enough values to unpack (expected 3, got 0)");
        if (flags[FLAG_POOL_FULL]) {
            if (attempt >= 6) return fail(c, SAGE_E_LIMIT, "window-id pool overflowed");
            c->pool_factor = std::max(1, c->pool_factor) * 4;   // long ablation chains: redo the step with a larger id pool
            continue;
        }
        if (c->debug_part)
            std::fprintf(stderr, "[sage debug] attempt %d: flags set_overflow=%u pool_full=%u bucket_overflow=%u part_mismatch=%u; pairs %llu, records read %llu; exact_buckets=%d fallback=%d batch_bytes=%lld\n",
                         attempt, flags[FLAG_SET_OVERFLOW], flags[FLAG_POOL_FULL], flags[FLAG_BUCKET_OVERFLOW], flags[FLAG_PART_MISMATCH],
                         (unsigned long long)stats[STAT_PAIRS], (unsigned long long)stats[STAT_RECORDS], (int)c->exact_buckets,
                         (int)c->count_fallback, (long long)c->batch_bytes);
        const bool partitioned = c->count_mode == 0 && !c->count_fallback;
        if (flags[FLAG_BUCKET_OVERFLOW] && partitioned && !c->exact_buckets) {
            c->exact_buckets = true;   // heavily repeated keys: redo the step with the counting pass
            continue;
        }
        // The partitioned count checks itself: regions sized by the counting pass cannot overflow, no tile may hold more
        // records than its stage, and the dedupe kernel must have read exactly one record per position pair.  A step
        // that fails any of these is redone with the per-segment HBM sets (exact whatever the key statistics) and
        // counted in step statistic 13, which the tests hold at zero.
        if (partitioned && !flags[FLAG_SET_OVERFLOW] &&
            (flags[FLAG_BUCKET_OVERFLOW] || flags[FLAG_PART_MISMATCH] || stats[STAT_RECORDS] != stats[STAT_PAIRS])) {
            if (c->debug_part) return fail(c, SAGE_E_LIMIT, "the partitioned distinct count failed its own checks (SAGE_DEBUG_PART is set: no fallback)");
            ++c->count_redos;
            c->count_fallback = true;
            continue;
        }
        if (flags[FLAG_SET_OVERFLOW]) {
            if (c->count_mode == 1 || c->count_fallback) return fail(c, SAGE_E_LIMIT, "distinct-key set overflowed");
            c->count_fallback = true;   // exactness first: an LDS set overflowed (adversarial hash skew) -> redo with HBM sets
            continue;
        }
        if (flags[FLAG_ROW_OOB])
            return fail(c, SAGE_E_ROW, "a token id is out of bounds for the embedding matrix");
        c->last_batches = n_batches;
        const bool ran_without_matcher = c->m_valid;
        if (c->m_valid) ++c->steps_without_matcher;
        else if (c->m_whole && n_seg > 0 && bounds[0] == 0 && bounds[n_seg] == c->n_lines) c->m_valid = true;   // every line was matched
        totals[0] = (int64_t)total_tokens;
        totals[1] = (int64_t)stats[STAT_DISTINCT];
        c->step_stats[0] = (int64_t)total_tokens;
        c->step_stats[1] = (int64_t)total_bytes;
        c->step_stats[2] = (int64_t)total_occ;
        c->step_stats[3] = (int64_t)stats[STAT_ROWS];
        c->step_stats[4] = (int64_t)stats[STAT_PAIRS] * 2;   // signed (t,c) records: both orders
        c->step_stats[5] = (int64_t)stats[STAT_PAIRS];
        c->step_stats[6] = c->use_f32 ? 4 : 8;   // bytes per stored embedding element on the main scoring pass
        c->step_stats[7] = (int64_t)stats[STAT_INEXACT];
        c->step_stats[8] = (int64_t)stats[STAT_WROWS];
        c->step_stats[9] = (int64_t)stats[STAT_LONG];
        c->step_stats[10] = (int64_t)c->dim;
        c->step_stats[11] = (int64_t)c->dp;
        c->step_stats[12] = (int64_t)stats[STAT_ABPAIRS];
        c->step_stats[16] = (int64_t)stats[STAT_WROWS_INEXACT];
        c->step_stats[17] = (int64_t)stats[STAT_MARKED_ROWS];
        c->step_stats[13] = c->count_redos;
        c->step_stats[14] = (c->count_mode == 1 || c->count_fallback) ? 1 : 0;   // which back end produced total_triples
        c->step_stats[15] = (c->exact_buckets ? 1 : 0) | (ran_without_matcher ? 2 : 0);
        // The share of ROWS that float32 cannot hold says little about the share of OCCURRENCES that touch one; the
        // step just measured the latter.  With k_dots_mixed (matrices of 4 GiB and more, or SAGE_DOTS2=0) the float64 pass
        // costs more than the float32 copy saves above 30 %: later steps (until the next sage_set_embeddings) then use the
        // float64 matrix alone.  With k_dots2_mixed the split wins at every share (79 % on the float64 pass: 9.2 ms per
        // 300k lines against 10.5 for the float64 matrix alone at D = 50, 50.6 against 65.6 at D = 300), so there is no
        // switch.  Results do not depend on this.
        const bool mixed2 = c->dots2 && (uint64_t)c->emb_rows * (uint64_t)c->dp * 8u < (1ull << 32) && c->emb_rows < (1ll << 24);
        const double thr = c->f64_switch >= 0.0 ? c->f64_switch : (mixed2 ? 2.0 : 0.30);
        if (c->use_f32 && c->score_mode == 0 && (double)stats[STAT_INEXACT] > thr * (double)total_occ) c->use_f32 = false;
        return SAGE_OK;
    }
}

}  // namespace

extern "C" {

const char* sage_version(void) { return SAGE_VERSION_STR; }

int sage_create(int device, sage_ctx** out_ctx) {
    if (!out_ctx) return SAGE_E_ARG;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return SAGE_E_HIP;
    if (hipSetDevice(device) != hipSuccess) return SAGE_E_HIP;
    sage_ctx* c = new sage_ctx();
    c->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
    }
    // test hook: start with a smaller id pool so that the overflow-and-redo path can be exercised
    if (const char* e = std::getenv("SAGE_POOL_FACTOR")) { const int v = std::atoi(e); if (v >= 0 && v <= 64) c->pool_factor = v; }
    if (scratch_for_device(device, &c->na_scratch) != hipSuccess) { delete c; return SAGE_E_HIP; }   // device is current
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SAGE_E_HIP; }
    // kernels that use more than 64 KiB of dynamic LDS must say so, on every device they run on (the device is current)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_key_partition3<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PART_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_key_partition3<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PART_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_key_dedupe3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SET3_LDS_BYTES) != hipSuccess) {
        (void)hipStreamDestroy(c->stream); delete c; return SAGE_E_HIP;
    }
    if (const char* e = std::getenv("SAGE_DEBUG_PART")) c->debug_part = std::atoi(e) != 0;
    if (const char* e = std::getenv("SAGE_DOTS2")) c->dots2 = std::atoi(e) != 0;
    if (const char* e = std::getenv("SAGE_LONG_WALK")) c->long_walk = std::atoi(e) != 0;
    if (const char* e = std::getenv("SAGE_MATCH_THREADS")) { const int x = std::atoi(e); if (x >= 64 && x <= 1024 && x % 64 == 0) c->match_threads = (uint32_t)x; }
    if (const char* e = std::getenv("SAGE_F64_SWITCH")) { const double x = std::atof(e); if (x >= 0.0) c->f64_switch = x; }
    if (const char* e = std::getenv("SAGE_M_RESIDENT_MAX_GB")) { const long long x = std::atoll(e); if (x >= 0) c->m_resident_max = (int64_t)x << 30; }
    if (const char* e = std::getenv("SAGE_TEST_PAIR_LIMIT")) { const long long x = std::atoll(e); if (x > 0) c->pair_limit = (uint64_t)x; }
    if (hipMalloc(&c->flags.p, FLAG_COUNT * 4) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return SAGE_E_HIP; }
    c->flags.cap = FLAG_COUNT * 4;
    (void)hipMemset(c->flags.p, 0, FLAG_COUNT * 4);
    *out_ctx = c;
    return SAGE_OK;
}

void sage_destroy(sage_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    collect_events(c);
    for (auto e : c->ev_free) (void)hipEventDestroy(e);
    DevBuf* all[] = {&c->buckets, &c->keys, &c->ebk, &c->parent, &c->parent_old, &c->old_to_new, &c->remap, &c->emb_stage, &c->emb_cnt, &c->lline_list, &c->lline_count, &c->vk_bytes, &c->vk_off, &c->vk_ids, &c->byte_id, &c->direct2, &c->bloom, &c->pmask, &c->pmask8, &c->corpus, &c->rem, &c->line_off, &c->emb, &c->M, &c->ntok,
                     &c->tokoff, &c->tile_sums, &c->grand_total, &c->tok_w0, &c->occ_pos, &c->set, &c->seg_line_rel,
                     &c->seg_tok_off, &c->set_base, &c->set_mask, &c->seg_pairs, &c->desc, &c->gcache, &c->bucket_off, &c->rec_a, &c->nocc, &c->occoff, &c->occ_tok,
                     &c->occ_a, &c->exact_list, &c->exact_count, &c->np_ab, &c->dot_off, &c->dots, &c->pool, &c->pool_cursor, &c->np_all, &c->pair_off, &c->seg_first_occ, &c->seg_tile_off,
                     &c->seg_bucket_base, &c->seg_nbk_mask, &c->cursor, &c->giant_list, &c->giant_count, &c->tile_tab, &c->emb32, &c->inexact, &c->inex_list, &c->inex_count, &c->long_list, &c->long_count, &c->w2v_counts, &c->w2v_keep, &c->w2v_dropped, &c->w2v_aprob, &c->w2v_aidx, &c->w2v_syn0, &c->w2v_syn1, &c->ex_ids, &c->ex_starts, &c->ex_widths, &c->loss,
                     &c->occ, &c->hist, &c->stats, &c->flags, &c->loss_fix, &c->tile_dbg};
    for (DevBuf* b : all) release(*b);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

#ifdef MATCH_STATS
extern "C" int sage_debug_match_stats(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_match_dbg), 16 * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_match_dbg), z, 16 * 8) != hipSuccess) return 1; }
    return 0;
}
#endif
#ifdef PHASE_CLOCKS
extern "C" int sage_debug_phase_clocks(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_dbg), 48 * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[48] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_dbg), z, 48 * 8) != hipSuccess) return 1; }
    return 0;
}
#endif
const char* sage_last_error(const sage_ctx* c) { return c ? c->err.c_str() : "null context"; }

int sage_set_vocab(sage_ctx* c, const uint8_t* key_bytes, const int64_t* key_offsets, const int32_t* key_ids,
                   int64_t n_keys, int32_t max_len) {
    if (!c || !key_bytes || !key_offsets || n_keys <= 0) return fail(c, SAGE_E_ARG, "bad argument");
    if (max_len < 1 || max_len > SAGE_MAX_KEY_LEN) return fail(c, SAGE_E_LIMIT, "max_len must be in 1..32");
    HIP_TRY(c, hipSetDevice(c->device));
    int64_t max_id = -1;
    for (int64_t i = 0; i < n_keys; ++i) {
        const int64_t id = key_ids ? key_ids[i] : i;
        const int64_t len = key_offsets[i + 1] - key_offsets[i];
        if (id < 0 || id > (int64_t)SAGE_MAX_ID) return fail(c, SAGE_E_LIMIT, "token ids must be below 2^20 - 1");
        if (len < 1) return fail(c, SAGE_E_ARG, "empty key");
        // a key longer than max_len can never match (model.py:90 tries lengths max_len..1 only; the reference merely
        // warns about such a vocabulary, SaGeVocabBuilder.py:52-54): it keeps its id but is left out of every table
        max_id = std::max(max_id, id);
    }
    // The tables are built on the GPU (k_vocab_build, one thread per key): only the packed keys travel.
    const int nw = (max_len + 7) / 8;
    const size_t key_plane = ((size_t)max_id + 1 + 15) & ~(size_t)15;   // ids per 8-byte-word plane
    // slot table: load factor <= 0.25: a probe of an absent key continues into the next bucket only when this one is
    // full (2 % of the buckets at this load, 14 % at 0.5), and that continuation is a serial, divergent path
    const uint64_t nbuckets = pow2ceil(std::max<uint64_t>(16, (uint64_t)n_keys));
    const uint32_t bmask = (uint32_t)(nbuckets - 1);
    // embedded-key table (VocabDev::ebk): keys of 3..max_len bytes, load <= 0.125 unless SAGE_ETABLE_LOAD says otherwise.
    // What the load buys is not fewer cache misses (the table is far larger than L2 either way) but fewer FULL buckets:
    // a probe that meets one cannot conclude "absent" and walks on serially while its wave waits (1.9 % of the buckets
    // at 0.25, 0.2 % at 0.125).  Measured on 152 MB: 6.03 ms at 0.125, 6.18 at 0.18, 6.55 at 0.25, 7.8 at 0.35; no
    // further gain below 0.125 (6.10 at 0.0625).
    int64_t n_long = 0;
    for (int64_t i = 0; i < n_keys; ++i) {
        const int64_t len = key_offsets[i + 1] - key_offsets[i];
        n_long += len >= 3 && len <= max_len;
    }
    double eload = 0.125;
    if (const char* e = std::getenv("SAGE_ETABLE_LOAD")) { const double x = std::atof(e); if (x >= 0.05 && x <= 0.9) eload = x; }
    const uint32_t enb = (uint32_t)std::max<int64_t>(16, (int64_t)std::ceil((double)n_long / (4.0 * eload)));
    const bool pm16 = max_len <= 16;      // the matcher instantiations for max_len <= 16 read 16-bit prefix-table entries
    const size_t total_key_bytes = (size_t)key_offsets[n_keys];
    int rc;
    c->have_vocab = false;      // the tables are rewritten in place: no vocabulary until all of them are complete
    c->m_valid = false;         // (sage_set_vocab_pruned re-validates the match table after it has moved the matches)
    if ((rc = ensure(c, c->buckets, nbuckets * 16))) return rc;
    if ((rc = ensure(c, c->keys, key_plane * (size_t)nw * 8))) return rc;
    if ((rc = ensure(c, c->ebk, (size_t)enb * 64))) return rc;
    if ((rc = ensure(c, c->byte_id, 256 * 4))) return rc;
    if ((rc = ensure(c, c->direct2, 65536 * 4))) return rc;
    if ((rc = ensure(c, c->bloom, SAGE_BLOOM_BITS / 8))) return rc;
    if ((rc = ensure(c, c->pmask, (size_t)SAGE_PMASK_SLOTS * 4))) return rc;
    if ((rc = ensure(c, c->pmask8, (size_t)SAGE_PMASK8_SLOTS * 4))) return rc;
    if ((rc = ensure(c, c->parent, ((size_t)max_id + 1) * 4))) return rc;
    if ((rc = ensure(c, c->vk_bytes, total_key_bytes + 16))) return rc;
    if ((rc = ensure(c, c->vk_off, (size_t)(n_keys + 1) * 8))) return rc;
    if (key_ids && (rc = ensure(c, c->vk_ids, (size_t)n_keys * 4))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->vk_bytes.p, key_bytes, total_key_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->vk_off.p, key_offsets, (size_t)(n_keys + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (key_ids) HIP_TRY(c, hipMemcpyAsync(c->vk_ids.p, key_ids, (size_t)n_keys * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->buckets.p, 0xFF, nbuckets * 16, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->ebk.p, 0xFF, (size_t)enb * 64, c->stream));      // meta words = SAGE_SLOT_EMPTY
    HIP_TRY(c, hipMemsetAsync(c->keys.p, 0, key_plane * (size_t)nw * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->byte_id.p, 0xFF, 256 * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->direct2.p, 0xFF, 65536 * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->bloom.p, 0, SAGE_BLOOM_BITS / 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->pmask.p, 0, (size_t)SAGE_PMASK_SLOTS * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->pmask8.p, 0, (size_t)SAGE_PMASK8_SLOTS * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->parent.p, 0, ((size_t)max_id + 1) * 4, c->stream));
    VocabBuild vb{c->vk_bytes.as<uint8_t>(), c->vk_off.as<int64_t>(), key_ids ? c->vk_ids.as<int32_t>() : nullptr, (uint32_t)n_keys,
                  (uint32_t)max_len, nw, c->buckets.as<uint32_t>(), bmask, c->keys.as<uint64_t>(), (uint32_t)key_plane,
                  c->byte_id.as<uint32_t>(), c->direct2.as<uint32_t>(), c->bloom.as<uint32_t>(), c->pmask.as<uint32_t>(),
                  c->pmask8.as<uint32_t>(), pm16 ? 1 : 0, c->ebk.as<uint32_t>(), enb};
    k_vocab_build<<<cdiv((uint64_t)n_keys, 256), 256, 0, c->stream>>>(vb);
    HIP_TRY(c, hipGetLastError());
    {   // parent[id]: the longest proper prefix that is a key (probes the tables the kernel above has just completed)
        const VocabDev vd{c->buckets.as<uint4>(), bmask, c->keys.as<uint64_t>(), (uint32_t)key_plane, c->byte_id.as<uint32_t>(),
                          c->direct2.as<uint32_t>(), c->bloom.as<uint32_t>(), c->pmask.p, c->pmask8.p, max_len,
                          c->ebk.as<uint4>(), enb, nullptr};
        const uint32_t pgrid = cdiv((uint64_t)n_keys, 256);
        switch (nw) {
        case 1: k_vocab_parents<1><<<pgrid, 256, 0, c->stream>>>(vb, vd, c->parent.as<uint32_t>()); break;
        case 2: k_vocab_parents<2><<<pgrid, 256, 0, c->stream>>>(vb, vd, c->parent.as<uint32_t>()); break;
        case 3: k_vocab_parents<3><<<pgrid, 256, 0, c->stream>>>(vb, vd, c->parent.as<uint32_t>()); break;
        default: k_vocab_parents<4><<<pgrid, 256, 0, c->stream>>>(vb, vd, c->parent.as<uint32_t>()); break;
        }
        HIP_TRY(c, hipGetLastError());
    }
    uint32_t byte_id[256];
    HIP_TRY(c, hipMemcpyAsync(byte_id, c->byte_id.p, sizeof byte_id, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int b = 0; b < 256; ++b)
        if (byte_id[b] == SAGE_SLOT_EMPTY) {
            char msg[64];
            std::snprintf(msg, sizeof msg, "missing byte 0x%02x", b);
            return fail(c, SAGE_E_MISSING_BYTE, msg);
        }
    c->bucket_mask = bmask;
    c->enb = enb;
    c->key_plane = (uint32_t)key_plane;
    c->max_len = max_len;
    c->max_id = max_id;
    c->have_vocab = true;
    c->exact_buckets = false;      // fallbacks a step took because of the old vocabulary's key statistics
    c->count_fallback = false;
    return SAGE_OK;
}

int sage_set_vocab_pruned(sage_ctx* c, const uint8_t* key_bytes, const int64_t* key_offsets, const int32_t* key_ids,
                          int64_t n_keys, int32_t max_len, const int32_t* old_to_new, int64_t n_old) {
    if (!c || !old_to_new) return fail(c, SAGE_E_ARG, "bad argument");
    if (!c->have_vocab) return fail(c, SAGE_E_STATE, "sage_set_vocab_pruned follows a sage_set_vocab");
    if (max_len != c->max_len) return fail(c, SAGE_E_ARG, "sage_set_vocab_pruned must keep max_len");
    if (n_old != c->max_id + 1) return fail(c, SAGE_E_ARG, "old_to_new must cover the previous vocabulary's id space");
    HIP_TRY(c, hipSetDevice(c->device));
    const bool move_matches = c->m_valid;
    // the previous vocabulary's parents outlive the rebuild: the remap below walks them
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::swap(c->parent, c->parent_old);
    int rc = sage_set_vocab(c, key_bytes, key_offsets, key_ids, n_keys, max_len);     // (leaves m_valid false)
    if (rc) return rc;
    if (!move_matches) return SAGE_OK;
    for (int64_t i = 0; i < n_old; ++i)
        if (old_to_new[i] < -1 || old_to_new[i] > c->max_id) return fail(c, SAGE_E_ARG, "old_to_new holds an id outside the new vocabulary");
    if ((rc = ensure(c, c->old_to_new, (size_t)n_old * 4))) return rc;
    if ((rc = ensure(c, c->remap, (size_t)n_old * 4))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->old_to_new.p, old_to_new, (size_t)n_old * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->flags.p, 0, FLAG_COUNT * 4, c->stream));
    k_build_remap<<<cdiv((uint64_t)n_old, 256), 256, 0, c->stream>>>(c->old_to_new.as<int32_t>(), c->parent_old.as<uint32_t>(),
                                                                     (uint32_t)n_old, c->remap.as<uint32_t>());
    {
        Prof p(c, SAGE_K_MATCH);       // what replaces the matcher on this step
        k_remap_matches<<<(uint32_t)std::min<uint64_t>(((uint64_t)c->nbytes + 255) / 256, (uint64_t)c->n_cu * 32u), 256, 0, c->stream>>>(
            c->M.as<uint32_t>(), (uint64_t)c->nbytes, c->remap.as<uint32_t>(), (uint32_t)n_old, c->flags.as<uint32_t>());
    }
    HIP_TRY(c, hipGetLastError());
    uint32_t flags[FLAG_COUNT];
    HIP_TRY(c, hipMemcpyAsync(flags, c->flags.p, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));     // (old_to_new is the caller's: the copy above has completed)
    collect_events(c);
    // a match whose every prefix was pruned cannot exist (single bytes are never pruned): the table stays invalid if one did
    if (flags[FLAG_SET_OVERFLOW]) return SAGE_OK;
    c->m_valid = true;
    return SAGE_OK;
}

int sage_set_corpus(sage_ctx* c, const uint8_t* bytes, const int64_t* line_offsets, int64_t n_lines) {
    if (!c || !line_offsets || n_lines < 0) return fail(c, SAGE_E_ARG, "bad argument");
    if (line_offsets[0] != 0) return fail(c, SAGE_E_ARG, "line_offsets[0] must be 0");
    for (int64_t i = 0; i < n_lines; ++i)
        if (line_offsets[i + 1] < line_offsets[i]) return fail(c, SAGE_E_ARG, "line_offsets must be non-decreasing");
    const int64_t nbytes = line_offsets[n_lines];
    if (nbytes > 0 && !bytes) return fail(c, SAGE_E_ARG, "bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    const size_t padded = (size_t)nbytes + 1024;  // k_match stages up to 256+40 bytes past the last position
    if ((rc = ensure(c, c->corpus, padded))) return rc;
    if ((rc = ensure(c, c->rem, padded))) return rc;
    if ((rc = ensure(c, c->line_off, (size_t)(n_lines + 1) * 8))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemset(c->corpus.p, 0, padded));
    HIP_TRY(c, hipMemset(c->rem.p, 0, padded));
    if (nbytes) HIP_TRY(c, hipMemcpy(c->corpus.p, bytes, (size_t)nbytes, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->line_off.p, line_offsets, (size_t)(n_lines + 1) * 8, hipMemcpyHostToDevice));
    c->h_line_off.assign(line_offsets, line_offsets + n_lines + 1);
    c->max_line_bytes = 0;
    for (int64_t l = 0; l < n_lines; ++l) c->max_line_bytes = std::max(c->max_line_bytes, line_offsets[l + 1] - line_offsets[l]);
    c->n_lines = n_lines;
    c->nbytes = nbytes;
    c->m_valid = false;
    c->m_whole = nbytes > 0 && nbytes * 4 <= c->m_resident_max;
    c->n_lline = 0;
    if (nbytes) {
        k_line_remaining<<<(uint32_t)std::min<uint64_t>(((uint64_t)nbytes + 255) / 256, 1u << 20), 256, 0, c->stream>>>(
            c->line_off.as<int64_t>(), n_lines, c->rem.as<uint8_t>(), nbytes);
        HIP_TRY(c, hipGetLastError());
        if (c->long_walk) {      // the lines the walks take a wave each: counted, then listed
            if ((rc = ensure(c, c->lline_count, 16))) return rc;
            uint32_t n_long = 0;
            HIP_TRY(c, hipMemsetAsync(c->lline_count.p, 0, 16, c->stream));
            k_long_lines<<<cdiv((uint64_t)n_lines, 256), 256, 0, c->stream>>>(c->line_off.as<int64_t>(), n_lines, nullptr, c->lline_count.as<uint32_t>());
            HIP_TRY(c, hipMemcpyAsync(&n_long, c->lline_count.p, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (n_long) {
                if ((rc = ensure(c, c->lline_list, (size_t)n_long * 4))) return rc;
                HIP_TRY(c, hipMemsetAsync(c->lline_count.p, 0, 16, c->stream));
                k_long_lines<<<cdiv((uint64_t)n_lines, 256), 256, 0, c->stream>>>(c->line_off.as<int64_t>(), n_lines, c->lline_list.as<uint32_t>(),
                                                                                 c->lline_count.as<uint32_t>());
                HIP_TRY(c, hipGetLastError());
                c->n_lline = n_long;
            }
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->have_corpus = true;
    c->exact_buckets = false;
    c->count_fallback = false;
    return SAGE_OK;
}

int sage_set_embeddings(sage_ctx* c, const double* emb, int64_t rows, int32_t dim) {
    if (!c || !emb || rows <= 0 || dim <= 0) return fail(c, SAGE_E_ARG, "bad argument");
    if (dim > 512) return fail(c, SAGE_E_LIMIT, "embedding dimension above 512 is not supported");
    HIP_TRY(c, hipSetDevice(c->device));
    const int dp = padded_dim(dim);  // whole 128-byte lines per lane group
    if (rows >= (1ll << 32)) return fail(c, SAGE_E_LIMIT, "too many embedding rows");
    int rc;
    const bool want32 = c->score_mode != 1;
    const size_t bm_words = ((size_t)rows + 31) / 32 + 1;
    if ((rc = ensure(c, c->emb, (size_t)rows * dp * 8))) return rc;
    if ((rc = ensure(c, c->emb_stage, (size_t)rows * dim * 8))) return rc;
    if ((rc = ensure(c, c->emb_cnt, 16))) return rc;
    if (want32) {
        if ((rc = ensure(c, c->emb32, (size_t)rows * dp * 4))) return rc;
        if ((rc = ensure(c, c->inexact, bm_words * 4))) return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // the matrix goes up as it is (one contiguous copy); padding, the float32 copy and the bitmap are made on the device
    HIP_TRY(c, hipMemcpyAsync(c->emb_stage.p, emb, (size_t)rows * dim * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->emb_cnt.p, 0, 16, c->stream));
    if (want32) HIP_TRY(c, hipMemsetAsync(c->inexact.p, 0, bm_words * 4, c->stream));
    k_embed_layout<<<cdiv((uint64_t)rows, 4), 256, 0, c->stream>>>(
        c->emb_stage.as<double>(), (uint32_t)rows, (uint32_t)dim, (uint32_t)dp, c->emb.as<double>(),
        want32 ? c->emb32.as<float>() : nullptr, want32 ? c->inexact.as<uint32_t>() : nullptr,
        c->emb_cnt.as<unsigned long long>());
    HIP_TRY(c, hipGetLastError());
    unsigned long long n_inexact = 0;
    HIP_TRY(c, hipMemcpyAsync(&n_inexact, c->emb_cnt.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->emb_rows = rows;
    c->dim = dim;
    c->dp = dp;
    c->n_inexact = (int64_t)n_inexact;
    c->use_f32 = want32 && ((c->score_mode == 0 && c->n_inexact * 2 <= rows) || (c->score_mode == 2 && c->n_inexact < rows));
    c->have_emb = true;
    return SAGE_OK;
}

// tokens of the lines [l0, l1) as the compact arrays of a step (tok_w0, line token offsets): matcher + base walk
static int w2v_tokenize_batch(sage_ctx* c, int64_t l0, int64_t l1, uint32_t* n_tok_out) {
    int rc;
    if ((rc = match_and_count(c, l0, l1, nullptr))) return rc;
    uint64_t gt[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(gt, c->grand_total.p, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (gt[0] >= 0xFFFFFF00ull) return fail(c, SAGE_E_LIMIT, "a batch must stay under 2^32 tokens");
    const uint32_t n_tok = (uint32_t)gt[0], n_occ = (uint32_t)gt[1], nl = (uint32_t)(l1 - l0);
    *n_tok_out = n_tok;
    if (n_tok == 0) return SAGE_OK;
    if ((rc = ensure(c, c->tok_w0, (size_t)n_tok * 4))) return rc;
    if ((rc = ensure(c, c->occ_pos, (size_t)std::max(n_occ, 1u) * 4))) return rc;
    if ((rc = ensure(c, c->occ_tok, (size_t)std::max(n_occ, 1u) * 4))) return rc;
    k_walk_write<false><<<cdiv(nl, 256) + long_blocks(c->n_lline), 256, 0, c->stream>>>(
        c->line_off.as<int64_t>(), l0, nl, c->h_line_off[l0], batch_M(c, c->h_line_off[l0]), c->tokoff.as<uint32_t>(),
        c->occoff.as<uint32_t>(), c->tok_w0.as<uint32_t>(), c->occ_pos.as<uint32_t>(), c->occ_tok.as<uint32_t>(),
        nullptr, nullptr, nullptr, c->lline_list.as<uint32_t>(), c->n_lline);
    HIP_TRY(c, hipGetLastError());
    return SAGE_OK;
}

int sage_train_embeddings(sage_ctx* c, int32_t dim, int32_t window, int32_t negative, double alpha, double min_alpha,
                          int32_t epochs, double sample, int32_t min_count, uint64_t seed, int64_t vsize,
                          double* out_vectors, int64_t* out_counts) {
    int rc;
    if ((rc = check_ready(c, false))) return rc;
    if (!out_vectors || !out_counts || dim <= 0 || window < 1 || window > 31 || negative < 1 || negative > 31 || epochs < 1 ||
        !(alpha > 0) || min_alpha < 0 || sample < 0 || min_count < 1)
        return fail(c, SAGE_E_ARG, "bad argument (1 <= window, negative <= 31)");
    if (dim > 512) return fail(c, SAGE_E_LIMIT, "embedding dimension above 512 is not supported");
    if (vsize <= c->max_id) return fail(c, SAGE_E_ARG, "vsize must exceed the largest token id");
    HIP_TRY(c, hipSetDevice(c->device));
    const int dp = padded_dim(dim);
    const size_t V = (size_t)vsize;

    // batches of lines
    std::vector<std::pair<int64_t, int64_t>> batches;
    for (int64_t l0 = 0; l0 < c->n_lines;) {
        int64_t l1 = l0 + 1;
        while (l1 < c->n_lines && c->h_line_off[l1 + 1] - c->h_line_off[l0] <= c->batch_bytes) ++l1;
        batches.emplace_back(l0, l1);
        l0 = l1;
    }
    // ---- pass 1: token counts (the vocabulary scan of word2vec)
    if ((rc = ensure(c, c->w2v_counts, V * 8))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->w2v_counts.p, 0, V * 8, c->stream));
    uint64_t total_tokens = 0;
    for (auto& b : batches) {
        uint32_t n_tok = 0;
        if ((rc = w2v_tokenize_batch(c, b.first, b.second, &n_tok))) return rc;
        total_tokens += n_tok;
        if (n_tok)
            k_token_hist<<<std::min<uint32_t>(cdiv(n_tok, 256), 256u * 8u), 256, 0, c->stream>>>(
                c->tok_w0.as<uint32_t>(), n_tok, c->w2v_counts.as<unsigned long long>());
    }
    std::vector<uint64_t> counts(V);
    HIP_TRY(c, hipMemcpyAsync(counts.data(), c->w2v_counts.p, V * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < V; ++i) out_counts[i] = (int64_t)counts[i];

    // ---- sampling tables
    double retained = 0;
    for (size_t i = 0; i < V; ++i) if (counts[i] >= (uint64_t)min_count) retained += (double)counts[i];
    std::vector<uint32_t> keep(V, 0);
    std::vector<uint8_t> dropped(V, 1);
    std::vector<double> w(V, 0.0);
    double wsum = 0;
    const double thr = sample * retained;   // gensim: threshold_count = sample * retain_total
    for (size_t i = 0; i < V; ++i) {
        if (counts[i] < (uint64_t)min_count || counts[i] == 0) continue;
        dropped[i] = 0;
        const double v = (double)counts[i];
        double p = (sample > 0 && thr > 0) ? (std::sqrt(v / thr) + 1.0) * (thr / v) : 1.0;
        if (p > 1.0) p = 1.0;
        keep[i] = (uint32_t)std::min(4294967295.0, std::floor(p * 4294967296.0));
        w[i] = std::pow(v, 0.75);           // ns_exponent
        wsum += w[i];
    }
    // alias table (Vose) of w / wsum over all V slots (zero-weight slots always redirect)
    std::vector<float> aprob(V, 0.0f);
    std::vector<uint32_t> aidx(V, 0);
    {
        std::vector<double> q(V);
        std::vector<uint32_t> small, large;
        for (size_t i = 0; i < V; ++i) {
            q[i] = wsum > 0 ? w[i] / wsum * (double)V : 1.0;
            (q[i] < 1.0 ? small : large).push_back((uint32_t)i);
        }
        while (!small.empty() && !large.empty()) {
            const uint32_t s_ = small.back(), l_ = large.back();
            small.pop_back();
            aprob[s_] = (float)q[s_]; aidx[s_] = l_;
            q[l_] = (q[l_] + q[s_]) - 1.0;
            if (q[l_] < 1.0) { large.pop_back(); small.push_back(l_); }
        }
        for (uint32_t i : large) { aprob[i] = 1.0f; aidx[i] = i; }
        for (uint32_t i : small) { aprob[i] = 1.0f; aidx[i] = i; }   // numerical leftovers
    }
    // ---- initial vectors: syn0 ~ U(-1,1)/dim (gensim 4 prep_vectors), syn1neg = 0
    std::vector<float> syn0(V * (size_t)dp, 0.0f);
    {
        uint64_t st = seed ^ 0x5851F42D4C957F2Dull;
        for (size_t r = 0; r < V; ++r)
            for (int d = 0; d < dim; ++d) {
                st += 0x9E3779B97F4A7C15ull;
                const uint64_t z = sage_mix64(st);
                syn0[r * dp + d] = (float)((((double)(z >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0) / dim);
            }
    }
    if ((rc = ensure(c, c->w2v_keep, V * 4))) return rc;
    if ((rc = ensure(c, c->w2v_dropped, V))) return rc;
    if ((rc = ensure(c, c->w2v_aprob, V * 4))) return rc;
    if ((rc = ensure(c, c->w2v_aidx, V * 4))) return rc;
    if ((rc = ensure(c, c->w2v_syn0, syn0.size() * 4))) return rc;
    if ((rc = ensure(c, c->w2v_syn1, syn0.size() * 4))) return rc;
    HIP_TRY(c, hipMemcpy(c->w2v_keep.p, keep.data(), V * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->w2v_dropped.p, dropped.data(), V, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->w2v_aprob.p, aprob.data(), V * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->w2v_aidx.p, aidx.data(), V * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->w2v_syn0.p, syn0.data(), syn0.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(c->w2v_syn1.p, 0, syn0.size() * 4));

    // ---- training
    for (int32_t ep = 0; ep < epochs && total_tokens > 0; ++ep) {
        uint64_t before = 0;
        for (auto& b : batches) {
            uint32_t n_tok = 0;
            if ((rc = w2v_tokenize_batch(c, b.first, b.second, &n_tok))) return rc;
            if (!n_tok) continue;
            W2VArgs a{c->tok_w0.as<uint32_t>(), c->tokoff.as<uint32_t>(), (uint32_t)(b.second - b.first), before, total_tokens,
                      c->w2v_keep.as<uint32_t>(), c->w2v_dropped.as<uint8_t>(), c->w2v_aprob.as<float>(), c->w2v_aidx.as<uint32_t>(),
                      (uint32_t)V, c->w2v_syn0.as<float>(), c->w2v_syn1.as<float>(), (uint32_t)dp, (uint32_t)window,
                      (uint32_t)negative, (uint32_t)ep, (uint32_t)epochs, (float)alpha, (float)min_alpha, seed};
            const uint32_t grid = std::min<uint32_t>(cdiv(a.n_lines, 256 / W2V_G), (uint32_t)c->n_cu * 8u);
            switch (dp / W2V_G) {
            case 2: k_w2v_train<2><<<grid, 256, 0, c->stream>>>(a); break;
            case 4: k_w2v_train<4><<<grid, 256, 0, c->stream>>>(a); break;
            case 8: k_w2v_train<8><<<grid, 256, 0, c->stream>>>(a); break;
            case 16: k_w2v_train<16><<<grid, 256, 0, c->stream>>>(a); break;
            case 24: k_w2v_train<24><<<grid, 256, 0, c->stream>>>(a); break;
            default: k_w2v_train<32><<<grid, 256, 0, c->stream>>>(a); break;
            }
            HIP_TRY(c, hipGetLastError());
            before += n_tok;
        }
    }
    HIP_TRY(c, hipMemcpyAsync(syn0.data(), c->w2v_syn0.p, syn0.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t r = 0; r < V; ++r)
        for (int d = 0; d < dim; ++d) out_vectors[r * dim + d] = (double)syn0[r * dp + d];
    return SAGE_OK;
}

int sage_tokenize(sage_ctx* c, int64_t line_begin, int64_t line_end, int32_t* out_ids, int32_t* out_starts,
                  int32_t* out_widths, int64_t capacity, int64_t* out_counts, int64_t* out_total) {
    int rc;
    if ((rc = check_ready(c, false))) return rc;
    if (line_begin < 0 || line_end < line_begin || line_end > c->n_lines) return fail(c, SAGE_E_ARG, "line range outside the corpus");
    const bool want_tokens = out_ids || out_starts || out_widths;
    if (want_tokens && !(out_ids && out_starts && out_widths)) return fail(c, SAGE_E_ARG, "pass all three token arrays or none");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(c->flags.p, 0, FLAG_COUNT * 4, c->stream));
    int64_t total = 0;
    int64_t l0 = line_begin;
    while (l0 < line_end) {
        int64_t l1 = l0 + 1;
        while (l1 < line_end && c->h_line_off[l1 + 1] - c->h_line_off[l0] <= c->batch_bytes) ++l1;
        uint64_t ntok = 0;
        if ((rc = match_and_count(c, l0, l1, &ntok))) return rc;
        const uint32_t nl = (uint32_t)(l1 - l0);
        if (out_counts) {
            std::vector<uint32_t> cnt(nl);
            HIP_TRY(c, hipMemcpyAsync(cnt.data(), c->ntok.p, (size_t)nl * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (uint32_t i = 0; i < nl; ++i) out_counts[(l0 - line_begin) + i] = cnt[i];
        }
        if (want_tokens && ntok) {
            if (total + (int64_t)ntok > capacity) return fail(c, SAGE_E_ARG, "token capacity too small");
            if ((rc = ensure(c, c->ex_ids, ntok * 4))) return rc;
            if ((rc = ensure(c, c->ex_starts, ntok * 4))) return rc;
            if ((rc = ensure(c, c->ex_widths, ntok * 4))) return rc;
            {
                Prof p(c, SAGE_K_WALK);
                k_walk_write<true><<<cdiv(nl, 256) + long_blocks(c->n_lline), 256, 0, c->stream>>>(
                    c->line_off.as<int64_t>(), l0, nl, c->h_line_off[l0], batch_M(c, c->h_line_off[l0]), c->tokoff.as<uint32_t>(),
                    nullptr, nullptr, nullptr, nullptr, c->ex_ids.as<int32_t>(), c->ex_starts.as<int32_t>(),
                    c->ex_widths.as<int32_t>(), c->lline_list.as<uint32_t>(), c->n_lline);
            }
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(out_ids + total, c->ex_ids.p, ntok * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(out_starts + total, c->ex_starts.p, ntok * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(out_widths + total, c->ex_widths.p, ntok * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        total += (int64_t)ntok;
        l0 = l1;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    collect_events(c);
    if (out_total) *out_total = total;
    return SAGE_OK;
}

int sage_tokenize_noalloc(sage_ctx* c, const uint8_t* d_text, int64_t n_bytes, const int64_t* d_line_offsets,
                          int64_t n_lines, int32_t* d_ids, int32_t* d_starts, int32_t* d_widths, int64_t capacity,
                          int64_t* d_counts, int64_t* out_total) {
    if (!c) return SAGE_E_ARG;
    if (!c->have_vocab) return fail(c, SAGE_E_STATE, "sage_set_vocab has not been called");
    if (!d_line_offsets || n_lines < 0 || n_bytes < 0 || (n_bytes > 0 && !d_text) || capacity < 0)
        return fail(c, SAGE_E_ARG, "bad argument");
    const bool want_tokens = d_ids || d_starts || d_widths;
    if (want_tokens && !(d_ids && d_starts && d_widths)) return fail(c, SAGE_E_ARG, "pass all three token arrays or none");
    if (n_bytes > SAGE_NOALLOC_MAX_BYTES)
        return fail(c, SAGE_E_LIMIT, "sage_tokenize_noalloc takes at most SAGE_NOALLOC_MAX_BYTES (1 MiB) of text per call");
    if (n_lines > SAGE_NOALLOC_MAX_LINES)
        return fail(c, SAGE_E_LIMIT, "sage_tokenize_noalloc takes at most SAGE_NOALLOC_MAX_LINES (65,536) lines per call");
    if (!c->na_scratch) return fail(c, SAGE_E_STATE, "the scratch array was not resolved for this device");
    HIP_TRY(c, hipSetDevice(c->device));
    if (out_total) *out_total = 0;
    if (n_lines == 0) return SAGE_OK;
    // the kernels trust the line offsets: check them on the host first (host memory only; <= 512 KiB)
    std::vector<int64_t> off((size_t)n_lines + 1);
    HIP_TRY(c, hipMemcpyAsync(off.data(), d_line_offsets, off.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (off[0] != 0 || off[(size_t)n_lines] != n_bytes) return fail(c, SAGE_E_ARG, "line offsets must run from 0 to n_bytes");
    for (int64_t i = 0; i < n_lines; ++i)
        if (off[(size_t)i + 1] < off[(size_t)i]) return fail(c, SAGE_E_ARG, "line_offsets must be non-decreasing");

    unsigned char* S = c->na_scratch;          // never the array's name: see the note at its definition
    uint8_t* text = S + NA_TEXT;
    uint8_t* rem = S + NA_REM;
    const MatchBufs B{text, rem, d_line_offsets, reinterpret_cast<uint32_t*>(S + NA_M),
                      reinterpret_cast<uint32_t*>(S + NA_NTOK), reinterpret_cast<uint32_t*>(S + NA_NOCC),
                      reinterpret_cast<uint32_t*>(S + NA_TOKOFF), reinterpret_cast<uint32_t*>(S + NA_OCCOFF),
                      reinterpret_cast<uint32_t*>(S + NA_TILES), reinterpret_cast<uint64_t*>(S + NA_TOTAL),
                      reinterpret_cast<uint32_t*>(S + NA_FLAGS), nullptr, 0u};
    const uint32_t nl = (uint32_t)n_lines;
    // the array keeps the last call's contents: what must start at zero is cleared here (the text's pad, which the
    // matcher's 64-byte tiles read past the last byte, the totals and the flags); everything else is written before
    // it is read.  The text is copied in because the caller's buffer has no such pad.
    if (n_bytes) HIP_TRY(c, hipMemcpyAsync(text, d_text, (size_t)n_bytes, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(text + n_bytes, 0, 1024, c->stream));
    HIP_TRY(c, hipMemsetAsync(S + NA_TOTAL, 0, 16, c->stream));
    HIP_TRY(c, hipMemsetAsync(S + NA_FLAGS, 0, FLAG_COUNT * 4, c->stream));
    if (n_bytes)
        k_line_remaining<<<cdiv((uint64_t)n_bytes, 256), 256, 0, c->stream>>>(d_line_offsets, n_lines, rem, n_bytes);   // <= 1 MiB
    launch_match_and_count(c, B, 0, nl, 0, (uint64_t)n_bytes);
    HIP_TRY(c, hipGetLastError());
    uint64_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, B.grand_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (d_counts) k_widen_counts<<<cdiv(nl, 256), 256, 0, c->stream>>>(B.ntok, nl, d_counts);
    if (want_tokens && total) {
        if ((int64_t)total > capacity) {
            (void)hipStreamSynchronize(c->stream);     // k_widen_counts may still be writing the caller's d_counts
            collect_events(c);
            return fail(c, SAGE_E_ARG, "token capacity too small");
        }
        k_walk_write<true><<<cdiv(nl, 256), 256, 0, c->stream>>>(d_line_offsets, 0, nl, 0, B.M, B.tokoff, nullptr, nullptr,
                                                                nullptr, nullptr, d_ids, d_starts, d_widths, nullptr, 0u);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    collect_events(c);      // with profiling on, the events of this call go back to the pool (no event is created in the steady state)
    if (out_total) *out_total = (int64_t)total;
    return SAGE_OK;
}

int sage_ablation_step(sage_ctx* c, const int64_t* bounds, int64_t n_seg, int32_t pad, double* out_loss,
                       int64_t* out_occ, int64_t vsize, int64_t* out_total_tokens, int64_t* out_total_triples,
                       int64_t* out_hist, int64_t hist_len) {
    if (!c) return SAGE_E_ARG;
    if (!out_loss || !out_occ || !out_hist || !out_total_tokens || !out_total_triples) return fail(c, SAGE_E_ARG, "null output");
    int64_t totals[2] = {0, 0};
    int rc = step_impl(c, bounds, n_seg, pad, vsize, hist_len, totals);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpy(out_loss, c->loss.p, (size_t)vsize * 8, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(out_occ, c->occ.p, (size_t)vsize * 8, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(out_hist, c->hist.p, (size_t)hist_len * 8, hipMemcpyDeviceToHost));
    *out_total_tokens = totals[0];
    *out_total_triples = totals[1];
    return SAGE_OK;
}

int sage_ablation_step_device(sage_ctx* c, const int64_t* bounds, int64_t n_seg, int32_t pad, double* d_loss,
                              int64_t* d_occ, int64_t vsize, int64_t* d_totals, int64_t* d_hist, int64_t hist_len) {
    if (!c) return SAGE_E_ARG;
    if (!d_loss || !d_occ || !d_hist || !d_totals) return fail(c, SAGE_E_ARG, "null output");
    int64_t totals[2] = {0, 0};
    int rc = step_impl(c, bounds, n_seg, pad, vsize, hist_len, totals);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(d_loss, c->loss.p, (size_t)vsize * 8, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_occ, c->occ.p, (size_t)vsize * 8, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_hist, c->hist.p, (size_t)hist_len * 8, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_totals, totals, 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SAGE_OK;
}

int sage_set_profiling(sage_ctx* c, int enabled) {
    if (!c) return SAGE_E_ARG;
    c->profiling = enabled != 0;
    return SAGE_OK;
}

int sage_reset_kernel_times(sage_ctx* c) {
    if (!c) return SAGE_E_ARG;
    for (int i = 0; i < SAGE_K_NCLASSES; ++i) { c->k_ms[i] = 0; c->k_launches[i] = 0; }
    return SAGE_OK;
}

int sage_get_kernel_times(sage_ctx* c, double* out_ms, int64_t* out_launches) {
    if (!c || !out_ms || !out_launches) return SAGE_E_ARG;
    for (int i = 0; i < SAGE_K_NCLASSES; ++i) { out_ms[i] = c->k_ms[i]; out_launches[i] = c->k_launches[i]; }
    return SAGE_OK;
}

int sage_set_batch_bytes(sage_ctx* c, int64_t bytes) {
    if (!c || bytes < 4096) return fail(c, SAGE_E_ARG, "batch_bytes must be at least 4096");
    c->batch_bytes = std::min<int64_t>(bytes, 0xF0000000ll);
    return SAGE_OK;
}

int sage_set_match_mode(sage_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 2) return fail(c, SAGE_E_ARG, "match mode must be 0, 1 or 2");
    c->match_mode = mode;
    c->m_valid = false;      // the next step matches again, with the matcher that was asked for
    return SAGE_OK;
}

int sage_invalidate_matches(sage_ctx* c) {
    if (!c) return SAGE_E_ARG;
    c->m_valid = false;
    return SAGE_OK;
}

int sage_set_score_mode(sage_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 2) return fail(c, SAGE_E_ARG, "score mode must be 0, 1 or 2");
    c->score_mode = mode;
    c->have_emb = false;   // takes effect at the next sage_set_embeddings
    return SAGE_OK;
}

int sage_set_count_mode(sage_ctx* c, int mode) {
    if (!c || (mode != 0 && mode != 1)) return fail(c, SAGE_E_ARG, "count mode must be 0 or 1");
    c->count_mode = mode;
    return SAGE_OK;
}

int sage_set_deterministic(sage_ctx* c, int enabled) {
    if (!c) return SAGE_E_ARG;
    c->deterministic = enabled != 0;
    return SAGE_OK;
}

int sage_debug_scan_total(sage_ctx* c, uint32_t value, int64_t n, uint64_t* out_total, uint32_t* out_last) {
    if (!c || !out_total || !out_last || n <= 0 || n > (int64_t)1 << 30) return fail(c, SAGE_E_ARG, "bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    // the step's own scan (scan_u32) over n copies of `value`, in the buffers the step scans its pair counts in
    if ((rc = ensure(c, c->np_all, (size_t)n * 4))) return rc;
    if ((rc = ensure(c, c->pair_off, ((size_t)n + 1) * 4))) return rc;
    if ((rc = ensure(c, c->grand_total, 16))) return rc;
    HIP_TRY(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->np_all.p), (int)value, (size_t)n, c->stream));
    if ((rc = scan_u32(c, c->np_all.as<uint32_t>(), (uint32_t)n, c->pair_off.as<uint32_t>(), c->grand_total.as<uint64_t>()))) return rc;
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out_total, c->grand_total.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_last, c->pair_off.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    collect_events(c);
    return SAGE_OK;
}

int sage_measure_unique_rows(sage_ctx* c, int32_t window, int64_t* out) {
    if (!c || !out || window < 1 || window > 4096) return fail(c, SAGE_E_ARG, "bad argument");
    if (c->last_n_occ == 0 || c->last_batches < 1) return fail(c, SAGE_E_STATE, "no step has run");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->tile_dbg, 64))) return rc;
    unsigned long long h[4] = {0, 0, 0, 0};
    for (int which = 0; which < 2; ++which) {      // 0: the exact list (or every occurrence), 1: the inexact list
        if (which == 1 && !c->last_split) break;
        HIP_TRY(c, hipMemsetAsync(c->tile_dbg.p, 0, 16, c->stream));
        const uint32_t* list = !c->last_split ? nullptr : which == 0 ? c->exact_list.as<uint32_t>() : c->inex_list.as<uint32_t>();
        const uint32_t* cnt = !c->last_split ? nullptr : which == 0 ? c->exact_count.as<uint32_t>() : c->inex_count.as<uint32_t>();
        k_unique_rows<<<(uint32_t)c->n_cu * 4u, 256, 0, c->stream>>>(c->desc.as<uint4>(), c->pool.as<uint32_t>(), list, cnt, c->last_n_occ,
                                                                    (uint32_t)window, c->tile_dbg.as<unsigned long long>());
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(h + 2 * which, c->tile_dbg.p, 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    out[0] = (int64_t)h[0]; out[1] = (int64_t)h[1]; out[2] = (int64_t)h[2]; out[3] = (int64_t)h[3];
    out[4] = (int64_t)c->last_n_occ; out[5] = c->last_batches;
    return SAGE_OK;
}

int sage_get_step_stats(sage_ctx* c, int64_t* out) {
    if (!c || !out) return SAGE_E_ARG;
    for (int i = 0; i < SAGE_STEP_STATS; ++i) out[i] = c->step_stats[i];
    return SAGE_OK;
}

}  // extern "C"
