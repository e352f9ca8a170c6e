"""ctypes binding of libsage_hip.so (C-ABI: include/sage_hip.h).  The library is built in-tree by
`__graft_entry__.build()` / `sage_amd/csrc/Makefile`; there is NO CPU fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
# SAGE_LIB=<path> loads another build of the same library (development: A/B runs of a variant under tools/abl/)
LIB_PATH = Path(os.environ["SAGE_LIB"]) if os.environ.get("SAGE_LIB") else _HERE / "libsage_hip.so"

K_CLASSES = ("match", "walk", "scan", "count", "score", "misc", "describe", "pairs", "logsig", "fix", "partition")

_ERRORS = {-1: ValueError, -2: Exception, -3: RuntimeError, -4: RuntimeError, -5: ValueError, -6: IndexError,
           -7: ValueError}

_lib = None


class SageHipError(RuntimeError):
    pass


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype)) if arr is not None else None


def load():
    """Loads libsage_hip.so and declares every entry point of include/sage_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise SageHipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C sage_amd/csrc` (there is no CPU fallback for the ablation hot path)")
    lib = C.CDLL(str(LIB_PATH))
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    pu8, pi32, pi64, pf64 = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
    lib.sage_version.restype = C.c_char_p
    lib.sage_version.argtypes = []
    lib.sage_create.restype = C.c_int
    lib.sage_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.sage_destroy.restype = None
    lib.sage_destroy.argtypes = [vp]
    lib.sage_last_error.restype = C.c_char_p
    lib.sage_last_error.argtypes = [vp]
    lib.sage_set_vocab.restype = C.c_int
    lib.sage_set_vocab.argtypes = [vp, pu8, pi64, pi32, i64, i32]
    lib.sage_set_vocab_pruned.restype = C.c_int
    lib.sage_set_vocab_pruned.argtypes = [vp, pu8, pi64, pi32, i64, i32, pi32, i64]
    lib.sage_set_corpus.restype = C.c_int
    lib.sage_set_corpus.argtypes = [vp, pu8, pi64, i64]
    lib.sage_set_embeddings.restype = C.c_int
    lib.sage_set_embeddings.argtypes = [vp, pf64, i64, i32]
    lib.sage_tokenize.restype = C.c_int
    lib.sage_tokenize.argtypes = [vp, i64, i64, pi32, pi32, pi32, i64, pi64, pi64]
    lib.sage_tokenize_noalloc.restype = C.c_int
    lib.sage_tokenize_noalloc.argtypes = [vp, vp, i64, vp, i64, vp, vp, vp, i64, vp, pi64]
    lib.sage_ablation_step.restype = C.c_int
    lib.sage_ablation_step.argtypes = [vp, pi64, i64, i32, pf64, pi64, i64, pi64, pi64, pi64, i64]
    lib.sage_ablation_step_device.restype = C.c_int
    lib.sage_ablation_step_device.argtypes = [vp, pi64, i64, i32, vp, vp, i64, vp, vp, i64]
    lib.sage_set_profiling.restype = C.c_int
    lib.sage_set_profiling.argtypes = [vp, C.c_int]
    lib.sage_reset_kernel_times.restype = C.c_int
    lib.sage_reset_kernel_times.argtypes = [vp]
    lib.sage_get_kernel_times.restype = C.c_int
    lib.sage_get_kernel_times.argtypes = [vp, pf64, pi64]
    lib.sage_set_batch_bytes.restype = C.c_int
    lib.sage_set_batch_bytes.argtypes = [vp, i64]
    lib.sage_set_match_mode.restype = C.c_int
    lib.sage_set_match_mode.argtypes = [vp, C.c_int]
    lib.sage_invalidate_matches.restype = C.c_int
    lib.sage_invalidate_matches.argtypes = [vp]
    lib.sage_set_score_mode.restype = C.c_int
    lib.sage_set_score_mode.argtypes = [vp, C.c_int]
    lib.sage_set_count_mode.restype = C.c_int
    lib.sage_set_count_mode.argtypes = [vp, C.c_int]
    lib.sage_set_deterministic.restype = C.c_int
    lib.sage_set_deterministic.argtypes = [vp, C.c_int]
    lib.sage_train_embeddings.restype = C.c_int
    lib.sage_train_embeddings.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_double,
                                          C.c_int32, C.c_uint64, C.c_int64, C.POINTER(C.c_double), pi64]
    lib.sage_get_step_stats.restype = C.c_int
    lib.sage_get_step_stats.argtypes = [vp, pi64]
    lib.sage_measure_unique_rows.restype = C.c_int
    lib.sage_measure_unique_rows.argtypes = [vp, C.c_int32, pi64]
    lib.sage_debug_scan_total.restype = C.c_int
    lib.sage_debug_scan_total.argtypes = [vp, C.c_uint32, i64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    _lib = lib
    return lib


EXPORTED = ["sage_version", "sage_create", "sage_destroy", "sage_last_error", "sage_set_vocab", "sage_set_vocab_pruned", "sage_set_corpus",
            "sage_set_embeddings", "sage_tokenize", "sage_tokenize_noalloc", "sage_ablation_step", "sage_ablation_step_device",
            "sage_set_profiling", "sage_reset_kernel_times", "sage_get_kernel_times", "sage_set_batch_bytes",
            "sage_set_match_mode", "sage_invalidate_matches", "sage_set_count_mode", "sage_set_score_mode", "sage_set_deterministic", "sage_train_embeddings", "sage_get_step_stats", "sage_debug_scan_total", "sage_measure_unique_rows"]


class HipSage:
    """One context on one GPU.  Thin, allocation-free-in-steady-state wrapper over the C-ABI."""

    HIST_LEN = 4096

    def __init__(self, device: int = 0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.sage_create(int(device), C.byref(h))
        if rc != 0:
            raise SageHipError(f"sage_create(device={device}) failed with code {rc}: no usable MI355X / HIP device")
        self._h = h
        self.device = int(device)
        self.vsize = 0
        self.hist_len = self.HIST_LEN   # bins of the ablated_sizes histogram; grown by the callers on overflow
        self.cache = {}                 # what is resident (sage_amd.parallel keeps (object, signature) pairs here)

    def invalidate(self):
        """Forget what is resident: the next run_sage_parallel uploads vocabulary, corpus and embeddings again."""
        self.cache.clear()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sage_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == 0:
            return
        msg = self._lib.sage_last_error(self._h).decode("utf-8", "replace")
        raise _ERRORS.get(rc, SageHipError)(msg)

    # ---- state ------------------------------------------------------------------------------------------
    def set_vocab(self, keys, ids=None, max_len: int = 16):
        """keys: sequence of distinct bytes; ids: their ids (default: position)."""
        flat = np.frombuffer(b"".join(keys), dtype=np.uint8)
        if flat.size == 0:
            flat = np.zeros(1, dtype=np.uint8)
        off = np.zeros(len(keys) + 1, dtype=np.int64)
        np.cumsum(np.fromiter(map(len, keys), dtype=np.int64, count=len(keys)), out=off[1:])
        ids_arr = None if ids is None else np.ascontiguousarray(ids, dtype=np.int32)
        self._check(self._lib.sage_set_vocab(self._h, _ptr(flat, C.c_uint8), _ptr(off, C.c_int64),
                                             _ptr(ids_arr, C.c_int32), len(keys), int(max_len)))
        self.vsize = (int(ids_arr.max()) if ids_arr is not None else len(keys) - 1) + 1

    def set_vocab_packed(self, flat: np.ndarray, off: np.ndarray, ids, max_len: int = 16):
        """The same from arrays that are already packed (SaGeTokenizer.packed_key_table)."""
        n = len(off) - 1
        buf = flat if flat.size else np.zeros(1, dtype=np.uint8)
        ids_arr = None if ids is None else np.ascontiguousarray(ids, dtype=np.int32)
        self._check(self._lib.sage_set_vocab(self._h, _ptr(buf, C.c_uint8), _ptr(off, C.c_int64),
                                             _ptr(ids_arr, C.c_int32), n, int(max_len)))
        self.vsize = (int(ids_arr.max()) if ids_arr is not None else n - 1) + 1

    def set_vocab_pruned(self, flat: np.ndarray, off: np.ndarray, ids, max_len: int, old_to_new: np.ndarray):
        """sage_set_vocab_pruned: the new (packed) vocabulary only lost keys; old_to_new[old id] = new id or -1.  A valid
        resident match table is carried over instead of recomputed."""
        n = len(off) - 1
        buf = flat if flat.size else np.zeros(1, dtype=np.uint8)
        ids_arr = None if ids is None else np.ascontiguousarray(ids, dtype=np.int32)
        o2n = np.ascontiguousarray(old_to_new, dtype=np.int32)
        self._check(self._lib.sage_set_vocab_pruned(self._h, _ptr(buf, C.c_uint8), _ptr(off, C.c_int64),
                                                    _ptr(ids_arr, C.c_int32), n, int(max_len), _ptr(o2n, C.c_int32), len(o2n)))
        self.vsize = (int(ids_arr.max()) if ids_arr is not None else n - 1) + 1

    def set_corpus(self, corpus: np.ndarray, line_offsets: np.ndarray):
        corpus = np.ascontiguousarray(corpus, dtype=np.uint8)
        line_offsets = np.ascontiguousarray(line_offsets, dtype=np.int64)
        buf = corpus if corpus.size else np.zeros(1, dtype=np.uint8)
        self._check(self._lib.sage_set_corpus(self._h, _ptr(buf, C.c_uint8), _ptr(line_offsets, C.c_int64),
                                              len(line_offsets) - 1))
        self.n_lines = len(line_offsets) - 1
        self.cache.pop("corpus", None)

    def set_embeddings(self, emb: np.ndarray):
        emb = np.ascontiguousarray(emb, dtype=np.float64)
        if emb.ndim != 2:
            raise ValueError("embeddings must be [rows, dim]")
        self._check(self._lib.sage_set_embeddings(self._h, _ptr(emb, C.c_double), emb.shape[0], emb.shape[1]))
        self.cache.pop("emb", None)

    # ---- hot path ---------------------------------------------------------------------------------------
    def token_counts(self, line_begin: int, line_end: int):
        counts = np.zeros(max(0, line_end - line_begin), dtype=np.int64)
        total = C.c_int64(0)
        self._check(self._lib.sage_tokenize(self._h, line_begin, line_end, None, None, None, 0,
                                            _ptr(counts, C.c_int64), C.byref(total)))
        return counts, total.value

    def tokenize(self, line_begin: int, line_end: int):
        """-> (ids, starts, widths int32 arrays over all lines back to back, counts int64 per line)."""
        counts, total = self.token_counts(line_begin, line_end)
        ids = np.empty(max(total, 1), dtype=np.int32)
        starts = np.empty(max(total, 1), dtype=np.int32)
        widths = np.empty(max(total, 1), dtype=np.int32)
        tot2 = C.c_int64(0)
        self._check(self._lib.sage_tokenize(self._h, line_begin, line_end, _ptr(ids, C.c_int32), _ptr(starts, C.c_int32),
                                            _ptr(widths, C.c_int32), total, _ptr(counts, C.c_int64), C.byref(tot2)))
        return ids[:total], starts[:total], widths[:total], counts

    NOALLOC_MAX_BYTES = 1 << 20
    NOALLOC_MAX_LINES = 1 << 16

    def tokenize_noalloc(self, d_text: int, n_bytes: int, d_line_offsets: int, n_lines: int, d_ids: int = 0,
                         d_starts: int = 0, d_widths: int = 0, capacity: int = 0, d_counts: int = 0) -> int:
        """sage_tokenize_noalloc: device buffers given as raw addresses (e.g. torch.Tensor.data_ptr()); allocates no
        device memory.  Returns the number of tokens."""
        total = C.c_int64(0)
        p = lambda a: C.c_void_p(a) if a else None
        self._check(self._lib.sage_tokenize_noalloc(self._h, p(d_text), int(n_bytes), p(d_line_offsets), int(n_lines),
                                                    p(d_ids), p(d_starts), p(d_widths), int(capacity), p(d_counts),
                                                    C.byref(total)))
        return total.value

    def ablation_step(self, segment_line_bounds, pad: int = 2, vsize: int | None = None):
        """-> (loss f64[vsize], occ i64[vsize], total_tokens, total_triples, hist i64[self.hist_len])."""
        b = np.ascontiguousarray(segment_line_bounds, dtype=np.int64)
        vsize = int(vsize or self.vsize)
        loss = np.zeros(vsize, dtype=np.float64)
        occ = np.zeros(vsize, dtype=np.int64)
        hist = np.zeros(self.hist_len, dtype=np.int64)
        tt, ttr = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.sage_ablation_step(self._h, _ptr(b, C.c_int64), len(b) - 1, int(pad),
                                                 _ptr(loss, C.c_double), _ptr(occ, C.c_int64), vsize, C.byref(tt),
                                                 C.byref(ttr), _ptr(hist, C.c_int64), self.hist_len))
        return loss, occ, tt.value, ttr.value, hist

    def ablation_step_device(self, segment_line_bounds, d_loss, d_occ, d_totals, d_hist, pad: int = 2):
        """Results into caller-owned device buffers given as raw addresses (e.g. torch.Tensor.data_ptr())."""
        b = np.ascontiguousarray(segment_line_bounds, dtype=np.int64)
        loss_ptr, vsize = d_loss
        hist_ptr, hist_len = d_hist
        self._check(self._lib.sage_ablation_step_device(self._h, _ptr(b, C.c_int64), len(b) - 1, int(pad),
                                                        C.c_void_p(loss_ptr), C.c_void_p(d_occ), int(vsize),
                                                        C.c_void_p(d_totals), C.c_void_p(hist_ptr), int(hist_len)))

    # ---- measurement ------------------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        self._check(self._lib.sage_set_profiling(self._h, 1 if on else 0))

    def reset_kernel_times(self):
        self._check(self._lib.sage_reset_kernel_times(self._h))

    def kernel_times(self):
        ms = np.zeros(len(K_CLASSES), dtype=np.float64)
        n = np.zeros(len(K_CLASSES), dtype=np.int64)
        self._check(self._lib.sage_get_kernel_times(self._h, _ptr(ms, C.c_double), _ptr(n, C.c_int64)))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(K_CLASSES)}

    def set_batch_bytes(self, nbytes: int):
        self._check(self._lib.sage_set_batch_bytes(self._h, int(nbytes)))

    def set_match_mode(self, mode: int):
        self._check(self._lib.sage_set_match_mode(self._h, int(mode)))

    def invalidate_matches(self):
        """The next step runs the matcher again (a cold step), whatever match table is resident."""
        self._check(self._lib.sage_invalidate_matches(self._h))

    def set_count_mode(self, mode: int):
        self._check(self._lib.sage_set_count_mode(self._h, int(mode)))

    def set_score_mode(self, mode: int):
        """0 = float32 row storage where lossless (default), 1 = float64 storage; call before set_embeddings."""
        self._check(self._lib.sage_set_score_mode(self._h, int(mode)))
        self.cache.pop("emb", None)     # the library drops its matrix: the next step must upload it again

    def set_deterministic(self, on: bool):
        """Fixed-point loss sums: bit-identical losses from run to run and for any batch size (sage_set_deterministic)."""
        self._check(self._lib.sage_set_deterministic(self._h, 1 if on else 0))

    def train_embeddings(self, vsize: int, dim: int, window: int = 5, negative: int = 15, alpha: float = 0.025,
                         min_alpha: float = 1e-4, epochs: int = 5, sample: float = 1e-3, min_count: int = 1, seed: int = 1):
        """-> (vectors float64 [vsize, dim], counts int64 [vsize]); see sage_train_embeddings in include/sage_hip.h."""
        vec = np.zeros((int(vsize), int(dim)), dtype=np.float64)
        counts = np.zeros(int(vsize), dtype=np.int64)
        self._check(self._lib.sage_train_embeddings(self._h, int(dim), int(window), int(negative), float(alpha), float(min_alpha),
                                                    int(epochs), float(sample), int(min_count), int(seed) & (2 ** 64 - 1),
                                                    int(vsize), _ptr(vec, C.c_double), _ptr(counts, C.c_int64)))
        return vec, counts

    def debug_scan_total(self, value: int, n: int):
        """-> (64-bit total, last 32-bit offset) of the step's scan over n copies of `value` (sage_debug_scan_total)."""
        total, last = C.c_uint64(0), C.c_uint32(0)
        self._check(self._lib.sage_debug_scan_total(self._h, int(value), int(n), C.byref(total), C.byref(last)))
        return total.value, last.value

    def step_stats(self):
        out = np.zeros(24, dtype=np.int64)
        self._check(self._lib.sage_get_step_stats(self._h, _ptr(out, C.c_int64)))
        return {"tokens": int(out[0]), "corpus_bytes": int(out[1]), "occurrences": int(out[2]),
                "rows": int(out[3]), "records": int(out[4]), "pairs": int(out[5]), "emb_elem_bytes": int(out[6]),
                "f64_fix_occurrences": int(out[7]), "window_rows": int(out[8]), "long_occurrences": int(out[9]),
                "dim": int(out[10]), "padded_dim": int(out[11]), "window_pairs": int(out[12]),
                "count_redos": int(out[13]), "count_from_hbm_sets": int(out[14]), "exact_buckets": int(out[15]) & 1,
                "no_matcher": (int(out[15]) >> 1) & 1, "window_rows_inexact": int(out[16]), "f64_rows_gathered": int(out[17])}

    def measure_unique_rows(self, window: int = 32):
        """Distinct rows per `window` consecutive occurrences over the last batch of the last step (sage_measure_unique_rows)."""
        out = np.zeros(6, dtype=np.int64)
        self._check(self._lib.sage_measure_unique_rows(self._h, int(window), _ptr(out, C.c_int64)))
        return {"exact_unique_rows": int(out[0]), "exact_window_ids": int(out[1]), "inexact_unique_rows": int(out[2]),
                "inexact_window_ids": int(out[3]), "occurrences": int(out[4]), "batches": int(out[5])}
