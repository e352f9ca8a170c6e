"""run_sage_parallel: the reference's paralleliser (utils.py:106-113,141-257) with the process pool, the dict
bookkeeping and the numpy scoring replaced by one call into libsage_hip.so per GPU.

Signature, return shapes and conventions are the reference's:
    run_sage_parallel(embeddings, partial_corpus, sage_model, workers_number)
        -> (total_tokens, total_triples, sage_losses: Dict[int,float], ablated_sizes: Dict[int,int])
`workers_number` no longer sets the parallelism (the GPU does); it still defines the worker shards and, with
the 10,000-line flush rule, the line ranges over which distinct triples are counted -- so total_triples is
bit-identical to a reference run with that many workers.
"""
from __future__ import annotations

import logging
import os
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

from ._lib import HipSage

FLUSH_LINES = 10000  # utils.py:193 chunk_size

_engines: Dict[int, HipSage] = {}
_adhoc: Dict[int, HipSage] = {}


def divide_data_by_num(data: Sequence, num_procs: int) -> Iterable[Sequence]:
    """utils.py:106-113: consecutive blocks of len//num_procs + 1 items (may yield fewer blocks than workers)."""
    step = len(data) // num_procs + 1
    for lo in range(0, len(data), step):
        yield data[lo:lo + step]


def segment_line_bounds(n_lines: int, workers: int, flush: int = FLUSH_LINES) -> np.ndarray:
    """Boundaries (n_segments+1 increasing line indices) of the ranges that share one `triples` dict in the
    reference: worker shards of n//W + 1 lines (utils.py:111-113), each cut after its rows flush, 2*flush, ...
    (utils.py:220 `row > 0 and row % chunk_size == 0`: the first range of a shard has flush+1 lines)."""
    step = n_lines // workers + 1
    cuts = [0]
    for lo in range(0, n_lines, step):
        hi = min(n_lines, lo + step)
        at = min(hi, lo + flush + 1)
        while True:
            cuts.append(at)
            if at >= hi:
                break
            at = min(hi, at + flush)
    return np.asarray(cuts, dtype=np.int64)


def current_device() -> int:
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except Exception:
        pass
    return 0


def get_engine(device: int | None = None) -> HipSage:
    dev = current_device() if device is None else device
    if dev not in _engines:
        _engines[dev] = HipSage(dev)
        if os.environ.get("SAGE_DETERMINISTIC", "") not in ("", "0"):
            _engines[dev].set_deterministic(True)
    return _engines[dev]


def pack_lines(lines: Sequence) -> Tuple[np.ndarray, np.ndarray]:
    """List[str|bytes] -> (uint8 bytes, int64 offsets): `bytes(d, 'utf-8')` of utils.py:216 for every line."""
    raw = [ln.encode("utf-8") if isinstance(ln, str) else bytes(ln) for ln in lines]
    off = np.zeros(len(raw) + 1, dtype=np.int64)
    if raw:
        np.cumsum(np.fromiter(map(len, raw), dtype=np.int64, count=len(raw)), out=off[1:])
    flat = np.frombuffer(b"".join(raw), dtype=np.uint8)
    return flat, off


class PackedCorpus:
    """A corpus already in the device layout (bytes + line offsets); accepted wherever a List[str] is.
    Avoids materialising millions of Python strings for large or memory-mapped corpora
    (`hostio.load_corpus_packed`).  Treated as immutable once uploaded; call `touch()` after editing in place."""

    def __init__(self, data: np.ndarray, offsets: np.ndarray):
        self.data = np.ascontiguousarray(data, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.version = 0

    def touch(self) -> None:
        self.version += 1

    def __len__(self):
        return len(self.offsets) - 1

    def line_bytes(self, i: int) -> bytes:
        return self.data[self.offsets[i]:self.offsets[i + 1]].tobytes()

    def slice_packed(self, lo: int, hi: int) -> "PackedCorpus":
        lo, hi = max(0, lo), min(len(self), hi)
        o = self.offsets[lo:hi + 1]
        return PackedCorpus(self.data[o[0]:o[-1]], o - o[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            lo, hi, step = i.indices(len(self))
            if step == 1:
                return self.slice_packed(lo, hi)
            return [self[j] for j in range(lo, hi, step)]
        if i < 0:
            i += len(self)
        return self.line_bytes(i).decode("utf-8")


# ---- upload caches ------------------------------------------------------------------------------------------------
# What is resident on the GPU is remembered in `engine.cache` as (the uploaded object itself, a signature of its
# content).  Holding the object keeps its id from being reused; the signature catches the in-place edits a caller
# is likely to make (whole-array updates, appended or replaced lines).  `engine.invalidate()` forces a re-upload.
def _emb_signature(emb: np.ndarray):
    rows = emb.shape[0]
    sample = emb[::max(1, rows // 509)]
    return (emb.shape, emb.dtype.str, float(sample.sum()), float(np.abs(sample).sum()), float(emb[-1, -1]))


def _corpus_signature(corpus):
    if isinstance(corpus, PackedCorpus):
        return ("packed", len(corpus), corpus.version, int(corpus.offsets[-1]))
    n = len(corpus)
    probe = tuple(corpus[i] for i in range(0, n, max(1, n // 61)))
    return ("list", n, hash(probe), corpus[-1] if n else None)


def sync_vocab(engine: HipSage, sage_model) -> None:
    tag = (sage_model.vocab_version, sage_model.max_len)
    hit = engine.cache.get("vocab")
    if hit is None or hit[0] is not sage_model or hit[1] != tag:
        flat, off, ids = sage_model.packed_key_table()
        pm = getattr(sage_model, "prune_map", None)
        if (hit is not None and hit[0] is sage_model and pm is not None and hit[1] == (pm[0], sage_model.max_len)
                and os.environ.get("SAGE_INCREMENTAL_MATCH", "1") != "0"):
            # the resident vocabulary is exactly the one this prune started from: carry the matches over
            engine.set_vocab_pruned(flat, off, ids, sage_model.max_len, pm[1])
        else:
            engine.set_vocab_packed(flat, off, ids, sage_model.max_len)
        engine.cache["vocab"] = (sage_model, tag)


def sync_embeddings(engine: HipSage, embeddings: np.ndarray) -> None:
    sig = _emb_signature(embeddings)
    hit = engine.cache.get("emb")
    if hit is None or hit[0] is not embeddings or hit[1] != sig:
        engine.set_embeddings(embeddings)
        engine.cache["emb"] = (embeddings, sig)


def _packed(corpus) -> Tuple[np.ndarray, np.ndarray]:
    if isinstance(corpus, PackedCorpus):
        return corpus.data, corpus.offsets
    return pack_lines(corpus)


def sync_corpus(engine: HipSage, partial_corpus) -> None:
    """Whole corpus resident (single GPU)."""
    sig = ("whole", _corpus_signature(partial_corpus))
    hit = engine.cache.get("corpus")
    if hit is None or hit[0] is not partial_corpus or hit[1] != sig:
        flat, off = _packed(partial_corpus)
        engine.set_corpus(flat, off)
        engine.cache["corpus"] = (partial_corpus, sig)
        engine.cache["max_line_bytes"] = int(np.diff(off).max()) if len(off) > 1 else 0


def sync_corpus_shard(engine: HipSage, partial_corpus, bounds: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Only this rank's count segments resident (multi-GPU).  Returns the rank's segment bounds relative to its
    first line.  The split is by corpus bytes over whole segments (`dist.partition_segments`)."""
    from .dist import local_plan
    sig = ("shard", rank, world, tuple(bounds.tolist()) if len(bounds) < 64 else (len(bounds), int(bounds[-1]), hash(bounds.tobytes())),
           _corpus_signature(partial_corpus))
    hit = engine.cache.get("corpus")
    if hit is not None and hit[0] is partial_corpus and hit[1] == sig:
        return engine.cache["local_bounds"]
    if isinstance(partial_corpus, PackedCorpus):
        off = partial_corpus.offsets
        line_lo, line_hi, local = local_plan(bounds, off, rank, world)
        sub = off[line_lo:line_hi + 1]
        engine.set_corpus(partial_corpus.data[sub[0]:sub[-1]], sub - sub[0])
    else:
        off = np.zeros(len(partial_corpus) + 1, dtype=np.int64)
        np.cumsum(np.fromiter((len(s.encode("utf-8")) if isinstance(s, str) else len(s) for s in partial_corpus),
                              dtype=np.int64, count=len(partial_corpus)), out=off[1:])
        line_lo, line_hi, local = local_plan(bounds, off, rank, world)
        flat, sub = pack_lines(partial_corpus[line_lo:line_hi])
        engine.set_corpus(flat, sub)
    engine.cache["corpus"] = (partial_corpus, sig)
    engine.cache["local_bounds"] = local
    engine.cache["max_line_bytes"] = int(np.diff(off).max()) if len(off) > 1 else 0
    return local


def check_hist_overflow(hist: np.ndarray) -> bool:
    """True when an ablation into >= len(hist)-1 pieces was clamped into the last bin (sage_hip.h: out_hist)."""
    return bool(hist[-1] != 0)


def run_sage_parallel(embeddings: np.ndarray, partial_corpus, sage_model, workers_number: int):
    """Drop-in for utils.py:141-190, single GPU.  (Multi-GPU: `sage_amd.dist.run_sage_distributed`.)"""
    engine = get_engine()
    sync_vocab(engine, sage_model)
    sync_corpus(engine, partial_corpus)
    sync_embeddings(engine, embeddings)
    bounds = segment_line_bounds(len(partial_corpus), workers_number)
    logging.info(f"GPU {engine.device}: {len(partial_corpus)} lines as {workers_number} worker shards = {len(bounds) - 1} count segments")
    vsize = sage_model.id_space()
    loss, occ, total_tokens, total_triples, hist = engine.ablation_step(bounds, pad=2, vsize=vsize)
    if check_hist_overflow(hist):
        # the reference's ablated_sizes dict is unbounded (model.py:208-209); an ablation cannot have more pieces than
        # its line has bytes, so one redo with that many bins is exact
        engine.hist_len = engine.cache["max_line_bytes"] + 2
        loss, occ, total_tokens, total_triples, hist = engine.ablation_step(bounds, pad=2, vsize=vsize)
    return total_tokens, total_triples, losses_to_dict(loss, occ), hist_to_dict(hist)


def losses_to_dict(loss: np.ndarray, occ: np.ndarray) -> Dict[int, float]:
    """The reference's loss dict has exactly the multi-byte ids that occur (utils.py:137-138), python floats."""
    ids = np.nonzero(occ)[0]
    return dict(zip(ids.tolist(), loss[ids].tolist()))


def hist_to_dict(hist: np.ndarray) -> Dict[int, int]:
    ks = np.nonzero(hist)[0]
    return dict(zip(ks.tolist(), hist[ks].tolist()))


def tokenize_lines(sage_model, lines: Sequence):
    """GPU tokenization of arbitrary lines on a side context, so a resident partial corpus is not evicted."""
    dev = current_device()
    if dev not in _adhoc:
        _adhoc[dev] = HipSage(dev)
    engine = _adhoc[dev]
    sync_vocab(engine, sage_model)
    flat, off = _packed(lines)
    engine.set_corpus(flat, off)
    return engine.tokenize(0, len(lines))
