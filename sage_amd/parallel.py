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
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

from ._lib import HipSage

FLUSH_LINES = 10000  # utils.py:193 chunk_size

_engines: Dict[int, HipSage] = {}
_adhoc: Dict[int, HipSage] = {}
_state: Dict[int, dict] = {}


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
        _state[dev] = {"vocab": None, "corpus": None, "emb": None}
    return _engines[dev]


def pack_lines(lines: Sequence) -> Tuple[np.ndarray, np.ndarray]:
    """List[str|bytes] -> (uint8 bytes, int64 offsets): `bytes(d, 'utf-8')` of utils.py:216 for every line."""
    raw = [ln.encode("utf-8") if isinstance(ln, str) else bytes(ln) for ln in lines]
    off = np.zeros(len(raw) + 1, dtype=np.int64)
    if raw:
        np.cumsum(np.fromiter((len(r) for r in raw), dtype=np.int64, count=len(raw)), out=off[1:])
    flat = np.frombuffer(b"".join(raw), dtype=np.uint8)
    return flat, off


def sync_vocab(engine: HipSage, sage_model, slot: dict) -> None:
    tag = (id(sage_model), sage_model.vocab_version, sage_model.max_len)
    if slot.get("vocab") != tag:
        keys, ids = sage_model.key_table()
        engine.set_vocab(keys, ids, sage_model.max_len)
        slot["vocab"] = tag


def sync_corpus(engine: HipSage, partial_corpus, slot: dict) -> None:
    if isinstance(partial_corpus, PackedCorpus):
        tag = ("packed", id(partial_corpus))
        if slot.get("corpus") != tag:
            engine.set_corpus(partial_corpus.data, partial_corpus.offsets)
            slot["corpus"] = tag
        return
    n = len(partial_corpus)
    probe = (partial_corpus[0], partial_corpus[n // 2], partial_corpus[-1]) if n else ()
    tag = (id(partial_corpus), n, probe)
    if slot.get("corpus") != tag:
        flat, off = pack_lines(partial_corpus)
        engine.set_corpus(flat, off)
        slot["corpus"] = tag


def sync_embeddings(engine: HipSage, embeddings: np.ndarray, slot: dict) -> None:
    tag = (id(embeddings), embeddings.shape, float(embeddings[0, 0]), float(embeddings[-1, -1]))
    if slot.get("emb") != tag:
        engine.set_embeddings(embeddings)
        slot["emb"] = tag


class PackedCorpus:
    """A corpus already in the device layout (bytes + line offsets); accepted wherever a List[str] is.
    Avoids materialising millions of Python strings for large synthetic or memory-mapped corpora."""

    def __init__(self, data: np.ndarray, offsets: np.ndarray):
        self.data = np.ascontiguousarray(data, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return self.data[self.offsets[i]:self.offsets[i + 1]].tobytes().decode("utf-8")


def run_sage_parallel(embeddings: np.ndarray, partial_corpus, sage_model, workers_number: int):
    """Drop-in for utils.py:141-190, single GPU.  (Multi-GPU: `sage_amd.dist.run_sage_distributed`.)"""
    engine = get_engine()
    slot = _state[engine.device]
    sync_vocab(engine, sage_model, slot)
    sync_corpus(engine, partial_corpus, slot)
    sync_embeddings(engine, embeddings, slot)
    bounds = segment_line_bounds(len(partial_corpus), workers_number)
    logging.info(f"Splitting data into {workers_number} chunks ({len(bounds) - 1} count segments) on GPU {engine.device}.")
    vsize = max(sage_model.byte_vocab.values()) + 1
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
        _state[("adhoc", dev)] = {"vocab": None}
    engine = _adhoc[dev]
    sync_vocab(engine, sage_model, _state[("adhoc", dev)])
    if isinstance(lines, PackedCorpus):
        engine.set_corpus(lines.data, lines.offsets)
    else:
        flat, off = pack_lines(lines)
        engine.set_corpus(flat, off)
    return engine.tokenize(0, len(lines))
