"""Multi-GPU form of the ablation step: corpus lines shard across the GPUs of a node on count-segment boundaries
(the reference's only parallelism is data parallel over lines, utils.py:143,155-157), each rank runs the step on
its own segments, and the vocabulary-sized result vectors are combined with RCCL (torch.distributed backend
"nccl" on ROCm).  This replaces the parent-side dict merge of utils.py:170-179.

Integer results (occurrences, histogram, totals) are summed with all_reduce -- exact and order-free.  The f64 loss
vectors are all-gathered and summed in rank order so the result does not depend on arrival order.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np


def partition_segments(bounds: Sequence[int], line_offsets: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Splits the segment list (n_seg+1 line boundaries) into `world` contiguous runs [seg_lo, seg_hi) balanced by
    corpus bytes.  Every count segment stays whole on one rank, so total_triples is independent of `world`."""
    bounds = np.asarray(bounds, dtype=np.int64)
    n_seg = len(bounds) - 1
    off = np.asarray(line_offsets, dtype=np.int64)
    seg_end_bytes = off[bounds[1:]] - off[bounds[0]]
    total = int(seg_end_bytes[-1]) if n_seg else 0
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        # first segment whose end passes the target goes to the earlier rank if that lands closer
        s = int(np.searchsorted(seg_end_bytes, target, side="left"))
        if s < n_seg:
            before = seg_end_bytes[s - 1] if s > 0 else 0
            if (seg_end_bytes[s] - target) <= (target - before):
                s += 1
        cuts.append(min(max(s, cuts[-1]), n_seg))
    cuts.append(n_seg)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def combine_device(dist, d_loss, d_occ, d_hist, d_tot) -> None:
    """In-place combine of per-rank device tensors over the default process group (RCCL on GPUs)."""
    world = dist.get_world_size()
    dist.all_reduce(d_occ, op=dist.ReduceOp.SUM)
    dist.all_reduce(d_hist, op=dist.ReduceOp.SUM)
    dist.all_reduce(d_tot, op=dist.ReduceOp.SUM)
    import torch
    parts = torch.empty((world,) + tuple(d_loss.shape), dtype=d_loss.dtype, device=d_loss.device)
    dist.all_gather_into_tensor(parts, d_loss.contiguous())
    acc = parts[0].clone()
    for r in range(1, world):       # fixed rank order => run-to-run reproducible sum
        acc += parts[r]
    d_loss.copy_(acc)


def combine_host(dist, loss: np.ndarray, occ: np.ndarray, hist: np.ndarray, totals: np.ndarray):
    """Same combine for host arrays (gloo): used by the CPU tests of the N>1 path."""
    import torch
    t_occ, t_hist, t_tot = torch.from_numpy(occ.copy()), torch.from_numpy(hist.copy()), torch.from_numpy(totals.copy())
    dist.all_reduce(t_occ, op=dist.ReduceOp.SUM)
    dist.all_reduce(t_hist, op=dist.ReduceOp.SUM)
    dist.all_reduce(t_tot, op=dist.ReduceOp.SUM)
    world = dist.get_world_size()
    mine = torch.from_numpy(loss.copy())
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    acc = parts[0].clone()
    for r in range(1, world):
        acc += parts[r]
    return acc.numpy(), t_occ.numpy(), t_hist.numpy(), t_tot.numpy()


def run_sharded_step(dist, bounds: Sequence[int], line_offsets: Sequence[int],
                     step_fn: Callable[[np.ndarray], tuple]):
    """One distributed step.  `step_fn(local_bounds) -> (loss, occ, total_tokens, total_triples, hist)` runs the
    rank's segments (product: HipSage.ablation_step on the resident corpus).  Returns the combined
    (loss, occ, total_tokens, total_triples, hist), identical on every rank."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = partition_segments(bounds, line_offsets, world)[rank]
    local = np.asarray(bounds[lo:hi + 1], dtype=np.int64)
    loss, occ, tt, ttr, hist = step_fn(local)
    loss, occ, hist, tot = combine_host(dist, loss, occ, hist, np.array([tt, ttr], dtype=np.int64))
    return loss, occ, int(tot[0]), int(tot[1]), hist


def run_sage_distributed(embeddings: np.ndarray, partial_corpus, sage_model, workers_number: int):
    """Multi-GPU drop-in for run_sage_parallel (utils.py:141): call it from every rank of an initialised
    torch.distributed job (one process per GPU).  Each rank keeps the whole partial corpus resident (it is read
    from the same file on every rank) and runs only its own count segments."""
    import torch.distributed as dist
    from . import parallel
    engine = parallel.get_engine()
    slot = parallel._state[engine.device]
    parallel.sync_vocab(engine, sage_model, slot)
    parallel.sync_corpus(engine, partial_corpus, slot)
    parallel.sync_embeddings(engine, embeddings, slot)
    bounds = parallel.segment_line_bounds(len(partial_corpus), workers_number)
    vsize = max(sage_model.byte_vocab.values()) + 1
    off = engine_line_offsets(partial_corpus)
    loss, occ, tt, ttr, hist = run_sharded_step(
        dist, bounds, off, lambda b: engine.ablation_step(b, pad=2, vsize=vsize) if len(b) > 1 else
        (np.zeros(vsize), np.zeros(vsize, dtype=np.int64), 0, 0, np.zeros(engine.HIST_LEN, dtype=np.int64)))
    return tt, ttr, parallel.losses_to_dict(loss, occ), parallel.hist_to_dict(hist)


def engine_line_offsets(partial_corpus) -> np.ndarray:
    from .parallel import PackedCorpus
    if isinstance(partial_corpus, PackedCorpus):
        return partial_corpus.offsets
    off = np.zeros(len(partial_corpus) + 1, dtype=np.int64)
    np.cumsum(np.fromiter((len(s.encode("utf-8")) if isinstance(s, str) else len(s) for s in partial_corpus),
                          dtype=np.int64, count=len(partial_corpus)), out=off[1:])
    return off
