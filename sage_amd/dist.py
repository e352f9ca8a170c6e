"""Multi-GPU form of the ablation step: corpus lines shard across the GPUs of a node on count-segment boundaries
(the reference's only parallelism is data parallel over lines, utils.py:143,155-157), each rank uploads and runs
only its own segments, and the vocabulary-sized result vectors are combined with RCCL (torch.distributed backend
"nccl" on ROCm).  This replaces the parent-side dict merge of utils.py:152-189.

One code path serves the product (`run_sage_distributed`), `bench.py --gpus N` and the tests: the step leaves its
results in torch tensors (`step_into`), `combine` merges them in place.  `combine` uses device collectives when the
process group has a CUDA backend (RCCL) and stages through host memory otherwise (a gloo-only group: CPU tests, and
the rehearsal of N ranks on one GPU).

Integer results (occurrences, histogram, totals) are summed with all_reduce -- exact and order-free.  The f64 loss
vectors are all-gathered and summed in rank order, so the combined vector is bit-identical on every rank and from
run to run; every rank can then take the same prune decision without a broadcast of the pruned vocabulary.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


# ---- process-group helpers ------------------------------------------------------------------------------------
def get_dist():
    """torch.distributed if a process group with more than one rank is initialised, else None."""
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    if dist.is_available() and dist.is_initialized():
        # (SAGE_DIST_SINGLE_RANK=1: a one-rank group counts too -- how the tests drive the RCCL collectives on a one-GPU box)
        if dist.get_world_size() > 1 or os.environ.get("SAGE_DIST_SINGLE_RANK") == "1":
            return dist
    return None


def world_info() -> Tuple[int, int]:
    dist = get_dist()
    return (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)


def device_collectives(dist) -> bool:
    """True when CUDA tensors can go straight into the group's collectives (backend nccl = RCCL on ROCm)."""
    return "cuda:nccl" in str(dist.get_backend_config())


def barrier() -> None:
    dist = get_dist()
    if dist:
        dist.barrier()


def assert_same_on_all_ranks(values, what: str) -> None:
    """Raises RuntimeError on EVERY rank if the integer tuple `values` is not the same on all of them."""
    dist = get_dist()
    if dist is None:
        return
    import torch
    mine = torch.tensor([int(v) for v in values], dtype=torch.int64)
    lo, hi = mine.clone(), mine.clone()
    _through_host(dist, lo, lambda x: dist.all_reduce(x, op=dist.ReduceOp.MIN))
    _through_host(dist, hi, lambda x: dist.all_reduce(x, op=dist.ReduceOp.MAX))
    if not torch.equal(lo, hi):
        raise RuntimeError(f"{what} differs between ranks: this rank has {tuple(mine.tolist())}, "
                           f"minimum {tuple(lo.tolist())}, maximum {tuple(hi.tolist())}")


def _through_host(dist, t, fn):
    """Runs the collective `fn` on a tensor the group's backend can take; the result lands in `t`."""
    if t.is_cuda and not device_collectives(dist):
        h = t.cpu()
        fn(h)
        t.copy_(h)
    elif not t.is_cuda and "cpu:" not in str(dist.get_backend_config()):
        import torch
        d = t.cuda()
        fn(d)
        torch.cuda.synchronize()
        t.copy_(d.cpu())
    else:
        fn(t)


def combine(dist, loss, occ, hist, tot) -> None:
    """In-place combine of the per-rank result tensors (torch, CPU or CUDA) over the default process group."""
    import torch
    world = dist.get_world_size()
    for t in (occ, hist, tot):
        _through_host(dist, t, lambda x: dist.all_reduce(x, op=dist.ReduceOp.SUM))

    def gather_and_sum(x):
        flat = torch.empty(world * x.numel(), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(flat, x.contiguous().view(-1))
        parts = flat.view((world,) + tuple(x.shape))
        acc = parts[0].clone()
        for r in range(1, world):       # fixed rank order => the same bits on every rank and in every run
            acc += parts[r]
        x.copy_(acc)
    _through_host(dist, loss, gather_and_sum)


def broadcast_array(arr: Optional[np.ndarray], src: int = 0) -> np.ndarray:
    """Rank `src`'s array on every rank (shape and dtype travel first); no-op without a process group."""
    dist = get_dist()
    if dist is None:
        return arr
    import torch
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        arr = np.ascontiguousarray(arr)
        meta = [(arr.shape, arr.dtype.str)]
    dist.broadcast_object_list(meta, src=src)
    shape, dtype = meta[0]
    t = torch.from_numpy(arr.copy()) if rank == src else torch.empty(shape, dtype=torch.from_numpy(np.empty(0, dtype=dtype)).dtype)
    _through_host(dist, t, lambda x: dist.broadcast(x, src=src))
    return t.numpy()


# ---- partition ----------------------------------------------------------------------------------------------------
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


def local_plan(bounds: Sequence[int], line_offsets: Sequence[int], rank: int, world: int):
    """-> (line_lo, line_hi, local_bounds): the rank's line range and its segment bounds relative to line_lo."""
    bounds = np.asarray(bounds, dtype=np.int64)
    lo, hi = partition_segments(bounds, line_offsets, world)[rank]
    line_lo, line_hi = int(bounds[lo]), int(bounds[hi])
    return line_lo, line_hi, bounds[lo:hi + 1] - line_lo


# ---- the step -----------------------------------------------------------------------------------------------------
class StepBuffers:
    """Device result buffers of one rank (torch CUDA tensors): what `sage_ablation_step_device` fills and `combine`
    merges.  Allocated once per (vsize, hist_len)."""

    def __init__(self, vsize: int, hist_len: int, device=None):
        import torch
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.vsize, self.hist_len = int(vsize), int(hist_len)
        self.loss = torch.zeros(self.vsize, dtype=torch.float64, device=dev)
        self.occ = torch.zeros(self.vsize, dtype=torch.int64, device=dev)
        self.hist = torch.zeros(self.hist_len, dtype=torch.int64, device=dev)
        self.tot = torch.zeros(2, dtype=torch.int64, device=dev)

    def zero_(self):
        for t in (self.loss, self.occ, self.hist, self.tot):
            t.zero_()


# exception classes a failed step is reported with on the ranks that did not fail themselves (index = status code)
_STEP_ERRORS = (None, IndexError, ValueError, RuntimeError, Exception)


def _status_code(exc: Optional[BaseException]) -> int:
    if exc is None:
        return 0
    for code in (1, 2, 3):
        if isinstance(exc, _STEP_ERRORS[code]):
            return code
    return 4


def raise_if_any_rank_failed(dist, exc: Optional[BaseException], what: str) -> None:
    """utils.py:167 (`res.get()` re-raises a worker's exception in the parent): the ranks exchange one status word each
    BEFORE any data collective, so a rank whose step raised does not leave the others waiting in all_reduce.  The rank
    that failed re-raises its own exception; every other rank raises the same class, naming the ranks that failed."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    status = torch.zeros(world, dtype=torch.int64)
    status[rank] = _status_code(exc)
    _through_host(dist, status, lambda x: dist.all_reduce(x, op=dist.ReduceOp.SUM))
    if exc is not None:
        raise exc
    failed = [(r, int(c)) for r, c in enumerate(status.tolist()) if c]
    if failed:
        cls = _STEP_ERRORS[min(failed[0][1], 4)]
        names = ", ".join(f"rank {r} ({_STEP_ERRORS[min(c, 4)].__name__})" for r, c in failed)
        raise cls(f"{what} failed on {names}; this rank ({rank}) stops with it -- see that rank's log for the message")


def step_into(engine, local_bounds, buf: StepBuffers, dist=None) -> None:
    """One ablation step of this rank's segments into `buf`, then (with a process group) the combine.
    This is the whole per-step N>1 path; bench.py times exactly this call.  A step that raises on one rank raises on
    every rank (`raise_if_any_rank_failed`)."""
    err = None
    try:
        if len(local_bounds) > 1:
            engine.ablation_step_device(local_bounds, (buf.loss.data_ptr(), buf.vsize), buf.occ.data_ptr(),
                                        buf.tot.data_ptr(), (buf.hist.data_ptr(), buf.hist_len))
        else:
            buf.zero_()     # a rank without segments (more ranks than count segments) contributes zeros
    except Exception as exc:      # noqa: BLE001 -- reported to the other ranks, then re-raised here
        if dist is None:
            raise
        err = exc
    if dist is not None:
        raise_if_any_rank_failed(dist, err, "the ablation step")
        combine(dist, buf.loss, buf.occ, buf.hist, buf.tot)


def run_sharded_step(dist, bounds: Sequence[int], line_offsets: Sequence[int],
                     step_fn: Callable[[np.ndarray], tuple]):
    """One distributed step over host arrays: `step_fn(global_bounds_of_this_rank) -> (loss, occ, total_tokens,
    total_triples, hist)`.  Returns the combined results, identical on every rank.  (The product's step runs on
    device buffers: `step_into`; this form takes any step function and is what the CPU tests drive.)"""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = partition_segments(bounds, line_offsets, world)[rank]
    local = np.asarray(bounds[lo:hi + 1], dtype=np.int64)
    err, out = None, None
    try:
        out = step_fn(local)
    except Exception as exc:      # noqa: BLE001
        err = exc
    raise_if_any_rank_failed(dist, err, "the ablation step")
    loss, occ, tt, ttr, hist = out
    t_loss, t_occ, t_hist = (torch.from_numpy(np.array(a)) for a in (loss, occ, hist))
    t_tot = torch.tensor([tt, ttr], dtype=torch.int64)
    combine(dist, t_loss, t_occ, t_hist, t_tot)
    return t_loss.numpy(), t_occ.numpy(), int(t_tot[0]), int(t_tot[1]), t_hist.numpy()


def run_sage_distributed(embeddings: np.ndarray, partial_corpus, sage_model, workers_number: int):
    """Multi-GPU drop-in for run_sage_parallel (utils.py:141-190): call it from every rank of an initialised
    torch.distributed job (one process per GPU, backend "nccl").  Every rank holds the partial corpus on the host
    (it is read from the same file everywhere) but uploads and runs only its own count segments; the result is
    identical on every rank.  Without a process group it is the single-GPU step."""
    from . import parallel
    dist = get_dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
    engine = parallel.get_engine()
    err, local = None, None
    try:
        parallel.sync_vocab(engine, sage_model)
        parallel.sync_embeddings(engine, embeddings)
        n_lines = len(partial_corpus)
        bounds = parallel.segment_line_bounds(n_lines, workers_number)
        local = parallel.sync_corpus_shard(engine, partial_corpus, bounds, rank, world)
    except Exception as exc:      # noqa: BLE001
        if dist is None:
            raise
        err = exc
    if dist is not None:
        raise_if_any_rank_failed(dist, err, "uploading the vocabulary, embeddings or corpus shard")
    vsize = sage_model.id_space()
    while True:
        buf = engine.cache.get("buffers")
        if buf is None or buf.vsize != vsize or buf.hist_len != engine.hist_len:
            buf = engine.cache["buffers"] = StepBuffers(vsize, engine.hist_len)
        step_into(engine, local, buf, dist)
        hist = buf.hist.cpu().numpy()
        if not parallel.check_hist_overflow(hist):
            break
        # the combined histogram is the same on every rank, so every rank takes this branch together
        engine.hist_len = engine.cache["max_line_bytes"] + 2
    loss, occ, tot = buf.loss.cpu().numpy(), buf.occ.cpu().numpy(), buf.tot.cpu().numpy()
    return int(tot[0]), int(tot[1]), parallel.losses_to_dict(loss, occ), parallel.hist_to_dict(hist)
