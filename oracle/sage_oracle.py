"""CPU oracle for the SaGe 2.0 ablation hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the
product path (`sage_amd`) never does and fails loudly when the HIP library is missing.

It is a plain-Python / numpy restatement of the reference's algorithm, dict-based like the reference so the
floating-point summation order inside one worker shard is the reference's too.  Each function names the
reference lines it restates (paths are under /root/reference/src/sage_tokenizer/).  Parity status: PINNED --
`tests/test_oracle_golden.py` checks every function below against vectors captured from the imported
reference by `tests/golden/make_golden.py` (the reference ships no tests or fixtures of its own).
"""
from __future__ import annotations

import multiprocessing as mp
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
from scipy.special import expit

FLUSH_LINES = 10000  # utils.py:193 chunk_size
DEFAULT_PAD = 2      # model.py:138


class OracleTokenizer:
    """Vocabulary state + greedy matcher (model.py:15-51, 81-101, 251-255)."""

    def __init__(self, vocab: Sequence[bytes], max_len: int = 16):
        self.max_len = int(max_len)
        self.set_vocabulary(vocab)

    def set_vocabulary(self, vocab: Sequence[bytes]) -> None:
        # model.py:44-51: id = position in the list; a repeated key keeps its LAST position.
        table: Dict[bytes, int] = {}
        for pos, key in enumerate(vocab):
            if type(key) is not bytes:
                raise AssertionError("vocabulary entries must be bytes")
            table[key] = pos
        # model.py:251-255
        for b in range(256):
            if bytes([b]) not in table:
                raise Exception(f"missing byte {bytes([b])}")
        self.table = table
        self.inverse = {v: k for k, v in table.items()}

    def vocab_size(self) -> int:
        return len(self.table)

    def byte_ids(self) -> List[int]:
        return [self.table[bytes([b])] for b in range(256)]

    def longest_at(self, sent: bytes, at: int, cap: int) -> Tuple[int, int]:
        """Longest key that is a prefix of sent[at:] with length <= cap (slices saturate at the line end
        exactly as `sent[i:i+j]` does in model.py:91).  Returns (id, length)."""
        table = self.table
        for j in range(cap, 0, -1):
            piece = sent[at:at + j]
            if piece in table:
                # model.py:98-99: the reported width is len(piece) but the cursor advances by j; they can
                # differ only when the slice was truncated by the line end, and then the walk stops anyway.
                return table[piece], j
        raise Exception("unreachable: single bytes are always present")

    def tokenize(self, sent: bytes) -> List[Tuple[int, int, int]]:
        """model.py:81-101 -> [(id, start, width)]."""
        if isinstance(sent, str):
            sent = sent.encode("utf-8")
        out = []
        at, n = 0, len(sent)
        while at < n:
            tid, j = self.longest_at(sent, at, self.max_len)
            out.append((tid, at, len(sent[at:at + j])))
            at += j
        return out

    def ablate_line(self, sent: bytes, triples: Dict[Tuple[int, int, int], int], sizes: Dict[int, int],
                    pad: int = DEFAULT_PAD, occurrences: Dict[int, int] | None = None) -> int:
        """model.py:138-248 (`fast_sage`): adds the signed (ablated, target, context) counts of one line to
        `triples`, the ablated-length histogram to `sizes`, returns the line's token count."""
        n = len(sent)
        toks = self.tokenize(sent)
        if not toks:
            # model.py:148: zip(*[]) cannot be unpacked into three names
            raise ValueError("not enough values to unpack (expected 3, got 0)")
        ids = [t[0] for t in toks]
        ntok = len(toks)
        base_index = {start: j for j, (_, start, _) in enumerate(toks)}  # model.py:163-165 (on_base)
        for loc, (cur, start, width) in enumerate(toks):
            if width <= 1:
                continue
            # model.py:176-182: first piece is at most width-1 bytes
            tid, j = self.longest_at(sent, start, width - 1)
            ablated = [tid]
            at = start + j
            beyond = None
            # model.py:185-202: ordinary greedy steps until a base-token start or the line end
            while at < n:
                tid, j = self.longest_at(sent, at, min(self.max_len, n - at))
                ablated.append(tid)
                at += j
                if at >= n:
                    beyond = n           # model.py:194-196: a BYTE count used as a token index
                    break
                if at in base_index:
                    beyond = base_index[at]
                    break
            k = len(ablated)
            sizes[k] = sizes.get(k, 0) + 1
            if occurrences is not None:
                occurrences[cur] = occurrences.get(cur, 0) + 1
            base_span = ids[loc:beyond]
            padleft = min(pad, loc)
            padright = min(pad, ntok - beyond)            # may be <= 0 (model.py:219)
            left = ids[loc - padleft:loc]
            right = ids[beyond:beyond + padright] if padright > 0 else []
            window_triples(left + ablated + right, pad, padleft, padright, cur, +1, triples)
            window_triples(left + base_span + right, pad, padleft, padright, cur, -1, triples)
        return ntok


def window_triples(window: List[int], pad: int, padleft: int, padright: int, cur: int, sign: int,
                   triples: Dict[Tuple[int, int, int], int]) -> None:
    """model.py:115-136 (`do_triples`): every ordered (target, context) at distance 1..pad with at least one
    of the two inside the centre [padleft, len-padright); keys whose count returns to 0 stay in the dict."""
    m = len(window)
    right_edge = m - padright
    for t in range(m):
        t_mid = padleft <= t < right_edge
        for c in range(max(0, t - pad), min(m, t + pad + 1)):
            if c == t:
                continue
            if t_mid or (padleft <= c < right_edge):
                key = (cur, window[t], window[c])
                triples[key] = triples.get(key, 0) + sign


def score_triples(losses: Dict[int, float], triples: Dict[Tuple[int, int, int], int], emb: np.ndarray) -> None:
    """utils.py:125-138 (`compute_losses`): losses[a] += count * log(expit(E[t].E[c])) in dict order."""
    if not triples:
        raise ValueError("not enough values to unpack (expected 3, got 0)")  # utils.py:132 on an empty dict
    keys = list(triples.keys())
    cnt = np.fromiter((triples[k] for k in keys), dtype=np.int64, count=len(keys))
    tgt = np.fromiter((k[1] for k in keys), dtype=np.int64, count=len(keys))
    ctx = np.fromiter((k[2] for k in keys), dtype=np.int64, count=len(keys))
    term = cnt * np.log(expit(np.einsum("ij,ij->i", emb[tgt], emb[ctx])))
    for (abl, _, _), v in zip(keys, term):
        losses[abl] = losses.get(abl, 0.0) + v


def shard_bounds(n_lines: int, workers: int) -> List[Tuple[int, int]]:
    """utils.py:106-113 (`divide_data_by_num`): consecutive blocks of n//W + 1 lines."""
    step = n_lines // workers + 1
    return [(lo, min(n_lines, lo + step)) for lo in range(0, n_lines, step)]


def segment_bounds(n_lines: int, workers: int, flush: int = FLUSH_LINES) -> List[Tuple[int, int]]:
    """Line ranges over which the reference counts distinct triples: each worker shard (utils.py:111-113)
    cut after its rows flush, 2*flush, ... (utils.py:220: `row > 0 and row % chunk_size == 0`), so the first
    segment of a shard has flush+1 lines."""
    segs = []
    for lo, hi in shard_bounds(n_lines, workers):
        a = lo
        while a < hi:
            b = min(hi, lo + flush + 1) if a == lo else min(hi, a + flush)
            segs.append((a, b))
            a = b
    return segs


def run_shard(tok: OracleTokenizer, lines: Iterable, emb: np.ndarray, flush: int = FLUSH_LINES,
              pad: int = DEFAULT_PAD, occurrences: Dict[int, int] | None = None):
    """utils.py:193-257 (`sage_per_chunk`) for one worker shard."""
    losses: Dict[int, float] = {}
    triples: Dict[Tuple[int, int, int], int] = {}
    sizes: Dict[int, int] = {}
    total_tokens = 0
    total_triples = 0
    for row, line in enumerate(lines):
        sent = line.encode("utf-8") if isinstance(line, str) else bytes(line)
        total_tokens += tok.ablate_line(sent, triples, sizes, pad, occurrences)
        if row > 0 and row % flush == 0:
            score_triples(losses, triples, emb)
            total_triples += len(triples)
            triples = {}
    if triples:
        score_triples(losses, triples, emb)
        total_triples += len(triples)
    return {k: -v for k, v in losses.items()}, total_tokens, total_triples, sizes


def _pool_entry(args):
    vocab, max_len, lines, emb, flush, pad = args
    return run_shard(OracleTokenizer(vocab, max_len), lines, emb, flush, pad)


def run_parallel(emb: np.ndarray, lines: Sequence, tok: OracleTokenizer, workers: int,
                 flush: int = FLUSH_LINES, pad: int = DEFAULT_PAD, processes: int = 0):
    """utils.py:141-190 (`run_sage_parallel`).  Shards are merged in shard order (the reference merges in
    completion order, so its float sums are order-nondeterministic; integers are order-free).
    `processes` > 0 runs the shards in a multiprocessing pool like the reference does."""
    bounds = shard_bounds(len(lines), workers)
    if processes > 0:
        jobs =[(_vocab_list(tok), tok.max_len, lines[lo:hi], emb, flush, pad) for lo, hi in bounds]
        with mp.Pool(processes=processes) as pool:
            parts = pool.map(_pool_entry, jobs, chunksize=1)
    else:
        parts = [run_shard(tok, lines[lo:hi], emb, flush, pad) for lo, hi in bounds]
    losses: Dict[int, float] = {}
    sizes: Dict[int, int] = {}
    total_tokens = total_triples = 0
    for part_losses, ntok, ntrip, part_sizes in parts:
        total_tokens += ntok
        total_triples += ntrip
        for k, v in part_losses.items():
            losses[k] = losses.get(k, 0) + v
        for k, v in part_sizes.items():
            sizes[k] = sizes.get(k, 0) + v
    return total_tokens, total_triples, losses, sizes


def _vocab_list(tok: OracleTokenizer) -> List[bytes]:
    """A list that rebuilds `tok.table` exactly (ids preserved even when the original list had repeats)."""
    top = max(tok.table.values())
    out: List[bytes] = [None] * (top + 1)  # type: ignore[list-item]
    for k, v in tok.table.items():
        out[v] = k
    # a vacated position (earlier copy of a repeated key) is filled with that id's eventual key so the
    # dict built from the list is unchanged: repeat any key that maps to a LATER id
    later = None
    for i in range(top, -1, -1):
        if out[i] is None:
            out[i] = later
        else:
            later = out[i]
    return out


# ----------------------------------------------------------------------------------------------------------
# Prune selection, SaGeVocabBuilder.py:106-192, as a pure function (host logic shared by the parity tests)
# ----------------------------------------------------------------------------------------------------------
def select_prune(losses: Dict[int, float], tok: OracleTokenizer, schedule: List[int], i: int):
    """Given the merged losses of round i, returns (i', target, sorted_losses, pruned_ids) following
    SaGeVocabBuilder.py:109-159: single bytes get 1e6, targets >= active are skipped, the cut is the first
    active-target entries of sorted((loss, id))."""
    scored = dict(losses)
    for tid in tok.byte_ids():
        scored[tid] = 1e6
    active = len(scored)
    target = schedule[i + 1]
    while active <= target:
        i += 1
        target = schedule[i + 1]  # IndexError off the end, like the reference
    ranked = sorted((v, k) for k, v in scored.items())
    pruned = [tid for _, tid in ranked[:active - target]]
    return i, target, ranked, pruned
