"""GPU: the BASELINE.json workload itself (1,000,000 lines, 262,144 keys, D=50, max_len=16, workers_number=4 -- what
bench.py times), checked through properties that need no oracle at this size: the step over all count segments is the
sum of the steps over two halves (integers exact, losses to 1e-9), an independent distinct-count back end (per-segment
HBM sets, count mode 1) and the unfiltered matcher give the same integers, and the totals tie together.  The value of
total_triples is also pinned to what every build of this repository since v1 has produced on this corpus
(profiles/r01_v*_bench_full.json): a regression pin, NOT an oracle value -- the oracle pins the algorithm at small
sizes (tests/test_gpu_parity.py), this pins the large-size plumbing (batches, 32-bit offsets, bucket sizing)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_LINES = 1_000_000
PINNED_TOTAL_TOKENS = 86_193_104
PINNED_TOTAL_TRIPLES = 3_311_102_221


def test_baseline_workload_properties():
    from sage_amd import synth
    from sage_amd._lib import HipSage
    from sage_amd.parallel import segment_line_bounds
    seed = synth.DEFAULT_SEED
    head, head_off = synth.make_corpus(14_000, seed=seed, style="en", n_types=1_000_000)
    vocab = synth.make_vocab(head, head_off, 262_144, 16, sample_bytes=6_000_000)
    corpus, off = synth.make_corpus(N_LINES, seed=seed, style="en", n_types=1_000_000)
    eng = HipSage(0)
    try:
        eng.set_vocab(vocab, None, 16)
        eng.set_corpus(corpus, off)
        per_line, total = eng.token_counts(0, N_LINES)
        counts = np.bincount(eng.tokenize(0, N_LINES)[0], minlength=len(vocab))
        emb = synth.make_embeddings(len(vocab), 50, seed=seed, unseen=~(counts > 0))
        eng.set_embeddings(emb)
        bounds = segment_line_bounds(N_LINES, 4)
        loss, occ, tt, ttr, hist = eng.ablation_step(bounds)
        assert tt == total == int(counts.sum()) == int(per_line.sum()) == PINNED_TOTAL_TOKENS
        assert ttr == PINNED_TOTAL_TRIPLES
        assert int(hist.sum()) == int(occ.sum()) and np.isfinite(loss).all()
        assert not np.any((occ > 0) & (counts == 0))          # only tokens the tokenization produces are ablated
        # halves add up (what sharding over GPUs relies on)
        n_seg = len(bounds) - 1
        mid = n_seg // 2
        l1, o1, t1, r1, h1 = eng.ablation_step(bounds[:mid + 1])
        l2, o2, t2, r2, h2 = eng.ablation_step(bounds[mid:])
        assert t1 + t2 == tt and r1 + r2 == ttr
        assert np.array_equal(o1 + o2, occ) and np.array_equal(h1 + h2, hist)
        np.testing.assert_allclose(l1 + l2, loss, rtol=1e-9, atol=1e-12 * np.abs(loss).max())
        # independent back ends: HBM sets for the distinct count, the unfiltered matcher
        eng.set_count_mode(1)
        _, o3, t3, r3, h3 = eng.ablation_step(bounds[:mid + 1])
        eng.set_count_mode(0)
        assert (t3, r3) == (t1, r1) and np.array_equal(o3, o1) and np.array_equal(h3, h1)
        eng.set_match_mode(1)
        _, o4, t4, r4, h4 = eng.ablation_step(bounds[mid:])
        eng.set_match_mode(0)
        assert (t4, r4) == (t2, r2) and np.array_equal(o4, o2) and np.array_equal(h4, h2)
    finally:
        eng.close()
