"""GPU: batches whose position pairs do not fit 32-bit offsets, and key statistics that defeat the bucket estimate.

Round 2 left an open question (an exact-size key bucket that overflowed "in about half of the runs" on a corpus of
another lexicon than the vocabulary's).  Its cause: that 1M-line corpus has 4,877,430,051 position pairs in one 512 MiB
batch, the scan that produces the pair offsets summed its grand total in 32 bits, so the total wrapped to 582 M, passed
the "fits 32 bits" check, and the wrapped offsets cut one tile with thousands of pooled occurrences -- of which the
partition kernel listed the first 512 by atomic arrival order, a different subset in its counting and its scatter pass.
The total is 64-bit now (the batch is halved and the step starts over), a tile whose list runs full scans its headers
instead of dropping entries, and every step checks records-read == position pairs.  These tests hold all of that."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_scan_grand_total_is_64_bit():
    from sage_amd._lib import HipSage
    eng = HipSage(0)
    try:
        for value, n in ((2000, 3_000_000), (1, 5000), (2_000_000, 2049), (0, 70_000)):     # (items stay below 2^21: a 2,048-item tile sum fits 32 bits)
            total, last = eng.debug_scan_total(value, n)
            assert total == value * n, (value, n, total)
            assert last == (value * n) % (1 << 32)
    finally:
        eng.close()


def _off_lexicon(n_lines):
    from sage_amd import synth
    seed = synth.DEFAULT_SEED
    head, ho = synth.make_corpus(14_000, seed=seed, style="en", n_types=1_000_000)
    vocab = synth.make_vocab(head, ho, 262_144, 16, sample_bytes=6_000_000)
    corpus, off = synth.make_corpus(n_lines, seed=seed + 1, style="en", n_types=1_000_000)   # ANOTHER lexicon
    return vocab, corpus, off


def test_off_lexicon_corpus_against_hbm_sets(monkeypatch):
    """Reduced size (60k lines): text of another lexicon tokenizes into short tokens whose ablations run long (pooled
    windows by the hundred per tile, buckets of repeated keys).  The partitioned count must agree with the per-segment
    HBM sets without ever falling back to them, with and without a pair limit that forces smaller batches."""
    from sage_amd import synth
    from sage_amd._lib import HipSage
    from sage_amd.parallel import segment_line_bounds
    n = 60_000
    vocab, corpus, off = _off_lexicon(n)
    emb = synth.make_embeddings(len(vocab), 50, seed=5)
    bounds = segment_line_bounds(n, 4)

    def run(count_mode):
        eng = HipSage(0)
        try:
            eng.set_vocab(vocab, None, 16); eng.set_corpus(corpus, off); eng.set_embeddings(emb)
            eng.set_count_mode(count_mode)
            out = eng.ablation_step(bounds)
            return out, eng.step_stats()
        finally:
            eng.close()
    (l1, o1, t1, r1, h1), s1 = run(1)
    (l0, o0, t0, r0, h0), s0 = run(0)
    assert s1["count_from_hbm_sets"] == 1 and s0["count_from_hbm_sets"] == 0 and s0["count_redos"] == 0
    assert (t0, r0) == (t1, r1) and np.array_equal(o0, o1) and np.array_equal(h0, h1)
    assert s0["long_occurrences"] > 0.05 * s0["occurrences"]          # the shape this test is about
    np.testing.assert_allclose(l0, l1, rtol=1e-9, atol=1e-12 * np.abs(l1).max())
    monkeypatch.setenv("SAGE_TEST_PAIR_LIMIT", str(s0["pairs"] // 3))  # one batch holds 3x the limit: halved twice
    (l2, o2, t2, r2, h2), s2 = run(0)
    assert s2["count_from_hbm_sets"] == 0 and s2["count_redos"] == 0
    assert (t2, r2) == (t1, r1) and np.array_equal(o2, o1) and np.array_equal(h2, h1)
    np.testing.assert_allclose(l2, l1, rtol=1e-9, atol=1e-12 * np.abs(l1).max())


def test_off_lexicon_corpus_full_size_wraps_32_bits():
    """The round-2 reproducer itself: 1M lines, 4.88e9 position pairs in one 512 MiB batch (> 2^32).  The step must
    notice (64-bit total), halve its batches, size the buckets with the counting pass (repeated keys overflow the
    estimate) and produce the pinned integers without touching the HBM-set fallback -- twice in a row, since the
    failure was intermittent."""
    from sage_amd import synth
    from sage_amd._lib import HipSage
    from sage_amd.parallel import segment_line_bounds
    n = 1_000_000
    vocab, corpus, off = _off_lexicon(n)
    eng = HipSage(0)
    try:
        eng.set_vocab(vocab, None, 16); eng.set_corpus(corpus, off)
        eng.set_embeddings(synth.make_embeddings(len(vocab), 50, seed=5))
        bounds = segment_line_bounds(n, 4)
        for _ in range(2):
            loss, occ, tt, ttr, hist = eng.ablation_step(bounds)
            st = eng.step_stats()
            assert st["pairs"] == 4_877_430_051 > 1 << 32
            assert (tt, ttr) == (146_582_285, 7_935_869_548)        # regression pins (the debug build's compared passes)
            assert st["count_redos"] == 0 and st["count_from_hbm_sets"] == 0 and st["exact_buckets"] == 1
        # one count segment range against the independent back end
        eng.set_count_mode(1)
        _, o1, t1, r1, h1 = eng.ablation_step(bounds[:9])
        eng.set_count_mode(0)
        _, o0, t0, r0, h0 = eng.ablation_step(bounds[:9])
        assert (t0, r0) == (t1, r1) and np.array_equal(o0, o1) and np.array_equal(h0, h1)
    finally:
        eng.close()
