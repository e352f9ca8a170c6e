"""GPU: size-independent properties of the ablation step at a size the Python oracle cannot reach (200k lines,
65,536 keys): the step must not depend on how the work is cut (batches, GPU shards, kernel variants), and its parts
must add up.  Integer outputs are compared exactly, losses to 1e-9 relative (different float64 summation orders)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_LINES = 200_000


@pytest.fixture(scope="module")
def world():
    from sage_amd import synth
    from sage_amd._lib import HipSage
    from sage_amd.parallel import segment_line_bounds
    head, head_off = synth.make_corpus(14_000, seed=99, n_types=300_000)
    vocab = synth.make_vocab(head, head_off, 65_536, 16, sample_bytes=4_000_000)
    corpus, off = synth.make_corpus(N_LINES, seed=99, n_types=300_000)
    eng = HipSage(0)
    eng.set_vocab(vocab, None, 16)
    eng.set_corpus(corpus, off)
    seen = np.bincount(eng.tokenize(0, N_LINES)[0], minlength=len(vocab)) > 0
    emb = synth.make_embeddings(len(vocab), 50, seed=1, unseen=~seen)
    eng.set_embeddings(emb)
    bounds = segment_line_bounds(N_LINES, 4)
    ref = eng.ablation_step(bounds)
    yield eng, vocab, corpus, off, emb, bounds, ref
    eng.close()


def _same(a, b, rtol=1e-9):
    loss_a, occ_a, tt_a, ttr_a, hist_a = a
    loss_b, occ_b, tt_b, ttr_b, hist_b = b
    assert tt_a == tt_b and ttr_a == ttr_b
    assert np.array_equal(occ_a, occ_b) and np.array_equal(hist_a, hist_b)
    scale = np.abs(loss_a).max()
    np.testing.assert_allclose(loss_a, loss_b, rtol=rtol, atol=1e-12 * scale)


def test_token_total_is_the_tokenization(world):
    eng, vocab, corpus, off, emb, bounds, ref = world
    counts, total = eng.token_counts(0, N_LINES)
    assert ref[2] == total == int(counts.sum())
    assert int(ref[4].sum()) == int(ref[1].sum())          # every ablated occurrence lands in ablated_sizes once
    assert ref[3] > 0 and np.isfinite(ref[0]).all()


def test_independent_of_batch_size(world):
    eng, vocab, corpus, off, emb, bounds, ref = world
    eng.set_batch_bytes(24 << 20)                          # 5 batches instead of 1
    try:
        _same(ref, eng.ablation_step(bounds))
    finally:
        eng.set_batch_bytes(256 << 20)


def test_kernel_variants_agree(world):
    eng, vocab, corpus, off, emb, bounds, ref = world
    eng.set_count_mode(1)                                  # HBM sets instead of partition + LDS dedupe
    try:
        _same(ref, eng.ablation_step(bounds))
    finally:
        eng.set_count_mode(0)
    for mode in (1, 2):                                    # unfiltered matcher; column probes of the slot table
        eng.set_match_mode(mode)
        try:
            _same(ref, eng.ablation_step(bounds))
        finally:
            eng.set_match_mode(0)
    eng.set_score_mode(1)                                  # float64 row storage
    eng.set_embeddings(emb)
    try:
        _same(ref, eng.ablation_step(bounds))
    finally:
        eng.set_score_mode(0)
        eng.set_embeddings(emb)


def test_segments_add_up(world):
    """The step over a run of count segments is the sum of the steps over its parts (what multi-GPU sharding relies
    on): cut the segment list in three and add."""
    eng, vocab, corpus, off, emb, bounds, ref = world
    n_seg = len(bounds) - 1
    cuts = [0, n_seg // 3, 2 * n_seg // 3, n_seg]
    loss = np.zeros_like(ref[0]); occ = np.zeros_like(ref[1]); hist = np.zeros_like(ref[4]); tt = 0; ttr = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        l, o, t, r, h = eng.ablation_step(bounds[a:b + 1])
        loss += l; occ += o; hist += h; tt += t; ttr += r
    _same(ref, (loss, occ, tt, ttr, hist))


def test_repeatable(world):
    eng, vocab, corpus, off, emb, bounds, ref = world
    again = eng.ablation_step(bounds)
    _same(ref, again, rtol=1e-12)


def test_deterministic_mode_is_bit_identical(world):
    """sage_set_deterministic: fixed-point sums make the loss vector bit-identical from run to run, for any batch size
    and for both row storage modes' shared terms; it agrees with the default float64-atomic sums to rounding."""
    eng, vocab, corpus, off, emb, bounds, ref = world
    eng.set_deterministic(True)
    try:
        a = eng.ablation_step(bounds)
        b = eng.ablation_step(bounds)
        assert np.array_equal(a[0], b[0])
        eng.set_batch_bytes(8 << 20)                 # many small batches: other tiles, other slices, other atomic orders
        c = eng.ablation_step(bounds)
        eng.set_batch_bytes(256 << 20)
        assert np.array_equal(a[0], c[0])
        assert a[2] == c[2] and a[3] == c[3] and np.array_equal(a[1], c[1]) and np.array_equal(a[4], c[4])
        # halves: integer sums add exactly, so the two halves differ from the whole only by the final conversions
        mid = (len(bounds) - 1) // 2
        h1, h2 = eng.ablation_step(bounds[:mid + 1]), eng.ablation_step(bounds[mid:])
        np.testing.assert_allclose(h1[0] + h2[0], a[0], rtol=1e-13, atol=1e-13 * np.abs(a[0]).max())
        _same(ref, a, rtol=1e-10)                    # the default mode's sums, to float64 rounding
    finally:
        eng.set_deterministic(False)
        eng.set_batch_bytes(256 << 20)
    d = eng.ablation_step(bounds)
    _same(ref, d, rtol=1e-12)


def test_batches_shrink_when_pair_offsets_would_overflow(monkeypatch):
    """32-bit pair and dot offsets bound a batch; a batch above the bound makes the step halve batch_bytes and start
    over (SAGE_TEST_PAIR_LIMIT lowers the bound so that a small corpus reaches it).  Results must not change."""
    from sage_amd import synth
    from sage_amd._lib import HipSage
    from sage_amd.parallel import segment_line_bounds
    corpus, off = synth.make_corpus(3000, seed=21, style="en", n_types=4000, mean_log_words=2.8)
    vocab = synth.make_vocab(corpus, off, 4000, 16, sample_bytes=400_000)
    emb = synth.make_embeddings(len(vocab), 50, seed=3)
    bounds = segment_line_bounds(len(off) - 1, 2, flush=100)
    def run():
        eng = HipSage(0)
        try:
            eng.set_vocab(vocab, None, 16); eng.set_corpus(corpus, off); eng.set_embeddings(emb)
            return eng.ablation_step(bounds), eng.step_stats()
        finally:
            eng.close()
    ref, st = run()
    assert st["pairs"] > 400_000
    monkeypatch.setenv("SAGE_TEST_PAIR_LIMIT", str(st["pairs"] // 5))     # whole corpus in one batch: 5x the bound
    got, _ = run()
    _same(ref, got)
