"""GPU: the BASELINE.json workloads at full single-GPU size, checked through properties that need no oracle at this
size.  configs[1]: 1,000,000 English lines, 262,144 keys, D=50, max_len=16, workers_number=4 (what bench.py times);
configs[3] (C4) at its size: Turkish-style multi-byte corpus, max_len 17, 5,000,000 lines (4.38 GB, nine 512 MiB batches);
configs[4] (C5) at its per-GPU size: D=300 on 2,500,000 English lines (a 20M-line corpus over 8 GPUs).  Properties: the
step over all count segments is the sum of the steps over two halves (integers exact, losses to 1e-9), an independent
distinct-count back end (per-segment HBM sets, count mode 1) and the unfiltered matcher give the same integers, float64
row storage gives the same losses as the float32 copy, the partitioned count never fell back, and the totals tie
together.  On the 1M-line English corpus total_tokens / total_triples are also pinned to what every build of this
repository since v1 has produced (profiles/r01_v*_bench_full.json): a regression pin, NOT an oracle value -- the oracle
pins the algorithm at small sizes (tests/test_gpu_parity.py), this pins the large-size plumbing (batches, 32-bit
offsets, bucket sizing).  (make_corpus draws its lines chunk by chunk from one stream, so the 1M-line corpus is the
head of the 2.5M-line one.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PINNED = {("en", 1_000_000): (86_193_104, 3_311_102_221)}
GENERATED = {"en": 2_500_000, "tr": 5_000_000}      # lines generated per style; smaller cases use the head
_inputs = {}


def _corpus_and_vocab(style, max_len, n_lines):
    from sage_amd import synth
    key = (style, max_len)
    if key not in _inputs:
        _inputs.clear()                       # one corpus at a time in host memory
        seed = synth.DEFAULT_SEED
        head, head_off = synth.make_corpus(14_000, seed=seed, style=style, n_types=1_000_000)
        vocab = synth.make_vocab(head, head_off, 262_144, max_len, sample_bytes=6_000_000)
        corpus, off = synth.make_corpus(GENERATED[style], seed=seed, style=style, n_types=1_000_000)
        _inputs[key] = (vocab, corpus, off)
    vocab, corpus, off = _inputs[key]
    return vocab, corpus[:off[n_lines]], off[:n_lines + 1]


@pytest.mark.parametrize("style,max_len,dim,N_LINES", [("en", 16, 50, 1_000_000), ("en", 16, 300, 2_500_000), ("tr", 17, 50, 5_000_000)])
def test_full_size_workload_properties(style, max_len, dim, N_LINES):
    from sage_amd import synth
    from sage_amd._lib import HipSage
    from sage_amd.parallel import segment_line_bounds
    vocab, corpus, off = _corpus_and_vocab(style, max_len, N_LINES)
    eng = HipSage(0)
    try:
        eng.set_vocab(vocab, None, max_len)
        eng.set_corpus(corpus, off)
        per_line, total = eng.token_counts(0, N_LINES)
        counts = np.bincount(eng.tokenize(0, N_LINES)[0], minlength=len(vocab))
        emb = synth.make_embeddings(len(vocab), dim, seed=synth.DEFAULT_SEED, unseen=~(counts > 0))
        eng.set_embeddings(emb)
        bounds = segment_line_bounds(N_LINES, 4)
        loss, occ, tt, ttr, hist = eng.ablation_step(bounds)
        assert tt == total == int(counts.sum()) == int(per_line.sum())
        if (style, N_LINES) in PINNED:
            assert (tt, ttr) == PINNED[(style, N_LINES)]
        assert int(hist.sum()) == int(occ.sum()) and np.isfinite(loss).all()
        assert not np.any((occ > 0) & (counts == 0))          # only tokens the tokenization produces are ablated
        multi = np.array([len(v) > 1 for v in vocab])
        assert np.array_equal(occ[multi], counts[multi]) and not occ[~multi].any()   # every multi-byte token is ablated once
        st = eng.step_stats()
        assert st["tokens"] == tt and st["occurrences"] == int(occ.sum()) and st["emb_elem_bytes"] == 4
        assert st["records"] == 2 * st["pairs"] and 0 < st["window_rows"] < st["rows"]
        assert st["count_redos"] == 0 and st["count_from_hbm_sets"] == 0      # the partitioned count passed its own checks
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
        # float64 row storage (no float32 copy, no fix-up pass) scores the same
        eng.set_score_mode(1)
        eng.set_embeddings(emb)
        l5, o5, t5, r5, h5 = eng.ablation_step(bounds[:mid + 1])
        assert (t5, r5) == (t1, r1) and np.array_equal(o5, o1)
        np.testing.assert_allclose(l5, l1, rtol=1e-9, atol=1e-12 * np.abs(loss).max())
    finally:
        eng.close()
