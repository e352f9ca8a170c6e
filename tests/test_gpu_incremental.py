"""GPU: a vocabulary that only lost keys is applied to the resident match table instead of matching the corpus again
(sage_set_vocab_pruned).  Prune after prune -- random prunes, prunes of the longest keys, of the most frequent ones --
the carried-over table must tokenize every line exactly like a cold context given the same vocabulary, and the ablation
step must return the same integers (and losses to float64 rounding) without launching a matcher."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cold(vocab, max_len, corpus, off, emb, bounds):
    from sage_amd._lib import HipSage
    eng = HipSage(0)
    try:
        eng.set_vocab(vocab, None, max_len); eng.set_corpus(corpus, off); eng.set_embeddings(emb)
        tok = eng.tokenize(0, len(off) - 1)
        return tok, eng.ablation_step(bounds, vsize=len(vocab))
    finally:
        eng.close()


@pytest.mark.parametrize("style,max_len", [("en", 16), ("tr", 17)])
def test_pruned_vocabulary_moves_the_matches(style, max_len):
    from sage_amd import SaGeTokenizer, synth
    from sage_amd.parallel import segment_line_bounds, sync_vocab
    from sage_amd._lib import HipSage
    n = 4000
    corpus, off = synth.make_corpus(n, seed=31 + max_len, style=style, n_types=6000, mean_log_words=3.0)
    vocab = synth.make_vocab(corpus, off, 12_000, max_len, sample_bytes=2_000_000)
    emb = synth.make_embeddings(len(vocab), 50, seed=8)       # rows stay: stale-matrix semantics, ids index what is there
    bounds = segment_line_bounds(n, 2, flush=700)
    model = SaGeTokenizer(vocab, max_len)
    eng = HipSage(0)
    rng = np.random.default_rng(5)
    try:
        sync_vocab(eng, model)
        eng.set_corpus(corpus, off); eng.set_embeddings(emb)
        eng.ablation_step(bounds, vsize=model.id_space())     # a step over every line: the match table is complete
        assert eng.step_stats()["no_matcher"] == 0
        for rnd in range(5):
            keys = list(model.get_vocabulary().keys())
            multi = [k for k in keys if len(k) > 1]
            if rnd == 1:        # the longest keys: every match of theirs falls back to a shorter prefix
                doomed = set(sorted(multi, key=len, reverse=True)[:len(multi) // 6])
            elif rnd == 2:      # the most frequent ones
                counts = np.bincount(eng.tokenize(0, n)[0], minlength=len(keys))
                doomed = set(k for k in sorted(multi, key=lambda k: -counts[model.get_vocabulary()[k]])[:len(multi) // 8])
            else:
                doomed = set(multi[i] for i in rng.choice(len(multi), size=len(multi) // 5, replace=False))
            survivors = [k for k in keys if k not in doomed]
            model.set_vocabulary(survivors)
            assert model.prune_map is not None
            sync_vocab(eng, model)                            # -> sage_set_vocab_pruned
            got_tok = eng.tokenize(0, n)
            got = eng.ablation_step(bounds, vsize=len(emb))
            assert eng.step_stats()["no_matcher"] == 1, rnd
            want_tok, want = _cold(survivors, max_len, corpus, off, emb, bounds)
            for a, b in zip(got_tok, want_tok):
                assert np.array_equal(a, b), rnd
            V = len(survivors)
            assert (got[2], got[3]) == (want[2], want[3]), rnd
            assert np.array_equal(got[1][:V], want[1]) and not got[1][V:].any() and np.array_equal(got[4], want[4])
            np.testing.assert_allclose(got[0][:V], want[0], rtol=1e-9, atol=1e-12 * max(1.0, np.abs(want[0]).max()))
        # a vocabulary that GAINED a key is not a prune: the full path runs and the matcher is launched again
        model.set_vocabulary(list(model.get_vocabulary().keys()) + [b"zzzzqq"])
        assert model.prune_map is None
        sync_vocab(eng, model)
        eng.ablation_step(bounds, vsize=len(emb))
        assert eng.step_stats()["no_matcher"] == 0
    finally:
        eng.close()


def test_build_vocab_is_the_same_with_and_without_carried_matches(tmp_path, monkeypatch):
    """The whole builder loop (5 prunes of the reference golden) with the incremental path switched off gives the same
    files as with it on (which tests/test_gpu_build_vocab.py holds against the reference)."""
    import sage_amd
    from sage_amd import synth
    from sage_amd.parallel import get_engine
    from tests.golden_io import load_json_gz
    g = load_json_gz("build_vocab.json.gz")
    outs = []
    for tag, incr in (("on", "1"), ("off", "0")):
        monkeypatch.setenv("SAGE_INCREMENTAL_MATCH", incr)
        root = tmp_path / tag
        sage_amd.setSageFolder(root)
        emb_dir = root / "results" / "g" / "embeddings"
        emb_dir.mkdir(parents=True)
        for label, seed in g["emb_seeds"].items():
            np.save(emb_dir / f"embeddings_{label}.npy", synth.make_embeddings(g["emb_rows"], g["emb_dim"], seed=seed))
        (root / "initial.vocab").write_text("".join(h + "\n" for h in g["vocab"]))
        corpus = bytes.fromhex(g["corpus_hex"])
        (root / "corpus.txt").write_bytes(corpus)
        (root / "partial.txt").write_bytes(corpus)
        b = sage_amd.SaGeVocabBuilder(list(g["schedule"]), list(g["embeddings_schedule"]), max_len=g["max_len"],
                                      workers_number=g["workers"], random_seed=692653)
        b.deterministic = True
        b.build_vocab("g", str(root / "corpus.txt"), str(root / "initial.vocab"), str(root / "partial.txt"), 1000)
        res = root / "results" / "g"
        outs.append({p.name: p.read_text(encoding="utf-8") for d in ("sage_vocabs", "stats") for p in sorted((res / d).iterdir())})
        get_engine().set_deterministic(False)
    assert outs[0].keys() == outs[1].keys()
    for name in outs[0]:
        assert outs[0][name] == outs[1][name], name       # deterministic mode: bit-identical losses, identical text
