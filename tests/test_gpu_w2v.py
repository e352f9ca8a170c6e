"""GPU: word2vec retraining on the GPU (sage_train_embeddings).

PARITY UNPINNED: the reference trains with gensim 4.3.2 (absent here, and lock-free training is not reproducible on
either side).  What is checked: the token counts are exactly those of the tokenization, the output layout is that of
embeddings.py:77-89, and the vectors have learnt the co-occurrence structure of a corpus built to have one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _topic_corpus(n_lines=3000, words_per_line=12, seed=5):
    rs = np.random.RandomState(seed)
    topic_a = [f"{w} ".encode() for w in ("apple", "pear", "plum", "grape", "melon", "peach", "cherry", "lemon", "mango", "fig")]
    topic_b = [f"{w} ".encode() for w in ("bolt", "screw", "nail", "hammer", "wrench", "drill", "saw", "plier", "clamp", "rivet")]
    unused = [b"zebra ", b"quartz "]
    vocab = [bytes([i]) for i in range(256)] + topic_a + topic_b + unused
    lines = []
    for i in range(n_lines):
        words = topic_a if i % 2 == 0 else topic_b
        lines.append(b"".join(words[j] for j in rs.randint(0, len(words), words_per_line)) + b"\n")
    return vocab, lines, range(256, 266), range(266, 276), range(276, 278)


def _cos(u, v):
    return float(u @ v / (np.linalg.norm(u) * np.linalg.norm(v) + 1e-30))


def _within_across(vec, ids_a, ids_b):
    words = list(ids_a) + list(ids_b)
    v = vec - vec[words].mean(axis=0)
    within = np.mean([_cos(v[i], v[j]) for i in ids_a for j in ids_a if i < j] +
                     [_cos(v[i], v[j]) for i in ids_b for j in ids_b if i < j])
    across = np.mean([_cos(v[i], v[j]) for i in ids_a for j in ids_b])
    return within, across


def test_w2v_counts_and_structure():
    from sage_amd._lib import HipSage
    from sage_amd.parallel import pack_lines
    vocab, lines, ids_a, ids_b, ids_unused = _topic_corpus()
    eng = HipSage(0)
    eng.set_vocab(vocab, None, 16)
    flat, off = pack_lines(lines)
    eng.set_corpus(flat, off)
    D = 32
    # sample=0: every word of this 21-word corpus is 'frequent'; gensim's default subsampling would keep 16 % of it
    vec, counts = eng.train_embeddings(len(vocab), D, window=5, negative=5, alpha=0.025, epochs=5, sample=0.0, min_count=1, seed=7)
    ids = eng.tokenize(0, len(lines))[0]
    assert counts.tolist() == np.bincount(ids, minlength=len(vocab)).tolist()      # exact: the vocabulary scan
    assert vec.shape == (len(vocab), D) and np.isfinite(vec).all()
    assert np.array_equal(vec, vec.astype(np.float32).astype(np.float64))          # float32 vectors widened, as gensim's
    # trained rows moved away from the U(-1,1)/D initialisation; rows of absent tokens did not
    assert np.abs(vec[list(ids_a)]).max() > 1.5 / D
    assert np.abs(vec[list(ids_unused)]).max() <= 1.0 / D + 1e-7
    # words of one topic share contexts, words of different topics never do (cosines after removing the direction all
    # 20 words share: with so small a vocabulary SGNS input vectors are dominated by a common component)
    within, across = _within_across(vec, ids_a, ids_b)
    assert within > across + 0.3, (within, across)
    # small batches (batch seams, several kernel launches per epoch) give the same kind of result
    eng.set_batch_bytes(8192)
    vec2, counts2 = eng.train_embeddings(len(vocab), D, window=5, negative=5, epochs=3, sample=0.0, seed=7)
    assert counts2.tolist() == counts.tolist()
    within2, across2 = _within_across(vec2, ids_a, ids_b)
    assert within2 > across2 + 0.3, (within2, across2)
    with pytest.raises(Exception):
        eng.train_embeddings(len(vocab), D, window=0)
    eng.close()


def test_get_embeddings_hip_backend(tmp_path, monkeypatch):
    """The host mirror of get_embeddings / train_embeddings (embeddings.py:28-89) with the HIP back end: matrix layout,
    uniform fallback rows for tokens the corpus never produces, checkpoint file reused on the second call."""
    import sage_amd
    from sage_amd import SaGeTokenizer
    from sage_amd.embeddings import get_embeddings
    from sage_amd.w2v_params import Word2VecParams
    monkeypatch.setenv("SAGE_W2V_BACKEND", "hip")
    sage_amd.setSageFolder(tmp_path)
    vocab, lines, ids_a, ids_b, ids_unused = _topic_corpus(n_lines=600)
    model = SaGeTokenizer(vocab, 16)
    corpus = [ln.decode() for ln in lines]
    params = Word2VecParams(D=20, N=5, ALPHA=0.025, window_size=5, min_count=1, sg=1)
    np.random.seed(11)
    emb = get_embeddings(len(vocab), tmp_path, corpus, model, 2, params)
    assert emb.shape == (len(vocab), 20) and emb.dtype == np.float64 and np.isfinite(emb).all()
    assert (tmp_path / f"embeddings_{len(vocab)}.npy").exists()
    half = 0.5 / 20
    assert np.abs(emb[list(ids_unused)]).max() <= half                            # embeddings.py:86
    assert not np.array_equal(emb[276], emb[276].astype(np.float32).astype(np.float64))   # a float64 draw, not a vector
    assert np.abs(emb[list(ids_a)]).max() > half
    again = get_embeddings(len(vocab), tmp_path, corpus, model, 2, params)         # embeddings.py:31-35: loaded, not retrained
    assert np.array_equal(again, emb)
    # and the ablation step accepts it as is
    from sage_amd.parallel import run_sage_parallel
    tt, ttr, losses, sizes = run_sage_parallel(emb, corpus, model, 2)
    assert tt > 0 and ttr > 0 and all(np.isfinite(v) for v in losses.values())


def test_build_vocab_end_to_end_without_gensim(tmp_path, monkeypatch):
    """The whole SaGeVocabBuilder.build_vocab loop with no pre-seeded embeddings and no gensim: every retrain label of
    the embeddings_schedule is trained on the GPU, the run leaves the reference's files behind and reaches its target."""
    import sage_amd
    from sage_amd import synth
    monkeypatch.setenv("SAGE_W2V_BACKEND", "hip")
    sage_amd.setSageFolder(tmp_path)
    corpus, off = synth.make_corpus(300, seed=21, n_types=1200, mean_log_words=2.6)
    vocab = synth.make_vocab(corpus, off, 1200, 16, sample_bytes=200_000)
    (tmp_path / "initial.vocab").write_text("".join(v.hex() + "\n" for v in vocab))
    (tmp_path / "corpus.txt").write_bytes(corpus.tobytes())
    builder = sage_amd.SaGeVocabBuilder([1200, 1000, 800, 600], [1200, 800], max_len=16, workers_number=2, random_seed=692653)
    builder.build_vocab("e2e", str(tmp_path / "corpus.txt"), str(tmp_path / "initial.vocab"), str(tmp_path / "partial.txt"), 300)
    res = tmp_path / "results" / "e2e"
    assert (res / "embeddings" / "embeddings_1200.npy").exists()
    assert not (res / "embeddings" / "embeddings_1000.npy").exists()      # not in the embeddings_schedule: stale matrix reused
    final = (res / "sage_vocabs" / "sage_vocab_600.vocab").read_text().splitlines()
    # (tokens the corpus never produces are not 'active' and survive the cut: the file is larger than its label)
    assert 600 <= len(final) < 1200 and all(bytes([b]).hex() in final for b in range(256))
    emb = np.load(res / "embeddings" / "embeddings_1200.npy")
    assert emb.shape == (1200, 50) and np.isfinite(emb).all()
    print(sorted(p.name for p in (res / "embeddings").iterdir()), sorted(p.name for p in (res / "sage_vocabs").iterdir())[:40])
