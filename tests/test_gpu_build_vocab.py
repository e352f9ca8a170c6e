"""GPU: the full SaGeVocabBuilder.build_vocab loop against every file the reference wrote for the same inputs
(tests/golden/build_vocab.json.gz): pruned vocabularies identical at every schedule step, integer stats identical,
sorted-loss order identical and values within 1e-5 relative.  The embeddings_schedule skips steps, so this pins
the stale-matrix + id-compaction semantics (SaGeVocabBuilder.py:89-91,192)."""
import json
import os

import numpy as np
import pytest

from tests.golden_io import load_json_gz

pytestmark = pytest.mark.gpu


def test_build_vocab_matches_reference(tmp_path):
    import sage_amd
    from sage_amd import synth
    g = load_json_gz("build_vocab.json.gz")
    sage_amd.setSageFolder(tmp_path)
    exp = "golden"
    emb_dir = tmp_path / "results" / exp / "embeddings"
    emb_dir.mkdir(parents=True)
    for label, seed in g["emb_seeds"].items():
        np.save(emb_dir / f"embeddings_{label}.npy", synth.make_embeddings(g["emb_rows"], g["emb_dim"], seed=seed))
    vocab_file = tmp_path / "initial.vocab"
    vocab_file.write_text("".join(h + "\n" for h in g["vocab"]))
    corpus = bytes.fromhex(g["corpus_hex"])
    (tmp_path / "corpus.txt").write_bytes(corpus)
    (tmp_path / "partial.txt").write_bytes(corpus)
    builder = sage_amd.SaGeVocabBuilder(list(g["schedule"]), list(g["embeddings_schedule"]), max_len=g["max_len"],
                                        workers_number=g["workers"], random_seed=692653)
    builder.build_vocab(exp, str(tmp_path / "corpus.txt"), str(vocab_file), str(tmp_path / "partial.txt"), 1000)

    res = tmp_path / "results" / exp
    produced = sorted(f"sage_vocabs/{p.name}" for p in (res / "sage_vocabs").iterdir()) + \
        sorted(f"stats/{p.name}" for p in (res / "stats").iterdir()) + ["seed.txt"]
    assert sorted(produced) == sorted(g["files"].keys())
    for name, want in g["files"].items():
        got = (res / name).read_text(encoding="utf-8")
        if name.endswith(".vocab") or name == "seed.txt":
            assert got == want, name                      # pruned / active / deleted vocabularies: identical
        elif name.endswith(".json"):
            a, b = json.loads(got), json.loads(want)
            assert a == b, name                           # every field is an integer or a dict of integers
            assert list(a.keys()) == list(b.keys())
        else:                                             # sorted_losses_before_ / worst_500_ / best_500_
            gl, wl = got.splitlines(), want.splitlines()
            assert len(gl) == len(wl), name
            gt, gv = zip(*(ln.rsplit("\t", 1) for ln in gl))
            wt, wv = zip(*(ln.rsplit("\t", 1) for ln in wl))
            assert gt == wt, name                         # identical ranking (ties broken by id)
            np.testing.assert_allclose(np.array(gv, dtype=float), np.array(wv, dtype=float), rtol=1e-5, atol=1e-9)


def test_run_sage_parallel_api_shapes():
    """The drop-in call: fresh caller-owned dicts with int keys / float values, only occurring multi-byte ids."""
    from sage_amd import SaGeTokenizer, synth
    from sage_amd.parallel import run_sage_parallel
    corpus, off = synth.make_corpus(60, seed=3, n_types=800, mean_log_words=2.4)
    vocab = synth.make_vocab(corpus, off, 900, 16, sample_bytes=50_000)
    model = SaGeTokenizer(vocab, 16)
    lines = synth.corpus_lines(corpus, off)
    emb = synth.make_embeddings(len(vocab), 50, seed=9)
    tt, ttr, losses, sizes = run_sage_parallel(emb, lines, model, 4)
    assert isinstance(tt, int) and isinstance(ttr, int)
    assert all(type(k) is int and type(v) is float for k, v in losses.items())
    assert all(type(k) is int and type(v) is int for k, v in sizes.items())
    assert all(len(model.id_to_bytes(k)) > 1 for k in losses)
    # tokenizer API on the GPU
    toks = model.tokenize(lines[0])
    assert sum(w for _, _, w in toks) == len(lines[0].encode())
    assert model.tokenize(lines[0], tokens_only=True) == [t[0] for t in toks]
    assert b"".join(model.tokenize_to_bytes(lines[0])) == lines[0].encode()
    assert model.hfe.to_bytes("".join(model.tokenize_to_encoded_str(lines[0]))) == lines[0].encode()
    # a missing single byte is the reference's error
    with pytest.raises(Exception, match="missing byte"):
        SaGeTokenizer(vocab[1:], 16)
    # an out-of-range row is numpy's IndexError in the reference (utils.py:133)
    with pytest.raises(IndexError):
        run_sage_parallel(emb[:300], lines, model, 1)


@pytest.mark.parametrize("tag", ["en16", "tr17"])
@pytest.mark.parametrize("packed", [False, True])
def test_gensim_corpus_file_matches_reference(tmp_path, tag, packed):
    """data/gensim_<V>.txt (embeddings.py:53-66, model.py:103-107, HFEncoding.py:38-39) byte for byte against the file
    the reference wrote for the same lines (tests/golden/make_golden.py gensim_file)."""
    from sage_amd import SaGeTokenizer
    from sage_amd.embeddings import write_gensim_corpus
    from sage_amd.parallel import PackedCorpus, pack_lines
    from tests.golden_io import load_npz, unpack
    g = load_json_gz("gensim_file.json.gz")[tag]
    t = load_npz(f"tokenize_{tag}.npz")
    model = SaGeTokenizer(unpack(t["vocab_bytes"], t["vocab_off"]), int(t["max_len"]))
    assert model.vocab_size() == g["vocab_size"]
    corpus = PackedCorpus(*pack_lines(g["lines"])) if packed else g["lines"]
    out = tmp_path / "gensim.txt"
    write_gensim_corpus(model, corpus, out, lines_per_call=37)      # several GPU calls, a ragged last one
    assert out.read_text(encoding="utf-8") == g["text"]


def test_c1_config_matches_reference(tmp_path):
    """BASELINE.json configs[0] exactly (10k-line partial corpus, 16,384 -> 8,192 keys, one ablation step, D=50,
    workers=4) against the files the reference wrote (tests/golden/make_golden.py c1)."""
    import hashlib
    import sage_amd
    from sage_amd import synth
    g = load_json_gz("c1_build_vocab.json.gz")
    V = g["vocab_size"]
    corpus, off = synth.make_corpus(g["n_lines"], seed=g["corpus_seed"], style="en", n_types=g["n_types"])
    assert hashlib.sha256(corpus.tobytes()).hexdigest() == g["corpus_sha256"]
    vocab = synth.make_vocab(corpus, off, V, g["max_len"], sample_bytes=6_000_000)
    assert hashlib.sha256(b"\n".join(v.hex().encode() for v in vocab)).hexdigest() == g["vocab_sha256"]
    sage_amd.setSageFolder(tmp_path)
    exp = "c1"
    emb_dir = tmp_path / "results" / exp / "embeddings"
    emb_dir.mkdir(parents=True)
    np.save(emb_dir / f"embeddings_{V}.npy", synth.make_embeddings(V, g["dim"], seed=g["emb_seed"]))
    vocab_file = tmp_path / "initial.vocab"
    vocab_file.write_text("".join(v.hex() + "\n" for v in vocab))
    (tmp_path / "corpus.txt").write_bytes(corpus.tobytes())
    (tmp_path / "partial.txt").write_bytes(corpus.tobytes())
    builder = sage_amd.SaGeVocabBuilder(list(g["schedule"]), list(g["embeddings_schedule"]), max_len=g["max_len"],
                                        workers_number=g["workers"], random_seed=692653, word2vec_d=g["dim"],
                                        word2vec_window_size=5)
    builder.build_vocab(exp, str(tmp_path / "corpus.txt"), str(vocab_file), str(tmp_path / "partial.txt"), 10)
    res = tmp_path / "results" / exp
    for name, want in g["files"].items():
        got = (res / name).read_text(encoding="utf-8")
        if name.endswith(".vocab"):
            assert got == want, name
        elif name.endswith(".json"):
            assert json.loads(got) == json.loads(want), name
        else:
            gl, wl = got.splitlines(), want.splitlines()
            assert len(gl) == len(wl), name
            gt, gv = zip(*(ln.rsplit("\t", 1) for ln in gl))
            wt, wv = zip(*(ln.rsplit("\t", 1) for ln in wl))
            np.testing.assert_allclose(np.array(gv, dtype=float), np.array(wv, dtype=float), rtol=1e-5, atol=1e-9)
            assert gt == wt, name
    assert not builder.near_ties, builder.near_ties     # the identical cut above is not a coin toss


def test_keys_longer_than_max_len_are_ignored_not_refused():
    """A vocabulary may hold keys longer than max_len (the reference only warns, SaGeVocabBuilder.py:52-54) and longer
    than the library's 32-byte key limit: they keep their ids and never match (model.py:90)."""
    from oracle import sage_oracle as orc
    from sage_amd import SaGeTokenizer, synth
    from sage_amd.parallel import run_sage_parallel
    corpus, off = synth.make_corpus(80, seed=4, n_types=600, mean_log_words=2.3)
    vocab = synth.make_vocab(corpus, off, 700, 20, sample_bytes=60_000)
    lines = synth.corpus_lines(corpus, off)
    # keys that DO occur in the text (line starts) but are longer than max_len = 12, one beyond the 32-byte table limit
    extra = [lines[0].encode()[:40], lines[1].encode()[:14], lines[2].encode()[:16], lines[3].encode()[:13], b" " * 64]
    vocab = vocab[:300] + [k for k in extra if k not in vocab] + vocab[300:]
    assert max(len(v) for v in vocab) == 64 and sum(len(v) > 12 for v in vocab) >= 5
    model = SaGeTokenizer(vocab, 12)
    emb = synth.make_embeddings(len(vocab), 50, seed=2)
    tt, ttr, losses, sizes = run_sage_parallel(emb, lines, model, 2)
    tok = orc.OracleTokenizer(vocab, 12)
    wt, wr, wl, ws = orc.run_parallel(emb, lines, tok, 2)
    assert (tt, ttr, sizes) == (wt, wr, ws) and sorted(losses) == sorted(wl)
    np.testing.assert_allclose([losses[k] for k in sorted(wl)], [wl[k] for k in sorted(wl)], rtol=1e-5, atol=1e-9)
    assert model.tokenize(lines[0]) == tok.tokenize(lines[0].encode())
