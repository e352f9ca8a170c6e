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
