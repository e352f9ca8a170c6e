"""CPU: the oracle restatement (oracle/sage_oracle.py) against vectors captured from the imported reference
(tests/golden/make_golden.py).  This is what pins the oracle (prompt §3 / SURVEY.md §8c)."""
import numpy as np
import pytest

from oracle import sage_oracle as orc
from sage_amd import synth
from tests.golden_io import load_json_gz, load_npz, unpack


@pytest.mark.parametrize("tag", ["en16", "tr17"])
def test_tokenize_matches_reference(tag):
    g = load_npz(f"tokenize_{tag}.npz")
    vocab = unpack(g["vocab_bytes"], g["vocab_off"])
    lines = unpack(g["line_bytes"], g["line_off"])
    tok = orc.OracleTokenizer(vocab, int(g["max_len"]))
    ids, starts, widths, counts = [], [], [], []
    for ln in lines:
        t = tok.tokenize(ln)
        counts.append(len(t))
        for a, b, c in t:
            ids.append(a); starts.append(b); widths.append(c)
    assert counts == g["counts"].tolist()
    assert ids == g["ids"].tolist()
    assert starts == g["starts"].tolist()
    assert widths == g["widths"].tolist()


@pytest.mark.parametrize("tag", ["en16", "tr17"])
def test_ablate_line_matches_fast_sage(tag):
    g = load_json_gz(f"fast_sage_{tag}.json.gz")
    vocab = [bytes.fromhex(h) for h in g["vocab"]]
    tok = orc.OracleTokenizer(vocab, g["max_len"])
    pads = set()
    for case in g["cases"]:
        triples, sizes = {}, {}
        ret = tok.ablate_line(bytes.fromhex(case["line"]), triples, sizes, pad=case["pad"])
        assert ret == case["ret"]
        # same keys in the same insertion order, same signed counts (zero-count keys included)
        assert [list(k) for k in triples.keys()] == case["keys"]
        assert list(triples.values()) == case["counts"]
        assert {str(k): v for k, v in sizes.items()} == case["sizes"]
        pads.add(case["pad"])
    assert pads == {1, 2, 3, 5}


@pytest.mark.parametrize("tag", ["en16", "tr17"])
def test_score_triples_matches_compute_losses(tag):
    g = load_npz(f"compute_losses_{tag}.npz")
    emb = synth.make_embeddings(int(g["emb_rows"]), int(g["emb_dim"]), seed=int(g["emb_seed"]))
    assert float(emb.sum()) == float(g["emb_checksum"])
    triples = {tuple(int(x) for x in k): int(c) for k, c in zip(g["keys"], g["counts"])}
    losses = {}
    orc.score_triples(losses, triples, emb)
    assert list(losses.keys()) == g["loss_ids"].tolist()
    np.testing.assert_allclose(np.array(list(losses.values())), g["loss_vals"], rtol=1e-12, atol=0)


def test_shard_sizes_match_divide_data_by_num():
    g = load_npz("parallel.npz")
    for (n, w), want in zip(g["shard_cases"].tolist(), g["shard_sizes"].tolist()):
        got = ",".join(str(hi - lo) for lo, hi in orc.shard_bounds(n, w))
        assert got == want, (n, w)


def test_segment_bounds_flush_rule():
    # first segment of a shard has flush+1 lines, later ones flush (utils.py:220)
    assert orc.segment_bounds(25, 1, flush=10) == [(0, 11), (11, 21), (21, 25)]
    assert orc.segment_bounds(11, 1, flush=10) == [(0, 11)]
    assert orc.segment_bounds(12, 1, flush=10) == [(0, 11), (11, 12)]
    assert orc.segment_bounds(8, 4) == [(0, 3), (3, 6), (6, 8)]
    segs = orc.segment_bounds(20600, 1)
    assert segs == [(0, 10001), (10001, 20001), (20001, 20600)]


@pytest.mark.parametrize("cs", [7, 20, 60])
def test_run_shard_small_flush(cs):
    g = load_npz("parallel.npz")
    vocab = unpack(g["vocab_bytes"], g["vocab_off"])
    lines = unpack(g["line_bytes"], g["line_off"][:62])
    emb = synth.make_embeddings(len(vocab), int(g["emb_dim"]), seed=int(g["emb_seed"]))
    assert float(emb.sum()) == float(g["emb_checksum"])
    tok = orc.OracleTokenizer(vocab, int(g["max_len"]))
    losses, tt, ttr, sizes = orc.run_shard(tok, lines, emb, flush=cs)
    assert tt == int(g[f"chunk{cs}_total_tokens"])
    assert ttr == int(g[f"chunk{cs}_total_triples"])
    assert sorted(losses.keys()) == g[f"chunk{cs}_loss_ids"].tolist()
    np.testing.assert_allclose([losses[k] for k in sorted(losses)], g[f"chunk{cs}_loss_vals"], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("workers", [3])
def test_run_parallel_matches_reference(workers):
    """The 20,600-line case (flush rule at 10,001 lines).  One W here keeps the CPU suite short; the C oracle
    test covers W=1,3,4 on the same fixture."""
    g = load_npz("parallel.npz")
    vocab = unpack(g["vocab_bytes"], g["vocab_off"])
    lines = unpack(g["line_bytes"], g["line_off"])
    emb = synth.make_embeddings(len(vocab), int(g["emb_dim"]), seed=int(g["emb_seed"]))
    tok = orc.OracleTokenizer(vocab, int(g["max_len"]))
    tt, ttr, losses, sizes = orc.run_parallel(emb, lines, tok, workers, processes=workers)
    assert tt == int(g[f"w{workers}_total_tokens"])
    assert ttr == int(g[f"w{workers}_total_triples"])
    assert sorted(losses.keys()) == g[f"w{workers}_loss_ids"].tolist()
    np.testing.assert_allclose([losses[k] for k in sorted(losses)], g[f"w{workers}_loss_vals"], rtol=1e-9, atol=1e-11)
    assert sorted(sizes.keys()) == g[f"w{workers}_size_keys"].tolist()
    assert [sizes[k] for k in sorted(sizes)] == g[f"w{workers}_size_vals"].tolist()


def test_oracle_c1_config_against_reference_files():
    """BASELINE.json configs[0] (10k lines, 16,384 keys, W=4): the oracle's step and prune selection against the
    reference's stats_8192.json, sorted_losses_before_8192.txt and deleted_vocab_8192.vocab (make_golden.py c1)."""
    import hashlib
    import json
    from sage_amd import synth
    from sage_amd.hf_bytes import HFEncoding
    from tests.golden_io import load_json_gz
    g = load_json_gz("c1_build_vocab.json.gz")
    V = g["vocab_size"]
    corpus, off = synth.make_corpus(g["n_lines"], seed=g["corpus_seed"], style="en", n_types=g["n_types"])
    assert hashlib.sha256(corpus.tobytes()).hexdigest() == g["corpus_sha256"]
    vocab = synth.make_vocab(corpus, off, V, g["max_len"], sample_bytes=6_000_000)
    emb = synth.make_embeddings(V, g["dim"], seed=g["emb_seed"])
    tok = orc.OracleTokenizer(vocab, g["max_len"])
    tt, ttr, losses, sizes = orc.run_parallel(emb, synth.corpus_lines(corpus, off), tok, g["workers"], processes=4)
    stats = json.loads(g["files"]["stats/stats_8192.json"])
    assert tt == stats["total_tokens"] and ttr == stats["total_triples"]
    assert {str(k): v for k, v in sizes.items()} == stats["ablated_sizes"]
    i, target, ranked, pruned = orc.select_prune(losses, tok, g["schedule"], 0)
    assert target == 8192 and len(ranked) == stats["current_active_vocab_size"] and len(pruned) == stats["num_tokens_to_prune"]
    enc = HFEncoding().to_encoded
    want = [ln.rsplit("\t", 1) for ln in g["files"]["sage_vocabs/sorted_losses_before_8192.txt"].splitlines()]
    assert [enc(tok.inverse[t]) for _, t in ranked] == [w[0] for w in want]
    np.testing.assert_allclose([v for v, _ in ranked], [float(w[1]) for w in want], rtol=1e-9, atol=1e-9)
    deleted = sorted(pruned)
    assert "".join(tok.inverse[t].hex() + "\n" for t in deleted) == g["files"]["sage_vocabs/deleted_vocab_8192.vocab"]
