"""CPU: host logic of the product (no GPU, no compute calls): file formats against the reference's golden files,
segment rule against the oracle, the C-ABI library loads and exports every symbol of include/sage_hip.h."""
import json
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import sage_oracle as orc
from tests.golden_io import load_json_gz, load_npz

REPO = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from sage_amd import _lib
    header = (REPO / "include" / "sage_hip.h").read_text()
    declared = set(re.findall(r"\b(sage_[a-z_]+)\s*\(", header))
    declared.discard("sage_ctx")
    assert declared == set(_lib.EXPORTED)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.sage_version().decode().count(".") == 2


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from sage_amd._lib import HipSage, SageHipError
    with pytest.raises(SageHipError):
        HipSage(0)


def test_product_does_not_import_oracle():
    for p in (REPO / "sage_amd").rglob("*.py"):
        text = p.read_text()
        assert "oracle" not in text.replace("sage_oracle", "oracle") or "import oracle" not in text, p
        assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), p


@pytest.mark.parametrize("n,w,flush", [(8, 4, 10000), (25, 1, 10), (11, 1, 10), (12, 1, 10), (20600, 1, 10000),
                                        (20600, 3, 10000), (20600, 4, 10000), (61, 1, 7), (3, 8, 10000), (1, 1, 5),
                                        (100, 7, 3)])
def test_segment_bounds_match_oracle(n, w, flush):
    from sage_amd.parallel import divide_data_by_num, segment_line_bounds
    b = segment_line_bounds(n, w, flush)
    segs = list(zip(b[:-1].tolist(), b[1:].tolist()))
    assert segs == orc.segment_bounds(n, w, flush)
    assert [len(x) for x in divide_data_by_num(list(range(n)), w)] == [hi - lo for lo, hi in orc.shard_bounds(n, w)]


def test_shard_sizes_match_reference_golden():
    from sage_amd.parallel import divide_data_by_num
    g = load_npz("parallel.npz")
    for (n, w), want in zip(g["shard_cases"].tolist(), g["shard_sizes"].tolist()):
        assert ",".join(str(len(b)) for b in divide_data_by_num(list(range(n)), w)) == want


def test_vocab_and_loss_file_formats(tmp_path):
    """write_vocab / load_vocab / save_sorted_losses / save_stats reproduce the reference's files byte for byte."""
    from sage_amd import hostio
    from sage_amd.tokenizer import SaGeTokenizer
    g = load_json_gz("build_vocab.json.gz")
    files = g["files"]
    vocab = [bytes.fromhex(h) for h in g["vocab"]]
    init = tmp_path / "initial.vocab"
    hostio.write_vocab({k: i for i, k in enumerate(vocab)}, init)
    assert init.read_text() == "".join(h + "\n" for h in g["vocab"])
    assert hostio.load_vocab(init) == vocab
    with pytest.raises(FileNotFoundError):
        hostio.load_vocab(tmp_path / "absent.vocab")
    # rebuild the first round's files from its golden sorted-loss file
    model = SaGeTokenizer(vocab, 16)                  # host state only; no GPU is touched until something is tokenized
    want = files["sage_vocabs/sorted_losses_before_2300.txt"]
    ranked = []
    for ln in want.splitlines():
        tok, val = ln.rsplit("\t", 1)
        ranked.append((float(val), model.str_vocab[tok]))
    hostio.save_sorted_losses(model, ranked, 2300, tmp_path)
    assert (tmp_path / "sorted_losses_before_2300.txt").read_text(encoding="utf-8") == want
    assert (tmp_path / "worst_500_2300.txt").read_text(encoding="utf-8") == files["sage_vocabs/worst_500_2300.txt"]
    assert (tmp_path / "best_500_2300.txt").read_text(encoding="utf-8") == files["sage_vocabs/best_500_2300.txt"]
    stats = json.loads(files["stats/stats_2300.json"])
    hostio.save_stats(stats, tmp_path, 2300)
    assert (tmp_path / "stats_2300.json").read_text() == files["stats/stats_2300.json"]
    # prune selection (SaGeVocabBuilder.py:146,159) from the golden ranking reproduces deleted_vocab
    n_prune = stats["num_tokens_to_prune"]
    doomed = {model.id_to_bytes(tid) for _, tid in sorted(ranked)[:n_prune]}
    deleted = {t: i for t, i in model.get_vocabulary().items() if t in doomed}
    hostio.write_vocab(deleted, tmp_path / "deleted.vocab")
    assert (tmp_path / "deleted.vocab").read_text() == files["sage_vocabs/deleted_vocab_2300.vocab"]


def test_tokenizer_vocab_state_matches_reference_rules():
    from sage_amd.hf_bytes import HFEncoding
    from sage_amd.tokenizer import SaGeTokenizer
    vocab = [bytes([i]) for i in range(256)] + [b"ab", b"abc", b"ab"]   # a repeated key keeps its LAST position
    t = SaGeTokenizer(vocab, 16)
    assert t.vocab_size() == 258 and t.get_vocabulary()[b"ab"] == 258 and t.id_to_bytes(257) == b"abc"
    keys, ids = t.key_table()
    assert dict(zip(keys, ids.tolist())) == t.get_vocabulary()
    flat, off, pids = t.packed_key_table()            # what the engine uploads: the same table, packed
    assert {flat[off[i]:off[i + 1]].tobytes(): int(pids[i]) for i in range(len(off) - 1)} == t.get_vocabulary()
    t2 = SaGeTokenizer(vocab[:-1], 16)                # no repeats: ids are the positions
    flat, off, pids = t2.packed_key_table()
    assert pids is None and [flat[off[i]:off[i + 1]].tobytes() for i in range(len(off) - 1)] == vocab[:-1]
    assert t2.id_space() == 258 and t.id_space() == 259 and t.str_vocab["ab"] == 258 and t.inv_str_vocab[257] == "abc"
    d = {}
    t.add_all_byte_ids(d, 1e6)
    assert len(d) == 256 and set(d.values()) == {1e6}
    with pytest.raises(Exception, match="missing byte"):
        SaGeTokenizer(vocab[1:], 16)
    with pytest.raises(AssertionError):
        SaGeTokenizer(vocab + ["str"], 16)
    assert HFEncoding().to_encoded(b" a\n") == "ĠaĊ"


def test_load_corpus_shuffle_and_cache(tmp_path):
    import random
    from sage_amd import hostio, setSageFolder
    setSageFolder(tmp_path)
    lines = [f"line {i}\n" for i in range(2500)]
    (tmp_path / "c.txt").write_text("".join(lines))
    random.seed(7)
    got = hostio.load_corpus(tmp_path / "c.txt", None, 2)
    random.seed(7)
    want = list(lines)
    random.shuffle(want)
    assert got == want[:2000]
    cached = tmp_path / "data" / "c_2000.txt"
    assert cached.read_text() == "".join(want[:2000])
    assert hostio.load_corpus(tmp_path / "c.txt", cached, 2) == want[:2000]


def test_partition_segments_balanced_and_whole():
    from sage_amd.dist import partition_segments
    from sage_amd.parallel import segment_line_bounds
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 400, size=53_123)
    off = np.concatenate([[0], np.cumsum(lens)])
    bounds = segment_line_bounds(len(lens), 4)
    for world in (1, 2, 3, 4, 8):
        parts = partition_segments(bounds, off, world)
        assert parts[0][0] == 0 and parts[-1][1] == len(bounds) - 1
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        sizes = [off[bounds[hi]] - off[bounds[lo]] for lo, hi in parts]
        assert sum(sizes) == off[-1]
        if world > 1:
            assert max(sizes) <= off[-1] / world + (off[bounds[1:]] - off[bounds[:-1]]).max()


def _load_corpus_fixture(tmp_path):
    import hashlib
    from sage_amd import synth
    g = load_json_gz("load_corpus.json.gz")
    a = g["corpus_args"]
    corpus, _ = synth.make_corpus(a["n_lines"], seed=g["corpus_seed"], style=a["style"], n_types=a["n_types"],
                                  mean_log_words=a["mean_log_words"], sigma_log_words=a["sigma_log_words"])
    text = corpus.tobytes()[:-1]               # the last line has no newline
    assert hashlib.sha256(text).hexdigest() == g["corpus_sha256"]
    (tmp_path / "wiki.txt").write_bytes(text)
    return g


@pytest.mark.parametrize("packed", [True, False])
def test_load_corpus_matches_reference_fixture(tmp_path, packed):
    """utils.py:71-103 captured from the reference (tests/golden/make_golden.py load_corpus): same lines in the same
    order under the same seed, file-identical partial corpus, same reload (where a line without a newline merges with
    its successor) -- for the byte-buffer loader build_vocab uses and for the List[str] form."""
    import hashlib
    import random
    from sage_amd import hostio, setSageFolder
    g = _load_corpus_fixture(tmp_path)
    for case in g["cases"]:
        root = tmp_path / f"run_{case['seed']}_{packed}"
        root.mkdir()
        setSageFolder(root)
        random.seed(case["seed"])
        loader = hostio.load_corpus_packed if packed else hostio.load_corpus
        got = loader(tmp_path / "wiki.txt", None, case["k"])
        lines = [got[i] for i in range(len(got))]
        assert lines == case["lines"]
        saved = root / "data" / case["file_name"]
        assert hashlib.sha256(saved.read_bytes()).hexdigest() == case["file_sha256"]
        again = loader(tmp_path / "wiki.txt", saved, case["k"])
        assert len(again) == case["reload_n"] and again[len(again) - 1] == case["reload_last"]
        if packed:
            assert int(got.offsets[-1]) == len(got.data) == sum(len(s.encode()) for s in case["lines"])


def test_load_corpus_packed_translates_carriage_returns(tmp_path):
    import random
    from sage_amd import hostio, setSageFolder
    setSageFolder(tmp_path)
    (tmp_path / "c.txt").write_bytes(b"one\r\ntwo\rthree\nfour")
    random.seed(1)
    want = hostio.load_corpus(tmp_path / "c.txt", tmp_path / "p1.txt", 1)
    random.seed(1)
    got = hostio.load_corpus_packed(tmp_path / "c.txt", tmp_path / "p2.txt", 1)
    assert [got[i] for i in range(len(got))] == want and sorted(want) == ["four", "one\n", "three\n", "two\n"]
    assert (tmp_path / "p1.txt").read_bytes() == (tmp_path / "p2.txt").read_bytes()
    empty = tmp_path / "empty.txt"
    empty.write_bytes(b"")
    assert len(hostio.load_corpus_packed(empty, tmp_path / "p3.txt", 1)) == 0


def test_near_ties_at_cut():
    from sage_amd.builder import near_ties_at_cut
    ranked = sorted([(-5.0, 1), (-3.0, 2), (-3.0 + 1e-12, 3), (-3.0 + 2e-12, 7), (-1.0, 4), (0.5, 5), (1e6, 0)])
    assert near_ties_at_cut(ranked, 1) == ([], [])                       # -5 | -3: clear
    low, high = near_ties_at_cut(ranked, 2)                               # -3 | -3+1e-12: a near-tie
    assert low == [(-3.0, 2)] and high == [(-3.0 + 1e-12, 3), (-3.0 + 2e-12, 7)]
    low, high = near_ties_at_cut(ranked, 3)
    assert [t for _, t in low] == [2, 3] and [t for _, t in high] == [7]
    assert near_ties_at_cut(ranked, 4) == ([], []) and near_ties_at_cut(ranked, 0) == ([], [])
    assert near_ties_at_cut(ranked, len(ranked)) == ([], [])
    exact = sorted([(-2.0, 9), (-2.0, 3), (4.0, 1)])
    assert near_ties_at_cut(exact, 1) == ([(-2.0, 3)], [(-2.0, 9)])      # exact tie: the id decides, still reported


def test_packed_corpus_behaves_like_a_list_of_lines():
    from sage_amd.parallel import PackedCorpus, pack_lines
    lines = ["ab\n", "çd e\n", "last"]
    flat, off = pack_lines(lines)
    pc = PackedCorpus(flat, off)
    assert len(pc) == 3 and [pc[i] for i in range(3)] == lines and pc[-1] == "last"
    sub = pc[1:3]
    assert isinstance(sub, PackedCorpus) and [sub[i] for i in range(len(sub))] == lines[1:3]
    assert pc[::2] == lines[::2]
    assert pc.line_bytes(1) == "çd e\n".encode()


def test_bench_shards_share_one_lexicon():
    """bench.py gives rank r the lines of `line_seed=r`: other lines of the SAME lexicon (the vocabulary is built from
    a sample of rank 0's text; text of another lexicon tokenizes into 1.7x the tokens and made rank >= 1 a different
    workload).  line_seed=0 is the single-GPU corpus unchanged."""
    from sage_amd import synth
    a, ao = synth.make_corpus(3000, seed=7, n_types=4000)
    b, bo = synth.make_corpus(3000, seed=7, n_types=4000, line_seed=0)
    c, co = synth.make_corpus(3000, seed=7, n_types=4000, line_seed=1)
    assert np.array_equal(a, b) and np.array_equal(ao, bo)
    assert not np.array_equal(a[:4096], c[:4096])
    words = lambda buf: set(buf.tobytes().lower().replace(b".\n", b" ").split())
    wa, wc = words(a), words(c)
    assert len(wa & wc) > 0.6 * min(len(wa), len(wc))          # same word types
    d, _ = synth.make_corpus(3000, seed=8, n_types=4000)
    assert len(wa & words(d)) < 0.2 * len(wa)                  # another seed is another language


def test_tokenizer_remembers_a_pure_prune():
    """set_vocabulary notes when the new vocabulary only lost keys (SaGeVocabBuilder.py:159-192) and keeps the id map the
    engine needs to move its resident matches (sage_set_vocab_pruned); anything else is a full rebuild."""
    from sage_amd import SaGeTokenizer
    base = [bytes([b]) for b in range(256)]
    vocab = base + [b"ab", b"abc", b"abcd", b"xy", b"xyz"]
    m = SaGeTokenizer(vocab, 8)
    assert m.prune_map is None and m.vocab_version == 1
    m.set_vocabulary(base + [b"ab", b"abcd", b"xyz"])          # pruned: abc, xy -- ids re-compacted
    version, o2n = m.prune_map
    assert version == 1 and len(o2n) == len(vocab)
    assert o2n[:256].tolist() == list(range(256))
    assert o2n[256:].tolist() == [256, -1, 257, -1, 258]
    m.set_vocabulary(base + [b"ab", b"abcd", b"xyz", b"new"])  # gained a key: not a prune
    assert m.prune_map is None
    m.set_vocabulary(base + [b"abcd", b"ab"])                  # lost one and REORDERED the rest: still only lost keys
    version, o2n = m.prune_map
    assert version == 3 and o2n[256:].tolist() == [257, 256, -1, -1]
    with pytest.raises(Exception, match="missing byte"):
        m.set_vocabulary(base[1:] + [b"ab"])
