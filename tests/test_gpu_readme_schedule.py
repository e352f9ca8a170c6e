"""GPU: BASELINE.json configs[2] (C3) -- the reference's WHOLE README schedule (README.md:11-13, run.sh:1-10):
262144 -> 229376 -> ... -> 16384, eleven consecutive prunes of a 262,144-key vocabulary, embeddings retrained at six
labels (rounds on stale rows of an older matrix after ids were re-compacted alternate with rounds on a fresh one),
max_len 17, workers_number 4, against what the REAL reference produced for the same inputs
(tests/golden/readme_schedule.json.gz, written by tests/golden/make_readme_schedule.py): every pruned / active / deleted
vocabulary identical, every stats file identical, the ranking of the losses identical (head, tail, every 64th rank and --
where the reference's own neighbours are further apart than float noise -- the whole token order), sampled values within
1e-5 relative, every retrained embedding matrix and every gensim corpus file identical to the byte.  word2vec itself is
played by tests/fake_gensim.py on both sides (gensim is not installed and is not reproducible).

Three settings: one rank; two ranks sharing cuda:0 (the product's sharding, combine and broadcast paths); and
deterministic mode (fixed-point loss sums)."""
import hashlib
import json
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.golden_io import GOLDEN, front_decode, load_json_gz, load_npz

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
_cache = {}


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def _inputs():
    if "in" not in _cache:
        from sage_amd import synth
        g = load_json_gz("readme_schedule.json.gz")
        v = load_npz("readme_schedule_vocab.npz")
        multi = front_decode(v["prefix_len"], v["suffix_len"], v["suffix_bytes"])
        perm = np.random.default_rng(int(v["perm_seed"])).permutation(len(multi))
        vocab = [bytes([b]) for b in range(256)] + [multi[i] for i in perm]
        assert sha(b"\n".join(k.hex().encode() for k in vocab)) == g["vocab_sha256"]
        corpus, off = synth.make_corpus(g["n_lines"], seed=g["corpus_seed"], style=g["style"], n_types=g["n_types"],
                                        mean_log_words=g["mean_log_words"])
        assert sha(corpus.tobytes()) == g["corpus_sha256"]
        _cache["in"] = (g, vocab, corpus.tobytes())
    return _cache["in"]


def _prepare(root: Path, vocab, corpus: bytes):
    root.mkdir(parents=True, exist_ok=True)
    (root / "initial.vocab").write_text("".join(k.hex() + "\n" for k in vocab))
    (root / "corpus.txt").write_bytes(corpus)
    (root / "partial.txt").write_bytes(corpus)        # a pre-existing partial corpus: no shuffle (utils.py:75-80)


def _build(root: Path, g, deterministic: bool):
    import sage_amd
    from tests import fake_gensim
    fake_gensim.install()
    os.environ["SAGE_W2V_BACKEND"] = "gensim"
    sage_amd.setSageFolder(root)
    b = sage_amd.SaGeVocabBuilder(list(g["schedule"]), list(g["embeddings_schedule"]), max_len=g["max_len"],
                                  workers_number=g["workers"], random_seed=g["random_seed"])
    b.deterministic = deterministic
    import time
    t0 = time.time()
    try:
        b.build_vocab("c3", str(root / "corpus.txt"), str(root / "initial.vocab"), str(root / "partial.txt"), g["n_lines"] // 1000)
        # (the reference took 1,933 s for this schedule with 4 worker processes on the 8 cores of the build container,
        # tests/golden/make_readme_schedule.py; both sides include the stand-in word2vec's six passes over the corpus file)
        print(f"build_vocab of the README schedule: {time.time() - t0:.1f} s (deterministic={deterministic})", flush=True)
    finally:
        os.environ.pop("SAGE_W2V_BACKEND", None)
        fake_gensim.uninstall()
        from sage_amd.parallel import get_engine
        get_engine().set_deterministic(False)
    return b


def _check(root: Path, g, near_ties):
    res = root / "results" / "c3"
    got_files = sorted(p.name for p in (res / "sage_vocabs").iterdir())
    want_files = sorted(list(g["vocab_files"]) + list(g["sorted_losses"]) + list(g["head_tail"]))
    assert got_files == want_files
    for name, want in g["vocab_files"].items():                       # pruned / active / deleted vocabularies: identical
        text = (res / "sage_vocabs" / name).read_text(encoding="utf-8")
        assert text.count("\n") == want["lines"], name
        assert sha(text.encode("utf-8")) == want["sha256"], name
    assert sorted(p.name for p in (res / "stats").iterdir()) == sorted(g["stats"])
    for name, want in g["stats"].items():                             # every field an integer (total_triples, ablated_sizes ...)
        got = json.loads((res / "stats" / name).read_text())
        assert got == json.loads(want), name
    assert (res / "seed.txt").read_text() == g["seed_txt"]
    for name, want in g["embeddings"].items():                        # retrained matrices: rows = the vocabulary at that label
        a = np.load(res / "embeddings" / name)
        assert list(a.shape) == want["shape"], name
        assert sha(a.tobytes()) == want["sha256"], name
    for name, want in g["gensim_files"].items():
        assert sha((root / "data" / name).read_bytes()) == want["sha256"], name
    for name, want in g["head_tail"].items():                         # worst_500_ / best_500_: the ends of the ranking
        gl, wl = (res / "sage_vocabs" / name).read_text(encoding="utf-8").splitlines(), want.splitlines()
        gt, gv = zip(*(ln.rsplit("\t", 1) for ln in gl))
        wt, wv = zip(*(ln.rsplit("\t", 1) for ln in wl))
        assert gt == wt, name
        np.testing.assert_allclose(np.array(gv, dtype=float), np.array(wv, dtype=float), rtol=1e-5, atol=1e-9)
    for name, want in g["sorted_losses"].items():
        lines = (res / "sage_vocabs" / name).read_text(encoding="utf-8").splitlines()
        assert len(lines) == want["lines"], name
        toks, vals = zip(*(ln.rsplit("\t", 1) for ln in lines))
        v = np.array(vals, dtype=np.float64)
        real = v[v < 1e5]
        np.testing.assert_allclose(real.sum(), want["sum"], rtol=1e-7, atol=1e-7 * want["abs_sum"])
        np.testing.assert_allclose(np.abs(real).sum(), want["abs_sum"], rtol=1e-8)
        rank_of = {t: i for i, t in enumerate(toks)}
        for i, tok, val in want["sample"]:                            # every 64th rank: same token (give or take a float tie), same value
            j = rank_of[tok]
            assert abs(j - i) <= 2, (name, i, j, tok)
            assert abs(v[j] - float(val)) <= 1e-5 * abs(float(val)) + 1e-9, (name, tok, v[j], val)
        if want["min_rel_gap"] > 1e-8:                                # the reference's own neighbours are further apart than float noise
            assert sha("\n".join(toks).encode("utf-8")) == want["tokens_sha256"], name
    assert not near_ties, near_ties                                   # no cut was a coin toss


@pytest.mark.parametrize("deterministic", [False, True])
def test_readme_schedule_on_one_rank(tmp_path, deterministic):
    g, vocab, corpus = _inputs()
    _prepare(tmp_path, vocab, corpus)
    b = _build(tmp_path, g, deterministic)
    _check(tmp_path, g, b.near_ties)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, root):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_json_gz("readme_schedule.json.gz")
    b = _build(Path(root), g, False)
    (Path(root) / f"near_ties_rank{rank}.json").write_text(json.dumps({str(k): len(v[0]) for k, v in b.near_ties.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_readme_schedule_on_two_ranks(tmp_path):
    """The same through `run_sage_distributed`: two ranks share cuda:0 (gloo; under "nccl" the same code hands the CUDA
    tensors to RCCL), each uploads its own count segments, the loss vectors are combined in rank order, rank 0 alone
    writes files and "trains", and both take the same eleven prune decisions."""
    import torch.multiprocessing as mp
    g, vocab, corpus = _inputs()
    _prepare(tmp_path, vocab, corpus)
    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    ties = [json.loads((tmp_path / f"near_ties_rank{r}.json").read_text()) for r in (0, 1)]
    _check(tmp_path, g, ties[0] or ties[1])
