#!/usr/bin/env python3
"""Captures golden vectors from the REAL reference (imported from /root/reference, read-only) for the hot
path of SURVEY.md §8a.  Run once in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference itself never travels: only the inputs and the outputs it produced are stored, as .npz/.json
under tests/golden/.  gensim is not installed here; the reference imports it at module load
(embeddings.py:6) but only touches it inside train_embeddings (embeddings.py:68), which is never reached
when embeddings_<label>.npy files are pre-seeded (embeddings.py:31-35) -- so an empty stub module is
registered before the import, exactly as SURVEY.md §8c describes.
"""
import gzip
import json
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))

_g = types.ModuleType("gensim")
_gm = types.ModuleType("gensim.models")
_g.models = _gm
sys.modules["gensim"] = _g
sys.modules["gensim.models"] = _gm
sys.path.insert(0, "/root/reference/src")

import sage_tokenizer  # noqa: E402  (the reference)
from sage_tokenizer.model import SaGeTokenizer  # noqa: E402
from sage_tokenizer import utils as ref_utils  # noqa: E402
from sage_tokenizer.SaGeVocabBuilder import SaGeVocabBuilder  # noqa: E402
from sage_tokenizer.paths import setSageFolder  # noqa: E402

from sage_amd import synth  # noqa: E402

SEED = synth.DEFAULT_SEED


def pack_lines(lines):
    raw = [ln if isinstance(ln, bytes) else ln.encode("utf-8") for ln in lines]
    off = np.zeros(len(raw) + 1, dtype=np.int64)
    np.cumsum([len(r) for r in raw], out=off[1:])
    return np.frombuffer(b"".join(raw), dtype=np.uint8).copy(), off


def pack_vocab(vocab):
    off = np.zeros(len(vocab) + 1, dtype=np.int64)
    np.cumsum([len(v) for v in vocab], out=off[1:])
    return np.frombuffer(b"".join(vocab), dtype=np.uint8).copy(), off


def special_lines(vocab):
    """Edge cases the reference code paths distinguish."""
    longest = max(vocab, key=len)
    multi = [v for v in vocab if len(v) > 1]
    out = [b"\n", b"a\n", b"\x00\xff\n", longest, longest + b"\n", longest[:-1], longest * 3 + b"\n",
           multi[0], multi[0] + multi[1], multi[5] + b"\n", b" ".join(multi[:12]) + b".\n",
           "çğıöşü İstanbul'da.\n".encode("utf-8"), b"no newline at the end", b"x" * 70 + b"\n"]
    return out


def case_tokenize(style, max_len, vsize, n_lines, tag):
    corpus, off = synth.make_corpus(3000, seed=SEED + max_len, style=style, n_types=3000,
                                    mean_log_words=2.7, sigma_log_words=0.7)
    vocab = synth.make_vocab(corpus, off, vsize, max_len, sample_bytes=400_000)
    model = SaGeTokenizer(vocab, max_len)
    buf = corpus.tobytes()
    lines = [buf[off[i]:off[i + 1]] for i in range(n_lines)] + special_lines(vocab)
    ids, starts, widths, counts = [], [], [], []
    for ln in lines:
        toks = model.tokenize(ln)
        counts.append(len(toks))
        for a, b, c in toks:
            ids.append(a); starts.append(b); widths.append(c)
    lb, lo = pack_lines(lines)
    vb, vo = pack_vocab(vocab)
    np.savez_compressed(HERE / f"tokenize_{tag}.npz", max_len=max_len, vocab_bytes=vb, vocab_off=vo,
                        line_bytes=lb, line_off=lo, ids=np.array(ids, dtype=np.int32),
                        starts=np.array(starts, dtype=np.int32), widths=np.array(widths, dtype=np.int32),
                        counts=np.array(counts, dtype=np.int64))
    return vocab, lines, model


def case_fast_sage(vocab, lines, model, max_len, tag, n_lines=40):
    """fast_sage per line: the whole triples dict in insertion order (zero-count keys included)."""
    pick = lines[:n_lines] + lines[-14:]
    pick = [ln for ln in pick if len(ln) > 0]
    rec = {"max_len": max_len, "cases": []}
    for pad in (2, 1, 3, 5):
        sub = pick if pad == 2 else pick[:6] + pick[-6:]
        for li, ln in enumerate(sub):
            tr, ab = {}, {}
            ret = model.fast_sage(ln, tr, ab, pad=pad)
            rec["cases"].append({"pad": pad, "line": ln.hex(), "ret": ret,
                                 "keys": [list(k) for k in tr.keys()], "counts": list(tr.values()),
                                 "sizes": {str(k): v for k, v in ab.items()}})
    vb, vo = pack_vocab(vocab)
    rec["vocab"] = [v.hex() for v in vocab]
    with gzip.open(HERE / f"fast_sage_{tag}.json.gz", "wt") as f:
        json.dump(rec, f, separators=(",", ":"))
    # compute_losses on the union of the pad=2 lines
    tr, ab = {}, {}
    for ln in pick:
        model.fast_sage(ln, tr, ab)
    emb = synth.make_embeddings(len(vocab), 50, seed=SEED + 7)
    losses = {}
    ref_utils.compute_losses(losses, tr, emb)
    keys = np.array(list(tr.keys()), dtype=np.int64)
    np.savez_compressed(HERE / f"compute_losses_{tag}.npz", keys=keys,
                        counts=np.array(list(tr.values()), dtype=np.int64), emb_seed=SEED + 7, emb_dim=50,
                        emb_rows=len(vocab), emb_checksum=float(emb.sum()),
                        loss_ids=np.array(list(losses.keys()), dtype=np.int64),
                        loss_vals=np.array(list(losses.values()), dtype=np.float64))


def case_parallel():
    """run_sage_parallel on >20,002 short lines, W in {1,3,4}: pins the 10,001/10,000 flush rule, the shard
    sizes, total_triples, and the final negation (utils.py:106-113,141-257)."""
    n_lines = 20_600
    corpus, off = synth.make_corpus(n_lines, seed=SEED + 3, style="en", n_types=1500,
                                    mean_log_words=1.0, sigma_log_words=0.5)
    vocab = synth.make_vocab(corpus, off, 1200, 16, sample_bytes=200_000)
    lines = synth.corpus_lines(corpus, off)
    emb = synth.make_embeddings(len(vocab), 50, seed=SEED + 11)
    model = SaGeTokenizer(vocab, 16)
    out = {}
    for W in (1, 3, 4):
        tt, ttr, losses, sizes = ref_utils.run_sage_parallel(emb, lines, model, W)
        out[f"w{W}_total_tokens"] = tt
        out[f"w{W}_total_triples"] = ttr
        out[f"w{W}_loss_ids"] = np.array(sorted(losses.keys()), dtype=np.int64)
        out[f"w{W}_loss_vals"] = np.array([losses[k] for k in sorted(losses.keys())], dtype=np.float64)
        sk = sorted(sizes.keys())
        out[f"w{W}_size_keys"] = np.array(sk, dtype=np.int64)
        out[f"w{W}_size_vals"] = np.array([sizes[k] for k in sk], dtype=np.int64)
        print(f"parallel W={W}: tokens {tt} triples {ttr} active {len(losses)}")
    # the flush rule again, cheaply, through sage_per_chunk's chunk_size argument (utils.py:193)
    small = lines[:61]
    for cs in (7, 20, 60):
        losses, tt, ttr, sizes = ref_utils.sage_per_chunk(0, model, small, emb, chunk_size=cs)
        out[f"chunk{cs}_total_triples"] = ttr
        out[f"chunk{cs}_total_tokens"] = tt
        out[f"chunk{cs}_loss_ids"] = np.array(sorted(losses.keys()), dtype=np.int64)
        out[f"chunk{cs}_loss_vals"] = np.array([losses[k] for k in sorted(losses.keys())], dtype=np.float64)
    # shard sizes for a few (n, W): len of each yielded block
    shard_cases = [(8, 4), (10, 3), (3, 8), (20600, 3), (20600, 4), (1, 1), (10001, 1), (20002, 2)]
    out["shard_cases"] = np.array(shard_cases, dtype=np.int64)
    out["shard_sizes"] = np.array([",".join(str(len(b)) for b in ref_utils.divide_data_by_num(list(range(n)), w))
                                   for n, w in shard_cases])
    vb, vo = pack_vocab(vocab)
    np.savez_compressed(HERE / "parallel.npz", max_len=16, vocab_bytes=vb, vocab_off=vo, line_bytes=corpus,
                        line_off=off, emb_seed=SEED + 11, emb_dim=50, emb_checksum=float(emb.sum()), **out)


def case_build_vocab():
    """Full SaGeVocabBuilder.build_vocab on a small corpus with pre-seeded embeddings and an
    embeddings_schedule that SKIPS steps: pins the stale-matrix + id-compaction semantics
    (SaGeVocabBuilder.py:89-91,192) and every output file."""
    n_lines = 1500
    corpus, off = synth.make_corpus(n_lines, seed=SEED + 5, style="en", n_types=4000,
                                    mean_log_words=2.9, sigma_log_words=0.6)
    vocab = synth.make_vocab(corpus, off, 4096, 16, sample_bytes=400_000)
    schedule = [4096, 3000, 2300, 2000, 1700, 1400]
    emb_schedule = [4096, 2300, 1400]
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        setSageFolder(tmp)
        exp = "golden"
        (tmp / "results" / exp / "embeddings").mkdir(parents=True)
        emb_seeds = {}
        for label in emb_schedule:
            emb_seeds[label] = SEED + label
            np.save(tmp / "results" / exp / "embeddings" / f"embeddings_{label}.npy",
                    synth.make_embeddings(4096, 50, seed=SEED + label))
        vocab_file = tmp / "initial.vocab"
        with open(vocab_file, "w") as f:
            for v in vocab:
                f.write(v.hex() + "\n")
        corpus_file = tmp / "corpus.txt"
        corpus_file.write_bytes(corpus.tobytes())
        partial = tmp / "partial.txt"
        partial.write_bytes(corpus.tobytes())  # pre-existing partial corpus => no shuffle (utils.py:75-80)
        builder = SaGeVocabBuilder(list(schedule), list(emb_schedule), max_len=16, workers_number=2,
                                   random_seed=SEED)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            builder.build_vocab(exp, str(corpus_file), str(vocab_file), str(partial), 1000)
        finally:
            os.chdir(cwd)
        res = tmp / "results" / exp
        files = {}
        for p in sorted((res / "sage_vocabs").iterdir()):
            files[f"sage_vocabs/{p.name}"] = p.read_text(encoding="utf-8")
        for p in sorted((res / "stats").iterdir()):
            files[f"stats/{p.name}"] = p.read_text(encoding="utf-8")
        files["seed.txt"] = (res / "seed.txt").read_text()
    rec = {"schedule": schedule, "embeddings_schedule": emb_schedule, "emb_seeds": {str(k): v for k, v in emb_seeds.items()},
           "emb_rows": 4096, "emb_dim": 50, "max_len": 16, "workers": 2, "vocab": [v.hex() for v in vocab],
           "corpus_hex": corpus.tobytes().hex(), "files": files}
    with gzip.open(HERE / "build_vocab.json.gz", "wt") as f:
        json.dump(rec, f, separators=(",", ":"))
    print("build_vocab files:", sorted(files.keys()))


def case_c1():
    """BASELINE.json configs[0] exactly: 10k-line partial corpus, 16,384 -> 8,192 keys, one ablation step, D=50,
    workers=4 (window 5 only reaches gensim), through the reference's whole build_vocab with a
    pre-seeded embeddings_16384.npy.  The inputs are regenerated from seeds by the test (sha256 pinned here); the
    outputs are every file the reference wrote."""
    import hashlib
    n_lines, V = 10_000, 16_384
    corpus, off = synth.make_corpus(n_lines, seed=SEED + 101, style="en", n_types=50_000)
    vocab = synth.make_vocab(corpus, off, V, 16, sample_bytes=6_000_000)
    assert len(vocab) == V
    emb_seed = SEED + 102
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        setSageFolder(tmp)
        exp = "c1"
        (tmp / "results" / exp / "embeddings").mkdir(parents=True)
        np.save(tmp / "results" / exp / "embeddings" / f"embeddings_{V}.npy", synth.make_embeddings(V, 50, seed=emb_seed))
        vocab_file = tmp / "initial.vocab"
        vocab_file.write_text("".join(v.hex() + "\n" for v in vocab))
        (tmp / "corpus.txt").write_bytes(corpus.tobytes())
        (tmp / "partial.txt").write_bytes(corpus.tobytes())   # pre-existing partial corpus => no shuffle
        builder = SaGeVocabBuilder([V, 8192], [V], max_len=16, workers_number=4, random_seed=SEED,
                                   word2vec_d=50, word2vec_window_size=5)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            builder.build_vocab(exp, str(tmp / "corpus.txt"), str(vocab_file), str(tmp / "partial.txt"), 10)
        finally:
            os.chdir(cwd)
        res = tmp / "results" / exp
        files = {}
        for sub in ("sage_vocabs", "stats"):
            for p in sorted((res / sub).iterdir()):
                if p.name.startswith(("worst_500", "best_500")):
                    continue                      # slices of sorted_losses_before_, pinned by the small golden already
                files[f"{sub}/{p.name}"] = p.read_text(encoding="utf-8")
    rec = {"n_lines": n_lines, "vocab_size": V, "schedule": [V, 8192], "embeddings_schedule": [V], "workers": 4,
           "max_len": 16, "dim": 50, "corpus_seed": SEED + 101, "n_types": 50_000, "emb_seed": emb_seed,
           "corpus_sha256": hashlib.sha256(corpus.tobytes()).hexdigest(),
           "vocab_sha256": hashlib.sha256(b"\n".join(v.hex().encode() for v in vocab)).hexdigest(), "files": files}
    with gzip.open(HERE / "c1_build_vocab.json.gz", "wt") as f:
        json.dump(rec, f, separators=(",", ":"))
    print("c1 files:", {k: len(v) for k, v in files.items()})
    print(files["stats/stats_8192.json"])


def case_gensim_file():
    """The data/gensim_<V>.txt the reference writes BEFORE it touches gensim (embeddings.py:53-66): with the empty
    gensim stub, train_embeddings tokenizes every line, writes the file and only then fails on
    gensim.models.Word2Vec (AttributeError), which is caught here."""
    from sage_tokenizer import embeddings as ref_emb
    from sage_tokenizer.Word2VecParams import Word2VecParams
    rec = {}
    for tag in ("en16", "tr17"):
        g = np.load(HERE / f"tokenize_{tag}.npz")
        vb, vo, lb, lo = g["vocab_bytes"].tobytes(), g["vocab_off"], g["line_bytes"].tobytes(), g["line_off"]
        vocab = [vb[vo[i]:vo[i + 1]] for i in range(len(vo) - 1)]
        lines = []
        for i in range(len(lo) - 1):
            try:
                lines.append(lb[lo[i]:lo[i + 1]].decode("utf-8"))    # partial_corpus is List[str]
            except UnicodeDecodeError:
                pass                                                  # (the golden holds a raw b"\x00\xff" line)
        model = SaGeTokenizer(vocab, int(g["max_len"]))
        with tempfile.TemporaryDirectory() as tmp:
            setSageFolder(Path(tmp))
            try:
                ref_emb.train_embeddings(model, lines, 1, Word2VecParams(50, 15, 0.025, 5, 1, 1))
                raise SystemExit("expected the stubbed gensim to fail after the file was written")
            except AttributeError:
                pass
            text = (Path(tmp) / "data" / f"gensim_{model.vocab_size()}.txt").read_text(encoding="utf-8")
        rec[tag] = {"lines": lines, "vocab_size": model.vocab_size(), "text": text}
        print(f"gensim file {tag}: {len(lines)} lines, {len(text)} chars")
    with gzip.open(HERE / "gensim_file.json.gz", "wt") as f:
        json.dump(rec, f, separators=(",", ":"), ensure_ascii=False)


def case_load_corpus():
    """load_corpus (utils.py:71-103): shuffle order under a seed, the [:N*1000] slice, the saved partial-corpus file,
    the reload path, a last line without a newline, multi-byte text."""
    import hashlib
    import random
    corpus, off = synth.make_corpus(2600, seed=SEED + 201, style="tr", n_types=800, mean_log_words=1.2, sigma_log_words=0.5)
    text = corpus.tobytes()[:-1]          # the last line loses its newline
    rec = {"corpus_seed": SEED + 201, "cases": []}
    for seed, k in ((7, 2), (692653, 1), (11, 5)):
        with tempfile.TemporaryDirectory() as tmp:
            tmp = Path(tmp)
            setSageFolder(tmp)
            (tmp / "wiki.txt").write_bytes(text)
            random.seed(seed)
            got = ref_utils.load_corpus(tmp / "wiki.txt", None, k)
            saved = sorted((tmp / "data").iterdir())
            assert len(saved) == 1
            file_bytes = saved[0].read_bytes()
            again = ref_utils.load_corpus(tmp / "wiki.txt", saved[0], k)
            assert file_bytes == "".join(got).encode("utf-8")     # writelines(partial): the file is the lines back to back
            rec["cases"].append({"seed": seed, "k": k, "n": len(got), "file_name": saved[0].name,
                                 "lines": got, "file_sha256": hashlib.sha256(file_bytes).hexdigest(),
                                 "reload_n": len(again), "reload_last": again[-1], "reload_equal": again == got})
            print(f"load_corpus seed={seed} k={k}: {len(got)} lines, file {saved[0].name} {len(file_bytes)} B, reload {len(again)} lines")
    rec["corpus_sha256"] = hashlib.sha256(text).hexdigest()
    rec["corpus_args"] = {"n_lines": 2600, "style": "tr", "n_types": 800, "mean_log_words": 1.2, "sigma_log_words": 0.5}
    with gzip.open(HERE / "load_corpus.json.gz", "wt") as f:
        json.dump(rec, f, separators=(",", ":"), ensure_ascii=False)


def main():
    single = {"build_vocab": case_build_vocab, "c1": case_c1, "gensim_file": case_gensim_file, "load_corpus": case_load_corpus}
    if len(sys.argv) > 1:          # regenerate one fixture without touching the others
        single[sys.argv[1]]()
        return
    vocab, lines, model = case_tokenize("en", 16, 3000, 110, "en16")
    case_fast_sage(vocab, lines, model, 16, "en16")
    vocab, lines, model = case_tokenize("tr", 17, 2500, 90, "tr17")
    case_fast_sage(vocab, lines, model, 17, "tr17", n_lines=25)
    case_parallel()
    case_build_vocab()
    case_gensim_file()
    case_load_corpus()
    case_c1()
    for p in sorted(HERE.iterdir()):
        print(f"{p.name:32s} {p.stat().st_size:>9d} B")


if __name__ == "__main__":
    main()
