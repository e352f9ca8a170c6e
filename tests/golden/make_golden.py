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


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "build_vocab":
        case_build_vocab()
        return
    vocab, lines, model = case_tokenize("en", 16, 3000, 110, "en16")
    case_fast_sage(vocab, lines, model, 16, "en16")
    vocab, lines, model = case_tokenize("tr", 17, 2500, 90, "tr17")
    case_fast_sage(vocab, lines, model, 17, "tr17", n_lines=25)
    case_parallel()
    case_build_vocab()
    for p in sorted(HERE.iterdir()):
        print(f"{p.name:32s} {p.stat().st_size:>9d} B")


if __name__ == "__main__":
    main()
