#!/usr/bin/env python3
"""Golden of BASELINE.json configs[2] (C3): the reference's WHOLE README schedule, run by the real reference.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_readme_schedule.py            # ~25 min, 4 worker processes

`SaGeVocabBuilder(full_vocab_schedule=[262144, ..., 16384], embeddings_schedule=[262144, 131072, 65536, 49152, 40960,
32768], max_len=17, workers_number=4, random_seed=692653)` (README.md:11-13, run.sh:1-10) imported from /root/reference
and run end to end: 11 consecutive prunes from a 262,144-key vocabulary, 6 retrain labels (so rounds that score with
stale rows of an older matrix after ids were re-compacted alternate with rounds on a fresh one,
SaGeVocabBuilder.py:89-91,192), max_len 17 on a Turkish-style multi-byte corpus.

Inputs (made here, never taken from the reference):
  corpus      240,000 short lines of `synth.make_corpus(style="tr", mean_log_words=2.2)` (28 MB, ~3.5 M tokens, 24 count
              segments at 4 workers); regenerated from its seed by the test, sha256 pinned here.
  vocabulary  262,144 keys of which > 99 % are ACTIVE on that corpus, so that the first target (229,376) already forces
              a prune: frequency-ranked substrings (`synth.make_vocab`, 600,000 candidates), tokenized once, the 261,888
              most used multi-byte keys kept.  Stored in the fixture (sorted + front-coded); ids = 256 single bytes, then
              a seeded permutation of the sorted multi-byte keys.
  word2vec    gensim cannot run here (and is not reproducible): tests/fake_gensim.py stands in for
              `gensim.models.Word2Vec` on BOTH sides, so the reference's own train_embeddings (embeddings.py:46-89) runs:
              it writes data/gensim_<V>.txt, fills the matrix in id order and draws the uniform fallback rows from
              numpy's global RNG.  Each retrain's matrix has exactly the rows of the vocabulary at that label.
Outputs stored (the reference's files are 60 MB, the fixture keeps what pins them):
  every .vocab file: sha256 + line count; every stats_<t>.json in full; of sorted_losses_before_<t>.txt the line count,
  the sha256 of the token column (the ranking), the smallest relative gap between neighbours, every 64th line
  (rank, token, value) and the sums of the values; worst_500_ / best_500_ in full; sha256 + shape of every
  embeddings_<label>.npy and sha256 of every data/gensim_<V>.txt.
"""
import gzip
import hashlib
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))

from tests import fake_gensim  # noqa: E402

fake_gensim.install()
sys.path.insert(0, "/root/reference/src")

from sage_tokenizer.SaGeVocabBuilder import SaGeVocabBuilder  # noqa: E402  (the reference)
from sage_tokenizer.paths import setSageFolder  # noqa: E402

from sage_amd import synth  # noqa: E402

SEED = synth.DEFAULT_SEED
N_LINES = 240_000        # many SHORT lines: the reference gathers one numpy row object per distinct triple of a 10,000-line
MEAN_LOG_WORDS = 2.2     # flush (utils.py:132-134): 7,500 lines of 950 bytes took 35 GB per worker here, 10,000 of 118 bytes take ~6
MAX_LEN = 17
V0 = 262_144
SCHEDULE = [262144, 229376, 196608, 163840, 131072, 98304, 65536, 57344, 49152, 40960, 32768, 16384]   # README.md:11
EMB_SCHEDULE = [262144, 131072, 65536, 49152, 40960, 32768]                                              # README.md:12
WORKERS = 4
SAMPLE_EVERY = 64
N_CAND = 600_000
N_TYPES = 1_000_000
OUT_TAG = "readme_schedule"
if os.environ.get("C3_SMALL"):      # dry run of this script at toy size (writes readme_schedule_small.*; not committed)
    N_LINES, V0, N_CAND, N_TYPES, OUT_TAG = 6000, 4096, 9000, 20_000, "readme_schedule_small"
    SCHEDULE = [4096, 3584, 3072, 2560, 2048, 1536, 1024, 896, 768, 640, 512, 384]
    EMB_SCHEDULE = [4096, 2048, 1024, 768, 640, 512]


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def make_inputs():
    corpus, off = synth.make_corpus(N_LINES, seed=SEED + 301, style="tr", n_types=N_TYPES, mean_log_words=MEAN_LOG_WORDS)
    t0 = time.time()
    cand = synth.make_vocab(corpus, off, N_CAND, MAX_LEN, sample_bytes=len(corpus) + 1)
    print(f"{len(cand)} candidate keys in {time.time() - t0:.0f} s", flush=True)
    buf = corpus.tobytes()
    table = set(cand)
    used = {}
    for i in range(N_LINES):
        s = buf[off[i]:off[i + 1]]
        at, n = 0, len(s)
        while at < n:
            for j in range(min(MAX_LEN, n - at), 0, -1):
                p = s[at:at + j]
                if p in table:
                    used[p] = used.get(p, 0) + 1
                    at += j
                    break
    multi = sorted((k for k in used if len(k) > 1), key=lambda k: (-used[k], k))[:V0 - 256]
    assert len(multi) == V0 - 256, len(multi)
    multi.sort()
    perm = np.random.default_rng(SEED + 302).permutation(len(multi))
    vocab = [bytes([b]) for b in range(256)] + [multi[i] for i in perm]
    return corpus, off, vocab, multi


def front_code(sorted_keys):
    pre, suf, tail = [], [], bytearray()
    last = b""
    for k in sorted_keys:
        p = 0
        while p < len(last) and p < len(k) and last[p] == k[p]:
            p += 1
        pre.append(p); suf.append(len(k) - p); tail += k[p:]
        last = k
    return np.array(pre, dtype=np.uint8), np.array(suf, dtype=np.uint8), np.frombuffer(bytes(tail), dtype=np.uint8)


def main():
    t_all = time.time()
    corpus, off, vocab, multi_sorted = make_inputs()
    pre, suf, tail = front_code(multi_sorted)
    np.savez_compressed(HERE / f"{OUT_TAG}_vocab.npz", prefix_len=pre, suffix_len=suf, suffix_bytes=tail,
                        perm_seed=SEED + 302)
    print(f"inputs ready after {time.time() - t_all:.0f} s", flush=True)
    rec = {"n_lines": N_LINES, "corpus_seed": SEED + 301, "style": "tr", "n_types": N_TYPES, "mean_log_words": MEAN_LOG_WORDS, "max_len": MAX_LEN,
           "schedule": SCHEDULE, "embeddings_schedule": EMB_SCHEDULE, "workers": WORKERS, "random_seed": SEED, "dim": 50,
           "corpus_sha256": sha(corpus.tobytes()),
           "vocab_sha256": sha(b"\n".join(v.hex().encode() for v in vocab)), "sample_every": SAMPLE_EVERY,
           "vocab_files": {}, "stats": {}, "sorted_losses": {}, "head_tail": {}, "embeddings": {}, "gensim_files": {}}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        tmp = Path(tmp)
        setSageFolder(tmp)
        exp = "c3"
        (tmp / "results" / exp).mkdir(parents=True)          # utils.py:272-275 makes the sub-folders without parents
        vocab_file = tmp / "initial.vocab"
        vocab_file.write_text("".join(v.hex() + "\n" for v in vocab))
        (tmp / "corpus.txt").write_bytes(corpus.tobytes())
        (tmp / "partial.txt").write_bytes(corpus.tobytes())   # a pre-existing partial corpus: no shuffle (utils.py:75-80)
        builder = SaGeVocabBuilder(list(SCHEDULE), list(EMB_SCHEDULE), max_len=MAX_LEN, workers_number=WORKERS,
                                   random_seed=SEED)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            t0 = time.time()
            builder.build_vocab(exp, str(tmp / "corpus.txt"), str(vocab_file), str(tmp / "partial.txt"), N_LINES // 1000)
            print(f"reference build_vocab: {time.time() - t0:.0f} s", flush=True)
        finally:
            os.chdir(cwd)
        res = tmp / "results" / exp
        for p in sorted((res / "sage_vocabs").iterdir()):
            text = p.read_text(encoding="utf-8")
            if p.name.endswith(".vocab"):
                rec["vocab_files"][p.name] = {"sha256": sha(text.encode("utf-8")), "lines": text.count("\n")}
            elif p.name.startswith(("worst_500_", "best_500_")):
                rec["head_tail"][p.name] = text
            elif p.name.startswith("sorted_losses_before_"):
                lines = text.splitlines()
                toks, vals = zip(*(ln.rsplit("\t", 1) for ln in lines))
                v = np.array(vals, dtype=np.float64)
                real = v[v < 1e5]                       # the 256 single bytes carry 1e6 (SaGeVocabBuilder.py:109)
                gaps = np.diff(real) / np.maximum(np.abs(real[1:]), np.abs(real[:-1]))
                rec["sorted_losses"][p.name] = {
                    "lines": len(lines), "tokens_sha256": sha("\n".join(toks).encode("utf-8")),
                    "min_rel_gap": float(gaps[gaps > 0].min()) if (gaps > 0).any() else 0.0,
                    "exact_ties": int((gaps == 0).sum()),
                    "sum": float(real.sum()), "abs_sum": float(np.abs(real).sum()),
                    "sample": [[i, toks[i], vals[i]] for i in range(0, len(lines), SAMPLE_EVERY)]}
        for p in sorted((res / "stats").iterdir()):
            rec["stats"][p.name] = p.read_text(encoding="utf-8")
        rec["seed_txt"] = (res / "seed.txt").read_text()
        for p in sorted((res / "embeddings").iterdir()):
            a = np.load(p)
            rec["embeddings"][p.name] = {"shape": list(a.shape), "sha256": sha(a.tobytes())}
        for p in sorted((tmp / "data").iterdir()):
            if p.name.startswith("gensim_"):
                rec["gensim_files"][p.name] = {"sha256": sha(p.read_bytes()), "bytes": p.stat().st_size}
    with gzip.open(HERE / f"{OUT_TAG}.json.gz", "wt") as f:
        json.dump(rec, f, separators=(",", ":"), ensure_ascii=False)
    for name in (f"{OUT_TAG}.json.gz", f"{OUT_TAG}_vocab.npz"):
        print(f"{name}: {(HERE / name).stat().st_size} B")
    print("vocab files:", {k: v["lines"] for k, v in rec["vocab_files"].items() if k.startswith("sage_vocab")})
    print("embeddings:", {k: v["shape"] for k, v in rec["embeddings"].items()})
    print("min gaps:", {k: (v["min_rel_gap"], v["exact_ties"]) for k, v in rec["sorted_losses"].items()})
    print(f"total {time.time() - t_all:.0f} s")


if __name__ == "__main__":
    main()
