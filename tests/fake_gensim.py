"""A deterministic stand-in for `gensim.models.Word2Vec` -- TEST INFRASTRUCTURE.

gensim (pinned 4.3.2 by the reference, pyproject.toml:8-11) is not installed in the build container or on the GPU box,
and its training is lock-free, so its vectors cannot be pinned anyway (word2vec parity is unpinned).  What
CAN be pinned is everything around the call at embeddings.py:68-75: the `data/gensim_<V>.txt` file handed to it
(embeddings.py:53-66), the `[V, D]` float64 matrix filled from `wv[token]` in id order with the `np.random.uniform`
fallback for tokens the corpus never produces (embeddings.py:77-89, which consumes numpy's GLOBAL random state), the
`.npy` checkpoint, and how the builder loop uses the matrices between retrains (stale rows after id re-compaction,
SaGeVocabBuilder.py:89-91,192).

`install()` registers this module as `gensim` / `gensim.models`.  `Word2Vec(corpus_file=...)` reads the corpus file it is
given (so a file that differs in one token changes `key_to_index`), keeps every token with at least `min_count`
occurrences, and gives token `t` the float32 vector drawn from `RandomState(crc32(t) ^ crc32(file name))` -- a pure
function of the token and of the vocabulary size in the file name, independent of numpy's global state.  Both
tests/golden/make_golden.py (around the imported reference) and the GPU tests (around sage_amd) use it, so the two
sides see identical "trained" vectors.
"""
from __future__ import annotations

import os
import sys
import types
import zlib
from collections import Counter

import numpy as np


class _KeyedVectors:
    def __init__(self, counts, salt: int, dim: int, min_count: int):
        self.key_to_index = {tok: i for i, (tok, n) in enumerate(counts.items()) if n >= min_count}
        self._salt, self._dim = salt, dim

    def __getitem__(self, token: str) -> np.ndarray:
        if token not in self.key_to_index:
            raise KeyError(token)
        seed = (zlib.crc32(token.encode("utf-8")) ^ self._salt) & 0xFFFFFFFF
        return np.random.RandomState(seed).normal(0.0, 0.25, self._dim).astype(np.float32)

    def __contains__(self, token) -> bool:
        return token in self.key_to_index


class Word2Vec:
    def __init__(self, corpus_file=None, vector_size=100, negative=5, alpha=0.025, window=5, min_count=5, sg=0, workers=3,
                 **_ignored):
        counts: Counter = Counter()
        with open(corpus_file, encoding="utf-8") as f:
            for line in f:
                counts.update(tok for tok in line.rstrip("\n").split(" ") if tok)
        salt = zlib.crc32(os.path.basename(str(corpus_file)).encode("utf-8"))
        self.params = dict(vector_size=vector_size, negative=negative, alpha=alpha, window=window, min_count=min_count, sg=sg,
                           workers=workers)
        self.wv = _KeyedVectors(counts, salt, int(vector_size), int(min_count))


def install() -> None:
    """Registers the stand-in as `gensim` and `gensim.models` (before the reference or sage_amd imports gensim)."""
    g = types.ModuleType("gensim")
    gm = types.ModuleType("gensim.models")
    gm.Word2Vec = Word2Vec
    g.models = gm
    g.__fake__ = True
    sys.modules["gensim"] = g
    sys.modules["gensim.models"] = gm


def uninstall() -> None:
    if getattr(sys.modules.get("gensim"), "__fake__", False):
        sys.modules.pop("gensim.models", None)
        sys.modules.pop("gensim", None)
