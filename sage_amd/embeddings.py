"""Embedding checkpoints and (host-side) word2vec retraining (reference: embeddings.py:28-89).

gensim stays on the host (north star) and is imported only when a retrain is really needed: a run whose
`embeddings_<label>.npy` files are present never touches it -- that is also the reference's resume mechanism
(embeddings.py:31-35).  The corpus file handed to gensim is written from GPU tokenization (one batched call,
model.py:103-107 semantics) instead of a per-line Python loop (embeddings.py:61-65)."""
from __future__ import annotations

import logging
import time
from pathlib import Path
from typing import List

import numpy as np

from .folders import getDataFolder
from .w2v_params import Word2VecParams


def get_embeddings(vocab_size: int, embeddings_folder, partial_corpus, sage_model, workers_number: int,
                   word2vec_params: Word2VecParams) -> np.ndarray:
    logging.info(f"training Embeddings at vocab size {vocab_size}")
    path = Path(embeddings_folder) / f"embeddings_{vocab_size}.npy"
    if path.exists():
        logging.info(f"Found trained embeddings. Loading it from {path.as_posix()}")
        return np.load(path.as_posix())
    logging.info("Start training embeddings with Word2Vec...")
    started = time.time()
    emb = train_embeddings(sage_model, partial_corpus, workers_number, word2vec_params)
    logging.info(f"Embeddings time: {time.time() - started}")
    logging.info(f"Save embeddings to {path.as_posix()}")
    np.save(path.as_posix(), emb, allow_pickle=True)
    return emb


def write_gensim_corpus(sage_model, partial_corpus, out_path: Path, lines_per_call: int = 200_000) -> None:
    """data/gensim_<V>.txt: one line per corpus line, tokens in HF-encoded form joined by single spaces
    (embeddings.py:61-65).  Tokenization is one GPU call per `lines_per_call` lines."""
    inv = sage_model.inv_str_vocab
    with open(out_path, "w", encoding="utf-8") as out:
        for lo in range(0, len(partial_corpus), lines_per_call):
            chunk = partial_corpus[lo:lo + lines_per_call]
            ids, _, _, counts = sage_model.tokenize_lines(chunk)
            ids = ids.tolist()
            at = 0
            for n in counts.tolist():
                out.write(" ".join(inv[i] for i in ids[at:at + n]) + "\n")
                at += n


def train_embeddings(sage_model, partial_corpus: List[str], workers: int, word2vec_params: Word2VecParams) -> np.ndarray:
    try:
        import gensim.models  # pinned 4.3.2 by the reference (pyproject.toml:8-11); host-side only
    except ImportError as exc:
        raise ImportError("gensim is needed to retrain embeddings; install gensim==4.3.2 or pre-seed "
                          "results/<exp>/embeddings/embeddings_<label>.npy for every label that is used") from exc
    corpus_file = getDataFolder() / f"gensim_{sage_model.vocab_size()}.txt"
    if corpus_file.exists():
        logging.info(f"Gensim format data file already exists: {corpus_file.as_posix()}")
    else:
        started = time.time()
        logging.info(f"starting tokenization of {len(partial_corpus)} lines for gensim")
        write_gensim_corpus(sage_model, partial_corpus, corpus_file)
        logging.info(f"Gensim format data written: {corpus_file.as_posix()}, time: {(time.time() - started):.2f}")
    w2v = gensim.models.Word2Vec(corpus_file=corpus_file.as_posix(), vector_size=word2vec_params.D,
                                 negative=word2vec_params.N, alpha=word2vec_params.ALPHA,
                                 window=word2vec_params.window_size, min_count=word2vec_params.min_count,
                                 sg=word2vec_params.sg, workers=workers)
    emb = np.zeros(shape=(sage_model.vocab_size(), word2vec_params.D))
    known = w2v.wv.key_to_index
    half = 0.5 / word2vec_params.D
    for idx, token in sage_model.inv_str_vocab.items():
        if token in known:
            emb[idx] = w2v.wv[token]
        else:  # below min_count: same random fallback, same RNG stream order as embeddings.py:86
            emb[idx] = np.random.uniform(low=-half, high=half, size=(1, word2vec_params.D))
    return emb
