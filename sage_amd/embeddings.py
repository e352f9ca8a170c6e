"""Embedding checkpoints and word2vec retraining (reference: embeddings.py:28-89).

A run whose `embeddings_<label>.npy` files are present never retrains -- that is also the reference's resume
mechanism (embeddings.py:31-35).  When a retrain is needed there are two back ends:
  "gensim"  the reference's host call (embeddings.py:68-75); the corpus file handed to gensim is written from GPU
            tokenization (one batched call, model.py:103-107 semantics) instead of a per-line Python loop;
  "hip"     `sage_train_embeddings` (include/sage_hip.h): the same skip-gram / negative-sampling algorithm on the GPU
            over the resident corpus, no corpus file and no gensim needed.
SAGE_W2V_BACKEND=gensim|hip chooses; the default is gensim when it can be imported (the reference's behaviour), else
hip.  Neither back end is reproducible bit for bit (lock-free training): parity of the vectors is unpinned."""
from __future__ import annotations

import logging
import os
import time
from pathlib import Path
from typing import List

import numpy as np

from .folders import getDataFolder
from .w2v_params import Word2VecParams


def get_embeddings(vocab_size: int, embeddings_folder, partial_corpus, sage_model, workers_number: int,
                   word2vec_params: Word2VecParams) -> np.ndarray:
    """embeddings.py:28-43: load `embeddings_<label>.npy` if it is there, else train and save it.  In a multi-GPU run
    rank 0 alone loads or trains (and writes the checkpoint); the matrix is then broadcast, so every rank scores its
    shard with the same rows and the saved file reproduces the run."""
    from . import dist as sdist
    rank, _ = sdist.world_info()
    emb = None
    if rank == 0:
        path = Path(embeddings_folder) / f"embeddings_{vocab_size}.npy"
        if path.exists():
            logging.info(f"embeddings for label {vocab_size}: loading {path.as_posix()}")
            emb = np.load(path.as_posix())
        else:
            logging.info(f"embeddings for label {vocab_size}: no checkpoint, training word2vec")
            started = time.time()
            emb = train_embeddings(sage_model, partial_corpus, workers_number, word2vec_params)
            logging.info(f"word2vec took {time.time() - started:.1f} s; saving {path.as_posix()}")
            np.save(path.as_posix(), emb, allow_pickle=True)
    return sdist.broadcast_array(emb, src=0)


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


GENSIM_EPOCHS = 5          # gensim.models.Word2Vec defaults the reference relies on (embeddings.py:68-75 passes none)
GENSIM_SAMPLE = 1e-3
GENSIM_MIN_ALPHA = 1e-4


def _backend() -> str:
    want = os.environ.get("SAGE_W2V_BACKEND", "").strip().lower()
    if want in ("gensim", "hip"):
        return want
    try:
        import gensim.models  # noqa: F401
        return "gensim"
    except ImportError:
        return "hip"


def fill_matrix(sage_model, D: int, vector_of) -> np.ndarray:
    """embeddings.py:77-89: one row per vocabulary id, in id order; `vector_of(idx, token)` returns the trained vector
    or None, and a missing vector gets the uniform draw from numpy's global RNG, in the same order as the reference."""
    emb = np.zeros(shape=(sage_model.vocab_size(), D))
    half = 0.5 / D
    for idx, token in sage_model.inv_str_vocab.items():
        v = vector_of(idx, token)
        if v is not None:
            emb[idx] = v
        else:
            emb[idx] = np.random.uniform(low=-half, high=half, size=(1, D))
    return emb


def train_embeddings_hip(sage_model, partial_corpus, word2vec_params: Word2VecParams, seed: int = 1) -> np.ndarray:
    """word2vec on the GPU over the resident corpus (sage_train_embeddings)."""
    from .parallel import get_engine, sync_corpus, sync_vocab
    if not word2vec_params.sg:
        raise ValueError("the HIP word2vec back end implements skip-gram (sg=1), the reference's setting")
    engine = get_engine()
    sync_vocab(engine, sage_model)
    sync_corpus(engine, partial_corpus)     # the whole corpus (a multi-GPU rank re-uploads its shard at the next step)
    vsize = sage_model.id_space()
    vec, counts = engine.train_embeddings(vsize, word2vec_params.D, window=word2vec_params.window_size,
                                          negative=word2vec_params.N, alpha=word2vec_params.ALPHA,
                                          min_alpha=GENSIM_MIN_ALPHA, epochs=GENSIM_EPOCHS, sample=GENSIM_SAMPLE,
                                          min_count=word2vec_params.min_count, seed=seed)
    trained = counts >= word2vec_params.min_count
    return fill_matrix(sage_model, word2vec_params.D, lambda idx, token: vec[idx] if trained[idx] else None)


def train_embeddings(sage_model, partial_corpus: List[str], workers: int, word2vec_params: Word2VecParams) -> np.ndarray:
    if _backend() == "hip":
        logging.info("word2vec back end: HIP (sage_train_embeddings)")
        return train_embeddings_hip(sage_model, partial_corpus, word2vec_params)
    return train_embeddings_gensim(sage_model, partial_corpus, workers, word2vec_params)


def train_embeddings_gensim(sage_model, partial_corpus: List[str], workers: int, word2vec_params: Word2VecParams) -> np.ndarray:
    try:
        import gensim.models  # pinned 4.3.2 by the reference (pyproject.toml:8-11); host-side only
    except ImportError as exc:
        raise ImportError("gensim is needed to retrain embeddings; install gensim==4.3.2 or pre-seed "
                          "results/<exp>/embeddings/embeddings_<label>.npy for every label that is used") from exc
    corpus_file = getDataFolder() / f"gensim_{sage_model.vocab_size()}.txt"
    if corpus_file.exists():
        logging.info(f"reusing the gensim corpus file {corpus_file.as_posix()}")
    else:
        started = time.time()
        logging.info(f"tokenizing {len(partial_corpus)} lines on the GPU for gensim")
        write_gensim_corpus(sage_model, partial_corpus, corpus_file)
        logging.info(f"{corpus_file.as_posix()} written in {(time.time() - started):.2f} s")
    w2v = gensim.models.Word2Vec(corpus_file=corpus_file.as_posix(), vector_size=word2vec_params.D,
                                 negative=word2vec_params.N, alpha=word2vec_params.ALPHA,
                                 window=word2vec_params.window_size, min_count=word2vec_params.min_count,
                                 sg=word2vec_params.sg, workers=workers)
    known = w2v.wv.key_to_index
    return fill_matrix(sage_model, word2vec_params.D, lambda idx, token: w2v.wv[token] if token in known else None)
