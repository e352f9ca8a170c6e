"""Host-side files and bookkeeping of a SaGe run, byte-compatible with the reference's formats
(utils.py:21-103, 260-304): hex .vocab files, partial-corpus cache, sorted-loss dumps, stats JSON, seed file,
logger and output folders.  None of this is on the GPU path."""
from __future__ import annotations

import json
import logging
import random
import time
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

from .folders import getDataFolder, getLogsFolder, getResultsFolder


def write_vocab(vocab: Dict[bytes, int], filename) -> None:
    """One lowercase-hex key per line, in increasing id order (utils.py:21-33)."""
    ordered = sorted(vocab.items(), key=lambda kv: kv[1])
    with open(filename, "w", encoding="utf-8") as out:
        out.write("".join(key.hex() + "\n" for key, _ in ordered))


def load_vocab(vocab_filepath) -> List[bytes]:
    """utils.py:55-68; bytes.fromhex skips the trailing newline (and any other ASCII whitespace)."""
    path = Path(vocab_filepath)
    if not path.exists():
        raise FileNotFoundError(f"Missing vocab file: {path.as_posix()}")
    with open(path, "r") as src:
        return [bytes.fromhex(line) for line in src.readlines()]


def write_sorted_losses_into_file(sl: Iterable[Tuple[float, int]], filename, sage_model) -> None:
    """`encoded_token<TAB>str(loss)` per line (utils.py:49-52)."""
    with open(filename, "w", encoding="utf-8") as out:
        for loss, tid in sl:
            out.write(f"{sage_model.id_to_encoded(tid)}\t{loss}\n")


def save_sorted_losses(sage_model, sorted_losses, target_vocab_size: int, vocab_folder) -> None:
    """utils.py:36-46: the full ranking plus its first and last 500 entries."""
    folder = Path(vocab_folder)
    full = folder / f"sorted_losses_before_{target_vocab_size}.txt"
    logging.info(f"Saving sorted losses to {full.as_posix()}")
    write_sorted_losses_into_file(sorted_losses, full, sage_model)
    write_sorted_losses_into_file(sorted_losses[:500], folder / f"worst_500_{target_vocab_size}.txt", sage_model)
    write_sorted_losses_into_file(sorted_losses[-500:], folder / f"best_500_{target_vocab_size}.txt", sage_model)


def save_stats(stats: dict, stats_folder, target_vocab_size: int) -> None:
    """utils.py:297-304: indent=2 JSON followed by a newline."""
    path = Path(stats_folder) / f"stats_{target_vocab_size}.json"
    logging.info(f"Saving stats to {path.as_posix()}")
    with open(path, "w") as out:
        json.dump(stats, out, indent=2)
        out.write("\n")


def load_corpus(corpus_filepath, partial_corpus_filepath: Optional[Path], partial_corpus_line_number: int) -> List[str]:
    """utils.py:71-103: reuse the partial corpus file if it exists, else shuffle the full corpus with the
    already-seeded `random`, keep the first N*1000 lines and save them.  Lines keep their trailing newline."""
    corpus_filepath = Path(corpus_filepath)
    if isinstance(partial_corpus_filepath, str):
        partial_corpus_filepath = Path(partial_corpus_filepath)
    started = time.time()
    if partial_corpus_filepath and partial_corpus_filepath.exists():
        logging.info(f"Found pre-existing partial corpus. Loading from {partial_corpus_filepath.as_posix()}...")
        with open(partial_corpus_filepath, "r") as src:
            partial = src.readlines()
        logging.info(f"Size of Corpus: {len(partial)}, time: {(time.time() - started):.2f}")
        return partial
    with open(corpus_filepath, "r") as src:
        everything = src.readlines()
    logging.info(f"Loading from Original Corpus. Number of lines: {len(everything)}")
    random.shuffle(everything)
    logging.info(f"Original Corpus read and shuffled. Time: {(time.time() - started):.2f}")
    started = time.time()
    partial = everything[:partial_corpus_line_number * 1000]
    if partial_corpus_filepath is None:
        partial_corpus_filepath = getDataFolder() / f"{corpus_filepath.stem}_{len(partial)}.txt"
    with open(partial_corpus_filepath, "w+") as out:
        out.writelines(partial)
    logging.info(f"Partial corpus saved at {partial_corpus_filepath.as_posix()}. Number of lines: {len(partial)}, "
                 f"time: {(time.time() - started):.2f}")
    return partial


def init_logger(experiment_name: str) -> None:
    """utils.py:260-268."""
    log_file = getLogsFolder() / f"{experiment_name}_{time.strftime('%Y%m%d_%H%M%S')}.log"
    logging.basicConfig(filename=log_file.as_posix(), format="%(asctime)s - %(message)s",
                        datefmt="%m/%d/%Y %H:%M:%S", level=logging.INFO)
    print(f"Logs will be stored in {log_file.as_posix()}")


def get_output_folder(experiment_name: str) -> Tuple[Path, Path, Path]:
    """results/<exp>/{embeddings,stats,sage_vocabs} (utils.py:271-283).  Unlike the reference this also creates
    results/<exp> itself, so a fresh run does not need the directory to pre-exist."""
    root = getResultsFolder() / experiment_name
    made = []
    for name in ("embeddings", "stats", "sage_vocabs"):
        d = root / name
        d.mkdir(parents=True, exist_ok=True)
        made.append(d)
    return made[0], made[1], made[2]


def set_random_seed(experiment_name: str, random_seed: int) -> None:
    """utils.py:286-294: seed.txt, then python and numpy generators."""
    root = getResultsFolder() / experiment_name
    root.mkdir(parents=True, exist_ok=True)
    with open(root / "seed.txt", "w+") as out:
        out.write(str(random_seed))
    random.seed(random_seed)
    np.random.seed(random_seed)
