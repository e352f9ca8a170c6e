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
    logging.info(f"ranking of {len(sorted_losses)} losses -> {full.as_posix()} (+ worst_500 / best_500)")
    write_sorted_losses_into_file(sorted_losses, full, sage_model)
    write_sorted_losses_into_file(sorted_losses[:500], folder / f"worst_500_{target_vocab_size}.txt", sage_model)
    write_sorted_losses_into_file(sorted_losses[-500:], folder / f"best_500_{target_vocab_size}.txt", sage_model)


def save_stats(stats: dict, stats_folder, target_vocab_size: int) -> None:
    """utils.py:297-304: indent=2 JSON followed by a newline."""
    path = Path(stats_folder) / f"stats_{target_vocab_size}.json"
    logging.info(f"round statistics -> {path.as_posix()}")
    with open(path, "w") as out:
        json.dump(stats, out, indent=2)
        out.write("\n")


def load_corpus(corpus_filepath, partial_corpus_filepath: Optional[Path], partial_corpus_line_number: int) -> List[str]:
    """utils.py:71-103 with the reference's return type: reuse the partial corpus file if it exists, else shuffle the
    full corpus with the already-seeded `random`, keep the first N*1000 lines and save them.  Lines keep their
    trailing newline.  (`build_vocab` itself uses `load_corpus_packed`: same lines, same file, no Python strings.)"""
    corpus_filepath = Path(corpus_filepath)
    if isinstance(partial_corpus_filepath, str):
        partial_corpus_filepath = Path(partial_corpus_filepath)
    started = time.time()
    if partial_corpus_filepath and partial_corpus_filepath.exists():
        with open(partial_corpus_filepath, "r") as src:
            partial = src.readlines()
        logging.info(f"partial corpus {partial_corpus_filepath.as_posix()}: {len(partial)} lines in {time.time() - started:.2f} s")
        return partial
    with open(corpus_filepath, "r") as src:
        everything = src.readlines()
    random.shuffle(everything)
    logging.info(f"corpus {corpus_filepath.as_posix()}: {len(everything)} lines read and shuffled in {time.time() - started:.2f} s")
    partial = everything[:partial_corpus_line_number * 1000]
    if partial_corpus_filepath is None:
        partial_corpus_filepath = getDataFolder() / f"{corpus_filepath.stem}_{len(partial)}.txt"
    with open(partial_corpus_filepath, "w+") as out:
        out.writelines(partial)
    logging.info(f"kept {len(partial)} lines, saved as {partial_corpus_filepath.as_posix()}")
    return partial


def _split_lines(data: np.ndarray) -> np.ndarray:
    """Line offsets of a byte buffer under `readlines()` rules for text without carriage returns: a line ends after
    each b"\n"; a last line without one is kept as it is (utils.py:79,84)."""
    ends = np.flatnonzero(data == 0x0A) + 1
    if len(ends) == 0 or ends[-1] != len(data):
        ends = np.append(ends, len(data))
    off = np.zeros(len(ends) + 1, dtype=np.int64)
    off[1:] = ends
    return off if len(data) else np.zeros(1, dtype=np.int64)


def _read_bytes(path: Path) -> np.ndarray:
    """The file as uint8, memory-mapped.  Text-mode reading translates "\r\n" and "\r" to "\n" (universal
    newlines); a file that contains carriage returns is translated the same way here, in memory."""
    if path.stat().st_size == 0:
        return np.zeros(0, dtype=np.uint8)
    data = np.memmap(path, dtype=np.uint8, mode="r")
    if (data == 0x0D).any():
        data = np.frombuffer(bytes(data).replace(b"\r\n", b"\n").replace(b"\r", b"\n"), dtype=np.uint8)
    return data


def load_corpus_packed(corpus_filepath, partial_corpus_filepath: Optional[Path], partial_corpus_line_number: int,
                       write: bool = True, reuse: Optional[bool] = None):
    """`load_corpus` (utils.py:71-103) straight into the device layout: a `PackedCorpus` (uint8 bytes + int64 line
    offsets) from a memory map, never a Python string per line.  Same lines in the same order as the reference under
    the same `random` state -- `random.shuffle` consumes the generator as a function of the list LENGTH only, so
    shuffling the line indices gives the reference's permutation -- and the same partial-corpus file.  Bytes are
    taken as they are: the reference would raise UnicodeDecodeError on a file that is not valid UTF-8, this does not.
    `write=False` (ranks other than 0 of a multi-GPU run) skips writing the partial-corpus file; `reuse` overrides the
    "does the partial-corpus file exist" test (the ranks of a multi-GPU run decide it together, before rank 0 writes)."""
    from .parallel import PackedCorpus
    corpus_filepath = Path(corpus_filepath)
    if isinstance(partial_corpus_filepath, str):
        partial_corpus_filepath = Path(partial_corpus_filepath)
    started = time.time()
    if reuse is None:
        reuse = bool(partial_corpus_filepath and partial_corpus_filepath.exists())
    if reuse:
        data = _read_bytes(partial_corpus_filepath)
        off = _split_lines(data)
        logging.info(f"partial corpus {partial_corpus_filepath.as_posix()}: {len(off) - 1} lines, {len(data)} bytes "
                     f"in {time.time() - started:.2f} s")
        return PackedCorpus(data, off)
    data = _read_bytes(corpus_filepath)
    off = _split_lines(data)
    n = len(off) - 1
    order = list(range(n))
    random.shuffle(order)              # the permutation random.shuffle(lines) applies: result[i] = lines[order[i]]
    keep = np.asarray(order[:partial_corpus_line_number * 1000], dtype=np.int64)
    lens = (off[1:] - off[:-1])[keep]
    new_off = np.zeros(len(keep) + 1, dtype=np.int64)
    np.cumsum(lens, out=new_off[1:])
    out = np.empty(int(new_off[-1]), dtype=np.uint8)
    step = 1 << 18                     # gather in blocks so the index array stays small
    for lo in range(0, len(keep), step):
        hi = min(len(keep), lo + step)
        src = np.repeat(off[keep[lo:hi]] - new_off[lo:hi], lens[lo:hi]) + np.arange(new_off[lo], new_off[hi], dtype=np.int64)
        out[new_off[lo]:new_off[hi]] = data[src]
    logging.info(f"corpus {corpus_filepath.as_posix()}: {n} lines read and shuffled, {len(keep)} kept, "
                 f"{time.time() - started:.2f} s")
    if write:
        if partial_corpus_filepath is None:
            partial_corpus_filepath = getDataFolder() / f"{corpus_filepath.stem}_{len(keep)}.txt"
        with open(partial_corpus_filepath, "wb") as dst:
            dst.write(out.tobytes() if len(out) < (1 << 30) else memoryview(out))
        logging.info(f"partial corpus saved as {partial_corpus_filepath.as_posix()}")
    return PackedCorpus(out, new_off)


def load_corpus_for_rank(corpus_filepath, partial_corpus_filepath, partial_corpus_line_number: int, rank: int):
    """Every rank of a multi-GPU run gets the same partial corpus.  Whether the partial-corpus file is reused is decided
    once, by what is on disk BEFORE anyone writes: if it exists every rank reads it; if not, every rank shuffles the
    full corpus from the same `random` state (set_random_seed) and rank 0 alone writes the file.  (A rank that found the
    file rank 0 had just written would re-split it -- and a line that lost its newline in the shuffle merges with its
    successor there, utils.py:75-80 vs :84-96 -- so the ranks would disagree on the line count.)  The ranks then compare
    line and byte counts and fail together if they differ."""
    from . import dist as sdist
    path = Path(partial_corpus_filepath) if isinstance(partial_corpus_filepath, str) else partial_corpus_filepath
    reuse = bool(path and path.exists())
    sdist.barrier()                    # nobody writes before everyone has looked
    corpus = load_corpus_packed(corpus_filepath, path, partial_corpus_line_number, write=rank == 0, reuse=reuse)
    sdist.barrier()                    # the file is complete before anyone goes on
    sdist.assert_same_on_all_ranks((len(corpus), int(corpus.offsets[-1]), int(reuse)), "partial corpus (lines, bytes, reused)")
    return corpus


def init_logger(experiment_name: str, rank: int = 0) -> None:
    """utils.py:260-268; ranks other than 0 of a multi-GPU run log to a file of their own."""
    suffix = f"_rank{rank}" if rank else ""
    log_file = getLogsFolder() / f"{experiment_name}_{time.strftime('%Y%m%d_%H%M%S')}{suffix}.log"
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


def set_random_seed(experiment_name: str, random_seed: int, write: bool = True) -> None:
    """utils.py:286-294: seed.txt, then python and numpy generators."""
    root = getResultsFolder() / experiment_name
    root.mkdir(parents=True, exist_ok=True)
    if write:
        with open(root / "seed.txt", "w+") as out:
            out.write(str(random_seed))
    random.seed(random_seed)
    np.random.seed(random_seed)
