"""Loaders for the committed golden vectors (tests/golden/, written by tests/golden/make_golden.py)."""
import gzip
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def unpack(flat: np.ndarray, off: np.ndarray):
    raw = flat.tobytes()
    return [raw[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def load_npz(name):
    return np.load(GOLDEN / name, allow_pickle=False)


def load_json_gz(name):
    with gzip.open(GOLDEN / name, "rt") as f:
        return json.load(f)


def front_decode(prefix_len: np.ndarray, suffix_len: np.ndarray, suffix_bytes: np.ndarray):
    """Keys of a sorted, front-coded list (tests/golden/make_readme_schedule.py): key i = the first prefix_len[i] bytes of
    key i-1 followed by its own suffix."""
    raw = suffix_bytes.tobytes()
    out, last, at = [], b"", 0
    for p, n in zip(prefix_len.tolist(), suffix_len.tolist()):
        last = last[:p] + raw[at:at + n]
        at += n
        out.append(last)
    return out
