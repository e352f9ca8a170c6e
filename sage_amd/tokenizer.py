"""SaGeTokenizer: the reference's tokenizer API (model.py:13-255) with the work done by the HIP library.

State (the four dicts) lives on the host exactly as in the reference, because SaGeVocabBuilder and the file
writers read it (SaGeVocabBuilder.py:57,68,83,109,159,166-192).  Matching itself -- `tokenize` and everything
built on it -- runs on the GPU through libsage_hip.so; there is no CPU matcher in this package.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple, Union

import numpy as np

from .hf_bytes import HFEncoding

Tokenizable = Union[str, bytes]


def require_all_single_bytes(vocab: Dict[bytes, int]) -> None:
    """model.py:251-255."""
    for b in range(256):
        key = bytes([b])
        if key not in vocab:
            raise Exception(f"missing byte {key}")


class SaGeTokenizer:
    def __init__(self, initial_vocabulary: Sequence[bytes], max_len: int = 16):
        self.hfe = HFEncoding()
        self.byte_vocab: Dict[bytes, int] = None
        self.inv_byte_vocab: Dict[int, bytes] = None
        self._str_vocab = None              # the two string-keyed dicts are built when first read (files, gensim)
        self._inv_str_vocab = None
        self._encoded: Dict[bytes, str] = {}    # HF-encoded form of every key seen so far (keys survive prunes)
        self._packed = None                 # (key bytes uint8, key offsets int64, ids int32): what the engine uploads
        self.prune_map = None
        self._id_space = 0
        self.vocab_version = 0      # bumped on every set_vocabulary; engines re-upload when it changes
        self.max_len = max_len
        self.set_vocabulary(initial_vocabulary)

    # ---- vocabulary state (model.py:28-79) ----------------------------------------------------------------
    @staticmethod
    def set_bytes_vocab(new_vocab: Sequence[bytes]) -> Dict[bytes, int]:
        table: Dict[bytes, int] = {}
        for position, key in enumerate(new_vocab):
            assert type(key) == bytes
            table[key] = position           # a repeated key keeps its last position, as a dict does
        return table

    def set_vocabulary(self, new_vocab: Sequence[bytes]) -> None:
        table = self.set_bytes_vocab(new_vocab)
        require_all_single_bytes(table)
        # A vocabulary that only lost keys (the builder's prune, SaGeVocabBuilder.py:159-192) is remembered as such: the
        # engine then moves its resident matches instead of matching the corpus again (sage_set_vocab_pruned).
        self.prune_map = None               # (version it applies to, old id -> new id or -1)
        before = self.byte_vocab
        if before is not None and len(table) <= len(before):
            get = before.get
            old_ids = np.fromiter((get(k, -1) for k in table), dtype=np.int64, count=len(table))
            if old_ids.size and old_ids.min() >= 0:
                o2n = np.full(self._id_space, -1, dtype=np.int32)
                o2n[old_ids] = np.fromiter(table.values(), dtype=np.int32, count=len(table))
                self.prune_map = (self.vocab_version, o2n)
        self.byte_vocab = table
        self.inv_byte_vocab = {i: k for k, i in table.items()}
        self._str_vocab = self._inv_str_vocab = None
        self._id_space = len(new_vocab)     # ids are list positions (model.py:44-51)
        # The device table is packed here, where the reference rebuilds its own lookup tables, so that the ablation
        # step that follows a prune (SaGeVocabBuilder.py:192 -> :100) only has to hash and upload it.
        keys = new_vocab if len(table) == len(new_vocab) else list(table.keys())
        ids = None if keys is new_vocab else np.fromiter(table.values(), dtype=np.int32, count=len(table))
        off = np.zeros(len(keys) + 1, dtype=np.int64)
        np.cumsum(np.fromiter(map(len, keys), dtype=np.int64, count=len(keys)), out=off[1:])
        self._packed = (np.frombuffer(b"".join(keys), dtype=np.uint8), off, ids)
        self.vocab_version += 1

    def _build_str_vocabs(self) -> None:
        enc, cache = self.hfe.to_encoded, self._encoded
        out = {}
        for k, i in self.byte_vocab.items():
            s = cache.get(k)
            if s is None:
                s = cache[k] = enc(k)
            out[s] = i
        self._str_vocab = out
        self._inv_str_vocab = {i: s for s, i in out.items()}

    @property
    def str_vocab(self) -> Dict[str, int]:
        if self._str_vocab is None:
            self._build_str_vocabs()
        return self._str_vocab

    @property
    def inv_str_vocab(self) -> Dict[int, str]:
        if self._inv_str_vocab is None:
            self._build_str_vocabs()
        return self._inv_str_vocab

    def id_to_bytes(self, token_id: int) -> bytes:
        return self.inv_byte_vocab[token_id]

    def id_to_encoded(self, token_id: int) -> str:
        return self.inv_str_vocab[token_id]

    def get_vocabulary(self) -> Dict[bytes, int]:
        return self.byte_vocab

    def vocab_size(self) -> int:
        return len(self.byte_vocab)

    def id_space(self) -> int:
        """Largest id + 1: the length of the id-indexed result vectors (== vocab_size() unless a key was repeated)."""
        return self._id_space

    def print_tokens(self, ids: List[int]) -> List[bytes]:
        return [self.inv_byte_vocab[i] for i in ids]

    def add_all_byte_ids(self, vocab: Dict[int, float], score: float = 1e400) -> None:
        for b in range(256):
            vocab[self.byte_vocab[bytes([b])]] = score

    # ---- matching, on the GPU (model.py:81-113) -----------------------------------------------------------
    def tokenize_lines(self, lines: Sequence[Tokenizable]):
        """Batch form of `tokenize`: -> (ids, starts, widths int32 arrays back to back, counts per line)."""
        from . import parallel
        return parallel.tokenize_lines(self, lines)

    def tokenize(self, sent: Tokenizable, tokens_only: bool = False):
        ids, starts, widths, _ = self.tokenize_lines([sent])
        if tokens_only:
            return ids.tolist()
        return list(zip(ids.tolist(), starts.tolist(), widths.tolist()))

    def tokenize_to_encoded_str(self, sent: Tokenizable) -> List[str]:
        inv = self.inv_str_vocab
        return [inv[i] for i in self.tokenize(sent, tokens_only=True)]

    def tokenize_to_bytes(self, sent: Tokenizable) -> List[bytes]:
        inv = self.inv_byte_vocab
        return [inv[i] for i in self.tokenize(sent, tokens_only=True)]

    # ---- what the engine uploads --------------------------------------------------------------------------
    def key_table(self) -> Tuple[List[bytes], np.ndarray]:
        keys = list(self.byte_vocab.keys())
        ids = np.fromiter(self.byte_vocab.values(), dtype=np.int32, count=len(keys))
        return keys, ids

    def packed_key_table(self):
        """(key bytes uint8, key offsets int64[n+1], ids int32[n] or None = positions), packed by set_vocabulary."""
        return self._packed
