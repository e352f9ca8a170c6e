"""Byte <-> printable-character table of the GPT-2 / HuggingFace byte-level pre-tokenizer, used for gensim token
strings and sorted_losses_*.txt (reference: HFEncoding.py:5-39).  Printable latin-1 bytes map to themselves, the
other 68 bytes to U+0100.. in increasing byte order."""
from __future__ import annotations


def _build_table():
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, spill = [], 0x100
    for b in range(256):
        if b in keep:
            table.append(chr(b))
        else:
            table.append(chr(spill))
            spill += 1
    return table


class HFEncoding:
    _TABLE = _build_table()
    _INVERSE = {ch: b for b, ch in enumerate(_TABLE)}

    def __init__(self):
        self.byte_map = {bytes([b]): ch for b, ch in enumerate(self._TABLE)}
        self.inv_byte_map = {ch: bytes([b]) for b, ch in enumerate(self._TABLE)}

    def to_encoded(self, byte_str: bytes) -> str:
        t = self._TABLE
        return "".join(t[b] for b in byte_str)

    def to_bytes(self, s: str) -> bytes:
        inv = self._INVERSE
        return bytes(inv[ch] for ch in s)
