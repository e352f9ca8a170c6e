"""Synthetic inputs for tests and bench: a Wikipedia-shaped English corpus, a Turkish-style
multi-byte UTF-8 corpus, a frequency-ordered substring vocabulary and gensim-shaped float64 embeddings.

Nothing here is on the product path; it only manufactures inputs of the shape the reference is run on
(`/root/reference/run.sh:1-10`, `/root/reference/README.md:11-13`).  Everything is vectorised numpy so a
1M-line / ~0.6 GB corpus is produced in well under a minute.
"""
from __future__ import annotations

import numpy as np

DEFAULT_SEED = 692653  # the reference's default random_seed (SaGeVocabBuilder.py:17)

_EN_LETTERS = "etaoinshrdlcumwfgypbvkjxqz"
_TR_LETTERS = ["a", "e", "i", "n", "r", "l", "ı", "k", "d", "m", "t", "y", "s", "u", "o", "b", "ü", "ş",
               "z", "g", "ç", "h", "ğ", "v", "c", "ö", "p", "f", "İ", "j"]


def _alphabet(style: str):
    letters = list(_EN_LETTERS) if style == "en" else list(_TR_LETTERS)
    enc = [ch.encode("utf-8") for ch in letters]
    weights = (np.arange(1, len(enc) + 1, dtype=np.float64)) ** -0.7
    return enc, weights / weights.sum()


def _ragged_concat(pieces_flat: np.ndarray, pieces_off: np.ndarray, picks: np.ndarray):
    """Concatenate pieces[picks[0]], pieces[picks[1]], ... ; returns (flat uint8, lengths int64)."""
    lens = (pieces_off[1:] - pieces_off[:-1])[picks]
    total = int(lens.sum())
    starts_out = np.zeros(len(picks), dtype=np.int64)
    if len(picks) > 1:
        np.cumsum(lens[:-1], out=starts_out[1:])
    src = np.repeat(pieces_off[:-1][picks] - starts_out, lens) + np.arange(total, dtype=np.int64)
    return pieces_flat[src], lens


def make_lexicon(rng: np.random.Generator, n_types: int, style: str = "en"):
    """Word types as (flat uint8 bytes, int64 offsets).  English: lowercase words, length
    clip(Gamma(4,1.2),1,14), letters ~ rank^-0.7.  Turkish style: stem + 0..5 suffixes from a 200-suffix
    inventory with ~25 % two-byte UTF-8 letters."""
    enc, w = _alphabet(style)
    char_len = np.array([len(e) for e in enc], dtype=np.int64)
    char_off = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum(char_len, out=char_off[1:])
    char_flat = np.frombuffer(b"".join(enc), dtype=np.uint8)

    def random_strings(count, nchars):
        picks = rng.choice(len(enc), size=int(nchars.sum()), p=w)
        flat, blens = _ragged_concat(char_flat, char_off, picks)
        # bytes per string = sum of blens over its characters
        ends = np.cumsum(nchars)
        csum = np.concatenate([[0], np.cumsum(blens)])
        off = np.concatenate([[0], csum[ends]])
        return flat, off.astype(np.int64)

    if style == "en":
        nchars = np.clip(np.rint(rng.gamma(4.0, 1.2, size=n_types)), 1, 14).astype(np.int64)
        flat, off = random_strings(n_types, nchars)
    else:
        n_stems = max(16, n_types // 6)
        stem_chars = np.clip(np.rint(rng.gamma(3.0, 1.1, size=n_stems)), 2, 8).astype(np.int64)
        sflat, soff = random_strings(n_stems, stem_chars)
        suf_chars = rng.integers(1, 4, size=200).astype(np.int64)
        xflat, xoff = random_strings(200, suf_chars)
        # piece table = stems followed by suffixes
        pflat = np.concatenate([sflat, xflat])
        poff = np.concatenate([soff, xoff[1:] + soff[-1]])
        nsuf = rng.integers(0, 6, size=n_types)
        stem_pick = rng.integers(0, n_stems, size=n_types)
        # build each word = stem + nsuf suffixes
        tot = int(n_types + nsuf.sum())
        picks = np.empty(tot, dtype=np.int64)
        wstart = np.zeros(n_types, dtype=np.int64)
        np.cumsum((nsuf + 1)[:-1], out=wstart[1:])
        picks[:] = n_stems + rng.integers(0, 200, size=tot)
        picks[wstart] = stem_pick
        flat, plens = _ragged_concat(pflat, poff, picks)
        csum = np.concatenate([[0], np.cumsum(plens)])
        off = np.concatenate([[0], csum[np.cumsum(nsuf + 1)]]).astype(np.int64)
    # ~200 number / punctuation types that get 3 % of the probability mass
    extra = [str(int(x)).encode() for x in rng.integers(0, 2100, size=150)]
    extra += [b",", b";", b":", b"(", b")", b"-", b"\"", b"'", b"%", b"&"] * 5
    eflat = np.frombuffer(b"".join(extra), dtype=np.uint8)
    eoff = np.concatenate([[0], np.cumsum([len(e) for e in extra])]).astype(np.int64)
    flat = np.concatenate([flat, eflat])
    off = np.concatenate([off, eoff[1:] + off[-1]])
    return flat, off, len(extra)


def make_corpus(n_lines: int, seed: int = DEFAULT_SEED, style: str = "en", n_types: int | None = None,
                mean_log_words: float = 4.3, sigma_log_words: float = 0.6, chunk_lines: int = 50_000,
                progress=None, line_seed: int = 0):
    """Returns (corpus uint8[nbytes], line_offsets int64[n_lines+1]); every line ends with b'.\\n'.

    Words/line ~ max(5, LogNormal(mean_log_words, sigma_log_words)) (defaults: ~580 B/line); word types
    ~ Zipf(1.05); first word capitalised (ASCII letters only).  `seed` fixes the lexicon (the language);
    `line_seed` != 0 draws other lines of the same language (the shards of a multi-GPU run must share one language: the
    vocabulary is built from a sample of `seed`'s text, and text of another lexicon tokenizes into 1.7x the tokens)."""
    rng = np.random.default_rng(seed)
    if n_types is None:
        n_types = 50_000 if n_lines <= 200_000 else 1_000_000
    lflat, loff, n_extra = make_lexicon(rng, n_types, style)
    if line_seed:
        rng = np.random.default_rng([seed, line_seed])
    n_all = len(loff) - 1
    p = np.empty(n_all, dtype=np.float64)
    p[:n_all - n_extra] = (np.arange(1, n_all - n_extra + 1, dtype=np.float64)) ** -1.05
    p[:n_all - n_extra] *= 0.97 / p[:n_all - n_extra].sum()
    p[n_all - n_extra:] = 0.03 / n_extra
    cdf = np.cumsum(p)
    cdf[-1] = 1.0
    sep = np.frombuffer(b" .\n", dtype=np.uint8)
    # piece table = lexicon + [" ", ".\n"]
    pflat = np.concatenate([lflat, sep])
    poff = np.concatenate([loff, [loff[-1] + 1, loff[-1] + 3]]).astype(np.int64)
    SPACE, EOL = n_all, n_all + 1

    out_chunks, len_chunks = [], []
    for c0 in range(0, n_lines, chunk_lines):
        nl = min(chunk_lines, n_lines - c0)
        wl = np.maximum(5, np.rint(rng.lognormal(mean_log_words, sigma_log_words, size=nl))).astype(np.int64)
        nw = int(wl.sum())
        wid = np.searchsorted(cdf, rng.random(nw), side="right").astype(np.int64)
        np.minimum(wid, n_all - 1, out=wid)
        picks = np.empty(2 * nw, dtype=np.int64)
        picks[0::2] = wid
        picks[1::2] = SPACE
        line_last = np.cumsum(wl) - 1
        picks[2 * line_last + 1] = EOL
        flat, plens = _ragged_concat(pflat, poff, picks)
        csum = np.cumsum(plens)
        line_end = csum[2 * line_last + 1]
        line_start = np.concatenate([[0], line_end[:-1]])
        first = flat[line_start]
        lower = (first >= 97) & (first <= 122)
        flat[line_start[lower]] = first[lower] - 32
        out_chunks.append(flat)
        if progress is not None and (c0 // chunk_lines) % 4 == 3:
            progress(f"generated {c0 + nl} lines")
        len_chunks.append(line_end - line_start)
    corpus = np.concatenate(out_chunks) if len(out_chunks) > 1 else out_chunks[0]
    lens = np.concatenate(len_chunks)
    offsets = np.zeros(n_lines + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    return np.ascontiguousarray(corpus), offsets


def corpus_lines(corpus: np.ndarray, offsets: np.ndarray):
    """List[str] view of the corpus, the shape `run_sage_parallel` takes (utils.py:141)."""
    buf = corpus.tobytes()
    return [buf[offsets[i]:offsets[i + 1]].decode("utf-8") for i in range(len(offsets) - 1)]


def make_vocab(corpus: np.ndarray, offsets: np.ndarray, vocab_size: int, max_len: int = 16,
               sample_bytes: int = 6_000_000):
    """256 single bytes followed by the most frequent byte substrings (2..max_len bytes, never crossing a
    line end) of a corpus sample, in descending frequency: the BPE-like initial vocabulary of SURVEY.md §8d.
    Returns List[bytes] of exactly `vocab_size` distinct keys (fewer only if the sample is too small)."""
    assert vocab_size >= 256
    n_lines = len(offsets) - 1
    last = int(np.searchsorted(offsets, sample_bytes, side="left"))
    last = max(1, min(n_lines, last))
    nb = int(offsets[last])
    data = corpus[:nb].astype(np.uint64)
    # bytes remaining in the line, for the "does not cross a line end" rule
    line_of = np.searchsorted(offsets[1:last + 1], np.arange(nb), side="right")
    rem = offsets[1:last + 1][line_of] - np.arange(nb)
    need = vocab_size - 256
    P = np.uint64(0x100000001B3)
    h = data.copy()
    cand_cnt, cand_len, cand_idx = [], [], []
    with np.errstate(over="ignore"):
        for L in range(2, max_len + 1):
            h = h[:nb - L + 1] * P + data[L - 1:]
            valid = rem[:nb - L + 1] >= L
            if not valid.any():
                break
            hv = h[valid]
            pos = np.nonzero(valid)[0]
            _, first, cnt = np.unique(hv, return_index=True, return_counts=True)
            keep = np.argsort(-cnt, kind="stable")[:need]
            cand_cnt.append(cnt[keep])
            cand_len.append(np.full(len(keep), L, dtype=np.int64))
            cand_idx.append(pos[first[keep]])
    vocab = [bytes([i]) for i in range(256)]
    if need > 0 and cand_cnt:
        cnt = np.concatenate(cand_cnt)
        ln = np.concatenate(cand_len)
        idx = np.concatenate(cand_idx)
        order = np.lexsort((idx, ln, -cnt))
        raw = corpus[:nb].tobytes()
        seen = set()
        for o in order:
            tok = raw[idx[o]:idx[o] + ln[o]]
            if tok not in seen:
                seen.add(tok)
                vocab.append(tok)
                if len(vocab) == vocab_size:
                    break
    return vocab


def make_embeddings(vocab_size: int, dim: int = 50, seed: int = DEFAULT_SEED, unseen: np.ndarray | None = None) -> np.ndarray:
    """[V, D] float64 shaped like `train_embeddings` output (embeddings.py:77-89): float32 word2vec vectors
    upcast to float64 for the tokens the tokenized corpus contains, the float64 uniform(+-0.5/D) draw of
    embeddings.py:86 for the others.  `unseen` marks the latter rows; by default 2 % of the rows, at random."""
    rs = np.random.RandomState(seed % (2 ** 32))
    emb = rs.normal(0.0, 0.25, size=(vocab_size, dim)).astype(np.float32).astype(np.float64)
    miss = rs.random_sample(vocab_size) < 0.02
    if unseen is not None:
        miss = np.asarray(unseen, dtype=bool)
    emb[miss] = rs.uniform(-0.5 / dim, 0.5 / dim, size=(int(miss.sum()), dim))
    return emb
