"""GPU: sage_tokenize_noalloc (the allocation-free variant of sage_tokenize, include/sage_hip.h) through the C-ABI,
against sage_tokenize itself on the same lines -- bit-exact ids, starts, widths and per-line counts -- on two calls in
a row, the second with a smaller input (the scratch array keeps the first call's contents).  torch only provides the
caller-side device buffers here; the entry point itself allocates nothing."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _expected(eng, flat, off):
    eng.set_corpus(flat, off)
    return eng.tokenize(0, len(off) - 1)


def _noalloc(eng, flat, off, capacity):
    import torch
    n_lines = len(off) - 1
    d_text = torch.from_numpy(np.ascontiguousarray(flat)).cuda() if len(flat) else torch.zeros(1, dtype=torch.uint8, device="cuda")
    d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.int64)).cuda()
    d_ids = torch.full((max(capacity, 1),), -7, dtype=torch.int32, device="cuda")
    d_starts = torch.full_like(d_ids, -7)
    d_widths = torch.full_like(d_ids, -7)
    d_counts = torch.full((max(n_lines, 1),), -7, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    total = eng.tokenize_noalloc(d_text.data_ptr(), len(flat), d_off.data_ptr(), n_lines, d_ids.data_ptr(),
                                 d_starts.data_ptr(), d_widths.data_ptr(), capacity, d_counts.data_ptr())
    torch.cuda.synchronize()
    out = (d_ids.cpu().numpy(), d_starts.cpu().numpy(), d_widths.cpu().numpy(), d_counts.cpu().numpy()[:n_lines])
    assert (out[0][total:] == -7).all()           # nothing is written past the tokens
    return total, out


@pytest.mark.parametrize("style,max_len", [("en", 16), ("tr", 17)])
def test_noalloc_tokenize_matches_tokenize_twice(style, max_len):
    from sage_amd import synth
    from sage_amd._lib import HipSage
    corpus, off = synth.make_corpus(1500, seed=31 + max_len, style=style, n_types=3000, mean_log_words=3.2)
    vocab = synth.make_vocab(corpus, off, 4000, max_len, sample_bytes=400_000)
    eng = HipSage(0)
    other = HipSage(0)                 # the expected values come from a context of their own
    try:
        eng.set_vocab(vocab, None, max_len)
        other.set_vocab(vocab, None, max_len)
        assert len(corpus) <= HipSage.NOALLOC_MAX_BYTES
        # call 1: the whole corpus; call 2: a smaller, different slice of it (lines 200..500)
        lo, hi = 200, 500
        small_off = off[lo:hi + 1] - off[lo]
        small = corpus[off[lo]:off[hi]]
        for flat, offs in ((corpus, off), (small, small_off)):
            want_ids, want_starts, want_widths, want_counts = _expected(other, flat, offs)
            total, (ids, starts, widths, counts) = _noalloc(eng, flat, offs, capacity=len(want_ids) + 5)
            assert total == len(want_ids)
            assert counts.tolist() == want_counts.tolist()
            assert ids[:total].tolist() == want_ids.tolist()
            assert starts[:total].tolist() == want_starts.tolist()
            assert widths[:total].tolist() == want_widths.tolist()
        # counts only; then limits and argument checks are refused with a message, before anything is launched
        import torch
        d_text = torch.from_numpy(np.ascontiguousarray(small)).cuda()
        d_off = torch.from_numpy(np.ascontiguousarray(small_off)).cuda()
        assert eng.tokenize_noalloc(d_text.data_ptr(), len(small), d_off.data_ptr(), hi - lo) == total
        with pytest.raises(ValueError, match="SAGE_NOALLOC_MAX_BYTES"):
            eng.tokenize_noalloc(d_text.data_ptr(), HipSage.NOALLOC_MAX_BYTES + 1, d_off.data_ptr(), hi - lo)
        with pytest.raises(ValueError, match="SAGE_NOALLOC_MAX_LINES"):
            eng.tokenize_noalloc(d_text.data_ptr(), len(small), d_off.data_ptr(), HipSage.NOALLOC_MAX_LINES + 1)
        with pytest.raises(ValueError, match="from 0 to n_bytes"):
            eng.tokenize_noalloc(d_text.data_ptr(), len(small) - 1, d_off.data_ptr(), hi - lo)
        d_ids = torch.zeros(8, dtype=torch.int32, device="cuda")
        with pytest.raises(ValueError, match="capacity"):
            eng.tokenize_noalloc(d_text.data_ptr(), len(small), d_off.data_ptr(), hi - lo, d_ids.data_ptr(),
                                 d_ids.data_ptr(), d_ids.data_ptr(), 8)
    finally:
        eng.close()
        other.close()


@pytest.mark.parametrize("tag", ["en16", "tr17"])
def test_noalloc_tokenize_against_the_reference_golden(tag):
    """Independent of sage_tokenize: the (id, start, width) triples the REFERENCE's tokenize produced
    (tests/golden/tokenize_<tag>.npz, model.py:81-101), including the edge-case lines (one byte, no newline, a key at the
    line end, raw 0x00 0xff)."""
    from sage_amd._lib import HipSage
    from tests.golden_io import load_npz, unpack
    g = load_npz(f"tokenize_{tag}.npz")
    vocab = unpack(g["vocab_bytes"], g["vocab_off"])
    eng = HipSage(0)
    try:
        eng.set_vocab(vocab, None, int(g["max_len"]))
        for _ in range(2):              # twice: the second call meets the scratch array as the first one left it
            total, (ids, starts, widths, counts) = _noalloc(eng, g["line_bytes"], g["line_off"], capacity=len(g["ids"]))
            assert total == len(g["ids"])
            assert counts.tolist() == g["counts"].tolist()
            assert ids[:total].tolist() == g["ids"].tolist()
            assert starts[:total].tolist() == g["starts"].tolist()
            assert widths[:total].tolist() == g["widths"].tolist()
    finally:
        eng.close()
