"""Matcher timing on 300k lines of the bench corpus: python tools/matchbench.py [mode ...] (0 queued, 2 column probes, 1 plain)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from sage_amd import synth
from sage_amd import _lib
from sage_amd._lib import HipSage
import os, pathlib
if os.environ.get('SAGE_LIB'): _lib.LIB_PATH = pathlib.Path(os.environ['SAGE_LIB'])
modes = [int(a) for a in sys.argv[1:]] or [0]
seed = synth.DEFAULT_SEED
head, ho = synth.make_corpus(14000, seed=seed, style="en", n_types=1_000_000)
V = int(os.environ.get('MB_V', '262144'))
vocab = synth.make_vocab(head, ho, V, 16, sample_bytes=6_000_000)
import collections
print('V', V, 'key length histogram', sorted(collections.Counter(len(k) for k in vocab).items()), flush=True)
corpus, off = synth.make_corpus(300_000, seed=seed, style="en", n_types=1_000_000)
e = HipSage(0); e.set_vocab(vocab, None, 16); e.set_corpus(corpus, off)
ref = None
for m in modes:
    e.set_match_mode(m)
    ids = e.tokenize(0, len(off) - 1)[0]
    if ref is None: ref = ids
    same = np.array_equal(ref, ids)
    e.set_profiling(True); e.reset_kernel_times()
    for _ in range(3): e.token_counts(0, len(off) - 1)
    kt = e.kernel_times(); print("mode", m, "match ms per pass", kt['match'][0] / 3, "bytes", len(corpus), "tokens equal to first mode:", same, flush=True)
    e.set_profiling(False)

try:
    import ctypes as C
    lib = C.CDLL(str(_lib.LIB_PATH))
    out = (C.c_ulonglong * 16)()
    lib.sage_debug_match_stats(out, 1)            # reset
    e.set_match_mode(0); e.token_counts(0, len(off) - 1)
    lib.sage_debug_match_stats(out, 0)
    names = ["13+", "9..12", "7..8", "5..6", "3..4"]
    nb = len(corpus)
    for i, n in enumerate(names):
        print(f"class {n}: candidates per byte {out[2*i]/nb:.3f}, hits per byte {out[2*i+1]/nb:.3f}")
except AttributeError:
    pass
