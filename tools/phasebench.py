"""Phase clocks of k_describe / k_logsig (debug build: hipcc ... -DPHASE_CLOCKS -o tools/abl/lib_phase.so; SAGE_LIB picks it)."""
import ctypes as C, os, pathlib, sys, numpy as np
sys.path.insert(0, '/root/repo')
from sage_amd import _lib, synth
from sage_amd._lib import HipSage
from sage_amd.parallel import segment_line_bounds
_lib.LIB_PATH = pathlib.Path(os.environ.get('SAGE_LIB', 'tools/abl/lib_phase.so'))
N = int(os.environ.get('SB_LINES', '1000000'))
seed = synth.DEFAULT_SEED
head, ho = synth.make_corpus(14000, seed=seed, style="en", n_types=1_000_000)
vocab = synth.make_vocab(head, ho, 262144, 16, sample_bytes=6_000_000)
corpus, off = synth.make_corpus(N, seed=seed, style="en", n_types=1_000_000)
e = HipSage(0); e.set_vocab(vocab, None, 16); e.set_corpus(corpus, off)
counts = np.bincount(e.tokenize(0, N)[0], minlength=len(vocab))
emb = synth.make_embeddings(len(vocab), 50, seed=seed, unseen=~(counts > 0))
e.set_embeddings(emb)
b = segment_line_bounds(N, 4)
for _ in range(2): e.ablation_step(b)
lib = C.CDLL(str(_lib.LIB_PATH))
out = (C.c_ulonglong * 48)()
lib.sage_debug_phase_clocks(out, 1)
e.ablation_step(b)
lib.sage_debug_phase_clocks(out, 0)
def show(name, base, labels, ntile_idx):
    v = [out[base + i] for i in range(len(labels))]
    tiles = out[base + ntile_idx]
    tot = sum(v)
    print(name, "tiles (thread 0's)", tiles, "clocks per tile", round(tot / max(tiles, 1)))
    for l, x in zip(labels, v):
        print(f"   {l:34s} {x / max(tiles, 1):9.0f}  {100.0 * x / max(tot, 1):5.1f} %")
show("k_key_partition3<1>", 32, ["clear + tile table", "load_inline (first round)", "phase A keys + count atomics", "barrier", "pooled A + barrier", "scan + reservations + barrier",
                                 "phase B (inline)", "pooled B + barrier", "copy-out + barrier"], 9)
show("k_describe<false>", 0, ["loads + span + barrier", "segment search", "walk", "bitmap", "outputs + atomics", "list ballots + barriers + atomic",
                              "list stores + seg atomics + barrier", "descriptor write-out + barrier"], 8)
show("k_logsig<false>", 16, ["tile set-up + 2 barriers", "loads (G window, dots)", "base sum", "owner search", "dot loop", "barrier", "accumulate"], 7)
