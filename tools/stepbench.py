"""Per-kernel times of the ablation step on 300k lines of the bench corpus (quick A/B runs; SAGE_LIB=path picks another build, SB_LINES / SB_DIM the size)."""
import os, pathlib, sys, numpy as np
sys.path.insert(0, '/root/repo')
from sage_amd import _lib, synth
from sage_amd._lib import HipSage
from sage_amd.parallel import segment_line_bounds
if os.environ.get('SAGE_LIB'): _lib.LIB_PATH = pathlib.Path(os.environ['SAGE_LIB'])
N = int(os.environ.get('SB_LINES', '300000'))
seed = synth.DEFAULT_SEED
head, ho = synth.make_corpus(14000, seed=seed, style="en", n_types=1_000_000)
vocab = synth.make_vocab(head, ho, 262144, 16, sample_bytes=6_000_000)
if os.environ.get("SB_SEED_OFF"):      # text of ANOTHER lexicon than the vocabulary's: 1.7x the tokens, heavily repeated keys (fallback paths)
    corpus, off = synth.make_corpus(N, seed=seed + int(os.environ["SB_SEED_OFF"]), style="en", n_types=1_000_000)
else:
    corpus, off = synth.make_corpus(N, seed=seed, line_seed=int(os.environ.get("SB_LINE_SEED", "0")), style="en", n_types=1_000_000)
if os.environ.get('SB_SPLIT_LONG'):     # timing experiment: lines longer than this many bytes are cut (how much of the walks is their longest line?)
    T = int(os.environ['SB_SPLIT_LONG']); extra = []
    for i in np.nonzero(np.diff(off) > T)[0]:
        extra.extend(range(int(off[i]) + T, int(off[i + 1]), T))
    off = np.sort(np.concatenate([off, np.array(extra, dtype=off.dtype)])); N = len(off) - 1
    print("lines after the cut", N, "longest", int(np.diff(off).max()), flush=True)
e = HipSage(0); e.set_vocab(vocab, None, 16); e.set_corpus(corpus, off)
counts = np.bincount(e.tokenize(0, N)[0], minlength=len(vocab))
unseen = ~(counts > 0)
if os.environ.get('SB_STALE'):      # float64 rows where the corpus does use them, as between two retrain labels (ids moved, rows did not)
    unseen = np.random.RandomState(1).rand(len(vocab)) < float(os.environ['SB_STALE'])
emb = synth.make_embeddings(len(vocab), int(os.environ.get('SB_DIM', '50')), seed=seed, unseen=unseen)
if os.environ.get('SB_SCORE_MODE'): e.set_score_mode(int(os.environ['SB_SCORE_MODE']))
e.set_embeddings(emb)
b = segment_line_bounds(N, 4)
for _ in range(3): r = e.ablation_step(b)     # warm-up: the first steps on a fresh box run at ramping clocks
e.set_profiling(True); e.reset_kernel_times()
R = 5
for _ in range(R):
    if not os.environ.get('SB_WARM_MATCHES'): e.invalidate_matches()     # every step a cold one (matcher included)
    r = e.ablation_step(b)
kt = e.kernel_times()
print("tokens", r[2], "triples", r[3], {k: round(v[0] / R, 3) for k, v in kt.items() if v[0] > 0}, "sum", round(sum(v[0] for v in kt.values()) / R, 2),
      "launches/step", {k: v[1] / R for k, v in kt.items() if k in ("partition", "count", "match")}, flush=True)
