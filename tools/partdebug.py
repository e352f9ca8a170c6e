"""Reproducer of the exact-size bucket overflow (DESIGN.md, round 2 open question): text of ANOTHER lexicon than the
vocabulary's, so that tiles hold hundreds of pooled occurrences and buckets fill with repeated keys.
SAGE_DEBUG_PART=1 makes the library compare its counting and scatter passes tile by tile and bucket by bucket and fail
instead of falling back.  PD_LINES / PD_RUNS / PD_SEED_OFF choose the size, the repeats and the lexicon."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sage_amd import synth
from sage_amd._lib import HipSage
from sage_amd.parallel import segment_line_bounds
N = int(os.environ.get("PD_LINES", "1000000"))
seed = synth.DEFAULT_SEED
t0 = time.time()
head, ho = synth.make_corpus(14000, seed=seed, style="en", n_types=1_000_000)
vocab = synth.make_vocab(head, ho, 262144, 16, sample_bytes=6_000_000)
corpus, off = synth.make_corpus(N, seed=seed + int(os.environ.get("PD_SEED_OFF", "1")), style="en", n_types=1_000_000)
print(f"inputs in {time.time() - t0:.0f} s: {len(corpus)} bytes", flush=True)
e = HipSage(0); e.set_vocab(vocab, None, 16); e.set_corpus(corpus, off)
counts = np.bincount(e.tokenize(0, N)[0], minlength=len(vocab))
e.set_embeddings(synth.make_embeddings(len(vocab), 50, seed=seed, unseen=~(counts > 0)))
b = segment_line_bounds(N, 4)
for run in range(int(os.environ.get("PD_RUNS", "4"))):
    t1 = time.time()
    try:
        r = e.ablation_step(b)
        st = e.step_stats()
        print(f"run {run}: tokens {r[2]} triples {r[3]} pairs {st['pairs']} redos {st['count_redos']} hbm_sets {st['count_from_hbm_sets']} "
              f"exact_buckets {st['exact_buckets']} in {time.time() - t1:.2f} s", flush=True)
    except Exception as ex:
        print(f"run {run}: FAILED {type(ex).__name__}: {ex}", flush=True)
