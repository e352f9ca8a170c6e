import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from sage_amd import synth
from sage_amd._lib import HipSage
seed=synth.DEFAULT_SEED
head,ho=synth.make_corpus(14000,seed=seed,style="en",n_types=1_000_000)
vocab=synth.make_vocab(head,ho,262144,16,sample_bytes=6_000_000)
corpus,off=synth.make_corpus(300_000,seed=seed,style="en",n_types=1_000_000)
e=HipSage(0); e.set_vocab(vocab,None,16); e.set_corpus(corpus,off)
e.token_counts(0,len(off)-1)
e.set_profiling(True); e.reset_kernel_times()
for _ in range(3): e.token_counts(0,len(off)-1)
kt=e.kernel_times(); print("match ms per pass", kt['match'][0]/3, "bytes", len(corpus))
