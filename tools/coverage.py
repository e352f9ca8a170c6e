import sys, numpy as np
sys.path.insert(0,'/root/repo')
from sage_amd import synth
from sage_amd._lib import HipSage
seed=synth.DEFAULT_SEED
head,ho=synth.make_corpus(14000,seed=seed,style="en",n_types=1_000_000)
vocab=synth.make_vocab(head,ho,262144,16,sample_bytes=6_000_000)
corpus,off=synth.make_corpus(300_000,seed=seed,style="en",n_types=1_000_000)
e=HipSage(0); e.set_vocab(vocab,None,16); e.set_corpus(corpus,off)
ids=e.tokenize(0,len(off)-1)[0]
c=np.sort(np.bincount(ids,minlength=len(vocab)))[::-1].astype(np.float64)
tot=c.sum()
for n in (256,640,1024,4096,16384,65536): print(n, round(c[:n].sum()/tot,3))
print('distinct used', int((c>0).sum()))
