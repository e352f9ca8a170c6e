#!/usr/bin/env python3
"""Throughput of sage_train_embeddings on the bench corpus shape (not part of bench.py's contract).

    python tools/bench_w2v.py --lines 200000 --epochs 1
"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=200_000)
    ap.add_argument("--vocab", type=int, default=262_144)
    ap.add_argument("--dim", type=int, default=50)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--negative", type=int, default=15)
    ap.add_argument("--window", type=int, default=5)
    args = ap.parse_args()
    from sage_amd import synth
    from sage_amd._lib import HipSage
    head, head_off = synth.make_corpus(min(args.lines, 14_000), seed=synth.DEFAULT_SEED, n_types=1_000_000)
    vocab = synth.make_vocab(head, head_off, args.vocab, 16, sample_bytes=6_000_000)
    corpus, off = synth.make_corpus(args.lines, seed=synth.DEFAULT_SEED, n_types=1_000_000)
    eng = HipSage(0)
    eng.set_vocab(vocab, None, 16)
    eng.set_corpus(corpus, off)
    t0 = time.time()
    vec, counts = eng.train_embeddings(len(vocab), args.dim, window=args.window, negative=args.negative, epochs=args.epochs)
    dt = time.time() - t0
    tokens = int(counts.sum())
    print(json.dumps({"lines": args.lines, "tokens": tokens, "epochs": args.epochs, "seconds": dt,
                      "trained_tokens_per_s": tokens * args.epochs / dt, "vocab": len(vocab), "dim": args.dim,
                      "negative": args.negative, "window": args.window,
                      "rows_trained": int((counts > 0).sum())}))


if __name__ == "__main__":
    main()
