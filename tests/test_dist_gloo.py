"""CPU: the N>1 path (segment partition + combine) with two gloo ranks.  The per-rank step is played by the oracle
(this is a test: the product's step is HipSage.ablation_step); the combined result must equal a one-rank run:
integers exactly, losses to float64 rounding."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs():
    from sage_amd import synth
    corpus, off = synth.make_corpus(180, seed=21, n_types=900, mean_log_words=2.2)
    vocab = synth.make_vocab(corpus, off, 800, 16, sample_bytes=60_000)
    emb = synth.make_embeddings(len(vocab), 50, seed=4)
    return corpus, off, vocab, emb


def _oracle_step(lines, vocab, emb, vsize, hist_len=64):
    from oracle import sage_oracle as orc
    tok = orc.OracleTokenizer(vocab, 16)

    def step(local_bounds):
        loss = np.zeros(vsize)
        occ = np.zeros(vsize, dtype=np.int64)
        hist = np.zeros(hist_len, dtype=np.int64)
        tt = ttr = 0
        for lo, hi in zip(local_bounds[:-1], local_bounds[1:]):
            occs = {}
            l, t, r, s = orc.run_shard(tok, lines[lo:hi], emb, flush=10 ** 9, occurrences=occs)
            tt += t
            ttr += r
            for k, v in l.items():
                loss[k] += v
            for k, v in occs.items():
                occ[k] += v
            for k, v in s.items():
                hist[k] += v
        return loss, occ, tt, ttr, hist
    return step


def _rank_main(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from sage_amd import synth
    from sage_amd.dist import run_sharded_step
    from sage_amd.parallel import segment_line_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    corpus, off, vocab, emb = _inputs()
    lines = synth.corpus_lines(corpus, off)
    bounds = segment_line_bounds(len(lines), 3, flush=25)
    loss, occ, tt, ttr, hist = run_sharded_step(dist, bounds, off, _oracle_step(lines, vocab, emb, len(vocab)))
    np.savez(Path(out_dir) / f"rank{rank}.npz", loss=loss, occ=occ, tt=tt, ttr=ttr, hist=hist)
    # host-side pieces of a multi-rank build_vocab: rank 0's embeddings reach every rank; every rank loads the same
    # partial corpus and only rank 0 writes its file
    import random
    from sage_amd import hostio, setSageFolder
    from sage_amd.dist import broadcast_array, world_info
    assert world_info() == (rank, world)
    got = broadcast_array(emb if rank == 0 else None, src=0)
    assert got.dtype == emb.dtype and np.array_equal(got, emb)
    root = Path(out_dir) / f"sage_rank{rank}"
    root.mkdir()
    setSageFolder(root)
    (root / "c.txt").write_bytes(corpus.tobytes())
    random.seed(99)
    part = hostio.load_corpus_for_rank(root / "c.txt", None, 1, rank)        # all 180 lines, shuffled
    assert len(part) == 180
    np.savez(Path(out_dir) / f"corpus{rank}.npz", data=part.data, off=part.offsets,
             wrote=(root / "data" / "c_180.txt").exists())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_combine_equals_one_rank(tmp_path):
    import torch.multiprocessing as mp
    from sage_amd import synth
    from sage_amd.parallel import segment_line_bounds
    port = _free_port()
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    corpus, off, vocab, emb = _inputs()
    lines = synth.corpus_lines(corpus, off)
    bounds = segment_line_bounds(len(lines), 3, flush=25)
    loss, occ, tt, ttr, hist = _oracle_step(lines, vocab, emb, len(vocab))(bounds)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for r in (r0, r1):
        assert int(r["tt"]) == tt and int(r["ttr"]) == ttr
        assert r["occ"].tolist() == occ.tolist()
        assert r["hist"].tolist() == hist.tolist()
        np.testing.assert_allclose(r["loss"], loss, rtol=1e-12, atol=1e-12)
    assert r0["loss"].tolist() == r1["loss"].tolist()   # every rank holds the identical combined vector
    c0, c1 = np.load(tmp_path / "corpus0.npz"), np.load(tmp_path / "corpus1.npz")
    assert np.array_equal(c0["data"], c1["data"]) and np.array_equal(c0["off"], c1["off"])
    assert bool(c0["wrote"]) and not bool(c1["wrote"])
