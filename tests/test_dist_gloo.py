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


def _failing_rank_main(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from sage_amd.dist import StepBuffers, step_into
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Engine:           # stands in for HipSage: rank 1's step fails like SAGE_E_ROW does (IndexError)
        def ablation_step_device(self, bounds, d_loss, d_occ, d_tot, d_hist, pad=2):
            if rank == 1:
                raise IndexError("a token id is out of bounds for the embedding matrix")
    buf = StepBuffers(8, 4, device=torch.device("cpu"))
    outcome = "no exception"
    try:
        step_into(Engine(), np.array([0, 5]), buf, dist)
    except IndexError as exc:
        outcome = f"IndexError: {exc}"
    (Path(out_dir) / f"outcome{rank}.txt").write_text(outcome)
    # the group is still usable afterwards: nobody is stuck in a collective
    step_into(type("Ok", (), {"ablation_step_device": lambda *a, **k: None})(), np.array([0, 5]), buf, dist)
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_rank_fails_every_rank(tmp_path):
    """utils.py:167: a worker's exception surfaces in the parent.  Here: one rank's step raises, and every rank raises
    the same class instead of waiting forever in the combine."""
    import torch.multiprocessing as mp
    mp.spawn(_failing_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    o0, o1 = (tmp_path / "outcome0.txt").read_text(), (tmp_path / "outcome1.txt").read_text()
    assert o1 == "IndexError: a token id is out of bounds for the embedding matrix"
    assert o0.startswith("IndexError: the ablation step failed on rank 1 (IndexError)")


def _ragged_corpus_rank_main(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import random
    import torch.distributed as dist
    from sage_amd import hostio, setSageFolder
    dist.init_process_group("gloo", rank=rank, world_size=world)
    root = Path(out_dir)
    setSageFolder(root / f"sage_rank{rank}")
    random.seed(7)
    # the source's last line has no newline and the shuffle moves it into the middle of the kept lines; the partial
    # corpus file is named explicitly and shared by the ranks, and does not exist yet
    part = hostio.load_corpus_for_rank(root / "ragged.txt", str(root / "partial.txt"), 1, rank)
    np.savez(root / f"ragged{rank}.npz", data=part.data, off=part.offsets)
    dist.barrier()
    random.seed(7)
    again = hostio.load_corpus_for_rank(root / "ragged.txt", str(root / "partial.txt"), 1, rank)    # now it exists: all reuse it
    np.savez(root / f"ragged_again{rank}.npz", data=again.data, off=again.offsets)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_a_corpus_without_final_newline(tmp_path):
    import torch.multiprocessing as mp
    lines = [f"line {i} of the source\n".encode() for i in range(40)]
    lines[-1] = b"the last line has no newline"
    (tmp_path / "ragged.txt").write_bytes(b"".join(lines))
    mp.spawn(_ragged_corpus_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "ragged0.npz"), np.load(tmp_path / "ragged1.npz")
    assert np.array_equal(a["data"], b["data"]) and np.array_equal(a["off"], b["off"]) and len(a["off"]) == 41
    pos = bytes(a["data"]).index(b"the last line has no newline")
    assert pos + 28 < len(a["data"])          # the shuffle did move it off the end (else the test shows nothing)
    a2, b2 = np.load(tmp_path / "ragged_again0.npz"), np.load(tmp_path / "ragged_again1.npz")
    assert np.array_equal(a2["data"], b2["data"]) and np.array_equal(a2["off"], b2["off"])
    assert len(a2["off"]) == 40               # the reference's reload quirk (two lines merged), the same on every rank
