"""GPU: the multi-GPU PRODUCT path -- `sage_amd.dist.run_sage_distributed`, i.e. per-rank shard upload,
`HipSage.ablation_step_device` into torch CUDA tensors and the combine over the process group -- on two fresh ranks
that share cuda:0 (one-GPU rehearsal: gloo group, collectives staged through the host; under "nccl" the same code
path hands the CUDA tensors to RCCL).  Every rank must return the reference's W=4 result for the 20,600-line golden
(tests/golden/parallel.npz, captured from utils.py:141-190): integers exact, losses within 1e-5 relative."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.golden_io import load_npz, unpack

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
WORKERS = 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _golden_inputs():
    from sage_amd import synth
    g = load_npz("parallel.npz")
    vocab = unpack(g["vocab_bytes"], g["vocab_off"])
    emb = synth.make_embeddings(len(vocab), int(g["emb_dim"]), seed=int(g["emb_seed"]))
    return g, vocab, emb


def _rank_main(rank, world, port, out_dir, packed):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sage_amd import SaGeTokenizer, synth
    from sage_amd.dist import run_sage_distributed
    from sage_amd.parallel import PackedCorpus, get_engine
    g, vocab, emb = _golden_inputs()
    model = SaGeTokenizer(vocab, int(g["max_len"]))
    corpus = PackedCorpus(g["line_bytes"], g["line_off"]) if packed else synth.corpus_lines(g["line_bytes"], g["line_off"])
    tt, ttr, losses, sizes = run_sage_distributed(emb, corpus, model, WORKERS)
    # a second call reuses the resident shard (nothing is uploaded again) and must give the same integers
    tt2, ttr2, losses2, sizes2 = run_sage_distributed(emb, corpus, model, WORKERS)
    assert (tt2, ttr2, sizes2, sorted(losses2)) == (tt, ttr, sizes, sorted(losses))
    resident = get_engine().n_lines
    ids = np.array(sorted(losses), dtype=np.int64)
    np.savez(Path(out_dir) / f"rank{rank}.npz", tt=tt, ttr=ttr, ids=ids, vals=np.array([losses[int(i)] for i in ids]),
             size_keys=np.array(sorted(sizes)), size_vals=np.array([sizes[k] for k in sorted(sizes)]), resident=resident)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("packed", [False, True])
def test_two_ranks_run_the_hip_step_and_match_the_reference(tmp_path, packed):
    import torch.multiprocessing as mp
    g, vocab, emb = _golden_inputs()
    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path), packed), nprocs=2, join=True)
    W = WORKERS
    n_lines = len(g["line_off"]) - 1
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for r in (r0, r1):
        assert int(r["tt"]) == int(g[f"w{W}_total_tokens"])
        assert int(r["ttr"]) == int(g[f"w{W}_total_triples"])
        assert r["ids"].tolist() == g[f"w{W}_loss_ids"].tolist()
        want = g[f"w{W}_loss_vals"]
        np.testing.assert_allclose(r["vals"], want, rtol=1e-5, atol=1e-9 * max(1.0, float(np.abs(want).max())))
        assert r["size_keys"].tolist() == g[f"w{W}_size_keys"].tolist()
        assert r["size_vals"].tolist() == g[f"w{W}_size_vals"].tolist()
    assert r0["vals"].tolist() == r1["vals"].tolist()            # bit-identical combined vector on every rank
    # each rank uploaded only its own count segments
    assert 0 < int(r0["resident"]) < n_lines and int(r0["resident"]) + int(r1["resident"]) == n_lines


def _rank_fail_main(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sage_amd import SaGeTokenizer
    from sage_amd.dist import run_sage_distributed
    from sage_amd.parallel import PackedCorpus
    g, vocab, emb = _golden_inputs()
    model = SaGeTokenizer(vocab, int(g["max_len"]))
    corpus = PackedCorpus(g["line_bytes"], g["line_off"])
    outcome = "no exception"
    try:
        run_sage_distributed(emb[:300] if rank == 1 else emb, corpus, model, WORKERS)     # rank 1: matrix too short
    except IndexError as exc:
        outcome = f"IndexError: {exc}"
    (Path(out_dir) / f"outcome{rank}.txt").write_text(outcome)
    tt, ttr, _, _ = run_sage_distributed(emb, corpus, model, WORKERS)      # the group still works afterwards
    (Path(out_dir) / f"after{rank}.txt").write_text(f"{tt} {ttr}")
    dist.barrier()
    dist.destroy_process_group()


def test_a_rank_whose_step_fails_fails_every_rank(tmp_path):
    """utils.py:167 (`res.get()` re-raises a worker's exception in the parent): rank 1's embedding matrix is too short,
    its HIP step ends in SAGE_E_ROW (numpy's IndexError in the reference, utils.py:133); rank 0 must raise too instead
    of waiting in the combine, and the next step must work."""
    import torch.multiprocessing as mp
    g, _, _ = _golden_inputs()
    mp.spawn(_rank_fail_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    o0, o1 = (tmp_path / "outcome0.txt").read_text(), (tmp_path / "outcome1.txt").read_text()
    assert o1.startswith("IndexError: a token id is out of bounds")
    assert o0.startswith("IndexError: the ablation step failed on rank 1 (IndexError)")
    want = f"{int(g[f'w{WORKERS}_total_tokens'])} {int(g[f'w{WORKERS}_total_triples'])}"
    assert (tmp_path / "after0.txt").read_text() == want == (tmp_path / "after1.txt").read_text()


def _rccl_rank_main(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SAGE_DIST_SINGLE_RANK="1")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from sage_amd import SaGeTokenizer
    from sage_amd import dist as sdist
    from sage_amd.parallel import PackedCorpus, get_engine, segment_line_bounds, sync_corpus_shard, sync_embeddings, sync_vocab
    g, vocab, emb = _golden_inputs()
    assert sdist.device_collectives(dist)                    # backend "nccl" = RCCL: CUDA tensors go straight into the collectives
    model = SaGeTokenizer(vocab, int(g["max_len"]))
    corpus = PackedCorpus(g["line_bytes"], g["line_off"])
    eng = get_engine()
    sync_vocab(eng, model); sync_embeddings(eng, emb)
    local = sync_corpus_shard(eng, corpus, segment_line_bounds(len(corpus), WORKERS), rank, world)
    buf = sdist.StepBuffers(model.id_space(), eng.hist_len)
    sdist.step_into(eng, local, buf, dist)                   # the step, then combine() over RCCL on the device tensors
    torch.cuda.synchronize()
    got = sdist.broadcast_array(emb[:7] if rank == 0 else None, src=0)     # the embeddings broadcast, through RCCL too
    assert np.array_equal(got, emb[:7])
    sdist.assert_same_on_all_ranks((len(corpus), 5), "test tuple")
    occ = buf.occ.cpu().numpy()
    ids = np.nonzero(occ)[0]
    np.savez(Path(out_dir) / "rccl.npz", tt=int(buf.tot[0]), ttr=int(buf.tot[1]), ids=ids, vals=buf.loss.cpu().numpy()[ids])
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_collectives_on_device_tensors(tmp_path):
    """The RCCL branch of the combine (`device_collectives`: backend "nccl", CUDA tensors handed to all_reduce /
    all_gather_into_tensor / broadcast without host staging).  A one-GPU box cannot hold two RCCL ranks, so the group has
    ONE rank: every collective of the product path still runs through RCCL on device memory, and the result must be the
    reference's W=4 golden."""
    import torch.multiprocessing as mp
    g, _, _ = _golden_inputs()
    mp.spawn(_rccl_rank_main, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl.npz")
    assert int(r["tt"]) == int(g[f"w{WORKERS}_total_tokens"]) and int(r["ttr"]) == int(g[f"w{WORKERS}_total_triples"])
    assert r["ids"].tolist() == g[f"w{WORKERS}_loss_ids"].tolist()
    want = g[f"w{WORKERS}_loss_vals"]
    np.testing.assert_allclose(r["vals"], want, rtol=1e-5, atol=1e-9 * max(1.0, float(np.abs(want).max())))


def test_distributed_entry_point_without_a_process_group():
    """run_sage_distributed on one rank (no process group) is run_sage_parallel."""
    from sage_amd import SaGeTokenizer, synth
    from sage_amd.dist import run_sage_distributed
    from sage_amd.parallel import run_sage_parallel
    corpus, off = synth.make_corpus(300, seed=5, n_types=900, mean_log_words=2.4)
    vocab = synth.make_vocab(corpus, off, 1000, 16, sample_bytes=80_000)
    model = SaGeTokenizer(vocab, 16)
    lines = synth.corpus_lines(corpus, off)
    emb = synth.make_embeddings(len(vocab), 50, seed=9)
    a = run_sage_distributed(emb, lines, model, 3)
    b = run_sage_parallel(emb, lines, model, 3)
    assert a[0] == b[0] and a[1] == b[1] and a[3] == b[3] and sorted(a[2]) == sorted(b[2])
    np.testing.assert_allclose([a[2][k] for k in sorted(a[2])], [b[2][k] for k in sorted(b[2])], rtol=1e-9, atol=1e-12)
