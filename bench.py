#!/usr/bin/env python3
"""bench.py -- one ablation step of SaGe 2.0 on MI355X (BASELINE.json metric / configs[1]).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (greedy tokenization + ablation counts + f64 scoring, i.e. the device work of
one run_sage_parallel call, utils.py:141-257) over the rank's resident corpus shard.  Workload at N=1: the 1M-line
synthetic English-Wikipedia-shaped corpus with a 262,144-key vocabulary (BASELINE.json configs[1]); at N>1 every
rank holds its own 1M-line shard (weak scaling) and the per-step result vectors (loss f64[V], occurrences
i64[V], histogram, totals) are combined over RCCL by the product's own `sage_amd.dist.step_into`.  Corpus,
vocabulary and embeddings are uploaded before the timed region; the timed region is K steps between barrier +
synchronize pairs; rank 0 prints ONE JSON line.

Besides `value` the line carries
  run_sage_parallel_ms  what SURVEY.md 8d names: one whole `sage_amd.parallel.run_sage_parallel` call right after a
                        `set_vocabulary` (vocabulary re-upload + step + result dicts), N=1 only;
  roofline              the dominant kernel: algorithmic bytes per launch (formulas below and in DESIGN.md) over its
                        average launch time from HIP events on the library's stream;
  step_roofline         the whole step against SURVEY.md 8d's bytes per token, with e = 4 (what is stored) and e = 8;
  cpu_baseline          the oracle port on this host's cores, on a bounded sample of the same corpus.
"""
from __future__ import annotations

import argparse
import json
import re
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# kernel class (sage_hip.h SAGE_K_*) -> the kernel that dominates it in the rocprofv3 traces under profiles/
CLASS_KERNEL = {"match": "k_match_lds", "score": "k_dots2<", "fix": "k_dots2_mixed", "count": "k_key_dedupe", "partition": "k_key_partition3",
                "describe": "k_describe<false>", "pairs": "k_pair_cache", "logsig": "k_logsig", "walk": "k_walk_write"}
# what limits each kernel class (Kernels), reported as `limiter`: only the "hbm" ones stream HBM.  `bound` is
# "hbm" for every kernel (nothing here is MFMA work): the HBM peak is what every fraction is taken against
CLASS_BOUND = {"match": "l2-miss-traffic", "score": "l2-fabric-gather", "pairs": "l2-fabric-gather", "fix": "l2-fabric-gather",
               "count": "hbm", "partition": "hbm", "describe": "lds-latency", "logsig": "hbm", "walk": "latency"}


def workload_tag(args):
    """Names the workload in the file names under profiles/ (a PMC pass belongs to ONE workload)."""
    lines = f"{args.lines // 1_000_000}m" if args.lines % 1_000_000 == 0 else f"{args.lines // 1000}k"
    return f"{args.style}{args.max_len}_{lines}_d{args.dim}_v{args.vocab // 1024}k"


def pmc_traffic(kernel_class, args):
    """HBM-side bytes per launch of a kernel from the newest committed PMC passes OF THIS WORKLOAD (profiles/collect.sh
    takes FETCH_SIZE and WRITE_SIZE in runs of their own and names its outputs after `workload_tag`): raw, with
    FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 counts 64 B per 128-B request) and `calibrated` = what
    profiles/r02_pmc_calibration.json measured for this library's access shapes (the same factor 2.00).  None when no
    pass of this workload is committed."""
    def version_key(path):   # r02_v3 after r01_v15: numbers compared as numbers
        return [int(x) for x in re.findall(r"\d+", path.name.split("_pmc_")[0].replace(workload_tag(args), ""))]
    files = sorted((REPO / "profiles").glob(f"r*_{workload_tag(args)}_pmc_traffic.json"), key=version_key)
    if not files and (args.lines, args.vocab, args.dim, args.max_len, args.style) == (1_000_000, 262_144, 50, 16, "en"):
        files = sorted((p for p in (REPO / "profiles").glob("r*_v*_pmc_traffic.json") if re.fullmatch(r"r\d+_v\d+_pmc_traffic\.json", p.name)),
                       key=lambda p: [int(x) for x in re.findall(r"\d+", p.name)])       # rounds 1-2 named the default workload's passes r0N_vM
    if not files:
        return None
    want = "__step_total__" if kernel_class == "__step_total__" else CLASS_KERNEL.get(kernel_class)
    for row in json.loads(files[-1].read_text()):
        if want and row["kernel"].startswith(want):
            out = {"source": files[-1].name, "raw": (row["fetch_GB_per_launch"] + row["write_GB_per_launch"]) * 1e9,
                   "fetch_x2": (row["fetch_x2_GB_per_launch"] + row["write_GB_per_launch"]) * 1e9}
            if "calibrated_GB_per_launch" in row:
                out["calibrated"] = row["calibrated_GB_per_launch"] * 1e9
            return out
    return None


def roofline_object(dom, avg_ms, launches_per_step, per_kernel_bytes, named_bytes, eb, args):
    """The contract's roofline object for the kernel with the largest share of the step: algorithmic bytes per launch over
    the average launch time measured live (HIP events on the library's stream), against HBM's 8 TB/s.  Besides it the
    MEASURED L2-miss-side rate (PMC bytes of this workload's committed pass over the same launch time)."""
    traffic = pmc_traffic(dom, args)
    alg = per_kernel_bytes[dom] / launches_per_step
    achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    out = {"bound": "hbm", "limiter": CLASS_BOUND.get(dom, "hbm"), "kernel": dom, "kernel_name": CLASS_KERNEL.get(dom, dom),
           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
           "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step,
           "algorithmic_bytes_per_launch": alg, "elem_bytes": eb,
           "note": "peak is HBM's 8 TB/s for every kernel.  `frac` = algorithmic bytes / launch time / peak; for the gather "
                   "kernels (score, fix) the algorithmic bytes count an embedding row ONCE per 32 consecutive occurrences "
                   "(unique_rows_per_32_occurrences) plus 64 B of descriptor per occurrence, so cache hits on repeated rows "
                   "cannot push it past 1.  `named_rows_GBs` is the rate at which rows reach the lanes when every window "
                   "row is counted every time (a delivery rate, not a fraction of HBM: L1/L2 serve the repeats).  "
                   "`traffic_frac` = measured L2-miss-side bytes (FETCH_SIZE x 2 + WRITE_SIZE) over the same time"}
    if traffic and avg_ms:
        out["traffic_bytes_per_launch"] = traffic["fetch_x2"]      # FETCH_SIZE x 2 + WRITE_SIZE of the committed PMC passes
        out["traffic_GBs"] = traffic["fetch_x2"] / 1e9 / (avg_ms * 1e-3)
        out["traffic_frac"] = out["traffic_GBs"] / HBM_PEAK_GBS
    if dom in named_bytes and avg_ms:
        out["compulsory_bytes_per_launch"], out["frac_compulsory"] = alg, out["frac"]     # (the same figure under round 2's names)
        out["named_rows_bytes_per_launch"] = named_bytes[dom] / launches_per_step
        out["named_rows_GBs"] = named_bytes[dom] / launches_per_step / (avg_ms * 1e-3) / 1e9
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--lines", type=int, default=1_000_000, help="lines per GPU")
    ap.add_argument("--vocab", type=int, default=262_144)
    ap.add_argument("--dim", type=int, default=50)
    ap.add_argument("--max-len", type=int, default=16)
    ap.add_argument("--workers", type=int, default=4, help="reference workers_number (defines count segments; run.sh uses 4)")
    ap.add_argument("--style", default="en")
    ap.add_argument("--cpu-lines", type=int, default=0, help="lines of the CPU-baseline sample (0 = auto, <0 = skip)")
    ap.add_argument("--batch-mb", type=int, default=0)
    ap.add_argument("--score-mode", type=int, default=0, help="0 = float32 row storage where lossless, 1 = float64 storage")
    ap.add_argument("--match-mode", type=int, default=-1, help="matcher variant (development switch)")
    ap.add_argument("--no-api-timing", action="store_true", help="skip the run_sage_parallel_ms and step_after_prune measurements")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="N>1: weak = every rank holds its own --lines shard; strong = ONE --lines corpus cut over the ranks by "
                         "the product's own sharding (partition_segments + sync_corpus_shard)")
    return ap.parse_args()


def cpu_baseline(corpus, offsets, vocab, max_len, emb, n_lines_req):
    """The oracle (a dict-based port of the reference algorithm, like the reference one process per shard) timed on
    this host: kind "port".  Bounded sample of the same corpus so the default run stays within minutes."""
    from oracle import sage_oracle as orc
    from sage_amd import synth
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1))
    cores = max(1, min(cores, 32))
    n = n_lines_req if n_lines_req > 0 else cores * 2500   # ~15 s on 16 cores
    n = min(n, len(offsets) - 1)
    sub_off = offsets[:n + 1]
    lines = synth.corpus_lines(corpus[:sub_off[-1]], sub_off)
    tok = orc.OracleTokenizer(vocab, max_len)
    t0 = time.time()
    tt, ttr, losses, sizes = orc.run_parallel(emb, lines, tok, cores, processes=cores)
    dt = time.time() - t0
    return {"value": tt / dt, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"first {n} lines ({tt} tokens) of the same corpus, oracle/sage_oracle.py run_parallel with "
                      f"{cores} worker processes (pool start + pickling of vocab/embeddings included, as in the reference), {dt:.1f} s"}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def algorithmic_bytes(st, dim, uniq=None):
    """Per step and rank, per kernel class: the bytes the algorithm has to move (Kernels, states each).
    e = bytes per stored embedding element on the main pass (4: the lossless float32 copy).  With float32 row storage the
    occurrences are split: the score kernel runs on the EXACT list only (its window rows, its descriptors), the mixed
    float64 kernel ("fix") on the rest, gathering marked rows from the float64 matrix and the others from the copy.
    For the two gather kernels the figure returned here counts every row EVERY time a window names it ("named rows");
    `compulsory_bytes` counts a row once per 32 consecutive occurrences and is what their roofline fractions use."""
    e = st["emb_elem_bytes"]
    tok, occ, nb = st["tokens"], st["occurrences"], st["corpus_bytes"]
    occ_fix, rows_fix, rows_f64 = st["f64_fix_occurrences"], st["window_rows_inexact"], st["f64_rows_gathered"]
    if e == 8:        # float64 storage: one score kernel over everything
        occ_fix = rows_fix = rows_f64 = 0
    return {
        "match": nb * (1 + 1 + 4),                              # text + rem read, 4-byte match entry written, per byte
        "walk": tok * 4 * 2 + tok * 12,                         # M entry of every token read by both passes; token arrays out
        "describe": st["rows"] * 4 + occ * 64,                  # 4 B of M per window/span element, 64-byte descriptor out
        "partition": occ * 64 + st["pairs"] * 8,                # descriptors in, one 8-byte record per position pair out
        "count": st["pairs"] * 8,                               # every record read once
        "pairs": tok * (dim * e + 16),                          # one row per token, 16-byte pair-cache entry out
        "score": (st["window_rows"] - rows_fix) * dim * e + (occ - occ_fix) * 64,   # rows of the exact list's windows + descriptors
        "logsig": st["window_pairs"] * 8 + occ * (16 + 32),     # dots read once; header + pair-cache window per occurrence
        "fix": (rows_fix - rows_f64) * dim * 4 + rows_f64 * dim * 8 + occ_fix * 64,
    }


def compulsory_bytes(st, uniq, dim):
    """The gather kernels' ALGORITHMIC bytes: a row counted ONCE per window of 32 consecutive occurrences -- neighbouring
    occurrences share their pad rows, so a kernel that holds 32 occurrences in flight (k_dots hands them out that way)
    need not move a row twice -- plus the descriptors (sage_measure_unique_rows over the last batch, scaled to the step
    by occurrences).  Never more than the named-row figure, so a fraction built on it cannot pass 1 through cache hits on
    repeated rows, which the named-row figure does at D = 300."""
    if not uniq or uniq["occurrences"] <= 0:
        return None
    scale = st["occurrences"] / uniq["occurrences"]
    e = st["emb_elem_bytes"]
    occ_fix = st["f64_fix_occurrences"] if e == 4 else 0
    out = {"score": uniq["exact_unique_rows"] * scale * dim * e + (st["occurrences"] - occ_fix) * 64}
    if occ_fix:
        share64 = st["f64_rows_gathered"] / max(1, st["window_rows_inexact"])
        out["fix"] = uniq["inexact_unique_rows"] * scale * dim * (4 + 4 * share64) + occ_fix * 64
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    # SAGE_BENCH_REHEARSE=1: run the N>1 code path on a ONE-GPU box (every rank on cuda:0, gloo, collectives staged
    # through host memory by sage_amd.dist).  For checking the multi-rank flow only; its numbers mean nothing.
    rehearse = os.environ.get("SAGE_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from sage_amd import SaGeTokenizer, parallel, synth
    from sage_amd.dist import StepBuffers, _through_host, step_into
    from sage_amd.parallel import PackedCorpus, segment_line_bounds

    def all_reduce_(t, op=None):
        _through_host(dist, t, lambda x: dist.all_reduce(x, op=op or dist.ReduceOp.SUM))

    seed = synth.DEFAULT_SEED
    t_gen = time.time()
    # the vocabulary comes from the head of shard 0 so that every rank builds the same one
    n_types = 1_000_000 if args.lines > 200_000 else 50_000
    head_lines = min(args.lines, 14_000)
    log(f"rank {rank}/{world}: generating vocabulary sample")
    head, head_off = synth.make_corpus(head_lines, seed=seed, style=args.style, n_types=n_types)
    vocab = synth.make_vocab(head, head_off, args.vocab, args.max_len, sample_bytes=6_000_000)
    if (rank == 0 or args.scaling == "strong") and args.lines == head_lines:
        corpus, offsets = head, head_off
    else:
        log(f"vocabulary of {len(vocab)} keys built; generating {args.lines} lines")
        corpus, offsets = synth.make_corpus(args.lines, seed=seed, line_seed=0 if args.scaling == "strong" else rank,
                                            style=args.style, n_types=n_types, progress=log)
    log(f"corpus: {len(corpus)} bytes")
    gen_s = time.time() - t_gen

    # the product's own objects: tokenizer state on the host, packed corpus, one engine per GPU
    model = SaGeTokenizer(vocab, args.max_len)
    packed = PackedCorpus(corpus, offsets)
    eng = parallel.get_engine(local_rank)
    if args.batch_mb:
        eng.set_batch_bytes(args.batch_mb << 20)
    if args.match_mode >= 0:
        eng.set_match_mode(args.match_mode)
    eng.set_score_mode(args.score_mode)
    t_up = time.time()
    parallel.sync_vocab(eng, model)
    bounds = segment_line_bounds(len(offsets) - 1, args.workers)
    strong = world > 1 and args.scaling == "strong"
    if strong:      # the product's sharding: whole count segments balanced by bytes, only this rank's lines uploaded
        bounds = parallel.sync_corpus_shard(eng, packed, bounds, rank, world)
        log(f"rank {rank}: {len(bounds) - 1} count segments, {eng.n_lines} lines resident")
    else:
        parallel.sync_corpus(eng, packed)
    # Embeddings as train_embeddings leaves them (embeddings.py:77-89): float32 word2vec vectors for the tokens the
    # tokenized corpus contains (word2vec is trained on that very tokenization, min_count=1), the float64 uniform
    # draw for the rest.  The corpus is tokenized here on the GPU; with several ranks the token counts are summed so
    # that every rank builds the same matrix.
    seen_counts = np.bincount(eng.tokenize(0, eng.n_lines)[0], minlength=len(vocab)).astype(np.int64)
    if world > 1:
        t = torch.from_numpy(seen_counts).cuda()
        all_reduce_(t)
        seen_counts = t.cpu().numpy()
    seen = seen_counts > 0
    emb = synth.make_embeddings(len(vocab), args.dim, seed=seed, unseen=~seen)
    log(f"embeddings: {int((~seen).sum())} of {len(vocab)} rows are float64 uniform draws (tokens the corpus never produces)")
    parallel.sync_embeddings(eng, emb)
    upload_s = time.time() - t_up
    log(f"uploaded in {upload_s:.1f} s")
    V = model.id_space()
    buf = StepBuffers(V, eng.hist_len)

    def step():   # the product's per-step path: sage_ablation_step_device into torch CUDA tensors + combine
        eng.invalidate_matches()      # a COLD step: the matcher runs, whatever match table an earlier step left resident
        step_into(eng, bounds, buf, dist if world > 1 else None)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
        log("warmup step done")
    eng.set_profiling(True)
    eng.reset_kernel_times()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    eng.set_profiling(False)
    log(f"{args.steps} timed steps: {dt:.3f} s")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        all_reduce_(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_tokens = int(buf.tot[0].item())      # whole job (already combined over ranks)
    total_triples = int(buf.tot[1].item())
    ms_per_step = dt / args.steps * 1e3
    st = eng.step_stats()                      # this rank's shard (taken before the API timing below runs more steps)
    kt = eng.kernel_times()
    uniq = eng.measure_unique_rows(32)         # over the descriptors the last timed step left resident
    loss_h, occ_h = buf.loss.cpu().numpy(), buf.occ.cpu().numpy()

    api_ms, api_parts = None, None
    if world == 1 and not args.no_api_timing:
        # SURVEY.md 8d's call: run_sage_parallel right after the prune loop's set_vocabulary (SaGeVocabBuilder.py:192),
        # i.e. vocabulary table rebuild + upload, the step, and the result dicts.  Corpus and embeddings stay resident.
        reps = []
        os.environ["SAGE_INCREMENTAL_MATCH"] = "0"    # a cold call: tables rebuilt, matcher run (the carried-over case is below)
        for _ in range(3):
            model.set_vocabulary(vocab)           # bumps vocab_version: the next call re-uploads the table
            torch.cuda.synchronize()
            ta = time.perf_counter()
            parallel.sync_vocab(eng, model)
            tb = time.perf_counter()
            tt_api, ttr_api, losses_api, sizes_api = parallel.run_sage_parallel(emb, packed, model, args.workers)
            tc = time.perf_counter()
            reps.append(((tc - ta) * 1e3, (tb - ta) * 1e3))
        os.environ.pop("SAGE_INCREMENTAL_MATCH", None)
        assert (tt_api, ttr_api) == (total_tokens, total_triples)
        api_ms, vocab_ms = sorted(reps)[1]
        api_parts = {"vocabulary_upload_ms": vocab_ms, "step_and_dicts_ms": api_ms - vocab_ms}
        log(f"run_sage_parallel after set_vocabulary: {reps} ms (total, of which vocabulary upload)")

    after_prune = None
    if world == 1 and not args.no_api_timing:
        # The step AFTER a prune (every step of a schedule but the first): SaGe's own cut on the losses just computed
        # (SaGeVocabBuilder.py:109-159: bytes protected, the lowest-loss ids go) down to 7/8 of the vocabulary, i.e. the
        # README's 262144 -> 229376.  The pruned vocabulary only lost keys, so the resident match table is moved
        # (sage_set_vocab_pruned) and the step launches no matcher; the same step with the table recomputed is timed
        # beside it.  Reported separately: `value` above is the cold step.
        scored = {int(i): float(loss_h[i]) for i in np.nonzero(occ_h)[0]}
        model.add_all_byte_ids(scored, score=1e6)
        # the README schedule scaled to this vocabulary; targets the active vocabulary already meets are skipped
        # (SaGeVocabBuilder.py:130-136) -- on the default corpus 208k of the 262,144 keys are active, so the cut is 196,608
        readme = [262144, 229376, 196608, 163840, 131072, 98304, 65536, 57344, 49152, 40960, 32768, 16384]
        target = next((t * len(vocab) // 262144 for t in readme[1:] if t * len(vocab) // 262144 < len(scored)), len(scored) - 1)
        ranked = sorted((v, k) for k, v in scored.items())
        doomed = {model.id_to_bytes(t) for _, t in ranked[:len(scored) - target]}
        survivors = [k for k in vocab if k not in doomed]
        # Two matrices, as a schedule meets them: STALE = the matrix of the unpruned vocabulary read with the re-compacted
        # ids, which is what the reference does between two retrain labels (SaGeVocabBuilder.py:89-91,192: ids move,
        # rows do not) -- other ids now meet the float64 uniform rows, most windows touch one and the float32 copy stops
        # paying; FRESH = what a retrain at this label leaves (embeddings.py:77-89): float32-valued rows for the tokens
        # the PRUNED vocabulary's tokenization of the corpus produces, the float64 uniform draw for the rest.
        model.set_vocabulary(survivors)
        parallel.sync_vocab(eng, model)
        seen_after = np.bincount(eng.tokenize(0, eng.n_lines)[0], minlength=len(survivors)) > 0
        emb_fresh = synth.make_embeddings(len(survivors), args.dim, seed=seed + 1, unseen=~seen_after)
        res = {}
        for matrix, E in (("stale_matrix", emb), ("fresh_matrix", emb_fresh)):
            res[matrix] = {}
            for mode, incr in (("carried_matches", "1"), ("rematched", "0")):
                model.set_vocabulary(vocab)
                parallel.sync_vocab(eng, model)
                eng.set_embeddings(emb)                       # (also resets the step's float32 / float64 storage choice)
                eng.invalidate_matches()
                step_into(eng, bounds, buf, None)             # the step before the prune (leaves a complete match table)
                os.environ["SAGE_INCREMENTAL_MATCH"] = incr
                model.set_vocabulary(survivors)
                eng.set_embeddings(E)
                torch.cuda.synchronize()
                ta = time.perf_counter()
                parallel.sync_vocab(eng, model)
                tb = time.perf_counter()
                eng.set_profiling(True)
                eng.reset_kernel_times()
                step_into(eng, bounds, buf, None)
                torch.cuda.synchronize()
                tc = time.perf_counter()
                eng.set_profiling(False)
                st2 = eng.step_stats()
                res[matrix][mode] = {
                    "vocabulary_ms": (tb - ta) * 1e3, "step_ms": (tc - tb) * 1e3, "total_ms": (tc - ta) * 1e3,
                    "no_matcher": st2["no_matcher"], "total_tokens": int(buf.tot[0].item()),
                    "total_triples": int(buf.tot[1].item()), "occurrences": st2["occurrences"],
                    "f64_fix_occurrences": st2["f64_fix_occurrences"], "emb_elem_bytes": st2["emb_elem_bytes"],
                    "kernel_ms": {k: round(v[0], 3) for k, v in eng.kernel_times().items() if v[0] > 0}}
            a, b = res[matrix]["carried_matches"], res[matrix]["rematched"]
            assert a["no_matcher"] == 1 and b["no_matcher"] == 0
            assert (a["total_tokens"], a["total_triples"]) == (b["total_tokens"], b["total_triples"])
        os.environ.pop("SAGE_INCREMENTAL_MATCH", None)
        eng.set_embeddings(emb)
        after_prune = {"active_before": len(scored), "target": target, "pruned_keys": len(vocab) - len(survivors),
                       "vocabulary_after": len(survivors), **res}
        model.set_vocabulary(vocab)
        log(f"step after a prune: {res}")

    if rank == 0:
        # ---- algorithmic bytes, statistics measured on this very corpus ----
        tokens_local = st["tokens"]
        occ = max(1, st["occurrences"])
        b_bar = st["corpus_bytes"] / tokens_local
        f_m = st["occurrences"] / tokens_local
        r_bar = st["rows"] / occ
        p_bar = st["records"] / occ
        eb = st["emb_elem_bytes"]   # 4 when the rows are gathered from the lossless float32 copy, else 8

        def A_of(e):
            return b_bar + 9 + f_m * (r_bar * args.dim * e + 16 * p_bar)
        named_all = algorithmic_bytes(st, args.dim)
        comp = compulsory_bytes(st, uniq, args.dim) or {}
        per_kernel_bytes = {**named_all, **comp}          # gather kernels: rows once per 32 occurrences (see compulsory_bytes)
        named_bytes = {k: named_all[k] for k in comp}
        dom = max(per_kernel_bytes, key=lambda k: kt[k][0])
        dom_ms, dom_n = kt[dom]
        launches_per_step = max(1, dom_n // args.steps)
        avg_ms = dom_ms / max(1, dom_n)
        step_s = dt / args.steps
        step_alg_bytes = sum(per_kernel_bytes.values())   # this rank's shard

        def step_gbs(e):
            return A_of(e) * total_tokens / step_s / 1e9 / world
        out = {
            "metric": "corpus tokens scored/sec per ablation step (node) @262k vocab + HBM GB/s",
            "value": total_tokens * args.steps / dt,
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "run_sage_parallel_ms": api_ms,
            "run_sage_parallel_parts": api_parts,
            "step_after_prune_ms": after_prune["stale_matrix"]["carried_matches"]["total_ms"] if after_prune else None,
            "step_after_prune": after_prune,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"embeddings": f"float32 word2vec-shaped rows upcast to float64, {int((~seen).sum())} float64 uniform rows (embeddings.py:77-89)",
                       "workload": f"{args.lines}-line synthetic {args.style}-wiki-shaped corpus {'cut over the GPUs' if strong else 'per GPU'}, single "
                                   f"{args.vocab}-key vocabulary ablation step (tokenize + score + counts), D={args.dim}, "
                                   f"max_len={args.max_len}, pad=2, workers_number={args.workers}",
                       "workload_shape": "substring-frequency vocabulary: ablations rejoin the base tokenization late, so this "
                                         "corpus is harder than the survey's measured text (b 5.3 / f_m 0.93 / r 7.3 / p 24.4); "
                                         "its own statistics are in step_roofline",
                       "lines_per_gpu": int(eng.n_lines), "workload_tag": workload_tag(args), "vocab": len(vocab), "dim": args.dim, "max_len": args.max_len,
                       "corpus_bytes_per_gpu": int(st["corpus_bytes"]), "total_tokens": total_tokens,
                       "total_triples": total_triples},
            "roofline": roofline_object(dom, avg_ms, launches_per_step, per_kernel_bytes, named_bytes, eb, args),
            # the same object for the largest kernel that memory traffic does limit (when the largest of all is the matcher,
            # whose algorithmic bytes say nothing about it: its `limiter` is the request rate of random 64-byte probes)
            "roofline_largest_memory_bound_kernel": (lambda k2: roofline_object(
                k2, kt[k2][0] / max(1, kt[k2][1]), max(1, kt[k2][1] // args.steps), per_kernel_bytes, named_bytes, eb, args))(
                max((k for k in per_kernel_bytes if CLASS_BOUND.get(k) in ("hbm", "l2-fabric-gather")), key=lambda k: kt[k][0])),
            "unique_rows_per_32_occurrences": uniq,
            "step_roofline": {"measured_hbm_bytes_per_step": pmc_traffic("__step_total__", args) if world == 1 else None,
                              # the step's algorithmic bytes = the sum of its kernels' (kernel_roofline), per token of the shard
                              "bytes_per_token_A": step_alg_bytes / tokens_local,
                              "achieved_GBs_per_gpu": step_alg_bytes / step_s / 1e9,
                              "frac": step_alg_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                              # SURVEY.md 8d's formula A = b + 9 + f_m (r D e + 16 p) names every row of every window and
                              # base span at D*e bytes: delivery rates (GB/s), not fractions -- at D = 300 they pass HBM's peak
                              "survey_model": {"bytes_per_token_e4": A_of(4), "bytes_per_token_e8": A_of(8),
                                               "GBs_e4": step_gbs(4), "GBs_e8": step_gbs(8)},
                              "b_bar": b_bar, "f_m": f_m, "r_bar": r_bar, "p_bar": p_bar, "emb_elem_bytes": eb,
                              "window_rows_per_occurrence": st["window_rows"] / occ,
                              "window_pairs_per_occurrence": st["window_pairs"] / occ,
                              "long_occurrence_share": st["long_occurrences"] / occ,
                              "f64_fix_occurrences": st["f64_fix_occurrences"]},
            "kernel_ms_per_step": {k: kt[k][0] / args.steps for k in kt},
            "timing_note": "the timed region is K cold steps (sage_invalidate_matches before each: the matcher runs every step) "
                           "between barrier + synchronize pairs; the per-kernel HIP events (sage_set_profiling) are recorded "
                           "inside it, on the library's stream",
            # every kernel class against the HBM roofline: algorithmic bytes per step / its time per step
            "kernel_roofline": {k: {"algorithmic_GB_per_step": per_kernel_bytes[k] / 1e9, "bound": "hbm", "limiter": CLASS_BOUND.get(k),
                                    "GBps": per_kernel_bytes[k] / (kt[k][0] / args.steps * 1e-3) / 1e9 if kt[k][0] > 0 else 0.0,
                                    "frac": per_kernel_bytes[k] / (kt[k][0] / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS if kt[k][0] > 0 else 0.0}
                                for k in per_kernel_bytes},
            "kernel_named_rows_GBps": {k: v / (kt[k][0] / args.steps * 1e-3) / 1e9 for k, v in named_bytes.items() if kt[k][0] > 0},
            # the same kernels by MEASURED L2-miss-side bytes (committed PMC pass) over the live kernel time
            "kernel_traffic_GBps": {k: (lambda t: t["fetch_x2"] * (kt[k][1] / args.steps) / (kt[k][0] / args.steps * 1e-3) / 1e9
                                        if t and kt[k][0] > 0 else None)(pmc_traffic(k, args)) for k in per_kernel_bytes if k not in ("walk", "fix")},
            "setup_s": {"generate": gen_s, "upload": upload_s},
        }
        if args.cpu_lines >= 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(corpus, offsets, vocab, args.max_len, emb, args.cpu_lines)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
