"""SaGeVocabBuilder: the reference's schedule loop (SaGeVocabBuilder.py:14-200) around the GPU ablation step.

Same constructor, same `build_vocab` arguments, same files under results/<exp>/ -- including the semantics that
make pruned vocabularies identical: ids are re-compacted after every prune while the embedding matrix is only
replaced on embeddings_schedule steps (stale rows are indexed by the new ids), single bytes are protected with a
score of 1e6, targets the active vocabulary already satisfies are skipped, and the cut is the first
`active - target` entries of sorted((loss, id)).

Under torch.distributed (one process per GPU) every rank runs this loop on the combined, bit-identical loss vector
and therefore takes the same prune decisions; rank 0 alone writes files and trains embeddings, which are then
broadcast (`sage_amd.dist`)."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Dict, List, Optional, Tuple, Union

from . import dist as sdist
from . import hostio
from .embeddings import get_embeddings
from .parallel import run_sage_parallel
from .tokenizer import SaGeTokenizer
from .w2v_params import Word2VecParams

BYTE_SCORE = 1e6            # SaGeVocabBuilder.py:109
NEAR_TIE_RTOL = 1e-9        # |delta loss| <= this * |loss| across the cut is reported (float sums are order-dependent)


def near_ties_at_cut(ranked: List[Tuple[float, int]], n_prune: int, rtol: float = NEAR_TIE_RTOL):
    """Entries on both sides of the cut `ranked[:n_prune] | ranked[n_prune:]` whose losses lie within `rtol` of the
    boundary pair.  The reference's cut is sorted((loss, id))[:k] (SaGeVocabBuilder.py:146,159); its float sums are
    merged in task-completion order (utils.py:159-183) and ours with atomics, so a pair this close could change
    sides between runs or GPU counts.  Returns (pruned_side, kept_side) as lists of (loss, id); both empty when the
    boundary is clear."""
    if n_prune <= 0 or n_prune >= len(ranked):
        return [], []
    below, above = ranked[n_prune - 1][0], ranked[n_prune][0]
    band = rtol * max(abs(below), abs(above))
    if above - below > band:
        return [], []
    lo = n_prune - 1
    while lo > 0 and above - ranked[lo - 1][0] <= band:
        lo -= 1
    hi = n_prune
    while hi + 1 < len(ranked) and ranked[hi + 1][0] - below <= band:
        hi += 1
    return ranked[lo:n_prune], ranked[n_prune:hi + 1]


class SaGeVocabBuilder:

    def __init__(self, full_vocab_schedule: List[int], embeddings_schedule: List[int], max_len: int = 16,
                 workers_number: int = 1, random_seed: int = 692653, word2vec_d: int = 50, word2vec_n: int = 15,
                 word2vec_alpha: float = 0.025, word2vec_window_size: int = 5, word2vec_min_count: int = 1,
                 word2vec_sg: bool = True):
        self.full_vocab_schedule = full_vocab_schedule
        self.embeddings_schedule = embeddings_schedule
        self.max_len = max_len
        self.workers_number = workers_number
        self.random_seed = random_seed
        self.word2vec_params = Word2VecParams(D=word2vec_d, N=word2vec_n, ALPHA=word2vec_alpha,
                                              window_size=word2vec_window_size, min_count=word2vec_min_count,
                                              sg=int(word2vec_sg))
        # the ablation step; None = run_sage_parallel on one GPU, run_sage_distributed under torch.distributed
        self.run_step = None
        self.near_ties: Dict[int, tuple] = {}   # target size -> (pruned side, kept side) of a near-tie at the cut
        # fixed-point loss sums: bit-identical losses from run to run (sage_set_deterministic); SAGE_DETERMINISTIC=1 too
        self.deterministic = False

    # ---- one round: score, choose the cut, write the files -------------------------------------------------------
    def _step_fn(self):
        if self.run_step is not None:
            return self.run_step
        return sdist.run_sage_distributed if sdist.get_dist() else run_sage_parallel

    def build_vocab(self, experiment_name: str, corpus_filepath: Union[str, Path], vocabulary_filepath: Union[str, Path],
                    partial_corpus_filepath: Optional[Union[str, Path]] = None, partial_corpus_line_number: int = 1000):
        corpus_filepath = Path(corpus_filepath)
        vocabulary_filepath = Path(vocabulary_filepath)
        # SaGeVocabBuilder.py:38: only a str survives here, a Path argument becomes None
        partial_corpus_filepath = Path(partial_corpus_filepath) if isinstance(partial_corpus_filepath, str) else None
        rank, world = sdist.world_info()
        writer = rank == 0

        hostio.init_logger(experiment_name, rank)
        logging.info(f"experiment {experiment_name}: rank {rank} of {world}, workers_number={self.workers_number} "
                     f"(defines the count segments; the work runs on the GPU)")
        embeddings_folder, stats_folder, vocab_folder = hostio.get_output_folder(experiment_name)
        hostio.set_random_seed(experiment_name, self.random_seed, write=writer)
        byte_vocab = hostio.load_vocab(vocabulary_filepath)
        logging.info(f"initial vocabulary {vocabulary_filepath.as_posix()}: {len(byte_vocab)} keys")
        longest = max(len(v) for v in byte_vocab)
        if self.max_len != longest:
            logging.warning(f"max_len={self.max_len} differs from the longest key ({longest} bytes)")
        sage_model = SaGeTokenizer(byte_vocab, self.max_len)
        if self.deterministic:
            from .parallel import get_engine
            get_engine().set_deterministic(True)
        partial_corpus = hostio.load_corpus_for_rank(corpus_filepath, partial_corpus_filepath,
                                                     partial_corpus_line_number, rank)

        schedule = self.full_vocab_schedule
        if not len(schedule) >= 2:
            raise Exception("Vocabulary schedule must contain more than 2 vocabulary sizes!")
        schedule.sort(reverse=True)  # in place, like the reference (mutates the caller's list)
        logging.info(f"schedule starts at {schedule[0]}; the vocabulary holds {sage_model.vocab_size()} keys")
        retrain_at = set(self.embeddings_schedule)
        # always loaded/trained for the first label, scheduled or not (SaGeVocabBuilder.py:73)
        embeddings = get_embeddings(schedule[0], embeddings_folder, partial_corpus, sage_model, self.workers_number,
                                    self.word2vec_params)
        i = 0
        while i < len(schedule) - 1:
            label = schedule[i]
            target = schedule[i + 1]
            actual = sage_model.vocab_size()
            logging.info(f"round {i}: label {label}, target {target}, vocabulary now {actual}")
            if label in retrain_at:
                embeddings = get_embeddings(label, embeddings_folder, partial_corpus, sage_model,
                                            self.workers_number, self.word2vec_params)
            if actual <= target:
                logging.info("nothing to prune for this target; next round")
                i += 1
                continue

            total_tokens, total_triples, token_to_losses, ablated_sizes = self._step_fn()(
                embeddings, partial_corpus, sage_model, self.workers_number)
            logging.info(f"ablation step done: {total_tokens} tokens, {total_triples} distinct triples")

            scored = len(token_to_losses)
            sage_model.add_all_byte_ids(token_to_losses, score=BYTE_SCORE)
            active = len(token_to_losses)
            inactive = actual - active
            values = list(token_to_losses.values())
            neg_loss = sum(1 for v in values if v < 0.0)
            zero_loss = sum(1 for v in values if v == 0.0)
            pos_loss = sum(1 for v in values if v > 0.0)
            logging.info(f"{scored} scored ids + single bytes = {active} active, {inactive} inactive; "
                         f"losses: {neg_loss} negative / {zero_loss} zero / {pos_loss} positive")

            while active <= target:  # may run off the end of the schedule with IndexError, as the reference does
                i += 1
                logging.info(f"active vocabulary ({active}) already within target {target}: moving to round {i}")
                target = schedule[i + 1]
            n_prune = active - target
            ranked = sorted((loss, tid) for tid, loss in token_to_losses.items())  # ties: ascending id
            logging.info(f"pruning {n_prune} of {active} down to {target}")
            tie_low, tie_high = near_ties_at_cut(ranked, n_prune)
            if tie_low:
                self.near_ties[target] = (tie_low, tie_high)
                show = lambda side: ", ".join(f"{sage_model.id_to_encoded(t)}={v!r}" for v, t in side[:8])
                logging.warning(f"near-tie at the cut for target {target}: {len(tie_low)} pruned [{show(tie_low)}] and "
                                f"{len(tie_high)} kept [{show(tie_high)}] entries lie within {NEAR_TIE_RTOL:g} relative of "
                                f"each other; float summation order could move them across the cut")

            doomed = {sage_model.id_to_bytes(tid) for _, tid in ranked[:n_prune]}
            assert not any(len(tok) == 1 for tok in doomed)
            vocab_now = sage_model.get_vocabulary()
            survivors = {tok: tid for tok, tid in vocab_now.items() if tok not in doomed}
            if writer:
                hostio.save_sorted_losses(sage_model, ranked, target, vocab_folder)
                hostio.save_stats({
                    "current_step_vocab_size": label, "total_tokens": total_tokens, "total_triples": total_triples,
                    "current_active_vocab_size": active, "current_inactive_vocab_size": inactive, "neg_loss": neg_loss,
                    "zero_loss": zero_loss, "pos_loss": pos_loss, "target_vocab_size": target,
                    "num_tokens_to_prune": n_prune, "ablated_sizes": ablated_sizes,
                }, stats_folder, target)
                active_vocab = {tok: tid for tok, tid in survivors.items() if tid in token_to_losses}
                deleted = {tok: tid for tok, tid in vocab_now.items() if tok in doomed}
                for stem, table in (("sage_vocab", survivors), ("active_vocab", active_vocab), ("deleted_vocab", deleted)):
                    path = vocab_folder / f"{stem}_{target}.vocab"
                    logging.info(f"{path.as_posix()}: {len(table)} keys")
                    hostio.write_vocab(table, path)
            # ids are re-compacted here; `embeddings` is NOT remapped (stale rows, SaGeVocabBuilder.py:89-91,192)
            sage_model.set_vocabulary(list(survivors.keys()))
            logging.info(f"round {i} finished: {len(survivors)} keys remain ({target} of them active)")
            i += 1
        sdist.barrier()
