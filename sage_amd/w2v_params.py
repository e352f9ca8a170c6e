"""word2vec hyper-parameters handed to gensim (reference: Word2VecParams.py:3-10).  Only D reaches the GPU path
(row width of the embedding matrix); window/negatives/alpha affect gensim alone."""
from dataclasses import dataclass


@dataclass
class Word2VecParams:
    D: int = 50
    N: int = 15
    ALPHA: float = 0.025
    window_size: int = 5
    min_count: int = 1
    sg: int = 1
