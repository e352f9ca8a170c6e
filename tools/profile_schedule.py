"""cProfile of build_vocab on the README-schedule golden inputs (where does the host time of a whole schedule go?)."""
import cProfile, pstats, sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests import test_gpu_readme_schedule as T

g, vocab, corpus = T._inputs()
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    root = Path(tmp)
    T._prepare(root, vocab, corpus)
    pr = cProfile.Profile()
    pr.enable()
    T._build(root, g, False)
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
