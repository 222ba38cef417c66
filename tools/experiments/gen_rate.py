import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pushworld_amd.generate import generate_level0_puzzles
with tempfile.TemporaryDirectory() as d:
    t0 = time.perf_counter()
    generate_level0_puzzles(d, num_puzzles=300, random_seed=0, filter_puzzles=False)
    t1 = time.perf_counter()
    from pushworld_amd.generate import filter_puzzles_by_solvability
    kept = filter_puzzles_by_solvability(d, 2, 300)
    t2 = time.perf_counter()
    print(f"generate 300: {t1 - t0:.2f} s; filter: {t2 - t1:.2f} s ({300 / (t2 - t1):.0f} puzzles/s), kept {kept}")
