"""Path constants (reference: python3/src/pushworld/config.py:20-33, hot-path subset)."""
import os

MODULE_PATH = os.path.split(os.path.abspath(__file__))[0]

PUZZLE_EXTENSION = ".pwp"

# The benchmark puzzles are vendored as data next to the package (levels 1-4 extracted,
# level 0 as the original zip), so `standard_padding` works without the reference checkout.
BENCHMARK_PATH = os.path.join(MODULE_PATH, "data")
BENCHMARK_PUZZLES_PATH = os.path.join(BENCHMARK_PATH, "puzzles")
BENCHMARK_SOLUTIONS_PATH = os.path.join(BENCHMARK_PATH, "solutions")
