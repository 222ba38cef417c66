"""Puzzle file discovery (reference: python3/src/pushworld/utils/filesystem.py:21-61).

The pool order of ``PushWorldEnv`` is the order in which this generator yields files, and
``reset`` draws from that pool with ``random.Random`` (gym_env.py:107-109,171-174), so the
``os.walk`` traversal (unsorted, like the reference) is part of the observable behaviour
(SURVEY trap T15).
"""
import os
from typing import Generator, Optional, Tuple


def iter_files_with_extension(file_or_directory_path: str, extension: str) -> Generator[str, None, None]:
    extension = extension.lower()
    root = file_or_directory_path.rstrip(os.path.sep)
    if os.path.isfile(root):
        if not root.lower().endswith(extension):
            raise ValueError(f"The given file does not have the expected extension ({extension}): {root}")
        yield root
        return
    for parent, _, filenames in os.walk(root):
        for filename in filenames:
            if filename.lower().endswith(extension):
                yield os.path.join(parent, filename)


def map_files_with_extension(input_file_or_directory_path: str, input_extension: str, output_directory_path: str,
                             output_extension: Optional[str] = None) -> Generator[Tuple[str, str], None, None]:
    """(input path, output path) for every file with ``input_extension`` below the input: the output path mirrors the
    input's sub-directory structure under ``output_directory_path`` (created as needed) with the extension replaced
    by ``output_extension`` (dropped when None) -- filesystem.py:64-129."""
    if output_extension is not None and not output_extension.startswith("."):
        output_extension = "." + output_extension
    root = input_file_or_directory_path.rstrip(os.path.sep)
    for src in iter_files_with_extension(input_file_or_directory_path, input_extension):
        sub = "" if src == root else os.path.relpath(os.path.dirname(src), root)
        out_dir = output_directory_path if sub in ("", ".") else os.path.join(output_directory_path, sub)
        os.makedirs(out_dir, exist_ok=True)
        stem = os.path.splitext(os.path.basename(src))[0]
        yield src, os.path.join(out_dir, stem + (output_extension or ""))


def get_puzzle_file_paths(puzzle_file_or_directory_path: Optional[str] = None) -> dict:
    """name -> path of every puzzle below a directory, the benchmark's puzzles by default (filesystem.py:132-159)."""
    from pushworld_amd.config import BENCHMARK_PUZZLES_PATH, PUZZLE_EXTENSION

    if puzzle_file_or_directory_path is None:
        puzzle_file_or_directory_path = BENCHMARK_PUZZLES_PATH
    out = {}
    for path in iter_files_with_extension(puzzle_file_or_directory_path, PUZZLE_EXTENSION):
        name = os.path.split(path)[1][: -len(PUZZLE_EXTENSION)]
        if name in out:
            raise ValueError(f'Found two puzzles with the same name "{name}": {path} {out[name]}')
        out[name] = path
    return out
