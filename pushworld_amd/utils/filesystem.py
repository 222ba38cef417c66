"""Puzzle file discovery (reference: python3/src/pushworld/utils/filesystem.py:21-61).

The pool order of ``PushWorldEnv`` is the order in which this generator yields files, and
``reset`` draws from that pool with ``random.Random`` (gym_env.py:107-109,171-174), so the
``os.walk`` traversal (unsorted, like the reference) is part of the observable behaviour
(SURVEY trap T15).
"""
import os
from typing import Generator


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


def get_puzzle_file_paths(puzzle_file_or_directory_path: str) -> dict:
    """name -> path of every puzzle below a directory (filesystem.py:132-159)."""
    from pushworld_amd.config import PUZZLE_EXTENSION

    out = {}
    for path in iter_files_with_extension(puzzle_file_or_directory_path, PUZZLE_EXTENSION):
        name = os.path.split(path)[1][: -len(PUZZLE_EXTENSION)]
        if name in out:
            raise ValueError(f'Found two puzzles with the same name "{name}": {path} {out[name]}')
        out[name] = path
    return out
