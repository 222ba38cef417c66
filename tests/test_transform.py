"""pushworld_amd.transform against golden vectors generated from the reference
(tests/golden/make_transform_golden.py) and the reference's own property test
(python3/test/test_transform.py:24-79: transformed plans solve transformed puzzles)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_transforms_equal_reference_strings(tmp_path):
    from pushworld_amd.transform import TRANSFORM_NAMES, create_transformed_puzzles, get_puzzle_transforms

    with open(os.path.join(ROOT, "tests", "golden", "golden_transforms.json")) as f:
        golden = json.load(f)
    assert len(golden) >= 6
    for rel, want in golden.items():
        with open(os.path.join(ROOT, rel)) as f:
            got = get_puzzle_transforms(f.read())
        assert list(got) == list(TRANSFORM_NAMES) and len(got) == 8
        assert got == want, rel
    src = tmp_path / "in" / "sub"
    src.mkdir(parents=True)
    rel = "tests/puzzles/ref_python/trivial_obstacle.pwp"
    (src / "p.pwp").write_text(open(os.path.join(ROOT, rel)).read())
    (src / "ignored.txt").write_text("x")
    create_transformed_puzzles(str(tmp_path / "in"), str(tmp_path / "out"))
    files = sorted(os.listdir(tmp_path / "out" / "sub"))
    assert files == sorted(f"p_{n}.pwp" for n in TRANSFORM_NAMES)
    assert (tmp_path / "out" / "sub" / "p_r90_flipped.pwp").read_text() == golden[rel]["r90_flipped"]


@pytest.mark.gpu
def test_transformed_plans_solve_transformed_puzzles():
    """test_transform.py:24-79 on the GPU step engine, for the reference's puzzle and two benchmark
    puzzles with their human solutions."""
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.transform import get_puzzle_transforms, transform_plan

    cases = [(os.path.join(ROOT, "tests", "puzzles", "ref_python", "shortest_path_tool.pwp"), [0, 2, 2, 2, 1, 3, 3, 3, 3])]
    chars = {"L": 0, "R": 1, "U": 2, "D": 3}
    for name in ("level1/2 Obstacle", "level2/Pull Dont Push"):
        with open(os.path.join(ROOT, "pushworld_amd", "data", "solutions", name + ".yaml")) as f:
            plan = next([chars[c] for c in line.split(":", 1)[1].strip()] for line in f if line.startswith("plan:"))
        cases.append((os.path.join(ROOT, "pushworld_amd", "data", "puzzles", name + ".pwp"), plan))
    for path, plan in cases:
        with open(path) as f:
            text = f.read()
        assert PushWorldPuzzle(text=text).is_valid_plan(plan)
        for tname, ttext in get_puzzle_transforms(text).items():
            pz = PushWorldPuzzle(text=ttext)
            assert pz.is_valid_plan(transform_plan(plan, tname)), (path, tname)
    with pytest.raises(ValueError):
        transform_plan([0], "r45")
