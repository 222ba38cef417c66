"""The hand-built collision cases of the C++ reference's unit tests (cpp/test/test_pushworld_puzzle.cc:84-257)
re-expressed as data.  The reference builds ``ObjectCollisions`` tables by hand; this engine only takes puzzles, so
each case is a small puzzle text (written for this repository) whose geometry produces exactly those tables, plus
the literal expectations of the reference test translated by the offset between the two coordinate frames
(parsed puzzles carry a border wall, pushworld_puzzle.cc:216-260, so file cell (c, r) is position (c + 1, r + 1)).

Each step: (state, action, expected next state, expected moved_object_indices or None where the reference test
does not check them).  Used by tests/test_oracle_golden.py (CPU oracles) and tests/test_gpu_expand.py (HIP)."""

L, R, U, D = 0, 1, 2, 3

OPEN = ". . .\n. A .\n. . ."


def _agent_movement():
    """test_agent_movement, cc:84-142: a free agent moves in all four directions (moved = [0]); an agent wall on a
    side blocks exactly that action (state unchanged, moved empty).  Reference frame: agent at (1, 1), no walls;
    here the agent sits at (2, 2) in the middle of an open 3 x 3 field: offset (+1, +1)."""
    s = ((2, 2),)
    cases = [("agent free", OPEN, [(s, L, ((1, 2),), [0]), (s, R, ((3, 2),), None), (s, U, ((2, 1),), None),
                                   (s, D, ((2, 3),), None)])]
    cases.append(("left agent wall", ". . .\nAW A .\n. . .", [(s, L, s, []), (s, R, ((3, 2),), [0])]))
    cases.append(("left+right agent walls", ". . .\nAW A AW\n. . .", [(s, L, s, []), (s, R, s, [])]))
    cases.append(("left+right+top", ". AW .\nAW A AW\n. . .", [(s, U, s, []), (s, D, ((2, 3),), [0])]))
    cases.append(("all four", ". AW .\nAW A AW\n. AW .", [(s, a, s, []) for a in (L, R, U, D)]))
    return cases


def _pushing():
    """test_pushing, cc:145-171: reference coordinates unchanged (agent (1, 1), object (2, 1))."""
    text = "A M0 . . .\n. . . . ."
    s0 = ((1, 1), (2, 1))
    s1 = ((2, 1), (3, 1))
    return [("pushing", text, [(s0, D, ((1, 2), (2, 1)), [0]), (s0, R, s1, [0, 1]), (s1, R, ((3, 1), (4, 1)), [0, 1])])]


def _transitive_pushing():
    """test_transitive_pushing, cc:175-222: reference agent (1, 1), objects (3, 1) and (5, 1); here one free row on
    top (the reference's final UP leaves the row): offset (0, +1)."""
    text = ". . . . . . .\nA . M0 . M1 . .\n. . . . . . ."
    s0 = ((1, 2), (3, 2), (5, 2))
    s1 = ((2, 2), (3, 2), (5, 2))
    s2 = ((3, 2), (4, 2), (5, 2))
    s3 = ((4, 2), (5, 2), (6, 2))
    return [("transitive pushing", text, [(s0, D, ((1, 3), (3, 2), (5, 2)), [0]), (s0, R, s1, None),
                                          (s1, R, s2, [0, 1]), (s2, R, s3, [0, 1, 2]),
                                          (s3, U, ((4, 1), (5, 2), (6, 2)), None)])]


def movement_cases():
    return _agent_movement() + _pushing() + _transitive_pushing()


# test_goal_checking, cc:225-257: satisfiesGoal looks at the goal objects only (state[1 .. G]); reference
# coordinates unchanged.  (text, [(state, satisfied)])
def goal_cases():
    rows = [["."] * 8 for _ in range(9)]
    rows[0][0] = "A"      # (1, 1)
    rows[1][1] = "M1"     # (2, 2)
    rows[2][2] = "M2"     # (3, 3)
    rows[4][1] = "G1"     # (2, 5)
    one = "\n".join(" ".join(r) for r in rows)
    rows[5][2] = "G2"     # (3, 6)
    two = "\n".join(" ".join(r) for r in rows)
    return [
        ("one goal", one, [(((1, 1), (2, 5), (3, 3)), True), (((2, 1), (2, 5), (3, 5)), True),
                           (((1, 1), (3, 5), (3, 3)), False), (((2, 1), (2, 2), (3, 6)), False)]),
        ("two goals", two, [(((5, 1), (2, 5), (3, 6)), True), (((2, 8), (2, 5), (3, 6)), True),
                            (((1, 1), (2, 5), (3, 3)), False), (((1, 1), (2, 2), (3, 6)), False)]),
    ]
