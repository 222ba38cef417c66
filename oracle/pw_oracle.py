"""CPU oracle for the PushWorld hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module is a plain-Python / numpy *restatement* of the reference algorithm
(google-deepmind/pushworld) for the one path this repository accelerates:

    agent move -> transitive push-chain closure -> wall/object collision test
    -> goal check / reward -> RGB pixel observation (+ centred zero padding)

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the *checker*.  Nothing under
``pushworld_amd/`` imports this file; the product path is the HIP library.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
itself (``/root/reference/python3/src``, build container only) and stores its
outputs as fixtures under ``tests/golden``; ``tests/test_oracle_golden.py``
checks this restatement against every one of them (parse products, collision
table digests, trajectories, rewards, render digests) and against the
reference's own known-answer tests (``python3/test/test_puzzle.py``,
``cpp/test/test_pushworld_puzzle.cc``) re-expressed as data.

Each function cites the reference lines it follows (paths relative to the
reference checkout).  Two object orders exist in the reference and both are
reproduced here (SURVEY trap T1):

* ``order="python"``  python3/src/pushworld/puzzle.py:170-257
* ``order="cpp"``     cpp/src/pushworld_puzzle.cc:262-321
"""

from __future__ import annotations

import numpy as np

NUM_ACTIONS = 4
LEFT, RIGHT, UP, DOWN = range(4)
# puzzle.py:43-50 / pushworld_puzzle.h:101-107.  y grows downwards.
DISPLACEMENTS = ((-1, 0), (1, 0), (0, -1), (0, 1))
ACTION_FROM_CHAR = {"L": LEFT, "R": RIGHT, "U": UP, "D": DOWN}

# puzzle.py:65-79
RGB = {
    "agent": ((0x00, 0xDC, 0x00), (0x00, 0x6E, 0x00)),
    "agent_wall": ((0xFA, 0xC7, 0x1E), (0x7D, 0x64, 0x0F)),
    "goal": (None, (0xB9, 0x00, 0x00)),
    "goal_object": ((0xDC, 0x00, 0x00), (0x6E, 0x00, 0x00)),
    "movable": ((0x46, 0x9B, 0xFF), (0x23, 0x48, 0x7F)),
    "wall": ((0x0A, 0x0A, 0x0A), (0x05, 0x05, 0x05)),
}


def read_cells(text):
    """Tokenise ``.pwp`` text into ``{element id: set of (x, y)}``.

    Follows puzzle.py:133-157 (whitespace split, ``+`` joins co-located ids,
    ids are lower-cased, ``.`` is empty, interior coordinates start at 1).
    Returns ``(cells, n_cols, n_rows)``.
    """
    cells = {}
    n_cols = None
    n_rows = 0
    for line in text.splitlines(keepends=True):
        # The reference iterates over file lines; a trailing newline does not
        # create an extra line, an interior blank line does (and then raises).
        tokens = line.split()
        n_rows += 1
        if n_cols is None:
            n_cols = len(tokens)
        elif len(tokens) != n_cols:
            raise ValueError(
                f"Row {n_rows} does not have the same number of elements as the first row."
            )
        for col, token in enumerate(tokens, start=1):
            for name in token.split("+"):
                name = name.lower()
                if name != ".":
                    cells.setdefault(name, set()).add((col, n_rows))
    if "a" not in cells:
        raise ValueError("Every puzzle must have an agent object, indicated by 'a'.")
    return cells, n_cols, n_rows


def _shifted(points, origin):
    ox, oy = origin
    return {(x - ox, y - oy) for (x, y) in points}


def _bbox_size(points):
    xs = [p[0] for p in points]
    ys = [p[1] for p in points]
    return max(xs) - min(xs) + 1, max(ys) - min(ys) + 1


def _overlap(a, b, offset):
    """True iff some p in ``a`` has ``p + offset`` in ``b`` (puzzle.py:509-513)."""
    dx, dy = offset
    return any((x + dx, y + dy) in b for (x, y) in a)


def static_table(action, shape, obstacles, width, height):
    """Positions from which ``shape`` moves into ``obstacles``.

    puzzle.py:522-564 == pushworld_puzzle.cc:147-172: candidate positions are
    obstacle - cell - displacement, kept when inside
    ``[0, width-w] x [0, height-h]`` and when the object does not overlap the
    obstacles *before* moving.
    """
    ddx, ddy = DISPLACEMENTS[action]
    w, h = _bbox_size(shape)
    cands = {(ox - cx - ddx, oy - cy - ddy) for (cx, cy) in shape for (ox, oy) in obstacles}
    return {
        (px, py)
        for (px, py) in cands
        if 0 <= px <= width - w and 0 <= py <= height - h and not _overlap(shape, obstacles, (px, py))
    }


def dynamic_table(action, pusher, pushee):
    """Relative positions (pusher - pushee) at which moving the pusher hits the
    pushee.  puzzle.py:567-593 == pushworld_puzzle.cc:123-138."""
    ddx, ddy = DISPLACEMENTS[action]
    cands = {(qx - px - ddx, qy - py - ddy) for (px, py) in pusher for (qx, qy) in pushee}
    return {c for c in cands if not _overlap(pusher, pushee, c)}


class OraclePuzzle:
    """Restatement of ``PushWorldPuzzle`` (puzzle.py:100-506) for both object
    orders.  States are tuples of ``(x, y)`` int tuples, agent first."""

    def __init__(self, text, order="python", build_tables=True):
        assert order in ("python", "cpp")
        self.order = order
        cells, n_cols, n_rows = read_cells(text)
        # border walls, puzzle.py:159-168
        self.width = W = n_cols + 2
        self.height = H = n_rows + 2
        wall = cells.setdefault("w", set())
        wall.update((x, 0) for x in range(W))
        wall.update((x, H - 1) for x in range(W))
        wall.update((0, y) for y in range(H))
        wall.update((W - 1, y) for y in range(H))

        names = list(cells)
        if order == "python":
            # puzzle.py:178-179: descending *string* order ("g4" before "g10"? no:
            # plain str comparison, so "g4" > "g10").
            scan = sorted(names, reverse=True)
        else:
            # pushworld_puzzle.cc:197,274: std::map iterates ascending.
            scan = sorted(names)

        origin = {}
        shape = {}
        goal_ids = []
        for name in scan:
            pts = cells[name]
            if name in ("w", "aw"):
                origin[name] = (0, 0)
                shape[name] = set(pts)
            else:
                origin[name] = (min(p[0] for p in pts), min(p[1] for p in pts))
                shape[name] = _shifted(pts, origin[name])
            if name[0] == "g":
                mov = "m" + name[1:]
                # puzzle.py:229-232 asserts; cc:285-294 throws invalid_argument.
                assert mov in cells, f"Goal has no associated movable object: {mov}"
                goal_ids.append(name)

        movables = ["a"] + ["m" + g[1:] for g in goal_ids]
        if order == "python":
            rest = [n for n in names if n[0] == "m" and n not in movables]  # file order, :235-237
        else:
            rest = [n for n in sorted(names) if n[0] == "m" and n not in movables]  # cc:309-315
        movables += rest

        self.names = movables
        self.goal_names = goal_ids
        self.num_movables = len(movables)
        self.num_goals = len(goal_ids)
        self.shapes = [shape[n] for n in movables]
        self.sizes = [_bbox_size(s) for s in self.shapes]
        self.initial_state = tuple(origin[n] for n in movables)
        self.goal_state = tuple(origin[g] for g in goal_ids)
        self.goal_shapes = [shape[g] for g in goal_ids]
        self.wall_cells = set(wall)
        self.has_agent_walls = "aw" in cells
        self.agent_wall_cells_raw = set(cells.get("aw", ()))
        # puzzle.py:273: the agent-wall set is updated in place with the walls,
        # and that same set object is what render() later draws (trap T2).
        self.agent_wall_cells = self.agent_wall_cells_raw | self.wall_cells

        self.static = None
        self.dynamic = None
        if build_tables:
            self.build_tables()

    # ---------------------------------------------------------------- tables
    def build_tables(self):
        """puzzle.py:259-308 / pushworld_puzzle.cc:324-359."""
        n, W, H = self.num_movables, self.width, self.height
        self.static = [[None] * n for _ in range(NUM_ACTIONS)]
        self.dynamic = [[[None] * n for _ in range(n)] for _ in range(NUM_ACTIONS)]
        for a in range(NUM_ACTIONS):
            self.static[a][0] = static_table(a, self.shapes[0], self.agent_wall_cells, W, H)
            for m in range(1, n):
                self.static[a][m] = static_table(a, self.shapes[m], self.wall_cells, W, H)
            for i in range(n):
                self.dynamic[a][i][0] = set()  # the agent is never a pushee
                for j in range(1, n):
                    self.dynamic[a][i][j] = dynamic_table(a, self.shapes[i], self.shapes[j])

    # ------------------------------------------------------------------ step
    def get_next_state(self, state, action):
        """puzzle.py:348-394 / pushworld_puzzle.cc:386-460.  Returns the next
        state; ``get_next_state_moved`` also returns the C++ moved list."""
        return self.get_next_state_moved(state, action)[0]

    def get_next_state_moved(self, state, action):
        stat = self.static[action]
        dyn = self.dynamic[action]
        if tuple(state[0]) in stat[0]:
            return tuple(map(tuple, state)), []
        n = self.num_movables
        pushed = [False] * n
        pushed[0] = True
        stack = [0]
        while stack:
            i = stack.pop()  # LIFO, puzzle.py:360 / cc:418
            xi, yi = state[i]
            for j in range(1, n):
                if pushed[j]:
                    continue
                xj, yj = state[j]
                if (xi - xj, yi - yj) not in dyn[i][j]:
                    continue
                if (xj, yj) in stat[j]:
                    return tuple(map(tuple, state)), []  # transitive stopping
                pushed[j] = True
                stack.append(j)
        ddx, ddy = DISPLACEMENTS[action]
        nxt = tuple((x + ddx, y + ddy) if pushed[k] else (x, y) for k, (x, y) in enumerate(state))
        moved = [k for k in range(n) if pushed[k]]  # cc:446-457: agent first, ascending
        return nxt, moved

    # ------------------------------------------------------------- goal/reward
    def count_achieved_goals(self, state):
        """puzzle.py:396-407."""
        return sum(
            1 for s, g in zip(state[1 : 1 + self.num_goals], self.goal_state) if tuple(s) == tuple(g)
        )

    def is_goal_state(self, state):
        """puzzle.py:409-411 / cc:462-469 (vacuously true with zero goals)."""
        return tuple(map(tuple, state[1 : 1 + self.num_goals])) == self.goal_state

    def is_valid_plan(self, plan, reject_early_goal=True):
        """puzzle.py:413-424 (Python rejects plans reaching the goal early);
        cc:471-479 does not (``reject_early_goal=False``)."""
        state = self.initial_state
        for a in plan:
            if reject_early_goal and self.is_goal_state(state):
                return False
            state = self.get_next_state(state, a)
        return self.is_goal_state(state)

    # ----------------------------------------------------------------- render
    def render(self, state, border_width=2, pixels_per_cell=20):
        """puzzle.py:426-469 + _draw_object :596-638.  uint8 (H*ppc, W*ppc, 3)."""
        bw, ppc = border_width, pixels_per_cell
        if bw < 1:
            raise ValueError("border_width must be >= 1")
        if ppc < 1 + 2 * bw:
            raise ValueError("pixels_per_cell must be >= 1 + 2*border_width")
        img = np.full((self.height * ppc, self.width * ppc, 3), 255, np.uint8)

        layers = []
        if self.has_agent_walls:
            layers.append((self.agent_wall_cells, (0, 0), RGB["agent_wall"]))
        layers.append((self.wall_cells, (0, 0), RGB["wall"]))
        for k, (shape, pos) in enumerate(zip(self.shapes, state)):
            if k == 0:
                col = RGB["agent"]
            elif k <= self.num_goals:
                col = RGB["goal_object"]
            else:
                col = RGB["movable"]
            layers.append((shape, tuple(pos), col))
        for shape, pos in zip(self.goal_shapes, self.goal_state):
            layers.append((shape, pos, RGB["goal"]))

        for shape, (ox, oy), (fill, edge) in layers:
            for (cx, cy) in shape:
                r0 = (oy + cy) * ppc
                c0 = (ox + cx) * ppc
                if fill is not None:
                    img[r0 : r0 + ppc, c0 : c0 + ppc] = fill
                for ddy in (-1, 0, 1):
                    for ddx in (-1, 0, 1):
                        if (ddx or ddy) and (cx + ddx, cy + ddy) not in shape:
                            ra = r0 + (ppc - bw if ddy > 0 else 0)
                            rb = ra + (ppc if ddy == 0 else bw)
                            ca = c0 + (ppc - bw if ddx > 0 else 0)
                            cb = ca + (ppc if ddx == 0 else bw)
                            img[ra:rb, ca:cb] = edge
        return img

    def observation(self, state, max_cell_height, max_cell_width, pixels_per_cell=20, border_width=2):
        """env_utils.py:44-91: uint8 -> float32 / 255, centred zero padding."""
        img = self.render(state, border_width, pixels_per_cell).astype(np.float32) / 255
        ph = max_cell_height * pixels_per_cell - img.shape[0]
        pw = max_cell_width * pixels_per_cell - img.shape[1]
        return np.pad(img, [(ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)])

    def observation_u8(self, state, max_cell_height, max_cell_width, pixels_per_cell=20, border_width=2):
        """Same geometry as ``observation`` but kept in uint8 (pad = 0)."""
        img = self.render(state, border_width, pixels_per_cell)
        ph = max_cell_height * pixels_per_cell - img.shape[0]
        pw = max_cell_width * pixels_per_cell - img.shape[1]
        return np.pad(img, [(ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)])


class OracleEnv:
    """Restatement of the step/reset bookkeeping of gym_env.py:150-226 (and
    dm_env.py:150-234) on top of ``OraclePuzzle``; no gym types, just tuples."""

    def __init__(self, puzzle, max_steps=None):
        self.puzzle = puzzle
        self.max_steps = max_steps
        self.state = None
        self.steps = 0

    def reset(self):
        self.state = self.puzzle.initial_state
        self.steps = 0
        return self.state

    def step(self, action):
        """Returns (state, reward: python float, terminated, truncated)."""
        if self.state is None:
            raise RuntimeError("reset() must be called before step() can be called.")
        self.steps += 1
        prev = self.state
        self.state = self.puzzle.get_next_state(prev, action)
        terminated = self.puzzle.is_goal_state(self.state)
        if terminated:
            reward = 10.0
        else:
            # gym_env.py:214-221: int - int - 0.01 in float64
            reward = (
                self.puzzle.count_achieved_goals(self.state) - self.puzzle.count_achieved_goals(prev) - 0.01
            )
        truncated = False if self.max_steps is None else self.steps >= self.max_steps
        return self.state, reward, terminated, truncated


class OracleNovelty:
    """Restatement of ``NoveltyHeuristic`` (cpp/src/heuristics/novelty.cc:20-77, Lipovetzky & Geffner's
    width-based novelty capped at 3): 1 if a moved object sits at a position it never had in any state
    given before, 2 if a (moved object, other object) pair of positions is new, 3 otherwise.  Every atom of
    the moved objects is recorded whatever the result.  Positions are any hashable values."""

    def __init__(self, state_size):
        self.state_size = state_size
        self.positions = [set() for _ in range(state_size)]
        self.pairs = {}

    def estimate(self, state, moved_object_indices):
        novelty = 3
        for i in moved_object_indices:
            p_i = state[i]
            if p_i not in self.positions[i]:  # novelty.cc:43-45
                self.positions[i].add(p_i)
                novelty = 1
            for j in range(self.state_size):
                if j == i:
                    continue
                lo, hi = (j, i) if j < i else (i, j)  # smaller index first, novelty.cc:50-72
                atom = (state[lo], state[hi])
                seen = self.pairs.setdefault((lo, hi), set())
                if atom not in seen:
                    seen.add(atom)
                    if novelty > 2:
                        novelty = 2
        return novelty


def position2d(x, y):
    """pushworld_puzzle.h:32-37 / cc:176-178: ``x * 10000 + y``."""
    return x * 10000 + y


def load(path, order="python", build_tables=True):
    with open(path, "r") as f:
        return OraclePuzzle(f.read(), order=order, build_tables=build_tables)
