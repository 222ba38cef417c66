"""Host library (C++ parser / packer behind the C ABI) against the golden parse products,
the reference's error behaviour, and -- through the packed bitboards -- the reference's
collision tables.  CPU only: no kernel is launched."""
import ctypes
import hashlib
import os
import re
import struct

import numpy as np
import pytest

from pushworld_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def digest_points(points):
    return sha(np.array(sorted(points), dtype=np.int32).reshape(-1, 2))


def test_library_exports_every_declared_symbol():
    """Every function declared in include/pushworld_amd.h is exported and bound."""
    with open(os.path.join(ROOT, "include", "pushworld_amd.h")) as f:
        src = f.read()
    declared = set(re.findall(r"\b(pw_[a-z0-9_]+)\s*\(", src))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    assert lib.pw_abi_version() == 4


def test_parse_products_match_reference(golden):
    """Python object order: dimensions, states, cells, walls for all 700 golden puzzles."""
    for k in golden.keys:
        m = golden.meta[k]
        p = _capi.ParsedPuzzle(golden.text(k), _capi.ORDER_PYTHON)
        assert (p.width, p.height, p.num_movables) == (m["width"], m["height"], m["num_movables"]), k
        assert [list(x) for x in p.initial_state] == m["initial_state"], k
        assert [list(x) for x in p.goal_state] == m["goal_state"], k
        assert [sorted(map(list, c)) for c in p.object_cells] == m["object_cells"], k
        assert [sorted(map(list, c)) for c in p.goal_cells] == m["goal_cells"], k
        assert digest_points(p.wall_cells) == m["walls_sha"], k
        assert digest_points(set(p.agent_wall_cells) | set(p.wall_cells)) == m["agent_walls_prop_sha"], k
        assert p.has_agent_walls == m["has_agent_walls"], k


def test_cpp_object_order(golden):
    """cpp/test/test_pushworld_puzzle.cc:467-477 (C++ order) vs python3/test/test_puzzle.py:203-211."""
    text = golden.text("cpptest:file_parsing.pwp")
    c = _capi.ParsedPuzzle(text, _capi.ORDER_CPP)
    assert c.names == ["a", "m1", "m4", "m0", "m2", "m3"]
    assert c.initial_state == ((1, 12), (1, 3), (6, 14), (4, 1), (2, 7), (3, 8))
    assert c.goal_state == ((3, 4), (6, 5))
    p = _capi.ParsedPuzzle(text, _capi.ORDER_PYTHON)
    assert p.names == ["a", "m4", "m1", "m0", "m2", "m3"]
    assert p.goal_state == ((6, 5), (3, 4))
    assert (p.width, p.height) == (12, 18)


def test_cpp_order_matches_oracle_everywhere(golden):
    from oracle import pw_oracle

    for k in [k for k in golden.keys if not k.startswith("l0:")]:
        text = golden.text(k)
        c = _capi.ParsedPuzzle(text, _capi.ORDER_CPP)
        o = pw_oracle.OraclePuzzle(text, order="cpp", build_tables=False)
        assert c.names == o.names and c.initial_state == o.initial_state and c.goal_state == o.goal_state, k


def test_parse_errors():
    """puzzle.py:141-157, :230-232 / pushworld_puzzle.cc:216,236,292."""
    with pytest.raises(ValueError, match="same number of elements"):
        _capi.ParsedPuzzle("a . .\n. .\n")
    with pytest.raises(ValueError, match="agent"):
        _capi.ParsedPuzzle(". . .\n. m0 .\n")
    with pytest.raises(AssertionError, match="m7"):
        _capi.ParsedPuzzle("a . g7\n. m0 .\n")
    with pytest.raises(ValueError, match="64"):
        _capi.ParsedPuzzle(" ".join(["a"] + ["."] * 70) + "\n")
    big = "\n".join(" ".join(f"m{r * 6 + c}" for c in range(6)) for r in range(6)) + "\na . . . . .\n"
    with pytest.raises(ValueError, match="32"):
        _capi.ParsedPuzzle(big)
    # lower-casing, '+' cells, ragged whitespace
    p = _capi.ParsedPuzzle("A   G0+M0  .\n  W  aw+m1 .\n")
    assert p.names == ["a", "m0", "m1"] and p.goal_state == ((2, 1),) and p.has_agent_walls


# ------------------------------------------------------------------ packed tables
HDR = struct.Struct("<I4B10I")  # PwPuzzleHeader, csrc/pw_format.h




def bitboard_tables(parsed):
    """The reference's collision tables recomputed from the row-bitboard predicate the
    kernels evaluate: collides <=> shift(mask_i) & mask_j != 0 and mask_i & mask_j == 0."""
    W, H, n = parsed.width, parsed.height, parsed.num_movables
    wall = np.zeros((H, W), bool)
    for x, y in parsed.wall_cells:
        wall[y, x] = True
    awall = wall.copy()
    for x, y in parsed.agent_wall_cells:
        awall[y, x] = True
    shapes = []
    for cells in parsed.object_cells:
        w = max(c[0] for c in cells) + 1
        h = max(c[1] for c in cells) + 1
        m = np.zeros((h, w), bool)
        for x, y in cells:
            m[y, x] = True
        shapes.append(m)
    disp = [(-1, 0), (1, 0), (0, -1), (0, 1)]

    def place(m, x, y, HH, WW, ox=0, oy=0):
        """Board of size HH x WW with shape m at (x+ox, y+oy); cells outside are dropped."""
        b = np.zeros((HH, WW), bool)
        h, w = m.shape
        for yy in range(h):
            for xx in range(w):
                if m[yy, xx]:
                    X, Y = x + xx + ox, y + yy + oy
                    if 0 <= X < WW and 0 <= Y < HH:
                        b[Y, X] = True
        return b

    static = [[set() for _ in range(n)] for _ in range(4)]
    dynamic = [[[set() for _ in range(n)] for _ in range(n)] for _ in range(4)]
    for a, (dx, dy) in enumerate(disp):
        for i in range(n):
            obst = awall if i == 0 else wall
            h, w = shapes[i].shape
            for y in range(0, H - h + 1):
                for x in range(0, W - w + 1):
                    now = place(shapes[i], x, y, H, W)
                    nxt = place(shapes[i], x + dx, y + dy, H, W)
                    if (nxt & obst).any() and not (now & obst).any():
                        static[a][i].add((x, y))
            for j in range(1, n):
                hj, wj = shapes[j].shape
                S = 2 * (max(h, hj) + max(w, wj)) + 8
                c = S // 2
                bj = place(shapes[j], c, c, S, S)
                for ry in range(-h - 1, hj + 2):
                    for rx in range(-w - 1, wj + 2):
                        now = place(shapes[i], c + rx, c + ry, S, S)
                        nxt = place(shapes[i], c + rx + dx, c + ry + dy, S, S)
                        if (nxt & bj).any() and not (now & bj).any():
                            dynamic[a][i][j].add((rx, ry))
    return static, dynamic


def test_bitboard_predicate_reproduces_reference_tables(golden):
    """The formulation the kernels use is table-for-table identical to the reference's
    collision sets (sizes and SHA-256 of sorted contents) on small and medium puzzles."""
    keys = [k for k in golden.keys if k.startswith(("pytest:", "cpptest:"))]
    keys += [k for k in golden.keys if k.startswith("l0:")][::40]
    keys += ["bench:level1/2 Obstacle.pwp", "bench:level1/Choose Wisely.pwp"]
    for k in keys:
        m = golden.meta[k]
        if m["width"] * m["height"] > 400:
            continue
        p = _capi.ParsedPuzzle(golden.text(k), _capi.ORDER_PYTHON)
        static, dynamic = bitboard_tables(p)
        n = p.num_movables
        h = hashlib.sha256()
        for a in range(4):
            for i in range(n):
                assert len(static[a][i]) == m["static_sizes"][a][i], (k, a, i)
                h.update(np.array(sorted(static[a][i]), np.int32).tobytes())
                for j in range(n):
                    assert len(dynamic[a][i][j]) == m["dynamic_sizes"][a][i][j], (k, a, i, j)
                    h.update(np.array(sorted(dynamic[a][i][j]), np.int32).tobytes())
        assert h.hexdigest() == m["tables_sha"], k


def test_puzzleset_host_packing(golden):
    """Host-only packing (device < 0) works without a GPU; engines cannot be created on it."""
    ps = [_capi.ParsedPuzzle(golden.text(k)) for k in golden.keys[:40]]
    s = _capi.PuzzleSet(ps, -1)
    assert len(s) == 40 and len(s.blob()) % 16 == 0 and len(s.blob()) > 40 * 64
    assert s.max_width == max(p.width for p in ps) and s.max_movables == max(p.num_movables for p in ps)
    with pytest.raises(RuntimeError, match="no device tables"):
        _capi.Engine(s, None, 3, 1, _capi.OBS_U8)


def test_packed_set_file_round_trip_and_rejects_corruption(golden, tmp_path):
    """pw_puzzleset_save / pw_puzzleset_load (host-only sets, no GPU): the loaded set is
    byte-identical; foreign, truncated, bit-flipped and offset-corrupted files are refused."""
    keys = [k for k in golden.keys if k.startswith(("bench:level1/", "pytest:", "rand:"))][::3]
    parsed = [_capi.ParsedPuzzle(golden.text(k)) for k in keys]
    pset = _capi.PuzzleSet(parsed, -1)
    path = str(tmp_path / "pool.pwset")
    pset.save(path)
    back = _capi.PuzzleSet.load(path, -1)
    assert len(back) == len(pset) == len(keys)
    assert (back.max_width, back.max_height, back.max_movables) == (pset.max_width, pset.max_height, pset.max_movables)
    assert back.blob() == pset.blob()
    raw = open(path, "rb").read()
    assert raw[:5] == b"PWSET" and len(raw) == 64 + 320 * len(keys) + len(pset.blob())

    def refused(data):
        bad = str(tmp_path / "bad.pwset")
        with open(bad, "wb") as f:
            f.write(data)
        with pytest.raises(ValueError):
            _capi.PuzzleSet.load(bad, -1)

    refused(b"")
    refused(b"not a puzzle set" * 10)
    refused(raw[:-16])                                   # truncated
    flipped = bytearray(raw)
    flipped[64 + 320 * len(keys) + 5] ^= 0x40            # payload bit flip -> checksum
    refused(bytes(flipped))
    wrong_version = bytearray(raw)
    wrong_version[8] = 9
    refused(bytes(wrong_version))
    with pytest.raises(ValueError):
        _capi.PuzzleSet.load(str(tmp_path / "missing.pwset"), -1)
    # a well-formed file (valid checksum) whose first table offset points outside the blob
    evil = bytearray(raw)
    evil[64:68] = struct.pack("<I", 0xFFFFFFF0)         # PwPuzzleHeader.base of puzzle 0
    h = 0xCBF29CE484222325
    for byte in evil[64:]:
        h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    evil[40:48] = struct.pack("<Q", h)
    refused(bytes(evil))
    evil[64:68] = raw[64:68]                             # restoring the offset (+ its checksum) loads again
    h = 0xCBF29CE484222325
    for byte in evil[64:]:
        h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    assert struct.pack("<Q", h) == raw[40:48]

    # Crafted files with a RECOMPUTED checksum: every index the kernels use unchecked is validated on load
    # (section extents, object table, positions, movable-cell list), not only the section start offsets.
    def crafted(patch):
        data = bytearray(raw)
        patch(data)
        hh = 0xCBF29CE484222325
        for byte in data[64:]:
            hh = ((hh ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        data[40:48] = struct.pack("<Q", hh)
        return bytes(data)

    H0 = 64                      # PwPuzzleHeader of puzzle 0 (csrc/pw_format.h)
    blob0 = 64 + 320 * len(keys)
    base0, = struct.unpack_from("<I", raw, H0)
    W0, Hh0, N0, G0 = raw[H0 + 4:H0 + 8]
    off_static0, off_mcells0, n_mcells0 = struct.unpack_from("<III", raw, H0 + 20)
    off_small0, = struct.unpack_from("<I", raw, H0 + 36)
    assert n_mcells0 >= 1 and N0 >= 1
    blob_len = len(pset.blob())

    def put(off, fmt, *vals):
        return lambda d: struct.pack_into(fmt, d, off, *vals)

    cases = {
        "static section extent": put(H0 + 20, "<I", blob_len - base0 - 8),          # start inside, end outside
        "movable-cell list extent": put(H0 + 28, "<I", 4000),                        # n_mcells * 4 beyond the blob
        "shape row offset": put(H0 + 64 + 2, "<H", 60000),                           # objtab[0].row_off
        "object height": put(H0 + 64 + 1, "<B", 65),                                 # objtab[0].h > 64
        "object width": put(H0 + 64 + 0, "<B", W0 + 1),                              # objtab[0].w > W
        "initial position": put(H0 + 256, "<b", W0),                                 # init[0].x outside the grid
        "negative initial position": put(H0 + 256 + 1, "<b", -3),
        "movable cell object index": put(blob0 + base0 + off_mcells0 + 3, "<B", N0),  # mcells[0].obj >= N
        "movable cell x": put(blob0 + base0 + off_mcells0, "<B", 200),               # mcells[0].cx >= w
        "static cell kind": put(blob0 + base0 + off_static0 + 1, "<B", 0x0F),        # kind 15
        "goal count": put(H0 + 7, "<B", N0),                                         # G >= N
        "puzzle count beyond the file": put(16, "<i", 1 << 30),                      # nothing is allocated on its word
        "blob size beyond the file": put(32, "<Q", 1 << 33),
        "small-board section extent": put(H0 + 36, "<I", blob_len - base0),          # off_small at the blob's end
        "small board != shape rows": put(blob0 + base0 + off_small0 + 7, "<B", 0x80),  # a cell the shape rows lack
        "zero width": put(H0 + 4, "<B", 0),
    }
    for what, patch in cases.items():
        data = crafted(patch)
        bad = str(tmp_path / "crafted.pwset")
        with open(bad, "wb") as f:
            f.write(data)
        with pytest.raises(ValueError):
            _capi.PuzzleSet.load(bad, -1)
            pytest.fail(f"crafted file accepted: {what}")
    assert crafted(lambda d: None) == raw   # the helper reproduces the genuine checksum


def test_parser_fuzz_vectors_from_the_reference():
    """600 random puzzle texts (ragged rows, unknown / empty / duplicated element names, blank lines, tabs,
    CRLF, goals without movables ...) with the reference's verdict (tests/golden/make_parser_fuzz_golden.py):
    the C++ parser behind the C ABI and the oracle raise the same exception type or produce the same
    dimensions, state, goals, object cells, walls and agent walls."""
    import json

    from oracle import pw_oracle
    from pushworld_amd.puzzle import PushWorldPuzzle

    with open(os.path.join(ROOT, "tests", "golden", "golden_parser_fuzz.json")) as f:
        cases = json.load(f)
    assert len(cases) == 600
    errors = {"ValueError": ValueError, "AssertionError": AssertionError, "IndexError": IndexError}
    n_ok = 0
    for i, case in enumerate(cases):
        text, want = case["text"], case["result"]
        if "error" in want:
            with pytest.raises(errors[want["error"]]):
                _capi.ParsedPuzzle(text)
            with pytest.raises(errors[want["error"]]):
                pw_oracle.OraclePuzzle(text, build_tables=False)
            continue
        n_ok += 1
        o = pw_oracle.OraclePuzzle(text, build_tables=False)
        p = _capi.ParsedPuzzle(text)
        assert [p.width, p.height] == want["dimensions"] == [o.width, o.height], i
        assert [list(xy) for xy in p.initial_state] == want["initial_state"] == [list(xy) for xy in o.initial_state], i
        assert [list(xy) for xy in p.goal_state] == want["goal_state"] == [list(xy) for xy in o.goal_state], i
        assert [sorted(map(list, c)) for c in p.object_cells] == want["object_cells"], i
        assert [sorted(map(list, s)) for s in o.shapes] == want["object_cells"], i
        assert sorted(map(list, p.wall_cells)) == want["walls"] == sorted(map(list, o.wall_cells)), i
        # the reference property returns AW u W (SURVEY trap T2); the C ABI exposes the raw "aw" cells
        assert sorted(map(list, o.agent_wall_cells)) == want["agent_walls"], i
    assert n_ok > 100


def _sets_from_overlap_tables(words, d, parsed):
    """The reference's collision sets (puzzle.py:259-311) read back out of the engine's overlap tables."""
    W, H, n = parsed.width, parsed.height, parsed.num_movables
    pair_off, wall_off, R, Hs = int(d[0]), int(d[1]), int(d[2]) & 0xffff, int(d[2]) >> 16
    assert pair_off > 0 and Hs == H + 2
    dims = []
    for cells in parsed.object_cells:
        dims.append((max(c[0] for c in cells) + 1, max(c[1] for c in cells) + 1))

    def bits(rows):  # uint64 rows -> bool [len(rows) + 2, 66], one cell of zero margin all around
        a = np.zeros((len(rows) + 2, 66), bool)
        a[1:-1, 1:65] = ((rows[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
        return a

    disp = [(-1, 0), (1, 0), (0, -1), (0, 1)]
    static = [[set() for _ in range(n)] for _ in range(4)]
    dynamic = [[[set() for _ in range(n)] for _ in range(n)] for _ in range(4)]
    for i in range(n):
        w, h = dims[i]
        t = bits(words[wall_off + i * Hs: wall_off + (i + 1) * Hs])  # t[y + 2, x + 2] = object i at (x, y) overlaps
        for a, (dx, dy) in enumerate(disp):
            now = t[2:2 + H - h + 1, 2:2 + W - w + 1]
            nxt = t[2 + dy:2 + dy + H - h + 1, 2 + dx:2 + dx + W - w + 1]
            ys, xs = np.nonzero(nxt & ~now)
            static[a][i] = set(zip(xs.tolist(), ys.tolist()))
        for j in range(1, n):
            o = bits(words[pair_off + (i * n + j) * R: pair_off + (i * n + j + 1) * R])  # o[ry + h + 1 - 1, rx + w + 1 - 1]
            for a, (dx, dy) in enumerate(disp):
                now = o[1:-1, 1:-1]
                nxt = np.roll(np.roll(o, -dy, axis=0), -dx, axis=1)[1:-1, 1:-1]  # nxt[r] = o[r + d]
                ys, xs = np.nonzero(nxt & ~now)
                dynamic[a][i][j] = set(zip((xs - (w - 1)).tolist(), (ys - (h - 1)).tolist()))
                # offsets just outside the table (r + d is its first / last row or column)
                for ry in range(-1, R + 1):
                    for rx in range(-1, 65):
                        if 0 <= ry < R and 0 <= rx < 64:
                            continue
                        yy, xx = ry + dy, rx + dx
                        if 0 <= yy < R and 0 <= xx < 64 and o[yy + 1, xx + 1]:
                            dynamic[a][i][j].add((rx - (w - 1), ry - (h - 1)))
    return static, dynamic


def test_overlap_tables_are_the_reference_collision_tables(golden):
    """PW_OPT_STEP_TABLES: the sets read back out of the engine's overlap tables have the sizes and the
    SHA-256 (over sorted contents) the reference's own tables have -- every benchmark puzzle with a movable beyond
    8 x 8 cells plus a sample of the others (mode 1 = the default: every puzzle)."""
    big = []
    for k in golden.keys:
        if k.startswith("bench:") and any(max(c[0] for c in cells) >= 8 or max(c[1] for c in cells) >= 8
                                           for cells in golden.meta[k]["object_cells"]):
            big.append(k)
    assert len(big) >= 40
    rest = [k for k in golden.keys if k not in big]
    keys = big + rest[::9]
    parsed = [_capi.ParsedPuzzle(golden.text(k), _capi.ORDER_PYTHON) for k in keys]
    pset = _capi.PuzzleSet(parsed, -1)
    words0, dir0 = pset.overlap_tables(3)
    assert [bool(d[0]) for d in dir0] == [k in big for k in keys]  # mode 3: exactly the puzzles with big movables
    assert pset.overlap_tables(2)[0].size == 1
    words, dirs = pset.overlap_tables(1)
    assert all(d[0] for d in dirs)
    # automatic (the default) = every puzzle
    wa, da = pset.overlap_tables(0)
    assert np.array_equal(wa, words) and np.array_equal(da, dirs)
    for k, p, d, d0 in zip(keys, parsed, dirs, dir0):
        if d0[0]:  # same tables whichever mode built them
            n_words = p.num_movables ** 2 * (int(d[2]) & 0xffff) + p.num_movables * (int(d[2]) >> 16)
            assert np.array_equal(words[int(d[0]): int(d[0]) + n_words], words0[int(d0[0]): int(d0[0]) + n_words])
        m = golden.meta[k]
        static, dynamic = _sets_from_overlap_tables(words, d, p)
        n = p.num_movables
        h = hashlib.sha256()
        for a in range(4):
            for i in range(n):
                assert len(static[a][i]) == m["static_sizes"][a][i], (k, a, i)
                h.update(np.array(sorted(static[a][i]), np.int32).tobytes())
                for j in range(n):
                    assert len(dynamic[a][i][j]) == m["dynamic_sizes"][a][i][j], (k, a, i, j)
                    h.update(np.array(sorted(dynamic[a][i][j]), np.int32).tobytes())
        assert h.hexdigest() == m["tables_sha"], k


def test_cpp_order_with_many_ids_against_the_reference():
    """The host packer's C++ order on the puzzles of golden_cpp_order.json (ids past 10: "m10" < "m2"), whose expectations
    come from the Python reference under the permutation pushworld_puzzle.cc:262-321 defines."""
    import json

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_cpp_order.json")) as f:
        fx = json.load(f)
    for key, ent in fx.items():
        c = _capi.ParsedPuzzle(ent["text"], _capi.ORDER_CPP)
        assert c.names == ent["cpp_names"], key
        assert [list(p) for p in c.initial_state] == ent["states_cpp"][0], key
        assert [list(p) for p in c.goal_state] == ent["goal_state_cpp"], key
        p = _capi.ParsedPuzzle(ent["text"], _capi.ORDER_PYTHON)
        assert p.names == ent["python_names"], key
