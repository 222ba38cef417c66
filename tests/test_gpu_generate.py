"""-m gpu: SURVEY 8-f4 on the device -- the generator kernel, the dihedral transform kernel and the packer kernel
(symbol grid -> PwPuzzleHeader + table blob) against their host counterparts: the same generator function on the
host, the reference-equal text transforms, and pw_puzzle_parse + pw_puzzleset_create of the text."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HDR = 320  # sizeof(PwPuzzleHeader), csrc/pw_format.h


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _sections(header: bytes, blob: bytes):
    """(header with `base` zeroed, the puzzle's own bytes of the blob) of one packed puzzle"""
    base, = struct.unpack_from("<I", header, 0)
    W, H, N, G = header[4:8]
    off_small, = struct.unpack_from("<I", header, 36)  # the last section: one uint64 per movable
    used = off_small + 8 * N
    return b"\0\0\0\0" + header[4:], blob[base:base + used], (W, H, N, G)


def _assert_same_tables(dev_set, texts, order):
    """every puzzle of a device-packed set == pw_puzzle_parse + pack_puzzle of its text, section by section"""
    from pushworld_amd import _capi

    dh, db = dev_set.headers(), dev_set.blob()
    assert len(dh) == HDR * len(texts)
    for i, text in enumerate(texts):
        host = _capi.PuzzleSet([_capi.ParsedPuzzle(text, order)], -1)
        want_h, want_b, dims = _sections(host.headers(), host.blob())
        got_h, got_b, _ = _sections(dh[HDR * i:HDR * (i + 1)], db)
        assert got_h == want_h, (i, dims, text)
        assert got_b == want_b, (i, dims, text)


KW = [dict(), dict(min_puzzle_size=4, max_puzzle_size=9, min_num_walls=0, max_num_walls=7, min_num_obstacles=0,
                   max_num_obstacles=6, max_num_goal_objects=2),
      dict(object_shapes="simple", min_puzzle_size=3, max_puzzle_size=5)]


@pytest.mark.parametrize("kw", KW)
def test_device_generator_equals_the_host_instance(torch_mod, kw):
    from pushworld_amd import generate

    n = 3000
    grids, dims = generate.generate_level0_grids(n, random_seed=5, device=0, **kw)
    hg, hd = generate.generate_level0_grids(n, random_seed=5, device=-1, **kw)
    assert (dims.cpu().numpy() == hd).all() and (grids.cpu().numpy() == hg).all()
    assert (hd > 0).all()


@pytest.mark.parametrize("name", ["default", "dense", "simple"])
def test_device_generator_distribution_matches_the_reference(torch_mod, name):
    """The KERNEL's own output against the reference's histograms (tests/golden/golden_generator_stats.json, 20 000 draws of the
    imported reference per configuration): the chi-square pin of tests/test_generate_stats.py applied to what the device wrote,
    not only to the host instance the device is compared with above."""
    import json

    import test_generate_stats as tgs

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_generator_stats.json")) as f:
        ref_stats = json.load(f)
    ref = ref_stats[name]
    mine, _ = tgs.own_stats(ref["kwargs"], ref_stats["_n"], ref_stats["_shapes"], device=0)
    for key in ("width", "height", "walls", "obstacles", "goals", "shape_m1", "shape_m2", "shape_agent", "shape_obstacles"):
        assert mine[key].sum() > 0 or sum(ref[key]) == 0, key
        p = tgs.chi2_p(mine[key], ref[key])
        assert p > 1e-3, (name, key, p, mine[key].tolist(), ref[key])


@pytest.mark.parametrize("order", ["python", "cpp"])
@pytest.mark.parametrize("kw", KW[:2])
def test_device_packer_equals_host_parse_and_pack(torch_mod, kw, order):
    """Same seed -> generator kernel -> packer kernel, against the host text of the same puzzles through
    pw_puzzle_parse + pw_puzzleset_create: header and every table section byte for byte."""
    from pushworld_amd import _capi, generate

    n = 400
    pset, grids, dims = generate.generate_level0_set(n, device=0, order=order, random_seed=21, **kw)
    assert len(pset) == n and pset.puzzles is None
    g, d = grids.cpu().numpy(), dims.cpu().numpy()
    texts = [generate.grid_to_text(g[i], d[i, 0], d[i, 1]) for i in range(n)]
    _assert_same_tables(pset, texts, _capi.ORDER_PYTHON if order == "python" else _capi.ORDER_CPP)
    assert pset.max_width == int(d[:, 0].max()) + 2 and pset.max_height == int(d[:, 1].max()) + 2


def _text_to_grid(text, slot):
    """symbol grid of a puzzle text whose cells hold one element each; None otherwise"""
    rows = [line.split() for line in text.splitlines() if line.strip()]
    h, w = len(rows), len(rows[0])
    if w > slot or h > slot:
        return None
    g = np.zeros((slot, slot), np.uint8)
    for y, row in enumerate(rows):
        for x, tok in enumerate(row):
            t = tok.lower()
            if "+" in t:
                return None
            if t == ".":
                continue
            if t == "w":
                g[y, x] = 1
            elif t == "aw":
                g[y, x] = 2
            elif t == "a":
                g[y, x] = 3
            elif t[0] in "mg" and t[1:].isdigit() and int(t[1:]) < 48:
                g[y, x] = (0x40 if t[0] == "m" else 0x80) | int(t[1:])
            else:
                return None
    return g.reshape(-1), w, h


@pytest.mark.parametrize("order", ["python", "cpp"])
def test_device_packer_on_benchmark_puzzles(golden, torch_mod, order):
    """The packer on real puzzles (agent walls, up to 19 movables, two-digit names, multi-cell shapes, goals
    descending / ascending by NAME): every benchmark / test puzzle whose cells hold a single element each."""
    torch = torch_mod
    from pushworld_amd import _capi

    slot = 62
    grids, dims, texts = [], [], []
    for key in golden.keys:
        if not key.startswith(("bench:", "pytest:", "cpptest:", "rand:")):
            continue
        text = golden.text(key)
        got = _text_to_grid(text, slot)
        if got is None:
            continue
        grids.append(got[0])
        dims.append((got[1], got[2]))
        texts.append(text)
    assert len(texts) > 150
    dg = torch.as_tensor(np.stack(grids)).to("cuda:0")
    dd = torch.as_tensor(np.array(dims, np.int32)).to("cuda:0")
    o = _capi.ORDER_PYTHON if order == "python" else _capi.ORDER_CPP
    pset = _capi.PuzzleSet.from_grids(dg, dd, 0, o)
    _assert_same_tables(pset, texts, o)
    assert pset.max_movables >= 19


def test_packer_reports_the_parsers_errors(torch_mod):
    torch = torch_mod
    from pushworld_amd import _capi

    def pack(cells, w=3, h=3):
        g = np.zeros((1, 25), np.uint8)
        for (x, y), s in cells.items():
            g[0, y * 5 + x] = s
        return _capi.PuzzleSet.from_grids(torch.as_tensor(g).to("cuda:0"), torch.tensor([[w, h]], dtype=torch.int32, device="cuda:0"), 0)

    with pytest.raises(ValueError, match="agent"):                       # puzzle.py:155-157
        pack({(0, 0): 0x41})
    with pytest.raises(AssertionError, match="Goal has no associated"):  # puzzle.py:230-232
        pack({(0, 0): 3, (1, 1): 0x82})
    with pytest.raises(ValueError, match="limits"):
        pack({(0, 0): 3, (1, 1): 0x7F})                                  # M63: element numbers stop at 47
    with pytest.raises(ValueError, match="dimensions"):
        pack({(0, 0): 3}, w=0)
    ok = pack({(0, 0): 3, (1, 1): 0x41, (2, 2): 0x81, (1, 0): 2})
    assert len(ok) == 1 and ok.max_movables == 2


def test_transform_kernel_equals_text_transforms(torch_mod):
    """pw_transform_grids against pushworld_amd.transform.get_puzzle_transforms (itself equal to the reference's
    strings, tests/test_transform.py) on non-square generated puzzles, and the packer on the variants."""
    from pushworld_amd import _capi, generate
    from pushworld_amd.transform import TRANSFORM_NAMES, get_puzzle_transforms, transform_grids

    n = 60
    grids, dims = generate.generate_level0_grids(n, random_seed=8, device=0, min_puzzle_size=4, max_puzzle_size=9,
                                                 max_num_goal_objects=2, max_num_obstacles=3)
    out, odims = transform_grids(grids, dims)
    g, d = grids.cpu().numpy(), dims.cpu().numpy()
    og, od = out.cpu().numpy(), odims.cpu().numpy()
    texts = []
    for i in range(n):
        want = get_puzzle_transforms(generate.grid_to_text(g[i], d[i, 0], d[i, 1]))
        for v, name in enumerate(TRANSFORM_NAMES):
            got = generate.grid_to_text(og[8 * i + v], od[8 * i + v, 0], od[8 * i + v, 1])
            assert got == want[name], (i, name)
            texts.append(got)
        assert (od[8 * i:8 * i + 8, 0] * od[8 * i:8 * i + 8, 1] == d[i, 0] * d[i, 1]).all()
    pset = _capi.PuzzleSet.from_grids(out, odims, 0)
    _assert_same_tables(pset, texts, _capi.ORDER_PYTHON)


def test_generated_set_trains_like_the_text_pool(torch_mod):
    """generate -> transform -> pack on the device, then a random walk with rendering: the batch behaves exactly
    like one built from the texts of the same puzzles through the host parser (states, rewards, flags, pixels)."""
    torch = torch_mod
    from pushworld_amd import generate
    from pushworld_amd.puzzle import PushWorldPuzzle
    from pushworld_amd.vec_env import VecPushWorld

    pset, grids, dims = generate.generate_level0_set(40, device=0, transforms=True, random_seed=4, max_num_goal_objects=2)
    assert len(pset) == 320
    g, d = grids.cpu().numpy(), dims.cpu().numpy()
    texts = [generate.grid_to_text(g[i], d[i, 0], d[i, 1]) for i in range(len(pset))]
    B, T = 1280, 40
    kw = dict(max_steps=15, pixels_per_cell=3, border_width=1, observation="uint8", device=0, autoreset=True, resample=True,
              seed=9)
    dev = VecPushWorld(pset, B, **kw)
    txt = VecPushWorld([PushWorldPuzzle(text=t) for t in texts], B, **kw)
    assert torch.equal(dev.reset(seed=9), txt.reset(seed=9))
    gen = torch.Generator(device=dev.device).manual_seed(2)
    solved = 0
    for t in range(T):
        a = torch.randint(0, 4, (B,), generator=gen, device=dev.device, dtype=torch.uint8)
        ro, rt = dev.step(a), txt.step(a)
        for x, y in zip(ro, rt):
            assert torch.equal(x, y), t
        assert torch.equal(dev.pos, txt.pos) and torch.equal(dev.puzzle_id, txt.puzzle_id)
        solved += int(ro[2].sum())
    dev.engine.validate(dev.puzzle_id, dev.pos)


def test_device_packed_set_survives_the_pool_file(torch_mod, tmp_path):
    """A device-generated set through pw_puzzleset_save / pw_puzzleset_load: the loader's extent and
    index validation accepts the packer's fixed-slot blob, and the loaded set has the same tables."""
    from pushworld_amd import _capi, generate

    pset, _, _ = generate.generate_level0_set(200, device=0, random_seed=6, max_num_goal_objects=2)
    path = str(tmp_path / "generated.pwset")
    pset.save(path)
    back = _capi.PuzzleSet.load(path, 0)
    assert len(back) == 200 and back.blob() == pset.blob() and back.headers() == pset.headers()
    assert (back.max_width, back.max_height, back.max_movables) == (pset.max_width, pset.max_height, pset.max_movables)


def test_solvability_filter_on_a_device_generated_set(torch_mod):
    """The filter of generate.py:262-297 on a set that never existed as text (search by set index): verdicts equal
    those of ``generate.solve`` on the texts of the same puzzles."""
    from pushworld_amd import generate

    pset, grids, dims = generate.generate_level0_set(24, device=0, random_seed=13, min_puzzle_size=5, max_puzzle_size=7)
    keep = generate.solvable_mask(pset, max_states=200_000)
    g, d = grids.cpu().numpy(), dims.cpu().numpy()
    want = [generate.solve(generate.grid_to_text(g[i], d[i, 0], d[i, 1]), max_states=200_000)[0] is not None for i in range(24)]
    assert keep.tolist() == want and 0 < keep.sum()
