"""No-GPU checks of the C-ABI boundary: the library builds/loads, exports every symbol
include/riab_hip.h declares, the ctypes structs mirror the header, and argument errors
are reported before any launch."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "riab_hip.h")


@pytest.fixture(scope="module")
def L():
    from ratinabox_amd import _lib
    return _lib


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(riab_[a-z_]+)\s*\(", src)))


def test_header_symbols_are_exported(L):
    syms = declared_symbols()
    assert "riab_agent_step" in syms and "riab_place_cells" in syms and len(syms) >= 10
    for s in syms:
        assert hasattr(L.lib, s), f"{s} declared in riab_hip.h but not exported by libriab_hip.so"
    assert sorted(L.PROTOTYPES) == syms, "ctypes prototypes and header declarations differ"
    assert L.lib.riab_abi_version() == L.ABI_VERSION


def test_library_is_in_tree(L):
    assert os.path.realpath(L.LIB_PATH).startswith(os.path.realpath(os.path.join(ROOT, "ratinabox_amd")))


def test_struct_layouts_match_header(L):
    # natural C layout of the three ABI structs on LP64
    assert C.sizeof(L.RiabEnv) == 4 * 8 + 8 + 4 + 4 + 8 + 4 * 4 and L.RiabEnv.polygon.offset == 56
    assert C.sizeof(L.RiabMotion) == 8 * 8 + 4 + 4 + 5 * 8 + 8 + 4 + 4 + 2 * 8 and L.RiabMotion.wall_grid.offset == 112   # (+ wall_grid, n + padding, wd, lmax)
    assert L.RiabRateIO.pos_ld.offset == 32 and L.RiabRateIO.rates.offset == 56
    assert L.RiabRateIO.dt.offset == 80 and L.RiabRateIO.seed.offset == 96
    assert C.sizeof(L.RiabRateIO) == 128
    src = open(HEADER).read()
    for name, val in (("RIAB_STATE_ROWS", L.STATE_ROWS), ("RIAB_HIST_ROWS", L.HIST_ROWS),
                      ("RIAB_MAX_WALLS", L.MAX_WALLS), ("RIAB_ABI_VERSION", L.ABI_VERSION)):
        assert int(re.search(rf"#define {name} (\d+)", src).group(1)) == val


def test_argument_errors_before_launch(L):
    """Negative codes are produced by validation only — no device needed."""
    io = L.RiabRateIO()
    env = L.RiabEnv()
    assert L.lib.riab_place_cells(None, io, None, 4, 0, 0, 0.2, None) == -1
    assert L.lib.riab_place_cells(env, io, C.c_void_p(16), 4, 0, 0, 0.2, None) == -1  # io has null pointers
    assert L.lib.riab_grid_cells(io, None, 4, 0, 0.0, None) == -1
    assert L.lib.riab_head_direction_cells(io, None, 4, None) == -1
    assert L.lib.riab_agent_step(None, None, None, 4, 0, None, None, None, None, None, 0, 0, 1, None, None, None) == -1
    m = L.RiabMotion()
    env.n_walls = 1000
    env.walls = 16
    assert L.lib.riab_agent_step(env, m, C.c_void_p(16), 4, 0, None, None, None, None, None, 0, 0, 1, None, None, None) == -3
    env.n_walls = 0
    io.pos_x = io.pos_y = io.rates = 16
    io.T, io.B, io.pos_ld = 1, 6, 8
    assert L.lib.riab_place_cells(env, io, C.c_void_p(16), 4, 0, 0, 0.2, None) == -2  # B % 4
    io.B, io.rates = 8, 20
    assert L.lib.riab_place_cells(env, io, C.c_void_p(16), 4, 0, 0, 0.2, None) == -2  # misaligned rows
    assert L.lib.riab_fill(None, 16, 0.0, None) == -1 and L.lib.riab_fill(C.c_void_p(16), 10, 0.0, None) == -2
    for code, frag in ((0, "ok"), (-1, "invalid"), (-2, "multiple of 4"), (-3, "too many"), (-4, "not supported")):
        assert frag in L.strerror(code)


def test_no_cpu_fallback_in_product():
    """The product package never imports the oracle and has no CPU compute path: outside comments and
    docstrings (which may cite it) no product source mentions `oracle` at all."""
    import io
    import tokenize
    pkg = os.path.join(ROOT, "ratinabox_amd")
    checked = 0
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            src = open(os.path.join(dirpath, f)).read()
            assert "import oracle" not in src and "from oracle" not in src, f
            code = [tok.string for tok in tokenize.generate_tokens(io.StringIO(src).readline)
                    if tok.type not in (tokenize.COMMENT, tokenize.STRING)]
            assert not any("oracle" in t for t in code), f"{f}: code refers to the oracle"
            checked += 1
    assert checked >= 8


def test_step_plan_validation_without_gpu(L):
    """riab_plan_*: argument validation and the 'history chunk full' report happen before any launch."""
    env, m = L.RiabEnv(), L.RiabMotion()
    assert not L.lib.riab_plan_create(None, m, C.c_void_p(16), 4, 0, 0, 0, C.c_void_p(16), None)
    assert not L.lib.riab_plan_create(env, m, None, 4, 0, 0, 0, C.c_void_p(16), None)
    h = L.lib.riab_plan_create(env, m, C.c_void_p(16), 4, 0, 7, 5, C.c_void_p(16), None)
    assert h and L.lib.riab_plan_step_index(h) == 5
    assert L.lib.riab_plan_step(h, 0, None) == -1
    pop = L.RiabPopulation()
    pop.kind, pop.n = 99, 4
    assert L.lib.riab_plan_add(h, pop) == -1
    pop.kind = L.POP_KINDS["place"]
    assert L.lib.riab_plan_add(h, pop) == 0
    assert L.lib.riab_plan_set_population_history(h, 3, C.c_void_p(16), None, 4) == -1
    assert L.lib.riab_plan_set_population_history(h, 0, C.c_void_p(16), None, 2) == 0
    assert L.lib.riab_plan_set_agent_history(h, C.c_void_p(16), 8) == 0
    assert L.lib.riab_plan_rows_free(h) == 2
    assert L.lib.riab_plan_step(h, 3, None) == L.EFULL  # nothing launched
    assert "full" in L.strerror(L.EFULL)
    m.has_drift = 1
    assert L.lib.riab_plan_set_motion(h, m, None) == -1
    L.lib.riab_plan_destroy(h)


def test_watch_list_and_fast_repeat_record(L):
    """The short road of a repeated simulate() (Agent._fast_record / RiabSimulate.watch): the arrays a population's
    tables derive from are watched by identity in Python and by CONTENT in the library (memcmp against a snapshot,
    before anything is launched: RIAB_ECHANGED) — host memory only, no device needed."""
    import numpy as np
    import ratinabox_amd as riab
    np.random.seed(0)
    env = riab.Environment({})
    ag = riab.Agent(env, {"n_agents": 8, "device": "cpu"})
    pcs = riab.PlaceCells(ag, {"n": 16})
    gcs = riab.GridCells(ag, {"n": 8})
    hdc = riab.HeadDirectionCells(ag, {"n": 4})
    rec = ag._fast_record([pcs, gcs, hdc])
    assert rec is not None and rec["n_watch"] == 2 + 3 + 2 + 1          # centres, widths | scales, phases, w | angles, tunings | walls
    w = C.addressof(rec["watch"])
    assert L.lib.riab_watch_compare(w, rec["n_watch"]) == 0
    pcs.place_cell_centres[-1] = [0.123, 0.456]                          # the in-place edit of reference tests/test_advanced.py:59
    assert L.lib.riab_watch_compare(w, rec["n_watch"]) == L.ECHANGED
    assert "snapshot" in L.strerror(L.ECHANGED)
    rec = ag._fast_record([pcs, gcs, hdc])                               # (a new record holds the new content)
    assert L.lib.riab_watch_compare(C.addressof(rec["watch"]), rec["n_watch"]) == 0
    env.walls[0, 0, 0] += 1e-9
    assert L.lib.riab_watch_compare(C.addressof(rec["watch"]), rec["n_watch"]) == L.ECHANGED
    env.walls[0, 0, 0] -= 1e-9
    # scalar parameters and replaced attributes are compared in Python: the getters return what the record holds
    N, arrs, getter, sc = rec["pops"][0]
    assert N is pcs and getter(pcs) == sc and all(getattr(pcs, name) is a for name, a in arrs)
    pcs.max_fr = 2.0
    assert getter(pcs) != sc
    pcs.max_fr = 1.0
    pcs.place_cell_widths = pcs.place_cell_widths.copy()                 # same content, another object: caught by identity
    assert not all(getattr(pcs, name) is a for name, a in arrs)
    # populations whose tables do not come straight from float64 array attributes have no record
    assert ag._fast_record([riab.BoundaryVectorCells(ag, {"n": 4})]) is None
    pcs.place_cell_centres = pcs.place_cell_centres.astype(np.float32)
    assert ag._fast_record([pcs]) is None
    assert L.lib.riab_watch_compare(None, 1) == -1 and L.lib.riab_watch_compare(None, 0) == 0
