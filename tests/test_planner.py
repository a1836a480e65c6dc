"""CPU: the symmetric kernel's host-side work planner (nb_debug_sym_plan, nbodysim_amd/csrc/nb_plan.cpp).
The 8-GPU partition cannot be run in this container, so its correctness is checked by construction: over all
ranks, every unordered pair of (2048- or 512-particle tile, 64-particle chunk) is covered exactly once — diagonal
items cover a tile's own chunks, symmetric items the chunks after it — stationary slab rows are unique, the
travelling-slab ranges the items write are disjoint and fill the (triangular) slab exactly, and the ranks'
work is balanced."""
import ctypes as C

import numpy as np
import pytest

import hooks
import nbodysim_amd as nb
from nbodysim_amd import _lib as L

SB, CH = 2048, 64


def plan(n, rank, world, cus=256, **tuning):
    p = None
    if tuning:
        p = hooks.default_params()
        for k, v in tuning.items():
            setattr(p, k, v)
    items, info = hooks.sym_plan(n, cus, rank, world, p)
    return items, info


def check_slab_ranges(items, info, n, esz=8):
    """Every non-diagonal item writes elements [r_base + c0*64, r_base + min((c0+cnt)*64, n)) of the travelling slab:
    the ranges are disjoint and tile the slab exactly (no hole, nothing outside)."""
    sym = items[items["diag"] == 0]
    lo = sym["r_base"] + sym["c0"].astype(np.int64) * CH
    hi = sym["r_base"] + np.minimum((sym["c0"].astype(np.int64) + sym["cnt"]) * CH, n)
    order = np.argsort(lo)
    lo, hi = lo[order], hi[order]
    total = info["slab_r_bytes"] // esz
    if len(lo) == 0:
        assert total == 0
        return
    assert lo[0] == 0 and 0 <= total - hi[-1] <= 1
    gaps = lo[1:] - hi[:-1]                 # back to back, but for the padding element after an odd-length (ragged) segment
    assert ((gaps == 0) | (gaps == 1)).all()
    units = int(sym["cnt"].sum())
    assert total <= units * CH + len(lo) and total > (units - len(set(sym["tile"]))) * CH      # = 64 per unit, ragged last chunk aside


@pytest.mark.parametrize("n,world,tile", [(16384, 1, 0), (20000, 1, 0), (25000, 1, 512), (25000, 1, 2048), (70001, 1, 0), (70001, 1, 2048),
                                          (262144, 1, 0), (131072, 1, 512), (262144, 2, 0), (262144, 4, 0), (262144, 8, 0),
                                          (65536, 8, 0), (1048576, 8, 0), (196608, 3, 0)])
def test_items_cover_every_tile_chunk_pair_exactly_once(n, world, tile):
    """tile = 0: the library's choice (wave-split tiles of 512 for single handles below 49 152 bodies, else 2048)."""
    tuning = {"sym_tile": tile} if tile else {}
    SB = plan(n, 0, world, **tuning)[1]["tile_particles"]
    assert SB == (tile or (512 if world == 1 and n < 49152 else 2048))
    quantum = (4 if SB == 512 else 1) * (2 if n >= 65536 else 1)       # chunks per item come in multiples of this
    tiles, chunks, cpt = -(-n // SB), -(-n // CH), SB // CH
    cover = np.zeros((tiles, chunks), np.int32)
    work = []
    blk_tiles = tiles if world == 1 else (n // world) // SB
    cross_totals = set()
    for rank in range(world):
        items, info = plan(n, rank, world, **tuning)
        nloc, Lc = info["items_local"], info["chunks_per_item"]
        assert len(items) == info["items"] > 0 and Lc >= 1 and 0 < nloc <= len(items) and Lc % quantum == 0
        assert info["items_local"] + info["items_cross"] + info["items_late"] == info["items"]
        assert info["tiles"] == tiles and info["rows_s"] == len(items)
        cross_totals.add(info["cross_units_total"])
        # slab rows: stationary rows unique and dense per rank
        assert sorted(items["s_row"]) == list(range(len(items)))
        check_slab_ranges(items, info, n)
        groups = items["group"].astype(np.int64)
        assert (np.diff(groups) >= 0).all() and (groups[:nloc] == 0).all() and (groups[nloc:] > 0).all()
        assert (groups == 2).any() == (world >= 8)           # from 8 ranks on a rank holds some local items back (late)
        w = 0
        for it in items:
            tile, c0, cnt, diag, group = int(it["tile"]), int(it["c0"]), int(it["cnt"]), int(it["diag"]), int(it["group"])
            assert 1 <= cnt <= Lc and c0 + cnt <= chunks
            if group == 2:
                assert cnt <= 2
            if group != 1:   # pairs inside the rank's own block: tile and chunks both in block `rank`
                assert tile // blk_tiles == rank or world == 1
                assert c0 + cnt <= min((tile // blk_tiles + 1) * blk_tiles * cpt, chunks)
            else:            # cross-block pairs: chunks strictly after the tile's block
                assert not diag and c0 >= (tile // blk_tiles + 1) * blk_tiles * cpt
            if diag:
                assert tile * cpt <= c0 and c0 + cnt <= min((tile + 1) * cpt, chunks)
                w += cnt * 48                      # one-sided body cost
            else:
                assert c0 >= (tile + 1) * cpt
                w += cnt * 56                      # symmetric body cost
            cover[tile, c0:c0 + cnt] += 1
        assert info["units_local"] + info["units_cross"] + info["units_late"] == int(items["cnt"].sum())
        work.append(w)
    assert len(cross_totals) == 1                            # rank-independent figure the ranks compare at start-up
    for tile in range(tiles):
        first = tile * cpt
        assert (cover[tile, first:] == 1).all(), tile      # own chunks (diagonal) and every later chunk: once
        assert (cover[tile, :first] == 0).all(), tile      # earlier chunks belong to the earlier tile's items
    if world > 1:
        assert max(work) / (sum(work) / world) < 1.02       # equal local blocks + equal cross runs


@pytest.mark.parametrize("world", [2, 3, 4])
def test_forced_late_items_keep_the_cover_exact(world):
    """nb_params.sym_late_us forces the held-back (late) group at any world size: still every pair exactly once."""
    n = 196608
    tiles, chunks, cpt = n // SB, n // CH, SB // CH
    cover = np.zeros((tiles, chunks), np.int32)
    for rank in range(world):
        items, info = plan(n, rank, world, sym_late_us=40.0)
        assert (items["group"] == 2).any() and info["items_late"] == int((items["group"] == 2).sum())
        assert sorted(items["s_row"]) == list(range(len(items)))
        check_slab_ranges(items, info, n)
        for it in items:
            cover[it["tile"], it["c0"]:it["c0"] + it["cnt"]] += 1
            if it["group"] == 2:
                assert it["tile"] // (tiles // world) == rank and it["c0"] + it["cnt"] <= (rank + 1) * (chunks // world)
        none, info0 = plan(n, rank, world, sym_late_us=-1.0)
        assert info0["items_late"] == 0
    for tile in range(tiles):
        assert (cover[tile, tile * cpt:] == 1).all() and (cover[tile, :tile * cpt] == 0).all()


def test_travelling_slab_is_triangular():
    """The travelling partials take tiles x n / 2 elements, not tiles x n: 2 GiB at N = 1 048 576 fp32 (4 GiB before),
    133 MiB at the headline N = 262 144, and the 8 ranks of a sharded run hold an eighth each."""
    for n, lo, hi in ((262144, 126, 136), (1048576, 2040, 2052)):
        items, info = plan(n, 0, 1)
        assert lo * 2**20 <= info["slab_r_bytes"] <= hi * 2**20, info
        tiles = n // SB
        assert info["slab_r_bytes"] == 8 * sum(n - (i + 1) * SB for i in range(tiles))
        assert info["coverage_entries"] == tiles * (tiles - 1) // 2
    whole = plan(1048576, 0, 1)[1]["slab_r_bytes"]
    parts = [plan(1048576, r, 8)[1]["slab_r_bytes"] for r in range(8)]
    assert sum(parts) == whole and max(parts) < 1.05 * whole / 8


def test_tuning_fields_replace_the_environment_switches():
    items, info = plan(262144, 0, 1, sym_chunks_per_item=16)
    assert info["chunks_per_item"] == 16 and int(items["cnt"].max()) == 16
    uniform, iu = plan(262144, 0, 1, flags=L.NB_FLAG_NO_GUIDED_TAIL)
    guided, ig = plan(262144, 0, 1)
    assert iu["items"] < ig["items"] and int(uniform["cnt"].min()) >= 1
    assert set(np.unique(uniform[uniform["diag"] == 0]["cnt"])) <= {iu["chunks_per_item"]} | set(range(1, iu["chunks_per_item"]))
    early, ie = plan(262144, 0, 1, sym_tail=(C.c_float * 3)(0.5, 0.7, 0.9))
    assert ie["items"] > ig["items"]                         # finer items start earlier in the launch


def test_plan_fills_the_chip():
    for n, world, lo, hi in ((262144, 1, 5000, 14000), (262144, 8, 3000, 6000), (16384, 1, 500, 2400)):
        items, info = plan(n, world // 2, world)
        assert lo <= len(items) <= hi, (n, world, len(items), info["chunks_per_item"])


def test_plan_rejects_bad_arguments():
    lib = hooks.lib()
    info = L.nb_sym_info()
    info.struct_size = C.sizeof(L.nb_sym_info)
    assert lib.nb_debug_sym_plan(0, 256, 0, 1, None, None, 0, C.byref(info)) == L.NB_EINVAL
    assert lib.nb_last_error_code() == L.NB_EINVAL
    assert lib.nb_debug_sym_plan(1000, 256, 3, 2, None, None, 0, C.byref(info)) != 0
    assert lib.nb_debug_sym_plan(100000, 256, 0, 3, None, None, 0, C.byref(info)) != 0   # blocks must be whole tiles
    info.struct_size = 4
    assert lib.nb_debug_sym_plan(262144, 256, 0, 1, None, None, 0, C.byref(info)) == L.NB_EINVAL


def test_local_items_are_a_rank_independent_share():
    """Every rank's local part (pairs inside its own block) is 1/world of its work: the early local items hide the
    all-gather, the held-back (late) ones — about 1400 chunk-units whatever the world size, at most half — the
    reduce-scatter."""
    for world in (2, 4, 8):
        items, info = plan(262144, world - 1, world)
        early, late = info["units_local"], info["units_late"]
        assert abs((early + late) / int(items["cnt"].sum()) - 1.0 / world) < 0.03
        assert (1200 <= late <= 1400 and late <= early) if world >= 8 else late == 0


def test_small_system_plans_follow_the_measured_rules():
    """Round 4's measurement-derived rules (planner bullet): whole-system wave-split plans use the finest uniform items
    (one chunk per wave: 4 chunks per item) and NO guided tail while the item count stays within 25 per CU — their step is
    9.1 + 7.3 x ceil(items / CUs) us, what counts is the count — then 8 chunks per item with the late tail up to the classic tiles'
    size; classic plans start the tail early only between 1.5 and 5 rounds of workgroups."""
    for n in (5632, 7168, 10000, 16384, 25000, 36000, 40000):
        items, info = plan(n, 0, 1)
        tiles = info["tiles"]
        assert info["tile_particles"] == 512 and info["chunks_per_item"] == 4 and info["items"] <= 25 * 256
        # uniform: every item holds 4 chunks except the one remainder per (tile, kind); the minimum possible count
        assert int(np.sum(items["cnt"] < 4)) <= 2 * tiles and int(items["cnt"].max()) == 4
        chunks = (n + CH - 1) // CH
        want = sum(-(-(min((I + 1) * 8, chunks) - I * 8) // 4) + -(-(chunks - min((I + 1) * 8, chunks)) // 4) for I in range(tiles))
        assert len(items) == want
    for n in (41000, 49000):
        items, info = plan(n, 0, 1)
        assert info["tile_particles"] == 512 and info["chunks_per_item"] == 8
        assert int(np.sum(items["cnt"] < 8)) > 4 * info["tiles"]                         # the late tail cut the end of the list finer
    # an explicit item size or explicit thresholds are taken as given
    items, info = plan(25000, 0, 1, sym_chunks_per_item=8)
    assert info["chunks_per_item"] == 8
    # the fine-item rule holds where it was measured (below 49 152 bodies): tiles of 512 FORCED at the headline size keep the
    # size-derived item length (a multiple of the wave quantum), not 8-16 chunks — ~65 000 items and 0.5 GiB of partials
    items, info = plan(262144, 0, 1, sym_tile=512)
    assert info["tile_particles"] == 512 and info["chunks_per_item"] >= 32 and info["chunks_per_item"] % 8 == 0
    assert len(items) < 12000              # (the 0.5 GiB of travelling partials is the price of 512 tiles of 512 at this n, whatever the item length)
    items64, info64 = plan(65536, 0, 1, sym_tile=512)
    assert info64["chunks_per_item"] >= 8 and len(items64) < 12000
    # classic tiles (fp64 handles): late tail below 1.5 rounds of workgroups, early tail between 1.5 and 5
    p = dict(precision=L.NB_FP64)
    late16, _ = plan(16384, 0, 1, **p)
    early48, _ = plan(49152, 0, 1, **p)
    forced_late = plan(49152, 0, 1, sym_tail=(C.c_float * 3)(0.85, 0.94, 0.98), **p)[0]
    forced_early = plan(16384, 0, 1, sym_tail=(C.c_float * 3)(0.65, 0.85, 0.95), **p)[0]
    assert len(late16) < len(forced_early) and len(early48) > len(forced_late)
