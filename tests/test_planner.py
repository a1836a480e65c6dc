"""CPU: the symmetric kernel's host-side work planner (nb_debug_sym_plan).  The 8-GPU partition cannot be
run in this container, so its correctness is checked by construction: over all ranks, every unordered pair
of (2048-particle tile, 64-particle chunk) is covered exactly once — diagonal items cover a tile's own
chunks, symmetric items the chunks after it — slab rows are unique, and the ranks' work is balanced."""
import ctypes as C

import numpy as np
import pytest

import nbodysim_amd as nb

SB, CH = 2048, 64


def plan(n, rank, world, cus=256):
    lib = nb.load()
    cnt, nloc, L = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert lib.nb_debug_sym_plan(n, cus, rank, world, None, 0, C.byref(cnt), C.byref(nloc), C.byref(L)) == 0
    items = np.zeros((cnt.value, 8), np.uint32)
    assert lib.nb_debug_sym_plan(n, cus, rank, world, items.ctypes.data, cnt.value, C.byref(cnt), C.byref(nloc), C.byref(L)) == 0
    return items, int(nloc.value), int(L.value)


@pytest.mark.parametrize("n,world", [(16384, 1), (20000, 1), (70001, 1), (262144, 1), (262144, 2), (262144, 4), (262144, 8),
                                     (65536, 8), (1048576, 8), (196608, 3)])
def test_items_cover_every_tile_chunk_pair_exactly_once(n, world):
    tiles, chunks, cpt = -(-n // SB), -(-n // CH), SB // CH
    cover = np.zeros((tiles, chunks), np.int32)
    work = []
    blk_tiles = tiles if world == 1 else (n // world) // SB
    for rank in range(world):
        items, nloc, L = plan(n, rank, world)
        assert len(items) > 0 and L >= 1 and 0 < nloc <= len(items)
        # slab rows: stationary rows unique and dense per rank; one travelling row per (tile, local|cross) part
        assert sorted(items[:, 3]) == list(range(len(items)))
        w = 0
        rrow_of = {}
        groups = items[:, 6]
        assert (np.diff(groups.astype(np.int64)) >= 0).all() and (groups[:nloc] == 0).all() and (groups[nloc:] > 0).all()
        assert (groups == 2).any() == (world >= 8)           # from 8 ranks on a rank holds some local items back (late)
        for idx, (tile, c0, cnt, s_row, r_row, diag, group, _) in enumerate(items):
            assert 1 <= cnt <= L and c0 + cnt <= chunks
            local = group != 1
            if group == 2:
                assert cnt <= 2
            if local:   # pairs inside the rank's own block: tile and chunks both in block `rank`
                assert tile // blk_tiles == rank or world == 1
                assert c0 + cnt <= min((tile // blk_tiles + 1) * blk_tiles * cpt, chunks)
            else:       # cross-block pairs: chunks strictly after the tile's block
                assert not diag and c0 >= (tile // blk_tiles + 1) * blk_tiles * cpt
            if not diag:
                assert rrow_of.setdefault((int(tile), int(group)), int(r_row)) == int(r_row)
            if diag:
                assert tile * cpt <= c0 and c0 + cnt <= min((tile + 1) * cpt, chunks)
                w += cnt * 48                      # one-sided body cost
            else:
                assert c0 >= (tile + 1) * cpt
                w += cnt * 56                      # symmetric body cost
            cover[tile, c0:c0 + cnt] += 1
        assert sorted(set(rrow_of.values())) == list(range(len(set(rrow_of.values()))))
        work.append(w)
    for tile in range(tiles):
        first = tile * cpt
        assert (cover[tile, first:] == 1).all(), tile      # own chunks (diagonal) and every later chunk: once
        assert (cover[tile, :first] == 0).all(), tile      # earlier chunks belong to the earlier tile's items
    if world > 1:
        assert max(work) / (sum(work) / world) < 1.02       # equal local blocks + equal cross runs


@pytest.mark.parametrize("world", [2, 3, 4])
def test_forced_late_items_keep_the_cover_exact(monkeypatch, world):
    """NB_SYM_LATE_US forces the held-back (late) group at any world size: still every pair exactly once."""
    monkeypatch.setenv("NB_SYM_LATE_US", "40")
    n = 196608
    tiles, chunks, cpt = n // SB, n // CH, SB // CH
    cover = np.zeros((tiles, chunks), np.int32)
    for rank in range(world):
        items, nloc, L = plan(n, rank, world)
        assert (items[:, 6] == 2).any()
        assert sorted(items[:, 3]) == list(range(len(items)))
        for tile, c0, cnt, s_row, r_row, diag, group, _ in items:
            cover[tile, c0:c0 + cnt] += 1
            if group == 2:
                assert tile // (tiles // world) == rank and c0 + cnt <= (rank + 1) * (chunks // world)
    for tile in range(tiles):
        assert (cover[tile, tile * cpt:] == 1).all() and (cover[tile, :tile * cpt] == 0).all()


def test_plan_fills_the_chip():
    for n, world, lo, hi in ((262144, 1, 5000, 14000), (262144, 8, 3000, 6000), (16384, 1, 500, 1400)):
        items, nloc, L = plan(n, world // 2, world)
        assert lo <= len(items) <= hi, (n, world, len(items), L)


def test_plan_rejects_bad_arguments():
    lib = nb.load()
    cnt = C.c_uint32()
    assert lib.nb_debug_sym_plan(0, 256, 0, 1, None, 0, C.byref(cnt), None, None) != 0
    assert lib.nb_debug_sym_plan(1000, 256, 3, 2, None, 0, C.byref(cnt), None, None) != 0
    assert lib.nb_debug_sym_plan(100000, 256, 0, 3, None, 0, C.byref(cnt), None, None) != 0   # blocks must be whole tiles


def test_local_items_are_a_rank_independent_share():
    """Every rank's local part (pairs inside its own block) is 1/world of its work: the early local items hide the
    all-gather, the held-back (late) ones — about 1400 chunk-units whatever the world size, at most half — the
    reduce-scatter."""
    for world in (2, 4, 8):
        items, nloc, L = plan(262144, world - 1, world)
        early, late = items[:nloc, 2].sum(), items[items[:, 6] == 2, 2].sum()
        assert abs((early + late) / items[:, 2].sum() - 1.0 / world) < 0.03
        assert (1200 <= late <= 1400 and late <= early) if world >= 8 else late == 0
