// nb_plan.cpp — planner of the symmetric scheme (see nb_plan.h).  Host only, no device calls.
//
// The particle range is cut into `world` equal blocks (block r is integrated by rank r; blocks are whole
// tiles).  Rank r evaluates
//   LOCAL items  every unordered pair INSIDE its own block: the diagonal items of its tiles and the
//                symmetric items whose chunks lie in the same block — they need only positions the rank
//                has just produced itself, so they run while the all-gather of the other blocks is in flight;
//   CROSS items  an equal share of the pairs between different blocks: the ordered list of all
//                (tile I, L-chunk slice after I's block) items is cut into `world` runs of equal work;
//   LATE items   (world > 1) a small tail of the local items, held back until the cross items are done: they
//                touch only the rank's own block, so they can run while the reduce-scatter of the partial
//                accelerations is in flight, and are folded in by the integrate step.
#include "nb_plan.h"

#include <cmath>

namespace nbk {

namespace {
struct Geometry {
    uint32_t n, world, sb, tiles, chunks, cpt, tpb;
    Geometry(uint32_t n_, uint32_t world_, uint32_t sb_) : n(n_), world(world_ ? world_ : 1), sb(sb_ ? sb_ : SYM_SB)
    {
        tiles = (n + sb - 1) / sb;
        chunks = (n + SYM_CH - 1) / SYM_CH;
        cpt = sb / SYM_CH;
        tpb = world > 1 ? (n / world) / sb : tiles;          // tiles per block
        if (tpb == 0) tpb = 1;
    }
    uint32_t block_of(uint32_t I) const { return I / tpb; }
    uint32_t block_end_chunk(uint32_t I) const               // first chunk after I's block
    {
        const uint64_t e = (uint64_t)(I / tpb + 1) * tpb * cpt;
        return (world > 1 && e < chunks) ? (uint32_t)e : chunks;
    }
    uint32_t diag_end(uint32_t I) const { return (I + 1) * cpt < chunks ? (I + 1) * cpt : chunks; }
};
}  // namespace

void sym_units(uint32_t n, uint32_t world, uint64_t *local_sym_units, uint64_t *cross_total, uint32_t sb)
{
    const Geometry g(n, world, sb);
    uint64_t local_max = 0, cross = 0;
    std::vector<uint64_t> local_of(g.world, 0);
    for (uint32_t I = 0; I < g.tiles; ++I) {
        const uint32_t be = g.block_end_chunk(I);
        const uint32_t b = g.world > 1 ? g.block_of(I) : 0;
        if (b < g.world) local_of[b] += be - g.diag_end(I);
        cross += g.chunks - be;
    }
    for (uint64_t v : local_of) if (v > local_max) local_max = v;
    if (local_sym_units) *local_sym_units = local_max;
    if (cross_total) *cross_total = cross;
}

uint64_t sym_slab_r_bound(uint32_t n, uint32_t world, uint32_t sb)
{
    uint64_t local = 0, cross = 0;
    sym_units(n, world, &local, &cross, sb);
    const uint64_t w = world ? world : 1;
    // a rank's cross run is cut at item boundaries: at most one item (<= one tile row of chunks) over its share
    return (local + (cross + w - 1) / w + (uint64_t)(n / SYM_CH) + (sb ? sb : SYM_SB) / SYM_CH) * SYM_CH;
}

void build_sym_plan(uint32_t n, uint32_t cus, uint32_t rank, uint32_t world, const SymTuning &tune, SymPlan &pl)
{
    const Geometry g(n, world, tune.sb);
    world = g.world;
    const uint32_t tiles = g.tiles, chunks = g.chunks;
    if (cus < 1) cus = 1;

    // work (in chunks) of this rank: its local pairs + 1/world of the cross pairs
    uint64_t local = 0, cross_total = 0;
    for (uint32_t I = 0; I < tiles; ++I) {
        const uint32_t be = g.block_end_chunk(I);
        if (g.block_of(I) == rank || world == 1) local += be - I * g.cpt;       // own chunks (diagonal) + later chunks of the block
        cross_total += chunks - be;
    }
    // workgroups wanted (profiles/history/r01_*sweep.log): with the guided tail 24 per CU run as fast as 32 (L = 43-48 vs 33
    // at N = 262 144) and write a quarter fewer slab rows
    // (a sharded rank of the reduce-scatter protocol runs its local and its cross items as two launches with a tail
    // each: 24 per CU there too — profiles/history/r02_shard_L_sweep.log: -2.5 % at 4 ranks, neutral at 8; the single launch of
    // a replicated rank is best at 16)
    const uint32_t target = (tune.wg_per_cu ? tune.wg_per_cu : (world > 1 ? 16u : 24u)) * cus;
    // Chunks per item.  Large systems: as many items as fill the chip `target` workgroups deep.  Small ones
    // (fewer chunk-units than that): one chunk per item would be the finest grain, but every item costs a
    // 16-KiB slab row that sym_gather re-reads and a prologue, while coarse items cost tail — the optimum
    // sits near items ~ 20 sqrt(units) at 256 CUs (profiles/history/r01_force_sym_small_n_sweep.log: L = 2 at
    // N = 16 384, 3 at 25 000-32 768, 4-6 at 65 536, -5 ... -10 % step time against L = 1).
    const uint64_t units = local + cross_total / world;
    uint32_t L = tune.forced_L;
    if (!L) {
        const uint32_t fill = (uint32_t)((units + target - 1) / target);
        // (round 4, wave-split kernels and chunk pairs in place: items ~ 16 sqrt(units) with an earlier tail measures 1-2 % faster at
        // N = 25 000 ... 65 536 than the 20 sqrt(units) of round 1; profiles/history/r04_tail_sweep.log)
        // A rank of a sharded run gets 12 chunks per item at 8 ranks instead of 10 (one rank's compute share on one GPU: 94.3 ->
        // 95.5 % of the ideal in the symmetric protocol, 94.3 -> 96.4 % in the all-reduce protocol), 16 instead of 14 at 4 ranks and
        // 22 at 2 ranks (both unchanged within the noise); its tail thresholds stay (profiles/history/r04_rank_tail_sweep.log).
        const uint32_t grain = (uint32_t)(std::sqrt((double)units) * 256.0 / (16.0 * (double)cus) + 0.5);
        L = fill > grain ? fill : grain;
    }
    if (L < 1) L = 1;
    // chunk pairs: even counts; wave-split tiles: one chunk (pair) per wave, i.e. multiples of 4 (8) — a forced L is taken as given
    const uint32_t unit = tune.quantum();
    if (unit > 1 && !tune.forced_L) L = L < unit ? unit : ((L + unit / 2) / unit) * unit;
    // WAVE-SPLIT plans of whole systems (round 4; tiles of 512, N < 49 152): the finest uniform items and NO guided tail.  With one
    // chunk per wave a workgroup is one chunk time long (7.3 us with equal masses), a CU holding k of them takes k chunk times, and
    // the measured step is 9.1 + 7.3 x ceil(items / CUs) us from 7 168 to 25 000 bodies (profiles/history/r04_small_n_plans.log) — what
    // counts is the item COUNT.  Tail pieces of such items hold fewer chunks than the workgroup has waves (idle waves, nothing
    // gained) and only raise the count; coarser items (the rule above: ~16 sqrt(units)) quantise the same work into fewer, longer
    // rounds.  Against the plans of earlier in the round: -21 % per step at N = 7 168 / 10 000 / 11 000, -16 % at 13 312, -8 ... -10 %
    // at 14 336 ... 20 000, -15 % at 21 000 / 22 000, -5.5 % at 25 000 (the reference's own workload: 97.9 -> 92.3 us), -3 % at
    // 32 768 / 36 000; two sizes lose 2-4 % to a plan of 8 chunks per item (12 288, 16 384).  Beyond ~25 items per CU the count no
    // longer quantises and the per-item costs show: from there 8 chunks per item with the late tail (40 000 ... 49 151: -1 ... -2 %).
    bool no_tail = false, late_tail = false;
    // The rule was measured below SYM_WS_MAX_N bodies only, where the library itself chooses such tiles; a tile of 512 FORCED at a
    // large n (nb_params.sym_tile) keeps the size-derived L above — one or two chunks per wave there would mean ~65 000 items and
    // half a GiB of travelling partials at N = 262 144.
    if (!tune.forced_L && !tune.tail_given && world == 1 && g.sb == SYM_SB_WS && tune.guided_tail && n < SYM_WS_MAX_N) {
        uint64_t items = 0;
        for (uint32_t I = 0; I < tiles; ++I) {
            const uint32_t d0 = I * g.cpt, dend = g.diag_end(I);
            items += (dend - d0 + unit - 1) / unit + (chunks - dend + unit - 1) / unit;
        }
        if (items <= 25ull * cus) { L = unit; no_tail = true; }
        else { L = 2 * unit; late_tail = true; }
    }
    // Cross items: EVERY tile's range of later-block chunks [be, chunks) is cut into `world` contiguous sub-ranges and
    // rank r takes sub-range (r + I) mod world of tile I.  Every rank therefore holds a slice of every tile: its stationary
    // rows and travelling segments — and with them the work of its sym_gather — are spread over all tiles instead of
    // piling up on the few tiles a contiguous run of the tile-major item list covers (round 2: a rank's gather ran at
    // 1.7 TB/s on ~20 heavy tiles; profiles/history/r03_shard_p8_*).  The sub-ranges of a tile partition [be, chunks) exactly, so
    // all ranks together still meet every (tile, chunk) pair once; shares differ by at most one unit per tile, rotated
    // over the ranks.  Units are chunk pairs for handles that sweep pairs (even_chunks).
    std::vector<SymItem> local_items, cross_items, late_items;
    std::vector<uint32_t> local_rows_of(tiles, 0), cross_rows_of(tiles, 0), late_rows_of(tiles, 0);
    for (uint32_t I = 0; I < tiles; ++I) {
        const uint32_t be = g.block_end_chunk(I);
        if (g.block_of(I) == rank || world == 1) {
            const uint32_t d0 = I * g.cpt, dend = g.diag_end(I);
            for (uint32_t c = d0; c < dend; c += L) {                               // diagonal, one-sided
                local_items.push_back(SymItem{I, c, dend - c < L ? dend - c : L, 0u, 0, 1u, 0u});
                ++local_rows_of[I];
            }
            for (uint32_t c = dend; c < be; c += L) {                               // rest of the block, symmetric
                local_items.push_back(SymItem{I, c, be - c < L ? be - c : L, 0u, 0, 0u, 0u});
                ++local_rows_of[I];
            }
        }
        if (be < chunks) {                                                          // later blocks: this rank's slice
            const uint64_t len_u = (chunks - be) / unit;                            // whole units; a leftover odd chunk joins the last slice
            const uint32_t slot = (rank + I) % world;
            const uint32_t lo = be + unit * (uint32_t)(len_u * slot / world);
            const uint32_t hi = slot + 1 == world ? chunks : be + unit * (uint32_t)(len_u * (slot + 1) / world);
            for (uint32_t c = lo; c < hi; c += L) {
                cross_items.push_back(SymItem{I, c, hi - c < L ? hi - c : L, 0u, 0, 0u, 1u});
                ++cross_rows_of[I];
            }
        }
    }
    // Late items: whole items off the end of the local list, at most late_units chunk-units and at most half
    // of the local work, re-cut into late_chunks-chunk items (they run alone on the chip: fine grain, short tail).
    if (world > 1 && tune.late_units > 0) {
        const uint32_t late_chunks = tune.late_chunks ? tune.late_chunks : 1u;
        const uint64_t budget = tune.late_units < local / 2 ? tune.late_units : local / 2;
        uint64_t taken = 0;
        std::vector<SymItem> held;
        while (!local_items.empty() && taken + local_items.back().cnt <= budget) {
            taken += local_items.back().cnt;
            --local_rows_of[local_items.back().tile];
            held.push_back(local_items.back());
            local_items.pop_back();
        }
        for (auto it = held.rbegin(); it != held.rend(); ++it)
            for (uint32_t c = 0; c < it->cnt; c += late_chunks) {
                SymItem q = *it;
                q.c0 = it->c0 + c; q.cnt = it->cnt - c < late_chunks ? it->cnt - c : late_chunks;
                q.group = 2u;
                late_items.push_back(q);
                ++late_rows_of[q.tile];
            }
    }
    // Guided tail: workgroups are dispatched in item order and an item is a fixed amount of VALU work, so a
    // launch ends with up to one item time of partly idle CUs (half of it on average: 3 % of a single-GPU step,
    // 7 % of a rank's step at world = 8).  The end of each launch's work is cut into finer items
    // (L/2, L/4, L/8 chunks from 85 %, 94 %, 98 % of the work on; profiles/history/r01_guided_tail_ab.log): -2 % step time.
    // Splitting happens after the cross runs were assigned, so every rank still sees the same run boundaries.
    if (tune.guided_tail && !no_tail) {
        auto guided = [&](std::vector<SymItem> &list, std::vector<uint32_t> &rows_of) {
            uint64_t total = 0, done = 0;
            for (const auto &it : list) total += it.cnt;
            // A launch of few rounds (items per resident workgroup slot, ~4 slots per CU) ends with a larger share of its work in the
            // last, partly filled round: the finer items start earlier there (0.65 / 0.85 / 0.95 below 5 rounds instead of 0.85 / 0.94 /
            // 0.98: -9 % at N = 16 384 (with tail pieces of fewer chunks than waves), -1 % at 65 536, +-1 % elsewhere, the headline plan
            // — 6 100 items before the tail is cut, 8 187 after: 6 rounds — unchanged; profiles/history/r04_tail_sweep.log, r04_defaults_check.log).  Explicit
            // thresholds (nb_params.sym_tail) are taken as given.  Not below 1.5 rounds either: there the extra items of an early tail cost
            // more than they balance (classic tiles, fp64 / 3-D handles at 16 384 ... 24 576 bodies: the late tail is 4-9 % faster;
            // profiles/history/r04_small_n_plans_classic.log).
            double at[3] = {tune.tail_at[0], tune.tail_at[1], tune.tail_at[2]};
            if (!tune.tail_given && world == 1 && !late_tail && (double)list.size() >= 1.5 * 4.0 * (double)cus && (double)list.size() < 5.0 * 4.0 * (double)cus)
                { at[0] = 0.65; at[1] = 0.85; at[2] = 0.95; }
            std::vector<SymItem> out;
            out.reserve(list.size() * 2);
            for (const auto &it : list) {
                const double f = total ? (double)done / (double)total : 0.0;
                const uint32_t div = f < at[0] ? 1u : f < at[1] ? 2u : f < at[2] ? 4u : 8u;
                uint32_t piece = (L + div - 1) / div;
                // tail pieces of a wave-split plan may hold fewer chunks than waves (idle waves cost nothing while the launch drains;
                // balance is what the tail is for): only the evenness chunk pairs need is kept
                const uint32_t tail_unit = tune.even_chunks ? 2u : 1u;
                piece = ((piece + tail_unit - 1) / tail_unit) * tail_unit;
                done += it.cnt;
                if (div == 1 || it.cnt <= piece) { out.push_back(it); continue; }
                --rows_of[it.tile];
                for (uint32_t c = 0; c < it.cnt; c += piece) {
                    SymItem q = it;
                    q.c0 = it.c0 + c; q.cnt = it.cnt - c < piece ? it.cnt - c : piece;
                    out.push_back(q);
                    ++rows_of[it.tile];
                }
            }
            list.swap(out);
        };
        guided(local_items, local_rows_of);
        guided(cross_items, cross_rows_of);
    }
    // Slab layout.  Stationary rows: a tile's rows are contiguous (local, cross, then late items).
    // Travelling partials: one segment per (tile, group) holding exactly the particle range the group's
    // symmetric items of that tile cover (contiguous by construction: items of a tile in one list are
    // consecutive chunk slices), packed back to back.
    auto span_of = [&](const std::vector<SymItem> &list, std::vector<uint32_t> &lo, std::vector<uint32_t> &hi) {
        lo.assign(tiles, 0xffffffffu); hi.assign(tiles, 0);
        for (const auto &it : list) {
            if (it.diag) continue;
            if (it.c0 < lo[it.tile]) lo[it.tile] = it.c0;
            if (it.c0 + it.cnt > hi[it.tile]) hi[it.tile] = it.c0 + it.cnt;
        }
    };
    std::vector<uint32_t> llo, lhi, clo, chi, tlo, thi;
    span_of(local_items, llo, lhi); span_of(cross_items, clo, chi); span_of(late_items, tlo, thi);
    pl.rowbase.assign(tiles + 1, 0);
    pl.rowmid.assign(tiles, 0);
    pl.segs.clear();
    std::vector<uint32_t> next_local(tiles), next_cross(tiles), next_late(tiles);
    std::vector<int64_t> base_local(tiles, 0), base_cross(tiles, 0), base_late(tiles, 0);
    uint32_t row = 0;
    uint64_t off = 0;
    auto up = [](uint64_t v, uint32_t cap) { return v < cap ? (uint32_t)v : cap; };
    auto add_seg = [&](uint32_t I, uint32_t clo_, uint32_t chi_, uint32_t group, std::vector<int64_t> &base_of) {
        const uint32_t lo = up((uint64_t)clo_ * SYM_CH, n), hi = up((uint64_t)chi_ * SYM_CH, n);
        pl.segs.push_back(SymSeg{I, lo, hi, group, off});
        base_of[I] = (int64_t)off - (int64_t)lo;
        off += (uint64_t)(hi - lo + 1u) & ~(uint64_t)1;   // even offsets: sym_gather loads two adjacent partials at once
    };
    for (uint32_t I = 0; I < tiles; ++I) {
        pl.rowbase[I] = row;
        next_local[I] = row; row += local_rows_of[I];
        next_cross[I] = row; row += cross_rows_of[I];
        pl.rowmid[I] = row;
        next_late[I] = row; row += late_rows_of[I];
        if (lhi[I] > llo[I]) add_seg(I, llo[I], lhi[I], 0u, base_local);
        if (chi[I] > clo[I]) add_seg(I, clo[I], chi[I], 1u, base_cross);
    }
    pl.rowbase[tiles] = row;
    pl.nsegs_main = (uint32_t)pl.segs.size();
    for (uint32_t I = 0; I < tiles; ++I)
        if (thi[I] > tlo[I]) add_seg(I, tlo[I], thi[I], 2u, base_late);
    pl.slab_r_elems = off;
    pl.units_local = pl.units_cross = pl.units_late = 0;
    for (auto &it : local_items) { it.s_row = next_local[it.tile]++; it.r_base = base_local[it.tile]; it.group = 0u; pl.units_local += it.cnt; }
    for (auto &it : cross_items) { it.s_row = next_cross[it.tile]++; it.r_base = base_cross[it.tile]; it.group = 1u; pl.units_cross += it.cnt; }
    for (auto &it : late_items)  { it.s_row = next_late[it.tile]++;  it.r_base = base_late[it.tile];  it.group = 2u; pl.units_late += it.cnt; }
    pl.n_local = (uint32_t)local_items.size();
    pl.n_cross = (uint32_t)cross_items.size();
    pl.n_late = (uint32_t)late_items.size();
    pl.items = std::move(local_items);
    pl.items.insert(pl.items.end(), cross_items.begin(), cross_items.end());
    pl.items.insert(pl.items.end(), late_items.begin(), late_items.end());
    pl.L = L;
    pl.tiles = tiles;
    pl.sb = g.sb;

    // Gather lists: for every tile the segments that hold partials of (some of) its particles, in segment
    // order — sym_gather reads slab_r[base + k] for each entry with lo <= k < hi: no scan over the segments
    // of other tiles.  Whole-system plan: tile g's list is the segments of tiles 0 .. g-1.
    auto build_cov = [&](size_t s0, size_t s1, std::vector<uint32_t> &begin, std::vector<SymCov> &cov) {
        begin.assign((size_t)tiles + 1, 0);
        for (size_t s = s0; s < s1; ++s) {
            const SymSeg &sg = pl.segs[s];
            for (uint32_t t = sg.lo / g.sb; t <= (sg.hi - 1) / g.sb; ++t) ++begin[t + 1];
        }
        for (uint32_t t = 0; t < tiles; ++t) begin[t + 1] += begin[t];
        cov.assign(begin[tiles], SymCov{0, 0u, 0u});
        std::vector<uint32_t> fill(begin.begin(), begin.end() - 1);
        for (size_t s = s0; s < s1; ++s) {
            const SymSeg &sg = pl.segs[s];
            for (uint32_t t = sg.lo / g.sb; t <= (sg.hi - 1) / g.sb; ++t)
                cov[fill[t]++] = SymCov{(int64_t)sg.off - (int64_t)sg.lo, sg.lo, sg.hi};
        }
    };
    build_cov(0, pl.nsegs_main, pl.cov_main_begin, pl.cov_main);
    build_cov(pl.nsegs_main, pl.segs.size(), pl.cov_late_begin, pl.cov_late);
}

}  // namespace nbk
