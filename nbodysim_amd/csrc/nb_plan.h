// nb_plan.h — host-side work planner of the symmetric force kernels (plain C++, no HIP).
//
// The reference fans attract() over contiguous i-chunks (Nbodysim/headers/Simulation.hpp:180-208); the
// symmetric kernels instead evaluate every UNORDERED pair of particles once, so the unit of work is a
// (tile of 2048 "stationary" particles) x (run of 64-particle "travelling" chunks) item.  This file
// builds the item list of one handle (whole system, or one rank's share of a sharded run), the layout
// of the two slab sets the items write, and the per-tile coverage lists sym_gather reads them back by.
// It is compiled by the host compiler into libnbody_hip.so and, on its own, into the CPU sanitizer
// build (`make -C nbodysim_amd/csrc asan`) that fuzzes it.
#pragma once
#include <stdint.h>

#include <vector>

namespace nbk {

constexpr uint32_t SYM_SB = 2048;   // stationary particles per workgroup / block-tile (4 waves x 512 different particles)
constexpr uint32_t SYM_SB_WS = 512; // block-tile of the WAVE-SPLIT kernels (small and mid-size systems): the 4 waves of a workgroup
                                    // hold the SAME 512 stationary particles and sweep DIFFERENT chunks of the item
constexpr uint32_t SYM_WS_MAX_N = 49152; // whole systems below this many bodies get wave-split tiles by default; the planner's rules for such tiles hold below it only
constexpr uint32_t SYM_CH = 64;     // travelling chunk (one particle per lane of a wave64)

// One workgroup of force_sym_*: tile `tile` (SymPlan::sb particles) against chunks [c0, c0 + cnt).
//   s_row    row of slab_s (sb elements each) receiving the stationary partial of the item
//   r_base   element offset into slab_r such that the travelling partial of particle j goes to
//            slab_r[r_base + j] (the item's segment starts at particle seg.lo: r_base = seg.off - seg.lo,
//            which can be negative)
//   diag     1 = the chunks are the tile's own particles (one-sided, no travelling output)
//   group    0 local (pairs inside the rank's block), 1 cross-block, 2 late (held-back local)
struct SymItem { uint32_t tile, c0, cnt, s_row; int64_t r_base; uint32_t diag, group; };
static_assert(sizeof(SymItem) == 32, "SymItem is read by the kernels as 32 bytes");

// Travelling partials of one (tile, group): particles [lo, hi) stored contiguously at slab_r[off ...).
// Only the particles the tile's items really meet are stored ("triangular" slab): the whole-system
// plan needs sum_I (n - (I+1) 2048) ~ tiles x n / 2 elements, not tiles x n.
struct SymSeg { uint32_t tile, lo, hi, group; uint64_t off; };

// Coverage entry of a tile's gather list: particles k in [lo, hi) read slab_r[base + k].
struct SymCov { int64_t base; uint32_t lo, hi; };
static_assert(sizeof(SymCov) == 16, "SymCov is read by the kernels as 16 bytes");

struct SymTuning {
    uint32_t forced_L = 0;        // chunks per item; 0 = automatic
    uint32_t wg_per_cu = 0;       // workgroups per CU the automatic choice aims at; 0 = 24 (one launch per step) / 16
    uint32_t late_units = 0;      // chunk-units of local work held back for the side stream (sharded ranks)
    uint32_t late_chunks = 2;     // chunks per late item
    bool guided_tail = true;      // finer items at the end of each launch
    bool even_chunks = false;     // cut items into EVEN chunk counts: the fp32 2-D kernel sweeps chunk PAIRS (sym_chunks2), and
                                  // an odd item wastes half a pair
    uint32_t sb = SYM_SB;         // particles per block-tile: SYM_SB, or SYM_SB_WS for the wave-split kernels, whose items are
                                  // cut into multiples of 4 chunks (one per wave; 8 with chunk pairs) — see quantum()
    uint32_t quantum() const { return (sb == SYM_SB_WS ? 4u : 1u) * (even_chunks ? 2u : 1u); }
    double tail_at[3] = {0.85, 0.94, 0.98};
    bool tail_given = false;      // tail_at was set by the caller (nb_params.sym_tail): no size-dependent adjustment
};

struct SymPlan {
    std::vector<SymItem> items;                 // [local | cross | late]
    uint32_t n_local = 0, n_cross = 0, n_late = 0, L = 0, tiles = 0, sb = SYM_SB;
    std::vector<uint32_t> rowbase, rowmid;      // stationary rows of tile g: [rowbase[g], rowmid[g]) local + cross,
                                                // [rowmid[g], rowbase[g + 1]) late
    std::vector<SymSeg> segs;                   // nsegs_main segments of local + cross first, then the late ones
    uint32_t nsegs_main = 0;
    uint64_t slab_r_elems = 0;                  // total elements of slab_r
    // gather lists (CSR over tiles): entries cov_*[begin[g] .. begin[g + 1]) are the segments meeting tile g
    std::vector<uint32_t> cov_main_begin, cov_late_begin;
    std::vector<SymCov> cov_main, cov_late;
    uint64_t units_local = 0, units_cross = 0, units_late = 0;   // chunk-units (tile x chunk) of each group
};

// Rank-independent size figures of a symmetric plan (every rank of a run must take the same decisions).
//   local_sym_units   (tile, chunk) units of one block's internal symmetric items
//   cross_total       (tile, chunk) units between different blocks, all ranks together
void sym_units(uint32_t n, uint32_t world, uint64_t *local_sym_units, uint64_t *cross_total, uint32_t sb = SYM_SB);

// Upper bound, equal on all ranks, of the slab_r elements one handle of the run needs.
uint64_t sym_slab_r_bound(uint32_t n, uint32_t world, uint32_t sb = SYM_SB);

void build_sym_plan(uint32_t n, uint32_t cus, uint32_t rank, uint32_t world, const SymTuning &tune, SymPlan &pl);

}  // namespace nbk
