// nb_capi.hip — the C ABI of include/nbody.h over the gfx950 kernels.
//
// One nb_sim stands for one `Simulation` of the reference
// (Nbodysim/headers/Simulation.hpp:49-75).  Device state is SoA:
//   pos[2]   full-n (x,y) replicas, double buffered (step n reads cur, writes next)
//   mass     full n            radius  full n (carried, float)
//   vel,acc  owned block only  partial [slabs][i_count] force partial sums
// No CPU fallback exists: every entry that computes needs a HIP device.
#include "nbody.h"
#include "nbody_debug.h"
#include "nb_internal.h"
#include "nb_kernels.hip.h"
#include "nb_kernels3d.hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include <unistd.h>

using namespace nbk;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
// (the thread-local error text / code and nb_params_default live in nb_host.c: plain C, shared with the CPU-only build)
static int hip_code(hipError_t e) { return e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation ? NB_ENOMEM : NB_EHIP; }

#define HIPCHK(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess)                                                         \
            return nb_fail(hip_code(e_), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" int nb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// ---------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------
// j_begin/j_end are virtual indices that skip [gap_begin, gap_begin + gap_len)
struct ForceJob { uint32_t j_begin, j_end, js, slab0; int P; uint32_t i_tiles; uint32_t gap_begin, gap_len; };

constexpr int F32_WS = 4;   // waves of a workgroup sharing one i-set in force_tiled_f32 (in-workgroup j-split)

struct nb_sim {
    nb_params p;
    size_t n = 0, i_begin = 0, i_count = 0;
    int dev = 0;
    int cus = 256;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool fp64 = false;
    bool dims3 = false;        // 3-D variant: real4 {x,y,z,m} positions, real4 velocities / accelerations / slabs
    size_t rsz = 4;            // sizeof(real)
    size_t esz = 8;            // bytes of one position / velocity / acceleration / slab element

    void *pos[2] = {nullptr, nullptr};
    bool own_pos = true;
    size_t pos_rows = 0;       // rows each replica holds (>= n: padded to world * ceil(n / world) for a sharded handle)
    int cur = 0;
    void *mass = nullptr;
    float *radius = nullptr;
    void *vel = nullptr, *acc = nullptr;
    void *partial = nullptr;
    uint32_t slabs_cap = 0;
    BodyRec *aos_dev = nullptr;     // n records (upload) / i_count records (sync)
    void *staging = nullptr;        // pinned host, i_count * 64 B
    void *bounce = nullptr;         // pinned host bounce ring (BOUNCE_SLOTS x BOUNCE_SLOT_BYTES): pageable caller memory never reaches HIP
    hipEvent_t ev_bounce[4] = {nullptr, nullptr, nullptr, nullptr};   // one per slot / per staged piece
    double *ered_dev = nullptr;     // energy partials
    double *pred_dev = nullptr;     // momentum partials (nb_momentum), allocated on first use
    size_t ered_blocks = 0;

    // launch geometry (per job: particles per lane and j-slices)
    ForceJob job_all{}, job_local{}, job_remote{};
    uint32_t slabs_all = 0, slabs_two_phase = 0;

    uint64_t frame = 0;
    float pending_dt = 0.f;
    bool in_step = false;
    bool mid_done = false;          // symmetric sharded protocol: nb_step_mid has run for the step in flight
    bool acc_valid = false;         // KDK: acc holds a(x_cur)
    bool uniform_mass = false;      // every body has the same mass: the per-pair mass multiply is hoisted
    float um_mass = 0.f;
    bool sym_pairs = false;         // symmetric fp32 kernel sweeps chunk pairs (want_pairs)
    bool mass_scaled = false;       // individual masses folded into the pair geometry (MM_SCALED, nb_kernels.hip.h)
    float *sigma = nullptr;         // m^(-1/2) per particle, for mass_scaled
    float mass_scaling_dev = -1.0f; // what the upload-time check measured: max |a_scaled - a_general| / max |a_general| (-1: not measured)

    // symmetric path (force_sym_f32): work items and its two slab sets
    bool sym = false;
    uint32_t sym_items = 0, sym_items_local = 0, sym_items_cross = 0, sym_items_late = 0;   // [local | cross | late]
    uint32_t sym_tiles = 0, sym_rows = 0, sym_L = 0, sym_cov_late_off = 0;
    uint32_t sym_sb = SYM_SB, sym_sb_shift = 11;   // particles per block-tile of the plan: 2048 (classic) or 512 (wave-split kernels)
    SymItem *sym_items_dev = nullptr;          // local items first, then the cross-block items
    uint32_t *sym_rowbase_dev = nullptr;       // 3 x tiles: first row / first late row / end row of every tile
    uint32_t *sym_cov_begin_dev = nullptr;     // 2 x (tiles + 1): coverage-list bounds of the main and the late gather
    SymCov *sym_cov_dev = nullptr;             // coverage entries: main lists, then (from sym_cov_late_off) the late ones
    nb_sym_info sym_info{};
    void *sym_slab_s = nullptr, *sym_slab_r = nullptr;       // float2 / double2 by precision
    bool broken = false;                       // a force launch was refused by the runtime: every later step returns NB_ESTATE
    // dynamic item tickets of the whole-system symmetric launch (sym_item_index, nb_kernels.hip.h)
    uint32_t *sym_ticket = nullptr;            // device: one counter on a line of its own, monotonic modulo 2^32
    uint32_t sym_ticket_base[3] = {0, 0, 0};   // what the launches so far have drawn, per launch kind (local or whole | cross | late: one counter each,
                                               // 128 bytes apart — a sharded rank's launches may run side by side)
    uint32_t sym_first_wave = 0;               // workgroups that keep their static item (the resident slots of the kernel variant); 0 = not yet known
    bool sym_first_wave_uniform = false, sym_first_wave_scaled = false;   // the mass model of the instantiation it was asked for (do_upload resets it on a change)
    // symmetric SHARDED protocol: this rank holds the items of the tiles dealt to it
    bool sym_sharded = false;
    // symmetric REPLICATED protocol (NB_FLAG_SHARD_ALLREDUCE): the handle holds this rank's share of the pairs like a
    // sharded one, but integrates ALL n particles itself after the host has all-reduced the partial accelerations:
    // one collective per step, every rank keeps the whole (bit-identical) state
    bool sym_replicated = false;
    void *acc_full = nullptr, *acc_owned = nullptr;     // reduce-scatter input (n) / output (i_count), (ax,ay) reals
    bool own_acc = true;
    // the local items run on a side stream so that their tail and the head of the cross items share the chip
    // and the late items run there while the reduce-scatter is in flight
    // pipelined snapshot (nb_snapshot_begin / _wait): D2H on its own stream, beside the steps that follow
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_copied = nullptr;
    nb_body *snap_out = nullptr;
    bool snap_direct = false, snap_pending = false;
    hipStream_t aux = nullptr;
    bool aux_local = false;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_late = nullptr;
    hipEvent_t ev_x[2] = {nullptr, nullptr};   // in-process exchanges: fences between the handles' streams
    uint64_t peers_enabled = 0;                // devices whose memory this handle's device has mapped

    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool, ev_used;
    std::vector<uint32_t> ev_weight;           // force passes each used event pair brackets (a pipeline launch: its steps)
    double prof_ms = 0.0;
    uint64_t prof_launches = 0;
};

static int bind(const nb_sim *s)
{
    int d = -1;
    HIPCHK(hipGetDevice(&d));
    if (d != s->dev) HIPCHK(hipSetDevice(s->dev));
    return NB_OK;
}

// Launch geometry of one one-sided force job.  The kernel is VALU-bound, so what
// matters is (a) enough independent work per lane — 2P particles per lane, P = 4 measured best — and
// (b) enough workgroups in flight to keep 5-8 waves per SIMD issuing and to even out the tail: about
// 32 workgroups per CU (profiles/history/r01_force_tiled_geometry_sweep.log).  When i-particles are scarce (sharded or small runs)
// the j range is cut into more tile-aligned slices (at most 128: every slice is a slab that
// `integrate` re-reads), and P drops only when even that cannot fill half the target.
static bool want_sym(const nb_sim *s);

// Slices are whole LDS tiles and none is empty: js = ceil(tiles / ceil(tiles / want)).
static uint32_t even_slices(uint32_t jn, uint32_t want)
{
    const uint32_t tiles = (jn + TJ - 1) / TJ;
    if (want < 1) want = 1;
    if (want > tiles) want = tiles;
    const uint32_t per = (tiles + want - 1) / want;
    return (tiles + per - 1) / per;
}

static ForceJob plan_job(const nb_sim *s, uint32_t jb, uint32_t je, uint32_t slab0,
                         uint32_t gap_begin = 0xffffffffu, uint32_t gap_len = 0)
{
    ForceJob j{jb, je, 0, slab0, 1, 0, gap_begin, gap_len};
    const uint32_t ic = (uint32_t)s->i_count;
    if (je <= jb) return j;
    const uint32_t jn = je - jb;
    if (s->p.sum_order == NB_SUM_SEQUENTIAL) { j.js = 1; j.P = 1; j.i_tiles = (ic + BLOCK - 1) / BLOCK; return j; }
    // fp32: the 4 waves of a workgroup share 64 i-lanes and split each j-tile (WS = 4), a lane owns 2P
    // particles -> 128P particles per workgroup; fp64: 256 i-lanes, P particles per lane.
    const uint32_t lanes_i = s->fp64 ? 1u : 2u;                 // particles per lane per P
    const uint32_t ilanes = s->fp64 ? (uint32_t)BLOCK : (uint32_t)BLOCK / F32_WS;
    const uint32_t target = 32u * (uint32_t)s->cus;             // workgroups wanted in the grid (profiles/history/r01_force_tiled_geometry_sweep.log)
    const uint32_t max_slices = 128;                            // bounds the slab traffic of `integrate`
    const uint32_t tiles = (jn + TJ - 1) / TJ;
    const int forced_p = s->p.lanes_p > 0 ? s->p.lanes_p : 0;
    const int pmax = s->fp64 ? 2 : 4;
    // Largest P (most independent chains per lane, fewest LDS reads per pair) that still yields
    // at least half the wanted workgroups; P halves only when both i and j are scarce.
    for (int P = pmax; P >= 1; P >>= 1) {
        if (forced_p > 0 && P != (forced_p > pmax ? pmax : forced_p)) continue;
        const uint32_t i_tiles = (ic + ilanes * lanes_i * P - 1) / (ilanes * lanes_i * P);
        uint32_t want = (target + i_tiles - 1) / i_tiles;
        if (want > max_slices) want = max_slices;
        if (s->p.j_slices > 0) want = (uint32_t)s->p.j_slices;
        j.P = P;
        j.i_tiles = i_tiles;
        j.js = even_slices(jn, want);
        if (forced_p > 0 || s->p.j_slices > 0 || P == 1) break;
        const uint32_t reach = i_tiles * (tiles < max_slices ? tiles : max_slices);   // most workgroups this P can give
        if (reach >= target / 2) break;
    }
    return j;
}

static void plan(nb_sim *s)
{
    const uint32_t n = (uint32_t)s->n, ib = (uint32_t)s->i_begin, ic = (uint32_t)s->i_count;
    s->job_all = plan_job(s, 0, n, 0);
    s->slabs_all = s->sym ? 1 : s->job_all.js;           // the symmetric path leaves one summed slab
    s->job_local = plan_job(s, ib, ib + ic, 0);
    // everything but the owned block, in one launch: virtual j range [0, n - ic) with a gap at the block
    s->job_remote = plan_job(s, 0, n - ic, s->job_local.js, ib, ic);
    s->slabs_two_phase = s->job_local.js + s->job_remote.js;
}

// Symmetric path: tiled runs with eps > 0 that are big enough to fill the chip with (tile, chunk-range)
// items — either the whole system on one GPU, or (shard_world > 1) this rank's share of the pairs of a
// sharded run.  NB_FLAG_NO_SYMMETRY forces the one-sided kernels.
// The branch-free pair body lets a particle meet itself (and coincident particles meet): r = 0 gives
// 0 x (eps^2)^(-3/2), which is exactly 0 only while (1/eps)^3 is finite — in fp32 down to eps ~ 1.5e-13.
// Below that (and for eps = 0) the one-sided kernels keep the reference's `if (r_sq > 0)` guard (Quadtree.hpp:139).
static bool needs_guard(const nb_sim *s) { return s->fp64 ? !(s->p.eps > 0.0f) : !(s->p.eps >= 1e-12f); }

// Block-tile of a handle's symmetric plan.  The WAVE-SPLIT kernels (force_sym_f32<..., WS>: tiles of 512, the 4 waves of a
// workgroup share the stationary particles and split the chunks) give the planner work units a quarter the size, items
// whose stationary row is 4 KiB instead of 16, and a sweep without barriers — what small and mid-size systems need to
// fill 1024 resident workgroup slots evenly — for four times the travelling partials per pair, which large systems do not
// pay back.  Measured (profiles/history/r04_ws_sweep.log): -3 ... -4 % step time at N = 25 000 (reference
// workload), -2.5 % at 16 384 and 32 768, neutral at 65 536, +1.3 ... +2.6 % at 131 072: used below 49 152 bodies.
// fp32 2-D, single handle (a rank of a sharded run keeps the classic tiles: its blocks are whole 2048-particle tiles).
// nb_params.sym_tile = 512 / 2048 forces one.  Rank-independent (n and parameters only).
static uint32_t sym_tile_of(const nb_params &p, size_t n)
{
    if (p.precision == NB_FP64 || p.dims == 3 || p.shard_world > 1 || (p.flags & NB_FLAG_SHARD_SINGLE)) return SYM_SB;
    if (p.sym_tile) return (uint32_t)p.sym_tile;
    return n < SYM_WS_MAX_N ? SYM_SB_WS : SYM_SB;
}

static bool sym_eligible(const nb_sim *s)
{
    if (s->p.flags & NB_FLAG_NO_SYMMETRY) return false;
    if (s->p.sum_order != NB_SUM_TILED || needs_guard(s)) return false;
    if (s->fp64 && s->p.rsqrt_mode != NB_RSQRT_EXACT) return false;
    if (s->p.integrator != NB_INTEGRATOR_KICK_DRIFT && s->i_count != s->n) return false;
    // smallest system worth the symmetric scheme: 16 384 bodies with the classic tiles (rounds 1-3); with the wave-split tiles and
    // their uniform one-chunk-per-wave plans it overtakes the one-sided kernel from ~5 600 bodies on (-4 ... -8 % at 5 632, -14 ... -17 %
    // at 6 144, -24 ... -28 % at 7 168, -32 ... -34 % at 10 000; +1 ... +6 % at 5 120, +25 % at 4 096:
    // profiles/history/r04_small_n_plans.log)
    if (s->n < (sym_tile_of(s->p, s->n) == SYM_SB_WS ? (size_t)5632 : 8 * (size_t)SYM_SB)) return false;
    const uint32_t world = s->p.shard_world > 1 ? (uint32_t)s->p.shard_world : 1u;
    // Travelling partials: one element per (tile, later particle) pair the handle evaluates — tiles x n / 2 for a
    // whole system (1 GiB at N = 524 288 fp32, 2 GiB at 1 048 576, 32 GiB at 4 194 304), 1/world of that for a rank.
    // Sized for the 288 GB of an MI355X: up to 96 GiB and a third of what is free now (N ~ 7 million fp32 is the
    // largest symmetric run; beyond that the one-sided kernel).  Ranks of a sharded run must all take the same
    // decision (it selects the exchange protocol), so there the limit does not look at this device's free memory
    // and uses the rank-independent bound; an allocation that does not fit fails nb_create loudly (NB_ENOMEM).
    size_t cap = (size_t)96 << 30;
    if (world == 1) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)24 << 30; }
        if (free_b / 3 < cap) cap = free_b / 3;
    }
    if (sym_slab_r_bound((uint32_t)s->n, world, sym_tile_of(s->p, s->n)) * s->esz > cap) return false;
    return true;
}

// NB_FLAG_SHARD_SINGLE: the sharded protocols with one rank (shard_world = 1, the handle owns everything): the rank's
// "share" is every pair, all of them local; the host-side exchange degenerates to copies.  Lets a one-GPU box run the
// whole split-step + collective path (nb_comm_*, nb_exchange_*).
static bool single_rank(const nb_sim *s)
{
    return (s->p.flags & NB_FLAG_SHARD_SINGLE) && s->p.shard_world == 1 && s->p.shard_rank == 0 && s->i_count == s->n;
}

static bool want_sym(const nb_sim *s)          // single handle owns everything
{
    return s->i_count == s->n && s->p.shard_world <= 1 && !single_rank(s) && sym_eligible(s);
}

static bool want_sym_replicated(const nb_sim *s)  // rank of a run that all-reduces accelerations and integrates everything everywhere
{
    const size_t w = (size_t)s->p.shard_world;
    if (!(s->p.flags & NB_FLAG_SHARD_ALLREDUCE) || s->i_count != s->n || s->p.integrator != NB_INTEGRATOR_KICK_DRIFT || !sym_eligible(s)) return false;
    if (single_rank(s)) return true;
    return s->p.shard_world > 1 && s->n / w >= 2 * (size_t)SYM_SB && s->n % (w * SYM_SB) == 0;
}

static bool want_sym_sharded(const nb_sim *s)  // rank of a sharded run
{
    const size_t w = (size_t)s->p.shard_world;
    if (single_rank(s)) return !(s->p.flags & NB_FLAG_SHARD_ALLREDUCE) && s->p.integrator == NB_INTEGRATOR_KICK_DRIFT && sym_eligible(s);
    return s->p.shard_world > 1 && s->i_count != s->n && sym_eligible(s) && s->n / w >= 2 * (size_t)SYM_SB &&
           s->n % (w * SYM_SB) == 0 && s->i_count == s->n / w && s->i_begin == (size_t)s->p.shard_rank * s->i_count;   // equal blocks of whole tiles
}

// Chunk-units of local work a rank holds back to run beside the reduce-scatter: 40 us of whole-chip work at the
// measured 35 units/us (fp32) or 14 (fp64) of 256 CUs.  Alone on the chip those items take 50-60 us and the
// hand-over between the streams ~15 us, so the split pays when the collective is exposed for longer than that
// share of a step: from 8 ranks on (a rank's step at N = 262 144 is ~1 ms there; at 2-4 ranks it measured
// neutral to -2 %, profiles/history/r01_late_items_ab.log).  nb_params.sym_late_us > 0 forces it for any world size,
// < 0 disables it.
static uint32_t late_units_for(const nb_params &p, bool fp64, int cus, uint32_t world)
{
    const double us = p.sym_late_us > 0.0f ? (double)p.sym_late_us : p.sym_late_us < 0.0f ? 0.0 : (world >= 8 ? 40.0 : 0.0);
    if (!(us > 0.0)) return 0;
    return (uint32_t)(us * (fp64 ? 14.0 : 35.0) * (double)cus / 256.0);
}

// Chunk PAIRS (sym_chunks2: two travelling particles per lane) pay from ~65 536 bodies on (-2 ... -3 % at 131 072 - 262 144,
// neutral at 65 536; below that the coarser items and the lower occupancy — 146-158 VGPRs, 3 waves per SIMD — cost more
// than the saved rotations: profiles/history/r03_chunk_pairs_sweep.log).  fp32 2-D only.  nb_params.sym_chunk_pairs = 1 / -1 forces it.
// Rank-independent (n and parameters only): it shapes the plan every rank must agree on.
static bool want_pairs(const nb_params &p, size_t n)
{
    if (p.precision == NB_FP64) return false;
    if (p.sym_chunk_pairs) return p.sym_chunk_pairs > 0;
    return n >= 65536;                           // 2-D and 3-D (3-D with individual masses keeps the single-chunk kernel: launch_sym_items)
}

static SymTuning tuning_of(const nb_params &p, bool fp64, int cus, uint32_t world, bool sharded, size_t n)
{
    SymTuning t;
    t.forced_L = p.sym_chunks_per_item > 0 ? (uint32_t)p.sym_chunks_per_item : 0u;
    t.late_units = sharded ? late_units_for(p, fp64, cus, world) : 0u;
    t.wg_per_cu = sharded ? 24u : 0u;             // reduce-scatter protocol: two launches per step (nb_plan.cpp)
    t.late_chunks = fp64 ? 1u : 2u;
    t.even_chunks = want_pairs(p, n);            // the kernel sweeps chunk pairs: even chunk counts
    t.sb = sym_tile_of(p, n);
    t.guided_tail = !(p.flags & NB_FLAG_NO_GUIDED_TAIL);
    if (p.sym_tail[0] > 0.0f || p.sym_tail[1] > 0.0f || p.sym_tail[2] > 0.0f)
        { for (int k = 0; k < 3; ++k) t.tail_at[k] = (double)p.sym_tail[k]; t.tail_given = true; }
    return t;
}

static void fill_sym_info(const SymPlan &pl, uint32_t n, uint32_t world, int cus, size_t esz, bool enabled, nb_sym_info *out)
{
    const uint32_t sz = out->struct_size;
    memset(out, 0, sizeof *out);
    out->struct_size = sz;
    out->enabled = enabled ? 1 : 0;
    out->chunks_per_item = pl.L;
    out->items = (uint32_t)pl.items.size();
    out->items_local = pl.n_local; out->items_cross = pl.n_cross; out->items_late = pl.n_late;
    out->tiles = pl.tiles;
    out->rows_s = pl.rowbase.empty() ? 0u : pl.rowbase[pl.tiles];
    out->segments = (uint32_t)pl.segs.size();
    out->cus = (uint32_t)cus;
    out->units_local = pl.units_local; out->units_cross = pl.units_cross; out->units_late = pl.units_late;
    sym_units(n, world, nullptr, &out->cross_units_total, pl.sb);
    out->slab_s_bytes = (uint64_t)out->rows_s * pl.sb * esz;
    out->tile_particles = pl.sb;
    out->slab_r_bytes = pl.slab_r_elems * esz;
    out->coverage_entries = pl.cov_main.size() + pl.cov_late.size();
}

#ifdef NB_TEST_HOOKS   /* include/nbody_debug.h: test build only */
// CPU-testable view of the planner (no device call).
extern "C" int nb_debug_sym_plan(size_t n, int cus, int rank, int world, const nb_params *tuning,
                                 nb_sym_item *items_out, size_t cap, nb_sym_info *info)
{
    static_assert(sizeof(nb_sym_item) == sizeof(SymItem) && offsetof(nb_sym_item, r_base) == offsetof(SymItem, r_base) &&
                  offsetof(nb_sym_item, group) == offsetof(SymItem, group), "nb_sym_item mirrors SymItem");
    if (n == 0 || n > 0x7fffff00u || cus < 1 || world < 1 || rank < 0 || rank >= world ||
        (world > 1 && (n % ((size_t)world * SYM_SB)) != 0)) return nb_fail(NB_EINVAL, "nb_debug_sym_plan: bad arguments");
    if (tuning && tuning->struct_size != sizeof(nb_params)) return nb_fail(NB_EINVAL, "nb_debug_sym_plan: tuning->struct_size");
    if (info && info->struct_size != sizeof(nb_sym_info)) return nb_fail(NB_EINVAL, "nb_debug_sym_plan: info->struct_size");
    nb_params p;
    if (tuning) p = *tuning; else nb_params_default(&p);
    p.shard_rank = rank; p.shard_world = world;        // what a handle of this (rank, world) would carry: selects the tile size too
    const bool fp64 = p.precision == NB_FP64;
    SymPlan pl;
    build_sym_plan((uint32_t)n, (uint32_t)cus, (uint32_t)rank, (uint32_t)world, tuning_of(p, fp64, cus, (uint32_t)world, world > 1, n), pl);
    if (info) fill_sym_info(pl, (uint32_t)n, (uint32_t)world, cus, fp64 ? 16 : 8, true, info);
    if (items_out) memcpy(items_out, pl.items.data(), (pl.items.size() < cap ? pl.items.size() : cap) * sizeof(SymItem));
    return NB_OK;
}
#endif

// ---------------------------------------------------------------------------
// host memory
// ---------------------------------------------------------------------------
// The copy engine is only ever pointed at host memory this library KNOWS to be page-locked over the whole transfer:
// its own buffers (hipHostMalloc) and the ranges in the registry below — whole pages registered through
// nb_host_register, or blocks handed out by nb_host_alloc.  Everything else (caller arrays, std::vector storage,
// numpy arrays) moves through the handle's page-locked bounce buffer in pipelined chunks.
// Why so strict: hipHostRegister / the runtime's own pinning of pageable copy sources work on whole
// PAGES; a malloc'ed array shares its first and last page with whatever the heap put next to it, and a registration
// that outlives the array (a std::vector that reallocated) keeps pinning pages that now belong to someone else.
// Round 2 saw one process abort inside nb_upload right after nb_host_register of an unaligned numpy array.  CAUSE
// UNKNOWN: no log of that run exists, and none of the nine deterministic constructions of round 3's pin probe (shared
// pages, overlapping registrations, freed-while-registered memory reused at the same address ...) aborts on this
// runtime (profiles/history/r03_pin_probe.log).  The rules above are therefore a DEFENSIVE change, not the fix of a known bug.
struct PinnedRange { uintptr_t lo, hi; bool owned; };
static std::mutex g_pin_mutex;
static std::vector<PinnedRange> g_pinned;

static bool pinned_covers(const void *p, size_t bytes)
{
    const uintptr_t a = (uintptr_t)p, b = a + bytes;
    std::lock_guard<std::mutex> lock(g_pin_mutex);
    for (const PinnedRange &r : g_pinned)
        if (a >= r.lo && b <= r.hi && b >= a) return true;
    return false;
}

static size_t host_page_size()
{
    const long ps = sysconf(_SC_PAGESIZE);
    return ps > 0 ? (size_t)ps : 4096;
}

constexpr size_t BOUNCE_SLOTS = 4;
constexpr size_t BOUNCE_SLOT_BYTES = (size_t)2 << 20;       // 4 x 2 MiB: a DMA, a host memcpy and two slots of slack in flight
constexpr size_t BOUNCE_BYTES = BOUNCE_SLOTS * BOUNCE_SLOT_BYTES;
#ifndef NB_SMALL_SYNC_BYTES
#define NB_SMALL_SYNC_BYTES ((size_t)4 << 20)           // (A/B builds: -DNB_SMALL_SYNC_BYTES=0 keeps the copy-engine path for every size)
#endif
constexpr size_t SMALL_SYNC_BYTES = NB_SMALL_SYNC_BYTES;   // nb_sync: below this the pack kernel writes the host staging buffer itself (no DMA)

static int ensure_bounce(nb_sim *s)
{
    if (!s->bounce) HIPCHK(hipHostMalloc(&s->bounce, BOUNCE_BYTES, hipHostMallocDefault));
    for (size_t k = 0; k < BOUNCE_SLOTS; ++k)
        if (!s->ev_bounce[k]) HIPCHK(hipEventCreateWithFlags(&s->ev_bounce[k], hipEventDisableTiming));
    return NB_OK;
}

// Host -> device on `st`; returns when the data has left `src` (the caller may free it).  Pageable sources are
// copied slot by slot into the bounce buffer; slot k is reused only after its DMA (event) is complete, so the
// host memcpy of chunk c + 1 overlaps the DMA of chunk c.
static int copy_h2d(nb_sim *s, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return NB_OK;
    if (pinned_covers(src, bytes)) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        return NB_OK;
    }
    { const int rc = ensure_bounce(s); if (rc) return rc; }
    size_t c = 0;
    for (size_t off = 0; off < bytes; off += BOUNCE_SLOT_BYTES, ++c) {
        const size_t k = c % BOUNCE_SLOTS, len = bytes - off < BOUNCE_SLOT_BYTES ? bytes - off : BOUNCE_SLOT_BYTES;
        char *slot = (char *)s->bounce + k * BOUNCE_SLOT_BYTES;
        if (c >= BOUNCE_SLOTS) HIPCHK(hipEventSynchronize(s->ev_bounce[k]));
        memcpy(slot, (const char *)src + off, len);
        HIPCHK(hipMemcpyAsync((char *)dst + off, slot, len, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipEventRecord(s->ev_bounce[k], s->stream));
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    return NB_OK;
}

// Device -> host on `st` (ordered after the work enqueued there), blocking.  Pageable destinations: all chunk DMAs
// are enqueued at once into the slots of `stage` (page-locked, >= bytes, or the bounce ring when stage is NULL) and
// the host copies chunk c out while chunk c + 1 is still in flight.
static int copy_d2h(nb_sim *s, void *dst, const void *src, size_t bytes, hipStream_t st = nullptr, void *stage = nullptr)
{
    if (bytes == 0) return NB_OK;
    if (!st) st = s->stream;
    if (pinned_covers(dst, bytes)) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return NB_OK;
    }
    { const int rc = ensure_bounce(s); if (rc) return rc; }
    if (stage) {
        // whole transfer staged: BOUNCE_SLOTS equal pieces, one event each — one piece where the transfer is so short (<= 1 MiB: the
        // positions of a small system, nb_sync_positions) that three more DMA launches and events cost more than the overlap buys
        const size_t pieces = bytes <= ((size_t)1 << 20) ? 1 : BOUNCE_SLOTS;
        const size_t piece = ((bytes + pieces - 1) / pieces + 63) & ~(size_t)63;
        size_t k = 0;
        for (size_t off = 0; off < bytes; off += piece, ++k) {
            const size_t len = bytes - off < piece ? bytes - off : piece;
            HIPCHK(hipMemcpyAsync((char *)stage + off, (const char *)src + off, len, hipMemcpyDeviceToHost, st));
            HIPCHK(hipEventRecord(s->ev_bounce[k], st));
        }
        // Copy-out on the calling thread, piece k as soon as its DMA has landed (the copy of piece k overlaps the DMA of piece k + 1).
        // Round 4 tried helper threads for pieces 1 .. 3 (9.68 -> 9.31 ms per frame of the reference caller's loop at N = 262 144):
        // dropped — after threads that had touched HIP exited, RCCL's ncclCommInitAll failed in the same process ("unhandled cuda
        // error", tests/test_host_gpu.py, twice in two runs); a caller that wants the 0.35 ms keeps `bodies` in nb_host_alloc memory
        // (direct DMA) or uses the pipelined snapshot.
        k = 0;
        for (size_t off = 0; off < bytes; off += piece, ++k) {
            const size_t len = bytes - off < piece ? bytes - off : piece;
            HIPCHK(hipEventSynchronize(s->ev_bounce[k]));
            memcpy((char *)dst + off, (const char *)stage + off, len);
        }
        return NB_OK;
    }
    size_t issued = 0, done = 0;
    const size_t chunks = (bytes + BOUNCE_SLOT_BYTES - 1) / BOUNCE_SLOT_BYTES;
    while (done < chunks) {
        while (issued < chunks && issued < done + BOUNCE_SLOTS) {
            const size_t off = issued * BOUNCE_SLOT_BYTES, len = bytes - off < BOUNCE_SLOT_BYTES ? bytes - off : BOUNCE_SLOT_BYTES;
            const size_t k = issued % BOUNCE_SLOTS;
            HIPCHK(hipMemcpyAsync((char *)s->bounce + k * BOUNCE_SLOT_BYTES, (const char *)src + off, len, hipMemcpyDeviceToHost, st));
            HIPCHK(hipEventRecord(s->ev_bounce[k], st));
            ++issued;
        }
        const size_t off = done * BOUNCE_SLOT_BYTES, len = bytes - off < BOUNCE_SLOT_BYTES ? bytes - off : BOUNCE_SLOT_BYTES;
        const size_t k = done % BOUNCE_SLOTS;
        HIPCHK(hipEventSynchronize(s->ev_bounce[k]));
        memcpy((char *)dst + off, (const char *)s->bounce + k * BOUNCE_SLOT_BYTES, len);
        ++done;
    }
    return NB_OK;
}

static int plan_sym(nb_sim *s)
{
    const uint32_t n = (uint32_t)s->n;
    const bool split = s->sym_sharded || s->sym_replicated;
    const uint32_t world = split ? (uint32_t)s->p.shard_world : 1u;
    const uint32_t rank = split ? (uint32_t)s->p.shard_rank : 0u;
    SymPlan pl;
    build_sym_plan(n, (uint32_t)s->cus, rank, world, tuning_of(s->p, s->fp64, s->cus, world, s->sym_sharded, s->n), pl);   // late items: sharded only
    s->sym_pairs = want_pairs(s->p, s->n);
    const uint32_t tiles = pl.tiles, row = pl.rowbase[tiles];
    s->sym_info.struct_size = (uint32_t)sizeof(nb_sym_info);
    fill_sym_info(pl, n, world, s->cus, s->esz, true, &s->sym_info);
    s->sym_items_local = pl.n_local; s->sym_items_cross = pl.n_cross; s->sym_items_late = pl.n_late;
    s->sym_items = (uint32_t)pl.items.size(); s->sym_tiles = tiles; s->sym_rows = row; s->sym_L = pl.L;
    s->sym_sb = pl.sb; s->sym_sb_shift = pl.sb == SYM_SB_WS ? 9u : 11u;
    // row bounds for the gathers: [lo | mid | hi] = rowbase[0..tiles), rowmid[0..tiles), rowbase[1..tiles]
    std::vector<uint32_t> bounds(3 * (size_t)tiles);
    for (uint32_t g = 0; g < tiles; ++g) { bounds[g] = pl.rowbase[g]; bounds[tiles + g] = pl.rowmid[g]; bounds[2 * (size_t)tiles + g] = pl.rowbase[g + 1]; }
    // coverage lists of the main and the late gather: [begin_main (tiles + 1) | begin_late (tiles + 1)], entries of both back to back
    std::vector<uint32_t> cbegin(pl.cov_main_begin);
    cbegin.insert(cbegin.end(), pl.cov_late_begin.begin(), pl.cov_late_begin.end());
    std::vector<SymCov> cov(pl.cov_main);
    cov.insert(cov.end(), pl.cov_late.begin(), pl.cov_late.end());
    s->sym_cov_late_off = (uint32_t)pl.cov_main.size();
    HIPCHK(hipMalloc((void **)&s->sym_items_dev, pl.items.size() * sizeof(SymItem)));
    HIPCHK(hipMalloc((void **)&s->sym_rowbase_dev, bounds.size() * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void **)&s->sym_cov_begin_dev, cbegin.size() * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void **)&s->sym_cov_dev, (cov.size() ? cov.size() : 1) * sizeof(SymCov)));
    HIPCHK(hipMalloc(&s->sym_slab_s, (size_t)(row ? row : 1) * pl.sb * s->esz));
    HIPCHK(hipMalloc(&s->sym_slab_r, (size_t)(pl.slab_r_elems ? pl.slab_r_elems : 1) * s->esz));
    int rc;
    if ((rc = copy_h2d(s, s->sym_items_dev, pl.items.data(), pl.items.size() * sizeof(SymItem)))) return rc;
    if ((rc = copy_h2d(s, s->sym_rowbase_dev, bounds.data(), bounds.size() * sizeof(uint32_t)))) return rc;
    if ((rc = copy_h2d(s, s->sym_cov_begin_dev, cbegin.data(), cbegin.size() * sizeof(uint32_t)))) return rc;
    if ((rc = copy_h2d(s, s->sym_cov_dev, cov.data(), cov.size() * sizeof(SymCov)))) return rc;
    HIPCHK(hipMalloc((void **)&s->sym_ticket, 3 * 128));
    HIPCHK(hipMemsetAsync(s->sym_ticket, 0, 3 * 128, s->stream));
    // Side stream for the local items when they are about one wave of workgroups (P = 8 at N = 262 144: 615 items
    // on 512 resident slots, 150 us where 128 us of work is due): run concurrently, the cross items fill the CUs
    // the last local workgroups leave idle (-1.7 % step time; with two LONG launches sharing the chip, P = 2, the
    // same trick costs 4 % — profiles/history/r01_aux_stream_ab.log — hence the bound).  nb_params.sym_aux_stream = 1 / -1 forces it.
    s->aux_local = s->sym_sharded && (s->p.sym_aux_stream ? s->p.sym_aux_stream > 0 : s->sym_items_local <= 4u * (uint32_t)s->cus);
    if (s->aux_local || s->sym_items_late) {         // the late items always run on the side stream
        HIPCHK(hipStreamCreateWithFlags(&s->aux, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s->ev_late, hipEventDisableTiming));
    }
    if (split) {
        if (s->p.acc_buffers[0]) { s->acc_full = s->p.acc_buffers[0]; s->acc_owned = s->p.acc_buffers[1]; s->own_acc = false; }
        else {
            HIPCHK(hipMalloc(&s->acc_full, (size_t)n * s->esz));
            if (s->sym_sharded) HIPCHK(hipMalloc(&s->acc_owned, s->i_count * s->esz));
        }
    }
    return NB_OK;
}

static void free_all(nb_sim *s)
{
    if (!s) return;
    (void)hipSetDevice(s->dev);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (auto &e : s->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto &e : s->ev_used) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (s->own_pos) { (void)hipFree(s->pos[0]); (void)hipFree(s->pos[1]); }
    (void)hipFree(s->mass); (void)hipFree(s->radius); (void)hipFree(s->sigma);
    (void)hipFree(s->vel); (void)hipFree(s->acc); (void)hipFree(s->partial);
    (void)hipFree(s->aos_dev); (void)hipFree(s->ered_dev); (void)hipFree(s->pred_dev);
    (void)hipFree(s->sym_items_dev); (void)hipFree(s->sym_rowbase_dev); (void)hipFree(s->sym_cov_begin_dev); (void)hipFree(s->sym_cov_dev);
    if (s->own_acc) { (void)hipFree(s->acc_full); (void)hipFree(s->acc_owned); }
    (void)hipFree(s->sym_slab_s); (void)hipFree(s->sym_slab_r);
    (void)hipFree(s->sym_ticket);
    if (s->copy_stream) { (void)hipStreamSynchronize(s->copy_stream); (void)hipStreamDestroy(s->copy_stream); }
    if (s->ev_packed) (void)hipEventDestroy(s->ev_packed);
    if (s->ev_copied) (void)hipEventDestroy(s->ev_copied);
    if (s->staging) (void)hipHostFree(s->staging);
    if (s->bounce) (void)hipHostFree(s->bounce);
    for (hipEvent_t e : s->ev_bounce) if (e) (void)hipEventDestroy(e);
    if (s->aux) { (void)hipStreamSynchronize(s->aux); (void)hipStreamDestroy(s->aux); }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    if (s->ev_late) (void)hipEventDestroy(s->ev_late);
    for (hipEvent_t e : s->ev_x) if (e) (void)hipEventDestroy(e);
    if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

static int launch_force(nb_sim *s, const ForceJob &j);
static int launch_integrate(nb_sim *s, uint32_t nslabs, double dt_kick, double dt_drift, int flags);

// The MEASURED rule of the mass-scaled body (NB_FLAG_MASS_SCALING_MEASURED; opt-in since ABI 6): decided per upload FROM THE
// DATA.  The accelerations of the uploaded bodies are evaluated twice — with the per-pair mass multiplies (MM_GENERAL) and with
// the masses folded into the pair geometry (MM_SCALED) — and the scaled body is taken only if the two agree to MASS_SCALING_TOL
// of the force scale (max |a|): 2e-6, a fifth of north_star's 1e-5.  Equal-ish light masses pass (Plummer spheres at the headline
// size: 4e-7, the two bodies' ordinary rounding difference); THE REFERENCE'S OWN BODIES DO NOT (Simulation.hpp:347-603: 4.0e-5 —
// its light bodies sit 1e3 ... 1e5 from the origin with eps = 1, so sigma * x is rounded at that magnitude while close pairs
// are a unit apart) and keep MM_GENERAL; a four-decade mixture with close heavy pairs fails likewise (6e-5:
// tests/test_headline_gpu.py).  What the measure is NOT (why it is no longer the default): it is a global absolute figure at
// t = 0 — a body whose own |a| is far below max |a| can carry a larger relative error and still pass, and a system that
// passes at upload and later drifts from the origin or forms close heavy pairs is not measured again.
// Cost: two force evaluations per upload (14 ms at N = 262 144).
constexpr float MASS_SCALING_TOL = 2e-6f;
static int choose_mass_scaling(nb_sim *s)
{
    const size_t bytes = s->i_count * s->esz;
    const uint32_t n = (uint32_t)s->i_count;
    void *saved = nullptr, *ref = nullptr;
    uint32_t *out = nullptr;
    int rc = NB_OK;
    const bool prof = s->prof;
    s->prof = false;                  // the check's two force launches are not the caller's: keep them out of nb_profile_read
    auto body = [&]() -> int {
        HIPCHK(hipMalloc(&saved, bytes));
        HIPCHK(hipMalloc(&ref, bytes));
        HIPCHK(hipMalloc((void **)&out, 2 * sizeof(uint32_t)));
        HIPCHK(hipMemcpyAsync(saved, s->acc, bytes, hipMemcpyDeviceToDevice, s->stream));      // the uploaded acc field survives the check
        HIPCHK(hipMemsetAsync(out, 0, 2 * sizeof(uint32_t), s->stream));
        int r;
        s->mass_scaled = false;
        if ((r = launch_force(s, s->job_all)) || (r = launch_integrate(s, s->slabs_all, 0.0, 0.0, 0))) return r;
        HIPCHK(hipMemcpyAsync(ref, s->acc, bytes, hipMemcpyDeviceToDevice, s->stream));
        s->mass_scaled = true;
        if ((r = launch_force(s, s->job_all)) || (r = launch_integrate(s, s->slabs_all, 0.0, 0.0, 0))) return r;
        max_deviation_f32<<<(n + BLOCK - 1) / BLOCK, BLOCK, 0, s->stream>>>((const float2 *)s->acc, (const float2 *)ref, n, out);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(s->acc, saved, bytes, hipMemcpyDeviceToDevice, s->stream));
        uint32_t *host = nullptr;
        HIPCHK(hipHostMalloc((void **)&host, 2 * sizeof(uint32_t), hipHostMallocDefault));
        hipError_t e = hipMemcpyAsync(host, out, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        float dev = 0.f, scale = 0.f;
        memcpy(&dev, &host[0], sizeof dev);
        memcpy(&scale, &host[1], sizeof scale);
        (void)hipHostFree(host);
        if (e != hipSuccess) return nb_fail(hip_code(e), "nb_upload: mass-scaling check: %s", hipGetErrorString(e));
        const bool comparable = dev == dev && scale == scale && scale > 0.f && scale < 3.0e38f;
        s->mass_scaling_dev = comparable ? dev / scale : HUGE_VALF;
        s->mass_scaled = comparable && dev <= MASS_SCALING_TOL * scale;
        return NB_OK;
    };
    rc = body();
    s->prof = prof;
    if (rc) s->mass_scaled = false;
    (void)hipFree(saved); (void)hipFree(ref); (void)hipFree(out);
    s->acc_valid = false;
    return rc;
}

// Caller-owned device memory (params.pos_buffers / acc_buffers) or a caller's stream: whatever the caller enqueued on those
// buffers before handing them over — a fill on another stream, a collective still in flight — must be THROUGH before the
// upload writes them.  The handle's stream is non-blocking (or foreign), so nothing orders it against that work; a zero-fill
// queued on the caller's stream could land after the upload and wipe the positions (found in round 5 by the sharded host's
// start-up validation).  One device-wide fence per creation / upload; never on the step path.
static int fence_foreign_work(nb_sim *s)
{
    if (s->own_pos && s->own_acc && s->own_stream) return NB_OK;
    HIPCHK(hipDeviceSynchronize());
    return NB_OK;
}

static int do_upload(nb_sim *s, const nb_body *in)
{
    { const int rc = fence_foreign_work(s); if (rc) return rc; }
    // Equal masses (the synthetic Plummer workload, most N-body ICs) let the force kernel hoist the
    // per-pair mass multiply: 8 instead of 9 packed ops per two pairs.  NB_FLAG_NO_UNIFORM_MASS disables it.
    s->uniform_mass = s->n > 0 && !(s->p.flags & NB_FLAG_NO_UNIFORM_MASS) && s->p.sum_order == NB_SUM_TILED;
    for (size_t i = 1; s->uniform_mass && i < s->n; ++i)
        if (memcmp(&in[i].mass, &in[0].mass, sizeof(float)) != 0) s->uniform_mass = false;
    s->um_mass = in[0].mass;
    // Mass scaling: individual masses folded into the pair geometry (MM_SCALED: 11 + 2 instead of 12 + 2 instructions per body of
    // the symmetric kernel, 8 + 2 instead of 9 + 2 in the one-sided one).  REPRESENTABLE when every mass is positive and
    // sigma = m^(-1/2), sigma * (the kernels' padding coordinate 1e18) and g^3 <= (sqrt(m_max) / eps)^3 all stay finite floats
    // with room to spare; exact rsqrt only (the Quake mode keeps the reference's arithmetic).  Whether it is also HARMLESS — its
    // displacement is rounded once more, 6e-8 |x_j| / |d| per pair force, which heavy close pairs turn into several 1e-5 of the
    // force scale — depends on the data, so it is the CALLER's decision (ABI 6: off unless asked for — the same pair arithmetic on
    // one GPU and on N, at t = 0 and later): NB_FLAG_MASS_SCALING takes it wherever representable, NB_FLAG_MASS_SCALING_MEASURED
    // lets choose_mass_scaling() above measure it on the uploaded bodies (unsharded handles), NB_FLAG_NO_MASS_SCALING overrides both.
    s->mass_scaled = false;
    s->mass_scaling_dev = -1.0f;
    bool scalable = false;
    if (!s->uniform_mass && s->p.sum_order == NB_SUM_TILED && !needs_guard(s) && !s->fp64 && !s->dims3 &&
        s->p.rsqrt_mode == NB_RSQRT_EXACT && (s->p.flags & (NB_FLAG_MASS_SCALING | NB_FLAG_MASS_SCALING_MEASURED)) && !(s->p.flags & NB_FLAG_NO_MASS_SCALING)) {
        double mmin = HUGE_VAL, mmax = 0.0;
        bool finite = true;
        for (size_t i = 0; i < s->n; ++i) {
            const double m = (double)in[i].mass;
            if (!(m == m) || m > 3.0e38) finite = false;
            if (m < mmin) mmin = m;
            if (m > mmax) mmax = m;
        }
        const double eps = (double)s->p.eps;
        scalable = finite && mmin >= 1e-30 && std::pow(mmax, 1.5) / (eps * eps * eps) <= 1e36;
    }
    const bool forced_scaling = scalable && (s->p.flags & NB_FLAG_MASS_SCALING);
    const bool auto_scaling = scalable && !forced_scaling && (s->p.flags & NB_FLAG_MASS_SCALING_MEASURED) && s->i_count == s->n && !s->sym_sharded && !s->sym_replicated;
    const bool was_uniform = s->sym_first_wave_uniform, was_scaled = s->sym_first_wave_scaled;
    { const int rc = copy_h2d(s, s->aos_dev, in, s->n * sizeof(nb_body)); if (rc) return rc; }
    const uint32_t n = (uint32_t)s->n, g = (n + BLOCK - 1) / BLOCK;
    // both replicas get the full initial positions
    for (int b = 0; b < 2; ++b) {
        if (s->dims3 && s->fp64)
            unpack_bodies3<double><<<g, BLOCK, 0, s->stream>>>(s->aos_dev, n, (double4 *)s->pos[b], (double4 *)s->vel, (double4 *)s->acc, s->radius,
                                                               (uint32_t)s->i_begin, (uint32_t)s->i_count);
        else if (s->dims3)
            unpack_bodies3<float><<<g, BLOCK, 0, s->stream>>>(s->aos_dev, n, (float4 *)s->pos[b], (float4 *)s->vel, (float4 *)s->acc, s->radius,
                                                              (uint32_t)s->i_begin, (uint32_t)s->i_count);
        else if (s->fp64)
            unpack_bodies<double><<<g, BLOCK, 0, s->stream>>>(s->aos_dev, n, (double2 *)s->pos[b], (double *)s->mass,
                                                             (double2 *)s->vel, (double2 *)s->acc, s->radius,
                                                             (uint32_t)s->i_begin, (uint32_t)s->i_count);
        else
            unpack_bodies<float><<<g, BLOCK, 0, s->stream>>>(s->aos_dev, n, (float2 *)s->pos[b], (float *)s->mass,
                                                            (float2 *)s->vel, (float2 *)s->acc, s->radius,
                                                            (uint32_t)s->i_begin, (uint32_t)s->i_count);
    }
    HIPCHK(hipGetLastError());
    if (forced_scaling || auto_scaling) {
        if (!s->sigma) HIPCHK(hipMalloc((void **)&s->sigma, s->n * sizeof(float)));
        mass_sigma<<<g, BLOCK, 0, s->stream>>>((const float *)s->mass, s->sigma, n);
        HIPCHK(hipGetLastError());
    }
    s->mass_scaled = forced_scaling;
    HIPCHK(hipStreamSynchronize(s->stream));  // `in` may be pageable and freed by the caller
    s->acc_valid = false;
    if (auto_scaling) { const int rc = choose_mass_scaling(s); if (rc) return rc; }
    // the resident-slot count behind the static / dynamic item split belongs to ONE kernel instantiation (its VGPR count sets the
    // occupancy): the check above launches MM_GENERAL whatever the verdict, and a re-upload can flip uniform_mass — ask again
    if (auto_scaling || was_uniform != s->uniform_mass || was_scaled != s->mass_scaled) s->sym_first_wave = 0;
    s->sym_first_wave_uniform = s->uniform_mass;
    s->sym_first_wave_scaled = s->mass_scaled;
    return NB_OK;
}

extern "C" nb_sim *nb_create(const nb_body *init, size_t n, const nb_params *params)
{
    nb_clear_error();
    nb_params p;
    if (params) {
        if (params->struct_size != sizeof(nb_params)) {
            nb_set_error("nb_create: params->struct_size %u != %zu (use nb_params_default)", params->struct_size, sizeof(nb_params));
            return nullptr;
        }
        p = *params;
    } else nb_params_default(&p);
    if (!init || n == 0) { nb_set_error("nb_create: no bodies"); return nullptr; }
    if (n > 0x7fffff00u) { nb_set_error("nb_create: n=%zu exceeds the 32-bit index range of the kernels", n); return nullptr; }
    if (!(p.eps >= 0.0f)) { nb_set_error("nb_create: eps must be >= 0"); return nullptr; }
    if (p.precision != NB_FP32 && p.precision != NB_FP64) { nb_set_error("nb_create: bad precision %d", p.precision); return nullptr; }
    if (p.rsqrt_mode != NB_RSQRT_EXACT && p.rsqrt_mode != NB_RSQRT_QUAKE) { nb_set_error("nb_create: bad rsqrt_mode %d", p.rsqrt_mode); return nullptr; }
    if (p.sum_order != NB_SUM_TILED && p.sum_order != NB_SUM_SEQUENTIAL) { nb_set_error("nb_create: bad sum_order %d", p.sum_order); return nullptr; }
    if (p.integrator != NB_INTEGRATOR_KICK_DRIFT && p.integrator != NB_INTEGRATOR_KDK) { nb_set_error("nb_create: bad integrator %d", p.integrator); return nullptr; }
    if (p.precision == NB_FP64 && (p.rsqrt_mode == NB_RSQRT_QUAKE || p.sum_order == NB_SUM_SEQUENTIAL)) {
        nb_set_error("nb_create: quake rsqrt / sequential order are fp32 (reference arithmetic) modes");
        return nullptr;
    }
    if (p.flags & ~(NB_FLAG_NO_SYMMETRY | NB_FLAG_NO_UNIFORM_MASS | NB_FLAG_NO_GUIDED_TAIL | NB_FLAG_SHARD_ALLREDUCE | NB_FLAG_SHARD_SINGLE | NB_FLAG_MASS_SCALING | NB_FLAG_NO_MASS_SCALING | NB_FLAG_STATIC_ITEMS | NB_FLAG_MASS_SCALING_MEASURED)) { nb_set_error("nb_create: unknown bits in flags 0x%x", (unsigned)p.flags); return nullptr; }
    if (p.extras & ~(NB_EXTRA_VCLAMP | NB_EXTRA_BOUNDARY)) { nb_set_error("nb_create: unknown bits in extras 0x%x", (unsigned)p.extras); return nullptr; }
    if (p.sym_chunks_per_item < 0 || p.sym_aux_stream < -1 || p.sym_aux_stream > 1 || p.j_slices < 0 || p.sym_chunk_pairs < -1 || p.sym_chunk_pairs > 1 ||
        (p.sym_tile != 0 && p.sym_tile != (int32_t)SYM_SB_WS && p.sym_tile != (int32_t)SYM_SB) ||
        (p.lanes_p != 0 && p.lanes_p != 1 && p.lanes_p != 2 && p.lanes_p != 4) || !(p.sym_late_us == p.sym_late_us)) {
        nb_set_error("nb_create: tuning field out of range (sym_chunks_per_item >= 0, sym_aux_stream in -1..1, lanes_p in {0,1,2,4}, j_slices >= 0, sym_tile in {0,512,2048})");
        return nullptr;
    }
    if (p.sym_tail[0] != 0.0f || p.sym_tail[1] != 0.0f || p.sym_tail[2] != 0.0f) {
        if (!(p.sym_tail[0] > 0.0f && p.sym_tail[0] <= p.sym_tail[1] && p.sym_tail[1] <= p.sym_tail[2] && p.sym_tail[2] <= 1.0f)) {
            nb_set_error("nb_create: sym_tail must be 0 < a <= b <= c <= 1 (or all 0 for the defaults)");
            return nullptr;
        }
    }
    if (p._reserved0 != 0) { nb_set_error("nb_create: params->_reserved0 must be 0 (use nb_params_default)"); return nullptr; }
    if (p.dims == 0) p.dims = 2;
    if (p.dims != 2 && p.dims != 3) { nb_set_error("nb_create: dims must be 2 or 3"); return nullptr; }
    if (p.dims == 3 && (p.sum_order != NB_SUM_TILED || p.extras != 0)) {
        nb_set_error("nb_create: dims = 3 supports the tiled sum without extras (the reference defines its sequential order and "
                     "iterate()'s clamp / boundary in the plane only)");
        return nullptr;
    }
    if (p.i_count == 0) { p.i_begin = 0; p.i_count = n; }
    if (p.i_begin + p.i_count > n) { nb_set_error("nb_create: owned block [%llu,+%llu) exceeds n=%zu", (unsigned long long)p.i_begin, (unsigned long long)p.i_count, n); return nullptr; }
    if ((p.pos_buffers[0] == nullptr) != (p.pos_buffers[1] == nullptr)) { nb_set_error("nb_create: give both pos_buffers or none"); return nullptr; }
    if (p.pos_rows != 0 && (!p.pos_buffers[0] || p.pos_rows < n)) { nb_set_error("nb_create: pos_rows describes caller-owned pos_buffers and must be >= n"); return nullptr; }
    if ((p.acc_buffers[0] == nullptr) != (p.acc_buffers[1] == nullptr)) { nb_set_error("nb_create: give both acc_buffers or none"); return nullptr; }
    if (p.shard_world < 0 || (p.shard_world > 1 && (p.shard_rank < 0 || p.shard_rank >= p.shard_world))) { nb_set_error("nb_create: bad shard_rank/shard_world %d/%d", p.shard_rank, p.shard_world); return nullptr; }

    int ndev = nb_device_count();
    if (ndev <= 0) { nb_fail(NB_ENODEVICE, "nb_create: no HIP device visible (this library has no CPU path)"); return nullptr; }
    int dev = p.device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) { nb_set_error("nb_create: device %d out of range (%d visible)", dev, ndev); return nullptr; }

    nb_sim *s = new (std::nothrow) nb_sim;
    if (!s) { nb_fail(NB_ENOMEM, "nb_create: out of host memory"); return nullptr; }
    s->p = p; s->n = n; s->i_begin = (size_t)p.i_begin; s->i_count = (size_t)p.i_count; s->dev = dev;
    s->frame = p.first_frame;
    s->fp64 = p.precision == NB_FP64;
    s->rsz = s->fp64 ? 8 : 4;
    s->dims3 = p.dims == 3;
    s->esz = (s->dims3 ? 4 : 2) * s->rsz;      // one position / velocity / acceleration / slab element: real2 or real4

    auto fail = [&](const char *what, hipError_t e) -> nb_sim * {
        free_all(s);
        (void)hipGetLastError();        // do not leave the failure sticky for the next handle's launch checks
        nb_fail(hip_code(e), "nb_create: %s: %s", what, hipGetErrorString(e));
        return nullptr;
    };
    hipError_t e;
    if ((e = hipSetDevice(dev)) != hipSuccess) return fail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return fail("hipGetDeviceProperties", e);
    s->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (p.stream) { s->stream = (hipStream_t)p.stream; s->own_stream = false; }
    else { if ((e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e); s->own_stream = true; }

    // decided once (eligibility looks at the free device memory, which the allocations below change)
    s->sym = want_sym(s);
    s->sym_sharded = want_sym_sharded(s);
    s->sym_replicated = want_sym_replicated(s);
    if (s->sym_replicated) s->sym = false;          // the whole-system plan is not built: this rank evaluates its share only
    plan(s);
    const size_t r2 = s->esz;
    if (p.pos_buffers[0]) { s->pos[0] = p.pos_buffers[0]; s->pos[1] = p.pos_buffers[1]; s->own_pos = false; s->pos_rows = p.pos_rows ? (size_t)p.pos_rows : n; }
    else {
        // a sharded handle's replicas hold world * ceil(n / world) rows: an all-gather over ragged blocks moves equal counts
        const size_t w = p.shard_world > 1 ? (size_t)p.shard_world : 1;
        s->pos_rows = w * ((n + w - 1) / w);
        if ((e = hipMalloc(&s->pos[0], s->pos_rows * r2)) != hipSuccess) return fail("hipMalloc pos", e);
        if ((e = hipMalloc(&s->pos[1], s->pos_rows * r2)) != hipSuccess) return fail("hipMalloc pos", e);
        if (s->pos_rows > n) {          // the padding rows travel with the last block: keep them defined
            if ((e = hipMemsetAsync((char *)s->pos[0] + n * r2, 0, (s->pos_rows - n) * r2, s->stream)) != hipSuccess) return fail("hipMemset pos", e);
            if ((e = hipMemsetAsync((char *)s->pos[1] + n * r2, 0, (s->pos_rows - n) * r2, s->stream)) != hipSuccess) return fail("hipMemset pos", e);
        }
    }
    s->slabs_cap = s->slabs_all > s->slabs_two_phase ? s->slabs_all : s->slabs_two_phase;
    if ((e = hipMalloc(&s->mass, n * s->rsz)) != hipSuccess) return fail("hipMalloc mass", e);
    if ((e = hipMalloc((void **)&s->radius, n * sizeof(float))) != hipSuccess) return fail("hipMalloc radius", e);
    if ((e = hipMalloc(&s->vel, s->i_count * r2)) != hipSuccess) return fail("hipMalloc vel", e);
    if ((e = hipMalloc(&s->acc, s->i_count * r2)) != hipSuccess) return fail("hipMalloc acc", e);
    if ((e = hipMalloc(&s->partial, (size_t)s->slabs_cap * s->i_count * r2)) != hipSuccess) return fail("hipMalloc partial", e);
    if ((e = hipMalloc((void **)&s->aos_dev, n * sizeof(nb_body))) != hipSuccess) return fail("hipMalloc aos", e);
    s->ered_blocks = (s->i_count + BLOCK - 1) / BLOCK;
    if ((e = hipMalloc((void **)&s->ered_dev, 2 * s->ered_blocks * sizeof(double))) != hipSuccess) return fail("hipMalloc energy", e);

    if ((s->sym || s->sym_sharded || s->sym_replicated) && plan_sym(s) != NB_OK) { free_all(s); (void)hipGetLastError(); return nullptr; }
    if (do_upload(s, init) != NB_OK) { free_all(s); (void)hipGetLastError(); return nullptr; }
    return s;
}

extern "C" void nb_destroy(nb_sim *s) { free_all(s); }

extern "C" int nb_upload(nb_sim *s, const nb_body *in)
{
    if (!s || !in) return nb_fail(NB_EINVAL, "nb_upload: NULL argument");
    if (bind(s)) return NB_EHIP;
    if (nb_snapshot_wait(s)) return nb_last_error_code();      // the AoS staging array may still feed a pipelined snapshot
    return do_upload(s, in);
}

// ---------------------------------------------------------------------------
// profiling events
// ---------------------------------------------------------------------------
static int prof_begin(nb_sim *s, std::pair<hipEvent_t, hipEvent_t> *pr, hipStream_t st = nullptr)
{
    if (s->ev_pool.empty()) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        s->ev_pool.push_back({a, b});
    }
    *pr = s->ev_pool.back();
    s->ev_pool.pop_back();
    HIPCHK(hipEventRecord(pr->first, st ? st : s->stream));
    return NB_OK;
}

static int prof_end(nb_sim *s, const std::pair<hipEvent_t, hipEvent_t> &pr, hipStream_t st = nullptr, uint32_t passes = 1)
{
    HIPCHK(hipEventRecord(pr.second, st ? st : s->stream));
    s->ev_used.push_back(pr);
    s->ev_weight.push_back(passes);
    return NB_OK;
}

static int prof_collect(nb_sim *s)
{
    for (size_t k = 0; k < s->ev_used.size(); ++k) {
        auto &pr = s->ev_used[k];
        HIPCHK(hipEventSynchronize(pr.second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
        s->prof_ms += ms;
        s->prof_launches += s->ev_weight[k];          // a pipeline launch counts as the force passes (steps) it holds
        s->ev_pool.push_back(pr);
    }
    s->ev_used.clear();
    s->ev_weight.clear();
    return NB_OK;
}

// ---------------------------------------------------------------------------
// force launch
// ---------------------------------------------------------------------------
template <int P, int RSQ, bool GUARD>
static void launch_tiled_f32(nb_sim *s, const ForceJob &j, float eps2)
{
    const uint32_t i_tiles = j.i_tiles;
    const uint32_t grid = grid_blocks(i_tiles, j.js);
    float2 *out = (float2 *)s->partial + (size_t)j.slab0 * s->i_count;
    if constexpr (!GUARD) {
        if (s->uniform_mass) {
            force_tiled_f32<P, RSQ, false, 8, true, F32_WS><<<grid, BLOCK, 0, s->stream>>>(
                (const float2 *)s->pos[s->cur], (const float *)s->mass, s->sigma, out,
                (uint32_t)s->i_begin, (uint32_t)s->i_count, j.j_begin, j.j_end, j.js, i_tiles, eps2, s->um_mass,
                j.gap_begin, j.gap_len);
            return;
        }
        if constexpr (RSQ == RSQ_EXACT) {
            if (s->mass_scaled) {      // individual masses folded into the pair geometry: no mass multiply in the body
                force_tiled_f32<P, RSQ_EXACT, false, 8, false, F32_WS, true><<<grid, BLOCK, 0, s->stream>>>(
                    (const float2 *)s->pos[s->cur], (const float *)s->mass, s->sigma, out,
                    (uint32_t)s->i_begin, (uint32_t)s->i_count, j.j_begin, j.j_end, j.js, i_tiles, eps2, 1.0f, j.gap_begin, j.gap_len);
                return;
            }
        }
    }
    force_tiled_f32<P, RSQ, GUARD, 8, false, F32_WS><<<grid, BLOCK, 0, s->stream>>>(
        (const float2 *)s->pos[s->cur], (const float *)s->mass, s->sigma, out,
        (uint32_t)s->i_begin, (uint32_t)s->i_count, j.j_begin, j.j_end, j.js, i_tiles, eps2, 1.0f, j.gap_begin, j.gap_len);
}

template <int P, bool GUARD>
static void launch_tiled_f64(nb_sim *s, const ForceJob &j, double eps2)
{
    const uint32_t i_tiles = j.i_tiles;
    const uint32_t grid = grid_blocks(i_tiles, j.js);
    double2 *out = (double2 *)s->partial + (size_t)j.slab0 * s->i_count;
    force_tiled_f64<P, GUARD, 4><<<grid, BLOCK, 0, s->stream>>>(
        (const double2 *)s->pos[s->cur], (const double *)s->mass, out,
        (uint32_t)s->i_begin, (uint32_t)s->i_count, j.j_begin, j.j_end, j.js, i_tiles, eps2, j.gap_begin, j.gap_len);
}

// Symmetric kernel over items [first, first + count) of this handle (HIP events around the launch
// when profiling).
static int launch_sym_items(nb_sim *s, uint32_t first, uint32_t count, hipStream_t st = nullptr)
{
    if (count == 0) return NB_OK;
    if (!st) st = s->stream;
    std::pair<hipEvent_t, hipEvent_t> pr;
    if (s->prof && prof_begin(s, &pr, st)) return NB_EHIP;
    const uint32_t n = (uint32_t)s->n;
    const SymItem *items = s->sym_items_dev + first;
    const bool quake = s->p.rsqrt_mode == NB_RSQRT_QUAKE;
    // Dynamic work items (sym_item_index, nb_kernels.hip.h): past the first resident wave a workgroup draws its item when it starts.
    // One counter per launch kind of the handle (this launch's is told by where its items begin), monotonic: `tb` = what the launches
    // of that kind have drawn so far.
    const uint32_t slot = first == 0 ? 0u : (first == s->sym_items_local ? 1u : 2u);
    const bool dyn = s->sym_ticket != nullptr && !(s->p.flags & NB_FLAG_STATIC_ITEMS);
    uint32_t *tk = nullptr;
    uint32_t fw = 0, tb = 0, drawn = 0;      // drawn: tickets this launch will take — committed to the host's record only once the launch is known to be enqueued
#define NB_TICKETS(KERNEL)                                                                                                        \
    do {                                                                                                                          \
        if (dyn) {                                                                                                                \
            if (!s->sym_first_wave) {                                                                                             \
                int per_cu = 0;                                                                                                   \
                HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, BLOCK, 0));                                  \
                s->sym_first_wave = (uint32_t)(per_cu > 0 ? per_cu : 1) * (uint32_t)s->cus;                                       \
            }                                                                                                                     \
            fw = s->sym_first_wave < count ? s->sym_first_wave : count;                                                           \
            tk = s->sym_ticket + slot * 32u; tb = s->sym_ticket_base[slot];                                                       \
            drawn = count - fw;                                                                                                   \
        }                                                                                                                         \
    } while (0)
    if (s->dims3 && s->fp64) {
        const double eps2 = (double)s->p.eps * (double)s->p.eps;
        const double4 *pos = (const double4 *)s->pos[s->cur];
        double4 *ss = (double4 *)s->sym_slab_s, *sr = (double4 *)s->sym_slab_r;
        if (s->uniform_mass) { NB_TICKETS(force_sym3_f64<true>);  force_sym3_f64<true><<<count, BLOCK, 0, st>>>(pos, items, ss, sr, n, eps2, (double)s->um_mass, tk, fw, tb); }
        else                 { NB_TICKETS(force_sym3_f64<false>); force_sym3_f64<false><<<count, BLOCK, 0, st>>>(pos, items, ss, sr, n, eps2, 1.0, tk, fw, tb); }
    } else if (s->dims3) {
        const float eps2 = s->p.eps * s->p.eps;
        const float4 *pos = (const float4 *)s->pos[s->cur];
        float4 *ss = (float4 *)s->sym_slab_s, *sr = (float4 *)s->sym_slab_r;
#define NB_SYM3_LAUNCH(RQ, UMB, PR, UMV) do { NB_TICKETS((force_sym3_f32<RQ, UMB, PR>)); force_sym3_f32<RQ, UMB, PR><<<count, BLOCK, 0, st>>>(pos, items, ss, sr, n, eps2, UMV, tk, fw, tb); } while (0)
        // chunk pairs in 3-D pay with equal masses only (-1 ... -3 %); with individual masses the pair body needs 216 VGPRs
        // (2 waves per SIMD) and loses 4 % (profiles/history/r03_chunk_pairs_3d.log): that case keeps the single-chunk sweep
        // (any plan, even chunk counts included, runs on either kernel) unless nb_params.sym_chunk_pairs = 1 forces it
        const bool pairs = s->sym_pairs && (s->uniform_mass || s->p.sym_chunk_pairs > 0);
        if (s->uniform_mass) {
            if (quake) { if (pairs) NB_SYM3_LAUNCH(RSQ_QUAKE, true, true, s->um_mass); else NB_SYM3_LAUNCH(RSQ_QUAKE, true, false, s->um_mass); }
            else       { if (pairs) NB_SYM3_LAUNCH(RSQ_EXACT, true, true, s->um_mass); else NB_SYM3_LAUNCH(RSQ_EXACT, true, false, s->um_mass); }
        } else {
            if (quake) { if (pairs) NB_SYM3_LAUNCH(RSQ_QUAKE, false, true, 1.0f); else NB_SYM3_LAUNCH(RSQ_QUAKE, false, false, 1.0f); }
            else       { if (pairs) NB_SYM3_LAUNCH(RSQ_EXACT, false, true, 1.0f); else NB_SYM3_LAUNCH(RSQ_EXACT, false, false, 1.0f); }
        }
#undef NB_SYM3_LAUNCH
    } else if (s->fp64) {
        const double eps2 = (double)s->p.eps * (double)s->p.eps;
        const double2 *pos = (const double2 *)s->pos[s->cur];
        const double *mass = (const double *)s->mass;
        if (s->uniform_mass) { NB_TICKETS(force_sym_f64<true>);  force_sym_f64<true><<<count, BLOCK, 0, st>>>(pos, mass, items, (double2 *)s->sym_slab_s, (double2 *)s->sym_slab_r, n, eps2, (double)s->um_mass, tk, fw, tb); }
        else                 { NB_TICKETS(force_sym_f64<false>); force_sym_f64<false><<<count, BLOCK, 0, st>>>(pos, mass, items, (double2 *)s->sym_slab_s, (double2 *)s->sym_slab_r, n, eps2, 1.0, tk, fw, tb); }
    } else {
        const float eps2 = s->p.eps * s->p.eps;
        const float2 *pos = (const float2 *)s->pos[s->cur];
        const float *mass = (const float *)s->mass;
        float2 *ss = (float2 *)s->sym_slab_s, *sr = (float2 *)s->sym_slab_r;
        const float *sg = s->sigma;
#define NB_SYM_GO(KERNEL, UMV) do { NB_TICKETS(KERNEL); KERNEL<<<count, BLOCK, 0, st>>>(pos, mass, sg, items, ss, sr, n, eps2, UMV, tk, fw, tb); } while (0)
#define NB_SYM_LAUNCH(RQ, MMODE, PR, UMV)                                                                                         \
        do {                                                                                                                      \
            if (s->sym_sb == SYM_SB_WS) NB_SYM_GO((force_sym_f32<RQ, MMODE, PR, true>), UMV);                                     \
            else                        NB_SYM_GO((force_sym_f32<RQ, MMODE, PR, false>), UMV);                                    \
        } while (0)
        const bool pairs = s->sym_pairs && !s->mass_scaled;          // chunk pairs (sym_chunks2): large systems, see want_pairs
        if (s->uniform_mass) {
            if (quake) { if (pairs) NB_SYM_LAUNCH(RSQ_QUAKE, MM_UNIFORM, true, s->um_mass); else NB_SYM_LAUNCH(RSQ_QUAKE, MM_UNIFORM, false, s->um_mass); }
            else       { if (pairs) NB_SYM_LAUNCH(RSQ_EXACT, MM_UNIFORM, true, s->um_mass); else NB_SYM_LAUNCH(RSQ_EXACT, MM_UNIFORM, false, s->um_mass); }
        } else if (s->mass_scaled) {                 // exact rsqrt only (decided at upload)
            NB_SYM_LAUNCH(RSQ_EXACT, MM_SCALED, false, 1.0f);
        } else {
            if (quake) { if (pairs) NB_SYM_LAUNCH(RSQ_QUAKE, MM_GENERAL, true, 1.0f); else NB_SYM_LAUNCH(RSQ_QUAKE, MM_GENERAL, false, 1.0f); }
            else       { if (pairs) NB_SYM_LAUNCH(RSQ_EXACT, MM_GENERAL, true, 1.0f); else NB_SYM_LAUNCH(RSQ_EXACT, MM_GENERAL, false, 1.0f); }
        }
#undef NB_SYM_LAUNCH
#undef NB_SYM_GO
    }
#undef NB_TICKETS
    {
        // The host's record of the tickets advances only with a launch that was accepted: if this one was refused, the device
        // counter did not move either, but a kernel of an EARLIER launch may still be drawing — nothing sound can follow on
        // this handle, so it is refused from here on (nb_step_begin) instead of indexing items[] with a base that may be off.
        const hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) {
            s->broken = true;
            return nb_fail(hip_code(e_), "launch of the symmetric force kernel failed: %s; the handle is unusable from here on", hipGetErrorString(e_));
        }
        s->sym_ticket_base[slot] += drawn;
    }
    if (s->prof && prof_end(s, pr, st)) return NB_EHIP;
    return NB_OK;
}

// Sum of the slabs.  fuse_step (whole-system handles): apply kick and drift in the same kernel; otherwise the
// summed (unsharded) or partial (sharded rank) acceleration of every particle is stored: slab 0 / acc_full.
// A sharded rank's late items are left out here (launch_sym_gather_late folds them in).
static int launch_sym_gather(nb_sim *s, bool fuse_step, double dt)
{
    const uint32_t n = (uint32_t)s->n, per = s->dims3 ? (uint32_t)GATHER_T : (uint32_t)GATHER_P, gg = (n + per - 1) / per, tiles = s->sym_tiles;
    void *dst = (s->sym_sharded || s->sym_replicated) ? s->acc_full : s->partial;
    const uint32_t *lo = s->sym_rowbase_dev, *hi = s->sym_rowbase_dev + tiles;      // [first row, first late row)
    const uint32_t *cb = s->sym_cov_begin_dev;                                      // coverage lists of the main gather
    const int nxt = s->cur ^ 1, kd = INTEG_KICK | INTEG_DRIFT;
    if (s->dims3 && s->fp64) {
        const double4 *ss = (const double4 *)s->sym_slab_s, *sr = (const double4 *)s->sym_slab_r;
        if (fuse_step)
            sym_gather3<double, true><<<gg, BLOCK, 0, s->stream>>>(ss, sr, lo, hi, cb, s->sym_cov_dev, n, 0u, n, (double4 *)dst, nullptr,
                                                                   (const double4 *)s->pos[s->cur], (double4 *)s->pos[nxt], (double4 *)s->vel, (double4 *)s->acc,
                                                                   dt, dt, kd);
        else
            sym_gather3<double, false><<<gg, BLOCK, 0, s->stream>>>(ss, sr, lo, hi, cb, s->sym_cov_dev, n, 0u, n, (double4 *)dst, nullptr,
                                                                    nullptr, nullptr, nullptr, nullptr, 0.0, 0.0, 0);
    } else if (s->dims3) {
        const float4 *ss = (const float4 *)s->sym_slab_s, *sr = (const float4 *)s->sym_slab_r;
        if (fuse_step)
            sym_gather3<float, true><<<gg, BLOCK, 0, s->stream>>>(ss, sr, lo, hi, cb, s->sym_cov_dev, n, 0u, n, (float4 *)dst, nullptr,
                                                                  (const float4 *)s->pos[s->cur], (float4 *)s->pos[nxt], (float4 *)s->vel, (float4 *)s->acc,
                                                                  (float)dt, (float)dt, kd);
        else
            sym_gather3<float, false><<<gg, BLOCK, 0, s->stream>>>(ss, sr, lo, hi, cb, s->sym_cov_dev, n, 0u, n, (float4 *)dst, nullptr,
                                                                   nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0);
    } else if (s->fp64) {
        const double2 *ss = (const double2 *)s->sym_slab_s, *sr = (const double2 *)s->sym_slab_r;
        if (fuse_step)
            sym_gather<double, true><<<gg, BLOCK, 0, s->stream>>>(ss, sr, lo, hi, cb, s->sym_cov_dev, n, 0u, n, (double2 *)dst, nullptr,
                                                                  (const double2 *)s->pos[s->cur], (double2 *)s->pos[nxt], (double2 *)s->vel, (double2 *)s->acc,
                                                                  dt, dt, s->p.extras, kd, s->sym_sb_shift);
        else
            sym_gather<double, false><<<gg, BLOCK, 0, s->stream>>>(ss, sr, lo, hi, cb, s->sym_cov_dev, n, 0u, n, (double2 *)dst, nullptr,
                                                                   nullptr, nullptr, nullptr, nullptr, 0.0, 0.0, 0, 0, s->sym_sb_shift);
    } else {
        const float2 *ss = (const float2 *)s->sym_slab_s, *sr = (const float2 *)s->sym_slab_r;
        if (fuse_step)
            sym_gather<float, true><<<gg, BLOCK, 0, s->stream>>>(ss, sr, lo, hi, cb, s->sym_cov_dev, n, 0u, n, (float2 *)dst, nullptr,
                                                                 (const float2 *)s->pos[s->cur], (float2 *)s->pos[nxt], (float2 *)s->vel, (float2 *)s->acc,
                                                                 (float)dt, (float)dt, s->p.extras, kd, s->sym_sb_shift);
        else
            sym_gather<float, false><<<gg, BLOCK, 0, s->stream>>>(ss, sr, lo, hi, cb, s->sym_cov_dev, n, 0u, n, (float2 *)dst, nullptr,
                                                                  nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0, 0, s->sym_sb_shift);
    }
    HIPCHK(hipGetLastError());
    return NB_OK;
}

// Sharded rank, after the reduce-scatter: acceleration of the owned block = acc_owned (the summed partials of all
// ranks) + the slabs of this rank's late items; kick and drift applied in the same kernel.
static int launch_sym_gather_late(nb_sim *s, double dt)
{
    const uint32_t n = (uint32_t)s->n, ic = (uint32_t)s->i_count, ib = (uint32_t)s->i_begin;
    const uint32_t per = s->dims3 ? (uint32_t)GATHER_T : (uint32_t)GATHER_P, gg = (ic + per - 1) / per, tiles = s->sym_tiles;
    const uint32_t *lo = s->sym_rowbase_dev + tiles, *hi = s->sym_rowbase_dev + 2 * (size_t)tiles;   // [first late row, end row)
    const uint32_t *cb = s->sym_cov_begin_dev + (tiles + 1);                        // coverage lists of the late segments
    const SymCov *cov = s->sym_cov_dev + s->sym_cov_late_off;
    const int nxt = s->cur ^ 1, kd = INTEG_KICK | INTEG_DRIFT;
    if (s->dims3 && s->fp64)
        sym_gather3<double, true><<<gg, BLOCK, 0, s->stream>>>((const double4 *)s->sym_slab_s, (const double4 *)s->sym_slab_r, lo, hi, cb, cov,
                                                               n, ib, ic, nullptr, (const double4 *)s->acc_owned,
                                                               (const double4 *)s->pos[s->cur], (double4 *)s->pos[nxt], (double4 *)s->vel, (double4 *)s->acc,
                                                               dt, dt, kd);
    else if (s->dims3)
        sym_gather3<float, true><<<gg, BLOCK, 0, s->stream>>>((const float4 *)s->sym_slab_s, (const float4 *)s->sym_slab_r, lo, hi, cb, cov,
                                                              n, ib, ic, nullptr, (const float4 *)s->acc_owned,
                                                              (const float4 *)s->pos[s->cur], (float4 *)s->pos[nxt], (float4 *)s->vel, (float4 *)s->acc,
                                                              (float)dt, (float)dt, kd);
    else if (s->fp64)
        sym_gather<double, true><<<gg, BLOCK, 0, s->stream>>>((const double2 *)s->sym_slab_s, (const double2 *)s->sym_slab_r, lo, hi, cb, cov,
                                                              n, ib, ic, nullptr, (const double2 *)s->acc_owned,
                                                              (const double2 *)s->pos[s->cur], (double2 *)s->pos[nxt], (double2 *)s->vel, (double2 *)s->acc,
                                                              dt, dt, s->p.extras, kd, s->sym_sb_shift);
    else
        sym_gather<float, true><<<gg, BLOCK, 0, s->stream>>>((const float2 *)s->sym_slab_s, (const float2 *)s->sym_slab_r, lo, hi, cb, cov,
                                                             n, ib, ic, nullptr, (const float2 *)s->acc_owned,
                                                             (const float2 *)s->pos[s->cur], (float2 *)s->pos[nxt], (float2 *)s->vel, (float2 *)s->acc,
                                                             (float)dt, (float)dt, s->p.extras, kd, s->sym_sb_shift);
    HIPCHK(hipGetLastError());
    return NB_OK;
}

// Whole-system symmetric force (+ optionally the kick/drift).
static int launch_force_sym(nb_sim *s, bool fuse_step = false, double dt = 0.0)
{
    int rc = launch_sym_items(s, 0, s->sym_items);
    if (rc) return rc;
    return launch_sym_gather(s, fuse_step, dt);
}

static int launch_force(nb_sim *s, const ForceJob &j)
{
    if (j.j_end <= j.j_begin || j.js == 0) return NB_OK;
    if (s->sym && &j == &s->job_all) return launch_force_sym(s);
    std::pair<hipEvent_t, hipEvent_t> pr;
    if (s->prof && prof_begin(s, &pr)) return NB_EHIP;
    const bool guard = needs_guard(s);
    const uint32_t ic = (uint32_t)s->i_count;
    if (s->dims3 && s->fp64) {
        const double eps2 = (double)s->p.eps * (double)s->p.eps;
        const uint32_t grid = grid_blocks(j.i_tiles, j.js);
        double4 *out = (double4 *)s->partial + (size_t)j.slab0 * s->i_count;
        const double4 *pos = (const double4 *)s->pos[s->cur];
#define NB_LAUNCH3D(PP, GD)                                                                                            \
        force_tiled3_f64<PP, GD, 4><<<grid, BLOCK, 0, s->stream>>>(pos, out, (uint32_t)s->i_begin, ic, j.j_begin, j.j_end, j.js, j.i_tiles, \
                                                                   eps2, j.gap_begin, j.gap_len)
        if (j.P == 2) { if (guard) NB_LAUNCH3D(2, true); else NB_LAUNCH3D(2, false); }
        else          { if (guard) NB_LAUNCH3D(1, true); else NB_LAUNCH3D(1, false); }
#undef NB_LAUNCH3D
        HIPCHK(hipGetLastError());
        if (s->prof && prof_end(s, pr)) return NB_EHIP;
        return NB_OK;
    }
    if (s->dims3) {
        const float eps2 = s->p.eps * s->p.eps;
        const uint32_t grid = grid_blocks(j.i_tiles, j.js);
        float4 *out = (float4 *)s->partial + (size_t)j.slab0 * s->i_count;
        const float4 *pos = (const float4 *)s->pos[s->cur];
        const bool quake = s->p.rsqrt_mode == NB_RSQRT_QUAKE, um = s->uniform_mass && !guard;
#define NB_LAUNCH3(PP, RQ, GD, UMM)                                                                                   \
        force_tiled3_f32<PP, RQ, GD, 8, UMM><<<grid, BLOCK, 0, s->stream>>>(pos, out, (uint32_t)s->i_begin, ic, j.j_begin, j.j_end, \
                                                                            j.js, j.i_tiles, eps2, um ? s->um_mass : 1.0f, j.gap_begin, j.gap_len)
#define NB_DISPATCH3(PP)                                                                                              \
        do {                                                                                                          \
            if (guard)      { if (quake) NB_LAUNCH3(PP, RSQ_QUAKE, true, false); else NB_LAUNCH3(PP, RSQ_EXACT, true, false); }   \
            else if (um)    { if (quake) NB_LAUNCH3(PP, RSQ_QUAKE, false, true); else NB_LAUNCH3(PP, RSQ_EXACT, false, true); }   \
            else            { if (quake) NB_LAUNCH3(PP, RSQ_QUAKE, false, false); else NB_LAUNCH3(PP, RSQ_EXACT, false, false); } \
        } while (0)
        if (j.P == 4) NB_DISPATCH3(4); else if (j.P == 2) NB_DISPATCH3(2); else NB_DISPATCH3(1);
#undef NB_DISPATCH3
#undef NB_LAUNCH3
        HIPCHK(hipGetLastError());
        if (s->prof && prof_end(s, pr)) return NB_EHIP;
        return NB_OK;
    }
    if (s->fp64) {
        const double eps2 = (double)s->p.eps * (double)s->p.eps;
        if (j.P == 2) { if (guard) launch_tiled_f64<2, true>(s, j, eps2); else launch_tiled_f64<2, false>(s, j, eps2); }
        else          { if (guard) launch_tiled_f64<1, true>(s, j, eps2); else launch_tiled_f64<1, false>(s, j, eps2); }
    } else {
        const float eps2 = s->p.eps * s->p.eps;   // Quadtree.hpp:19  e_sq(epsilon * epsilon)
        if (s->p.sum_order == NB_SUM_SEQUENTIAL) {
            const uint32_t grid = (ic + BLOCK - 1) / BLOCK;
            float2 *out = (float2 *)s->partial + (size_t)j.slab0 * s->i_count;
            if (s->p.rsqrt_mode == NB_RSQRT_QUAKE)
                force_seq_f32<RSQ_QUAKE><<<grid, BLOCK, 0, s->stream>>>((const float2 *)s->pos[s->cur], (const float *)s->mass, out,
                                                                          (uint32_t)s->i_begin, ic, j.j_begin, j.j_end, eps2);
            else
                force_seq_f32<RSQ_EXACT><<<grid, BLOCK, 0, s->stream>>>((const float2 *)s->pos[s->cur], (const float *)s->mass, out,
                                                                          (uint32_t)s->i_begin, ic, j.j_begin, j.j_end, eps2);
        } else {
            const bool quake = s->p.rsqrt_mode == NB_RSQRT_QUAKE;
#define NB_DISPATCH(PP)                                                                           \
            do {                                                                                  \
                if (quake) { if (guard) launch_tiled_f32<PP, RSQ_QUAKE, true>(s, j, eps2);  \
                             else       launch_tiled_f32<PP, RSQ_QUAKE, false>(s, j, eps2); } \
                else       { if (guard) launch_tiled_f32<PP, RSQ_EXACT, true>(s, j, eps2);  \
                             else       launch_tiled_f32<PP, RSQ_EXACT, false>(s, j, eps2); } \
            } while (0)
            if (j.P == 4) NB_DISPATCH(4); else if (j.P == 2) NB_DISPATCH(2); else NB_DISPATCH(1);
#undef NB_DISPATCH
        }
    }
    HIPCHK(hipGetLastError());
    if (s->prof && prof_end(s, pr)) return NB_EHIP;
    return NB_OK;
}

static int launch_integrate(nb_sim *s, uint32_t nslabs, double dt_kick, double dt_drift, int flags)
{
    const uint32_t ic = (uint32_t)s->i_count, g = (ic + BLOCK - 1) / BLOCK;
    const bool strict = s->p.sum_order == NB_SUM_SEQUENTIAL;
    const int nxt = s->cur ^ 1;
    if (s->dims3 && s->fp64)
        integrate3<double><<<g, BLOCK, 0, s->stream>>>((const double4 *)s->pos[s->cur], (double4 *)s->pos[nxt], (double4 *)s->vel, (double4 *)s->acc,
                                                       (const double4 *)s->partial, nslabs, (uint32_t)s->i_begin, ic, dt_kick, dt_drift, flags);
    else if (s->dims3)
        integrate3<float><<<g, BLOCK, 0, s->stream>>>((const float4 *)s->pos[s->cur], (float4 *)s->pos[nxt], (float4 *)s->vel, (float4 *)s->acc,
                                                      (const float4 *)s->partial, nslabs, (uint32_t)s->i_begin, ic, (float)dt_kick, (float)dt_drift, flags);
    else if (s->fp64)
        integrate<double, false><<<g, BLOCK, 0, s->stream>>>((const double2 *)s->pos[s->cur], (double2 *)s->pos[nxt], (double2 *)s->vel,
                                                             (double2 *)s->acc, (const double2 *)s->partial, nslabs,
                                                             (uint32_t)s->i_begin, ic, dt_kick, dt_drift, s->p.extras, flags);
    else if (strict)
        integrate<float, true><<<g, BLOCK, 0, s->stream>>>((const float2 *)s->pos[s->cur], (float2 *)s->pos[nxt], (float2 *)s->vel,
                                                           (float2 *)s->acc, (const float2 *)s->partial, nslabs,
                                                           (uint32_t)s->i_begin, ic, (float)dt_kick, (float)dt_drift, s->p.extras, flags);
    else
        integrate<float, false><<<g, BLOCK, 0, s->stream>>>((const float2 *)s->pos[s->cur], (float2 *)s->pos[nxt], (float2 *)s->vel,
                                                            (float2 *)s->acc, (const float2 *)s->partial, nslabs,
                                                            (uint32_t)s->i_begin, ic, (float)dt_kick, (float)dt_drift, s->p.extras, flags);
    HIPCHK(hipGetLastError());
    return NB_OK;
}

// ---------------------------------------------------------------------------
// stepping
// ---------------------------------------------------------------------------
static bool sharded(const nb_sim *s) { return s->i_count != s->n; }
static bool two_phase(const nb_sim *s) { return sharded(s) && s->p.sum_order != NB_SUM_SEQUENTIAL && !s->sym_sharded; }

static int step_begin_enqueue(nb_sim *s)
{
    if (s->sym_replicated) {
        // every position is already here (each rank integrates everything): all items in one launch, then this rank's
        // PARTIAL acceleration of every particle -> acc_full, which the host all-reduces before nb_step_finish
        int rc = launch_sym_items(s, 0, s->sym_items);
        if (rc) return rc;
        return launch_sym_gather(s, false, 0.0);
    }
    if (s->sym_sharded) {
        // pairs inside my own block: no remote data needed.  On the side stream (ordered after everything
        // enqueued so far), so the cross items of nb_step_mid fill the CUs its last workgroups leave idle.
        if (!s->aux_local) return launch_sym_items(s, 0, s->sym_items_local);
        HIPCHK(hipEventRecord(s->ev_fork, s->stream));
        HIPCHK(hipStreamWaitEvent(s->aux, s->ev_fork, 0));
        const int rc = launch_sym_items(s, 0, s->sym_items_local, s->aux);
        if (rc) return rc;
        HIPCHK(hipEventRecord(s->ev_join, s->aux));
        return NB_OK;
    }
    // local j-block first: its positions are already resident, so this overlaps the exchange
    if (two_phase(s)) return launch_force(s, s->job_local);
    return NB_OK;
}

extern "C" int nb_step_begin(nb_sim *s, float dt)
{
    if (!s) return nb_fail(NB_EINVAL, "nb_step_begin: NULL handle");
    if (s->in_step) return nb_fail(NB_ESTATE, "nb_step_begin: previous step not finished");
    if (s->broken) return nb_fail(NB_ESTATE, "nb_step_begin: an earlier force launch of this handle was refused by the runtime; destroy the handle");
    if (s->p.integrator != NB_INTEGRATOR_KICK_DRIFT) return nb_fail(NB_EINVAL, "nb_step_begin: split stepping is the kick-drift integrator's (a KDK handle steps with nb_step)");
    if (bind(s)) return NB_EHIP;
    s->pending_dt = dt > 0.0f ? dt : s->p.dt;
    const int rc = step_begin_enqueue(s);
    if (rc) return rc;              // the handle is NOT left "in step": the caller sees the real error and may retry or destroy it
    s->in_step = true;
    return NB_OK;
}

extern "C" int nb_step_mid(nb_sim *s)
{
    if (!s) return nb_fail(NB_EINVAL, "nb_step_mid: NULL handle");
    if (!s->in_step) return nb_fail(NB_ESTATE, "nb_step_mid: no step in flight");
    if (!s->sym_sharded) return NB_OK;                 // nothing between begin and finish in the other protocols
    if (s->mid_done) return nb_fail(NB_ESTATE, "nb_step_mid: already called for this step");
    if (bind(s)) return NB_EHIP;
    int rc = launch_sym_items(s, s->sym_items_local, s->sym_items_cross);   // cross-block pairs: need the gathered positions
    if (rc) return rc;
    if (s->sym_items_late) {
        // the held-back local items go to the side stream once the cross items are through: they run while the
        // host's reduce-scatter of acc_full is in flight and are folded in by nb_step_finish
        HIPCHK(hipEventRecord(s->ev_fork, s->stream));
        HIPCHK(hipStreamWaitEvent(s->aux, s->ev_fork, 0));
        if ((rc = launch_sym_items(s, s->sym_items_local + s->sym_items_cross, s->sym_items_late, s->aux))) return rc;
        HIPCHK(hipEventRecord(s->ev_late, s->aux));
    }
    if (s->aux_local) HIPCHK(hipStreamWaitEvent(s->stream, s->ev_join, 0));   // the local items' slabs
    if ((rc = launch_sym_gather(s, false, 0.0))) return rc;                   // partial acceleration of every particle -> acc_full
    s->mid_done = true;                                // only once everything was enqueued: a failed call can be seen and repeated
    return NB_OK;
}

extern "C" int nb_step_finish(nb_sim *s)
{
    if (!s) return nb_fail(NB_EINVAL, "nb_step_finish: NULL handle");
    if (!s->in_step) return nb_fail(NB_ESTATE, "nb_step_finish: no step in flight");
    if (bind(s)) return NB_EHIP;
    s->in_step = false;
    int rc;
    uint32_t nslabs;
    if (s->sym_replicated) {
        // acc_full holds the all-reduced acceleration of every particle: kick and drift them all
        const uint32_t nn = (uint32_t)s->n, g = (nn + BLOCK - 1) / BLOCK;
        const int nxt = s->cur ^ 1;
        const float dt = s->pending_dt;
        if (s->dims3 && s->fp64)
            integrate3<double><<<g, BLOCK, 0, s->stream>>>((const double4 *)s->pos[s->cur], (double4 *)s->pos[nxt], (double4 *)s->vel, (double4 *)s->acc,
                                                           (const double4 *)s->acc_full, 1u, 0u, nn, (double)dt, (double)dt, INTEG_KICK | INTEG_DRIFT);
        else if (s->dims3)
            integrate3<float><<<g, BLOCK, 0, s->stream>>>((const float4 *)s->pos[s->cur], (float4 *)s->pos[nxt], (float4 *)s->vel, (float4 *)s->acc,
                                                          (const float4 *)s->acc_full, 1u, 0u, nn, dt, dt, INTEG_KICK | INTEG_DRIFT);
        else if (s->fp64)
            integrate<double, false><<<g, BLOCK, 0, s->stream>>>((const double2 *)s->pos[s->cur], (double2 *)s->pos[nxt], (double2 *)s->vel,
                                                                 (double2 *)s->acc, (const double2 *)s->acc_full, 1u, 0u, nn,
                                                                 (double)dt, (double)dt, s->p.extras, INTEG_KICK | INTEG_DRIFT);
        else
            integrate<float, false><<<g, BLOCK, 0, s->stream>>>((const float2 *)s->pos[s->cur], (float2 *)s->pos[nxt], (float2 *)s->vel,
                                                                (float2 *)s->acc, (const float2 *)s->acc_full, 1u, 0u, nn, dt, dt,
                                                                s->p.extras, INTEG_KICK | INTEG_DRIFT);
        HIPCHK(hipGetLastError());
        s->cur ^= 1;
        s->frame += 1;
        s->acc_valid = false;
        return NB_OK;
    }
    if (s->sym_sharded) {
        if (!s->mid_done) { s->in_step = true; return nb_fail(NB_ESTATE, "nb_step_finish: symmetric sharded handle needs nb_step_mid (and the reduce-scatter) first"); }
        s->mid_done = false;
        // the host has reduce-scattered acc_full into acc_owned: it is the one slab of the owned block
        const uint32_t ic = (uint32_t)s->i_count, g = (ic + BLOCK - 1) / BLOCK;
        const int nxt = s->cur ^ 1;
        const float dt = s->pending_dt;
        if (s->sym_items_late) {
            HIPCHK(hipStreamWaitEvent(s->stream, s->ev_late, 0));
            if ((rc = launch_sym_gather_late(s, (double)dt))) return rc;
        } else if (s->dims3 && s->fp64)
            integrate3<double><<<g, BLOCK, 0, s->stream>>>((const double4 *)s->pos[s->cur], (double4 *)s->pos[nxt], (double4 *)s->vel, (double4 *)s->acc,
                                                           (const double4 *)s->acc_owned, 1u, (uint32_t)s->i_begin, ic, (double)dt, (double)dt,
                                                           INTEG_KICK | INTEG_DRIFT);
        else if (s->dims3)
            integrate3<float><<<g, BLOCK, 0, s->stream>>>((const float4 *)s->pos[s->cur], (float4 *)s->pos[nxt], (float4 *)s->vel, (float4 *)s->acc,
                                                          (const float4 *)s->acc_owned, 1u, (uint32_t)s->i_begin, ic, dt, dt, INTEG_KICK | INTEG_DRIFT);
        else if (s->fp64)
            integrate<double, false><<<g, BLOCK, 0, s->stream>>>((const double2 *)s->pos[s->cur], (double2 *)s->pos[nxt], (double2 *)s->vel,
                                                                 (double2 *)s->acc, (const double2 *)s->acc_owned, 1u, (uint32_t)s->i_begin, ic,
                                                                 (double)dt, (double)dt, s->p.extras, INTEG_KICK | INTEG_DRIFT);
        else
            integrate<float, false><<<g, BLOCK, 0, s->stream>>>((const float2 *)s->pos[s->cur], (float2 *)s->pos[nxt], (float2 *)s->vel,
                                                                (float2 *)s->acc, (const float2 *)s->acc_owned, 1u, (uint32_t)s->i_begin, ic, dt, dt,
                                                                s->p.extras, INTEG_KICK | INTEG_DRIFT);
        HIPCHK(hipGetLastError());
        s->cur ^= 1;
        s->frame += 1;
        return NB_OK;
    }
    if (s->sym && s->p.integrator == NB_INTEGRATOR_KICK_DRIFT) {
        // whole system, symmetric kernel: the gather launch applies the kick and the drift itself
        if ((rc = launch_force_sym(s, true, (double)s->pending_dt))) return rc;
        s->cur ^= 1;
        s->frame += 1;
        s->acc_valid = false;
        return NB_OK;
    }
    if (two_phase(s)) {
        if ((rc = launch_force(s, s->job_remote))) return rc;
        nslabs = s->slabs_two_phase;
    } else {
        if ((rc = launch_force(s, s->job_all))) return rc;
        nslabs = s->slabs_all;
    }
    const double dt = s->pending_dt;
    if ((rc = launch_integrate(s, nslabs, dt, dt, INTEG_KICK | INTEG_DRIFT))) return rc;
    s->cur ^= 1;
    s->frame += 1;                                  // Simulation.hpp:74
    s->acc_valid = false;
    return NB_OK;
}

static int step_kdk(nb_sim *s, double dt)
{
    int rc;
    if (!s->acc_valid) {                            // a(x_n), first step only
        if ((rc = launch_force(s, s->job_all))) return rc;
        if ((rc = launch_integrate(s, s->slabs_all, 0.0, 0.0, 0))) return rc;   // acc <- slabs only
    } else {
        // acc already holds a(x_n): re-present it as the single slab 0
        HIPCHK(hipMemcpyAsync(s->partial, s->acc, s->i_count * s->esz, hipMemcpyDeviceToDevice, s->stream));
    }
    // half kick + drift (acc from slab(s)), then force at x_{n+1} and the second half kick
    if ((rc = launch_integrate(s, s->acc_valid ? 1 : s->slabs_all, 0.5 * dt, dt, INTEG_KICK | INTEG_DRIFT))) return rc;
    s->cur ^= 1;
    if ((rc = launch_force(s, s->job_all))) return rc;
    if ((rc = launch_integrate(s, s->slabs_all, 0.5 * dt, 0.0, INTEG_KICK))) return rc;
    s->acc_valid = true;
    s->frame += 1;
    return NB_OK;
}

extern "C" int nb_step(nb_sim *s, float dt, int nsteps)
{
    if (!s) return nb_fail(NB_EINVAL, "nb_step: NULL handle");
    if (nsteps < 0) return nb_fail(NB_EINVAL, "nb_step: nsteps < 0");
    if (sharded(s) || s->sym_replicated || s->sym_sharded) return nb_fail(NB_ESTATE, "nb_step: sharded handle — drive it with nb_step_begin / exchange / nb_step_finish");
    if (s->in_step) return nb_fail(NB_ESTATE, "nb_step: a split step is in flight");
    if (bind(s)) return NB_EHIP;
    const float h = dt > 0.0f ? dt : s->p.dt;
    for (int k = 0; k < nsteps; ++k) {
        int rc;
        if (s->p.integrator == NB_INTEGRATOR_KDK) { if ((rc = step_kdk(s, h))) return rc; continue; }
        if ((rc = nb_step_begin(s, h))) return rc;
        if ((rc = nb_step_finish(s))) return rc;
        // keep the event list bounded on long runs
        if (s->prof && s->ev_used.size() >= 2048 && (rc = prof_collect(s))) return rc;
    }
    return NB_OK;
}

extern "C" int nb_accelerations(nb_sim *s)
{
    if (!s) return nb_fail(NB_EINVAL, "nb_accelerations: NULL handle");
    if (s->in_step) return nb_fail(NB_ESTATE, "nb_accelerations: a split step is in flight");
    if (bind(s)) return NB_EHIP;
    int rc;
    if ((rc = launch_force(s, s->job_all))) return rc;
    return launch_integrate(s, s->slabs_all, 0.0, 0.0, 0);   // acc <- sum of slabs, nothing else
}

extern "C" int nb_wait(nb_sim *s)
{
    if (!s) return nb_fail(NB_EINVAL, "nb_wait: NULL handle");
    if (bind(s)) return NB_EHIP;
    if (s->aux) HIPCHK(hipStreamSynchronize(s->aux));
    HIPCHK(hipStreamSynchronize(s->stream));
    return NB_OK;
}

// ---------------------------------------------------------------------------
// host views
// ---------------------------------------------------------------------------
static int ensure_staging(nb_sim *s)
{
    if (!s->staging) HIPCHK(hipHostMalloc(&s->staging, s->i_count * sizeof(nb_body), hipHostMallocDefault));
    return NB_OK;
}

// Page-locked host memory the library may DMA into / out of directly.  Whole pages only: a registration pins every
// page it touches, and pages shared with unrelated heap data (the first and last page of a malloc'ed array) must not be
// pinned on behalf of one array — so an unaligned range is REFUSED, not rounded (the caller does not own the rest of
// those pages).  Page-aligned storage comes from nb_host_alloc, posix_memalign / aligned_alloc or mmap.
extern "C" int nb_host_register(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return nb_fail(NB_EINVAL, "nb_host_register: NULL argument");
    const size_t ps = host_page_size();
    if (((uintptr_t)ptr % ps) != 0 || (bytes % ps) != 0)
        return nb_fail(NB_EINVAL, "nb_host_register: [%p, +%zu) is not a whole number of %zu-byte pages; registration pins whole pages, "
                                  "and the pages an unaligned array shares with its heap neighbours are not the caller's to pin "
                                  "(use nb_host_alloc, or page-aligned storage)", ptr, bytes, ps);
    if (pinned_covers(ptr, 1)) return nb_fail(NB_ESTATE, "nb_host_register: %p is already registered", ptr);
    HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    std::lock_guard<std::mutex> lock(g_pin_mutex);
    g_pinned.push_back(PinnedRange{(uintptr_t)ptr, (uintptr_t)ptr + bytes, false});
    return NB_OK;
}

// Only ranges registered through nb_host_register, by their start address, and before their storage is released.
extern "C" int nb_host_unregister(void *ptr)
{
    if (!ptr) return nb_fail(NB_EINVAL, "nb_host_unregister: NULL argument");
    {
        std::lock_guard<std::mutex> lock(g_pin_mutex);
        size_t k = 0;
        while (k < g_pinned.size() && !(g_pinned[k].lo == (uintptr_t)ptr && !g_pinned[k].owned)) ++k;
        if (k == g_pinned.size()) return nb_fail(NB_EINVAL, "nb_host_unregister: %p was not registered with nb_host_register", ptr);
        g_pinned.erase(g_pinned.begin() + (long)k);
    }
    HIPCHK(hipHostUnregister(ptr));
    return NB_OK;
}

// Page-locked, page-aligned host memory owned by the library (hipHostMalloc): the simplest destination for
// nb_sync / nb_snapshot_begin that the copy engine writes directly.  NULL on failure (nb_last_error).
extern "C" void *nb_host_alloc(size_t bytes)
{
    if (!bytes) { nb_fail(NB_EINVAL, "nb_host_alloc: zero bytes"); return nullptr; }
    if (nb_device_count() <= 0) { nb_fail(NB_ENODEVICE, "nb_host_alloc: no HIP device visible"); return nullptr; }
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); nb_fail(hip_code(e), "nb_host_alloc: hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
    std::lock_guard<std::mutex> lock(g_pin_mutex);
    g_pinned.push_back(PinnedRange{(uintptr_t)p, (uintptr_t)p + bytes, true});
    return p;
}

extern "C" int nb_host_free(void *ptr)
{
    if (!ptr) return NB_OK;
    {
        std::lock_guard<std::mutex> lock(g_pin_mutex);
        size_t k = 0;
        while (k < g_pinned.size() && !(g_pinned[k].lo == (uintptr_t)ptr && g_pinned[k].owned)) ++k;
        if (k == g_pinned.size()) return nb_fail(NB_EINVAL, "nb_host_free: %p did not come from nb_host_alloc", ptr);
        g_pinned.erase(g_pinned.begin() + (long)k);
    }
    HIPCHK(hipHostFree(ptr));
    return NB_OK;
}

// AoS view of the owned block into aos_dev, on the handle's stream
// SoA -> 64-byte Body records of `cnt` owned particles from offset `o` of the owned block, written to `out` (device memory, or
// page-locked host memory the device can address: the small-transfer path of nb_sync writes the staging buffer directly).
static int launch_pack_range(nb_sim *s, BodyRec *out, uint32_t o, uint32_t cnt)
{
    if (cnt == 0) return NB_OK;
    const uint32_t g = (cnt + BLOCK - 1) / BLOCK, ib = (uint32_t)s->i_begin + o;
    const size_t e = s->esz * (size_t)o;                 // vel / acc are indexed from the start of the owned block
    const char *vel = (const char *)s->vel + e, *acc = (const char *)s->acc + e;
    if (s->dims3 && s->fp64)
        pack_bodies3<double><<<g, BLOCK, 0, s->stream>>>(out, (const double4 *)s->pos[s->cur], (const double4 *)vel, (const double4 *)acc, s->radius, ib, cnt);
    else if (s->dims3)
        pack_bodies3<float><<<g, BLOCK, 0, s->stream>>>(out, (const float4 *)s->pos[s->cur], (const float4 *)vel, (const float4 *)acc, s->radius, ib, cnt);
    else if (s->fp64)
        pack_bodies<double><<<g, BLOCK, 0, s->stream>>>(out, (const double2 *)s->pos[s->cur], (const double *)s->mass,
                                                        (const double2 *)vel, (const double2 *)acc, s->radius, ib, cnt);
    else
        pack_bodies<float><<<g, BLOCK, 0, s->stream>>>(out, (const float2 *)s->pos[s->cur], (const float *)s->mass,
                                                       (const float2 *)vel, (const float2 *)acc, s->radius, ib, cnt);
    HIPCHK(hipGetLastError());
    return NB_OK;
}

static int launch_pack(nb_sim *s) { return launch_pack_range(s, s->aos_dev, 0, (uint32_t)s->i_count); }

extern "C" int nb_snapshot_wait(nb_sim *s)
{
    if (!s) return nb_fail(NB_EINVAL, "nb_snapshot_wait: NULL handle");
    if (!s->snap_pending) return NB_OK;
    if (bind(s)) return NB_EHIP;
    HIPCHK(hipEventSynchronize(s->ev_copied));
    if (!s->snap_direct) memcpy(s->snap_out, s->staging, s->i_count * sizeof(nb_body));
    s->snap_pending = false;
    s->snap_out = nullptr;
    return NB_OK;
}

extern "C" int nb_snapshot_begin(nb_sim *s, nb_body *out)
{
    if (!s || !out) return nb_fail(NB_EINVAL, "nb_snapshot_begin: NULL argument");
    if (s->snap_pending) return nb_fail(NB_ESTATE, "nb_snapshot_begin: a snapshot is already in flight (call nb_snapshot_wait)");
    if (s->in_step) return nb_fail(NB_ESTATE, "nb_snapshot_begin: a split step is in flight");
    if (bind(s)) return NB_EHIP;
    if (!s->copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&s->ev_packed, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s->ev_copied, hipEventDisableTiming));
    }
    s->snap_direct = pinned_covers(out, s->i_count * sizeof(nb_body));
    if (!s->snap_direct && ensure_staging(s)) return NB_EHIP;
    // aos_dev is free again: the previous snapshot was waited for, and nb_sync / nb_upload synchronise before returning
    int rc = launch_pack(s);
    if (rc) return rc;
    HIPCHK(hipEventRecord(s->ev_packed, s->stream));
    HIPCHK(hipStreamWaitEvent(s->copy_stream, s->ev_packed, 0));
    HIPCHK(hipMemcpyAsync(s->snap_direct ? (void *)out : s->staging, s->aos_dev, s->i_count * sizeof(nb_body), hipMemcpyDeviceToHost, s->copy_stream));
    HIPCHK(hipEventRecord(s->ev_copied, s->copy_stream));
    s->snap_out = out;
    s->snap_pending = true;
    return NB_OK;
}

extern "C" int nb_sync(nb_sim *s, nb_body *out)
{
    if (!s || !out) return nb_fail(NB_EINVAL, "nb_sync: NULL argument");
    if (bind(s)) return NB_EHIP;
    int rc = nb_snapshot_wait(s);                     // aos_dev / staging may still be feeding a pipelined snapshot
    if (rc) return rc;
    const size_t bytes = s->i_count * sizeof(nb_body);
    const bool direct = pinned_covers(out, bytes);
    if (!direct && ensure_staging(s)) return NB_EHIP;
    if (!direct && bytes <= SMALL_SYNC_BYTES) {
        // Small systems (the reference's own 25 000 bodies = 1.6 MB per frame, main.cpp:621-627): the copy engine's start-up costs
        // as much as the transfer.  The pack kernel writes the records STRAIGHT INTO the page-locked staging buffer over PCIe, in
        // two halves with an event each; the host copies half 0 into the caller's (pageable) array while half 1 is on the wire.
        // No DMA, no device-side AoS round trip.  (Measured on the reference's default start: profiles/r06_frames_*.)
        { const int e = ensure_bounce(s); if (e) return e; }
        const uint32_t ic = (uint32_t)s->i_count, c0 = ic < 2 * BLOCK ? ic : ((ic / 2 + BLOCK - 1) / BLOCK) * BLOCK;
        BodyRec *stage = (BodyRec *)s->staging;
        if ((rc = launch_pack_range(s, stage, 0, c0))) return rc;
        HIPCHK(hipEventRecord(s->ev_bounce[0], s->stream));
        if ((rc = launch_pack_range(s, stage + c0, c0, ic - c0))) return rc;
        HIPCHK(hipEventRecord(s->ev_bounce[1], s->stream));
        HIPCHK(hipEventSynchronize(s->ev_bounce[0]));
        memcpy(out, stage, (size_t)c0 * sizeof(nb_body));
        HIPCHK(hipEventSynchronize(s->ev_bounce[1]));
        memcpy(out + c0, stage + c0, (size_t)(ic - c0) * sizeof(nb_body));
        return NB_OK;
    }
    if ((rc = launch_pack(s))) return rc;
    // registered destination: one DMA; pageable destination: staged in pieces, the host copies piece k out while
    // piece k + 1 is still in flight
    if ((rc = copy_d2h(s, out, s->aos_dev, bytes, s->stream, direct ? nullptr : s->staging))) return rc;
    return NB_OK;
}

extern "C" int nb_sync_positions(nb_sim *s, float *out_xy)
{
    if (!s || !out_xy) return nb_fail(NB_EINVAL, "nb_sync_positions: NULL argument");
    if (bind(s)) return NB_EHIP;
    if (nb_snapshot_wait(s)) return nb_last_error_code();
    if (ensure_staging(s)) return NB_EHIP;
    const uint32_t ic = (uint32_t)s->i_count, g = (ic + BLOCK - 1) / BLOCK;
    const size_t bytes = s->i_count * (s->dims3 ? 3 : 2) * sizeof(float);       // (x, y) or (x, y, z) per body
    const void *src = s->aos_dev;
    if (s->dims3) {
        if (s->fp64) pack_positions3<double><<<g, BLOCK, 0, s->stream>>>((float *)s->aos_dev, (const double4 *)s->pos[s->cur], (uint32_t)s->i_begin, ic);
        else         pack_positions3<float><<<g, BLOCK, 0, s->stream>>>((float *)s->aos_dev, (const float4 *)s->pos[s->cur], (uint32_t)s->i_begin, ic);
        HIPCHK(hipGetLastError());
    } else if (s->fp64) {
        pack_positions<double><<<g, BLOCK, 0, s->stream>>>((float2 *)s->aos_dev, (const double2 *)s->pos[s->cur], (uint32_t)s->i_begin, ic);
        HIPCHK(hipGetLastError());
    } else {
        src = (const float2 *)s->pos[s->cur] + s->i_begin;        // fp32 2-D positions are already (x, y) floats
    }
    return copy_d2h(s, out_xy, src, bytes, s->stream, s->staging);
}

extern "C" int nb_energy(nb_sim *s, double *kinetic, double *potential)
{
    if (!s || !kinetic || !potential) return nb_fail(NB_EINVAL, "nb_energy: NULL argument");
    if (bind(s)) return NB_EHIP;
    const uint32_t g = (uint32_t)s->ered_blocks;
    const double eps2 = (double)s->p.eps * (double)s->p.eps;
    if (s->dims3 && s->fp64)
        energy_partials3<double><<<g, BLOCK, 0, s->stream>>>((const double4 *)s->pos[s->cur], (const double4 *)s->vel, (uint32_t)s->n,
                                                             (uint32_t)s->i_begin, (uint32_t)s->i_count, eps2, s->ered_dev, s->ered_dev + g);
    else if (s->dims3)
        energy_partials3<float><<<g, BLOCK, 0, s->stream>>>((const float4 *)s->pos[s->cur], (const float4 *)s->vel, (uint32_t)s->n,
                                                            (uint32_t)s->i_begin, (uint32_t)s->i_count, eps2, s->ered_dev, s->ered_dev + g);
    else if (s->fp64)
        energy_partials<double><<<g, BLOCK, 0, s->stream>>>((const double2 *)s->pos[s->cur], (const double *)s->mass, (const double2 *)s->vel,
                                                            (uint32_t)s->n, (uint32_t)s->i_begin, (uint32_t)s->i_count, eps2,
                                                            s->ered_dev, s->ered_dev + g);
    else
        energy_partials<float><<<g, BLOCK, 0, s->stream>>>((const float2 *)s->pos[s->cur], (const float *)s->mass, (const float2 *)s->vel,
                                                           (uint32_t)s->n, (uint32_t)s->i_begin, (uint32_t)s->i_count, eps2,
                                                           s->ered_dev, s->ered_dev + g);
    HIPCHK(hipGetLastError());
    std::vector<double> h(2 * (size_t)g);
    { const int rc = copy_d2h(s, h.data(), s->ered_dev, h.size() * sizeof(double)); if (rc) return rc; }
    double K = 0.0, U = 0.0;
    for (uint32_t b = 0; b < g; ++b) { K += h[b]; U += h[g + b]; }
    *kinetic = K;
    *potential = U;
    return NB_OK;
}

extern "C" int nb_momentum(nb_sim *s, double *p_xyz, double *l_z)
{
    if (!s || !p_xyz) return nb_fail(NB_EINVAL, "nb_momentum: NULL argument");
    if (bind(s)) return NB_EHIP;
    const uint32_t g = (uint32_t)s->ered_blocks;
    if (!s->pred_dev) HIPCHK(hipMalloc((void **)&s->pred_dev, 4 * (size_t)g * sizeof(double)));
    if (s->dims3 && s->fp64)
        momentum_partials3<double><<<g, BLOCK, 0, s->stream>>>((const double4 *)s->pos[s->cur], (const double4 *)s->vel, (uint32_t)s->i_begin, (uint32_t)s->i_count, s->pred_dev);
    else if (s->dims3)
        momentum_partials3<float><<<g, BLOCK, 0, s->stream>>>((const float4 *)s->pos[s->cur], (const float4 *)s->vel, (uint32_t)s->i_begin, (uint32_t)s->i_count, s->pred_dev);
    else if (s->fp64)
        momentum_partials<double><<<g, BLOCK, 0, s->stream>>>((const double2 *)s->pos[s->cur], (const double *)s->mass, (const double2 *)s->vel,
                                                              (uint32_t)s->i_begin, (uint32_t)s->i_count, s->pred_dev);
    else
        momentum_partials<float><<<g, BLOCK, 0, s->stream>>>((const float2 *)s->pos[s->cur], (const float *)s->mass, (const float2 *)s->vel,
                                                             (uint32_t)s->i_begin, (uint32_t)s->i_count, s->pred_dev);
    HIPCHK(hipGetLastError());
    std::vector<double> h(4 * (size_t)g);
    { const int rc = copy_d2h(s, h.data(), s->pred_dev, h.size() * sizeof(double)); if (rc) return rc; }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < 4; ++c)
        for (uint32_t b = 0; b < g; ++b) acc[c] += h[(size_t)c * g + b];
    p_xyz[0] = acc[0]; p_xyz[1] = acc[1]; p_xyz[2] = acc[2];
    if (l_z) *l_z = acc[3];
    return NB_OK;
}

extern "C" uint64_t nb_frame(const nb_sim *s) { return s ? s->frame : 0; }
extern "C" size_t nb_count(const nb_sim *s) { return s ? s->n : 0; }
extern "C" size_t nb_owned_begin(const nb_sim *s) { return s ? s->i_begin : 0; }
extern "C" size_t nb_owned_count(const nb_sim *s) { return s ? s->i_count : 0; }
extern "C" void *nb_pos_buffer(nb_sim *s, int which) { return s ? s->pos[which == NB_POS_NEXT ? (s->cur ^ 1) : s->cur] : nullptr; }
extern "C" void *nb_stream(nb_sim *s) { return s ? (void *)s->stream : nullptr; }
extern "C" size_t nb_pos_rows(const nb_sim *s) { return s ? s->pos_rows : 0; }
extern "C" int nb_device(const nb_sim *s) { return s ? s->dev : -1; }
extern "C" int nb_shard_rank(const nb_sim *s, int *world)
{
    if (world) *world = s ? s->p.shard_world : 0;
    return s ? s->p.shard_rank : -1;
}
extern "C" int nb_element_layout(const nb_sim *s, int *reals_per_element, int *bytes_per_real)
{
    if (!s) return nb_fail(NB_EINVAL, "nb_element_layout: NULL handle");
    if (reals_per_element) *reals_per_element = s->dims3 ? 4 : 2;
    if (bytes_per_real) *bytes_per_real = (int)s->rsz;
    return NB_OK;
}
// ---- in-process exchanges (one host process drives all the handles; no RCCL) -------------------------------------
// Everything below is STREAM-ORDERED: the handles' streams wait for each other through events, the host never blocks.
// cross_fence: every handle's stream waits for the work enqueued so far on every other handle's stream.
static int cross_fence(nb_sim *const *sims, int count, int which)
{
    for (int a = 0; a < count; ++a) {
        nb_sim *s = sims[a];
        if (bind(s)) return NB_EHIP;
        if (!s->ev_x[which]) HIPCHK(hipEventCreateWithFlags(&s->ev_x[which], hipEventDisableTiming));
        HIPCHK(hipEventRecord(s->ev_x[which], s->stream));
    }
    for (int a = 0; a < count; ++a) {
        if (bind(sims[a])) return NB_EHIP;
        for (int b = 0; b < count; ++b)
            if (b != a) HIPCHK(hipStreamWaitEvent(sims[a]->stream, sims[b]->ev_x[which], 0));
    }
    return NB_OK;
}

// Kernels of handle s read the peers' buffers directly: map them.  Only "already enabled" is not an error — on a node
// that refuses peer access the caller gets NB_EHIP here instead of a faulting kernel later.
static int enable_peers(nb_sim *s, nb_sim *const *sims, int count)
{
    for (int r = 0; r < count; ++r) {
        const int d = sims[r]->dev;
        if (d == s->dev || d >= 64 || (s->peers_enabled >> d & 1u)) continue;
        int can = 0;
        HIPCHK(hipDeviceCanAccessPeer(&can, s->dev, d));
        if (!can) return nb_fail(NB_EHIP, "device %d cannot access device %d's memory (no peer access): in-process exchange impossible", s->dev, d);
        const hipError_t e = hipDeviceEnablePeerAccess(d, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        else if (e != hipSuccess) { (void)hipGetLastError(); return nb_fail(NB_EHIP, "hipDeviceEnablePeerAccess(%d) on device %d: %s", d, s->dev, hipGetErrorString(e)); }
        s->peers_enabled |= (uint64_t)1 << d;
    }
    return NB_OK;
}

extern "C" int nb_exchange_positions(nb_sim *const *sims, int count)
{
    if (!sims || count < 1) return nb_fail(NB_EINVAL, "nb_exchange_positions: no handles");
    for (int a = 0; a < count; ++a)
        if (!sims[a] || sims[a]->n != sims[0]->n || sims[a]->esz != sims[0]->esz)
            return nb_fail(NB_EINVAL, "nb_exchange_positions: handles must shard the same system");
    int rc = cross_fence(sims, count, 0);                        // every owner's new block is complete before anyone copies it
    if (rc) return rc;
    for (int o = 0; o < count; ++o) {
        const nb_sim *own = sims[o];
        const char *src = (const char *)own->pos[own->cur] + own->i_begin * own->esz;
        for (int d = 0; d < count; ++d) {
            if (d == o) continue;
            nb_sim *dst = sims[d];
            if (bind(dst)) return NB_EHIP;
            HIPCHK(hipMemcpyPeerAsync((char *)dst->pos[dst->cur] + own->i_begin * own->esz, dst->dev, src, own->dev,
                                      own->i_count * own->esz, dst->stream));
        }
    }
    return cross_fence(sims, count, 1);                          // nobody overwrites a block that is still being copied
}

extern "C" int nb_exchange_accelerations(nb_sim *const *sims, int count)
{
    if (!sims || count < 1 || count > 64) return nb_fail(NB_EINVAL, "nb_exchange_accelerations: 1..64 handles");
    PartialPtrs src;
    for (int a = 0; a < count; ++a) {
        if (!sims[a] || !sims[a]->sym_sharded || sims[a]->n != sims[0]->n || sims[a]->esz != sims[0]->esz ||
            sims[a]->p.shard_world != count || sims[a]->p.shard_rank != a) {
            return nb_fail(NB_EINVAL, "nb_exchange_accelerations: needs the `count` handles of one symmetric sharded run, in rank order");
        }
        src.p[a] = sims[a]->acc_full;
    }
    int rc = cross_fence(sims, count, 0);                        // all partial accelerations are complete
    if (rc) return rc;
    for (int o = 0; o < count; ++o) {
        nb_sim *s = sims[o];
        if (bind(s)) return NB_EHIP;
        if ((rc = enable_peers(s, sims, count))) return rc;
        const uint32_t ic = (uint32_t)s->i_count, g = (ic + BLOCK - 1) / BLOCK;
        if (s->dims3 && s->fp64) sum_partials<double4><<<g, BLOCK, 0, s->stream>>>(src, count, (uint32_t)s->i_begin, ic, (double4 *)s->acc_owned);
        else if (s->dims3)       sum_partials<float4><<<g, BLOCK, 0, s->stream>>>(src, count, (uint32_t)s->i_begin, ic, (float4 *)s->acc_owned);
        else if (s->fp64)        sum_partials<double2><<<g, BLOCK, 0, s->stream>>>(src, count, (uint32_t)s->i_begin, ic, (double2 *)s->acc_owned);
        else                     sum_partials<float2><<<g, BLOCK, 0, s->stream>>>(src, count, (uint32_t)s->i_begin, ic, (float2 *)s->acc_owned);
        HIPCHK(hipGetLastError());
    }
    return cross_fence(sims, count, 1);                          // acc_full may be rewritten only after every peer has read it
}

extern "C" int nb_shard_protocol(const nb_sim *s)
{
    if (!s) return NB_SHARD_NONE;
    if (s->sym_replicated) return NB_SHARD_ALLREDUCE;
    if (s->sym_sharded) return NB_SHARD_SYMMETRIC;          // also the single-rank form (NB_FLAG_SHARD_SINGLE: i_count == n)
    return s->i_count == s->n ? NB_SHARD_NONE : NB_SHARD_ALLGATHER;
}

// In-process all-reduce of the replicated protocol (a host that drives all `count` handles of one run itself):
// every handle's nb_acc_buffer(0) receives the sum, in rank order, of all handles' partial accelerations — the same
// bits everywhere, so the replicas stay identical.  Call it between nb_step_begin and nb_step_finish.
extern "C" int nb_exchange_allreduce(nb_sim *const *sims, int count)
{
    if (!sims || count < 1 || count > 64) return nb_fail(NB_EINVAL, "nb_exchange_allreduce: 1..64 handles");
    PartialPtrs src;
    for (int a = 0; a < count; ++a) {
        if (!sims[a] || !sims[a]->sym_replicated || sims[a]->n != sims[0]->n || sims[a]->esz != sims[0]->esz ||
            sims[a]->p.shard_world != count || sims[a]->p.shard_rank != a)
            return nb_fail(NB_EINVAL, "nb_exchange_allreduce: needs the `count` handles of one replicated (NB_FLAG_SHARD_ALLREDUCE) run, in rank order");
        src.p[a] = sims[a]->acc_full;
    }
    int rc = cross_fence(sims, count, 0);
    if (rc) return rc;
    for (int o = 0; o < count; ++o) {               // sums into each handle's scratch (its `partial` array) ...
        nb_sim *s = sims[o];
        if (bind(s)) return NB_EHIP;
        if ((rc = enable_peers(s, sims, count))) return rc;
        const uint32_t nn = (uint32_t)s->n, g = (nn + BLOCK - 1) / BLOCK;
        if (s->dims3 && s->fp64) sum_partials<double4><<<g, BLOCK, 0, s->stream>>>(src, count, 0u, nn, (double4 *)s->partial);
        else if (s->dims3)       sum_partials<float4><<<g, BLOCK, 0, s->stream>>>(src, count, 0u, nn, (float4 *)s->partial);
        else if (s->fp64)        sum_partials<double2><<<g, BLOCK, 0, s->stream>>>(src, count, 0u, nn, (double2 *)s->partial);
        else                     sum_partials<float2><<<g, BLOCK, 0, s->stream>>>(src, count, 0u, nn, (float2 *)s->partial);
        HIPCHK(hipGetLastError());
    }
    if ((rc = cross_fence(sims, count, 1))) return rc;           // ... and only when every handle has read every input
    for (int o = 0; o < count; ++o) {               // over the inputs
        nb_sim *s = sims[o];
        if (bind(s)) return NB_EHIP;
        HIPCHK(hipMemcpyAsync(s->acc_full, s->partial, s->n * s->esz, hipMemcpyDeviceToDevice, s->stream));
    }
    return NB_OK;
}
extern "C" void *nb_acc_buffer(nb_sim *s, int which)
{
    if (!s || !(s->sym_sharded || s->sym_replicated)) return nullptr;
    return which == 0 ? s->acc_full : s->acc_owned;
}

extern "C" int nb_dump(nb_sim *s, const char *path)
{
    if (!s || !path) return nb_fail(NB_EINVAL, "nb_dump: NULL argument");
    if (sharded(s)) return nb_fail(NB_ESTATE, "nb_dump: sharded handle holds only its block; gather on the host and use nb_write_bodies");
    std::vector<nb_body> host(s->n);
    int rc = nb_sync(s, host.data());
    if (rc) return rc;
    return nb_write_bodies(path, host.data(), s->n, s->frame, &s->p);
}

// ---------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------
extern "C" int nb_profile_enable(nb_sim *s, int on)
{
    if (!s) return nb_fail(NB_EINVAL, "nb_profile_enable: NULL handle");
    s->prof = on != 0;
    return NB_OK;
}

extern "C" int nb_profile_read(nb_sim *s, double *force_ms_total, uint64_t *force_launches, int reset)
{
    if (!s) return nb_fail(NB_EINVAL, "nb_profile_read: NULL handle");
    if (bind(s)) return NB_EHIP;
    int rc = prof_collect(s);
    if (rc) return rc;
    if (force_ms_total) *force_ms_total = s->prof_ms;
    if (force_launches) *force_launches = s->prof_launches;
    if (reset) { s->prof_ms = 0.0; s->prof_launches = 0; }
    return NB_OK;
}

#ifdef NB_TEST_HOOKS   /* include/nbody_debug.h: test build only */
// Quadtree::fast_inv_sqrt (Quadtree.hpp:106-111) as the device evaluates it, on an array: y_scalar from the scalar form
// used by the sequential kernel, y_packed from the packed form of the tiled / symmetric kernels.  n must be even.
extern "C" int nb_debug_fast_inv_sqrt(const float *x, float *y_scalar, float *y_packed, size_t n)
{
    if (!x || !y_scalar || !y_packed || n == 0 || (n & 1) || n > 0x7fffff00u) return nb_fail(NB_EINVAL, "nb_debug_fast_inv_sqrt: bad arguments (n even)");
    if (nb_device_count() <= 0) return nb_fail(NB_ENODEVICE, "nb_debug_fast_inv_sqrt: no HIP device visible");
    float *dx = nullptr, *ds = nullptr, *dp = nullptr, *host = nullptr;
    int rc = NB_OK;
    hipError_t e;
    // page-locked staging: pageable caller memory is never handed to HIP (see copy_h2d)
    if ((e = hipHostMalloc((void **)&host, 3 * n * sizeof(float), hipHostMallocDefault)) != hipSuccess ||
        (e = hipMalloc((void **)&dx, n * sizeof(float))) != hipSuccess || (e = hipMalloc((void **)&ds, n * sizeof(float))) != hipSuccess ||
        (e = hipMalloc((void **)&dp, n * sizeof(float))) != hipSuccess) {
        rc = nb_fail(hip_code(e), "nb_debug_fast_inv_sqrt: allocation: %s", hipGetErrorString(e));
    } else {
        memcpy(host, x, n * sizeof(float));
        if ((e = hipMemcpy(dx, host, n * sizeof(float), hipMemcpyHostToDevice)) == hipSuccess) {
            quake_rsqrt_array<<<(unsigned)((n + BLOCK - 1) / BLOCK), BLOCK>>>(dx, ds, dp, (uint32_t)n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(host + n, ds, n * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(host + 2 * n, dp, n * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = nb_fail(hip_code(e), "nb_debug_fast_inv_sqrt: %s", hipGetErrorString(e));
        else { memcpy(y_scalar, host + n, n * sizeof(float)); memcpy(y_packed, host + 2 * n, n * sizeof(float)); }
    }
    (void)hipFree(dx); (void)hipFree(ds); (void)hipFree(dp);
    if (host) (void)hipHostFree(host);
    return rc;
}

// Test hook: move the handle's item-ticket counters (device) and the host's record of them to `value`, as if the launches so far
// had drawn that many items — so that a test can step a handle ACROSS the 2^32 wrap of the counters (580 000 steps away at
// N = 262 144) in a few steps.
extern "C" int nb_debug_ticket_seed(nb_sim *s, uint32_t value)
{
    if (!s) return nb_fail(NB_EINVAL, "nb_debug_ticket_seed: NULL handle");
    if (!s->sym_ticket) return nb_fail(NB_ESTATE, "nb_debug_ticket_seed: the handle has no symmetric plan");
    if (bind(s)) return NB_EHIP;
    if (s->aux) HIPCHK(hipStreamSynchronize(s->aux));
    HIPCHK(hipStreamSynchronize(s->stream));
    uint32_t words[3 * 32];
    memset(words, 0, sizeof words);
    for (int k = 0; k < 3; ++k) { words[k * 32] = value; s->sym_ticket_base[k] = value; }
    HIPCHK(hipMemcpy(s->sym_ticket, words, sizeof words, hipMemcpyHostToDevice));
    return NB_OK;
}
#endif

extern "C" int nb_sym_plan_info(const nb_sim *s, nb_sym_info *out)
{
    if (!s || !out) return nb_fail(NB_EINVAL, "nb_sym_plan_info: NULL argument");
    if (out->struct_size != sizeof(nb_sym_info)) return nb_fail(NB_EINVAL, "nb_sym_plan_info: out->struct_size %u != %zu", out->struct_size, sizeof(nb_sym_info));
    if (s->sym || s->sym_sharded || s->sym_replicated) *out = s->sym_info;
    else { memset(out, 0, sizeof *out); out->struct_size = (uint32_t)sizeof(nb_sym_info); out->cus = (uint32_t)s->cus; }
    return NB_OK;
}

extern "C" int nb_describe(nb_sim *s, char *buf, size_t buflen)
{
    if (!s || !buf || !buflen) return nb_fail(NB_EINVAL, "nb_describe: NULL argument");
    const ForceJob &a = s->job_all;
    const bool seq = s->p.sum_order == NB_SUM_SEQUENTIAL;
    snprintf(buf, buflen,
             "n=%zu owned=[%zu,+%zu) %s%s rsqrt=%s sum=%s | force: block=%d waves/i-set=%d i/lane=%d i_tiles=%u j_slices(all)=%u grid=%u tile_j=%d | "
             "two-phase P/slices local=%d/%u remote=%d/%u | uniform_mass=%d mass_scaled=%d mass_scaling_check=%.1e | symmetric=%d tile=%u chunk_pairs=%d items=%u chunks/item=%u late=%u slabs=%.1f+%.1f MiB | CUs=%d",
             s->n, s->i_begin, s->i_count, s->fp64 ? "fp64" : "fp32", s->dims3 ? " 3-D" : "",
             s->p.rsqrt_mode == NB_RSQRT_QUAKE ? "quake" : "exact", seq ? "sequential" : "tiled",
             BLOCK, (seq || s->fp64) ? 1 : F32_WS, seq ? 1 : (s->fp64 ? a.P : 2 * a.P), a.i_tiles, a.js,
             seq ? a.i_tiles : grid_blocks(a.i_tiles, a.js), TJ,
             s->job_local.P, s->job_local.js, s->job_remote.P, s->job_remote.js, (int)s->uniform_mass, (int)s->mass_scaled, (double)s->mass_scaling_dev,
             (int)(s->sym || s->sym_sharded || s->sym_replicated), s->sym_sb,
             (int)(s->sym_pairs && !s->mass_scaled && (!s->dims3 || s->uniform_mass || s->p.sym_chunk_pairs > 0)), s->sym_items, s->sym_L, s->sym_items_late,
             (double)s->sym_info.slab_s_bytes / 1048576.0, (double)s->sym_info.slab_r_bytes / 1048576.0, s->cus);
    return NB_OK;
}
