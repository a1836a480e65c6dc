/* include/nbody.h — C ABI of libnbody_hip.so, the MI355X (gfx950) drop-in for the
 * reference's pairwise-gravity + kick/drift path.
 *
 * The reference (7IBBE77S/nbodysim, paths relative to Nbodysim/) has no FFI: its
 * boundary is the public surface of `class Simulation`
 * (headers/Simulation.hpp:49-75): a constructor, `void step()`, and the public
 * `std::vector<Body> bodies`, driven by one caller, `simulation_thread`
 * (source/main.cpp:612-635).  Every entry point below names the piece of that
 * surface it replaces.  Plain C types only: pointers, sizes, scalars.
 *
 * Error convention (the reference has none — void returns, no exceptions):
 * int-returning functions give 0 on success and a negative NB_E* code on
 * failure; pointer-returning functions give NULL; nb_last_error() holds the
 * text and nb_last_error_code() the NB_E* code of the last failure on the
 * calling thread (so a NULL from nb_create still tells NB_ENODEVICE from
 * NB_ENOMEM from NB_EINVAL).  A handle is not thread-safe (the reference's
 * Simulation is single-caller too, main.cpp:621).
 *
 * There is NO CPU fallback: without a HIP device nb_create() fails with
 * NB_ENODEVICE.  The library reads NO environment variables: every switch is a
 * field of nb_params.
 */
#ifndef NBODY_H
#define NBODY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: what this header declares is ALL it exports (`nm -D` shows exactly these
 * nb_* symbols; tests/test_abi.py).  Test hooks live in include/nbody_debug.h and exist only in a -DNB_TEST_HOOKS build. */
#pragma GCC visibility push(default)

#define NB_ABI_VERSION 6

/* ---- particle record -------------------------------------------------------
 * Bit-compatible with the reference's `struct alignas(16) Body`
 * (headers/Body.hpp:6-13) built on `struct alignas(16) Vec2 {float x, y;}`
 * (headers/Vec2.hpp:17-20).  The source comments claim 32 bytes; the real
 * sizeof is 64 because each Vec2 is padded to 16 (measured with g++ 11.4 and
 * clang 22 — tests/golden/layout.json).  Padding is indeterminate in the
 * reference (Vec2's copy-ctor copies x,y only, Vec2.hpp:26); this library
 * always writes it as zero and never reads it. */
typedef struct nb_vec2 {
    float x, y;
    float _pad[2];
} nb_vec2;

/* z component of a 3-D handle (nb_params.dims = 3): the 8 bytes of padding that follow y in the
 * reference's alignas(16) Vec2 (Vec2.hpp:17-20). */
#define NB_Z(v) ((v)._pad[0])

typedef struct nb_body {
    nb_vec2 pos;     /* offset  0  Body::pos    */
    nb_vec2 vel;     /* offset 16  Body::vel    */
    nb_vec2 acc;     /* offset 32  Body::acc    */
    float   mass;    /* offset 48  Body::mass   */
    float   radius;  /* offset 52  Body::radius (carried, unused by gravity) */
    float   _pad[2];
} nb_body;

#if defined(__cplusplus)
static_assert(sizeof(nb_vec2) == 16, "Vec2 is 16 bytes (Vec2.hpp:17)");
static_assert(sizeof(nb_body) == 64, "Body is 64 bytes (Body.hpp:6)");
static_assert(offsetof(nb_body, vel) == 16 && offsetof(nb_body, acc) == 32 &&
              offsetof(nb_body, mass) == 48 && offsetof(nb_body, radius) == 52,
              "Body field offsets");
#else
_Static_assert(sizeof(nb_vec2) == 16, "Vec2 is 16 bytes (Vec2.hpp:17)");
_Static_assert(sizeof(nb_body) == 64, "Body is 64 bytes (Body.hpp:6)");
_Static_assert(offsetof(nb_body, vel) == 16 && offsetof(nb_body, acc) == 32 &&
               offsetof(nb_body, mass) == 48 && offsetof(nb_body, radius) == 52,
               "Body field offsets");
#endif

/* ---- enums ---------------------------------------------------------------- */
enum { NB_OK = 0, NB_EINVAL = -1, NB_ENODEVICE = -2, NB_EHIP = -3, NB_ENOMEM = -4,
       NB_EIO = -5, NB_EFORMAT = -6, NB_ESTATE = -7 };

/* arithmetic type of the device path */
enum { NB_FP32 = 0,   /* the reference's type (everything is float, Vec2.hpp:20) */
       NB_FP64 = 1 }; /* build extension (BASELINE config 5) */

/* inverse square root flavour */
enum { NB_RSQRT_EXACT = 0,   /* hardware v_rsq_f32 (fp32) / 1/sqrt (fp64): headline mode */
       NB_RSQRT_QUAKE = 1 }; /* Quadtree::fast_inv_sqrt, Quadtree.hpp:106-111: reference arithmetic */

/* order in which one particle's j-terms are summed */
enum { NB_SUM_TILED = 0,       /* LDS-tiled, packed FP32, FMA-contracted: fast path */
       NB_SUM_SEQUENTIAL = 1 };/* j ascending, one running sum, no FMA — the order of
                                  Quadtree.hpp:134-144; with NB_RSQRT_QUAKE this is
                                  bit-identical to the compiled reference */

/* nb_params.extras bit flags: the non-gravity parts of Simulation::iterate */
enum { NB_EXTRA_VCLAMP   = 1,   /* |v| <= 1000,            Simulation.hpp:133-137 */
       NB_EXTRA_BOUNDARY = 2 }; /* soft boundary + damping, Simulation.hpp:140-155 */

/* integrator */
enum { NB_INTEGRATOR_KICK_DRIFT = 0, /* Simulation.hpp:129-131,160-163 (reference) */
       NB_INTEGRATOR_KDK = 1 };      /* kick-drift-kick leapfrog (build extension) */

/* nb_params.flags bit mask: algorithm switches (all 0 = the fast defaults) */
enum { NB_FLAG_NO_SYMMETRY     = 1,   /* one-sided kernels only (every ordered pair evaluated); a sharded
                                         handle then uses the NB_SHARD_ALLGATHER protocol */
       NB_FLAG_NO_UNIFORM_MASS = 2,   /* keep the per-pair mass multiply even when all masses are equal */
       NB_FLAG_NO_GUIDED_TAIL  = 4,   /* symmetric planner: uniform work items (no finer items at the end) */
       NB_FLAG_SHARD_ALLREDUCE = 8,   /* with shard_world > 1 and i_count = n: the NB_SHARD_ALLREDUCE protocol below */
       NB_FLAG_MASS_SCALING    = 32,  /* individual masses, fp32, exact rsqrt, tiled sum, eps > 0: fold the masses into the pair
                                         geometry (nb_kernels.hip.h MM_SCALED: the travelling particle carries m^(-1/2) and its
                                         pre-multiplied position, one multiply less per pair in the symmetric kernel, none at all
                                         in the one-sided one: 2.5 % of kernel time).  The pair displacement is then no longer an exact
                                         difference of two floats (relative error 6e-8 |x_j| / |d| per pair force), and a body's pair with
                                         ITSELF no longer cancels exactly (a spurious self-acceleration of up to 6e-8 |x| m / eps^3; total
                                         momentum is conserved to that level only) — harmless for light, similar masses, several 1e-5 of the
                                         force scale for a broad mass spectrum with close heavy pairs (tests/test_headline_gpu.py).
                                         OFF BY DEFAULT (ABI 6): the same pair arithmetic on one GPU and on N, at t = 0 and later.
                                         THIS BIT: fold wherever representable (m > 0 everywhere, m_max^(3/2) / eps^3 inside the float
                                         range), whatever the data — the caller's decision; the ranks of a sharded run must agree on it */
       NB_FLAG_MASS_SCALING_MEASURED = 1024, /* fold only if harmless FOR THE BODIES UPLOADED: at nb_create / nb_upload of an unsharded
                                         handle the library evaluates their accelerations with and without the folding (two force
                                         evaluations) and folds only if they agree to 2e-6 of the force scale, max |a| (nb_describe:
                                         mass_scaled, mass_scaling_check).  A global figure at upload time: it does not bound the relative
                                         error of a body whose own |a| is far below the largest, and it is not repeated as the system
                                         evolves — the default of ABI 5, opt-in since.  The reference's own bodies (Simulation.hpp:347-603)
                                         fail it (4e-5) and keep both multiplies.  Ignored by sharded handles (no folding) */
       NB_FLAG_NO_MASS_SCALING = 512, /* never fold: both per-pair mass multiplies stay (12 + 2 instructions per body).  The default; the bit
                                         overrides the two above (a host that passes a caller's flags through and must not fold) */
       /* 64 and 128 were NB_FLAG_PIPELINE / NB_FLAG_ONE_LAUNCH_STEP of ABI 4: two in-launch fusions of the step (a persistent step
          pipeline, one launch per step) that measured level or slower than two launches per step (docs/rounds/r04.md) and were removed
          in ABI 5 — nb_create rejects the bits like any unknown one */
       NB_FLAG_STATIC_ITEMS    = 256, /* the symmetric launches normally hand out their work items dynamically (every workgroup past the first
                                         resident wave draws the next item of the list when it starts, so that the XCDs of a part, which are not
                                         equally fast, end together: -1.4 ... -2.3 % per step from 65 536 bodies up, same bits); this bit keeps
                                         item = workgroup index (A/B runs) */
       NB_FLAG_SHARD_SINGLE    = 16 };/* shard_world = 1, i_count = n: run the sharded symmetric protocol (or, with
                                         NB_FLAG_SHARD_ALLREDUCE, the replicated one) with ONE rank — every pair is "local",
                                         the reduce-scatter / all-gather degenerate to copies.  For rehearsing the exchange
                                         path (nb_comm_*, nb_exchange_*) on a single GPU; never faster than a plain handle */

/* ---- parameters ----------------------------------------------------------- */
typedef struct nb_params {
    uint32_t struct_size;  /* = sizeof(nb_params); set by nb_params_default */
    float    eps;          /* softening length; reference: 1.0f (Simulation.hpp:59 -> Quadtree.hpp:19) */
    float    dt;           /* default time step; reference: SIMULATION_DT = 0.01f (main.cpp:39) */
    int32_t  precision;    /* NB_FP32 | NB_FP64 */
    int32_t  rsqrt_mode;   /* NB_RSQRT_* */
    int32_t  sum_order;    /* NB_SUM_* */
    int32_t  integrator;   /* NB_INTEGRATOR_* */
    int32_t  extras;       /* NB_EXTRA_* bit mask; 0 = pure gravity */
    int32_t  device;       /* HIP device ordinal, -1 = current device */
    int32_t  j_slices;     /* 0 = auto; >0 forces the number of j-slices of the force grid */
    uint64_t i_begin;      /* first particle this handle integrates (sharding) */
    uint64_t i_count;      /* number of particles it integrates; 0 = all n */
    void    *stream;       /* hipStream_t to enqueue on, NULL = a private stream */
    void    *pos_buffers[2];/* optional caller-owned device buffers for the two full-n
                              position replicas (n * 2 * sizeof(real) bytes each, (x,y)
                              interleaved) so that a host-side collective can fill them;
                              NULL = allocated by the library.  ORDERING (this, acc_buffers and
                              `stream`): nb_create and nb_upload write these buffers on the handle's
                              stream, which nothing orders against the caller's other streams — so
                              when any of the three is the caller's, both calls first wait for ALL
                              work previously enqueued on the device (one hipDeviceSynchronize per
                              call, never on the step path): a fill or a collective the caller
                              queued on those buffers beforehand cannot land after the upload.
                              Both return with the upload complete.  Between steps the ordering is
                              the caller's, as described at nb_step_begin */
    int32_t  shard_rank;   /* this handle's rank and the number of ranks of a sharded run     */
    int32_t  shard_world;  /* (0/0 or x/1 = not sharded by the library's symmetric protocol)  */
    void    *acc_buffers[2];/* optional caller-owned device buffers of the symmetric sharded
                              protocol: [0] n*(ax,ay) floats = this rank's PARTIAL acceleration
                              of every particle (reduce-scatter input), [1] i_count*(ax,ay) =
                              the summed acceleration of the owned block (its output);
                              NULL = allocated by the library */
    int32_t  dims;         /* 2 (default, the reference) or 3: build extension — z of
                              pos / vel / acc lives in the first padding float of each vector
                              (macro NB_Z), sizeof of a body stays 64; tiled sum, no extras */
    int32_t  flags;        /* NB_FLAG_* bit mask */
    /* Tuning of the launch geometry; 0 = automatic everywhere (what the measurements in profiles/ chose).
     * Ranks of one sharded run must pass identical values: they select the pair split (nb_sym_info). */
    int32_t  sym_chunks_per_item; /* symmetric kernel: 64-particle chunks per work item (L) */
    int32_t  sym_aux_stream;      /* sharded symmetric: local items on a side stream: 0 auto, 1 always, -1 never */
    float    sym_late_us;         /* sharded symmetric: whole-chip microseconds of local work held back to run
                                     beside the reduce-scatter: 0 auto (40 us from 8 ranks on), < 0 none */
    int32_t  lanes_p;             /* one-sided kernel: packed particle pairs per lane (1, 2, 4) */
    float    sym_tail[3];         /* guided-tail thresholds (fractions of a launch's work from which items are cut
                                     into L/2, L/4, L/8 chunks); all 0 = 0.85, 0.94, 0.98 */
    int32_t  sym_chunk_pairs;     /* symmetric fp32 kernels: sweep the travelling chunks in PAIRS (two particles per lane, items
                                     cut into even chunk counts): 0 auto (from 65 536 bodies on; in 3-D with equal masses only),
                                     1 always, -1 never */
    uint64_t first_frame;         /* value nb_frame() starts from: 0 for a new run, the dump header's frame for a restart
                                     (the reference's Simulation::frame, Simulation.hpp:53, starts at 0: :60) */
    int32_t  sym_tile;            /* symmetric fp32 2-D kernels, single handle: stationary particles per work-item tile.  2048 = the
                                     classic form (4 waves x 512 different particles); 512 = the wave-split form (the 4 waves share 512
                                     particles and split the item's chunks: finer work units and 4-KiB slab rows for small and
                                     mid-size systems, 4x the travelling partials per pair); 0 = automatic (512 below 49 152 bodies) */
    int32_t  _reserved0;          /* keeps sizeof(nb_params) a multiple of 8 whatever the compiler; must be 0 */
    uint64_t pos_rows;            /* rows (particles) each caller-owned pos_buffers[] holds; 0 = n.  A sharded run whose world size does
                                     not divide n all-gathers ceil(n / world) rows per rank, so its replicas need world * ceil(n / world)
                                     rows: replicas the library allocates itself are sized that way, caller-owned ones must be and say so
                                     here (rows past n are never read by the kernels) */
} nb_params;

typedef struct nb_sim nb_sim; /* opaque; stands for one `Simulation` (Simulation.hpp:49) */

/* Fill *p with the reference's defaults (eps 1.0, dt 0.01, fp32, exact rsqrt, tiled sum). */
void nb_params_default(nb_params *p);

/* Replaces Simulation::Simulation() (Simulation.hpp:58-65) minus the hard-coded
 * ICs: the caller supplies the n initial bodies (AoS, 64-B records); they are
 * unpacked to SoA device arrays.  frame starts at 0. */
nb_sim *nb_create(const nb_body *init, size_t n, const nb_params *params);

/* Releases everything (the reference relies on ~Simulation via shared_ptr, main.cpp:657). */
void nb_destroy(nb_sim *s);

/* Replaces Simulation::step() (Simulation.hpp:67-75) called nsteps times with
 * time step dt (dt <= 0 -> params.dt; the reference reads the global
 * SIMULATION_DT once per step, Simulation.hpp:69).  Each step = force
 * evaluation at x_n (attract, :176) -> v += a dt -> x += v dt (iterate,
 * :129-163), ++frame (:74).  collide() (:72) is NOT performed: it is not
 * gravity and is a no-op for radius-0 bodies.  Work is ENQUEUED on
 * the handle's stream; nb_wait / nb_sync / nb_energy order after it. */
int nb_step(nb_sim *s, float dt, int nsteps);

/* Block until all enqueued work of the handle is complete. */
int nb_wait(nb_sim *s);

/* Bring the host view up to date: writes the owned block (i_begin..i_begin+i_count)
 * into out[0..i_count) — pos, vel, acc (of the last force evaluation), mass,
 * radius, padding zeroed.  This is what makes `Simulation::bodies`
 * (Simulation.hpp:54) coherent after step() for the snapshot copy at
 * main.cpp:623-627. */
int nb_sync(nb_sim *s, nb_body *out);

/* Host memory the copy engine may read / write DIRECTLY (nb_sync, nb_snapshot_begin, nb_sync_positions, nb_create,
 * nb_upload): only ranges this library knows to be page-locked over the WHOLE transfer — blocks from nb_host_alloc and
 * ranges registered here.  Any other pointer (malloc'ed arrays, `std::vector<Body>` storage — what the reference's
 * caller holds, main.cpp:623-627 — numpy arrays) is always legal and moves through the handle's page-locked staging
 * buffer in pipelined pieces (the host copy of piece k overlaps the DMA of piece k + 1).
 *
 * nb_host_register page-locks [ptr, ptr + bytes) in place.  ptr must be page-aligned and bytes a whole number of
 * pages (sysconf(_SC_PAGESIZE)); anything else is refused with NB_EINVAL: a registration pins whole pages, and the
 * first / last page of an unaligned heap array also holds its neighbours' data, which is not the caller's to pin.
 * Unregister (by the same ptr) BEFORE the storage is released; never register storage that may be reallocated
 * behind your back (a std::vector's).  nb_host_alloc / nb_host_free hand out page-locked blocks owned by the library. */
int   nb_host_register(void *ptr, size_t bytes);
int   nb_host_unregister(void *ptr);
void *nb_host_alloc(size_t bytes);
int   nb_host_free(void *ptr);

/* Positions only (8 bytes/body instead of 64): the fast path for a viewer that
 * only draws (main.cpp:623-627 consumer).  out holds 2*i_count floats (x,y) — 3*i_count
 * (x,y,z) for a dims = 3 handle.  For NB_FP64 handles values are rounded to float. */
int nb_sync_positions(nb_sim *s, float *out_xy);

/* Pipelined snapshot for a caller that does `step(); copy bodies` every frame (main.cpp:621-627) and can take the
 * copy one step late: nb_snapshot_begin packs the owned block as of the work enqueued so far and starts its D2H
 * copy on a separate copy stream, then returns; steps enqueued afterwards run concurrently with the transfer
 * (16.8 MB per frame at N = 262 144).  nb_snapshot_wait blocks until `out` is complete.  One snapshot in flight
 * per handle.  If `out` is page-locked memory known to the library (nb_host_alloc / nb_host_register) the copy engine
 * writes it directly; otherwise the data lands in the library's staging buffer and is copied to `out` inside
 * nb_snapshot_wait — which a caller issues after enqueueing the next step, so that host copy overlaps the GPU too.
 * nb_sync stays the simple blocking form. */
int nb_snapshot_begin(nb_sim *s, nb_body *out);
int nb_snapshot_wait(nb_sim *s);

/* Push host-side edits of the bodies back (the reference's `bodies` is public
 * and the GUI appends to it through SPAWN_QUEUE, main.cpp:43).  in holds the n
 * bodies of the whole system. */
int nb_upload(nb_sim *s, const nb_body *in);

/* Evaluate accelerations at the current positions without integrating
 * (Simulation::attract(), Simulation.hpp:176-214, as a direct sum). */
int nb_accelerations(nb_sim *s);

/* Total energy, fp64 accumulation on the device, softening-consistent with the
 * force (Quadtree.hpp:140-142): K = sum m v^2/2, U = -sum_{i<j} m_i m_j/sqrt(r^2+eps^2).
 * Replaces the unusable calculateMetrics (main.cpp:91-194) / Body::kinetic_energy
 * (Body.hpp:98-101).  On a sharded handle K and U are the owned block's share
 * (U of block B = - sum_{i in B} sum_{j > i} m_i m_j / sqrt(...): every unordered pair is
 * counted by the handle that owns its lower index), so the shares of all ranks add up to the total. */
int nb_energy(nb_sim *s, double *kinetic, double *potential);

/* Total linear momentum sum m v (the reference's Body::momentum, Body.hpp:103-106, summed) into p_xyz[3] (p_xyz[2] = 0
 * for a 2-D handle) and, if l_z is not NULL, the angular momentum about the origin sum m (x vy - y vx); fp64
 * accumulation on the device in a fixed order.  Both are conserved by the pairwise force (Newton's third law) up to
 * rounding, whatever the softening; a sharded handle returns its owned block's share (the shares add up). */
int nb_momentum(nb_sim *s, double *p_xyz, double *l_z);

/* Counters: Simulation::frame (Simulation.hpp:53) and sizes. */
uint64_t nb_frame(const nb_sim *s);
size_t   nb_count(const nb_sim *s);        /* n of the whole system */
size_t   nb_owned_begin(const nb_sim *s);
size_t   nb_owned_count(const nb_sim *s);

/* ---- dump / restore (build-defined: the reference has no file I/O) ----
 * File = 64-byte header {"NBODYAMD", u32 version, u32 sizeof(nb_body), u64 n,
 * u64 frame, f32 eps, f32 dt, i32 precision, i32 rsqrt_mode, zero pad} followed
 * by n raw nb_body records (padding zeroed) — the reference's only externally
 * visible state, `std::vector<Body>`. */
int nb_dump(nb_sim *s, const char *path);
/* Host-only halves of the same format (usable without a GPU). */
int nb_write_bodies(const char *path, const nb_body *bodies, size_t n, uint64_t frame,
                    const nb_params *params);
int nb_read_header(const char *path, size_t *n, uint64_t *frame, nb_params *params);
int nb_read_bodies(const char *path, nb_body *out, size_t n);

/* ---- sharded stepping (one process per GPU) ----------------------
 * A handle created with i_count < n integrates its block only and needs the
 * other blocks' positions each step.  The exchange itself is the host's
 * (torch.distributed / RCCL all-gather writing straight into the device
 * position replica), overlapped with the local-tile force:
 *
 *   nb_step_begin(s, dt)   enqueue: force from the OWNED j-block (already resident)
 *   ... host runs the all-gather into nb_pos_buffer(s, NB_POS_CURRENT) on its comm stream ...
 *   nb_step_finish(s)      enqueue: force from the remote j-blocks (one launch over
 *                          "everything but my block"), kick, drift;
 *                          new owned positions land in the NEXT replica, which
 *                          becomes CURRENT.
 * Stream ordering between the two calls and the collective is the caller's
 * (they share params.stream, or use events). */
enum { NB_POS_CURRENT = 0, NB_POS_NEXT = 1 };
/* Which exchange a sharded handle needs between nb_step_begin and nb_step_finish:
 *   NB_SHARD_ALLGATHER   (i_count < n): begin = local-tile force; host all-gathers the positions
 *                        into nb_pos_buffer(CURRENT) (may overlap begin); finish = remote force,
 *                        kick, drift.  One-sided kernels; one collective per step.
 *   NB_SHARD_SYMMETRIC   (shard_world > 1, tiled, eps >= 1e-12 (fp32) / > 0 (fp64), large n, equal blocks of whole 2048-particle
 *                        tiles): every rank evaluates 1/world of the UNORDERED pairs with the symmetric
 *                        kernel: begin = the pairs inside its own block (overlaps the all-gather still in
 *                        flight); host waits for the all-gather; nb_step_mid = its share of the cross-block
 *                        pairs, then the partial acceleration of ALL particles into nb_acc_buffer(0); host
 *                        reduce-scatters it (sum) into nb_acc_buffer(1); finish = kick, drift of the owned
 *                        block; host starts the all-gather of the new positions.  (From 8 ranks on,
 *                        nb_step_mid also starts a held-back share of the own-block pairs on a side stream;
 *                        it runs while the reduce-scatter is in flight and nb_step_finish adds its result to
 *                        nb_acc_buffer(1) — nothing changes for the host.)
 *   NB_SHARD_ALLREDUCE   (NB_FLAG_SHARD_ALLREDUCE, shard_world > 1, i_begin = 0, i_count = n, otherwise as
 *                        NB_SHARD_SYMMETRIC): every rank evaluates its 1/world of the unordered pairs and then
 *                        integrates ALL n particles itself: begin = all its items, then its partial acceleration of
 *                        every particle into nb_acc_buffer(0); the host ALL-REDUCES that buffer in place (sum; every
 *                        rank must receive the same bits, which ring / tree all-reduces deliver); finish = kick, drift of
 *                        all n.  One collective per step, no position exchange; every rank holds the whole state
 *                        (nb_sync returns all n bodies, nb_energy the total). */
enum { NB_SHARD_NONE = 0, NB_SHARD_ALLGATHER = 1, NB_SHARD_SYMMETRIC = 2, NB_SHARD_ALLREDUCE = 3 };
int   nb_shard_protocol(const nb_sim *s);
/* Exchange for a host that drives SEVERAL sharded handles from one process (e.g. one per GPU of
 * the node, no RCCL): every handle's owned block of its CURRENT replica is copied into the CURRENT
 * replica of every other handle (peer copies; the handles may sit on different devices).  Call it
 * after nb_step_finish (before the next nb_step_finish / nb_step_mid); it waits for the handles'
 * enqueued work and returns when the copies are done. */
int   nb_exchange_positions(nb_sim *const *sims, int count);
/* Same host, NB_SHARD_SYMMETRIC handles (all `count` ranks of the run, in rank order): the in-process
 * reduce-scatter — every handle's nb_acc_buffer(1) receives the sum, in rank order, of all handles'
 * partial accelerations of its block.  Call it between nb_step_mid and nb_step_finish. */
int   nb_exchange_accelerations(nb_sim *const *sims, int count);
/* Same host, NB_SHARD_ALLREDUCE handles: the in-process all-reduce — every handle's nb_acc_buffer(0) receives the sum,
 * in rank order, of all handles' partial accelerations.  Call it between nb_step_begin and nb_step_finish. */
int   nb_exchange_allreduce(nb_sim *const *sims, int count);
void *nb_acc_buffer(nb_sim *s, int which);   /* 0: full-n partial, 1: owned block sum (NULL if unused) */
int   nb_step_begin(nb_sim *s, float dt);
int   nb_step_mid(nb_sim *s);      /* NB_SHARD_SYMMETRIC only (no-op otherwise): see below */
int   nb_step_finish(nb_sim *s);
void *nb_pos_buffer(nb_sim *s, int which);   /* device pointer, n*(x,y) reals */
size_t nb_pos_rows(const nb_sim *s);         /* rows each position replica holds: >= n (padded to world * ceil(n / world) for a sharded handle) */
void *nb_stream(nb_sim *s);                  /* hipStream_t in use */

/* Element layout of the device buffers above: positions / accelerations are `reals_per_element` reals per particle
 * (2: (x, y); 4 for dims = 3: {x, y, z, m} / {ax, ay, az, 0}) of `bytes_per_real` bytes (4 or 8). */
int   nb_element_layout(const nb_sim *s, int *reals_per_element, int *bytes_per_real);
int   nb_device(const nb_sim *s);            /* HIP device ordinal the handle lives on */
int   nb_shard_rank(const nb_sim *s, int *world);   /* nb_params.shard_rank (and shard_world) the handle was created with */

/* ---- C-level multi-GPU exchange over RCCL -----------------------------------------
 * north_star / SURVEY §8e: "host code stays in C ... an RCCL all-gather of positions over xGMI each step, overlapped
 * with local-tile force compute on a second HIP stream".  Replaces the reference's only fan-out, std::async over
 * i-chunks inside one process (Simulation.hpp:180-213).  An nb_comm binds sharded handles to an RCCL communicator,
 * gives each a communication stream, and runs the step loop of their protocol (nb_shard_protocol) entirely
 * STREAM-ORDERED: force / kick / drift on the handle's stream, ncclAllGather / ncclReduceScatter / ncclAllReduce on
 * the communication stream, HIP events between the two; the host only enqueues and never blocks inside the loop.
 * Two ways to form the communicator:
 *   nb_comm_create_all   ONE process drives all ranks, one handle per device (ncclCommInitAll); the collectives of a
 *                        step are issued as one ncclGroup;
 *   nb_comm_create_rank  one process per GPU (ncclCommInitRank): rank 0 calls nb_comm_unique_id and hands the
 *                        NB_COMM_ID_BYTES to the other ranks out of band (a file, MPI, a torch.distributed broadcast).
 * Handles: created with shard_rank / shard_world, rank r owning the block [r n/world, +n/world) (equal blocks, which
 * the in-place all-gather and the reduce-scatter need; in the all-gather protocol world need not divide n: blocks of
 * ceil(n / world) particles, the last one shorter, equal counts moved over padded replicas — nb_params.pos_rows),
 * or all n particles for NB_SHARD_ALLREDUCE; an unsharded
 * handle forms a communicator of one rank (its all-gather is RCCL's one-rank no-op).  RCCL is loaded at first use
 * (librccl.so.1); without it nb_comm_create_* fail with NB_ENODEVICE — there is no other transport behind this API.
 * Like a handle, an nb_comm is driven by ONE thread, and its handles must outlive it (nb_comm_destroy first).  Between
 * nb_comm_step calls the handles may be read (nb_comm_flush, then nb_sync / nb_energy / nb_momentum) but not stepped by
 * other means.  If a call fails half-way through a step (a HIP or RCCL error), the run cannot be continued: the communicator is
 * marked failed (an open ncclGroup is closed, further nb_comm_step / _flush / _wait return NB_ESTATE) and nb_comm_destroy
 * then ABORTS the RCCL communicator (ncclCommAbort) instead of synchronising with peers that may never arrive; destroy the
 * communicator and the handles.  A FAILED communicator leaks on purpose what could block if released: its communication
 * stream always, and the RCCL communicator itself where the transport has no ncclCommAbort (ncclCommDestroy may wait for
 * the very peers that never arrived). */
typedef struct nb_comm nb_comm;
#define NB_COMM_ID_BYTES 128
int      nb_comm_unique_id(void *id_out /* NB_COMM_ID_BYTES */);
nb_comm *nb_comm_create_rank(nb_sim *s, const void *id, int rank, int world);
nb_comm *nb_comm_create_all(nb_sim *const *sims, int count);
/* Enqueue nsteps steps (dt <= 0 -> each handle's params.dt).  Returns as soon as everything is enqueued. */
int      nb_comm_step(nb_comm *c, float dt, int nsteps);
/* Make the handles' compute streams wait (on the device) for the collectives still in flight, so that nb_sync /
 * nb_energy / nb_momentum / nb_wait on a handle see complete replicas; nb_comm_wait also blocks the host until
 * everything enqueued on the compute and communication streams is done. */
int      nb_comm_flush(nb_comm *c);
int      nb_comm_wait(nb_comm *c);
void     nb_comm_destroy(nb_comm *c);        /* waits; the handles stay valid and are destroyed by their owner */
int      nb_comm_info(const nb_comm *c, int *protocol, int *world, int *local_handles, int *rccl_version);
/* NB_OK iff the transport (librccl) can be loaded in this process; every rank of a one-process-per-GPU job checks this, and
 * the ranks agree on the answer, BEFORE any of them enters the blocking communicator formation of nb_comm_create_rank. */
int      nb_comm_available(int *rccl_version);
/* Per-phase timing inside the library's loop (attribution of a sharded step when no Python driver is around to time it):
 * with profiling on, HIP events are recorded on each handle's COMPUTE stream around every compute-stream operation of the
 * schedule; nb_comm_phase_read returns, for local handle `handle`, the milliseconds summed over the steps since the last
 * reset — as the compute stream sees them, waits for the collectives included:
 *   NB_PH_LOCAL    nb_step_begin: pairs inside the own block / local j-block / (all-reduce protocol) all of the rank's pairs
 *   NB_PH_AG_WAIT  waiting for the all-gather of the previous step
 *   NB_PH_CROSS    nb_step_mid: the rank's cross-block pairs + gather of the slabs (symmetric protocol)
 *   NB_PH_REDUCE   waiting for the reduce-scatter / all-reduce of the accelerations
 *   NB_PH_FINISH   nb_step_finish: (all-gather protocol: remote blocks' force,) kick, drift
 * and the number of steps they cover.  The host runs at most 128 marked steps ahead of the device.  Off by default. */
enum { NB_PH_LOCAL = 0, NB_PH_AG_WAIT = 1, NB_PH_CROSS = 2, NB_PH_REDUCE = 3, NB_PH_FINISH = 4, NB_COMM_PHASES = 5 };
int      nb_comm_profile(nb_comm *c, int on);
int      nb_comm_phase_read(nb_comm *c, int handle, double *phase_ms /* NB_COMM_PHASES */, uint64_t *steps, int reset);
/* The RCCL id of a one-process-per-GPU launch travelling through a FILE (hosts without MPI or a torch process group):
 * rank 0 publishes {magic, nonce, id} atomically (temporary file + rename; a stale file of an earlier launch is removed
 * first), the other ranks wait for a file that carries THEIR launch's nonce — a left-over file of a crashed run, or one
 * still being replaced, is ignored — at most timeout_ms (NB_EIO after that).  Host-only (no GPU needed). */
int      nb_comm_id_publish(const char *path, uint64_t nonce, const void *id /* NB_COMM_ID_BYTES */);
int      nb_comm_id_await(const char *path, uint64_t nonce, void *id_out /* NB_COMM_ID_BYTES */, int timeout_ms);

/* ---- measurement ------------------------------------------------------------ */
/* When enabled, every force-kernel launch is bracketed by HIP events on the
 * handle's stream; nb_profile_read returns the summed kernel time and launch
 * count since the last reset (it synchronises). */
int nb_profile_enable(nb_sim *s, int on);
int nb_profile_read(nb_sim *s, double *force_ms_total, uint64_t *force_launches, int reset);

/* Describe the launch geometry chosen for the force kernel (for logs/DESIGN). */
int nb_describe(nb_sim *s, char *buf, size_t buflen);

/* ---- host utilities ----------------------------------------------------------- */
/* Synthetic workload: 3-D Plummer sphere (a = M = G = 1,
 * Aarseth-Henon-Wielen sampling, r <= 20) projected on (x,y),(vx,vy); equal
 * masses 1/n, radius 0, acc 0.  mt19937(seed) raw outputs mapped (u+0.5)/2^32. */
int nb_plummer_2d(nb_body *out, size_t n, uint32_t seed);
/* The same sample without the projection: a true 3-D Plummer sphere (z in NB_Z) for dims = 3. */
int nb_plummer_3d(nb_body *out, size_t n, uint32_t seed);
/* The bodies the reference's `Simulation()` starts from — Simulation.hpp:58-65 -> uniform_disc(n),
 * :347-603 (n = 25 000 there): 1e9 central mass, a Lorenz-attractor trace of light bodies, sorted by
 * radius, near-circular speeds; bit-identical to the compiled reference (tests/golden/default_ics.json). */
int nb_default_ics(nb_body *out, size_t n);

/* ---- the symmetric kernel's work plan ------------------------------------------ */
/* One work item (= one workgroup of force_sym_*): tile `tile` (nb_sym_info.tile_particles particles) against the 64-particle
 * chunks [c0, c0 + cnt).  s_row: row of the stationary slab it writes; the travelling partial of particle
 * j goes to element r_base + j of the travelling slab (diag items write none); group: 0 = local (pairs
 * inside the rank's own block), 1 = cross-block, 2 = late (held-back local). */
typedef struct nb_sym_item {
    uint32_t tile, c0, cnt, s_row;
    int64_t  r_base;
    uint32_t diag, group;
} nb_sym_item;

/* Figures of a plan.  Everything that must agree between the ranks of a sharded run (they split one set of
 * pairs between them) is here, so a host can verify the agreement before the first step. */
typedef struct nb_sym_info {
    uint32_t struct_size;       /* = sizeof(nb_sym_info), set by the caller */
    int32_t  enabled;           /* 1 = the handle runs the symmetric kernel */
    uint32_t chunks_per_item;   /* L */
    uint32_t items, items_local, items_cross, items_late;
    uint32_t tiles, rows_s, segments, cus;
    uint64_t units_local, units_cross, units_late;   /* (tile, chunk) units of each group */
    uint64_t cross_units_total; /* cross-block units of ALL ranks: equal on every rank of a run */
    uint64_t slab_s_bytes, slab_r_bytes;             /* the two partial-sum slab sets (written once, read once per step) */
    uint64_t coverage_entries;
    uint32_t tile_particles;    /* stationary particles per tile of this plan: 2048 (classic) or 512 (wave-split kernels) */
    uint32_t _reserved0;
} nb_sym_info;
int nb_sym_plan_info(const nb_sim *s, nb_sym_info *out);

/* Number of visible HIP devices (0 if none / runtime unavailable). */
int nb_device_count(void);

/* Text of the last error on this thread ("" if none) and its NB_E* code (NB_OK if none). */
const char *nb_last_error(void);
int nb_last_error_code(void);

/* NB_ABI_VERSION the library was built with. */
int nb_abi_version(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* NBODY_H */
