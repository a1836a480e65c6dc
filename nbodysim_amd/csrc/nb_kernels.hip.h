// nb_kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels of the hot path.
//
// Path replaced (reference, relative to Nbodysim/headers):
//   Simulation::attract()  Simulation.hpp:176-214  -> force_* kernels (direct O(N^2))
//   Quadtree::acc leaf loop Quadtree.hpp:134-144    -> the pair body
//   Quadtree::fast_inv_sqrt Quadtree.hpp:106-111    -> quake_rsqrt / RSQ_QUAKE
//   Simulation::iterate()  Simulation.hpp:129-163   -> integrate_* kernels
//
// Design: the pair body is pure VALU work (packed FP32 + v_rsq_f32); there is no
// contraction over j with i-independent operands to feed MFMA.  Kernels in this file:
//   force_sym_f32 / force_sym_f64   symmetric fast path: every unordered pair once (Newton's third
//                                   law); stationary particles in registers, 64-particle travelling
//                                   chunks rotated through the lanes with ds_bpermute_b32
//   force_tiled_f32 / force_tiled_f64   one-sided: j-particles through a double-buffered LDS tile,
//                                   one broadcast ds_read_b128 per j, 2P packed i-particles per lane,
//                                   grid- and workgroup-level j-split
//   force_seq_f32                   the reference's summation order, bit-exact parity mode
//   sym_gather / integrate          fixed-order sums of the partial slabs + kick/drift (no atomics:
//                                   results are reproducible run to run)
//   pack/unpack/energy/sum_partials AoS <-> SoA, diagnostics, in-process reduce-scatter
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nb_plan.h"   // SymItem, SymCov, SYM_SB, SYM_CH: shared with the host-side planner

namespace nbk {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int BLOCK = 256;  // 4 wave64 per workgroup
constexpr int TJ = 256;     // j-particles per LDS tile (one per thread per stage)

enum { RSQ_EXACT = 0, RSQ_QUAKE = 1 };

// Coordinates given to the padding lanes of a partial j-tile.  With mass 0 they contribute
// nothing in the general path; in the uniform-mass path (no per-pair mass multiply) they sit
// so far away that 1/r^3 = (7e-19)^3 underflows to exactly 0 while r^2 = 2e36 stays finite.
constexpr float PAD_XY = 1.0e18f;

template <typename real> struct vec2_of;
template <> struct vec2_of<float> { typedef float2 type; };
template <> struct vec2_of<double> { typedef double2 type; };

// Slab stores of the symmetric kernels.  WT = write-through (`sc1`): the bytes leave this XCD's L2 for memory at once instead of
// staying dirty there until the kernel's end flushes them — the partials are read by the NEXT launch only (force_sym_f32 below:
// -1.8 % step time at N = 25 000, -0.7 % at 65 536, neutral at 262 144, same bits; profiles/history/r04_write_through_ab.log).
// WT = false is the plain store.
template <bool WT>
__device__ __forceinline__ void store8(float2 *p, float2 v)
{
    if constexpr (WT) {
        static_assert(sizeof(float2) == sizeof(unsigned long long), "8-byte element");
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else *p = v;
}
template <bool WT>
__device__ __forceinline__ void store16(float4 *p, float4 v)
{
    if constexpr (WT) {
        // one global_store_dwordx4 sc1; the compiler does not count stores inside asm: the callers drain with s_waitcnt vmcnt(0)
        // before they signal; s_nop 1 = the wait states before the data registers may be rewritten (§5.7 of the HIP guide)
        const v4f r = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(r) : "memory");
    } else *p = v;
}

// ---------------------------------------------------------------------------
// expf_libm — the soft boundary of Simulation::iterate calls std::exp on a float (Simulation.hpp:147): glibc's expf.  The device's
// own expf is a different algorithm (bodies beyond the boundary came out 1e-7 apart: the one place a `sequential` + `quake` step was
// not bit-identical to the compiled reference).  This is glibc 2.35's algorithm restated (sysdeps/ieee754/flt-32/e_expf.c, from ARM's
// optimized routines): exp(x) = 2^(k/32) 2^(r/32) with k = round(32 x / ln 2) taken from the low bits of (z + 1.5 * 2^52), a 32-entry
// table of 2^(i/32) and a cubic in double, rounded to float once.  Pinned on the host against libm's expf, bit for bit, over 2.4e8
// floats covering the whole finite range (tests/test_oracle.py: the same statements compiled by gcc, with and without FMA
// contraction — both agree with libm everywhere, so the contraction the device compiler chooses does not matter).
// Table: asuint64(2^(i/32)) - (i << 47), correctly rounded.
// ---------------------------------------------------------------------------
__device__ static const uint64_t EXPF_TAB[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL
};

__device__ __forceinline__ float expf_libm(float x)
{
#pragma clang fp contract(off)                                  // only the three explicit fma below: z + 1.5 * 2^52 must see the ROUNDED product
    if (x != x) return x + x;
    if (x > 0x1.62e42ep6f) return __builtin_inff();          // overflow (glibc: __math_oflowf)
    if (x < -0x1.9fe368p6f) return 0.0f;                      // below the smallest subnormal
    const double z = (0x1.71547652b82fep+0 * 32.0) * (double)x;
    double kd = z + 0x1.8p+52;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= 0x1.8p+52;
    const double r = z - kd;
    const uint64_t t = EXPF_TAB[ki & 31u] + (ki << 47);
    const double sc = __longlong_as_double((long long)t);
    const double p = __builtin_fma(0x1.c6af84b912394p-5 / 32768.0, r, 0x1.ebfce50fac4f3p-3 / 1024.0);
    double y = __builtin_fma(0x1.62e42ff0c52d6p-1 / 32.0, r, 1.0);
    y = __builtin_fma(p, r * r, y);
    return (float)(y * sc);
}

__device__ __forceinline__ float exp_like_reference(float x) { return expf_libm(x); }
__device__ __forceinline__ double exp_like_reference(double x) { return exp(x); }      // fp64 handles: no reference to match

// ---------------------------------------------------------------------------
// kick_drift_one — Simulation::iterate after attract(), Simulation.hpp:129-163, for ONE owned
// particle whose summed acceleration is `a`: acc <- a; v += a dt; [clamp :133-137];
// [soft boundary :140-155]; x_next = x + v dt.  STRICT keeps every operation individually
// rounded (the reference build has no FMA contraction) for bit parity.
// flags: INTEG_KICK applies the kick (+extras), INTEG_DRIFT writes pos_next; 0 = store acc only.
// ---------------------------------------------------------------------------
enum { INTEG_KICK = 1, INTEG_DRIFT = 2 };

// v, x: the particle's velocity and position, loaded by the caller (sym_gather_block loads them before it sums, so that the
// two loads are not one more memory latency at the end of a latency-bound kernel); unused unless flags has INTEG_KICK.
template <typename real, bool STRICT>
__device__ __forceinline__
void kick_drift_loaded(typename vec2_of<real>::type a, typename vec2_of<real>::type v, const typename vec2_of<real>::type x, uint32_t li,
                       typename vec2_of<real>::type *__restrict__ pos_next,
                       typename vec2_of<real>::type *__restrict__ vel,
                       typename vec2_of<real>::type *__restrict__ acc,
                       uint32_t i_begin, real dt_kick, real dt_drift, int extras, int flags)
{
    typedef typename vec2_of<real>::type real2;
    acc[li] = a;
    if (!(flags & INTEG_KICK)) return;              // acceleration gather only
    if constexpr (STRICT) {
#pragma clang fp contract(off)
        v.x += a.x * dt_kick;                       // Simulation.hpp:130-131
        v.y += a.y * dt_kick;
    } else {
        v.x = __builtin_fma(a.x, dt_kick, v.x);
        v.y = __builtin_fma(a.y, dt_kick, v.y);
    }
    if (extras & 1) {                               // Simulation.hpp:133-137
#pragma clang fp contract(off)
        const real MAX_VELOCITY = (real)1000.0;
        const real vm = v.x * v.x + v.y * v.y;
        if (vm > MAX_VELOCITY * MAX_VELOCITY) {
            const real scale = MAX_VELOCITY / sqrt(vm);
            v.x *= scale; v.y *= scale;
        }
    }
    if (extras & 2) {                               // Simulation.hpp:140-155
#pragma clang fp contract(off)
        const real SOFT_BOUNDARY = (real)80000.0;   // 100000.0f * 0.8f
        const real d2 = x.x * x.x + x.y * x.y;
        if (d2 > SOFT_BOUNDARY * SOFT_BOUNDARY) {
            const real dist = sqrt(d2);
            const real ratio = dist / SOFT_BOUNDARY;
            const real force = (real)0.9f * exp_like_reference(ratio - (real)1.0);   // std::exp(float) = glibc's expf, restated above
            const real k = (real)-1.0 / dist;
            const real fdt = force * dt_kick;
            v.x += (x.x * k) * fdt;
            v.y += (x.y * k) * fdt;
            v.x *= (real)0.9995f;
            v.y *= (real)0.9995f;
        }
    }
    vel[li] = v;
    if (flags & INTEG_DRIFT) {
        real2 xn;
        if constexpr (STRICT) {
#pragma clang fp contract(off)
            xn.x = x.x + v.x * dt_drift;            // Simulation.hpp:161-162
            xn.y = x.y + v.y * dt_drift;
        } else {
            xn.x = __builtin_fma(v.x, dt_drift, x.x);
            xn.y = __builtin_fma(v.y, dt_drift, x.y);
        }
        pos_next[i_begin + li] = xn;
    }
}

template <typename real, bool STRICT>
__device__ __forceinline__
void kick_drift_one(typename vec2_of<real>::type a, uint32_t li,
                    const typename vec2_of<real>::type *__restrict__ pos_cur,
                    typename vec2_of<real>::type *__restrict__ pos_next,
                    typename vec2_of<real>::type *__restrict__ vel,
                    typename vec2_of<real>::type *__restrict__ acc,
                    uint32_t i_begin, real dt_kick, real dt_drift, int extras, int flags)
{
    typename vec2_of<real>::type v, x;
    v.x = v.y = x.x = x.y = 0;
    if (flags & INTEG_KICK) { v = vel[li]; x = pos_cur[i_begin + li]; }
    kick_drift_loaded<real, STRICT>(a, v, x, li, pos_next, vel, acc, i_begin, dt_kick, dt_drift, extras, flags);
}

// ---------------------------------------------------------------------------
// Block -> (i_tile, j_slice) decode.  Workgroups are dealt round-robin over the
// 8 XCDs (b % 8 shares an XCD, MI355X_MICROARCH §Workgroup dispatch): slices
// are keyed on b % 8 so that all workgroups streaming the same j-slice sit on
// one XCD and share its 4 MiB L2.  Speed only; any placement is correct.
// ---------------------------------------------------------------------------
struct TileMap { uint32_t i_tile, slice; bool valid; };

// Any js >= 1.  Slices are dealt in groups of 8 (one per XCD): with g = ceil(js/8) groups a
// workgroup b serves slice (b%8) + 8*((b/8) % g) of i-tile (b/8)/g; for js < 8 the eight XCD
// lanes are shared between 8/js' i-tiles (js' = js rounded up to 1, 2, 4).  Combinations that
// fall outside (slice >= js or i_tile >= i_tiles) are empty workgroups that exit at once.
__host__ __device__ __forceinline__ uint32_t small_js(uint32_t js) { return js <= 1 ? 1u : js <= 2 ? 2u : 4u; }

__device__ __forceinline__ TileMap decode_block(uint32_t b, uint32_t i_tiles, uint32_t js)
{
    TileMap m;
    const uint32_t xcd = b & 7u, k = b >> 3;
    if (js > 4u) {
        const uint32_t g = (js + 7u) >> 3;
        m.slice = xcd + 8u * (k % g);
        m.i_tile = k / g;
    } else {
        const uint32_t jr = small_js(js), per = 8u / jr;
        m.slice = xcd % jr;
        m.i_tile = k * per + xcd / jr;
    }
    m.valid = m.i_tile < i_tiles && m.slice < js;
    return m;
}

// grid size matching decode_block
static inline uint32_t grid_blocks(uint32_t i_tiles, uint32_t js)
{
    if (js > 4u) return i_tiles * ((js + 7u) / 8u) * 8u;
    const uint32_t per = 8u / small_js(js);
    return ((i_tiles + per - 1) / per) * 8u;
}

// Quadtree.hpp:106-111, operation order kept, no FMA contraction.
__device__ __forceinline__ float quake_rsqrt(float number)
{
#pragma clang fp contract(off)
    const float y = __uint_as_float(0x5f3759dfu - (__float_as_uint(number) >> 1));
    return y * (1.5f - (number * 0.5f * y * y));
}

// fast_inv_sqrt on an array: both device forms (scalar as in force_seq_f32, packed as in the tiled / symmetric
// kernels) — for the bit-exact check against the reference's golden grid (nb_debug_fast_inv_sqrt).
__device__ __forceinline__ v2f quake_rsqrt2(v2f t);

__device__ __forceinline__ v2f quake_rsqrt2(v2f t)
{
#pragma clang fp contract(off)
    v2f y = {__uint_as_float(0x5f3759dfu - (__float_as_uint(t.x) >> 1)),
             __uint_as_float(0x5f3759dfu - (__float_as_uint(t.y) >> 1))};
    const v2f half = {0.5f, 0.5f}, c15 = {1.5f, 1.5f};
    return y * (c15 - (t * half * y * y));
}

#ifdef NB_TEST_HOOKS   // nb_debug_fast_inv_sqrt (include/nbody_debug.h): test build only
__global__ __launch_bounds__(BLOCK)
void quake_rsqrt_array(const float *__restrict__ x, float *__restrict__ y_scalar, float *__restrict__ y_packed, uint32_t n)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    y_scalar[i] = quake_rsqrt(x[i]);
    const v2f p = quake_rsqrt2((v2f){x[i], x[i ^ 1u] });       // the neighbour rides in the other half (n is even or i^1 < n is checked by the host)
    y_packed[i] = p.x;
}
#endif


// ---------------------------------------------------------------------------
// force_tiled_f32 — the fast path.
//   pos      (x,y) interleaved, full-n replica of the positions at t_n
//   mass     full n
//   partial  [js][i_count] partial accelerations of this launch's slabs
//   i_begin/i_count   the owned block (particles this GPU integrates)
//   j_begin/j_end     the j range of this launch, cut into js slices; the range is in
//                     VIRTUAL indices that skip [gap_begin, gap_begin+gap_len): a sharded
//                     rank sweeps "everything but my own block" in one launch
// WS = how many of the workgroup's 4 waves share one set of i-particles and split each
// j-tile between them (in-workgroup j-split):
//   WS = 1  every wave owns its own i's (256 i-lanes), each wave walks the whole tile;
//   WS = 4  the 4 waves own the SAME 64 i-lanes and walk one quarter of the tile each;
//           their partial sums are combined through LDS in wave order at the end
//           (deterministic), so the workgroup still writes ONE slab row per particle.
//           Four times the workgroups for the same number of slabs: this is what keeps
//           256 CUs busy when i-particles are scarce (small N, sharded ranks) without
//           multiplying the slab traffic.
// Lane l (of i-lane group) of tile T owns particles
//   i_begin + T*IT + p*(2*LANES_I) + 2l + {0,1},  p < P,  IT = LANES_I*2P,  LANES_I = 256/WS.
// ---------------------------------------------------------------------------
template <int P, int RSQ, bool GUARD, int UNROLL, bool UM = false, int WS = 1, bool MS = false>
__device__ __forceinline__
void force_tiled_f32_body(const float2 *__restrict__ pos, const float *__restrict__ mass, const float *__restrict__ sigma,
                          float2 *__restrict__ partial,
                          uint32_t i_begin, uint32_t i_count,
                          uint32_t j_begin, uint32_t j_end,
                          uint32_t js, uint32_t i_tiles, float eps2, float um_mass = 1.0f,
                          uint32_t gap_begin = 0xffffffffu, uint32_t gap_len = 0)
{
    static_assert(WS == 1 || WS == 2 || WS == 4, "WS waves share an i-set");
    static_assert(!(MS && (UM || GUARD)), "mass scaling is for individual masses with eps > 0");
    constexpr uint32_t LANES_I = BLOCK / WS;          // distinct i-lanes in the workgroup
    constexpr uint32_t IT = LANES_I * 2 * P;          // particles per workgroup
    constexpr uint32_t JW = TJ / WS;                  // j's of a tile walked by one wave group
    constexpr uint32_t RED = (WS - 1) * P * LANES_I;  // v4f slots of the final cross-wave reduction
    constexpr uint32_t SMEM = 2 * TJ > RED ? 2 * TJ : RED;
    __shared__ v4f smem[SMEM];
    v4f (*tile)[TJ] = reinterpret_cast<v4f (*)[TJ]>(smem);

    const TileMap tm = decode_block(blockIdx.x, i_tiles, js);
    if (!tm.valid) return;
    const uint32_t t = threadIdx.x;
    const uint32_t lane_i = t % LANES_I;              // which i-lane
    const uint32_t w = t / LANES_I;                   // which share of every tile (0 when WS == 1)

    // slice bounds, multiples of TJ from j_begin
    const uint32_t jn = j_end - j_begin;
    const uint32_t slice_len = (((jn + js - 1) / js + TJ - 1) / TJ) * TJ;
    const uint32_t s0 = j_begin + min(tm.slice * slice_len, jn);
    const uint32_t s1 = j_begin + min((tm.slice + 1) * slice_len, jn);

    // i-particles of this lane
    v2f xi[P], yi[P], ax[P], ay[P];
    uint32_t li[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        li[p] = tm.i_tile * IT + (uint32_t)p * (LANES_I * 2) + 2u * lane_i;
        const uint32_t l0 = min(li[p], i_count - 1), l1 = min(li[p] + 1, i_count - 1);
        const float2 p0 = pos[i_begin + l0], p1 = pos[i_begin + l1];
        xi[p] = (v2f){p0.x, p1.x};
        yi[p] = (v2f){p0.y, p1.y};
        ax[p] = (v2f){0.f, 0.f};
        ay[p] = (v2f){0.f, 0.f};
    }
    const v2f e2 = {eps2, eps2};

    const uint32_t ntiles = (s1 - s0 + TJ - 1) / TJ;
    // One j-particle as the LDS tile holds it: {x, y, m, m}; mass-scaled (MS, see MM_SCALED at force_sym_f32): the
    // pre-multiplied position and the scaled softening {sigma x, sigma y, -sigma, sigma^2 eps^2}, sigma = m^(-1/2) —
    // the body then needs no mass multiply at all (8 + 2 instructions per two pairs, like the equal-mass form).
    auto stage = [&](uint32_t j) -> v4f {
        if (j >= s1) return MS ? (v4f){PAD_XY, PAD_XY, -1.0f, eps2} : (v4f){PAD_XY, PAD_XY, 0.f, 0.f};
        const uint32_t jg = j + (j >= gap_begin ? gap_len : 0u);
        const float2 pj = pos[jg];
        if constexpr (MS) { const float sg = sigma[jg]; return (v4f){pj.x * sg, pj.y * sg, -sg, (sg * sg) * eps2}; }
        else { const float mj = mass[jg]; return (v4f){pj.x, pj.y, mj, mj}; }
    };
    tile[0][t] = stage(s0 + t);
    __syncthreads();

    for (uint32_t it = 0; it < ntiles; ++it) {
        // prefetch the next tile into registers while this one is consumed
        const v4f nxt = stage(s0 + (it + 1) * TJ + t);

        const v4f *__restrict__ cur = tile[it & 1] + w * JW;
#pragma unroll UNROLL
        for (int jj = 0; jj < (int)JW; ++jj) {
            const v4f q = cur[jj];                 // broadcast ds_read_b128
            const v2f xj = {q.x, q.x}, yj = {q.y, q.y}, mj = {q.z, q.w};
#pragma unroll
            for (int p = 0; p < P; ++p) {
                v2f dx, dy, r2, inv;
                if constexpr (MS) {
                    const v2f ns = {q.z, q.z};
                    dx = __builtin_elementwise_fma(ns, xi[p], xj);     // sigma_j (x_j - x_i)
                    dy = __builtin_elementwise_fma(ns, yi[p], yj);
                } else {
                    dx = xj - xi[p];               // v_pk_add_f32 (neg)
                    dy = yj - yi[p];
                }
                if constexpr (MS) {
                    r2 = __builtin_elementwise_fma(dx, dx, (v2f){q.w, q.w});
                    r2 = __builtin_elementwise_fma(dy, dy, r2);
                    inv = (v2f){__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                } else if constexpr (GUARD) {
                    // eps == 0 or too small for 1/r^3 of a coincident pair to stay finite: keep the reference's
                    // `if (r_sq > 0)` around `fast_inv_sqrt(r_sq + e_sq)` (Quadtree.hpp:139-140)
                    r2 = __builtin_elementwise_fma(dy, dy, dx * dx);
                    const v2f t2 = r2 + e2;
                    if constexpr (RSQ == RSQ_EXACT)
                        inv = (v2f){__builtin_amdgcn_rsqf(t2.x), __builtin_amdgcn_rsqf(t2.y)};
                    else
                        inv = quake_rsqrt2(t2);
                    inv.x = r2.x > 0.f ? inv.x : 0.f;
                    inv.y = r2.y > 0.f ? inv.y : 0.f;
                } else {
                    r2 = __builtin_elementwise_fma(dx, dx, e2);
                    r2 = __builtin_elementwise_fma(dy, dy, r2);
                    if constexpr (RSQ == RSQ_EXACT)
                        inv = (v2f){__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                    else
                        inv = quake_rsqrt2(r2);
                }
                const v2f inv2 = inv * inv;
                v2f s;
                if constexpr (UM || MS) s = inv * inv2;     // 1 / r^3 (UM: the common mass is applied once at the end; MS: g^3 carries m_j)
                else s = (mj * inv) * inv2;                 // m / r^3
                ax[p] = __builtin_elementwise_fma(s, dx, ax[p]);
                ay[p] = __builtin_elementwise_fma(s, dy, ay[p]);
            }
        }
        if (it + 1 < ntiles) tile[(it + 1) & 1][t] = nxt;
        __syncthreads();
    }

    if constexpr (WS > 1) {
        // combine the WS partial sums of each particle in wave-group order 0,1,..,WS-1
        // (the tile buffers are free: the loop ended on a barrier)
        if (w > 0) {
#pragma unroll
            for (int p = 0; p < P; ++p)
                smem[((w - 1) * P + p) * LANES_I + lane_i] = (v4f){ax[p].x, ay[p].x, ax[p].y, ay[p].y};
        }
        __syncthreads();
        if (w > 0) return;
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int k = 0; k < WS - 1; ++k) {
                const v4f r = smem[(k * P + p) * LANES_I + lane_i];
                ax[p] += (v2f){r.x, r.z};
                ay[p] += (v2f){r.y, r.w};
            }
        }
    }

    float2 *__restrict__ out = partial + (size_t)tm.slice * i_count;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if constexpr (UM) { ax[p] *= um_mass; ay[p] *= um_mass; }
        if (li[p] + 1 < i_count) {
            *reinterpret_cast<float4 *>(&out[li[p]]) = make_float4(ax[p].x, ay[p].x, ax[p].y, ay[p].y);
        } else if (li[p] < i_count) {
            out[li[p]] = make_float2(ax[p].x, ay[p].x);
        }
    }
}

template <int P, int RSQ, bool GUARD, int UNROLL, bool UM = false, int WS = 1, bool MS = false>
__global__ __launch_bounds__(BLOCK)
void force_tiled_f32(const float2 *__restrict__ pos, const float *__restrict__ mass, const float *__restrict__ sigma,
                     float2 *__restrict__ partial,
                     uint32_t i_begin, uint32_t i_count,
                     uint32_t j_begin, uint32_t j_end,
                     uint32_t js, uint32_t i_tiles, float eps2, float um_mass,
                     uint32_t gap_begin, uint32_t gap_len)
{
    force_tiled_f32_body<P, RSQ, GUARD, UNROLL, UM, WS, MS>(pos, mass, sigma, partial, i_begin, i_count, j_begin, j_end, js, i_tiles, eps2,
                                                            um_mass, gap_begin, gap_len);
}

// ---------------------------------------------------------------------------
// force_sym_f32 — symmetric (Newton's third law) fast path: every UNORDERED pair
// is evaluated once and applied to both particles.
//
// A wave keeps 2P "stationary" particles per lane (512 per wave at P = 4, 2048 per
// workgroup) in registers and streams 64-particle "travelling" chunks through its
// lanes: one particle per lane together with its accumulator, rotated by one lane
// per step with ds_bpermute_b32 (the LDS crossbar: no VALU cost, no LDS storage),
// so after 64 steps every stationary lane has met every travelling particle and the
// travelling accumulators are back in their home lanes.  A body (two stationary
// particles x one travelling particle, BOTH directions = 4 ordered interactions)
// costs 10 packed ops + 2 v_rsq_f32 (12 + 2 with individual masses) against
// 2 x (8|9 + 2) for the one-sided kernel: ~1.6x fewer VALU cycles per interaction.
//
// Work items (built on the host, one workgroup each):
//   diagonal item  (I)         block-tile I against the chunks of its OWN 2048 particles,
//                              one-sided (each ordered pair is met from both ends anyway);
//   symmetric item (I, c0, cnt) block-tile I against cnt chunks that lie strictly AFTER
//                              tile I: stationary side accumulates in registers across the
//                              chunks, travelling side is combined over the 4 waves through
//                              LDS (wave order) and written once per chunk.
// Outputs (all plain stores, summed later in a fixed order -> deterministic):
//   slab_S[row][2048]   stationary partial of the item (row = item.s_row)
//   slab_R[r_base + j]  travelling partials: the force of tile I on particle j after it; only the
//                       particle range a (tile, group)'s items really cover is stored (nb_plan.h)
// sym_gather adds, for particle k of tile g: its slab_S rows + the slab_R entries of its tile's
// coverage list (tiles I < g).
// A rank of a sharded run holds only its share of the items (its own block's internal pairs plus
// an equal run of the cross-block items); its gather then yields a PARTIAL acceleration for every
// particle, and the ranks' partials are summed by the host's reduce-scatter.
// Requires eps > 0 (r = 0 then contributes exactly 0); eps == 0 uses force_tiled_f32<GUARD>.
// ---------------------------------------------------------------------------
#ifndef NB_SYM_UNROLL
#define NB_SYM_UNROLL 2      // rotation steps unrolled together (1 / 2 / 4 measured level: profiles/r05_first_item_vs_unroll.log)
#endif
#ifndef NB_SYM_WAVES
#define NB_SYM_WAVES 1       // __launch_bounds__ minimum waves per SIMD for force_sym_f32
#endif
#ifndef NB_SYM_P
#define NB_SYM_P 4
#endif
#ifndef NB_SYM_UNROLL2
#define NB_SYM_UNROLL2 2     // rotation steps unrolled together in sym_chunks2 (removes the register copies of the rotation: -1.4 %)
#endif
constexpr int SYM_P = NB_SYM_P;                          // packed stationary pairs per lane
constexpr int SYM_UNROLL = NB_SYM_UNROLL;
constexpr int SYM_UNROLL2 = NB_SYM_UNROLL2;
constexpr uint32_t SYM_WT = 64 * 2 * SYM_P;              // stationary particles per wave  (512)
static_assert(4 * SYM_WT == SYM_SB, "block-tile of the planner (nb_plan.h) = 4 waves x 64 lanes x 2 SYM_P particles");

__device__ __forceinline__ float lane_rot(float v, int addr)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// Mass handling of the symmetric fp32 kernels (template parameter MM):
//   MM_UNIFORM  all bodies have the same mass: the per-pair mass multiplies are hoisted out of the body
//               (10 packed + 2 v_rsq_f32 per two stationary particles x one travelling particle, both directions);
//   MM_GENERAL  individual masses, applied per pair: si = m_j / r^3, sj = m_i / r^3 (12 + 2);
//   MM_SCALED   individual masses folded into the GEOMETRY of the pair (11 + 2): the travelling particle j carries
//               sigma_j = m_j^(-1/2) and its pre-multiplied position X_j = sigma_j x_j, and the pair is evaluated in
//               coordinates stretched by sigma_j:   D = sigma_j (x_j - x_i) = fma(-sigma_j, x_i, X_j)   (the fma takes the
//               place of the subtraction),  t = |D|^2 + sigma_j^2 eps^2,  g = rsq(t) = (r^2 + eps^2)^(-1/2) / sigma_j.
//               Then g^3 D = m_j d / (r^2 + eps^2)^(3/2) IS the acceleration of the stationary particle — no multiply
//               by m_j — and the travelling side accumulates m_i g^3 D = m_j (m_i d / (...)^(3/2)), divided by m_j
//               (= multiplied by sigma_j^2) once per chunk.  One of the two per-pair mass multiplies is gone.
//               Price: D is no longer an exact difference of two floats (X_j is rounded once: relative error
//               6e-8 |x_j| / |d| of the pair's force), and m must be positive with m_max^(3/2) / eps^3 and 1 / m_min
//               inside the float range — checked at upload (nb_capi.hip: mass_scaling_ok), otherwise MM_GENERAL runs.
//               The pair of a body with ITSELF (diagonal items) does not cancel either: D = fma(-sigma, x, round(sigma x)) is
//               the rounding residue, so g^3 D is a spurious self-acceleration of up to 6e-8 |x| m / eps^3 and momentum is
//               conserved to that level only (include/nbody.h, NB_FLAG_MASS_SCALING).
enum { MM_UNIFORM = 0, MM_GENERAL = 1, MM_SCALED = 2 };

// WS (wave split, force_sym_f32<..., WS = true>): the 4 waves of the workgroup hold the SAME stationary particles and
// take the item's chunks in turn (wave w: chunks w, w + 4, ...), so a travelling partial is complete inside ONE wave
// and is stored straight from the registers: no LDS combine, no barrier per chunk.
template <int RSQ, int MM, bool DIAG, bool WS = false, bool WT = false>
__device__ __forceinline__
void sym_chunks(const float2 *__restrict__ pos, const float *__restrict__ mass, const float *__restrict__ sigma,
                float2 *__restrict__ slab_r_row, uint32_t n, uint32_t c0, uint32_t cnt,
                const v2f (&xi)[SYM_P], const v2f (&yi)[SYM_P], const v2f (&mi)[SYM_P],
                v2f (&ax)[SYM_P], v2f (&ay)[SYM_P], float eps2, float um_mass, float2 (*red)[4][64])
{
    constexpr bool UM = MM == MM_UNIFORM, MS = MM == MM_SCALED;
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    const int addr = (int)(((lane + 1u) & 63u) * 4u);    // pull from lane+1: particles move down one lane per step
    const v2f e2 = {eps2, eps2};

    // chunk c0 into registers.  MM_SCALED: (xq, yq) hold the PRE-MULTIPLIED position sigma (x, y), mq holds -sigma and
    // eq = sigma^2 eps^2; padding lanes take sigma = 1 at PAD_XY (their g^3 underflows to 0 like everywhere else)
    float xq = PAD_XY, yq = PAD_XY, mq = MS ? -1.0f : 0.f, eq = eps2;
    auto fetch = [&](uint32_t j, float &x, float &y, float &m, float &e) {
        x = PAD_XY; y = PAD_XY; m = MS ? -1.0f : 0.f; e = eps2;
        if (j < n) {
            const float2 pj = pos[j];
            if constexpr (MS) { const float sg = sigma[j]; x = pj.x * sg; y = pj.y * sg; m = -sg; e = (sg * sg) * eps2; }
            else { x = pj.x; y = pj.y; if constexpr (!UM) m = mass[j]; }
        }
    };
    constexpr uint32_t CS = WS ? 4u : 1u;              // chunk stride of this wave
    const uint32_t cfirst = WS ? w : 0u;
    fetch(cfirst < cnt ? (c0 + cfirst) * SYM_CH + lane : n, xq, yq, mq, eq);
    for (uint32_t c = cfirst; c < cnt; c += CS) {
        // next chunk in flight behind the 64 steps
        float xn, yn, mn, en;
        fetch(c + CS < cnt ? (c0 + c + CS) * SYM_CH + lane : n, xn, yn, mn, en);
        v2f aqx = {0.f, 0.f}, aqy = {0.f, 0.f};
#pragma unroll SYM_UNROLL
        for (int step = 0; step < 64; ++step) {
            // positions of the next step do not depend on this step's arithmetic: rotate them early
            const float xr = lane_rot(xq, addr), yr = lane_rot(yq, addr);
            float mr = 0.f, er = 0.f;
            if constexpr (!UM) mr = lane_rot(mq, addr);
            if constexpr (MS) er = lane_rot(eq, addr);
            const v2f xj = {xq, xq}, yj = {yq, yq};
#pragma unroll
            for (int p = 0; p < SYM_P; ++p) {
                v2f dx, dy, r2;
                if constexpr (MS) {
                    const v2f ns = {mq, mq};                               // -sigma_j
                    dx = __builtin_elementwise_fma(ns, xi[p], xj);         // sigma_j (x_j - x_i)
                    dy = __builtin_elementwise_fma(ns, yi[p], yj);
                    r2 = __builtin_elementwise_fma(dx, dx, (v2f){eq, eq});
                } else {
                    dx = xj - xi[p];
                    dy = yj - yi[p];
                    r2 = __builtin_elementwise_fma(dx, dx, e2);
                }
                r2 = __builtin_elementwise_fma(dy, dy, r2);
                v2f inv;
                if constexpr (RSQ == RSQ_EXACT) inv = (v2f){__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                else inv = quake_rsqrt2(r2);
                const v2f inv3 = inv * (inv * inv);
                if constexpr (UM || MS) {
                    ax[p] = __builtin_elementwise_fma(inv3, dx, ax[p]);
                    ay[p] = __builtin_elementwise_fma(inv3, dy, ay[p]);
                    if constexpr (!DIAG) {
                        if constexpr (MS) {
                            const v2f sj = mi[p] * inv3;                   // m_i g^3: the m_j it still carries is divided out per chunk
                            aqx = __builtin_elementwise_fma(-sj, dx, aqx);
                            aqy = __builtin_elementwise_fma(-sj, dy, aqy);
                        } else {
                            aqx = __builtin_elementwise_fma(-inv3, dx, aqx);
                            aqy = __builtin_elementwise_fma(-inv3, dy, aqy);
                        }
                    }
                } else {
                    const v2f si = (v2f){mq, mq} * inv3;       // force ON the stationary pair: m_j / r^3
                    ax[p] = __builtin_elementwise_fma(si, dx, ax[p]);
                    ay[p] = __builtin_elementwise_fma(si, dy, ay[p]);
                    if constexpr (!DIAG) {
                        const v2f sj = mi[p] * inv3;           // force ON the travelling particle: m_i / r^3
                        aqx = __builtin_elementwise_fma(-sj, dx, aqx);
                        aqy = __builtin_elementwise_fma(-sj, dy, aqy);
                    }
                }
            }
            xq = xr; yq = yr;
            if constexpr (!UM) mq = mr;
            if constexpr (MS) eq = er;
            if constexpr (!DIAG) {
                aqx.x = lane_rot(aqx.x, addr); aqx.y = lane_rot(aqx.y, addr);
                aqy.x = lane_rot(aqy.x, addr); aqy.y = lane_rot(aqy.y, addr);
            }
        }
        if constexpr (!DIAG) {
            // after 64 rotations lane l holds the accumulator of travelling particle (chunk, l):
            // combine the 4 waves (4 different stationary sets) in wave order and store once.
            float2 r = make_float2(aqx.x + aqx.y, aqy.x + aqy.y);
            if constexpr (UM) { r.x *= um_mass; r.y *= um_mass; }
            if constexpr (MS) { const float s2 = mq * mq; r.x *= s2; r.y *= s2; }      // / m_j: the particle is back in its home lane
            if constexpr (WS) {
                const uint32_t j = (c0 + c) * SYM_CH + lane;      // this wave alone met the chunk: its sum is the partial
                if (j < n) store8<WT>(&slab_r_row[j], r);
            } else {
                float2 (*rb)[64] = red[c & 1u];
                rb[w][lane] = r;
                __syncthreads();
                if (w == 0) {
                    const uint32_t j = (c0 + c) * SYM_CH + lane;
                    float2 a = rb[0][lane];
#pragma unroll
                    for (int k = 1; k < 4; ++k) { a.x += rb[k][lane].x; a.y += rb[k][lane].y; }
                    if (j < n) store8<WT>(&slab_r_row[j], a);
                }
            }
        }
        xq = xn; yq = yn;
        if constexpr (!UM) mq = mn;
        if constexpr (MS) eq = en;
    }
}

// sym_chunks2 — the same sweep with TWO travelling particles per lane (a PAIR of 64-particle chunks, c and c + 1, at a
// time).  The packed halves now hold the two travelling particles (q0, q1) and the stationary particle is the operand
// broadcast into both halves (op_sel: free), where sym_chunks packs two stationary particles against one broadcast
// travelling particle.  Same 10 + 2 (12 + 2) instructions per four ordered interactions, but
//   * one rotation step serves 16 pairs per lane instead of 8 with 8 instead of 6 ds_bpermute_b32 (x, y and the two
//     accumulator components of TWO particles ride in four register pairs): a third fewer rotations per pair, and twice
//     the VALU work between two dependent hops of the accumulators through the LDS crossbar;
//   * the travelling accumulators need no half-summing, and the 4-wave combine + barrier happens once per chunk pair.
// The stationary accumulators become per-particle pairs {from q0, from q1} (32 registers instead of 16), summed at the
// end.  An odd chunk count leaves the second half of the last pair empty (PAD particles: half of that pair's work is
// wasted), so the planner cuts items into even chunk counts for handles that run this kernel (SymTuning::even_chunks).
template <int RSQ, int MM, bool DIAG, bool WS = false, bool WT = false>
__device__ __forceinline__
void sym_chunks2(const float2 *__restrict__ pos, const float *__restrict__ mass,
                 float2 *__restrict__ slab_r_row, uint32_t n, uint32_t c0, uint32_t cnt,
                 const v2f (&xi)[SYM_P], const v2f (&yi)[SYM_P], const v2f (&mi)[SYM_P],
                 v2f (&ax)[SYM_P], v2f (&ay)[SYM_P], float eps2, float um_mass, float4 (*red)[4][64])
{
    static_assert(MM != MM_SCALED, "the mass-scaled body keeps the one-chunk form");
    constexpr bool UM = MM == MM_UNIFORM;
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    const int addr = (int)(((lane + 1u) & 63u) * 4u);
    const v2f e2 = {eps2, eps2};
    v2f bx[2 * SYM_P], by[2 * SYM_P];                  // stationary particle s = 2p + h: {partial from q0, partial from q1}
#pragma unroll
    for (int k = 0; k < 2 * SYM_P; ++k) { bx[k] = (v2f){0.f, 0.f}; by[k] = (v2f){0.f, 0.f}; }

    auto fetch = [&](uint32_t c, v2f &x, v2f &y, v2f &m) {          // chunks c and c + 1 of the item (if inside it)
        x = (v2f){PAD_XY, PAD_XY}; y = x; m = (v2f){0.f, 0.f};
        if (c < cnt) {
            const uint32_t j = (c0 + c) * SYM_CH + lane;
            if (j < n) { const float2 pj = pos[j]; x.x = pj.x; y.x = pj.y; if constexpr (!UM) m.x = mass[j]; }
        }
        if (c + 1 < cnt) {
            const uint32_t j = (c0 + c + 1) * SYM_CH + lane;
            if (j < n) { const float2 pj = pos[j]; x.y = pj.x; y.y = pj.y; if constexpr (!UM) m.y = mass[j]; }
        }
    };
    constexpr uint32_t CS = WS ? 8u : 2u;              // WS: wave w sweeps the chunk pairs 2w, 2w + 8, ...
    v2f xq, yq, mq;
    fetch(WS ? 2u * w : 0u, xq, yq, mq);
    for (uint32_t c = WS ? 2u * w : 0u; c < cnt; c += CS) {
        v2f xn, yn, mn;
        fetch(c + CS, xn, yn, mn);                     // next pair in flight behind the 64 steps
        v2f aqx = {0.f, 0.f}, aqy = {0.f, 0.f};         // {q0, q1}
#pragma unroll SYM_UNROLL2
        for (int step = 0; step < 64; ++step) {
            const v2f xr = {lane_rot(xq.x, addr), lane_rot(xq.y, addr)};
            const v2f yr = {lane_rot(yq.x, addr), lane_rot(yq.y, addr)};
            v2f mr = {0.f, 0.f};
            if constexpr (!UM) mr = (v2f){lane_rot(mq.x, addr), lane_rot(mq.y, addr)};
#pragma unroll
            for (int p = 0; p < SYM_P; ++p) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float xs = h ? xi[p].y : xi[p].x, ys = h ? yi[p].y : yi[p].x;
                    const v2f dx = xq - (v2f){xs, xs};
                    const v2f dy = yq - (v2f){ys, ys};
                    v2f r2 = __builtin_elementwise_fma(dx, dx, e2);
                    r2 = __builtin_elementwise_fma(dy, dy, r2);
                    v2f inv;
                    if constexpr (RSQ == RSQ_EXACT) inv = (v2f){__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                    else inv = quake_rsqrt2(r2);
                    const v2f inv3 = inv * (inv * inv);
                    if constexpr (UM) {
                        bx[2 * p + h] = __builtin_elementwise_fma(inv3, dx, bx[2 * p + h]);
                        by[2 * p + h] = __builtin_elementwise_fma(inv3, dy, by[2 * p + h]);
                        if constexpr (!DIAG) {
                            aqx = __builtin_elementwise_fma(-inv3, dx, aqx);
                            aqy = __builtin_elementwise_fma(-inv3, dy, aqy);
                        }
                    } else {
                        const v2f si = mq * inv3;                          // force ON the stationary particle: m_q / r^3 per half
                        bx[2 * p + h] = __builtin_elementwise_fma(si, dx, bx[2 * p + h]);
                        by[2 * p + h] = __builtin_elementwise_fma(si, dy, by[2 * p + h]);
                        if constexpr (!DIAG) {
                            const float ms = h ? mi[p].y : mi[p].x;
                            const v2f sj = (v2f){ms, ms} * inv3;           // force ON the travelling pair: m_s / r^3
                            aqx = __builtin_elementwise_fma(-sj, dx, aqx);
                            aqy = __builtin_elementwise_fma(-sj, dy, aqy);
                        }
                    }
                }
            }
            xq = xr; yq = yr;
            if constexpr (!UM) mq = mr;
            if constexpr (!DIAG) {
                aqx = (v2f){lane_rot(aqx.x, addr), lane_rot(aqx.y, addr)};
                aqy = (v2f){lane_rot(aqy.x, addr), lane_rot(aqy.y, addr)};
            }
        }
        if constexpr (!DIAG) {
            // lane l holds the accumulators of travelling particles (chunk c, l) and (chunk c + 1, l): combine the 4 waves
            // in wave order; wave 0 stores chunk c, wave 1 chunk c + 1
            float4 r = make_float4(aqx.x, aqy.x, aqx.y, aqy.y);
            if constexpr (UM) { r.x *= um_mass; r.y *= um_mass; r.z *= um_mass; r.w *= um_mass; }
            if constexpr (WS) {
                const uint32_t j0 = (c0 + c) * SYM_CH + lane, j1 = j0 + SYM_CH;
                if (j0 < n) store8<WT>(&slab_r_row[j0], make_float2(r.x, r.y));
                if (c + 1 < cnt && j1 < n) store8<WT>(&slab_r_row[j1], make_float2(r.z, r.w));
            } else {
                float4 (*rb)[64] = red[(c >> 1) & 1u];
                rb[w][lane] = r;
                __syncthreads();
                if (w < 2 && c + w < cnt) {
                    const uint32_t j = (c0 + c + w) * SYM_CH + lane;
                    float2 a = w ? make_float2(rb[0][lane].z, rb[0][lane].w) : make_float2(rb[0][lane].x, rb[0][lane].y);
#pragma unroll
                    for (int k = 1; k < 4; ++k) {
                        a.x += w ? rb[k][lane].z : rb[k][lane].x;
                        a.y += w ? rb[k][lane].w : rb[k][lane].y;
                    }
                    if (j < n) store8<WT>(&slab_r_row[j], a);
                }
            }
        }
        xq = xn; yq = yn;
        if constexpr (!UM) mq = mn;
    }
#pragma unroll
    for (int p = 0; p < SYM_P; ++p) {
        ax[p] += (v2f){bx[2 * p].x + bx[2 * p].y, bx[2 * p + 1].x + bx[2 * p + 1].y};
        ay[p] += (v2f){by[2 * p].x + by[2 * p].y, by[2 * p + 1].x + by[2 * p + 1].y};
    }
}

// WS = true is the WAVE-SPLIT form for small and mid-size systems (tiles of SYM_SB_WS = 512 particles): all 4 waves
// keep the SAME 512 stationary particles and share out the item's chunks (sym_chunks<..., WS>), so the unit of work a
// planner can place is one wave x one chunk — a quarter of the classic tile's — the travelling partials need no
// cross-wave combine (no barrier inside the sweep), and an item's stationary row is 4 KiB instead of 16.  The four
// waves' stationary sums are added once, at the end, through LDS in wave order.  Price: four times the travelling
// partials per pair (one per 512 x 64 instead of 2048 x 64 pairs), which is why large systems keep the classic form.
#ifndef NB_STAMP
#define NB_STAMP(k)          // measurement hook of tools/sym_timeline.hip (stamps inside a workgroup); nothing in the library
#endif
template <int RSQ, int MM, bool PAIRS = false, bool WS = false, bool WT = false>
__device__ __forceinline__
void force_sym_f32_body(const float2 *__restrict__ pos, const float *__restrict__ mass, const float *__restrict__ sigma,
                        const SymItem it,
                        float2 *__restrict__ slab_s, float2 *__restrict__ slab_r,
                        uint32_t n, float eps2, float um_mass)
{
    constexpr bool UM = MM == MM_UNIFORM;
    constexpr uint32_t SB = WS ? SYM_SB_WS : SYM_SB;
    static_assert(!(PAIRS && MM == MM_SCALED), "the mass-scaled body keeps the one-chunk form");
    static_assert(SYM_WT == SYM_SB_WS && (!WS || SYM_P == 4), "a wave-split tile is one wave's stationary set; wave w finishes register pair w");
    __shared__ float4 red4[WS ? 4 * SYM_P : 2 * 4][64];          // classic: [2][4][64] chunk combine; WS: [4 waves][SYM_P][64] final sums
    float2 (*red)[4][64] = reinterpret_cast<float2 (*)[4][64]>(red4);
    float4 (*red2)[4][64] = reinterpret_cast<float4 (*)[4][64]>(red4);
    const bool diag = it.diag != 0;
    const uint32_t s_row = it.s_row;
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;

    v2f xi[SYM_P], yi[SYM_P], mi[SYM_P], ax[SYM_P], ay[SYM_P];
    uint32_t li[SYM_P];
#pragma unroll
    for (int p = 0; p < SYM_P; ++p) {
        li[p] = (WS ? 0u : w * SYM_WT) + (uint32_t)p * 128u + 2u * lane;      // index inside the block-tile
        const uint32_t g0 = it.tile * SB + li[p], g1 = g0 + 1;
        // particles past the end sit at PAD_XY with mass 0: they neither feel nor exert force
        float2 p0 = make_float2(PAD_XY, PAD_XY), p1 = p0;
        float m0 = 0.f, m1 = 0.f;
        if (g0 < n) { p0 = pos[g0]; if constexpr (!UM) m0 = mass[g0]; }
        if (g1 < n) { p1 = pos[g1]; if constexpr (!UM) m1 = mass[g1]; }
        xi[p] = (v2f){p0.x, p1.x}; yi[p] = (v2f){p0.y, p1.y}; mi[p] = (v2f){m0, m1};
        ax[p] = (v2f){0.f, 0.f}; ay[p] = (v2f){0.f, 0.f};
    }
    NB_STAMP(1);                                                  // (tools/sym_timeline.hip only; empty in the product)
    float2 *__restrict__ rrow = slab_r + it.r_base;               // rrow[j] = travelling partial of particle j
    if constexpr (PAIRS) {
        if (diag) sym_chunks2<RSQ, MM, true, WS, WT>(pos, mass, rrow, n, it.c0, it.cnt, xi, yi, mi, ax, ay, eps2, um_mass, red2);
        else      sym_chunks2<RSQ, MM, false, WS, WT>(pos, mass, rrow, n, it.c0, it.cnt, xi, yi, mi, ax, ay, eps2, um_mass, red2);
    } else {
        if (diag) sym_chunks<RSQ, MM, true, WS, WT>(pos, mass, sigma, rrow, n, it.c0, it.cnt, xi, yi, mi, ax, ay, eps2, um_mass, red);
        else      sym_chunks<RSQ, MM, false, WS, WT>(pos, mass, sigma, rrow, n, it.c0, it.cnt, xi, yi, mi, ax, ay, eps2, um_mass, red);
    }

    NB_STAMP(2);
    float2 *__restrict__ out = slab_s + (size_t)s_row * SB;
    if constexpr (WS) {
        // the 4 waves hold partial sums of the SAME 512 particles (each over its own chunks): add them in wave order;
        // wave w finishes register pair p = w of every lane and stores its 128 particles (1 KiB)
#pragma unroll
        for (int p = 0; p < SYM_P; ++p) red4[w * SYM_P + p][lane] = make_float4(ax[p].x, ay[p].x, ax[p].y, ay[p].y);
        __syncthreads();
        float4 a = red4[w][lane];                                 // wave 0's sum of pair p = w
#pragma unroll
        for (int k = 1; k < 4; ++k) { const float4 b = red4[k * SYM_P + w][lane]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        if constexpr (UM) { a.x *= um_mass; a.y *= um_mass; a.z *= um_mass; a.w *= um_mass; }
        store16<WT>(reinterpret_cast<float4 *>(&out[w * 128u + 2u * lane]), a);
    } else {
#pragma unroll
        for (int p = 0; p < SYM_P; ++p) {
            if constexpr (UM) { ax[p] *= um_mass; ay[p] *= um_mass; }
            store16<WT>(reinterpret_cast<float4 *>(&out[li[p]]), make_float4(ax[p].x, ay[p].x, ax[p].y, ay[p].y));
        }
    }
}

// Which item a workgroup runs.  Workgroups are dealt round-robin over the 8 XCDs and every XCD walks ITS share in index order
// (tools/sym_timeline.hip: no inversion inside an XCD, XCDs up to 8 % of a launch apart), so with item = blockIdx every XCD gets
// the same work — and the XCDs of one part are not equally fast: the same XCDs end 2-5 % before the others launch after launch,
// whatever items they were dealt (profiles/history/r04_xcd_speed.log), and stand idle until the slowest is through.  With `ticket` set, the
// workgroups of the first wave (blockIdx < first_wave: they start together, an atomic each would only queue them up) keep their
// static item and every later workgroup draws the next item of the list when it STARTS: a faster XCD starts more workgroups and so
// takes more items.  The counter is monotonic over the handle's life (base = what earlier launches drew); any order gives the same
// bits, every item writes its own slab rows.
__device__ __forceinline__ uint32_t sym_item_index(uint32_t *ticket, uint32_t first_wave, uint32_t base)
{
    uint32_t idx = blockIdx.x;
    if (ticket != nullptr && idx >= first_wave) {             // uniform over the workgroup
        __shared__ uint32_t sh_idx;
        if (threadIdx.x == 0) sh_idx = first_wave + (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base);
        __syncthreads();
        idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh_idx);
    }
    return idx;
}

template <int RSQ, int MM, bool PAIRS = false, bool WS = false>
__global__ __launch_bounds__(BLOCK, NB_SYM_WAVES)
void force_sym_f32(const float2 *__restrict__ pos, const float *__restrict__ mass, const float *__restrict__ sigma,
                   const SymItem *__restrict__ items,
                   float2 *__restrict__ slab_s, float2 *__restrict__ slab_r,
                   uint32_t n, float eps2, float um_mass, uint32_t *ticket, uint32_t first_wave, uint32_t ticket_base)
{
#ifndef NB_SYM_WT
#define NB_SYM_WT true       // write-through (sc1) slab stores: the partials are read by the NEXT launch only, so nothing is gained by
#endif                       // keeping them dirty in this XCD's L2 until the kernel's end flushes them: -1.8 % step time at N = 25 000,
                             // -0.7 % at 65 536, neutral at 262 144, same bits (profiles/history/r04_write_through_ab.log)
    force_sym_f32_body<RSQ, MM, PAIRS, WS, NB_SYM_WT>(pos, mass, sigma, items[sym_item_index(ticket, first_wave, ticket_base)], slab_s, slab_r, n, eps2, um_mass);
}

// sigma[i] = m_i^(-1/2) for the mass-scaled kernels (correctly rounded sqrt and divide)
__global__ __launch_bounds__(BLOCK)
void mass_sigma(const float *__restrict__ mass, float *__restrict__ sigma, uint32_t n)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) sigma[i] = 1.0f / sqrtf(mass[i]);
}

// out[0] = max over particles and components of |a - b|, out[1] = max |b| (bit patterns of non-negative floats order like the
// floats, so an unsigned atomic max does; a NaN difference — bit pattern above every finite float's — wins the max and the host
// reads it as "not comparable").  The upload-time check of the mass-scaled body (nb_capi.hip: choose_mass_scaling).
__global__ __launch_bounds__(BLOCK)
void max_deviation_f32(const float2 *__restrict__ a, const float2 *__restrict__ b, uint32_t n, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    float d = 0.f, m = 0.f;
    if (i < n) {
        const float2 p = a[i], q = b[i];
        const float dx = fabsf(p.x - q.x), dy = fabsf(p.y - q.y);
        d = (dx > dy || dx != dx) ? dx : dy;              // a NaN stays
        m = fmaxf(fabsf(q.x), fabsf(q.y));
    }
    uint32_t du = __float_as_uint(d), mu = __float_as_uint(m);
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t d2 = (uint32_t)__shfl_xor((int)du, off), m2 = (uint32_t)__shfl_xor((int)mu, off);
        du = du > d2 ? du : d2;
        mu = mu > m2 ? mu : m2;
    }
    if ((threadIdx.x & 63u) == 0u) { atomicMax(&out[0], du); atomicMax(&out[1], mu); }
}

// acc_sum[k] = sum of particle k's stationary rows (its tile's items held here, in item order;
//              none if row_lo[g] == row_hi[g])
//            + sum over the coverage list of its tile (the travelling segments held here that meet the
//              tile, in segment order) of the entries that cover k.
//
// 64 particles per workgroup, 8 threads per particle PAIR: thread (q, p) adds the rows r = q (mod 8) of the
// lists of particles 2p and 2p + 1 in ascending order (one 16-byte load per row for the two float2 partials: the
// gather is the one HBM-bound kernel of the step), the 8 partials are then added in q order — a fixed association
// whatever the launch, so results are reproducible run to run.
constexpr int GATHER_Q = 8, GATHER_T = BLOCK / GATHER_Q, GATHER_V = 2, GATHER_P = GATHER_T * GATHER_V;
static_assert(SYM_SB % GATHER_P == 0 && SYM_SB_WS % GATHER_P == 0, "the particles of a gather workgroup share a tile");

// two adjacent elements (element index even: 16-byte aligned for float2)
__device__ __forceinline__ void load_pair(const float2 *__restrict__ p, float2 &a, float2 &b)
{
    const float4 v = *reinterpret_cast<const float4 *>(p);
    a = make_float2(v.x, v.y); b = make_float2(v.z, v.w);
}
__device__ __forceinline__ void load_pair(const double2 *__restrict__ p, double2 &a, double2 &b) { a = p[0]; b = p[1]; }

// FUSE: the summed acceleration goes straight into kick_drift_one (whole-system handles: the owned
// block is everything; sharded ranks: the LATE local items' slabs on top of the reduce-scattered sum
// `base`), saving the acc_sum round trip and a launch; otherwise it is stored to acc_sum (sharded ranks:
// the partial of every particle, to be reduce-scattered).
// The launch covers particles [k0, k0 + kn), k0 a multiple of the tile size 1 << sb_shift (2048, or 512 for the
// wave-split plans); tile g's stationary rows are
// [row_lo[g], row_hi[g]), its coverage entries cov[cov_begin[g] .. cov_begin[g + 1]).  Segment bounds are
// multiples of 64 (or n) and segment offsets even (nb_plan.cpp), so a pair (k, k + 1), k even, is covered together.
template <typename real, bool FUSE>
__device__ __forceinline__
void sym_gather_block(uint32_t blk,
                const typename vec2_of<real>::type *__restrict__ slab_s,
                const typename vec2_of<real>::type *__restrict__ slab_r,
                const uint32_t *__restrict__ row_lo, const uint32_t *__restrict__ row_hi,
                const uint32_t *__restrict__ cov_begin, const SymCov *__restrict__ cov,
                uint32_t n, uint32_t k0, uint32_t kn,
                typename vec2_of<real>::type *__restrict__ acc_sum,
                const typename vec2_of<real>::type *__restrict__ base,
                const typename vec2_of<real>::type *__restrict__ pos_cur,
                typename vec2_of<real>::type *__restrict__ pos_next,
                typename vec2_of<real>::type *__restrict__ vel,
                typename vec2_of<real>::type *__restrict__ acc,
                real dt_kick, real dt_drift, int extras, int flags, uint32_t sb_shift)
{
    typedef typename vec2_of<real>::type real2;
    __shared__ real2 part[GATHER_Q][GATHER_P];
    const uint32_t p = threadIdx.x % GATHER_T, q = threadIdx.x / GATHER_T;
    const uint32_t li = blk * GATHER_P + 2u * p, k = k0 + li;
    real2 a0, a1; a0.x = a0.y = a1.x = a1.y = 0;
    // thread f < GATHER_P finishes particle f of the workgroup: its velocity and position are fetched now, beside the sums
    const uint32_t f = threadIdx.x, lf = blk * GATHER_P + f;
    real2 vf, xf; vf.x = vf.y = xf.x = xf.y = 0;
    if constexpr (FUSE) { if (f < (uint32_t)GATHER_P && lf < kn && (flags & INTEG_KICK)) { vf = vel[lf]; xf = pos_cur[k0 + lf]; } }
    if (li < kn) {
        const uint32_t g = k >> sb_shift, loc = k & ((1u << sb_shift) - 1u);
        const uint32_t r0 = row_lo[g], r1 = row_hi[g];
#pragma unroll 4      // four independent 16-byte loads in flight per thread; the adds keep their order
        for (uint32_t r = r0 + q; r < r1; r += GATHER_Q) {
            real2 b0, b1;
            load_pair(slab_s + ((size_t)r << sb_shift) + loc, b0, b1);   // rows are whole tiles: loc + 1 is inside
            a0.x += b0.x; a0.y += b0.y; a1.x += b1.x; a1.y += b1.y;
        }
        // Coverage entries, four at a time: the four descriptors, then the four partial pairs they point at, are loaded as
        // independent batches (entry by entry the loop was a chain of two dependent memory latencies per entry: the last tiles
        // of a plan — the longest lists — set the duration of the whole gather); the adds keep their order.  An entry that does
        // not cover k reads element 0 of the slab (present whenever a list is non-empty) and adds nothing.
        const uint32_t c1 = cov_begin[g + 1];
        for (uint32_t i = cov_begin[g] + q; i < c1; i += 4u * GATHER_Q) {
            SymCov cv[4];
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t idx = i + (uint32_t)u * GATHER_Q;
                in[u] = idx < c1;
                cv[u] = cov[in[u] ? idx : c1 - 1u];
            }
            real2 b0[4], b1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                in[u] = in[u] && k >= cv[u].lo && k < cv[u].hi;
                load_pair(slab_r + (in[u] ? cv[u].base + (int64_t)k : (int64_t)0), b0[u], b1[u]);   // the element after an odd-length segment is padding
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!in[u]) continue;
                a0.x += b0[u].x; a0.y += b0[u].y;
                if (k + 1 < cv[u].hi) { a1.x += b1[u].x; a1.y += b1[u].y; }
            }
        }
    }
    part[q][2u * p] = a0;
    part[q][2u * p + 1] = a1;
    __syncthreads();
    if (f < (uint32_t)GATHER_P && lf < kn) {
        real2 t = part[0][f];
#pragma unroll
        for (int j = 1; j < GATHER_Q; ++j) { t.x += part[j][f].x; t.y += part[j][f].y; }
        if (base) { const real2 bb = base[lf]; t.x += bb.x; t.y += bb.y; }
        if constexpr (FUSE) kick_drift_loaded<real, false>(t, vf, xf, lf, pos_next, vel, acc, k0, dt_kick, dt_drift, extras, flags);
        else acc_sum[k0 + lf] = t;
    }
}

template <typename real, bool FUSE>
__global__ __launch_bounds__(BLOCK)
void sym_gather(const typename vec2_of<real>::type *__restrict__ slab_s,
                const typename vec2_of<real>::type *__restrict__ slab_r,
                const uint32_t *__restrict__ row_lo, const uint32_t *__restrict__ row_hi,
                const uint32_t *__restrict__ cov_begin, const SymCov *__restrict__ cov,
                uint32_t n, uint32_t k0, uint32_t kn,
                typename vec2_of<real>::type *__restrict__ acc_sum,
                const typename vec2_of<real>::type *__restrict__ base,
                const typename vec2_of<real>::type *__restrict__ pos_cur,
                typename vec2_of<real>::type *__restrict__ pos_next,
                typename vec2_of<real>::type *__restrict__ vel,
                typename vec2_of<real>::type *__restrict__ acc,
                real dt_kick, real dt_drift, int extras, int flags, uint32_t sb_shift)
{
    sym_gather_block<real, FUSE>(blockIdx.x, slab_s, slab_r, row_lo, row_hi, cov_begin, cov, n, k0, kn, acc_sum, base, pos_cur, pos_next, vel, acc,
                                 dt_kick, dt_drift, extras, flags, sb_shift);
}

// ---------------------------------------------------------------------------
// force_sym_f64 — the symmetric scheme in double precision (BASELINE config 5).
// Same items, tiles (2048 particles per workgroup) and slabs as force_sym_f32; a lane holds
// 8 stationary particles as scalars (no packed fp64 exists), the travelling particle and its
// accumulator are rotated as pairs of 32-bit halves (10 ds_bpermute_b32 per step, 8 with UM).
// Body, both directions: 2 add + 2 fma + v_rsq_f64 + 6 ops (1/r^3 with its correction, rsqrt3_f64) + 4 fma
// (+2 mul with individual masses) for 128 ordered interactions per wave.
// ---------------------------------------------------------------------------
constexpr int SYM_P64 = 8;    // stationary particles per lane: 64 * 8 = SYM_WT per wave
__device__ __forceinline__ double vgpr_const(double k);
__device__ __forceinline__ double rsqrt_f64(double x, double k0375);
__device__ __forceinline__ double rsqrt3_f64(double x, double k15, double k1875);

__device__ __forceinline__ double lane_rot64(double v, int addr)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)(unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

constexpr double PAD_XY64 = 1.0e150;   // r^2 = 2e300 finite, (1/r)^3 = 3.5e-451 underflows to 0

template <bool UM, bool DIAG>
__device__ __forceinline__
void sym_chunks_f64(const double2 *__restrict__ pos, const double *__restrict__ mass,
                    double2 *__restrict__ slab_r_row, uint32_t n, uint32_t c0, uint32_t cnt,
                    const double (&xi)[SYM_P64], const double (&yi)[SYM_P64], const double (&mi)[SYM_P64],
                    double (&ax)[SYM_P64], double (&ay)[SYM_P64], double eps2, double um_mass, double2 (*red)[4][64])
{
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    const int addr = (int)(((lane + 1u) & 63u) * 4u);
    const double k15 = vgpr_const(1.5), k1875 = vgpr_const(1.875);
    double xq = PAD_XY64, yq = PAD_XY64, mq = 0.0;
    {
        const uint32_t j = c0 * SYM_CH + lane;
        if (j < n) { const double2 pj = pos[j]; xq = pj.x; yq = pj.y; if constexpr (!UM) mq = mass[j]; }
    }
    for (uint32_t c = 0; c < cnt; ++c) {
        double xn = PAD_XY64, yn = PAD_XY64, mn = 0.0;
        {
            const uint32_t j = (c0 + c + 1) * SYM_CH + lane;
            if (c + 1 < cnt && j < n) { const double2 pj = pos[j]; xn = pj.x; yn = pj.y; if constexpr (!UM) mn = mass[j]; }
        }
        double aqx = 0.0, aqy = 0.0;
        for (int step = 0; step < 64; ++step) {
            const double xr = lane_rot64(xq, addr), yr = lane_rot64(yq, addr);
            double mr = 0.0;
            if constexpr (!UM) mr = lane_rot64(mq, addr);
#pragma unroll
            for (int p = 0; p < SYM_P64; ++p) {
                const double dx = xq - xi[p], dy = yq - yi[p];
                const double r2 = __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2));
                const double inv3 = rsqrt3_f64(r2, k15, k1875);
                if constexpr (UM) {
                    ax[p] = __builtin_fma(inv3, dx, ax[p]);
                    ay[p] = __builtin_fma(inv3, dy, ay[p]);
                    if constexpr (!DIAG) { aqx = __builtin_fma(-inv3, dx, aqx); aqy = __builtin_fma(-inv3, dy, aqy); }
                } else {
                    const double si = mq * inv3;
                    ax[p] = __builtin_fma(si, dx, ax[p]);
                    ay[p] = __builtin_fma(si, dy, ay[p]);
                    if constexpr (!DIAG) {
                        const double sj = mi[p] * inv3;
                        aqx = __builtin_fma(-sj, dx, aqx);
                        aqy = __builtin_fma(-sj, dy, aqy);
                    }
                }
            }
            xq = xr; yq = yr;
            if constexpr (!UM) mq = mr;
            if constexpr (!DIAG) { aqx = lane_rot64(aqx, addr); aqy = lane_rot64(aqy, addr); }
        }
        if constexpr (!DIAG) {
            double2 r = make_double2(aqx, aqy);
            if constexpr (UM) { r.x *= um_mass; r.y *= um_mass; }
            double2 (*rb)[64] = red[c & 1u];
            rb[w][lane] = r;
            __syncthreads();
            if (w == 0) {
                const uint32_t j = (c0 + c) * SYM_CH + lane;
                double2 a = rb[0][lane];
#pragma unroll
                for (int k = 1; k < 4; ++k) { a.x += rb[k][lane].x; a.y += rb[k][lane].y; }
                if (j < n) slab_r_row[j] = a;
            }
        }
        xq = xn; yq = yn;
        if constexpr (!UM) mq = mn;
    }
}

template <bool UM>
__global__ __launch_bounds__(BLOCK)
void force_sym_f64(const double2 *__restrict__ pos, const double *__restrict__ mass,
                   const SymItem *__restrict__ items,
                   double2 *__restrict__ slab_s, double2 *__restrict__ slab_r,
                   uint32_t n, double eps2, double um_mass, uint32_t *ticket, uint32_t first_wave, uint32_t ticket_base)
{
    static_assert(64 * SYM_P64 == SYM_WT, "fp64 and fp32 symmetric kernels share the tile geometry");
    __shared__ double2 red[2][4][64];
    const SymItem it = items[sym_item_index(ticket, first_wave, ticket_base)];
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    double xi[SYM_P64], yi[SYM_P64], mi[SYM_P64], ax[SYM_P64], ay[SYM_P64];
    uint32_t li[SYM_P64];
#pragma unroll
    for (int p = 0; p < SYM_P64; ++p) {
        li[p] = w * SYM_WT + (uint32_t)p * 64u + lane;
        const uint32_t g = it.tile * SYM_SB + li[p];
        xi[p] = PAD_XY64; yi[p] = PAD_XY64; mi[p] = 0.0;
        if (g < n) { const double2 q = pos[g]; xi[p] = q.x; yi[p] = q.y; if constexpr (!UM) mi[p] = mass[g]; }
        ax[p] = 0.0; ay[p] = 0.0;
    }
    double2 *__restrict__ rrow = slab_r + it.r_base;
    if (it.diag) sym_chunks_f64<UM, true>(pos, mass, rrow, n, it.c0, it.cnt, xi, yi, mi, ax, ay, eps2, um_mass, red);
    else         sym_chunks_f64<UM, false>(pos, mass, rrow, n, it.c0, it.cnt, xi, yi, mi, ax, ay, eps2, um_mass, red);
    double2 *__restrict__ out = slab_s + (size_t)it.s_row * SYM_SB;
#pragma unroll
    for (int p = 0; p < SYM_P64; ++p) {
        if constexpr (UM) { ax[p] *= um_mass; ay[p] *= um_mass; }
        out[li[p]] = make_double2(ax[p], ay[p]);
    }
}

// ---------------------------------------------------------------------------
// force_seq_f32 — reference summation order (NB_SUM_SEQUENTIAL).
// One lane per i, j ascending over the WHOLE range in one running sum, every
// operation individually rounded (no FMA), guard `r_sq > 0` kept: the exact
// arithmetic of Quadtree.hpp:134-144.  With RSQ_QUAKE the result is
// bit-identical to the compiled reference (tests/test_parity_gpu.py).
// ---------------------------------------------------------------------------
template <int RSQ>
__global__ __launch_bounds__(BLOCK)
void force_seq_f32(const float2 *__restrict__ pos, const float *__restrict__ mass,
                   float2 *__restrict__ partial,
                   uint32_t i_begin, uint32_t i_count,
                   uint32_t j_begin, uint32_t j_end, float eps2)
{
#pragma clang fp contract(off)
    __shared__ v4f tile[2][TJ];
    const uint32_t t = threadIdx.x;
    const uint32_t li = blockIdx.x * BLOCK + t;
    const float2 pi = pos[i_begin + min(li, i_count - 1)];
    float sx = 0.f, sy = 0.f;

    const uint32_t ntiles = (j_end - j_begin + TJ - 1) / TJ;
    {
        const uint32_t j = j_begin + t;
        float2 pj = make_float2(0.f, 0.f); float mj = 0.f;
        if (j < j_end) { pj = pos[j]; mj = mass[j]; }
        tile[0][t] = (v4f){pj.x, pj.y, mj, 0.f};
    }
    __syncthreads();
    for (uint32_t it = 0; it < ntiles; ++it) {
        float2 pn = make_float2(0.f, 0.f); float mn = 0.f;
        const uint32_t jn1 = j_begin + (it + 1) * TJ + t;
        if (jn1 < j_end) { pn = pos[jn1]; mn = mass[jn1]; }
        const v4f *__restrict__ cur = tile[it & 1];
        const uint32_t cnt = min((uint32_t)TJ, j_end - (j_begin + it * TJ));
        for (uint32_t jj = 0; jj < cnt; ++jj) {
            const v4f q = cur[jj];
            const float rx = q.x - pi.x;
            const float ry = q.y - pi.y;
            const float r_sq = rx * rx + ry * ry;
            const float tt = r_sq + eps2;
            float inv;
            if constexpr (RSQ == RSQ_QUAKE) inv = quake_rsqrt(tt);
            else inv = 1.0f / sqrtf(tt);   // correctly rounded sqrt and divide (hipcc default)
            const float inv3 = inv * inv * inv;
            const float s = q.z * inv3;
            const float cx = rx * s, cy = ry * s;
            sx = r_sq > 0.f ? sx + cx : sx;
            sy = r_sq > 0.f ? sy + cy : sy;
        }
        if (it + 1 < ntiles) tile[(it + 1) & 1][t] = (v4f){pn.x, pn.y, mn, 0.f};
        __syncthreads();
    }
    if (li < i_count) partial[li] = make_float2(sx, sy);
}

// ---------------------------------------------------------------------------
// force_tiled_f64 — fp64 extension (BASELINE config 5).  One i per lane per
// register slot (P slots), same tiling; 1/sqrt from v_rsq_f64 refined by one
// third-order step (relative error ~1e-16 after refinement).
// ---------------------------------------------------------------------------
// A floating-point constant held in a VGPR pair for the whole kernel.  gfx950 (GFX9 encoding) cannot put a 64-bit
// literal into a VOP3 v_fma_f64, so `fma(e, 1.875, 1.5)` would otherwise be compiled as v_mov_b32 x2 (re-materialising
// 1.5 in the destination) + v_fmac_f64 with the literal 1.875: two extra VALU instructions per pair body, 5 % of the
// fp64 force kernels.  The empty asm makes the value opaque; every kernel creates its constants once, before its
// loops (k15 = vgpr_const(1.5) ...), and hands them down.
__device__ __forceinline__ double vgpr_const(double k)
{
    asm("" : "+v"(k));
    return k;
}

__device__ __forceinline__ double rsqrt_f64(double x, double k0375)
{
    double y = __builtin_amdgcn_rsq(x);            // v_rsq_f64, ~2^-26 relative
    const double e = __builtin_fma(-x * y, y, 1.0);  // 1 - x y^2
    // y * (1 + e/2 + 3e^2/8): third-order correction
    const double c = __builtin_fma(e, k0375, 0.5);
    return __builtin_fma(y * e, c, y);
}

// x^(-3/2) in one go: with y = v_rsq_f64(x) and e = 1 - x y^2,  x^(-3/2) = y^3 (1 - e)^(-3/2)
// = y^3 (1 + 3e/2 + 15e^2/8 + O(e^3)),  e ~ 1e-7: six operations after the v_rsq_f64 where
// rsqrt_f64 followed by inv * inv * inv takes seven (the force kernels are VALU-bound: -5 %).
__device__ __forceinline__ double rsqrt3_f64(double x, double k15, double k1875)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double y2 = y * y;
    const double e = __builtin_fma(-x, y2, 1.0);
    const double y3 = y2 * y;
    const double c = __builtin_fma(e, k1875, k15);
    return __builtin_fma(y3 * e, c, y3);
}

template <int P, bool GUARD, int UNROLL>
__global__ __launch_bounds__(BLOCK)
void force_tiled_f64(const double2 *__restrict__ pos, const double *__restrict__ mass,
                     double2 *__restrict__ partial,
                     uint32_t i_begin, uint32_t i_count,
                     uint32_t j_begin, uint32_t j_end,
                     uint32_t js, uint32_t i_tiles, double eps2,
                     uint32_t gap_begin, uint32_t gap_len)
{
    constexpr uint32_t IT = BLOCK * P;
    struct alignas(16) JD { double x, y, m, pad; };
    __shared__ JD tile[2][TJ];
    const double k15 = vgpr_const(1.5), k1875 = vgpr_const(1.875);

    const TileMap tm = decode_block(blockIdx.x, i_tiles, js);
    if (!tm.valid) return;
    const uint32_t t = threadIdx.x;
    const uint32_t jn = j_end - j_begin;
    const uint32_t slice_len = (((jn + js - 1) / js + TJ - 1) / TJ) * TJ;
    const uint32_t s0 = j_begin + min(tm.slice * slice_len, jn);
    const uint32_t s1 = j_begin + min((tm.slice + 1) * slice_len, jn);

    double xi[P], yi[P], ax[P], ay[P];
    uint32_t li[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        li[p] = tm.i_tile * IT + (uint32_t)p * BLOCK + t;
        const double2 p0 = pos[i_begin + min(li[p], i_count - 1)];
        xi[p] = p0.x; yi[p] = p0.y; ax[p] = 0.0; ay[p] = 0.0;
    }
    const uint32_t ntiles = (s1 - s0 + TJ - 1) / TJ;
    {
        const uint32_t j = s0 + t;
        double2 pj = make_double2(0.0, 0.0); double mj = 0.0;
        if (j < s1) { const uint32_t jg = j + (j >= gap_begin ? gap_len : 0u); pj = pos[jg]; mj = mass[jg]; }
        tile[0][t] = JD{pj.x, pj.y, mj, 0.0};
    }
    __syncthreads();
    for (uint32_t it = 0; it < ntiles; ++it) {
        double2 pn = make_double2(0.0, 0.0); double mn = 0.0;
        const uint32_t jn1 = s0 + (it + 1) * TJ + t;
        if (jn1 < s1) { const uint32_t jg = jn1 + (jn1 >= gap_begin ? gap_len : 0u); pn = pos[jg]; mn = mass[jg]; }
        const JD *__restrict__ cur = tile[it & 1];
#pragma unroll UNROLL
        for (int jj = 0; jj < TJ; ++jj) {
            const double xj = cur[jj].x, yj = cur[jj].y, mj = cur[jj].m;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const double dx = xj - xi[p], dy = yj - yi[p];
                double r2, inv3;
                if constexpr (GUARD) {
                    r2 = __builtin_fma(dy, dy, dx * dx);
                    inv3 = r2 > 0.0 ? rsqrt3_f64(r2, k15, k1875) : 0.0;
                } else {
                    r2 = __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2));
                    inv3 = rsqrt3_f64(r2, k15, k1875);
                }
                const double s = mj * inv3;
                ax[p] = __builtin_fma(s, dx, ax[p]);
                ay[p] = __builtin_fma(s, dy, ay[p]);
            }
        }
        if (it + 1 < ntiles) tile[(it + 1) & 1][t] = JD{pn.x, pn.y, mn, 0.0};
        __syncthreads();
    }
    double2 *__restrict__ out = partial + (size_t)tm.slice * i_count;
#pragma unroll
    for (int p = 0; p < P; ++p)
        if (li[p] < i_count) out[li[p]] = make_double2(ax[p], ay[p]);
}

// ---------------------------------------------------------------------------
// integrate — Simulation::iterate after attract(), Simulation.hpp:129-163, for
// the owned block: a = sum of slabs (fixed order); v += a dt; [clamp];
// [boundary]; x_next = x + v dt.  STRICT keeps every operation individually
// rounded (the reference build has no FMA contraction) for bit parity.
// pos_cur/pos_next are full-n replicas; vel/acc are indexed by local i.
// flags: INTEG_KICK applies the kick (+extras), INTEG_DRIFT writes pos_next;
// flags = 0 only gathers the slabs into acc (nb_accelerations, KDK bootstrap).
// ---------------------------------------------------------------------------

template <typename real, bool STRICT>
__global__ __launch_bounds__(BLOCK)
void integrate(const typename vec2_of<real>::type *__restrict__ pos_cur,
               typename vec2_of<real>::type *__restrict__ pos_next,
               typename vec2_of<real>::type *__restrict__ vel,
               typename vec2_of<real>::type *__restrict__ acc,
               const typename vec2_of<real>::type *__restrict__ partial,
               uint32_t nslabs, uint32_t i_begin, uint32_t i_count,
               real dt_kick, real dt_drift, int extras, int flags)
{
    typedef typename vec2_of<real>::type real2;
    const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
    if (li >= i_count) return;
    real2 a = partial[li];
    for (uint32_t s = 1; s < nslabs; ++s) {
        const real2 b = partial[(size_t)s * i_count + li];
        a.x += b.x; a.y += b.y;
    }
    kick_drift_one<real, STRICT>(a, li, pos_cur, pos_next, vel, acc, i_begin, dt_kick, dt_drift, extras, flags);
}

// ---------------------------------------------------------------------------
// sum_partials — in-process reduce-scatter of the symmetric sharded protocol: dst[k] = sum over the
// handles (in handle order) of their partial acceleration of particle first + k.  `src` holds the
// handles' acc_full pointers (peer-accessible device memory); T is the element type.
// ---------------------------------------------------------------------------
struct PartialPtrs { const void *p[64]; };

__device__ __forceinline__ void add_to(float2 &a, const float2 &b) { a.x += b.x; a.y += b.y; }
__device__ __forceinline__ void add_to(double2 &a, const double2 &b) { a.x += b.x; a.y += b.y; }
__device__ __forceinline__ void add_to(float4 &a, const float4 &b) { a.x += b.x; a.y += b.y; a.z += b.z; }
__device__ __forceinline__ void add_to(double4 &a, const double4 &b) { a.x += b.x; a.y += b.y; a.z += b.z; }

template <typename T>
__global__ __launch_bounds__(BLOCK)
void sum_partials(PartialPtrs src, int count, uint32_t first, uint32_t cnt, T *__restrict__ dst)
{
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= cnt) return;
    T a = static_cast<const T *>(src.p[0])[first + k];
    for (int r = 1; r < count; ++r) add_to(a, static_cast<const T *>(src.p[r])[first + k]);
    dst[k] = a;
}

// ---------------------------------------------------------------------------
// AoS (64-byte Body records, Body.hpp:6-13) <-> SoA
// ---------------------------------------------------------------------------
struct BodyRec { float4 q[4]; };  // pos|pad, vel|pad, acc|pad, mass radius pad pad

template <typename real>
__global__ __launch_bounds__(BLOCK)
void unpack_bodies(const BodyRec *__restrict__ aos, uint32_t n,
                   typename vec2_of<real>::type *__restrict__ pos, real *__restrict__ mass,
                   typename vec2_of<real>::type *__restrict__ vel,
                   typename vec2_of<real>::type *__restrict__ acc,
                   float *__restrict__ radius,
                   uint32_t i_begin, uint32_t i_count)
{
    typedef typename vec2_of<real>::type real2;
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 p = aos[i].q[0], m = aos[i].q[3];
    real2 pp; pp.x = (real)p.x; pp.y = (real)p.y;
    pos[i] = pp;
    mass[i] = (real)m.x;
    radius[i] = m.y;
    if (i >= i_begin && i - i_begin < i_count) {
        const float4 v = aos[i].q[1], a = aos[i].q[2];
        real2 vv; vv.x = (real)v.x; vv.y = (real)v.y;
        real2 aa; aa.x = (real)a.x; aa.y = (real)a.y;
        vel[i - i_begin] = vv;
        acc[i - i_begin] = aa;
    }
}

template <typename real>
__global__ __launch_bounds__(BLOCK)
void pack_bodies(BodyRec *__restrict__ aos,
                 const typename vec2_of<real>::type *__restrict__ pos, const real *__restrict__ mass,
                 const typename vec2_of<real>::type *__restrict__ vel,
                 const typename vec2_of<real>::type *__restrict__ acc,
                 const float *__restrict__ radius,
                 uint32_t i_begin, uint32_t i_count)
{
    const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
    if (li >= i_count) return;
    const auto p = pos[i_begin + li];
    const auto v = vel[li];
    const auto a = acc[li];
    BodyRec r;
    r.q[0] = make_float4((float)p.x, (float)p.y, 0.f, 0.f);
    r.q[1] = make_float4((float)v.x, (float)v.y, 0.f, 0.f);
    r.q[2] = make_float4((float)a.x, (float)a.y, 0.f, 0.f);
    r.q[3] = make_float4((float)mass[i_begin + li], radius[i_begin + li], 0.f, 0.f);
    aos[li] = r;
}

template <typename real>
__global__ __launch_bounds__(BLOCK)
void pack_positions(float2 *__restrict__ out, const typename vec2_of<real>::type *__restrict__ pos,
                    uint32_t i_begin, uint32_t i_count)
{
    const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
    if (li >= i_count) return;
    const auto p = pos[i_begin + li];
    out[li] = make_float2((float)p.x, (float)p.y);
}

// ---------------------------------------------------------------------------
// energy — fp64 accumulation whatever the state precision.
//   ksum[b] = sum over the block's owned i of m v^2 / 2
//   usum[b] = - sum_i m_i sum_{j > i} m_j / sqrt(r^2 + eps^2)     (every unordered pair once: j-tiles below the
//             block's first particle are skipped, so the sweep costs n^2/2 pair evaluations; the shares of the
//             handles of a sharded run still add up to the total)
// Per-block partials are summed on the host in block order (deterministic).
// ---------------------------------------------------------------------------
template <typename real>
__global__ __launch_bounds__(BLOCK)
void energy_partials(const typename vec2_of<real>::type *__restrict__ pos, const real *__restrict__ mass,
                     const typename vec2_of<real>::type *__restrict__ vel,
                     uint32_t n, uint32_t i_begin, uint32_t i_count, double eps2,
                     double *__restrict__ ksum, double *__restrict__ usum)
{
    struct alignas(16) JD { double x, y, m, pad; };
    __shared__ JD tile[TJ];
    __shared__ double red[2][BLOCK / 64];
    const uint32_t t = threadIdx.x;
    const uint32_t li = blockIdx.x * BLOCK + t;
    const bool live = li < i_count;
    const uint32_t gi = i_begin + (live ? li : i_count - 1);
    const double xi = (double)pos[gi].x, yi = (double)pos[gi].y;
    const double k0375 = vgpr_const(0.375);
    double u = 0.0;
    const uint32_t first = ((i_begin + blockIdx.x * BLOCK) / TJ) * TJ;      // j-tile holding the block's first particle
    for (uint32_t j0 = first; j0 < n; j0 += TJ) {
        const uint32_t j = j0 + t;
        __syncthreads();
        if (j < n) tile[t] = JD{(double)pos[j].x, (double)pos[j].y, (double)mass[j], 0.0};
        else tile[t] = JD{0.0, 0.0, 0.0, 0.0};
        __syncthreads();
        const uint32_t cnt = min((uint32_t)TJ, n - j0);
        for (uint32_t jj = 0; jj < cnt; ++jj) {
            const double dx = tile[jj].x - xi, dy = tile[jj].y - yi;
            const double r2 = __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2));
            const double w = (j0 + jj > gi) ? tile[jj].m : 0.0;
            u = __builtin_fma(w, rsqrt_f64(r2, k0375), u);      // v_rsq_f64 + third-order step: 1.4e-16 relative
        }
    }
    double k = 0.0, uu = 0.0;
    if (live) {
        const double m = (double)mass[gi];
        const double vx = (double)vel[li].x, vy = (double)vel[li].y;
        k = 0.5 * m * (vx * vx + vy * vy);
        uu = -m * u;
    }
    // wave64 reduction with shuffles, then across the 4 waves through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        k += __shfl_down(k, off, 64);
        uu += __shfl_down(uu, off, 64);
    }
    if ((t & 63) == 0) { red[0][t >> 6] = k; red[1][t >> 6] = uu; }
    __syncthreads();
    if (t == 0) {
        double ks = 0.0, us = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) { ks += red[0][w]; us += red[1][w]; }
        ksum[blockIdx.x] = ks;
        usum[blockIdx.x] = us;
    }
}

// ---------------------------------------------------------------------------
// momentum — Body::momentum (Body.hpp:103-106: mass * vel) summed over the owned block in fp64, plus the angular
// momentum about the origin Lz = sum m (x vy - y vx).  psum is [4][blocks]: px | py | pz (0 in 2-D) | Lz; per-block
// partials are summed on the host in block order (deterministic), like the energy.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void block_reduce4(double (&v)[4], double *__restrict__ out, uint32_t blocks)
{
    __shared__ double red[4][BLOCK / 64];
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[c] += __shfl_down(v[c], off, 64);
        if ((t & 63) == 0) red[c][t >> 6] = v[c];
    }
    __syncthreads();
    if (t < 4) {
        double a = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) a += red[t][w];
        out[(size_t)t * blocks + blockIdx.x] = a;
    }
}

template <typename real>
__global__ __launch_bounds__(BLOCK)
void momentum_partials(const typename vec2_of<real>::type *__restrict__ pos, const real *__restrict__ mass,
                       const typename vec2_of<real>::type *__restrict__ vel, uint32_t i_begin, uint32_t i_count,
                       double *__restrict__ psum)
{
    const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (li < i_count) {
        const double m = (double)mass[i_begin + li];
        const double x = (double)pos[i_begin + li].x, y = (double)pos[i_begin + li].y;
        const double vx = (double)vel[li].x, vy = (double)vel[li].y;
        v[0] = m * vx; v[1] = m * vy; v[3] = m * (x * vy - y * vx);
    }
    block_reduce4(v, psum, gridDim.x);
}

} // namespace nbk
