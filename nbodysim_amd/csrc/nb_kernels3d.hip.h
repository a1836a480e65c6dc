// nb_kernels3d.hip.h — the 3-D variant of the hot path (build extension).
//
// The reference is strictly 2-D (`Vec2`), but each `alignas(16) Vec2 {float x, y;}` carries 8 bytes
// of padding right after y (Nbodysim/headers/Vec2.hpp:17-20), so z fits at offset +8 of pos / vel /
// acc without moving x, y or changing sizeof(Body) = 64 (Body.hpp:6-13).  With nb_params.dims = 3
// the library reads and writes those slots and evaluates the same softened force with a third
// component (20 algorithmic flop per pair instead of 14).  There is no reference arithmetic to be
// bit-exact with here: parity is against the fp64 restatement only ("parity unpinned").
//
// Device layout: positions as real4 {x, y, z, m} (float4: one 16-byte load per particle; double4: two; the
// mass rides along), velocities / accelerations / slab rows as real4 {·, ·, ·, 0}.
// Kernels mirror the 2-D ones of nb_kernels.hip.h, in fp32 and fp64:
//   force_sym3_f32 / _f64    symmetric (Newton's third law) fast path, lane-rotated travelling particles
//   force_tiled3_f32 / _f64  one-sided LDS-tiled kernel (small n, eps = 0, all-gather sharding, cross-check);
//                            the j range skips [gap_begin, gap_begin + gap_len) like the 2-D kernels
//   sym_gather3 / integrate3 / pack3 / unpack3 / energy3 / (sum_partials of nb_kernels.hip.h)
#pragma once
#include "nb_kernels.hip.h"

namespace nbk {

template <typename real> struct vec4_of;
template <> struct vec4_of<float> { typedef float4 type; };
template <> struct vec4_of<double> { typedef double4 type; };

template <typename real> __device__ __forceinline__ typename vec4_of<real>::type make_real4(real x, real y, real z, real w);
template <> __device__ __forceinline__ float4 make_real4<float>(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
template <> __device__ __forceinline__ double4 make_real4<double>(double x, double y, double z, double w) { return make_double4(x, y, z, w); }

// ---------------------------------------------------------------------------
// kick/drift for one particle, 3 components (the 2-D extras — velocity clamp, soft
// boundary — are defined by the reference in the plane only and are not applied).
// ---------------------------------------------------------------------------
template <typename real>
__device__ __forceinline__
void kick_drift_one3(typename vec4_of<real>::type a, uint32_t li, const typename vec4_of<real>::type *__restrict__ pos_cur,
                     typename vec4_of<real>::type *__restrict__ pos_next, typename vec4_of<real>::type *__restrict__ vel,
                     typename vec4_of<real>::type *__restrict__ acc, uint32_t i_begin, real dt_kick, real dt_drift, int flags)
{
    typedef typename vec4_of<real>::type real4;
    a.w = 0;
    acc[li] = a;
    if (!(flags & INTEG_KICK)) return;
    real4 v = vel[li];
    const real4 x = pos_cur[i_begin + li];
    v.x = __builtin_fma(a.x, dt_kick, v.x);
    v.y = __builtin_fma(a.y, dt_kick, v.y);
    v.z = __builtin_fma(a.z, dt_kick, v.z);
    vel[li] = v;
    if (flags & INTEG_DRIFT) {
        real4 xn;
        xn.x = __builtin_fma(v.x, dt_drift, x.x);
        xn.y = __builtin_fma(v.y, dt_drift, x.y);
        xn.z = __builtin_fma(v.z, dt_drift, x.z);
        xn.w = x.w;                                  // the mass travels with the position
        pos_next[i_begin + li] = xn;
    }
}

template <typename real>
__global__ __launch_bounds__(BLOCK)
void integrate3(const typename vec4_of<real>::type *__restrict__ pos_cur, typename vec4_of<real>::type *__restrict__ pos_next,
                typename vec4_of<real>::type *__restrict__ vel, typename vec4_of<real>::type *__restrict__ acc,
                const typename vec4_of<real>::type *__restrict__ partial, uint32_t nslabs,
                uint32_t i_begin, uint32_t i_count, real dt_kick, real dt_drift, int flags)
{
    const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
    if (li >= i_count) return;
    auto a = partial[li];
    for (uint32_t s = 1; s < nslabs; ++s) {
        const auto b = partial[(size_t)s * i_count + li];
        a.x += b.x; a.y += b.y; a.z += b.z;
    }
    kick_drift_one3<real>(a, li, pos_cur, pos_next, vel, acc, i_begin, dt_kick, dt_drift, flags);
}

// ---------------------------------------------------------------------------
// force_tiled3_f32 — one-sided, LDS-tiled (the 3-D twin of force_tiled_f32, WS = 4).
// Tile entries are the position records themselves: float4 {x, y, z, m}.
// ---------------------------------------------------------------------------
template <int P, int RSQ, bool GUARD, int UNROLL, bool UM>
__global__ __launch_bounds__(BLOCK)
void force_tiled3_f32(const float4 *__restrict__ pos, float4 *__restrict__ partial,
                      uint32_t i_begin, uint32_t i_count, uint32_t j_begin, uint32_t j_end,
                      uint32_t js, uint32_t i_tiles, float eps2, float um_mass, uint32_t gap_begin, uint32_t gap_len)
{
    constexpr int WS = 4;
    constexpr uint32_t LANES_I = BLOCK / WS, IT = LANES_I * 2 * P, JW = TJ / WS;
    constexpr uint32_t RED = (WS - 1) * P * LANES_I * 2;      // two float4 per (p, lane): 6 sums in 8 slots
    constexpr uint32_t SMEM = 2 * TJ > RED ? 2 * TJ : RED;
    __shared__ v4f smem[SMEM];
    v4f (*tile)[TJ] = reinterpret_cast<v4f (*)[TJ]>(smem);

    const TileMap tm = decode_block(blockIdx.x, i_tiles, js);
    if (!tm.valid) return;
    const uint32_t t = threadIdx.x, lane_i = t % LANES_I, w = t / LANES_I;
    const uint32_t jn = j_end - j_begin;
    const uint32_t slice_len = (((jn + js - 1) / js + TJ - 1) / TJ) * TJ;
    const uint32_t s0 = j_begin + min(tm.slice * slice_len, jn);
    const uint32_t s1 = j_begin + min((tm.slice + 1) * slice_len, jn);

    v2f xi[P], yi[P], zi[P], ax[P], ay[P], az[P];
    uint32_t li[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        li[p] = tm.i_tile * IT + (uint32_t)p * (LANES_I * 2) + 2u * lane_i;
        const float4 p0 = pos[i_begin + min(li[p], i_count - 1)], p1 = pos[i_begin + min(li[p] + 1, i_count - 1)];
        xi[p] = (v2f){p0.x, p1.x}; yi[p] = (v2f){p0.y, p1.y}; zi[p] = (v2f){p0.z, p1.z};
        ax[p] = ay[p] = az[p] = (v2f){0.f, 0.f};
    }
    const v2f e2 = {eps2, eps2};
    const uint32_t ntiles = (s1 - s0 + TJ - 1) / TJ;
    const v4f pad = {PAD_XY, PAD_XY, PAD_XY, 0.f};
    {
        const uint32_t j = s0 + t;
        v4f q = pad;
        if (j < s1) { const float4 r = pos[j + (j >= gap_begin ? gap_len : 0u)]; q = (v4f){r.x, r.y, r.z, r.w}; }
        tile[0][t] = q;
    }
    __syncthreads();
    for (uint32_t it = 0; it < ntiles; ++it) {
        v4f qn = pad;
        const uint32_t jn1 = s0 + (it + 1) * TJ + t;
        if (jn1 < s1) { const float4 r = pos[jn1 + (jn1 >= gap_begin ? gap_len : 0u)]; qn = (v4f){r.x, r.y, r.z, r.w}; }
        const v4f *__restrict__ cur = tile[it & 1] + w * JW;
#pragma unroll UNROLL
        for (int jj = 0; jj < (int)JW; ++jj) {
            const v4f q = cur[jj];
            const v2f xj = {q.x, q.x}, yj = {q.y, q.y}, zj = {q.z, q.z}, mj = {q.w, q.w};
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const v2f dx = xj - xi[p], dy = yj - yi[p], dz = zj - zi[p];
                v2f r2, inv;
                if constexpr (GUARD) {
                    r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                    const v2f t2 = r2 + e2;
                    if constexpr (RSQ == RSQ_EXACT) inv = (v2f){__builtin_amdgcn_rsqf(t2.x), __builtin_amdgcn_rsqf(t2.y)};
                    else inv = quake_rsqrt2(t2);
                    inv.x = r2.x > 0.f ? inv.x : 0.f;
                    inv.y = r2.y > 0.f ? inv.y : 0.f;
                } else {
                    r2 = __builtin_elementwise_fma(dx, dx, e2);
                    r2 = __builtin_elementwise_fma(dy, dy, r2);
                    r2 = __builtin_elementwise_fma(dz, dz, r2);
                    if constexpr (RSQ == RSQ_EXACT) inv = (v2f){__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                    else inv = quake_rsqrt2(r2);
                }
                const v2f inv2 = inv * inv;
                v2f s;
                if constexpr (UM) s = inv * inv2; else s = (mj * inv) * inv2;
                ax[p] = __builtin_elementwise_fma(s, dx, ax[p]);
                ay[p] = __builtin_elementwise_fma(s, dy, ay[p]);
                az[p] = __builtin_elementwise_fma(s, dz, az[p]);
            }
        }
        if (it + 1 < ntiles) tile[(it + 1) & 1][t] = qn;
        __syncthreads();
    }
    // combine the WS wave groups in order through LDS
    if (w > 0) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            smem[(((w - 1) * P + p) * LANES_I + lane_i) * 2 + 0] = (v4f){ax[p].x, ay[p].x, az[p].x, 0.f};
            smem[(((w - 1) * P + p) * LANES_I + lane_i) * 2 + 1] = (v4f){ax[p].y, ay[p].y, az[p].y, 0.f};
        }
    }
    __syncthreads();
    if (w > 0) return;
    float4 *__restrict__ out = partial + (size_t)tm.slice * i_count;
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int k = 0; k < WS - 1; ++k) {
            const v4f r0 = smem[((k * P + p) * LANES_I + lane_i) * 2 + 0], r1 = smem[((k * P + p) * LANES_I + lane_i) * 2 + 1];
            ax[p] += (v2f){r0.x, r1.x}; ay[p] += (v2f){r0.y, r1.y}; az[p] += (v2f){r0.z, r1.z};
        }
        if constexpr (UM) { ax[p] *= um_mass; ay[p] *= um_mass; az[p] *= um_mass; }
        if (li[p] < i_count) out[li[p]] = make_float4(ax[p].x, ay[p].x, az[p].x, 0.f);
        if (li[p] + 1 < i_count) out[li[p] + 1] = make_float4(ax[p].y, ay[p].y, az[p].y, 0.f);
    }
}

// ---------------------------------------------------------------------------
// force_sym3_f32 — symmetric fast path in 3-D.  Same items / tiles / slab scheme as force_sym_f32
// (DESIGN.md §4.1): 8 stationary particles per lane as 4 packed pairs, one travelling particle per
// lane rotated with ds_bpermute_b32 (x, y, z [, m] and six accumulator halves: 9-10 per step).
// Body, both directions: 3 pk_add + 3 pk_fma + 2 rsq + 2 pk_mul + 6 pk_fma = 14 packed + 2 trans.
// ---------------------------------------------------------------------------
template <int RSQ, bool UM, bool DIAG>
__device__ __forceinline__
void sym3_chunks(const float4 *__restrict__ pos, float4 *__restrict__ slab_r_row, uint32_t n, uint32_t c0, uint32_t cnt,
                 const v2f (&xi)[SYM_P], const v2f (&yi)[SYM_P], const v2f (&zi)[SYM_P], const v2f (&mi)[SYM_P],
                 v2f (&ax)[SYM_P], v2f (&ay)[SYM_P], v2f (&az)[SYM_P], float eps2, float um_mass, float4 (*red)[4][64])
{
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    const int addr = (int)(((lane + 1u) & 63u) * 4u);
    const v2f e2 = {eps2, eps2};
    float xq = PAD_XY, yq = PAD_XY, zq = PAD_XY, mq = 0.f;
    {
        const uint32_t j = c0 * SYM_CH + lane;
        if (j < n) { const float4 pj = pos[j]; xq = pj.x; yq = pj.y; zq = pj.z; mq = pj.w; }
    }
    for (uint32_t c = 0; c < cnt; ++c) {
        float xn = PAD_XY, yn = PAD_XY, zn = PAD_XY, mn = 0.f;
        {
            const uint32_t j = (c0 + c + 1) * SYM_CH + lane;
            if (c + 1 < cnt && j < n) { const float4 pj = pos[j]; xn = pj.x; yn = pj.y; zn = pj.z; mn = pj.w; }
        }
        v2f aqx = {0.f, 0.f}, aqy = {0.f, 0.f}, aqz = {0.f, 0.f};
#pragma unroll SYM_UNROLL
        for (int step = 0; step < 64; ++step) {
            const float xr = lane_rot(xq, addr), yr = lane_rot(yq, addr), zr = lane_rot(zq, addr);
            float mr = 0.f;
            if constexpr (!UM) mr = lane_rot(mq, addr);
            const v2f xj = {xq, xq}, yj = {yq, yq}, zj = {zq, zq};
#pragma unroll
            for (int p = 0; p < SYM_P; ++p) {
                const v2f dx = xj - xi[p], dy = yj - yi[p], dz = zj - zi[p];
                v2f r2 = __builtin_elementwise_fma(dx, dx, e2);
                r2 = __builtin_elementwise_fma(dy, dy, r2);
                r2 = __builtin_elementwise_fma(dz, dz, r2);
                v2f inv;
                if constexpr (RSQ == RSQ_EXACT) inv = (v2f){__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                else inv = quake_rsqrt2(r2);
                const v2f inv3 = inv * (inv * inv);
                v2f si = inv3, sj = inv3;
                if constexpr (!UM) { si = (v2f){mq, mq} * inv3; sj = mi[p] * inv3; }
                ax[p] = __builtin_elementwise_fma(si, dx, ax[p]);
                ay[p] = __builtin_elementwise_fma(si, dy, ay[p]);
                az[p] = __builtin_elementwise_fma(si, dz, az[p]);
                if constexpr (!DIAG) {
                    aqx = __builtin_elementwise_fma(-sj, dx, aqx);
                    aqy = __builtin_elementwise_fma(-sj, dy, aqy);
                    aqz = __builtin_elementwise_fma(-sj, dz, aqz);
                }
            }
            xq = xr; yq = yr; zq = zr;
            if constexpr (!UM) mq = mr;
            if constexpr (!DIAG) {
                aqx.x = lane_rot(aqx.x, addr); aqx.y = lane_rot(aqx.y, addr);
                aqy.x = lane_rot(aqy.x, addr); aqy.y = lane_rot(aqy.y, addr);
                aqz.x = lane_rot(aqz.x, addr); aqz.y = lane_rot(aqz.y, addr);
            }
        }
        if constexpr (!DIAG) {
            float4 r = make_float4(aqx.x + aqx.y, aqy.x + aqy.y, aqz.x + aqz.y, 0.f);
            if constexpr (UM) { r.x *= um_mass; r.y *= um_mass; r.z *= um_mass; }
            float4 (*rb)[64] = red[c & 1u];
            rb[w][lane] = r;
            __syncthreads();
            if (w == 0) {
                const uint32_t j = (c0 + c) * SYM_CH + lane;
                float4 a = rb[0][lane];
#pragma unroll
                for (int k = 1; k < 4; ++k) { a.x += rb[k][lane].x; a.y += rb[k][lane].y; a.z += rb[k][lane].z; }
                if (j < n) slab_r_row[j] = a;
            }
        }
        xq = xn; yq = yn; zq = zn;
        if constexpr (!UM) mq = mn;
    }
}

// sym3_chunks2 — chunk PAIRS in 3-D (see sym_chunks2 in nb_kernels.hip.h): two travelling particles per lane in the two
// halves of the packed registers, the stationary particle broadcast; 12 ds_bpermute_b32 per 16 pairs instead of 9 per 8.
template <int RSQ, bool UM, bool DIAG>
__device__ __forceinline__
void sym3_chunks2(const float4 *__restrict__ pos, float4 *__restrict__ slab_r_row, uint32_t n, uint32_t c0, uint32_t cnt,
                  const v2f (&xi)[SYM_P], const v2f (&yi)[SYM_P], const v2f (&zi)[SYM_P], const v2f (&mi)[SYM_P],
                  v2f (&ax)[SYM_P], v2f (&ay)[SYM_P], v2f (&az)[SYM_P], float eps2, float um_mass, float4 (*red)[4][64])
{
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    const int addr = (int)(((lane + 1u) & 63u) * 4u);
    const v2f e2 = {eps2, eps2};
    v2f bx[2 * SYM_P], by[2 * SYM_P], bz[2 * SYM_P];
#pragma unroll
    for (int k = 0; k < 2 * SYM_P; ++k) bx[k] = by[k] = bz[k] = (v2f){0.f, 0.f};
    auto fetch = [&](uint32_t c, v2f &x, v2f &y, v2f &z, v2f &m) {
        x = (v2f){PAD_XY, PAD_XY}; y = x; z = x; m = (v2f){0.f, 0.f};
        if (c < cnt) {
            const uint32_t j = (c0 + c) * SYM_CH + lane;
            if (j < n) { const float4 pj = pos[j]; x.x = pj.x; y.x = pj.y; z.x = pj.z; m.x = pj.w; }
        }
        if (c + 1 < cnt) {
            const uint32_t j = (c0 + c + 1) * SYM_CH + lane;
            if (j < n) { const float4 pj = pos[j]; x.y = pj.x; y.y = pj.y; z.y = pj.z; m.y = pj.w; }
        }
    };
    v2f xq, yq, zq, mq;
    fetch(0, xq, yq, zq, mq);
    for (uint32_t c = 0; c < cnt; c += 2) {
        v2f xn, yn, zn, mn;
        fetch(c + 2, xn, yn, zn, mn);
        v2f aqx = {0.f, 0.f}, aqy = {0.f, 0.f}, aqz = {0.f, 0.f};
#pragma unroll 2
        for (int step = 0; step < 64; ++step) {
            const v2f xr = {lane_rot(xq.x, addr), lane_rot(xq.y, addr)};
            const v2f yr = {lane_rot(yq.x, addr), lane_rot(yq.y, addr)};
            const v2f zr = {lane_rot(zq.x, addr), lane_rot(zq.y, addr)};
            v2f mr = {0.f, 0.f};
            if constexpr (!UM) mr = (v2f){lane_rot(mq.x, addr), lane_rot(mq.y, addr)};
#pragma unroll
            for (int p = 0; p < SYM_P; ++p) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float xs = h ? xi[p].y : xi[p].x, ys = h ? yi[p].y : yi[p].x, zs = h ? zi[p].y : zi[p].x;
                    const v2f dx = xq - (v2f){xs, xs}, dy = yq - (v2f){ys, ys}, dz = zq - (v2f){zs, zs};
                    v2f r2 = __builtin_elementwise_fma(dx, dx, e2);
                    r2 = __builtin_elementwise_fma(dy, dy, r2);
                    r2 = __builtin_elementwise_fma(dz, dz, r2);
                    v2f inv;
                    if constexpr (RSQ == RSQ_EXACT) inv = (v2f){__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                    else inv = quake_rsqrt2(r2);
                    const v2f inv3 = inv * (inv * inv);
                    v2f si = inv3, sj = inv3;
                    if constexpr (!UM) { const float ms = h ? mi[p].y : mi[p].x; si = mq * inv3; sj = (v2f){ms, ms} * inv3; }
                    bx[2 * p + h] = __builtin_elementwise_fma(si, dx, bx[2 * p + h]);
                    by[2 * p + h] = __builtin_elementwise_fma(si, dy, by[2 * p + h]);
                    bz[2 * p + h] = __builtin_elementwise_fma(si, dz, bz[2 * p + h]);
                    if constexpr (!DIAG) {
                        aqx = __builtin_elementwise_fma(-sj, dx, aqx);
                        aqy = __builtin_elementwise_fma(-sj, dy, aqy);
                        aqz = __builtin_elementwise_fma(-sj, dz, aqz);
                    }
                }
            }
            xq = xr; yq = yr; zq = zr;
            if constexpr (!UM) mq = mr;
            if constexpr (!DIAG) {
                aqx = (v2f){lane_rot(aqx.x, addr), lane_rot(aqx.y, addr)};
                aqy = (v2f){lane_rot(aqy.x, addr), lane_rot(aqy.y, addr)};
                aqz = (v2f){lane_rot(aqz.x, addr), lane_rot(aqz.y, addr)};
            }
        }
        if constexpr (!DIAG) {
            float4 r0 = make_float4(aqx.x, aqy.x, aqz.x, 0.f), r1 = make_float4(aqx.y, aqy.y, aqz.y, 0.f);
            if constexpr (UM) { r0.x *= um_mass; r0.y *= um_mass; r0.z *= um_mass; r1.x *= um_mass; r1.y *= um_mass; r1.z *= um_mass; }
            float4 (*rb)[4][64] = red + 2u * ((c >> 1) & 1u);         // double buffered: [half q0 | half q1][4 waves][64 lanes]
            rb[0][w][lane] = r0;
            rb[1][w][lane] = r1;
            __syncthreads();
            if (w < 2 && c + w < cnt) {                                // wave 0 stores chunk c, wave 1 chunk c + 1
                const uint32_t j = (c0 + c + w) * SYM_CH + lane;
                float4 a = rb[w][0][lane];
#pragma unroll
                for (int k = 1; k < 4; ++k) { a.x += rb[w][k][lane].x; a.y += rb[w][k][lane].y; a.z += rb[w][k][lane].z; }
                if (j < n) slab_r_row[j] = a;
            }
        }
        xq = xn; yq = yn; zq = zn;
        if constexpr (!UM) mq = mn;
    }
#pragma unroll
    for (int p = 0; p < SYM_P; ++p) {
        ax[p] += (v2f){bx[2 * p].x + bx[2 * p].y, bx[2 * p + 1].x + bx[2 * p + 1].y};
        ay[p] += (v2f){by[2 * p].x + by[2 * p].y, by[2 * p + 1].x + by[2 * p + 1].y};
        az[p] += (v2f){bz[2 * p].x + bz[2 * p].y, bz[2 * p + 1].x + bz[2 * p + 1].y};
    }
}

template <int RSQ, bool UM, bool PAIRS = false>
__global__ __launch_bounds__(BLOCK)
void force_sym3_f32(const float4 *__restrict__ pos, const SymItem *__restrict__ items,
                    float4 *__restrict__ slab_s, float4 *__restrict__ slab_r, uint32_t n, float eps2, float um_mass,
                    uint32_t *ticket, uint32_t first_wave, uint32_t ticket_base)
{
    __shared__ float4 red[PAIRS ? 4 : 2][4][64];
    const SymItem it = items[sym_item_index(ticket, first_wave, ticket_base)];
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    v2f xi[SYM_P], yi[SYM_P], zi[SYM_P], mi[SYM_P], ax[SYM_P], ay[SYM_P], az[SYM_P];
    uint32_t li[SYM_P];
#pragma unroll
    for (int p = 0; p < SYM_P; ++p) {
        li[p] = w * SYM_WT + (uint32_t)p * 128u + 2u * lane;
        const uint32_t g0 = it.tile * SYM_SB + li[p], g1 = g0 + 1;
        float4 p0 = make_float4(PAD_XY, PAD_XY, PAD_XY, 0.f), p1 = p0;
        if (g0 < n) p0 = pos[g0];
        if (g1 < n) p1 = pos[g1];
        xi[p] = (v2f){p0.x, p1.x}; yi[p] = (v2f){p0.y, p1.y}; zi[p] = (v2f){p0.z, p1.z}; mi[p] = (v2f){p0.w, p1.w};
        ax[p] = ay[p] = az[p] = (v2f){0.f, 0.f};
    }
    float4 *__restrict__ rrow = slab_r + it.r_base;
    if constexpr (PAIRS) {
        if (it.diag) sym3_chunks2<RSQ, UM, true>(pos, rrow, n, it.c0, it.cnt, xi, yi, zi, mi, ax, ay, az, eps2, um_mass, red);
        else         sym3_chunks2<RSQ, UM, false>(pos, rrow, n, it.c0, it.cnt, xi, yi, zi, mi, ax, ay, az, eps2, um_mass, red);
    } else {
        if (it.diag) sym3_chunks<RSQ, UM, true>(pos, rrow, n, it.c0, it.cnt, xi, yi, zi, mi, ax, ay, az, eps2, um_mass, red);
        else         sym3_chunks<RSQ, UM, false>(pos, rrow, n, it.c0, it.cnt, xi, yi, zi, mi, ax, ay, az, eps2, um_mass, red);
    }
    float4 *__restrict__ out = slab_s + (size_t)it.s_row * SYM_SB;
#pragma unroll
    for (int p = 0; p < SYM_P; ++p) {
        if constexpr (UM) { ax[p] *= um_mass; ay[p] *= um_mass; az[p] *= um_mass; }
        out[li[p]] = make_float4(ax[p].x, ay[p].x, az[p].x, 0.f);
        out[li[p] + 1] = make_float4(ax[p].y, ay[p].y, az[p].y, 0.f);
    }
}

// ---------------------------------------------------------------------------
// force_tiled3_f64 — one-sided 3-D kernel in double (one i per lane per register slot, like force_tiled_f64).
// ---------------------------------------------------------------------------
template <int P, bool GUARD, int UNROLL>
__global__ __launch_bounds__(BLOCK)
void force_tiled3_f64(const double4 *__restrict__ pos, double4 *__restrict__ partial,
                      uint32_t i_begin, uint32_t i_count, uint32_t j_begin, uint32_t j_end,
                      uint32_t js, uint32_t i_tiles, double eps2, uint32_t gap_begin, uint32_t gap_len)
{
    constexpr uint32_t IT = BLOCK * P;
    __shared__ double4 tile[2][TJ];
    const double k15 = vgpr_const(1.5), k1875 = vgpr_const(1.875);
    const TileMap tm = decode_block(blockIdx.x, i_tiles, js);
    if (!tm.valid) return;
    const uint32_t t = threadIdx.x;
    const uint32_t jn = j_end - j_begin;
    const uint32_t slice_len = (((jn + js - 1) / js + TJ - 1) / TJ) * TJ;
    const uint32_t s0 = j_begin + min(tm.slice * slice_len, jn);
    const uint32_t s1 = j_begin + min((tm.slice + 1) * slice_len, jn);
    double xi[P], yi[P], zi[P], ax[P], ay[P], az[P];
    uint32_t li[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        li[p] = tm.i_tile * IT + (uint32_t)p * BLOCK + t;
        const double4 p0 = pos[i_begin + min(li[p], i_count - 1)];
        xi[p] = p0.x; yi[p] = p0.y; zi[p] = p0.z; ax[p] = ay[p] = az[p] = 0.0;
    }
    const uint32_t ntiles = (s1 - s0 + TJ - 1) / TJ;
    const double4 none = make_double4(0.0, 0.0, 0.0, 0.0);        // mass 0: contributes nothing (eps > 0 or guarded)
    {
        const uint32_t j = s0 + t;
        tile[0][t] = j < s1 ? pos[j + (j >= gap_begin ? gap_len : 0u)] : none;
    }
    __syncthreads();
    for (uint32_t it = 0; it < ntiles; ++it) {
        const uint32_t jn1 = s0 + (it + 1) * TJ + t;
        const double4 qn = jn1 < s1 ? pos[jn1 + (jn1 >= gap_begin ? gap_len : 0u)] : none;
        const double4 *__restrict__ cur = tile[it & 1];
#pragma unroll UNROLL
        for (int jj = 0; jj < TJ; ++jj) {
            const double4 q = cur[jj];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const double dx = q.x - xi[p], dy = q.y - yi[p], dz = q.z - zi[p];
                double r2, inv3;
                if constexpr (GUARD) {
                    r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
                    inv3 = r2 > 0.0 ? rsqrt3_f64(r2, k15, k1875) : 0.0;
                } else {
                    r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2)));
                    inv3 = rsqrt3_f64(r2, k15, k1875);
                }
                const double sc = q.w * inv3;
                ax[p] = __builtin_fma(sc, dx, ax[p]);
                ay[p] = __builtin_fma(sc, dy, ay[p]);
                az[p] = __builtin_fma(sc, dz, az[p]);
            }
        }
        if (it + 1 < ntiles) tile[(it + 1) & 1][t] = qn;
        __syncthreads();
    }
    double4 *__restrict__ out = partial + (size_t)tm.slice * i_count;
#pragma unroll
    for (int p = 0; p < P; ++p)
        if (li[p] < i_count) out[li[p]] = make_double4(ax[p], ay[p], az[p], 0.0);
}

// ---------------------------------------------------------------------------
// force_sym3_f64 — the symmetric scheme in 3-D double precision: same items, tiles and slabs as the other
// symmetric kernels; a lane holds SYM_P64 = 8 stationary particles as scalars, the travelling particle (x, y, z
// [, m]) and its accumulator rotate as pairs of 32-bit halves (12-14 ds_bpermute_b32 per step).
// Body, both directions: 3 add + 3 fma + v_rsq_f64 + 6 (cube correction) + 6 fma (+2 mul with individual masses).
// ---------------------------------------------------------------------------
template <bool UM, bool DIAG>
__device__ __forceinline__
void sym3_chunks_f64(const double4 *__restrict__ pos, double4 *__restrict__ slab_r_row, uint32_t n, uint32_t c0, uint32_t cnt,
                     const double (&xi)[SYM_P64], const double (&yi)[SYM_P64], const double (&zi)[SYM_P64], const double (&mi)[SYM_P64],
                     double (&ax)[SYM_P64], double (&ay)[SYM_P64], double (&az)[SYM_P64], double eps2, double um_mass,
                     double4 (*red)[4][64])
{
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    const int addr = (int)(((lane + 1u) & 63u) * 4u);
    const double k15 = vgpr_const(1.5), k1875 = vgpr_const(1.875);
    double xq = PAD_XY64, yq = PAD_XY64, zq = PAD_XY64, mq = 0.0;
    {
        const uint32_t j = c0 * SYM_CH + lane;
        if (j < n) { const double4 pj = pos[j]; xq = pj.x; yq = pj.y; zq = pj.z; mq = pj.w; }
    }
    for (uint32_t c = 0; c < cnt; ++c) {
        double xn = PAD_XY64, yn = PAD_XY64, zn = PAD_XY64, mn = 0.0;
        {
            const uint32_t j = (c0 + c + 1) * SYM_CH + lane;
            if (c + 1 < cnt && j < n) { const double4 pj = pos[j]; xn = pj.x; yn = pj.y; zn = pj.z; mn = pj.w; }
        }
        double aqx = 0.0, aqy = 0.0, aqz = 0.0;
        for (int step = 0; step < 64; ++step) {
            const double xr = lane_rot64(xq, addr), yr = lane_rot64(yq, addr), zr = lane_rot64(zq, addr);
            double mr = 0.0;
            if constexpr (!UM) mr = lane_rot64(mq, addr);
#pragma unroll
            for (int p = 0; p < SYM_P64; ++p) {
                const double dx = xq - xi[p], dy = yq - yi[p], dz = zq - zi[p];
                const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2)));
                const double inv3 = rsqrt3_f64(r2, k15, k1875);
                double si = inv3, sj = inv3;
                if constexpr (!UM) { si = mq * inv3; sj = mi[p] * inv3; }
                ax[p] = __builtin_fma(si, dx, ax[p]);
                ay[p] = __builtin_fma(si, dy, ay[p]);
                az[p] = __builtin_fma(si, dz, az[p]);
                if constexpr (!DIAG) {
                    aqx = __builtin_fma(-sj, dx, aqx);
                    aqy = __builtin_fma(-sj, dy, aqy);
                    aqz = __builtin_fma(-sj, dz, aqz);
                }
            }
            xq = xr; yq = yr; zq = zr;
            if constexpr (!UM) mq = mr;
            if constexpr (!DIAG) { aqx = lane_rot64(aqx, addr); aqy = lane_rot64(aqy, addr); aqz = lane_rot64(aqz, addr); }
        }
        if constexpr (!DIAG) {
            double4 r = make_double4(aqx, aqy, aqz, 0.0);
            if constexpr (UM) { r.x *= um_mass; r.y *= um_mass; r.z *= um_mass; }
            double4 (*rb)[64] = red[c & 1u];
            rb[w][lane] = r;
            __syncthreads();
            if (w == 0) {
                const uint32_t j = (c0 + c) * SYM_CH + lane;
                double4 a = rb[0][lane];
#pragma unroll
                for (int k = 1; k < 4; ++k) { a.x += rb[k][lane].x; a.y += rb[k][lane].y; a.z += rb[k][lane].z; }
                if (j < n) slab_r_row[j] = a;
            }
        }
        xq = xn; yq = yn; zq = zn;
        if constexpr (!UM) mq = mn;
    }
}

template <bool UM>
__global__ __launch_bounds__(BLOCK)
void force_sym3_f64(const double4 *__restrict__ pos, const SymItem *__restrict__ items,
                    double4 *__restrict__ slab_s, double4 *__restrict__ slab_r, uint32_t n, double eps2, double um_mass,
                    uint32_t *ticket, uint32_t first_wave, uint32_t ticket_base)
{
    __shared__ double4 red[2][4][64];
    const SymItem it = items[sym_item_index(ticket, first_wave, ticket_base)];
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    double xi[SYM_P64], yi[SYM_P64], zi[SYM_P64], mi[SYM_P64], ax[SYM_P64], ay[SYM_P64], az[SYM_P64];
    uint32_t li[SYM_P64];
#pragma unroll
    for (int p = 0; p < SYM_P64; ++p) {
        li[p] = w * SYM_WT + (uint32_t)p * 64u + lane;
        const uint32_t g = it.tile * SYM_SB + li[p];
        xi[p] = yi[p] = zi[p] = PAD_XY64; mi[p] = 0.0;
        if (g < n) { const double4 q = pos[g]; xi[p] = q.x; yi[p] = q.y; zi[p] = q.z; mi[p] = q.w; }
        ax[p] = ay[p] = az[p] = 0.0;
    }
    double4 *__restrict__ rrow = slab_r + it.r_base;
    if (it.diag) sym3_chunks_f64<UM, true>(pos, rrow, n, it.c0, it.cnt, xi, yi, zi, mi, ax, ay, az, eps2, um_mass, red);
    else         sym3_chunks_f64<UM, false>(pos, rrow, n, it.c0, it.cnt, xi, yi, zi, mi, ax, ay, az, eps2, um_mass, red);
    double4 *__restrict__ out = slab_s + (size_t)it.s_row * SYM_SB;
#pragma unroll
    for (int p = 0; p < SYM_P64; ++p) {
        if constexpr (UM) { ax[p] *= um_mass; ay[p] *= um_mass; az[p] *= um_mass; }
        out[li[p]] = make_double4(ax[p], ay[p], az[p], 0.0);
    }
}

// sym_gather3 — the 3-D twin of sym_gather (nb_kernels.hip.h): sum of the stationary rows of particle k's tile +
// the entries of its tile's coverage list, over particles [k0, k0 + kn); FUSE applies kick and drift (adding
// `base`, the reduce-scattered sum of a sharded rank, when given), otherwise the sum is stored to acc_sum[k].
template <typename real, bool FUSE>
__global__ __launch_bounds__(BLOCK)
void sym_gather3(const typename vec4_of<real>::type *__restrict__ slab_s, const typename vec4_of<real>::type *__restrict__ slab_r,
                 const uint32_t *__restrict__ row_lo, const uint32_t *__restrict__ row_hi,
                 const uint32_t *__restrict__ cov_begin, const SymCov *__restrict__ cov, uint32_t n, uint32_t k0, uint32_t kn,
                 typename vec4_of<real>::type *__restrict__ acc_sum, const typename vec4_of<real>::type *__restrict__ base,
                 const typename vec4_of<real>::type *__restrict__ pos_cur, typename vec4_of<real>::type *__restrict__ pos_next,
                 typename vec4_of<real>::type *__restrict__ vel, typename vec4_of<real>::type *__restrict__ acc,
                 real dt_kick, real dt_drift, int flags)
{
    typedef typename vec4_of<real>::type real4;
    __shared__ real4 part[GATHER_Q][GATHER_T];      // one particle per thread here: the elements are 16 / 32 bytes already
    const uint32_t p = threadIdx.x % GATHER_T, q = threadIdx.x / GATHER_T;
    const uint32_t li = blockIdx.x * GATHER_T + p, k = k0 + li;
    real4 a = make_real4<real>(0, 0, 0, 0);
    if (li < kn) {
        const uint32_t g = k / SYM_SB, loc = k % SYM_SB;
        const uint32_t r1 = row_hi[g];
#pragma unroll 4      // four independent loads in flight per thread; the adds keep their order
        for (uint32_t r = row_lo[g] + q; r < r1; r += GATHER_Q) {
            const real4 b = slab_s[(size_t)r * SYM_SB + loc];
            a.x += b.x; a.y += b.y; a.z += b.z;
        }
        // coverage entries four at a time, as in sym_gather_block (nb_kernels.hip.h): descriptors, then the partials they point at,
        // as independent batches; an entry that does not cover k reads element 0 of the slab and adds nothing; same order of adds
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
            real4 b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                in[u] = in[u] && k >= cv[u].lo && k < cv[u].hi;
                b[u] = slab_r[in[u] ? cv[u].base + (int64_t)k : (int64_t)0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (in[u]) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; }
        }
    }
    part[q][p] = a;
    __syncthreads();
    if (q == 0 && li < kn) {
        real4 s = part[0][p];
#pragma unroll
        for (int j = 1; j < GATHER_Q; ++j) { s.x += part[j][p].x; s.y += part[j][p].y; s.z += part[j][p].z; }
        if (base) { const real4 b = base[li]; s.x += b.x; s.y += b.y; s.z += b.z; }
        if constexpr (FUSE) kick_drift_one3<real>(s, li, pos_cur, pos_next, vel, acc, k0, dt_kick, dt_drift, flags);
        else { s.w = 0; acc_sum[k] = s; }
    }
}

// ---------------------------------------------------------------------------
// AoS (64-byte records, z in the first padding slot of pos / vel / acc) <-> SoA real4
// ---------------------------------------------------------------------------
template <typename real>
__global__ __launch_bounds__(BLOCK)
void unpack_bodies3(const BodyRec *__restrict__ aos, uint32_t n, typename vec4_of<real>::type *__restrict__ pos,
                    typename vec4_of<real>::type *__restrict__ vel, typename vec4_of<real>::type *__restrict__ acc,
                    float *__restrict__ radius, uint32_t i_begin, uint32_t i_count)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 p = aos[i].q[0], m = aos[i].q[3];
    pos[i] = make_real4<real>((real)p.x, (real)p.y, (real)p.z, (real)m.x);
    radius[i] = m.y;
    if (i >= i_begin && i - i_begin < i_count) {
        const float4 v = aos[i].q[1], a = aos[i].q[2];
        vel[i - i_begin] = make_real4<real>((real)v.x, (real)v.y, (real)v.z, 0);
        acc[i - i_begin] = make_real4<real>((real)a.x, (real)a.y, (real)a.z, 0);
    }
}

template <typename real>
__global__ __launch_bounds__(BLOCK)
void pack_bodies3(BodyRec *__restrict__ aos, const typename vec4_of<real>::type *__restrict__ pos,
                  const typename vec4_of<real>::type *__restrict__ vel, const typename vec4_of<real>::type *__restrict__ acc,
                  const float *__restrict__ radius, uint32_t i_begin, uint32_t i_count)
{
    const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
    if (li >= i_count) return;
    const auto p = pos[i_begin + li], v = vel[li], a = acc[li];
    BodyRec r;
    r.q[0] = make_float4((float)p.x, (float)p.y, (float)p.z, 0.f);
    r.q[1] = make_float4((float)v.x, (float)v.y, (float)v.z, 0.f);
    r.q[2] = make_float4((float)a.x, (float)a.y, (float)a.z, 0.f);
    r.q[3] = make_float4((float)p.w, radius[i_begin + li], 0.f, 0.f);
    aos[li] = r;
}

// positions only, as packed (x, y, z) floats: the viewer's fast path in 3-D (12 bytes per body)
template <typename real>
__global__ __launch_bounds__(BLOCK)
void pack_positions3(float *__restrict__ out, const typename vec4_of<real>::type *__restrict__ pos, uint32_t i_begin, uint32_t i_count)
{
    const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
    if (li >= i_count) return;
    const auto p = pos[i_begin + li];
    out[3 * (size_t)li + 0] = (float)p.x;
    out[3 * (size_t)li + 1] = (float)p.y;
    out[3 * (size_t)li + 2] = (float)p.z;
}

// energy in fp64: K = sum m v^2 / 2, U = - sum_i m_i sum_{j > i} m_j / sqrt(r^2 + eps^2) (every unordered pair once)
template <typename real>
__global__ __launch_bounds__(BLOCK)
void energy_partials3(const typename vec4_of<real>::type *__restrict__ pos, const typename vec4_of<real>::type *__restrict__ vel,
                      uint32_t n, uint32_t i_begin, uint32_t i_count, double eps2, double *__restrict__ ksum, double *__restrict__ usum)
{
    struct alignas(16) JD { double x, y, z, m; };
    __shared__ JD tile[TJ];
    __shared__ double red[2][BLOCK / 64];
    const uint32_t t = threadIdx.x, li = blockIdx.x * BLOCK + t;
    const bool live = li < i_count;
    const uint32_t gi = i_begin + (live ? li : i_count - 1);
    const auto pi = pos[gi];
    const double xi = pi.x, yi = pi.y, zi = pi.z;
    const double k0375 = vgpr_const(0.375);
    double u = 0.0;
    const uint32_t first = ((i_begin + blockIdx.x * BLOCK) / TJ) * TJ;
    for (uint32_t j0 = first; j0 < n; j0 += TJ) {
        const uint32_t j = j0 + t;
        __syncthreads();
        if (j < n) { const auto q = pos[j]; tile[t] = JD{(double)q.x, (double)q.y, (double)q.z, (double)q.w}; }
        else tile[t] = JD{0.0, 0.0, 0.0, 0.0};
        __syncthreads();
        const uint32_t cnt = min((uint32_t)TJ, n - j0);
        for (uint32_t jj = 0; jj < cnt; ++jj) {
            const double dx = tile[jj].x - xi, dy = tile[jj].y - yi, dz = tile[jj].z - zi;
            const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2)));
            const double wgt = (j0 + jj > gi) ? tile[jj].m : 0.0;
            u = __builtin_fma(wgt, rsqrt_f64(r2, k0375), u);
        }
    }
    double k = 0.0, uu = 0.0;
    if (live) {
        const double m = (double)pi.w;
        const auto v = vel[li];
        k = 0.5 * m * ((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z);
        uu = -m * u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { k += __shfl_down(k, off, 64); uu += __shfl_down(uu, off, 64); }
    if ((t & 63) == 0) { red[0][t >> 6] = k; red[1][t >> 6] = uu; }
    __syncthreads();
    if (t == 0) {
        double ks = 0.0, us = 0.0;
        for (int wv = 0; wv < BLOCK / 64; ++wv) { ks += red[0][wv]; us += red[1][wv]; }
        ksum[blockIdx.x] = ks;
        usum[blockIdx.x] = us;
    }
}

// momentum of the owned block (see momentum_partials): px | py | pz | Lz (z component of x cross m v)
template <typename real>
__global__ __launch_bounds__(BLOCK)
void momentum_partials3(const typename vec4_of<real>::type *__restrict__ pos, const typename vec4_of<real>::type *__restrict__ vel,
                        uint32_t i_begin, uint32_t i_count, double *__restrict__ psum)
{
    const uint32_t li = blockIdx.x * BLOCK + threadIdx.x;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (li < i_count) {
        const auto p = pos[i_begin + li];
        const auto w = vel[li];
        const double m = (double)p.w;
        v[0] = m * (double)w.x; v[1] = m * (double)w.y; v[2] = m * (double)w.z;
        v[3] = m * ((double)p.x * (double)w.y - (double)p.y * (double)w.x);
    }
    block_reduce4(v, psum, gridDim.x);
}

} // namespace nbk
