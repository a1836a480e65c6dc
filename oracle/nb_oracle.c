/* oracle/nb_oracle.c — TEST INFRASTRUCTURE ONLY (see nb_oracle.h).
 *
 * CPU restatement of the reference hot path.  Build (oracle/Makefile):
 *   gcc -O3 -march=native -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * -ffp-contract=off matters: the golden vectors come from a reference build
 * without FMA contraction, and GCC's default (-ffp-contract=fast) would fuse
 * a*b+c under -march=native.
 *
 * Vectorisation is ACROSS i (IB independent running sums, one per i); every
 * running sum still adds its j terms in ascending j with the reference's
 * exact operation order, so the result is bit-identical to the scalar loop of
 * Quadtree.hpp:134-144 whatever the SIMD width or thread count.
 */
#include "nb_oracle.h"

#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IB 16 /* i-block = SIMD lanes worth of independent running sums */

static int g_threads = 0; /* 0 = OpenMP default */

int nbo_set_threads(int nthreads)
{
    g_threads = nthreads > 0 ? nthreads : 0;
    return nbo_get_threads();
}

int nbo_get_threads(void)
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

/* Quadtree.hpp:106-111:
 *   y = bit_cast<float>(0x5f3759df - (bit_cast<uint32_t>(number) >> 1));
 *   return y * (1.5f - (number * 0.5f * y * y));
 * C evaluates number*0.5f*y*y left to right: ((number*0.5f)*y)*y. */
static inline float quake_rsqrt(float number)
{
    uint32_t u;
    float y;
    memcpy(&u, &number, 4);
    u = 0x5f3759dfu - (u >> 1);
    memcpy(&y, &u, 4);
    return y * (1.5f - (number * 0.5f * y * y));
}

float nbo_fast_inv_sqrt(float x) { return quake_rsqrt(x); }

/* One i-block of the pairwise loop, Quadtree.hpp:134-144:
 *   Vec2 r = body.pos - pos;              (Vec2.hpp:87-95, per-component sub)
 *   float r_sq = r.mag_sq();              (Vec2.hpp:216-219: x*x + y*y)
 *   if (r_sq > 0) {
 *     float inv_dist = fast_inv_sqrt(r_sq + e_sq);
 *     float inv_dist_cubed = inv_dist * inv_dist * inv_dist;
 *     acc += r * (body.mass * inv_dist_cubed);   (Vec2.hpp:97-105 then :139-144)
 *   }
 */
#define ACCEL_BLOCK(T, RSQRT)                                                  \
    do {                                                                       \
        T xi[IB], yi[IB], sx[IB], sy[IB];                                      \
        for (int k = 0; k < IB; ++k) {                                         \
            size_t ii = ib + (size_t)k < i_end ? ib + (size_t)k : i_end - 1;   \
            xi[k] = x[ii]; yi[k] = y[ii]; sx[k] = 0; sy[k] = 0;                \
        }                                                                      \
        for (size_t j = 0; j < n; ++j) {                                       \
            const T xj = x[j], yj = y[j], mj = m[j];                           \
            _Pragma("omp simd")                                                \
            for (int k = 0; k < IB; ++k) {                                     \
                T rx = xj - xi[k];                                             \
                T ry = yj - yi[k];                                             \
                T r_sq = rx * rx + ry * ry;                                    \
                T inv = RSQRT(r_sq + eps2);                                    \
                T inv3 = inv * inv * inv;                                      \
                T s = mj * inv3;                                               \
                T cx = rx * s, cy = ry * s;                                    \
                sx[k] = r_sq > 0 ? sx[k] + cx : sx[k];                         \
                sy[k] = r_sq > 0 ? sy[k] + cy : sy[k];                         \
            }                                                                  \
        }                                                                      \
        for (int k = 0; k < IB && ib + (size_t)k < i_end; ++k) {               \
            ax[ib + k] = sx[k]; ay[ib + k] = sy[k];                            \
        }                                                                      \
    } while (0)

#define RSQ_EXACT_F32(v) (1.0f / sqrtf(v))
#define RSQ_EXACT_F64(v) (1.0 / sqrt(v))

void nbo_accel_f32(size_t n, const float *x, const float *y, const float *m,
                   float eps2, int rsqrt_mode, size_t i_begin, size_t i_end,
                   float *ax, float *ay)
{
    if (i_end <= i_begin) return;
    const long nblk = (long)((i_end - i_begin + IB - 1) / IB);
    const int nt = nbo_get_threads();
    (void)nt;
    if (rsqrt_mode == NBO_RSQRT_QUAKE) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt)
        for (long b = 0; b < nblk; ++b) {
            const size_t ib = i_begin + (size_t)b * IB;
            ACCEL_BLOCK(float, quake_rsqrt);
        }
    } else {
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt)
        for (long b = 0; b < nblk; ++b) {
            const size_t ib = i_begin + (size_t)b * IB;
            ACCEL_BLOCK(float, RSQ_EXACT_F32);
        }
    }
}

/* The reference's per-pair TERMS (every operation in fp32, Quake or exact rsqrt, exactly as above) summed in
 * double instead of the reference's single fp32 running sum: what Quadtree.hpp:134-144 would return without its
 * own summation rounding (which grows ~ sqrt(n) * 6e-8 and reaches ~2e-5 of the force at n = 262 144).  Used to
 * tell a kernel's error from the reference's own noise at large n; not a reference result in itself. */
void nbo_accel_f32_terms_acc64(size_t n, const float *x, const float *y, const float *m,
                               float eps2, int rsqrt_mode, size_t i_begin, size_t i_end,
                               double *ax, double *ay)
{
    if (i_end <= i_begin) return;
    const int nt = nbo_get_threads();
    (void)nt;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nt)
    for (long i = (long)i_begin; i < (long)i_end; ++i) {
        const float xi = x[i], yi = y[i];
        double sx = 0.0, sy = 0.0;
        for (size_t j = 0; j < n; ++j) {
            const float rx = x[j] - xi, ry = y[j] - yi;
            const float r_sq = rx * rx + ry * ry;
            if (r_sq > 0) {
                const float inv = rsqrt_mode == NBO_RSQRT_QUAKE ? quake_rsqrt(r_sq + eps2) : RSQ_EXACT_F32(r_sq + eps2);
                const float inv3 = inv * inv * inv;
                const float s = m[j] * inv3;
                const float cx = rx * s, cy = ry * s;
                sx += (double)cx; sy += (double)cy;
            }
        }
        ax[i] = sx; ay[i] = sy;
    }
}

void nbo_accel_f64(size_t n, const double *x, const double *y, const double *m,
                   double eps2, size_t i_begin, size_t i_end,
                   double *ax, double *ay)
{
    if (i_end <= i_begin) return;
    const long nblk = (long)((i_end - i_begin + IB - 1) / IB);
    const int nt = nbo_get_threads();
    (void)nt;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt)
    for (long b = 0; b < nblk; ++b) {
        const size_t ib = i_begin + (size_t)b * IB;
        ACCEL_BLOCK(double, RSQ_EXACT_F64);
    }
}

/* Simulation.hpp:129-163 (the three loops of iterate(), after attract()).
 * extras=0 keeps only the lines that act on unit-scale data:
 *   :130-131  vel += acc*dt          :161-162  pos += vel*dt
 * extras=1 adds :133-137 (|v| clamp at 1000) and :140-155 (soft boundary). */
void nbo_kick_drift_f32(size_t n, float *x, float *y, float *vx, float *vy,
                        const float *ax, const float *ay, float dt, int extras)
{
    const float BOUNDARY_RADIUS = 100000.0f;
    const float SOFT_BOUNDARY = BOUNDARY_RADIUS * 0.8f;
    const float BOUNDARY_FORCE = 0.9f;
    const float DAMPING = 0.9995f;
    const float MAX_VELOCITY = 1000.0f;

    for (size_t i = 0; i < n; ++i) {
        vx[i] += ax[i] * dt;
        vy[i] += ay[i] * dt;
        if (extras) {
            float velMagSq = vx[i] * vx[i] + vy[i] * vy[i];
            if (velMagSq > MAX_VELOCITY * MAX_VELOCITY) {
                float scale = MAX_VELOCITY / sqrtf(velMagSq);
                vx[i] *= scale;
                vy[i] *= scale;
            }
        }
    }
    if (extras) {
        const float SOFT_BOUNDARY_SQ = SOFT_BOUNDARY * SOFT_BOUNDARY;
        for (size_t i = 0; i < n; ++i) {
            float distSq = x[i] * x[i] + y[i] * y[i];
            if (distSq > SOFT_BOUNDARY_SQ) {
                float dist = sqrtf(distSq);
                float ratio = dist / SOFT_BOUNDARY;
                float force = BOUNDARY_FORCE * expf(ratio - 1.0f);
                float k = -1.0f / dist;
                float dirx = x[i] * k, diry = y[i] * k;
                float fdt = force * dt;
                vx[i] += dirx * fdt;
                vy[i] += diry * fdt;
                vx[i] *= DAMPING;
                vy[i] *= DAMPING;
            }
        }
    }
    for (size_t i = 0; i < n; ++i) {
        x[i] += vx[i] * dt;
        y[i] += vy[i] * dt;
    }
}

void nbo_step_f32(size_t n, float *x, float *y, float *vx, float *vy,
                  const float *m, float *ax, float *ay,
                  float eps2, float dt, int nsteps, int rsqrt_mode, int extras)
{
    for (int s = 0; s < nsteps; ++s) {
        nbo_accel_f32(n, x, y, m, eps2, rsqrt_mode, 0, n, ax, ay);
        nbo_kick_drift_f32(n, x, y, vx, vy, ax, ay, dt, extras);
    }
}

void nbo_step_f64(size_t n, double *x, double *y, double *vx, double *vy,
                  const double *m, double *ax, double *ay,
                  double eps2, double dt, int nsteps)
{
    for (int s = 0; s < nsteps; ++s) {
        nbo_accel_f64(n, x, y, m, eps2, 0, n, ax, ay);
        for (size_t i = 0; i < n; ++i) {
            vx[i] += ax[i] * dt;
            vy[i] += ay[i] * dt;
        }
        for (size_t i = 0; i < n; ++i) {
            x[i] += vx[i] * dt;
            y[i] += vy[i] * dt;
        }
    }
}

void nbo_energy_f64(size_t n, const double *x, const double *y,
                    const double *vx, const double *vy, const double *m,
                    double eps2, double *kinetic, double *potential)
{
    double K = 0.0, U = 0.0;
    for (size_t i = 0; i < n; ++i)
        K += 0.5 * m[i] * (vx[i] * vx[i] + vy[i] * vy[i]);
    const int nt = nbo_get_threads();
    (void)nt;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : U) num_threads(nt)
    for (long i = 0; i < (long)n; ++i) {
        double u = 0.0;
        for (size_t j = (size_t)i + 1; j < n; ++j) {
            double rx = x[j] - x[i], ry = y[j] - y[i];
            u += m[j] / sqrt(rx * rx + ry * ry + eps2);
        }
        U -= m[i] * u;
    }
    *kinetic = K;
    *potential = U;
}

/* ---- 3-D extension: same arithmetic with a z term, fp64 (no reference to pin to) ---- */
void nbo_accel3_f64(size_t n, const double *x, const double *y, const double *z, const double *m,
                    double eps2, double *ax, double *ay, double *az)
{
    const int nt = nbo_get_threads();
    (void)nt;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nt)
    for (long i = 0; i < (long)n; ++i) {
        double sx = 0.0, sy = 0.0, sz = 0.0;
        for (size_t j = 0; j < n; ++j) {
            const double rx = x[j] - x[i], ry = y[j] - y[i], rz = z[j] - z[i];
            const double r_sq = rx * rx + ry * ry + rz * rz;
            if (r_sq > 0) {
                const double inv = 1.0 / sqrt(r_sq + eps2);
                const double s = m[j] * (inv * inv * inv);
                sx += rx * s; sy += ry * s; sz += rz * s;
            }
        }
        ax[i] = sx; ay[i] = sy; az[i] = sz;
    }
}

void nbo_step3_f64(size_t n, double *x, double *y, double *z, double *vx, double *vy, double *vz,
                   const double *m, double *ax, double *ay, double *az, double eps2, double dt, int nsteps)
{
    for (int s = 0; s < nsteps; ++s) {
        nbo_accel3_f64(n, x, y, z, m, eps2, ax, ay, az);
        for (size_t i = 0; i < n; ++i) { vx[i] += ax[i] * dt; vy[i] += ay[i] * dt; vz[i] += az[i] * dt; }
        for (size_t i = 0; i < n; ++i) { x[i] += vx[i] * dt; y[i] += vy[i] * dt; z[i] += vz[i] * dt; }
    }
}

void nbo_energy3_f64(size_t n, const double *x, const double *y, const double *z,
                     const double *vx, const double *vy, const double *vz, const double *m,
                     double eps2, double *kinetic, double *potential)
{
    double K = 0.0, U = 0.0;
    for (size_t i = 0; i < n; ++i) K += 0.5 * m[i] * (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]);
    const int nt = nbo_get_threads();
    (void)nt;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : U) num_threads(nt)
    for (long i = 0; i < (long)n; ++i) {
        double u = 0.0;
        for (size_t j = (size_t)i + 1; j < n; ++j) {
            const double rx = x[j] - x[i], ry = y[j] - y[i], rz = z[j] - z[i];
            u += m[j] / sqrt(rx * rx + ry * ry + rz * rz + eps2);
        }
        U -= m[i] * u;
    }
    *kinetic = K;
    *potential = U;
}
