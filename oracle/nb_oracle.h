/* oracle/nb_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, no HIP, no product code) of the reference's
 * pairwise softened-gravity + kick/drift path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product library (libnbody_hip.so) never links or calls it.
 *
 * Every function cites the reference lines it restates
 * (paths relative to /root/reference/Nbodysim/headers).
 *
 * Pinning: validated bit-for-bit against the compiled reference
 * (oracle/_ref/libnbref.so, built from /root/reference by oracle/Makefile)
 * and against the committed fixtures tests/golden/ that the compiled reference
 * produced (oracle/make_golden.py).  The reference itself ships no tests or
 * golden vectors, so those fixtures are the only pin.
 */
#ifndef NB_ORACLE_H
#define NB_ORACLE_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1/sqrt flavour */
enum { NBO_RSQRT_EXACT = 0,   /* 1/sqrt(x), IEEE */
       NBO_RSQRT_QUAKE = 1 }; /* Quadtree::fast_inv_sqrt, Quadtree.hpp:106-111 */

/* Quadtree.hpp:106-111 */
float nbo_fast_inv_sqrt(float x);

/* threads used by the O(N^2) loops (OpenMP over i-blocks); returns the count in effect */
int nbo_set_threads(int nthreads);
int nbo_get_threads(void);

/* Pairwise accelerations for i in [i_begin, i_end) against ALL j in [0, n),
 * j ascending, one running sum per i, no FMA contraction — the operation
 * order of Quadtree.hpp:134-144.  ax/ay are indexed by absolute i. */
void nbo_accel_f32(size_t n, const float *x, const float *y, const float *m,
                   float eps2, int rsqrt_mode, size_t i_begin, size_t i_end,
                   float *ax, float *ay);
void nbo_accel_f64(size_t n, const double *x, const double *y, const double *m,
                   double eps2, size_t i_begin, size_t i_end,
                   double *ax, double *ay);
/* the reference's fp32 per-pair terms summed in double (its arithmetic without its summation rounding) */
void nbo_accel_f32_terms_acc64(size_t n, const float *x, const float *y, const float *m,
                               float eps2, int rsqrt_mode, size_t i_begin, size_t i_end,
                               double *ax, double *ay);

/* nsteps of: accel; v += a*dt; x += v*dt  (Simulation.hpp:117,129-131,160-163).
 * extras != 0 also applies the velocity clamp and the soft boundary of
 * Simulation.hpp:133-155.  ax/ay hold the last evaluated acceleration. */
void nbo_step_f32(size_t n, float *x, float *y, float *vx, float *vy,
                  const float *m, float *ax, float *ay,
                  float eps2, float dt, int nsteps, int rsqrt_mode, int extras);
void nbo_step_f64(size_t n, double *x, double *y, double *vx, double *vy,
                  const double *m, double *ax, double *ay,
                  double eps2, double dt, int nsteps);

/* Kick/drift only (no force evaluation), Simulation.hpp:129-163. */
void nbo_kick_drift_f32(size_t n, float *x, float *y, float *vx, float *vy,
                        const float *ax, const float *ay, float dt, int extras);

/* Total energy in fp64, softening-consistent with Quadtree.hpp:140-142:
 * K = sum 1/2 m v^2 ; U = -sum_{i<j} m_i m_j / sqrt(r^2 + eps^2). */
void nbo_energy_f64(size_t n, const double *x, const double *y,
                    const double *vx, const double *vy, const double *m,
                    double eps2, double *kinetic, double *potential);

/* ---- 3-D build extension ----------------------------------------
 * The reference is 2-D only; these restate the SAME softened pair force with a z term
 * (20 flop per pair) in fp64.  There is no reference output to pin them to: the 3-D
 * parity claims are "against the fp64 restatement" only (parity unpinned). */
void nbo_accel3_f64(size_t n, const double *x, const double *y, const double *z, const double *m,
                    double eps2, double *ax, double *ay, double *az);
void nbo_step3_f64(size_t n, double *x, double *y, double *z, double *vx, double *vy, double *vz,
                   const double *m, double *ax, double *ay, double *az, double eps2, double dt, int nsteps);
void nbo_energy3_f64(size_t n, const double *x, const double *y, const double *z,
                     const double *vx, const double *vy, const double *vz, const double *m,
                     double eps2, double *kinetic, double *potential);

#ifdef __cplusplus
}
#endif
#endif
