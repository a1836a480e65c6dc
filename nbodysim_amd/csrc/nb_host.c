/* nb_host.c — host-only pieces of libnbody_hip.so (plain C, no HIP):
 * synthetic initial conditions and the build-defined dump format.
 *
 * The reference has neither a Plummer generator nor any file I/O;
 * both are defined here and documented in DESIGN.md as build-defined.  Its own
 * ICs (Simulation::uniform_disc, Simulation.hpp:347-603, a Lorenz-attractor
 * trace) are restated in nb_default_ics.
 */
#define _FILE_OFFSET_BITS 64
#define _POSIX_C_SOURCE 200809L
#include "nbody.h"
#include "nb_internal.h"

#include <errno.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

/* ---- errors: text and NB_E* code of the last failure on the calling thread ---- */
static _Thread_local char g_err[512] = "";
static _Thread_local int g_err_code = NB_OK;

void nb_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    g_err_code = NB_EINVAL;
}

int nb_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    g_err_code = code;
    return code;
}

void nb_clear_error(void) { g_err[0] = 0; g_err_code = NB_OK; }
const char *nb_last_error(void) { return g_err; }
int nb_last_error_code(void) { return g_err_code; }
int nb_abi_version(void) { return NB_ABI_VERSION; }

void nb_params_default(nb_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->struct_size = (uint32_t)sizeof(nb_params);
    p->eps = 1.0f;   /* Simulation.hpp:59  quadtree(1.0f, 1.0f, 16) */
    p->dt = 0.01f;   /* main.cpp:39        SIMULATION_DT{0.01f} */
    p->precision = NB_FP32;
    p->rsqrt_mode = NB_RSQRT_EXACT;
    p->sum_order = NB_SUM_TILED;
    p->integrator = NB_INTEGRATOR_KICK_DRIFT;
    p->extras = 0;
    p->device = -1;
    p->dims = 2;
}

/* ---- mt19937 (Matsumoto & Nishimura 1998), same stream as std::mt19937(seed),
 * which is the PRNG the reference seeds its ICs with (Simulation.hpp:349) ---- */
typedef struct { uint32_t mt[624]; int idx; } nb_mt;

static void mt_seed(nb_mt *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static uint32_t mt_next(nb_mt *g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* portable uniform in (0,1): raw 32-bit output mapped (u + 0.5) / 2^32 in double */
static double mt_u01(nb_mt *g) { return ((double)mt_next(g) + 0.5) * (1.0 / 4294967296.0); }

/* 3-D Plummer model, Aarseth, Henon & Wielen (1974) sampling, a = M = G = 1,
 * truncated at r <= 20, projected on the (x,y) plane. */
static int plummer(nb_body *out, size_t n, uint32_t seed, int dims);

int nb_plummer_2d(nb_body *out, size_t n, uint32_t seed) { return plummer(out, n, seed, 2); }
int nb_plummer_3d(nb_body *out, size_t n, uint32_t seed) { return plummer(out, n, seed, 3); }

static int plummer(nb_body *out, size_t n, uint32_t seed, int dims)
{
    if (!out && n) return nb_fail(NB_EINVAL, "nb_plummer: out is NULL");
    nb_mt g;
    mt_seed(&g, seed);
    const double two_pi = 6.283185307179586476925286766559;
    const float mass = n ? (float)(1.0 / (double)n) : 0.0f;
    for (size_t i = 0; i < n; ++i) {
        double r;
        do {
            double x1 = mt_u01(&g);
            r = 1.0 / sqrt(pow(x1, -2.0 / 3.0) - 1.0);
        } while (!(r <= 20.0));
        double x2 = mt_u01(&g), x3 = mt_u01(&g);
        double z = (1.0 - 2.0 * x2) * r;
        double rho = sqrt(fmax(r * r - z * z, 0.0));
        double px = rho * cos(two_pi * x3), py = rho * sin(two_pi * x3);

        double q, gq;
        do { /* von Neumann rejection on g(q) = q^2 (1-q^2)^(7/2), max < 0.1 */
            q = mt_u01(&g);
            gq = 0.1 * mt_u01(&g);
        } while (gq > q * q * pow(1.0 - q * q, 3.5));
        double v = q * sqrt(2.0) * pow(1.0 + r * r, -0.25);
        double x6 = mt_u01(&g), x7 = mt_u01(&g);
        double w = (1.0 - 2.0 * x6) * v;
        double vr = sqrt(fmax(v * v - w * w, 0.0));
        double vx = vr * cos(two_pi * x7), vy = vr * sin(two_pi * x7);

        memset(&out[i], 0, sizeof(nb_body));
        out[i].pos.x = (float)px; out[i].pos.y = (float)py;
        out[i].vel.x = (float)vx; out[i].vel.y = (float)vy;
        if (dims == 3) { NB_Z(out[i].pos) = (float)z; NB_Z(out[i].vel) = (float)w; }
        out[i].mass = mass;
        out[i].radius = 0.0f;
    }
    return NB_OK;
}

/* ---- the reference's own initial conditions ----------------------------------
 * What `Simulation()` starts from (Simulation.hpp:58-65 -> uniform_disc(n), :347-603), restated:
 * a 1e9 central mass; n-1 bodies laid along a forward-Euler trace of the Lorenz system
 * (sigma 10, rho 28, beta 8/3, h = 0.01, from (0.1, 0, 0)) scaled by sqrt(n)*300.7/10 in x and y;
 * masses from three uniform ranges picked with probabilities 0.825 : 0.125 : 0.025, radius =
 * cbrt(mass); bodies sorted by distance from the centre; speeds set to sqrt(M(<r)/r) along the
 * "unit" tangent that Vec2::normalize leaves behind — which divides x by the length TWICE
 * (Vec2.hpp:226-235), so vx is ~1/r of what a circular orbit needs.  That is the workload the
 * reference runs, so it is reproduced as is.  Arithmetic is fp32, one rounding per operation
 * (built with -ffp-contract=off), random numbers are std::mt19937(0) through libstdc++'s
 * uniform_real_distribution<float>: float(u32) / 2^32, stepped down when it rounds to 1.
 * tests/golden/default_ics.json holds the sha256 of the reference's 25 000 bodies. */
static float disc_u01(nb_mt *g)
{
    const float r = (float)mt_next(g) / 4294967296.0f;
    return r >= 1.0f ? nextafterf(1.0f, 0.0f) : r;
}

/* Order of the reference's std::sort (Simulation.hpp:585-589; comparator: |pos|^2 ascending).  A few
 * bodies tie in fp32 |pos|^2 and std::sort is not stable, so the order of the ties is whatever the
 * algorithm of the reference's standard library leaves: libstdc++ (GCC 11.4, the toolchain the golden
 * vectors were made with) — introsort: median-of-three quicksort down to runs of 16 (heapsort past depth
 * 2*floor(log2 n), never reached here), then one insertion-sort pass.  Restated below on (key, index) pairs. */
typedef struct { float key; uint32_t idx; } disc_key;

static void disc_swap(disc_key *a, disc_key *b) { const disc_key t = *a; *a = *b; *b = t; }

static void disc_sift_down(disc_key *v, size_t start, size_t len)
{
    /* plain binary max-heap sift (only reached if quicksort degenerates; ties may then order differently) */
    for (size_t root = start, child; (child = 2 * root + 1) < len; root = child) {
        if (child + 1 < len && v[child].key < v[child + 1].key) ++child;
        if (!(v[root].key < v[child].key)) return;
        disc_swap(&v[root], &v[child]);
    }
}

static void disc_quick(disc_key *v, size_t first, size_t last, int depth)
{
    while (last - first > 16) {
        if (depth-- == 0) {
            disc_key *h = v + first;
            const size_t len = last - first;
            for (size_t i = len / 2; i-- > 0;) disc_sift_down(h, i, len);
            for (size_t end = len; end-- > 1;) { disc_swap(&h[0], &h[end]); disc_sift_down(h, 0, end); }
            return;
        }
        /* median of (first+1, middle, last-1) goes to `first` and is the pivot */
        disc_key *r = &v[first], *a = &v[first + 1], *b = &v[first + (last - first) / 2], *c = &v[last - 1];
        if (a->key < b->key) {
            if (b->key < c->key) disc_swap(r, b); else if (a->key < c->key) disc_swap(r, c); else disc_swap(r, a);
        } else if (a->key < c->key) disc_swap(r, a);
        else if (b->key < c->key) disc_swap(r, c);
        else disc_swap(r, b);
        size_t lo = first + 1, hi = last;
        for (;;) {
            while (v[lo].key < v[first].key) ++lo;
            --hi;
            while (v[first].key < v[hi].key) --hi;
            if (!(lo < hi)) break;
            disc_swap(&v[lo], &v[hi]);
            ++lo;
        }
        disc_quick(v, lo, last, depth);
        last = lo;
    }
}

static void disc_sort(disc_key *v, size_t n)
{
    if (n < 2) return;
    int lg = 0;
    for (size_t m = n; m > 1; m >>= 1) ++lg;
    disc_quick(v, 0, n, 2 * lg);
    const size_t guarded = n > 16 ? 16 : n;
    for (size_t i = 1; i < n; ++i) {
        const disc_key val = v[i];
        size_t j = i;
        if (i < guarded && val.key < v[0].key) { for (; j > 0; --j) v[j] = v[j - 1]; }
        else { for (; val.key < v[j - 1].key; --j) v[j] = v[j - 1]; }      /* a smaller-or-equal key is known to sit to the left */
        v[j] = val;
    }
}

int nb_default_ics(nb_body *out, size_t n)
{
    if (!out && n) return nb_fail(NB_EINVAL, "nb_default_ics: out is NULL");
    if (n == 0) return NB_OK;
    if (n > (1u << 24)) return nb_fail(NB_EINVAL, "nb_default_ics: n = %zu too large", n);
    static const float lo[3] = {0.00005f, 1.2f, 5.0f}, hi[3] = {0.8f, 2.5f, 50.0f}, weight[3] = {0.825f, 0.125f, 0.025f};
    float wsum = 0.0f, cum[3], run = 0.0f;
    for (int k = 0; k < 3; ++k) wsum += weight[k];
    for (int k = 0; k < 3; ++k) { const float w = weight[k] / wsum; run += w; cum[k] = run; }

    nb_mt g;
    mt_seed(&g, 0u);
    const float outer = sqrtf((float)n) * 300.7f, scale = outer / 10.0f;
    const float sigma = 10.0f, rho = 28.0f, beta = 8.0f / 3.0f, h = 0.01f;
    float lx = 0.1f, ly = 0.0f, lz = 0.0f;

    memset(out, 0, n * sizeof(nb_body));
    out[0].mass = 1e9f;
    out[0].radius = 200.0f;
    for (size_t i = 1; i < n; ++i) {
        const float dx = sigma * (ly - lx);
        const float t0 = rho - lz, t1 = lx * t0, dy = t1 - ly;
        const float t2 = lx * ly, t3 = beta * lz, dz = t2 - t3;
        const float sx = dx * h, sy = dy * h, sz = dz * h;
        lx += sx; ly += sy; lz += sz;
        nb_body *b = &out[i];
        b->pos.x = lx * scale;
        b->pos.y = ly * scale;
        float vx = -b->pos.y, vy = b->pos.x;
        const float x2 = vx * vx, y2 = vy * vy, len = sqrtf(x2 + y2);
        if (len > 0.0f) { vx /= len; vx /= len; vy /= len; }
        b->vel.x = vx; b->vel.y = vy;
        const float pick = disc_u01(&g);
        int k = 0;
        while (k < 3 && !(pick <= cum[k])) ++k;
        if (k == 3) k = 0;
        const float span = hi[k] - lo[k], mu = disc_u01(&g) * span;
        b->mass = mu + lo[k];
        b->radius = cbrtf(b->mass);
    }
    disc_key *order = (disc_key *)malloc(n * sizeof *order);
    nb_body *tmp = (nb_body *)malloc(n * sizeof *tmp);
    if (!order || !tmp) { free(order); free(tmp); return nb_fail(NB_ENOMEM, "nb_default_ics: out of memory"); }
    for (size_t i = 0; i < n; ++i) {
        const float x2 = out[i].pos.x * out[i].pos.x, y2 = out[i].pos.y * out[i].pos.y;
        order[i].key = x2 + y2;
        order[i].idx = (uint32_t)i;
    }
    disc_sort(order, n);
    memcpy(tmp, out, n * sizeof *tmp);
    for (size_t i = 0; i < n; ++i) out[i] = tmp[order[i].idx];
    free(order); free(tmp);
    float enclosed = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        nb_body *b = &out[i];
        enclosed += b->mass;
        if (b->pos.x == 0.0f && b->pos.y == 0.0f) continue;
        const float x2 = b->pos.x * b->pos.x, y2 = b->pos.y * b->pos.y, r = sqrtf(x2 + y2);
        const float v = sqrtf(enclosed / r);
        b->vel.x *= v; b->vel.y *= v;
    }
    return NB_OK;
}

/* ---- dump format ------------------------------------------------------------ */
typedef struct nb_file_header {
    char     magic[8];      /* "NBODYAMD" */
    uint32_t version;       /* 1 */
    uint32_t body_size;     /* 64 */
    uint64_t n;
    uint64_t frame;
    float    eps;
    float    dt;
    int32_t  precision;
    int32_t  rsqrt_mode;
    int32_t  dims;          /* 0 or 2: planar (the reference); 3: z kept in the first padding float */
    int32_t  sum_order;     /* the three fields below were zero padding in files written before they existed: */
    int32_t  integrator;    /* 0 = the defaults (tiled sum, kick-drift, no extras), so old files read the same */
    int32_t  extras;
} nb_file_header;

_Static_assert(sizeof(nb_file_header) == 64, "dump header is 64 bytes");

int nb_write_bodies(const char *path, const nb_body *bodies, size_t n, uint64_t frame,
                    const nb_params *params)
{
    if (!path || (!bodies && n)) return nb_fail(NB_EINVAL, "nb_write_bodies: NULL argument");
    FILE *f = fopen(path, "wb");
    if (!f) return nb_fail(NB_EIO, "nb_write_bodies: cannot open %s: %s", path, strerror(errno));
    nb_file_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "NBODYAMD", 8);
    h.version = 1;
    h.body_size = (uint32_t)sizeof(nb_body);
    h.n = (uint64_t)n;
    h.frame = frame;
    if (params) {
        h.eps = params->eps; h.dt = params->dt;
        h.precision = params->precision; h.rsqrt_mode = params->rsqrt_mode;
        h.dims = params->dims;
        h.sum_order = params->sum_order; h.integrator = params->integrator; h.extras = params->extras;
    }
    const int keep_z = h.dims == 3;
    int rc = NB_OK;
    if (fwrite(&h, sizeof h, 1, f) != 1) rc = NB_EIO;
    /* records are written with padding forced to zero, whatever the caller holds */
    enum { CHUNK = 4096 };
    nb_body *buf = (nb_body *)malloc(sizeof(nb_body) * CHUNK);
    if (!buf) { fclose(f); return nb_fail(NB_ENOMEM, "nb_write_bodies: out of memory"); }
    for (size_t base = 0; rc == NB_OK && base < n; base += CHUNK) {
        size_t c = n - base < CHUNK ? n - base : CHUNK;
        memset(buf, 0, sizeof(nb_body) * c);
        for (size_t k = 0; k < c; ++k) {
            const nb_body *b = &bodies[base + k];
            buf[k].pos.x = b->pos.x; buf[k].pos.y = b->pos.y;
            buf[k].vel.x = b->vel.x; buf[k].vel.y = b->vel.y;
            buf[k].acc.x = b->acc.x; buf[k].acc.y = b->acc.y;
            buf[k].mass = b->mass;   buf[k].radius = b->radius;
            if (keep_z) { NB_Z(buf[k].pos) = NB_Z(b->pos); NB_Z(buf[k].vel) = NB_Z(b->vel); NB_Z(buf[k].acc) = NB_Z(b->acc); }
        }
        if (fwrite(buf, sizeof(nb_body), c, f) != c) rc = NB_EIO;
    }
    free(buf);
    if (fclose(f) != 0) rc = NB_EIO;
    if (rc != NB_OK) nb_fail(rc, "nb_write_bodies: short write to %s", path);
    return rc;
}

/* The header is untrusted input: n must be a size the kernels can index and must match the file length
 * exactly (64-byte header + n 64-byte records), so that no caller sizes a buffer from a forged count. */
static int read_header(FILE *f, const char *path, nb_file_header *h)
{
    if (fread(h, sizeof *h, 1, f) != 1) return nb_fail(NB_EFORMAT, "%s: truncated header", path);
    if (memcmp(h->magic, "NBODYAMD", 8) != 0) return nb_fail(NB_EFORMAT, "%s: bad magic", path);
    if (h->version != 1 || h->body_size != sizeof(nb_body))
        return nb_fail(NB_EFORMAT, "%s: unsupported version %u / record size %u", path, h->version, h->body_size);
    if (h->n == 0 || h->n > 0x7fffff00u) return nb_fail(NB_EFORMAT, "%s: body count %llu out of range", path, (unsigned long long)h->n);
    if (fseek(f, 0, SEEK_END) != 0) return nb_fail(NB_EIO, "%s: cannot seek", path);
    const long long len = (long long)ftello(f);
    if (len < 0 || (unsigned long long)len != sizeof *h + h->n * sizeof(nb_body))
        return nb_fail(NB_EFORMAT, "%s: %lld bytes on disk, header promises %llu bodies (%llu bytes)", path, len,
                       (unsigned long long)h->n, (unsigned long long)(sizeof *h + h->n * sizeof(nb_body)));
    if (fseek(f, (long)sizeof *h, SEEK_SET) != 0) return nb_fail(NB_EIO, "%s: cannot seek", path);
    if ((h->precision != NB_FP32 && h->precision != NB_FP64) || (h->rsqrt_mode != NB_RSQRT_EXACT && h->rsqrt_mode != NB_RSQRT_QUAKE) ||
        (h->sum_order != NB_SUM_TILED && h->sum_order != NB_SUM_SEQUENTIAL) ||
        (h->integrator != NB_INTEGRATOR_KICK_DRIFT && h->integrator != NB_INTEGRATOR_KDK) ||
        (h->extras & ~(NB_EXTRA_VCLAMP | NB_EXTRA_BOUNDARY)) || !(h->eps >= 0.0f))
        return nb_fail(NB_EFORMAT, "%s: header holds parameters outside their enums", path);
    return NB_OK;
}

int nb_read_header(const char *path, size_t *n, uint64_t *frame, nb_params *params)
{
    if (!path) return nb_fail(NB_EINVAL, "nb_read_header: NULL path");
    FILE *f = fopen(path, "rb");
    if (!f) return nb_fail(NB_EIO, "nb_read_header: cannot open %s: %s", path, strerror(errno));
    nb_file_header h;
    int rc = read_header(f, path, &h);
    fclose(f);
    if (rc != NB_OK) return rc;
    if (n) *n = (size_t)h.n;
    if (frame) *frame = h.frame;
    if (params) {
        nb_params_default(params);
        params->eps = h.eps; params->dt = h.dt;
        params->precision = h.precision; params->rsqrt_mode = h.rsqrt_mode;
        params->dims = h.dims == 3 ? 3 : 2;
        params->sum_order = h.sum_order; params->integrator = h.integrator; params->extras = h.extras;
    }
    return NB_OK;
}

int nb_read_bodies(const char *path, nb_body *out, size_t n)
{
    if (!path || (!out && n)) return nb_fail(NB_EINVAL, "nb_read_bodies: NULL argument");
    FILE *f = fopen(path, "rb");
    if (!f) return nb_fail(NB_EIO, "nb_read_bodies: cannot open %s: %s", path, strerror(errno));
    nb_file_header h;
    int rc = read_header(f, path, &h);
    if (rc == NB_OK && (size_t)h.n != n)
        rc = nb_fail(NB_EINVAL, "%s: holds %llu bodies, caller asked for %zu", path, (unsigned long long)h.n, n);
    if (rc == NB_OK && fread(out, sizeof(nb_body), n, f) != n)
        rc = nb_fail(NB_EFORMAT, "%s: truncated body records", path);
    fclose(f);
    return rc;
}

/* ---- the RCCL id of a one-process-per-GPU launch through a file (include/nbody.h: nb_comm_id_publish / _await) ----
 * File = {"NBCOMMID", u64 nonce, NB_COMM_ID_BYTES id}.  The nonce names the LAUNCH (every rank is started with the
 * same one): a file left behind by a run that crashed before rank 0 could unlink it, or by a rank 0 that has not yet
 * replaced it, carries another nonce and is ignored by the waiting ranks instead of feeding ncclCommInitRank an id
 * nobody else holds (which blocks for ever inside RCCL). */
typedef struct nb_idfile { char magic[8]; uint64_t nonce; unsigned char id[NB_COMM_ID_BYTES]; } nb_idfile;

int nb_comm_id_publish(const char *path, uint64_t nonce, const void *id)
{
    if (!path || !id) return nb_fail(NB_EINVAL, "nb_comm_id_publish: NULL argument");
    char tmp[4096];
    if (snprintf(tmp, sizeof tmp, "%s.tmp.%ld", path, (long)getpid()) >= (int)sizeof tmp) return nb_fail(NB_EINVAL, "nb_comm_id_publish: path too long");
    (void)unlink(path);                       /* a stale file of an earlier launch must not be read while this one is being written */
    nb_idfile rec;
    memcpy(rec.magic, "NBCOMMID", 8);
    rec.nonce = nonce;
    memcpy(rec.id, id, NB_COMM_ID_BYTES);
    FILE *f = fopen(tmp, "wb");
    if (!f) return nb_fail(NB_EIO, "nb_comm_id_publish: cannot create %s: %s", tmp, strerror(errno));
    const int ok = fwrite(&rec, sizeof rec, 1, f) == 1;
    if (fclose(f) != 0 || !ok || rename(tmp, path) != 0) {
        (void)unlink(tmp);
        return nb_fail(NB_EIO, "nb_comm_id_publish: cannot publish %s: %s", path, strerror(errno));
    }
    return NB_OK;
}

int nb_comm_id_await(const char *path, uint64_t nonce, void *id_out, int timeout_ms)
{
    if (!path || !id_out || timeout_ms < 0) return nb_fail(NB_EINVAL, "nb_comm_id_await: bad argument");
    struct timespec t0, now;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int stale = 0;
    for (;;) {
        FILE *f = fopen(path, "rb");
        if (f) {
            nb_idfile rec;
            const int whole = fread(&rec, sizeof rec, 1, f) == 1;
            fclose(f);
            if (whole && memcmp(rec.magic, "NBCOMMID", 8) == 0) {
                if (rec.nonce == nonce) { memcpy(id_out, rec.id, NB_COMM_ID_BYTES); return NB_OK; }
                stale = 1;                    /* another launch's file: keep waiting for ours */
            }
        }
        clock_gettime(CLOCK_MONOTONIC, &now);
        const double ms = (double)(now.tv_sec - t0.tv_sec) * 1e3 + (double)(now.tv_nsec - t0.tv_nsec) * 1e-6;
        if (ms >= (double)timeout_ms)
            return nb_fail(NB_EIO, "nb_comm_id_await: no RCCL id of this launch (nonce %llu) in %s after %d ms%s", (unsigned long long)nonce, path,
                           timeout_ms, stale ? " (the file there belongs to another launch)" : "");
        const struct timespec nap = {0, 20000000L};
        nanosleep(&nap, NULL);
    }
}
