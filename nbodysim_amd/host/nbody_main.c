/* nbody_main.c — plain-C host driver over the C ABI (include/nbody.h).
 *
 * The reference's only host is a raylib GUI whose simulation thread does
 *     simulation->step();  SHARED_BODIES = simulation->bodies;     (main.cpp:621-627)
 * This is the headless equivalent: init -> step() loop -> dump, all in C, all
 * compute in libnbody_hip.so.
 *
 *   nbody_main [-n N] [-s steps] [-dt DT] [-eps EPS] [-seed S] [-fp64] [-quake]
 *              [-sequential] [-kdk] [-dump FILE] [-load FILE] [-sync-every K]
 *              [-reference-ics]  the reference's own start: Simulation()'s 25 000-body disc (or -n N of it),
 *                            eps = 1, dt = 0.01, velocity clamp + soft boundary (Simulation.hpp:58-65,116-163)
 *              [-shards P]   P sharded handles driven from this one process (device r mod #GPUs),
 *                            exchanged with nb_exchange_positions: multi-GPU without RCCL
 *              [-no-symmetry] one-sided kernels / all-gather protocol (nb_params.flags)
 *              [-mass-scaling | -mass-scaling-measured | -no-mass-scaling]  individual masses: fold them into the pair geometry wherever
 *                                                  representable / only if the library's upload-time measurement finds it harmless / never (default)
 *              [-allreduce]  with -shards: the replicated protocol (every handle integrates all n; in-process all-reduce)
 *              [-late-us US] sharded symmetric protocol: local work held back for the side stream (nb_params.sym_late_us)
 *              [-rccl]       with -shards P: exchange through RCCL instead (nb_comm_create_all + nb_comm_step: collectives on a
 *                            communication stream per handle, event-ordered, no host sync in the loop); needs one device
 *                            per shard.  -shards 1 -rccl runs the sharded protocol with ONE rank (NB_FLAG_SHARD_SINGLE):
 *                            the whole split-step + RCCL path on a one-GPU box
 *              [-rank R -world P -idfile F [-nonce X] [-deadline S] [-device D]]  ONE PROCESS PER GPU, plain C: this process is
 *                            rank R of P; rank 0 creates the RCCL id (nb_comm_unique_id) and publishes it through file F together
 *                            with the launch's nonce X (nb_comm_id_publish: a stale F is removed first, the new one appears
 *                            atomically); the others wait for a file carrying THEIR nonce (nb_comm_id_await), so a file left by
 *                            a crashed run is never mistaken for this launch's.  Give every rank of one launch the same X (default
 *                            0; e.g. $(date +%s%N)).  Forming the communicator is bounded by -deadline seconds (default 120): a
 *                            rank whose peers never arrive exits with status 3 instead of waiting inside RCCL for ever.  Every
 *                            rank builds the same initial bodies, owns block R, and the whole step loop is nb_comm_step.
 *                            Launch: X=$(date +%s%N); for r in 0..P-1: nbody_main -rank $r -world P -idfile F -nonce $X &
 *                            (-world 1 runs the sharded protocol with one rank: the single-GPU rehearsal)
 * -load FILE restarts from a dump: eps, dt, precision, rsqrt mode, sum order, integrator and extras come from its
 * header unless the command line gives them (options are applied in order, so put -load first to override).
 */
#include "nbody.h"

#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* deadline on communicator formation: ncclCommInitRank blocks until every rank has arrived, with no timeout of its own */
static const char *g_deadline_what = "";
static void on_deadline(int sig)
{
    (void)sig;
    static const char msg[] = "nbody_main: deadline expired while ";
    if (write(2, msg, sizeof msg - 1) < 0) _exit(3);
    if (write(2, g_deadline_what, strlen(g_deadline_what)) < 0 || write(2, "\n", 1) < 0) _exit(3);
    _exit(3);      /* never a re-exec: the process may have initialised the GPU */
}

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)
#define CHECK(call) do { int rc_ = (call); if (rc_ != NB_OK) DIE("%s -> %d: %s", #call, rc_, nb_last_error()); } while (0)

int main(int argc, char **argv)
{
    size_t n = 65536;
    uint64_t frame0 = 0;
    int steps = 20, sync_every = 0, shards = 1, reference_ics = 0, n_given = 0, rccl = 0, shards_given = 0;
    int rank = -1, world = 0, device = -1, deadline_s = 120;
    unsigned long long nonce = 0;
    const char *idfile = NULL;
    unsigned seed = 42;
    const char *dump = NULL, *load = NULL;
    nb_params p;
    nb_params_default(&p);
    p.eps = 0.01f;
    p.dt = 1e-3f;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-n") && i + 1 < argc) { n = (size_t)strtoull(argv[++i], NULL, 10); n_given = 1; }
        else if (!strcmp(argv[i], "-reference-ics")) {
            reference_ics = 1;
            p.eps = 1.0f; p.dt = 0.01f; p.extras = NB_EXTRA_VCLAMP | NB_EXTRA_BOUNDARY;
        }
        else if (!strcmp(argv[i], "-s") && i + 1 < argc) steps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-dt") && i + 1 < argc) p.dt = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "-eps") && i + 1 < argc) p.eps = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "-seed") && i + 1 < argc) seed = (unsigned)strtoul(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "-fp64")) p.precision = NB_FP64;
        else if (!strcmp(argv[i], "-quake")) p.rsqrt_mode = NB_RSQRT_QUAKE;
        else if (!strcmp(argv[i], "-sequential")) p.sum_order = NB_SUM_SEQUENTIAL;
        else if (!strcmp(argv[i], "-kdk")) p.integrator = NB_INTEGRATOR_KDK;
        else if (!strcmp(argv[i], "-dump") && i + 1 < argc) dump = argv[++i];
        else if (!strcmp(argv[i], "-load") && i + 1 < argc) {
            /* the header's parameters become the defaults of this run; later options override them */
            load = argv[++i];
            nb_params fp;
            CHECK(nb_read_header(load, &n, &frame0, &fp));     /* validates n against the file length */
            p.eps = fp.eps; p.dt = fp.dt; p.precision = fp.precision; p.rsqrt_mode = fp.rsqrt_mode;
            p.sum_order = fp.sum_order; p.integrator = fp.integrator; p.extras = fp.extras; p.dims = fp.dims;
            p.first_frame = frame0;                             /* the frame counter continues where the dump left off */
        }
        else if (!strcmp(argv[i], "-no-symmetry")) p.flags |= NB_FLAG_NO_SYMMETRY;
        else if (!strcmp(argv[i], "-mass-scaling")) p.flags |= NB_FLAG_MASS_SCALING;
        else if (!strcmp(argv[i], "-mass-scaling-measured")) p.flags |= NB_FLAG_MASS_SCALING_MEASURED;
        else if (!strcmp(argv[i], "-no-mass-scaling")) p.flags |= NB_FLAG_NO_MASS_SCALING;
        else if (!strcmp(argv[i], "-allreduce")) p.flags |= NB_FLAG_SHARD_ALLREDUCE;
        else if (!strcmp(argv[i], "-late-us") && i + 1 < argc) p.sym_late_us = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "-sync-every") && i + 1 < argc) sync_every = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-shards") && i + 1 < argc) { shards = atoi(argv[++i]); shards_given = 1; }
        else if (!strcmp(argv[i], "-rccl")) rccl = 1;
        else if (!strcmp(argv[i], "-rank") && i + 1 < argc) rank = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-world") && i + 1 < argc) world = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-idfile") && i + 1 < argc) idfile = argv[++i];
        else if (!strcmp(argv[i], "-nonce") && i + 1 < argc) nonce = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "-deadline") && i + 1 < argc) deadline_s = atoi(argv[++i]);
        else DIE("unknown argument %s", argv[i]);
    }

    nb_body *bodies;
    if (reference_ics && !n_given) n = 25000;
    /* page-locked host memory from the library: nb_sync DMAs straight into it (a malloc'ed array works too, staged) */
    if (load) {
        bodies = (nb_body *)nb_host_alloc(n * sizeof *bodies); /* n <= 0x7fffff00 and consistent with the file: checked by nb_read_header */
        if (!bodies) DIE("nb_host_alloc: %s", nb_last_error());
        CHECK(nb_read_bodies(load, bodies, n));
        printf("loaded %zu bodies (frame %llu, eps %g, dt %g, %s%s%s) from %s\n", n, (unsigned long long)frame0, p.eps, p.dt,
               p.precision == NB_FP64 ? "fp64" : "fp32", p.rsqrt_mode == NB_RSQRT_QUAKE ? ", quake" : "",
               p.sum_order == NB_SUM_SEQUENTIAL ? ", sequential" : "", load);
    } else {
        bodies = (nb_body *)nb_host_alloc(n * sizeof *bodies);
        if (!bodies) DIE("nb_host_alloc: %s", nb_last_error());
        if (reference_ics) CHECK(nb_default_ics(bodies, n));
        else CHECK(nb_plummer_2d(bodies, n, seed));
    }

    if (world > 0) {
        /* one process per GPU: this process is rank `rank` of `world`; the communicator is formed from an id file */
        /* blocks of ceil(n / P) particles, the last one shorter when P does not divide n (all-gather protocol: equal counts over
         * padded replicas, which the library allocates) */
        const size_t blk = (n + (size_t)(world > 0 ? world : 1) - 1) / (size_t)(world > 0 ? world : 1);
        if (rank < 0 || rank >= world || !idfile || (size_t)(world - 1) * blk >= n) DIE("-rank R -world P -idfile F: 0 <= R < P, every rank must own at least one particle");
        const int ndev = nb_device_count();
        nb_params q = p;
        q.device = device >= 0 ? device : (ndev > 0 ? rank % ndev : 0);
        q.shard_rank = rank; q.shard_world = world;
        q.i_begin = (uint64_t)rank * blk; q.i_count = n - (size_t)rank * blk < blk ? n - (size_t)rank * blk : blk;
        if (p.flags & NB_FLAG_SHARD_ALLREDUCE) { q.i_begin = 0; q.i_count = n; }
        if (world == 1) q.flags |= NB_FLAG_SHARD_SINGLE;
        nb_sim *h = nb_create(bodies, n, &q);
        if (!h) DIE("nb_create(rank %d): %s", rank, nb_last_error());
        char id[NB_COMM_ID_BYTES];
        if (deadline_s < 1) deadline_s = 1;
        CHECK(nb_comm_available(NULL));                    /* a rank without RCCL stops HERE, before anyone waits for it inside RCCL */
        if (rank == 0) {
            CHECK(nb_comm_unique_id(id));
            CHECK(nb_comm_id_publish(idfile, nonce, id));
        } else {
            CHECK(nb_comm_id_await(idfile, nonce, id, deadline_s * 1000));
        }
        g_deadline_what = "forming the RCCL communicator (ncclCommInitRank): a peer never arrived";
        signal(SIGALRM, on_deadline);
        alarm((unsigned)deadline_s);
        nb_comm *comm = nb_comm_create_rank(h, id, rank, world);
        alarm(0);
        if (!comm) DIE("nb_comm_create_rank(rank %d of %d): %s", rank, world, nb_last_error());
        int proto = 0, ver = 0;
        CHECK(nb_comm_info(comm, &proto, NULL, NULL, &ver));
        CHECK(nb_comm_step(comm, p.dt, 2));                /* warm-up: RCCL channels, first launches */
        CHECK(nb_comm_wait(comm));
        const double t0r = now_s();
        CHECK(nb_comm_step(comm, p.dt, steps));
        const double t_enq = now_s() - t0r;
        CHECK(nb_comm_wait(comm));
        const double pers = (now_s() - t0r) / steps;
        double pm[3], lz;
        CHECK(nb_momentum(h, pm, &lz));
        CHECK(nb_sync(h, bodies + (proto == NB_SHARD_ALLREDUCE ? 0 : (size_t)rank * blk)));
        printf("rank %d of %d on device %d: protocol=%s RCCL %d.%d.%d frame=%llu  %.3f ms/step (this rank's clock)  %.3e pair interactions/s  "
               "host enqueue %.1f us/step  block momentum (%.6e, %.6e)\n", rank, world, q.device,
               proto == NB_SHARD_SYMMETRIC ? "symmetric" : proto == NB_SHARD_ALLREDUCE ? "allreduce" : proto == NB_SHARD_ALLGATHER ? "allgather" : "none",
               ver / 10000, ver / 100 % 100, ver % 100, (unsigned long long)nb_frame(h), pers * 1e3, (double)n * (double)n / pers, t_enq / steps * 1e6, pm[0], pm[1]);
        if (dump) {                                        /* every rank dumps its own block (whole system for the replicated protocol) */
            char name[4096];
            snprintf(name, sizeof name, "%s.rank%d", dump, rank);
            const size_t first = proto == NB_SHARD_ALLREDUCE ? 0 : (size_t)rank * blk, cnt = proto == NB_SHARD_ALLREDUCE ? n : (size_t)q.i_count;
            CHECK(nb_write_bodies(name, bodies + first, cnt, nb_frame(h), &p));
            printf("dumped %zu bodies to %s\n", cnt, name);
        }
        nb_comm_destroy(comm);
        nb_destroy(h);
        if (rank == 0) unlink(idfile);
        CHECK(nb_host_free(bodies));
        return 0;
    }
    if (shards > 1 || (shards_given && rccl)) {
        /* one process, `shards` handles: each integrates a contiguous block */
        if (shards < 1 || shards > 64) DIE("-shards 1..64");
        nb_sim *h[64];
        const int ndev = nb_device_count();
        const size_t blk = (n + (size_t)shards - 1) / (size_t)shards;      /* the last block is shorter when shards does not divide n */
        if ((size_t)(shards - 1) * blk >= n) DIE("-shards %d: every shard must own at least one of the %zu particles", shards, n);
        if (shards == 1) p.flags |= NB_FLAG_SHARD_SINGLE;      /* one rank: every pair is local, the collectives are one-rank copies */
        for (int r = 0; r < shards; ++r) {
            nb_params q = p;
            q.i_begin = (uint64_t)r * blk; q.i_count = n - (size_t)r * blk < blk ? n - (size_t)r * blk : blk; q.device = ndev > 0 ? r % ndev : 0;
            if (p.flags & NB_FLAG_SHARD_ALLREDUCE) { q.i_begin = 0; q.i_count = n; }     /* replicated: every handle integrates all n */
            q.shard_rank = r; q.shard_world = shards;   /* lets the library pick the symmetric protocol where it applies */
            h[r] = nb_create(bodies, n, &q);
            if (!h[r]) DIE("nb_create(shard %d): %s", r, nb_last_error());
        }
        const int symmetric = nb_shard_protocol(h[0]) == NB_SHARD_SYMMETRIC;
        const int replicated = nb_shard_protocol(h[0]) == NB_SHARD_ALLREDUCE;
        char desc0[1024];
        CHECK(nb_describe(h[0], desc0, sizeof desc0));
        printf("shard 0: %s\n", desc0);
        nb_comm *comm = NULL;
        if (rccl) {
            comm = nb_comm_create_all(h, shards);
            if (!comm) DIE("nb_comm_create_all: %s", nb_last_error());
            int ver = 0;
            CHECK(nb_comm_info(comm, NULL, NULL, NULL, &ver));
            printf("exchange: RCCL %d.%d.%d, one communication stream per handle, event-ordered\n", ver / 10000, ver / 100 % 100, ver % 100);
        }
        const double t0s = now_s();
        if (comm) {
            CHECK(nb_comm_step(comm, p.dt, steps));       /* the whole loop is the library's: INTEGRATION.md 5 shows it */
        } else {
            for (int s = 0; s < steps; ++s) {             /* in-process exchange (peer copies / peer reads), also stream-ordered */
                for (int r = 0; r < shards; ++r) CHECK(nb_step_begin(h[r], p.dt));
                if (symmetric) {                          /* cross-block pairs, then the in-process reduce-scatter */
                    for (int r = 0; r < shards; ++r) CHECK(nb_step_mid(h[r]));
                    CHECK(nb_exchange_accelerations(h, shards));
                }
                if (replicated) CHECK(nb_exchange_allreduce(h, shards));
                for (int r = 0; r < shards; ++r) CHECK(nb_step_finish(h[r]));
                if (!replicated) CHECK(nb_exchange_positions(h, shards));
            }
        }
        const double t_enq = now_s() - t0s;               /* the host has only enqueued so far */
        if (comm) CHECK(nb_comm_wait(comm));
        for (int r = 0; r < shards; ++r) CHECK(nb_wait(h[r]));
        const double pers = (now_s() - t0s) / steps;
        if (replicated) CHECK(nb_sync(h[0], bodies));                 /* every handle holds the whole state */
        else for (int r = 0; r < shards; ++r) CHECK(nb_sync(h[r], bodies + (size_t)r * blk));
        printf("protocol=%s ", symmetric ? "symmetric" : replicated ? "allreduce" : "allgather");
        printf("shards=%d on %d device(s): frame=%llu  %.3f ms/step  %.3e pair interactions/s  host enqueue %.1f us/step\n", shards, ndev,
               (unsigned long long)nb_frame(h[0]), pers * 1e3, (double)n * (double)n / pers, t_enq / steps * 1e6);
        printf("body[0]: pos=(%.6f, %.6f) vel=(%.6f, %.6f)\n", bodies[0].pos.x, bodies[0].pos.y, bodies[0].vel.x, bodies[0].vel.y);
        if (dump) { CHECK(nb_write_bodies(dump, bodies, n, nb_frame(h[0]), &p)); printf("dumped to %s\n", dump); }
        if (comm) nb_comm_destroy(comm);
        for (int r = 0; r < shards; ++r) nb_destroy(h[r]);
        CHECK(nb_host_free(bodies));
        return 0;
    }
    nb_sim *sim = nb_create(bodies, n, &p);
    if (!sim) DIE("nb_create: %s", nb_last_error());
    char desc[1024];
    CHECK(nb_describe(sim, desc, sizeof desc));
    printf("%s\n", desc);

    double k0, u0, k1, u1;
    CHECK(nb_energy(sim, &k0, &u0));
    CHECK(nb_step(sim, p.dt, 2));           /* warm-up */
    CHECK(nb_wait(sim));
    const double t0 = now_s();
    if (sync_every > 0) {                   /* the reference's pattern: step, then refresh bodies */
        for (int s = 0; s < steps; ++s) {
            CHECK(nb_step(sim, p.dt, 1));
            if ((s + 1) % sync_every == 0) CHECK(nb_sync(sim, bodies));
        }
    } else {
        CHECK(nb_step(sim, p.dt, steps));
    }
    CHECK(nb_wait(sim));
    const double t1 = now_s();
    CHECK(nb_energy(sim, &k1, &u1));
    CHECK(nb_sync(sim, bodies));

    const double per = (t1 - t0) / steps;
    printf("frame=%llu  %.3f ms/step  %.2f steps/s  %.3e pair interactions/s  %.2f TFLOP/s (14 flop/pair)\n",
           (unsigned long long)nb_frame(sim), per * 1e3, 1.0 / per, (double)n * (double)n / per,
           14.0 * (double)n * (double)n / per / 1e12);
    printf("energy: E0=%.9e  E1=%.9e  drift=%.3e\n", k0 + u0, k1 + u1, (k1 + u1 - k0 - u0) / (k0 + u0));
    printf("body[0]: pos=(%.6f, %.6f) vel=(%.6f, %.6f) acc=(%.6f, %.6f)\n", bodies[0].pos.x, bodies[0].pos.y,
           bodies[0].vel.x, bodies[0].vel.y, bodies[0].acc.x, bodies[0].acc.y);
    if (dump) {
        CHECK(nb_dump(sim, dump));
        printf("dumped to %s\n", dump);
    }
    nb_destroy(sim);
    CHECK(nb_host_free(bodies));
    return 0;
}
