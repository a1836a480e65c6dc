// nb_fuzz.cpp — driver of the CPU sanitizer build (`make -C nbodysim_amd/csrc asan`): the host-only code of
// the library (nb_plan.cpp: the symmetric planner; nb_host.c: initial conditions, the dump format) under
// AddressSanitizer + UndefinedBehaviorSanitizer.  No GPU, no HIP.  Not part of libnbody_hip.so.
//
//   nb_host_fuzz [cases] [seed] [tmpdir]
//
// 1. planner: random (n, world, cus, tuning); over all ranks every (tile, chunk) unit must be covered exactly
//    once, stationary rows dense and unique, travelling-slab ranges disjoint and inside the slab, coverage
//    lists consistent with the segments.
// 2. host I/O: write/read round trips, truncated and forged headers (a forged body count must be refused
//    before any buffer is sized from it), the generators at ragged sizes.
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "nb_plan.h"
#include "nb_sched.h"
#include "nbody.h"

using namespace nbk;

#define REQUIRE(cond, ...)                                                        \
    do {                                                                          \
        if (!(cond)) {                                                            \
            std::fprintf(stderr, "FAIL %s:%d: %s | ", __FILE__, __LINE__, #cond); \
            std::fprintf(stderr, __VA_ARGS__);                                    \
            std::fprintf(stderr, "\n");                                           \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

static void check_plan_case(uint32_t n, uint32_t world, uint32_t cus, const SymTuning &tune)
{
    const uint32_t sb = tune.sb;
    const uint32_t tiles = (n + sb - 1) / sb, chunks = (n + SYM_CH - 1) / SYM_CH, cpt = sb / SYM_CH;
    std::vector<uint8_t> cover((size_t)tiles * chunks, 0);
    uint64_t cross_seen = 0, cross_total = 0;
    sym_units(n, world, nullptr, &cross_total, sb);
    for (uint32_t rank = 0; rank < world; ++rank) {
        SymPlan pl;
        build_sym_plan(n, cus, rank, world, tune, pl);
        REQUIRE(pl.tiles == tiles && pl.sb == sb && pl.items.size() == (size_t)pl.n_local + pl.n_cross + pl.n_late, "n=%u world=%u", n, world);
        REQUIRE(pl.slab_r_elems <= sym_slab_r_bound(n, world, sb), "bound n=%u world=%u: %llu > %llu", n, world,
                (unsigned long long)pl.slab_r_elems, (unsigned long long)sym_slab_r_bound(n, world, sb));
        if (!tune.forced_L) REQUIRE(pl.L % tune.quantum() == 0, "L = %u is not a multiple of the kernel's chunk quantum %u", pl.L, tune.quantum());
        std::vector<uint8_t> row_used(pl.rowbase[tiles], 0);
        std::vector<std::pair<int64_t, int64_t>> ranges;
        for (const SymItem &it : pl.items) {
            REQUIRE(it.tile < tiles && it.cnt >= 1 && it.c0 + it.cnt <= chunks, "item out of range n=%u", n);
            REQUIRE(it.s_row < row_used.size() && !row_used[it.s_row], "stationary row reused n=%u", n);
            row_used[it.s_row] = 1;
            const uint32_t lo_row = it.group == 2 ? pl.rowmid[it.tile] : pl.rowbase[it.tile];
            const uint32_t hi_row = it.group == 2 ? pl.rowbase[it.tile + 1] : pl.rowmid[it.tile];
            REQUIRE(it.s_row >= lo_row && it.s_row < hi_row, "row outside its tile's range");
            if (it.diag) REQUIRE(it.c0 >= it.tile * cpt && it.c0 + it.cnt <= std::min((it.tile + 1) * cpt, chunks), "diag chunks");
            else {
                REQUIRE(it.c0 >= (it.tile + 1) * cpt, "symmetric item meets its own tile");
                const int64_t lo = it.r_base + (int64_t)it.c0 * SYM_CH;
                const int64_t hi = it.r_base + (int64_t)std::min<uint64_t>((uint64_t)(it.c0 + it.cnt) * SYM_CH, n);
                REQUIRE(lo >= 0 && hi <= (int64_t)pl.slab_r_elems && lo < hi, "travelling range outside the slab");
                ranges.push_back({lo, hi});
            }
            if (it.group == 1) cross_seen += it.cnt;
            for (uint32_t c = it.c0; c < it.c0 + it.cnt; ++c) {
                REQUIRE(cover[(size_t)it.tile * chunks + c] == 0, "unit (%u,%u) covered twice, n=%u world=%u", it.tile, c, n, world);
                cover[(size_t)it.tile * chunks + c] = 1;
            }
        }
        std::sort(ranges.begin(), ranges.end());
        int64_t end = 0;
        // back to back, except for the one padding element that follows an odd-length segment (even segment offsets)
        for (auto &r : ranges) { REQUIRE(r.first == end || r.first == end + 1, "travelling ranges overlap or leave a hole"); end = r.second; }
        REQUIRE((int64_t)pl.slab_r_elems - end >= 0 && (int64_t)pl.slab_r_elems - end <= 1, "travelling slab not filled exactly");
        for (const SymSeg &sg : pl.segs) REQUIRE(sg.off % 2 == 0, "odd segment offset");
        // coverage lists: every (segment, particle) of the segments appears in its tile's list exactly once
        auto check_cov = [&](size_t s0, size_t s1, const std::vector<uint32_t> &begin, const std::vector<SymCov> &cov) {
            REQUIRE(begin.size() == (size_t)tiles + 1 && begin[tiles] == cov.size(), "coverage CSR shape");
            uint64_t want = 0, got = 0;
            for (size_t s = s0; s < s1; ++s) want += pl.segs[s].hi - pl.segs[s].lo;
            for (uint32_t t = 0; t < tiles; ++t)
                for (uint32_t i = begin[t]; i < begin[t + 1]; ++i) {
                    const uint32_t a = std::max(cov[i].lo, t * sb), b = std::min<uint64_t>(cov[i].hi, (uint64_t)(t + 1) * sb);
                    REQUIRE(a < b, "coverage entry does not meet its tile");
                    REQUIRE(cov[i].base + (int64_t)a >= 0 && cov[i].base + (int64_t)b <= (int64_t)pl.slab_r_elems, "coverage outside the slab");
                    got += b - a;
                }
            REQUIRE(want == got, "coverage lists miss particles: %llu vs %llu", (unsigned long long)want, (unsigned long long)got);
        };
        check_cov(0, pl.nsegs_main, pl.cov_main_begin, pl.cov_main);
        check_cov(pl.nsegs_main, pl.segs.size(), pl.cov_late_begin, pl.cov_late);
    }
    for (uint32_t t = 0; t < tiles; ++t)
        for (uint32_t c = 0; c < chunks; ++c)
            REQUIRE(cover[(size_t)t * chunks + c] == (c >= t * cpt ? 1 : 0), "unit (%u,%u) cover %d, n=%u world=%u", t, c,
                    cover[(size_t)t * chunks + c], n, world);
    REQUIRE(cross_seen == cross_total, "cross units %llu != %llu", (unsigned long long)cross_seen, (unsigned long long)cross_total);
}

static void fuzz_planner(int cases, std::mt19937 &rng)
{
    for (int k = 0; k < cases; ++k) {
        const uint32_t world = 1 + rng() % 8;
        uint32_t n;
        if (world == 1) n = 1 + rng() % 200000;                                   // ragged sizes, down to one particle
        else n = world * SYM_SB * (2 + rng() % 12);                              // equal blocks of whole tiles (>= 2 each)
        const uint32_t cus = 1 + rng() % 304;
        SymTuning t;
        if (rng() % 3 == 0) t.forced_L = 1 + rng() % 70;
        if (world > 1 && rng() % 2) t.late_units = rng() % 3000;
        t.late_chunks = 1 + rng() % 3;
        t.guided_tail = rng() % 4 != 0;
        t.even_chunks = rng() % 2 != 0;           // handles that sweep chunk pairs
        if (rng() % 2) t.sb = SYM_SB_WS;          // wave-split tiles (the product uses them for single handles; the planner takes any world)
        if (rng() % 5 == 0) t.wg_per_cu = 8 + rng() % 32;
        if (rng() % 4 == 0) { t.tail_at[0] = 0.3; t.tail_at[1] = 0.6; t.tail_at[2] = 0.9; }
        check_plan_case(n, world, cus, t);
    }
    check_plan_case(262144, 8, 256, SymTuning{});                                 // the benchmark's split
    { SymTuning e; e.even_chunks = true; e.late_units = 1400; check_plan_case(262144, 8, 256, e); check_plan_case(262144, 1, 256, e); check_plan_case(70001, 1, 256, e); }
    { SymTuning w; w.sb = SYM_SB_WS; check_plan_case(25000, 1, 256, w); check_plan_case(65536, 1, 256, w); check_plan_case(1, 1, 256, w); check_plan_case(513, 1, 2, w);
      w.even_chunks = true; check_plan_case(65536, 1, 256, w); check_plan_case(70001, 1, 256, w); check_plan_case(262144, 8, 256, w); }
    check_plan_case(1, 1, 256, SymTuning{});
    check_plan_case(2048, 1, 256, SymTuning{});
    check_plan_case(2049, 1, 1, SymTuning{});
}

static void write_raw(const std::string &path, const void *data, size_t bytes)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    REQUIRE(f, "cannot create %s", path.c_str());
    REQUIRE(std::fwrite(data, 1, bytes, f) == bytes, "short write");
    std::fclose(f);
}

static void fuzz_host(const std::string &dir, std::mt19937 &rng)
{
    const std::string path = dir + "/fuzz.nbd";
    for (size_t n : {(size_t)1, (size_t)2, (size_t)4097, (size_t)10000}) {
        std::vector<nb_body> b(n), back(n);
        REQUIRE(nb_plummer_2d(b.data(), n, 7) == NB_OK && nb_plummer_3d(back.data(), n, 7) == NB_OK, "plummer");
        nb_params p;
        nb_params_default(&p);
        p.eps = 0.25f; p.extras = NB_EXTRA_VCLAMP; p.integrator = NB_INTEGRATOR_KDK; p.sum_order = NB_SUM_SEQUENTIAL;
        REQUIRE(nb_write_bodies(path.c_str(), b.data(), n, 17, &p) == NB_OK, "%s", nb_last_error());
        size_t n2 = 0; uint64_t frame = 0; nb_params q;
        REQUIRE(nb_read_header(path.c_str(), &n2, &frame, &q) == NB_OK && n2 == n && frame == 17, "%s", nb_last_error());
        REQUIRE(q.eps == 0.25f && q.extras == NB_EXTRA_VCLAMP && q.integrator == NB_INTEGRATOR_KDK && q.sum_order == NB_SUM_SEQUENTIAL,
                "header does not carry the parameters");
        REQUIRE(nb_read_bodies(path.c_str(), back.data(), n) == NB_OK, "%s", nb_last_error());
        REQUIRE(std::memcmp(b.data(), back.data(), n * sizeof(nb_body)) == 0, "round trip differs");
        REQUIRE(nb_read_bodies(path.c_str(), back.data(), n + 1) == NB_EINVAL && nb_last_error_code() == NB_EINVAL, "wrong count accepted");
    }
    // forged / damaged files: refused with NB_EFORMAT before anything is allocated or read from them
    std::vector<unsigned char> img(64 + 3 * 64, 0);
    std::memcpy(img.data(), "NBODYAMD", 8);
    const uint32_t one = 1, sz = 64;
    std::memcpy(&img[8], &one, 4); std::memcpy(&img[12], &sz, 4);
    auto set_n = [&](uint64_t n) { std::memcpy(&img[16], &n, 8); };
    size_t n2 = 0;
    set_n(3); write_raw(path, img.data(), img.size());
    REQUIRE(nb_read_header(path.c_str(), &n2, nullptr, nullptr) == NB_OK && n2 == 3, "%s", nb_last_error());
    for (uint64_t forged : {(uint64_t)0, (uint64_t)4, (uint64_t)1 << 58, ~(uint64_t)0, (uint64_t)0x7fffff01u}) {
        set_n(forged); write_raw(path, img.data(), img.size());
        REQUIRE(nb_read_header(path.c_str(), &n2, nullptr, nullptr) == NB_EFORMAT && nb_last_error_code() == NB_EFORMAT, "forged n=%llu accepted",
                (unsigned long long)forged);
        std::vector<nb_body> small(3);
        REQUIRE(nb_read_bodies(path.c_str(), small.data(), (size_t)forged) != NB_OK, "forged n read");
    }
    set_n(3);
    for (size_t cut : {(size_t)0, (size_t)10, (size_t)63, (size_t)64, (size_t)100, img.size() - 1}) {
        write_raw(path, img.data(), cut);
        REQUIRE(nb_read_header(path.c_str(), &n2, nullptr, nullptr) == NB_EFORMAT, "truncated file (%zu bytes) accepted", cut);
    }
    for (int k = 0; k < 200; ++k) {                                               // random byte damage in the header
        std::vector<unsigned char> d(img);
        d[rng() % 64] ^= (unsigned char)(1u << (rng() % 8));
        write_raw(path, d.data(), d.size());
        nb_params q;
        const int rc = nb_read_header(path.c_str(), &n2, nullptr, &q);
        REQUIRE(rc == NB_OK || rc == NB_EFORMAT, "rc=%d", rc);
        if (rc == NB_OK) REQUIRE(n2 == 3, "damaged header changed n to %zu and was accepted", n2);
    }
    REQUIRE(nb_read_header((dir + "/does-not-exist").c_str(), &n2, nullptr, nullptr) == NB_EIO, "missing file");
    // the reference's initial conditions at ragged sizes (sort, enclosed-mass pass)
    for (size_t n : {(size_t)1, (size_t)2, (size_t)17, (size_t)1000, (size_t)25000}) {
        std::vector<nb_body> b(n);
        REQUIRE(nb_default_ics(b.data(), n) == NB_OK, "%s", nb_last_error());
        REQUIRE(b[0].mass == 1e9f, "central body");
        for (size_t i = 1; i < n; ++i)
            REQUIRE(b[i].pos.x * b[i].pos.x + b[i].pos.y * b[i].pos.y >= b[i - 1].pos.x * b[i - 1].pos.x + b[i - 1].pos.y * b[i - 1].pos.y, "not sorted at %zu", i);
    }
    REQUIRE(nb_plummer_2d(nullptr, 4, 1) == NB_EINVAL && nb_abi_version() == NB_ABI_VERSION, "argument checks");
    ::unlink(path.c_str());
}

// The sharded step's schedule (nb_sched.cpp): for random protocols / handle counts, every WAIT follows a RECORD of its
// event on the same handle (this step or, for the all-gather event, the previous one), collectives sit on the
// communication stream with the right count, and group brackets pair up.
static void fuzz_schedule(int cases, std::mt19937 &rng)
{
    using namespace nbk;
    for (int c = 0; c < cases; ++c) {
        const int proto = (int)(rng() % 4), handles = 1 + (int)(rng() % 8);
        const uint64_t block = 1 + rng() % 100000, full = block * (uint64_t)(1 + rng() % 8);
        for (int pending = 0; pending < 2; ++pending) {
            std::vector<nb_comm_op> ops;
            build_comm_schedule(proto, handles, block, full, pending != 0, ops);
            std::vector<std::vector<bool>> recorded((size_t)handles, std::vector<bool>(EV_COUNT, false));
            if (pending) for (auto &r : recorded) r[EV_AG] = true;      // recorded by the previous step
            int depth = 0;
            for (const nb_comm_op &o : ops) {
                if (o.kind == OP_GROUP_START) { REQUIRE(depth == 0 && handles > 1, "group start (protocol %d, %d handles)", proto, handles); ++depth; continue; }
                if (o.kind == OP_GROUP_END) { --depth; REQUIRE(depth == 0, "group end (protocol %d, %d handles)", proto, handles); continue; }
                REQUIRE(o.handle >= 0 && o.handle < handles, "handle index (protocol %d, %d handles)", proto, handles);
                if (o.kind == OP_RECORD) { REQUIRE(o.event >= 0 && o.event < EV_COUNT, "event index (protocol %d, %d handles)", proto, handles); recorded[(size_t)o.handle][(size_t)o.event] = true; }
                if (o.kind == OP_WAIT) REQUIRE(o.event >= 0 && o.event < EV_COUNT && recorded[(size_t)o.handle][(size_t)o.event], "wait before record (protocol %d, %d handles)", proto, handles);
                if (o.kind == OP_ALLGATHER || o.kind == OP_REDUCE_SCATTER) REQUIRE(o.stream == ST_COMM && o.count == block, "block collective (protocol %d, %d handles)", proto, handles);
                if (o.kind == OP_ALLREDUCE) REQUIRE(o.stream == ST_COMM && o.count == full, "all-reduce (protocol %d, %d handles)", proto, handles);
                if (o.kind <= OP_FINISH) REQUIRE(o.stream == ST_COMPUTE && depth == 0, "compute op (protocol %d, %d handles)", proto, handles);
            }
            REQUIRE(depth == 0, "unbalanced group (protocol %d, %d handles)", proto, handles);
        }
    }
}

// The RCCL id file of a one-process-per-GPU launch (nb_comm_id_publish / _await, nb_host.c): round trip, a stale file of
// another launch, truncated and damaged files, over-long paths — under the sanitizers.
static void fuzz_idfile(const std::string &dir, std::mt19937 &rng)
{
    const std::string path = dir + "/nb_fuzz_" + std::to_string((unsigned long)getpid()) + ".id";
    unsigned char id[NB_COMM_ID_BYTES], got[NB_COMM_ID_BYTES];
    for (int round = 0; round < 20; ++round) {
        for (unsigned char &b : id) b = (unsigned char)rng();
        const uint64_t nonce = ((uint64_t)rng() << 32) | rng();
        REQUIRE(nb_comm_id_publish(path.c_str(), nonce, id) == NB_OK, "publish: %s", nb_last_error());
        REQUIRE(nb_comm_id_await(path.c_str(), nonce, got, 200) == NB_OK && memcmp(id, got, sizeof id) == 0, "await: %s", nb_last_error());
        REQUIRE(nb_comm_id_await(path.c_str(), nonce + 1, got, 30) == NB_EIO, "a file of another launch was taken for this one");
        // damage: truncate to a random length / flip a byte of the magic: never an id
        std::vector<unsigned char> raw(8 + 8 + NB_COMM_ID_BYTES);
        FILE *f = std::fopen(path.c_str(), "rb");
        REQUIRE(f && std::fread(raw.data(), 1, raw.size(), f) == raw.size(), "id file has the documented size");
        std::fclose(f);
        const size_t cut = rng() % raw.size();
        write_raw(path, raw.data(), cut);
        REQUIRE(nb_comm_id_await(path.c_str(), nonce, got, 30) == NB_EIO, "truncated id file accepted (%zu bytes)", cut);
        raw[rng() % 8] ^= 0x40;
        write_raw(path, raw.data(), raw.size());
        REQUIRE(nb_comm_id_await(path.c_str(), nonce, got, 30) == NB_EIO, "id file with a damaged magic accepted");
    }
    unlink(path.c_str());
    const std::string longpath(5000, 'x');
    REQUIRE(nb_comm_id_publish((dir + "/" + longpath).c_str(), 1, id) != NB_OK, "over-long path accepted");
    REQUIRE(nb_comm_id_publish(nullptr, 1, id) == NB_EINVAL && nb_comm_id_await(path.c_str(), 1, nullptr, 10) == NB_EINVAL, "NULL arguments");
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? std::atoi(argv[1]) : 300;
    const unsigned seed = argc > 2 ? (unsigned)std::strtoul(argv[2], nullptr, 10) : 12345u;
    const std::string dir = argc > 3 ? argv[3] : "/tmp";
    std::mt19937 rng(seed);
    fuzz_planner(cases, rng);
    fuzz_host(dir, rng);
    fuzz_schedule(cases, rng);
    fuzz_idfile(dir, rng);
    std::printf("OK planner_cases=%d seed=%u\n", cases, seed);
    return 0;
}
