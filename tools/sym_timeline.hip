// tools/sym_timeline.hip — where does a short launch of the symmetric kernel lose its time?  (not part of the product)
//
// Runs the PRODUCT kernel body (force_sym_f32_body from nbodysim_amd/csrc/nb_kernels.hip.h) over the PRODUCT plan
// (nb_plan.cpp) inside a wrapper that stamps, per workgroup, start / end in the 100 MHz real-time counter and in shader
// clocks, and the XCC / SE / CU it ran on.  Prints: launch wall time (HIP events), shader clock during the launch,
// the ideal time of the plan's VALU work at that clock, the dispatch ramp (first / last workgroup start), workgroup
// durations, the busiest and the idlest CU (sum of its workgroups' VALU work), and the drain (time from the median
// workgroup end to the last one).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Inbodysim_amd/csrc -Iinclude -o build/sym_timeline tools/sym_timeline.hip nbodysim_amd/csrc/nb_plan.cpp
//   build/sym_timeline [n=25000] [tile=512|2048] [L=0] [masses=1: 0 equal | 1 individual | 2 individual, mass-scaled body] [pairs=0] [reps=20] [shift=0] [stagger=0]
//   stagger = k > 0 (experiment): among the first `stagger` items of the list every other adjacent pair of symmetric items of the same
//   tile is MERGED into one item of twice the chunks, so that the workgroups that start together on a CU no longer end together
// stamps INSIDE a workgroup (the body calls NB_STAMP(1) once the stationary particles are loaded, NB_STAMP(2) after the sweep of
// the item's chunks): wave 0 waits for its outstanding loads and records the real-time counter
#include <hip/hip_runtime.h>
__device__ unsigned long long *g_inner = nullptr;          // [workgroups][2]
#define NB_STAMP(k) do { if (g_inner && threadIdx.x < 64u) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
                         if (threadIdx.x == 0u) g_inner[(size_t)blockIdx.x * 2u + (k) - 1u] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#include "nb_kernels.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

using namespace nbk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

struct Stamp { unsigned long long t0, t1, c0, c1; unsigned hw_id, xcc_id; };

template <int MM, bool PAIRS, bool WS>
__global__ __launch_bounds__(BLOCK, NB_SYM_WAVES)
void stamped(const float2 *pos, const float *mass, const float *sigma, const SymItem *items, float2 *slab_s, float2 *slab_r, uint32_t n, float eps2, float um, Stamp *st, uint32_t shift,
             uint32_t delay_ticks, uint32_t delay_div)
{
    if (blockIdx.x < shift) return;                     // `shift` idle workgroups in front: item i runs as workgroup i + shift (moves every item to another XCD)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    if (delay_ticks && blockIdx.x < 1024u) {            // experiment: de-phase the workgroups of the first resident wave (k = 0..3 by index / delay_div)
        const unsigned long long until = t0 + (unsigned long long)delay_ticks * ((blockIdx.x / delay_div) & 3u);
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
    }
    force_sym_f32_body<RSQ_EXACT, MM, PAIRS, WS, true>(pos, mass, sigma, items[blockIdx.x - shift], slab_s, slab_r, n, eps2, um);   // write-through slab stores, like the product
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        Stamp s;
        s.t0 = t0; s.t1 = t1; s.c0 = c0; s.c1 = c1;
        s.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        s.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        st[blockIdx.x - shift] = s;
    }
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 25000u;
    const uint32_t tile = argc > 2 ? (uint32_t)atoi(argv[2]) : 512u;
    const uint32_t L = argc > 3 ? (uint32_t)atoi(argv[3]) : 0u;
    const int mass_mode = argc > 4 ? atoi(argv[4]) : 1;          // 0 equal masses, 1 individual (12 + 2), 2 individual folded into the geometry (11 + 2)
    const bool general = mass_mode != 0;
    const bool pairs = argc > 5 ? atoi(argv[5]) != 0 : false;
    const int reps = argc > 6 ? atoi(argv[6]) : 20;
    const uint32_t shift = argc > 7 ? (uint32_t)atoi(argv[7]) : 0u;
    const int stagger = argc > 8 ? atoi(argv[8]) : 0;
    const uint32_t delay_ticks = argc > 9 ? (uint32_t)atoi(argv[9]) : 0u;      // experiment: first-wave workgroup k of a CU starts k x this many 10-ns ticks late
    const uint32_t delay_div = argc > 10 ? (uint32_t)atoi(argv[10]) : 256u;    // ... with k = (index / delay_div) mod 4            // > 0: desynchronise the workgroups that share a CU (see below)
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;

    SymTuning t; t.sb = tile; t.forced_L = L; t.even_chunks = pairs;
    SymPlan pl; build_sym_plan(n, (uint32_t)cus, 0, 1, t, pl);
    if (stagger > 0) {
        std::vector<SymItem> out;
        size_t merged = 0;
        for (size_t i = 0; i < pl.items.size(); ++i) {
            const SymItem &a = pl.items[i];
            if ((int)out.size() < stagger && i + 1 < pl.items.size() && ((out.size() / 2) & 1u) == 0u) {
                const SymItem &b = pl.items[i + 1];
                if (!a.diag && !b.diag && a.tile == b.tile && a.c0 + a.cnt == b.c0 && a.r_base == b.r_base) {
                    SymItem m = a; m.cnt = a.cnt + b.cnt; out.push_back(m); ++i; ++merged; continue;
                }
            }
            out.push_back(a);
        }
        fprintf(stderr, "stagger: %zu pairs merged, %zu -> %zu items\n", merged, pl.items.size(), out.size());
        pl.items.swap(out);
    }
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float2> hp(n); std::vector<float> hm(n);
    for (uint32_t i = 0; i < n; ++i) { hp[i] = make_float2(U(rng), U(rng)); hm[i] = general ? 0.5f + 0.5f * std::fabs(U(rng)) : 1.0f / n; }
    float2 *pos, *ss, *sr; float *mass, *sigma; SymItem *items; Stamp *st;
    const size_t rows = pl.rowbase[pl.tiles];
    CK(hipMalloc(&pos, n * sizeof(float2))); CK(hipMalloc(&mass, n * sizeof(float)));
    CK(hipMalloc(&ss, rows * tile * sizeof(float2))); CK(hipMalloc(&sr, (pl.slab_r_elems + 2) * sizeof(float2)));
    CK(hipMalloc(&items, pl.items.size() * sizeof(SymItem))); CK(hipMalloc(&st, pl.items.size() * sizeof(Stamp)));
    CK(hipMemcpy(pos, hp.data(), n * sizeof(float2), hipMemcpyHostToDevice));
    CK(hipMemcpy(mass, hm.data(), n * sizeof(float), hipMemcpyHostToDevice));
    std::vector<float> hsig(n);
    for (uint32_t i = 0; i < n; ++i) hsig[i] = 1.0f / std::sqrt(hm[i]);
    CK(hipMalloc(&sigma, n * sizeof(float)));
    CK(hipMemcpy(sigma, hsig.data(), n * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(items, pl.items.data(), pl.items.size() * sizeof(SymItem), hipMemcpyHostToDevice));
    const uint32_t grid = (uint32_t)pl.items.size();
    unsigned long long *inner_dev = nullptr;
    CK(hipMalloc(&inner_dev, (size_t)(grid + shift) * 2 * sizeof(unsigned long long)));
    CK(hipMemset(inner_dev, 0, (size_t)(grid + shift) * 2 * sizeof(unsigned long long)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_inner), &inner_dev, sizeof inner_dev));
    const float eps2 = 1e-4f, um = 1.0f / n;
    auto launch = [&]() {
#define GO(MMV, PR, WSV) stamped<MMV, PR, WSV><<<grid + shift, BLOCK>>>(pos, mass, sigma, items, ss, sr, n, eps2, um, st, shift, delay_ticks, delay_div)
        if (mass_mode == 2) { if (tile == SYM_SB_WS) GO(MM_SCALED, false, true); else GO(MM_SCALED, false, false); return; }
        if (tile == SYM_SB_WS) { if (general) { if (pairs) GO(MM_GENERAL, true, true); else GO(MM_GENERAL, false, true); }
                                 else         { if (pairs) GO(MM_UNIFORM, true, true); else GO(MM_UNIFORM, false, true); } }
        else                   { if (general) { if (pairs) GO(MM_GENERAL, true, false); else GO(MM_GENERAL, false, false); }
                                 else         { if (pairs) GO(MM_UNIFORM, true, false); else GO(MM_UNIFORM, false, false); } }
#undef GO
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 50; ++k) launch();              // warm the clocks
    CK(hipDeviceSynchronize());
    std::vector<float> wall;
    for (int k = 0; k < reps; ++k) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); wall.push_back(ms * 1e3f);
    }
    // back-to-back rate (what a step loop sees)
    CK(hipEventRecord(e0)); for (int k = 0; k < 200; ++k) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms200; CK(hipEventElapsedTime(&ms200, e0, e1));
    std::sort(wall.begin(), wall.end());
    std::vector<Stamp> hs(grid);
    CK(hipMemcpy(hs.data(), st, grid * sizeof(Stamp), hipMemcpyDeviceToHost));

    // VALU cycles of the plan: a symmetric body (2 stationary x 1 travelling, both directions) = 10 packed (12 with masses) x 4 + 2 x 8
    // cycles; a diagonal one 8 (9) x 4 + 16; a wave runs 4 bodies per rotation step, 64 steps per chunk
    const double cyc_sym = (mass_mode == 2 ? 11 : general ? 12 : 10) * 4 + 16, cyc_diag = (mass_mode == 2 ? 8 : general ? 9 : 8) * 4 + 16;
    double work = 0;                                        // SIMD cycles of the whole launch
    std::vector<double> item_cyc(grid);
    for (uint32_t i = 0; i < grid; ++i) {
        const SymItem &it = pl.items[i];
        const double per_chunk_wave = 64.0 * 4.0 * (it.diag ? cyc_diag : cyc_sym);
        // classic: 4 waves each sweep every chunk of the item; wave-split: the chunks are shared out
        item_cyc[i] = per_chunk_wave * it.cnt * (tile == SYM_SB_WS ? 1.0 : 4.0);
        work += item_cyc[i];
    }
    unsigned long long tmin = ~0ull, tmax = 0, last_start = 0;
    double clk_sum = 0; size_t clk_n = 0;
    std::vector<double> dur, ends, starts;
    std::map<unsigned, double> cu_work; std::map<unsigned, int> cu_items;
    for (uint32_t i = 0; i < grid; ++i) {
        const Stamp &s = hs[i];
        tmin = std::min(tmin, s.t0); tmax = std::max(tmax, s.t1); last_start = std::max(last_start, s.t0);
        if (s.t1 > s.t0 + 100) { clk_sum += (double)(s.c1 - s.c0) / (double)(s.t1 - s.t0) * 100.0; ++clk_n; }   // MHz
        const unsigned cu = (s.xcc_id & 0xf) << 16 | ((s.hw_id >> 13) & 0x7) << 8 | ((s.hw_id >> 8) & 0xf);      // xcc | se | cu
        cu_work[cu] += item_cyc[i]; cu_items[cu] += 1;
    }
    for (uint32_t i = 0; i < grid; ++i) { dur.push_back((hs[i].t1 - hs[i].t0) * 0.01); ends.push_back((hs[i].t1 - tmin) * 0.01); starts.push_back((hs[i].t0 - tmin) * 0.01); }
    std::sort(dur.begin(), dur.end()); std::sort(ends.begin(), ends.end()); std::sort(starts.begin(), starts.end());
    const double mhz = clk_n ? clk_sum / clk_n : 0.0;
    const double simds = 4.0 * cus;
    const double ideal_us = work / simds / (mhz > 0 ? mhz : 2400.0);
    double wmax = 0, wmin = 1e300; for (auto &kv : cu_work) { wmax = std::max(wmax, kv.second); wmin = std::min(wmin, kv.second); }
    int imax = 0, imin = 1 << 30; for (auto &kv : cu_items) { imax = std::max(imax, kv.second); imin = std::min(imin, kv.second); }
    printf("n=%u tile=%u L=%u masses=%d pairs=%d | items=%u rows=%zu | CUs=%d seen=%zu\n", n, tile, pl.L, mass_mode, (int)pairs, grid, rows, cus, cu_work.size());
    printf("  launch wall (events): min %.1f median %.1f us | back-to-back %.1f us per launch\n", wall.front(), wall[wall.size() / 2], ms200 * 1e3 / 200);
    printf("  shader clock during the workgroups: %.0f MHz | VALU work of the plan at that clock on %d SIMDs: %.1f us (%.1f at 2400 MHz)\n",
           mhz, (int)simds, ideal_us, work / simds / 2400.0);
    printf("  inside the kernel: first start -> last end %.1f us | last workgroup START at %.1f us (90%% started by %.1f)\n",
           (tmax - tmin) * 0.01, (last_start - tmin) * 0.01, starts[(size_t)(0.9 * grid)]);
    // dispatch order: does a workgroup ever start before one of LOWER index?  (running maximum of the start times in index order:
    // an inversion is a workgroup that starts more than `slack` before that maximum).  HIP promises no order; this is what is observed.
    {
        unsigned long long run_max = 0; size_t inv1 = 0, inv10 = 0; double worst = 0.0;
        for (uint32_t i = 0; i < grid; ++i) {
            if (hs[i].t0 + 100 < run_max) ++inv1;                  // > 1 us earlier than an earlier-indexed workgroup's start
            if (hs[i].t0 + 1000 < run_max) ++inv10;                // > 10 us
            if (run_max > hs[i].t0) worst = std::max(worst, (run_max - hs[i].t0) * 0.01);
            run_max = std::max(run_max, hs[i].t0);
        }
        printf("  dispatch order: %zu of %u workgroups started > 1 us before a lower-indexed one (%zu by > 10 us; largest inversion %.2f us)\n", inv1, grid, inv10, worst);
        // the same among the workgroups of ONE XCD (workgroups are dealt round-robin over the XCDs and each XCD walks its share on its own)
        std::map<unsigned, unsigned long long> xmax; size_t xinv1 = 0; double xworst = 0.0; size_t rr_ok = 0;
        const unsigned x0 = hs[0].xcc_id & 0xf;
        for (uint32_t i = 0; i < grid; ++i) {
            const unsigned x = hs[i].xcc_id & 0xf;
            if (((x + 8u - x0) & 7u) == (i & 7u)) ++rr_ok;
            unsigned long long &m = xmax[x];
            if (hs[i].t0 + 100 < m) ++xinv1;
            if (m > hs[i].t0) xworst = std::max(xworst, (m - hs[i].t0) * 0.01);
            m = std::max(m, hs[i].t0);
        }
        {   // per XCD: VALU work dealt to it and when its last workgroup ended
            std::map<unsigned, double> xw; std::map<unsigned, unsigned long long> xe;
            double wsum = 0;
            for (uint32_t i = 0; i < grid; ++i) { const unsigned x = hs[i].xcc_id & 0xf; xw[x] += item_cyc[i]; wsum += item_cyc[i]; xe[x] = std::max(xe[x], hs[i].t1); }
            printf("  workgroup 0 sits on XCD %u; per XCD (work / mean, last end in us):", hs[0].xcc_id & 0xf);
            for (auto &kv : xw) printf("  %u: %.3f %.1f", kv.first, kv.second / (wsum / xw.size()), (xe[kv.first] - tmin) * 0.01);
            printf("\n");
            // shader clock seen by the workgroups of each XCD (s_memtime ticks per 100 MHz tick, workgroups longer than 1 us)
            std::map<unsigned, double> xc; std::map<unsigned, size_t> xn;
            for (uint32_t i = 0; i < grid; ++i) {
                const Stamp &q = hs[i];
                if (q.t1 > q.t0 + 100) { xc[q.xcc_id & 0xf] += (double)(q.c1 - q.c0) / (double)(q.t1 - q.t0) * 100.0; ++xn[q.xcc_id & 0xf]; }
            }
            printf("  per XCD shader clock (MHz):");
            for (auto &kv : xc) printf("  %u: %.0f", kv.first, kv.second / (double)xn[kv.first]);
            printf("\n");
        }
        printf("  ... within one XCD: %zu workgroups started > 1 us before a lower-indexed one of the same XCD (largest inversion %.2f us); "
               "%zu of %u workgroups sit on XCD (index mod 8) + const\n", xinv1, xworst, rr_ok, grid);
    }
    printf("  workgroup duration: min %.1f median %.1f p90 %.1f max %.1f us\n", dur.front(), dur[grid / 2], dur[(size_t)(0.9 * grid)], dur.back());
    printf("  workgroup ends: 10%% %.1f median %.1f 90%% %.1f last %.1f us\n", ends[(size_t)(0.1 * grid)], ends[grid / 2], ends[(size_t)(0.9 * grid)], ends.back());
    printf("  per CU: items min %d max %d | VALU work max / mean = %.3f, min / mean = %.3f -> busiest CU alone needs %.1f us at that clock\n",
           imin, imax, wmax / (work / cu_work.size()), wmin / (work / cu_work.size()), wmax / 4.0 / (mhz > 0 ? mhz : 2400.0));
    {   // residency: how long does a CU hold 4 / 3 / 2 / 1 / 0 workgroups between the launch's first start and last end?  (mean over CUs)
        std::map<unsigned, std::vector<std::pair<double, int>>> ev;
        for (uint32_t i = 0; i < grid; ++i) {
            const Stamp &q = hs[i];
            const unsigned cu = (q.xcc_id & 0xf) << 16 | ((q.hw_id >> 13) & 0x7) << 8 | ((q.hw_id >> 8) & 0xf);
            ev[cu].push_back({(q.t0 - tmin) * 0.01, +1}); ev[cu].push_back({(q.t1 - tmin) * 0.01, -1});
        }
        double held[6] = {0, 0, 0, 0, 0, 0};
        const double end = (tmax - tmin) * 0.01;
        unsigned busiest = 0; double bw = -1; for (auto &kv : cu_work) if (kv.second > bw) { bw = kv.second; busiest = kv.first; }
        for (auto &kv : ev) {
            auto &v = kv.second; std::sort(v.begin(), v.end());
            double t = 0; int k = 0;
            for (auto &e : v) { held[k > 5 ? 5 : k] += e.first - t; t = e.first; k += e.second; }
            held[0] += end - t;
        }
        printf("  residency, mean over CUs (us with k workgroups resident):");
        for (int k = 0; k <= 5; ++k) printf("  %d: %.1f", k, held[k] / ev.size());
        printf("\n  busiest CU, its workgroups (start-end us):");
        std::vector<std::pair<std::pair<double, double>, double>> w;      // (start, end), shader clock in MHz while it ran
        for (uint32_t i = 0; i < grid; ++i) {
            const Stamp &q = hs[i];
            const unsigned cu = (q.xcc_id & 0xf) << 16 | ((q.hw_id >> 13) & 0x7) << 8 | ((q.hw_id >> 8) & 0xf);
            if (cu == busiest) w.push_back({{(q.t0 - tmin) * 0.01, (q.t1 - tmin) * 0.01}, (double)(q.c1 - q.c0) / (double)(q.t1 - q.t0 + 1) * 100.0});
        }
        std::sort(w.begin(), w.end());
        for (auto &x : w) printf(" %.1f-%.1f@%.0f", x.first.first, x.first.second, x.second);
        printf("\n  ... the indices of its first-round workgroups:");
        for (uint32_t i = 0; i < grid; ++i) {
            const Stamp &q = hs[i];
            const unsigned cu = (q.xcc_id & 0xf) << 16 | ((q.hw_id >> 13) & 0x7) << 8 | ((q.hw_id >> 8) & 0xf);
            if (cu == busiest && (q.t0 - tmin) * 0.01 < 1.0) printf(" %u", i);
        }
        printf("\n");
        // shader clock by start time: do the workgroups of the first round run slower (clock ramp after the gap between launches)?
        double c_first = 0, c_rest = 0; size_t n_first = 0, n_rest = 0;
        for (uint32_t i = 0; i < grid; ++i) {
            const Stamp &q = hs[i];
            if (q.t1 <= q.t0 + 100) continue;
            const double mhz_i = (double)(q.c1 - q.c0) / (double)(q.t1 - q.t0) * 100.0;
            if ((q.t0 - tmin) * 0.01 < 1.0) { c_first += mhz_i; ++n_first; } else { c_rest += mhz_i; ++n_rest; }
        }
        {   // inside a workgroup: start -> stationary particles loaded -> sweep done -> end, first round against the later ones
            std::vector<unsigned long long> hin((size_t)(grid + shift) * 2);
            CK(hipMemcpy(hin.data(), inner_dev, hin.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<double> pro[2], swp[2], epi[2];
            for (uint32_t i = 0; i < grid; ++i) {
                const Stamp &q = hs[i];
                const unsigned long long s1 = hin[(size_t)(i + shift) * 2], s2 = hin[(size_t)(i + shift) * 2 + 1];
                if (!s1 || !s2 || pl.items[i].diag) continue;
                const int late = (q.t0 - tmin) * 0.01 < 1.0 ? 0 : 1;
                pro[late].push_back((s1 - q.t0) * 0.01); swp[late].push_back((s2 - s1) * 0.01); epi[late].push_back((q.t1 - s2) * 0.01);
            }
            for (int k = 0; k < 2; ++k) {
                if (pro[k].empty()) continue;
                std::sort(pro[k].begin(), pro[k].end()); std::sort(swp[k].begin(), swp[k].end()); std::sort(epi[k].begin(), epi[k].end());
                auto med = [](const std::vector<double> &v) { return v[v.size() / 2]; };
                printf("  inside a workgroup, %s (%zu symmetric items): stationary loads %.2f us (p90 %.2f) | sweep %.2f us (p90 %.2f) | combine + store %.2f us (p90 %.2f)\n",
                       k == 0 ? "FIRST round" : "later rounds", pro[k].size(), med(pro[k]), pro[k][(size_t)(0.9 * pro[k].size())],
                       med(swp[k]), swp[k][(size_t)(0.9 * swp[k].size())], med(epi[k]), epi[k][(size_t)(0.9 * epi[k].size())]);
            }
        }
        printf("  shader clock of the workgroups that start in the first microsecond: %.0f MHz (%zu), of the later ones: %.0f MHz (%zu)\n",
               n_first ? c_first / n_first : 0.0, n_first, n_rest ? c_rest / n_rest : 0.0, n_rest);
    }
    return 0;
}
