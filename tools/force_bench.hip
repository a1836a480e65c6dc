// tools/force_bench.hip — tuning / diagnosis harness for the force kernel (not part of the product).
//
// Includes the product kernel bodies from nbodysim_amd/csrc/nb_kernels.hip.h and
//   1. sweeps launch geometry (i per lane, j-slices, occupancy cap via dynamic LDS),
//      timing each with HIP events (interleaved rounds, median/min reported);
//   2. runs a stamped diagnostic wrapper that records, per workgroup, the XCC / SE / CU
//      it ran on and its start/end clocks, and prints the placement histogram.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Inbodysim_amd/csrc -Iinclude -o build/force_bench tools/force_bench.hip
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

struct Stamp { unsigned long long t0, t1; unsigned hw_id, xcc_id; };

template <int P, int UNROLL>
__global__ __launch_bounds__(BLOCK)
void force_diag(const float2 *pos, const float *mass, float2 *partial, uint32_t n, uint32_t js, uint32_t i_tiles, float eps2, Stamp *st)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    force_tiled_f32_body<P, RSQ_EXACT, false, UNROLL>(pos, mass, nullptr, partial, 0, n, 0, n, js, i_tiles, eps2);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        Stamp s;
        s.t0 = t0; s.t1 = t1;
        s.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        s.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        st[blockIdx.x] = s;
    }
}

struct Ctx { float2 *pos; float *mass; float2 *partial; uint32_t n; float eps2; hipStream_t stream; };

template <int P, int UNROLL, bool UMASS = false, int WS = 1>
static float time_once(const Ctx &c, uint32_t js, size_t dyn_lds)
{
    const uint32_t i_tiles = (c.n + BLOCK / WS * 2 * P - 1) / (BLOCK / WS * 2 * P);
    const uint32_t grid = grid_blocks(i_tiles, js);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, c.stream));
    force_tiled_f32<P, RSQ_EXACT, false, UNROLL, UMASS, WS><<<grid, BLOCK, dyn_lds, c.stream>>>(c.pos, c.mass, nullptr, c.partial, 0, c.n, 0, c.n, js, i_tiles, c.eps2, 1.0f / c.n, 0xffffffffu, 0u);
    CK(hipEventRecord(e1, c.stream));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms;
}

struct Cfg { int P, unroll; uint32_t js; size_t lds; bool wave = false; bool um = false; int ws = 1; };

static float run_cfg(const Ctx &c, const Cfg &f)
{
    if (f.ws == 4) {
        if (f.um && f.P == 4) return time_once<4, 8, true, 4>(c, f.js, f.lds);
        if (f.um && f.P == 2) return time_once<2, 8, true, 4>(c, f.js, f.lds);
        if (!f.um && f.P == 4) return time_once<4, 8, false, 4>(c, f.js, f.lds);
        if (!f.um && f.P == 2) return time_once<2, 8, false, 4>(c, f.js, f.lds);
        fprintf(stderr, "unsupported ws cfg\n"); exit(1);
    }
    if (f.um) {
        if (f.P == 4 && f.unroll == 8) return time_once<4, 8, true>(c, f.js, f.lds);
        if (f.P == 2 && f.unroll == 8) return time_once<2, 8, true>(c, f.js, f.lds);
        if (f.P == 4 && f.unroll == 4) return time_once<4, 4, true>(c, f.js, f.lds);
        if (f.P == 2 && f.unroll == 16) return time_once<2, 16, true>(c, f.js, f.lds);
        fprintf(stderr, "unsupported um cfg\n"); exit(1);
    }
    if (f.P == 1 && f.unroll == 8) return time_once<1, 8>(c, f.js, f.lds);
    if (f.P == 2 && f.unroll == 8) return time_once<2, 8>(c, f.js, f.lds);
    if (f.P == 1 && f.unroll == 16) return time_once<1, 16>(c, f.js, f.lds);
    if (f.P == 2 && f.unroll == 16) return time_once<2, 16>(c, f.js, f.lds);
    if (f.P == 2 && f.unroll == 4) return time_once<2, 4>(c, f.js, f.lds);
    if (f.P == 4 && f.unroll == 4) return time_once<4, 4>(c, f.js, f.lds);
    if (f.P == 4 && f.unroll == 8) return time_once<4, 8>(c, f.js, f.lds);
    fprintf(stderr, "unsupported cfg\n"); exit(1);
}

template <int P>
static void diag(const Ctx &c, uint32_t js, size_t dyn_lds)
{
    const uint32_t i_tiles = (c.n + BLOCK * 2 * P - 1) / (BLOCK * 2 * P);
    const uint32_t grid = grid_blocks(i_tiles, js);
    Stamp *d; CK(hipMalloc(&d, grid * sizeof(Stamp)));
    CK(hipMemset(d, 0, grid * sizeof(Stamp)));
    force_diag<P, 8><<<grid, BLOCK, dyn_lds, c.stream>>>(c.pos, c.mass, c.partial, c.n, js, i_tiles, c.eps2, d);
    CK(hipStreamSynchronize(c.stream));
    std::vector<Stamp> h(grid);
    CK(hipMemcpy(h.data(), d, grid * sizeof(Stamp), hipMemcpyDeviceToHost));
    CK(hipFree(d));
    unsigned long long tmin = ~0ull, tmax = 0;
    std::map<unsigned, int> per_cu;   // key = xcc<<8 | se<<4.. | cu
    std::map<unsigned, int> per_xcc;
    std::vector<double> dur;
    for (auto &s : h) {
        if (s.t1 == 0) continue;   // invalid tile (grid rounding)
        tmin = std::min(tmin, s.t0); tmax = std::max(tmax, s.t1);
        const unsigned cu = (s.hw_id >> 8) & 0xf, sh = (s.hw_id >> 12) & 1, se = (s.hw_id >> 13) & 0x7, xcc = s.xcc_id & 0xf;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        per_xcc[xcc]++;
        dur.push_back((double)(s.t1 - s.t0) / 100e6 * 1e3);
    }
    std::sort(dur.begin(), dur.end());
    std::map<int, int> hist;
    for (auto &kv : per_cu) hist[kv.second]++;
    printf("  diag P=%d js=%u lds=%zu grid=%u: span=%.3f ms  block dur min/med/max = %.3f/%.3f/%.3f ms  distinct CUs=%zu\n",
           P, js, dyn_lds, grid, (double)(tmax - tmin) / 100e6 * 1e3, dur.front(), dur[dur.size() / 2], dur.back(), per_cu.size());
    printf("    blocks-per-CU histogram:");
    for (auto &kv : hist) printf("  %d blocks x %d CUs;", kv.first, kv.second);
    printf("\n    blocks per XCC:");
    for (auto &kv : per_xcc) printf(" x%u=%d", kv.first, kv.second);
    // start-time spread: how many blocks start later than 10% into the span
    int late = 0;
    for (auto &s : h) if (s.t1 && (double)(s.t0 - tmin) > 0.1 * (double)(tmax - tmin)) late++;
    printf("\n    blocks starting >10%% into the span: %d\n", late);
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 262144;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d  N=%u\n", prop.gcnArchName, prop.multiProcessorCount, n);
    std::mt19937 rng(1);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<float2> hp(n); std::vector<float> hm(n, 1.0f / n);
    for (auto &p : hp) p = make_float2(g(rng), g(rng));
    Ctx c; c.n = n; c.eps2 = 1e-4f;
    CK(hipStreamCreate(&c.stream));
    CK(hipMalloc(&c.pos, n * sizeof(float2))); CK(hipMalloc(&c.mass, n * sizeof(float)));
    CK(hipMalloc(&c.partial, (size_t)64 * n * sizeof(float2)));
    CK(hipMemcpy(c.pos, hp.data(), n * sizeof(float2), hipMemcpyHostToDevice));
    CK(hipMemcpy(c.mass, hm.data(), n * sizeof(float), hipMemcpyHostToDevice));

    std::vector<Cfg> cfgs;
    for (bool um : {true, false}) {
        cfgs.push_back({4, 8, 32, 0, false, um, 1});
        cfgs.push_back({4, 8, 64, 0, false, um, 1});
        for (uint32_t js : {4u, 8u, 16u, 32u}) {
            cfgs.push_back({4, 8, js, 0, false, um, 4});
            cfgs.push_back({2, 8, js, 0, false, um, 4});
        }
    }
    const int rounds = 5;
    std::vector<std::vector<float>> ms(cfgs.size());
    for (size_t k = 0; k < cfgs.size(); ++k) run_cfg(c, cfgs[k]);   // warm every variant
    for (int r = 0; r < rounds; ++r)
        for (size_t k = 0; k < cfgs.size(); ++k) ms[k].push_back(run_cfg(c, cfgs[k]));
    const double pairs = (double)n * n;
    for (size_t k = 0; k < cfgs.size(); ++k) {
        std::sort(ms[k].begin(), ms[k].end());
        const double med = ms[k][rounds / 2], mn = ms[k][0];
        const uint32_t per = BLOCK / cfgs[k].ws * 2 * cfgs[k].P;
        const uint32_t i_tiles = (n + per - 1) / per;
        printf("WS=%d%s P=%d unroll=%2d js=%2u lds=%6zu grid=%5u : med %.3f ms  min %.3f ms  -> %.2f TFLOP/s (%.1f%% of 157.3)\n",
               cfgs[k].ws, cfgs[k].um ? " UM" : "   ", cfgs[k].P, cfgs[k].unroll, cfgs[k].js, cfgs[k].lds, grid_blocks(i_tiles, cfgs[k].js), med, mn,
               14.0 * pairs / (med * 1e-3) / 1e12, 14.0 * pairs / (med * 1e-3) / 1e12 / 157.3 * 100);
    }
    printf("--- placement diagnostics\n");
    diag<2>(c, 16, 0);
    return 0;
}
