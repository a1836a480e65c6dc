// tools/ubench_valu.hip — VALU issue-rate microbenchmark for gfx950.
//
// Answers the questions the force-kernel design depends on:
//   * cycles per wave64 v_fma_f32 vs v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
//   * cycles per v_rsq_f32 (transcendental rate)
//   * the pair-body mix (9 packed + 2 rsq per two pairs)
// at 1, 2, 4 and 8 waves per SIMD with every CU busy.  Cycles are shader clocks
// (s_memtime); the effective clock is printed from s_memtime/s_memrealtime.
//
//   hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip && ./ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

enum { K_FMA = 0, K_PKFMA, K_PKMUL, K_PKADD, K_RSQ, K_BODY, K_BODY_SCALAR, K_COUNT };
static const char *kname[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
                              "v_rsq_f32", "pair-body(9pk+2rsq per 2 pairs)", "pair-body scalar(9+1rsq per pair)"};
// VALU instructions per inner iteration of each kernel
static const int kinsts[] = {16, 16, 16, 16, 16, 4 * 11, 4 * 10};

template <int KIND>
__global__ __launch_bounds__(256) void ubench(float *out, unsigned long long *cyc, unsigned long long *rt, int iters, float seed)
{
    float l = (float)threadIdx.x * 1e-3f + seed;
    v2f a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (v2f){l + k, l - k};
    v2f b = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == K_FMA) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k].x) : "v"(b.x), "v"(c.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k].y) : "v"(b.y), "v"(c.y));
            }
        } else if constexpr (KIND == K_PKFMA) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        } else if constexpr (KIND == K_PKMUL) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        } else if constexpr (KIND == K_PKADD) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        } else if constexpr (KIND == K_RSQ) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k].x));
                asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k].y));
            }
        } else if constexpr (KIND == K_BODY) {
            // 4 independent two-pair bodies; a[0..3] = ax accumulators, a[4..7] = ay.
            // hipcc lowers this to 2 v_pk_add + 4 v_pk_fma + 3 v_pk_mul + 2 v_rsq (checked in the .s).
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v2f xi = a[(k + 1) & 3], yi = a[(k + 2) & 3];
                v2f dx = (v2f){b.x, b.x} - xi;
                v2f dy = (v2f){b.y, b.y} - yi;
                v2f r2 = __builtin_elementwise_fma(dx, dx, c);
                r2 = __builtin_elementwise_fma(dy, dy, r2);
                v2f inv = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                v2f inv2 = inv * inv;
                v2f s = ((v2f){c.x, c.x} * inv) * inv2;
                a[k] = __builtin_elementwise_fma(s, dx, a[k]);
                a[4 + k] = __builtin_elementwise_fma(s, dy, a[4 + k]);
                asm volatile("" : "+v"(a[k]), "+v"(a[4 + k]));
            }
        } else if constexpr (KIND == K_BODY_SCALAR) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float dx = b.x - a[(k + 1) & 3].y, dy = b.y - a[(k + 2) & 3].y;
                float r2 = __builtin_fmaf(dx, dx, c.x);
                r2 = __builtin_fmaf(dy, dy, r2);
                float inv = __builtin_amdgcn_rsqf(r2);
                float inv2 = inv * inv;
                float s = (b.x * inv) * inv2;
                a[k].x = __builtin_fmaf(s, dx, a[k].x);
                a[4 + k].x = __builtin_fmaf(s, dy, a[4 + k].x);
                asm volatile("" : "+v"(a[k].x), "+v"(a[4 + k].x));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += a[k].x + a[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        cyc[w] = t1 - t0;
        rt[w] = r1 - r0;
    }
}

template <int KIND>
static void run(int waves_per_simd, int iters)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;  // 256-thread block = 4 waves = 1 wave per SIMD of one CU
    const int nw = blocks * 4;
    float *out; unsigned long long *cyc, *rt;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CK(hipMalloc(&cyc, nw * 8)); CK(hipMalloc(&rt, nw * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ubench<KIND><<<blocks, 256>>>(out, cyc, rt, iters / 10, 1.0f);  // warm
    CK(hipEventRecord(e0));
    ubench<KIND><<<blocks, 256>>>(out, cyc, rt, iters, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hc(nw), hr(nw);
    CK(hipMemcpy(hc.data(), cyc, nw * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), rt, nw * 8, hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end()); std::sort(hr.begin(), hr.end());
    double med_c = (double)hc[nw / 2], med_r = (double)hr[nw / 2];
    double clk_ghz = med_c / (med_r / 100e6) / 1e9;  // s_memrealtime ticks at 100 MHz
    double insts = (double)kinsts[KIND] * iters;
    // cycles one SIMD spends per wave-instruction when waves_per_simd waves share it
    double cyc_per_inst = med_c / (insts * waves_per_simd);
    double wall_cyc_per_inst = (ms * 1e-3) * clk_ghz * 1e9 / (insts * waves_per_simd);
    printf("%-36s waves/SIMD=%d  cyc/inst(in-kernel)=%.3f  cyc/inst(wall)=%.3f  clk=%.3f GHz  ms=%.3f\n",
           kname[KIND], waves_per_simd, cyc_per_inst, wall_cyc_per_inst, clk_ghz, ms);
    CK(hipFree(out)); CK(hipFree(cyc)); CK(hipFree(rt));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  gcn=%s\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.gcnArchName);
    const int iters = 20000;
    for (int w : {1, 2, 4, 8}) {
        run<K_FMA>(w, iters);
        run<K_PKFMA>(w, iters);
        run<K_PKMUL>(w, iters);
        run<K_PKADD>(w, iters);
        run<K_RSQ>(w, iters);
        run<K_BODY>(w, iters);
        run<K_BODY_SCALAR>(w, iters);
    }
    return 0;
}
