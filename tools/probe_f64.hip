// tools/probe_f64.hip — accuracy of v_rsq_f64 (+ refinement steps) and fp64 issue rates on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ void acc_probe(const double *x, double *raw, double *n1, double *n3, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    raw[i] = y;
    double e = __builtin_fma(-v * y, y, 1.0);
    n1[i] = __builtin_fma(y * 0.5, e, y);
    double c = __builtin_fma(e, 0.375, 0.5);
    n3[i] = __builtin_fma(y * e, c, y);
}

template <int KIND>
__global__ __launch_bounds__(1024) void rate_probe(double *out, unsigned long long *cyc, int iters)
{
    extern __shared__ char pad[];   // occupancy cap: one 1024-thread workgroup per CU -> exactly 4 waves per SIMD
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0 + threadIdx.x * 1e-6 + k;
    const double b = 1.0000001, c = 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == 1) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[k]));
            if (KIND == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            if (KIND == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            if (KIND == 4) asm volatile("v_rsq_f32 %0, %0" : "+v"(*(float *)&a[k]));
            if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float *)&a[k]) : "v"((float)b), "v"((float)c));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
    (void)pad;
}

template <int KIND> void rate(const char *name)
{
    int cus = 256, iters = 4000;
    double *out; unsigned long long *cyc;
    CK(hipMalloc(&out, (size_t)cus * 1024 * 8)); CK(hipMalloc(&cyc, cus * 16 * 8));
    CK(hipFuncSetAttribute((const void *)rate_probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    rate_probe<KIND><<<cus, 1024, 100 * 1024>>>(out, cyc, iters / 4);
    rate_probe<KIND><<<cus, 1024, 100 * 1024>>>(out, cyc, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(cus * 16);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    // 4 waves per SIMD, each issuing 32*iters instructions
    printf("%-14s cycles per wave-instruction per SIMD (4 waves/SIMD, in-wave clock): %.2f   [min wave %.2f max wave %.2f]\n", name,
           med / (32.0 * iters * 4), (double)h.front() / (32.0 * iters * 4), (double)h.back() / (32.0 * iters * 4));
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main()
{
    const int n = 1 << 20;
    std::mt19937_64 rng(1);
    std::vector<double> x(n);
    for (auto &v : x) v = std::exp(std::uniform_real_distribution<double>(-20, 20)(rng));
    double *dx, *d0, *d1, *d3;
    CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&d0, n * 8)); CK(hipMalloc(&d1, n * 8)); CK(hipMalloc(&d3, n * 8));
    CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
    acc_probe<<<n / 256, 256>>>(dx, d0, d1, d3, n);
    std::vector<double> r0(n), r1(n), r3(n);
    CK(hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r3.data(), d3, n * 8, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0, e3 = 0;
    for (int i = 0; i < n; ++i) {
        long double ex = 1.0L / sqrtl((long double)x[i]);
        e0 = std::max(e0, (double)fabsl((r0[i] - ex) / ex));
        e1 = std::max(e1, (double)fabsl((r1[i] - ex) / ex));
        e3 = std::max(e3, (double)fabsl((r3[i] - ex) / ex));
    }
    printf("v_rsq_f64 max rel err: raw %.3e   +1 Newton %.3e   +3rd-order %.3e   (eps_f64 = 1.1e-16)\n", e0, e1, e3);
    rate<0>("v_fma_f64"); rate<2>("v_mul_f64"); rate<3>("v_add_f64"); rate<1>("v_rsq_f64");
    return 0;
}
