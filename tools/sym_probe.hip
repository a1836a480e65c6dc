// tools/sym_probe.hip — feasibility probe for a symmetric (Newton's third law) pair kernel:
// a wave keeps 2P i-particles per lane stationary and rotates Q j-particles per lane (with their
// accumulators) through the 64 lanes; each (i-pair, j) body updates BOTH sides.
// Measures SIMD cycles per body against the one-sided body of the product kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE> __device__ __forceinline__ float rot(float v, int addr)
{
    if constexpr (MODE == 0) return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
    else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, false)); // wave_ror:1
}

// MODE 0: ds_bpermute rotation, 1: DPP wave_ror:1, 2: no rotation (upper bound), 3: one-sided body (today's kernel)
template <int P, int Q, int MODE>
__global__ __launch_bounds__(256) void probe(const float2 *pos, float2 *out, int n, int chunks, float eps2)
{
    __shared__ float2 lpos[4][Q][64];
    __shared__ float2 lacc[4][Q][64];
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int addr = ((lane + 1) & 63) * 4;
    v2f xi[P], yi[P], ax[P], ay[P];
    for (int p = 0; p < P; ++p) {
        int i0 = (gw * 64 * 2 * P + p * 128 + 2 * lane) % n;
        xi[p] = (v2f){pos[i0].x, pos[i0 + 1].x}; yi[p] = (v2f){pos[i0].y, pos[i0 + 1].y};
        ax[p] = (v2f){0, 0}; ay[p] = (v2f){0, 0};
    }
    const v2f e2 = {eps2, eps2};
    float sx = 0, sy = 0;
    for (int c = 0; c < chunks; ++c) {
        float xq[Q], yq[Q]; v2f aqx[Q], aqy[Q];
        for (int q = 0; q < Q; ++q) {
            int j = ((c * Q + q) * 64 + lane + gw * 7) % n;
            xq[q] = pos[j].x; yq[q] = pos[j].y; aqx[q] = (v2f){0, 0}; aqy[q] = (v2f){0, 0};
            if constexpr (MODE == 6) { lpos[threadIdx.x >> 6][q][lane] = pos[j]; lacc[threadIdx.x >> 6][q][lane] = make_float2(0, 0); }
        }
#pragma unroll 2
        for (int step = 0; step < 64; ++step) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const v2f xj = {xq[q], xq[q]}, yj = {yq[q], yq[q]};
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const v2f dx = xj - xi[p], dy = yj - yi[p];
                    v2f r2 = __builtin_elementwise_fma(dx, dx, e2);
                    r2 = __builtin_elementwise_fma(dy, dy, r2);
                    const v2f inv = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                    const v2f s = inv * (inv * inv);
                    ax[p] = __builtin_elementwise_fma(s, dx, ax[p]);
                    ay[p] = __builtin_elementwise_fma(s, dy, ay[p]);
                    if constexpr (MODE != 3) {
                        aqx[q] = __builtin_elementwise_fma(-s, dx, aqx[q]);
                        aqy[q] = __builtin_elementwise_fma(-s, dy, aqy[q]);
                    }
                }
            }
            if constexpr (MODE == 4) {
#pragma unroll
                for (int q = 0; q < Q; ++q) { xq[q] = rot<0>(xq[q], addr); yq[q] = rot<0>(yq[q], addr);
                    asm volatile("" : "+v"(aqx[q]), "+v"(aqy[q])); }
            } else if constexpr (MODE == 5) {
#pragma unroll
                for (int q = 0; q < Q; ++q) asm volatile("" : "+v"(xq[q]), "+v"(yq[q]), "+v"(aqx[q]), "+v"(aqy[q]));
            } else if constexpr (MODE == 6) {
                // positions come from LDS with a rotating index; the step's j-side sum goes back with ds_add_f32
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int slot = (lane + step + 1) & 63, cur = (lane + step) & 63;
                    atomicAdd(&lacc[threadIdx.x >> 6][q][cur].x, aqx[q].x + aqx[q].y);
                    atomicAdd(&lacc[threadIdx.x >> 6][q][cur].y, aqy[q].x + aqy[q].y);
                    aqx[q] = (v2f){0, 0}; aqy[q] = (v2f){0, 0};
                    const float2 pn = lpos[threadIdx.x >> 6][q][slot];
                    xq[q] = pn.x; yq[q] = pn.y;
                }
            } else if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    xq[q] = rot<MODE>(xq[q], addr); yq[q] = rot<MODE>(yq[q], addr);
                    aqx[q].x = rot<MODE>(aqx[q].x, addr); aqx[q].y = rot<MODE>(aqx[q].y, addr);
                    aqy[q].x = rot<MODE>(aqy[q].x, addr); aqy[q].y = rot<MODE>(aqy[q].y, addr);
                }
            } else if constexpr (MODE == 3) {
#pragma unroll
                for (int q = 0; q < Q; ++q) { xq[q] = rot<0>(xq[q], addr); yq[q] = rot<0>(yq[q], addr); }
            }
        }
        for (int q = 0; q < Q; ++q) { sx += aqx[q].x + aqx[q].y; sy += aqy[q].x + aqy[q].y;
            if constexpr (MODE == 6) { sx += lacc[threadIdx.x >> 6][q][lane].x; sy += lacc[threadIdx.x >> 6][q][lane].y; } }
    }
    float2 r = make_float2(sx, sy);
    for (int p = 0; p < P; ++p) { r.x += ax[p].x + ax[p].y; r.y += ay[p].x + ay[p].y; }
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int P, int Q, int MODE> void run(const float2 *pos, float2 *out, int n, const char *name)
{
    const int blocks = 256 * 8, chunks = 64 / Q;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<P, Q, MODE><<<blocks, 256>>>(pos, out, n, chunks, 1e-4f);
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        probe<P, Q, MODE><<<blocks, 256>>>(pos, out, n, chunks, 1e-4f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    // bodies per wave = chunks * 64 steps * Q * P ; each body = 2 i x 1 j (128 pairs per wave-body)
    const double bodies = (double)blocks * 4 * chunks * 64 * Q * P;
    const double ordered = bodies * 128 * (MODE == 3 ? 1 : 2);   // ordered interactions delivered
    printf("%-34s P=%d Q=%d: %.3f ms  %.2f ns per wave-body per SIMD  -> %.2fe12 ordered interactions/s\n", name, P, Q, ms[2],
           ms[2] * 1e6 / (bodies / 1024.0), ordered / (ms[2] * 1e-3) / 1e12);
}

int main()
{
    const int n = 262144;
    std::vector<float2> h(n);
    srand(1);
    for (auto &p : h) p = make_float2(rand() / (float)RAND_MAX * 4 - 2, rand() / (float)RAND_MAX * 4 - 2);
    float2 *pos, *out; CK(hipMalloc(&pos, n * 8)); CK(hipMalloc(&out, 256 * 8 * 256 * 8));
    CK(hipMemcpy(pos, h.data(), n * 8, hipMemcpyHostToDevice));
    run<4, 1, 3>(pos, out, n, "one-sided (today), bpermute x,y");
    run<4, 1, 5>(pos, out, n, "two-sided, NO rotation (bound)");
    run<4, 1, 4>(pos, out, n, "two-sided, rotate x,y only (2 bperm)");
    run<4, 1, 0>(pos, out, n, "two-sided, 6 bpermute");
    run<4, 1, 6>(pos, out, n, "two-sided, LDS read + 2 ds_add_f32");
    run<4, 2, 6>(pos, out, n, "two-sided, LDS read + 2 ds_add_f32");
    run<4, 1, 1>(pos, out, n, "two-sided, DPP wave_ror");
    return 0;
}
