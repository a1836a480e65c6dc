// tools/ubench_valu2.hip — hand-allocated VALU issue-rate probes for gfx950.
//
// Every variant is a fully hand-written asm loop over NAMED registers so that the
// VGPR banks of each operand are known (register index mod 4).  16 independent
// chains per wave, so dependency latency is hidden even at one wave per SIMD.
// Reports ns and cycles per wave-instruction per SIMD from wall time (hipEvents)
// and the in-kernel clock from s_memtime / s_memrealtime of the SAME wave.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu2 tools/ubench_valu2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

// clobber list shared by all variants: v8..v47, s20..s23
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23", \
             "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \
             "v40","v41","v42","v43","v44","v45","v46","v47","s20","s21","s22","s23","scc","memory"

#define INIT \
    "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n" \
    "v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n" \
    "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n" \
    "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n" \
    "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n" \
    "v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n" \
    "v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0\n" \
    "v_mov_b32 v36, 1.0\n v_mov_b32 v37, 1.0\n v_mov_b32 v38, 1.0\n v_mov_b32 v39, 1.0\n" \
    "v_mov_b32 v40, 1.0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 1.0\n v_mov_b32 v43, 0\n" \
    "v_mov_b32 v44, 1.0\n v_mov_b32 v45, 1.0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n" \
    "s_mov_b32 s22, 1.0\n s_mov_b32 s23, 1.0\n"

#define LOOP_BEGIN "s_mov_b32 s20, %0\n 1:\n"
#define LOOP_END   "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"

// 16 single-register chains on v8..v23
#define R16(OP, TAIL) \
    OP " v8, v8"   TAIL "\n" OP " v9, v9"   TAIL "\n" OP " v10, v10" TAIL "\n" OP " v11, v11" TAIL "\n" \
    OP " v12, v12" TAIL "\n" OP " v13, v13" TAIL "\n" OP " v14, v14" TAIL "\n" OP " v15, v15" TAIL "\n" \
    OP " v16, v16" TAIL "\n" OP " v17, v17" TAIL "\n" OP " v18, v18" TAIL "\n" OP " v19, v19" TAIL "\n" \
    OP " v20, v20" TAIL "\n" OP " v21, v21" TAIL "\n" OP " v22, v22" TAIL "\n" OP " v23, v23" TAIL "\n"
// 16 register-pair chains on v[8:9]..v[38:39]
#define P16(OP, TAIL) \
    OP " v[8:9], v[8:9]"     TAIL "\n" OP " v[10:11], v[10:11]" TAIL "\n" OP " v[12:13], v[12:13]" TAIL "\n" OP " v[14:15], v[14:15]" TAIL "\n" \
    OP " v[16:17], v[16:17]" TAIL "\n" OP " v[18:19], v[18:19]" TAIL "\n" OP " v[20:21], v[20:21]" TAIL "\n" OP " v[22:23], v[22:23]" TAIL "\n" \
    OP " v[24:25], v[24:25]" TAIL "\n" OP " v[26:27], v[26:27]" TAIL "\n" OP " v[28:29], v[28:29]" TAIL "\n" OP " v[30:31], v[30:31]" TAIL "\n" \
    OP " v[32:33], v[32:33]" TAIL "\n" OP " v[34:35], v[34:35]" TAIL "\n" OP " v[36:37], v[36:37]" TAIL "\n" OP " v[38:39], v[38:39]" TAIL "\n"

struct Variant { const char *name; int flops_per_lane_inst; };
static const Variant variants[] = {
    {"v_fma_f32 d,d,v40,v41 (3 vgpr)", 2},
    {"v_fma_f32 d,d,1.0,v41 (2 vgpr)", 2},
    {"v_fma_f32 d,d,s22,v41 (sgpr)", 2},
    {"v_mul_f32 d,d,v40 (VOP2)", 1},
    {"v_add_f32 d,d,v41 (VOP2)", 1},
    {"v_fmac_f32 d,v40,v41 (VOP2 fmac)", 2},
    {"v_pk_fma_f32 D,D,v[40:41],v[42:43]", 4},
    {"v_pk_fma_f32 D,D,v[44:45],v[46:47]", 4},
    {"v_pk_fma_f32 D,D,s[22:23],v[42:43]", 4},
    {"v_pk_mul_f32 D,D,v[44:45]", 2},
    {"v_pk_add_f32 D,D,v[46:47]", 2},
    {"v_rsq_f32 d,d", 1},
    {"v_rsq_f32 x8 + v_pk_fma x8 interleaved", 0},
    {"v_rcp_f32 d,d", 1},
    {"v_sqrt_f32 d,d", 1},
};
constexpr int NVAR = sizeof(variants) / sizeof(variants[0]);

template <int V>
__global__ __launch_bounds__(256) void probe(unsigned long long *cyc, unsigned long long *rt, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (V == 0)  asm volatile(INIT LOOP_BEGIN R16("v_fma_f32", ", v40, v41") LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 1)  asm volatile(INIT LOOP_BEGIN R16("v_fma_f32", ", 1.0, v41") LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 2)  asm volatile(INIT LOOP_BEGIN R16("v_fma_f32", ", s22, v41") LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 3)  asm volatile(INIT LOOP_BEGIN R16("v_mul_f32", ", v40") LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 4)  asm volatile(INIT LOOP_BEGIN R16("v_add_f32", ", v41") LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 5)  asm volatile(INIT LOOP_BEGIN
        "v_fmac_f32 v8, v40, v41\n v_fmac_f32 v9, v40, v41\n v_fmac_f32 v10, v40, v41\n v_fmac_f32 v11, v40, v41\n"
        "v_fmac_f32 v12, v40, v41\n v_fmac_f32 v13, v40, v41\n v_fmac_f32 v14, v40, v41\n v_fmac_f32 v15, v40, v41\n"
        "v_fmac_f32 v16, v40, v41\n v_fmac_f32 v17, v40, v41\n v_fmac_f32 v18, v40, v41\n v_fmac_f32 v19, v40, v41\n"
        "v_fmac_f32 v20, v40, v41\n v_fmac_f32 v21, v40, v41\n v_fmac_f32 v22, v40, v41\n v_fmac_f32 v23, v40, v41\n"
        LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 6)  asm volatile(INIT LOOP_BEGIN P16("v_pk_fma_f32", ", v[40:41], v[42:43]") LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 7)  asm volatile(INIT LOOP_BEGIN P16("v_pk_fma_f32", ", v[44:45], v[46:47]") LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 8)  asm volatile(INIT LOOP_BEGIN P16("v_pk_fma_f32", ", s[22:23], v[42:43]") LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 9)  asm volatile(INIT LOOP_BEGIN P16("v_pk_mul_f32", ", v[44:45]") LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 10) asm volatile(INIT LOOP_BEGIN P16("v_pk_add_f32", ", v[46:47]") LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 11) asm volatile(INIT LOOP_BEGIN R16("v_rsq_f32", "") LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 12) asm volatile(INIT LOOP_BEGIN
        "v_rsq_f32 v8, v8\n v_pk_fma_f32 v[24:25], v[24:25], v[44:45], v[46:47]\n"
        "v_rsq_f32 v9, v9\n v_pk_fma_f32 v[26:27], v[26:27], v[44:45], v[46:47]\n"
        "v_rsq_f32 v10, v10\n v_pk_fma_f32 v[28:29], v[28:29], v[44:45], v[46:47]\n"
        "v_rsq_f32 v11, v11\n v_pk_fma_f32 v[30:31], v[30:31], v[44:45], v[46:47]\n"
        "v_rsq_f32 v12, v12\n v_pk_fma_f32 v[32:33], v[32:33], v[44:45], v[46:47]\n"
        "v_rsq_f32 v13, v13\n v_pk_fma_f32 v[34:35], v[34:35], v[44:45], v[46:47]\n"
        "v_rsq_f32 v14, v14\n v_pk_fma_f32 v[36:37], v[36:37], v[44:45], v[46:47]\n"
        "v_rsq_f32 v15, v15\n v_pk_fma_f32 v[38:39], v[38:39], v[44:45], v[46:47]\n"
        LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 13) asm volatile(INIT LOOP_BEGIN R16("v_rcp_f32", "") LOOP_END :: "s"(iters) : CLOB);
    if constexpr (V == 14) asm volatile(INIT LOOP_BEGIN R16("v_sqrt_f32", "") LOOP_END :: "s"(iters) : CLOB);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        cyc[w] = t1 - t0;
        rt[w] = r1 - r0;
    }
}

template <int V>
static void run(int wps, int iters)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * wps;
    const int nw = blocks * 4;
    unsigned long long *cyc, *rt;
    CK(hipMalloc(&cyc, nw * 8)); CK(hipMalloc(&rt, nw * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<V><<<blocks, 256>>>(cyc, rt, iters / 4);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    probe<V><<<blocks, 256>>>(cyc, rt, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hc(nw), hr(nw);
    CK(hipMemcpy(hc.data(), cyc, nw * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), rt, nw * 8, hipMemcpyDeviceToHost));
    // per-wave clock = its own cycles / its own real time; take the median wave
    std::vector<double> clk(nw), dur(nw);
    for (int i = 0; i < nw; ++i) { clk[i] = (double)hc[i] / ((double)hr[i] / 100e6) / 1e9; dur[i] = (double)hr[i] / 100e6 * 1e3; }
    std::sort(clk.begin(), clk.end()); std::sort(dur.begin(), dur.end());
    const double insts = 16.0 * iters;                 // per wave
    const double ns_per_inst = ms * 1e6 / (insts * wps);  // SIMD time per wave-instruction
    const double cyc_per_inst = ns_per_inst * clk[nw / 2];
    const double tflops = variants[V].flops_per_lane_inst * 64.0 / ns_per_inst * 1e9 * 4 * prop.multiProcessorCount / 1e12;
    printf("%-42s w/SIMD=%d  ms=%7.3f  wave_ms[med]=%7.3f  ns/inst=%.3f  clk=%.3f GHz  cyc/inst=%.2f  -> %.1f TFLOP/s\n",
           variants[V].name, wps, ms, dur[nw / 2], ns_per_inst, clk[nw / 2], cyc_per_inst, tflops);
    CK(hipFree(cyc)); CK(hipFree(rt)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int V>
static void run_all(int iters)
{
    for (int w : {1, 2, 4}) run<V>(w, iters / w);
    if constexpr (V + 1 < NVAR) run_all<V + 1>(iters);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s CUs=%d\n", prop.gcnArchName, prop.multiProcessorCount);
    run_all<0>(400000);
    return 0;
}
