// tools/pin_probe.hip — ONE deterministic construction per run of the host-memory situations that the round-2
// abort inside nb_upload could have come from.  Not product code: it does on
// purpose what the library no longer does (hands pageable pointers to hipMemcpyAsync, registers unaligned ranges,
// frees registered memory).  Each scenario is its own process (tools/pin_probe.sh), runs once, prints what every
// HIP call returned, and ends with "scenario X: completed" — or dies, and the shell records the signal.
//
//   hipcc --offload-arch=gfx950 -O1 -o build/pin_probe tools/pin_probe.hip
//   build/pin_probe <scenario>
//
// control        pageable 2.56 MB H2D copy, nothing registered
// adjacent       one 8 MiB page-aligned buffer; register the unaligned sub-range [B+16, B+16+2.56 MB); then
//                hipMemcpyAsync H2D from the pageable bytes that FOLLOW it (they share the boundary page)
// adjacent_heap  the sequence of tests/test_host_gpu.py:164-169 on the brk heap: mmap threshold raised, `out` and
//                `want` malloc'ed back to back (2.56 MB each, 16-byte aligned, sharing a page), register(out),
//                hipMemcpyAsync H2D from want
// before         the copy source ENDS inside the first registered page
// overlap_reg    register [B+16, +2.56 MB) and then a second range that overlaps its last page
// stale_reuse    register a malloc'ed (mmapped) block, free() it WITHOUT unregistering, malloc the same size again
//                (glibc hands the address range back), H2D copy from the new block, then register it
// stale_reuse_mmap  the same with the second block forced to be an mmap at the old address (fixed M_MMAP_THRESHOLD)
// unreg_freed    register an mmapped block, free() it, hipHostUnregister(old pointer)   (what host/Simulation.hpp:202
//                did after the vector reallocated)
// d2h_adjacent   like `adjacent`, device -> host into the pageable neighbour
#include <hip/hip_runtime.h>

#include <malloc.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const size_t BYTES = 40000 * 64;   // the 2.56 MB arrays of the test

#define SAY(call)                                                                                       \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        printf("  %-70s -> %s\n", #call, hipGetErrorName(e_));                                          \
        fflush(stdout);                                                                                 \
        if (e_ != hipSuccess) (void)hipGetLastError();                                                  \
    } while (0)

static void fill(void *p, size_t n, int v) { memset(p, v, n); }

int main(int argc, char **argv)
{
    const char *sc = argc > 1 ? argv[1] : "control";
    setvbuf(stdout, NULL, _IOLBF, 0);
    printf("scenario %s\n", sc);
    void *dev = nullptr;
    hipStream_t st;
    SAY(hipSetDevice(0));
    SAY(hipMalloc(&dev, BYTES));
    SAY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

    if (!strcmp(sc, "control")) {
        char *src = (char *)malloc(BYTES);
        fill(src, BYTES, 1);
        printf("  src %p (page offset %zu)\n", (void *)src, (size_t)((uintptr_t)src & 4095));
        SAY(hipMemcpyAsync(dev, src, BYTES, hipMemcpyHostToDevice, st));
        SAY(hipStreamSynchronize(st));
        free(src);
    } else if (!strcmp(sc, "adjacent") || !strcmp(sc, "d2h_adjacent") || !strcmp(sc, "before") || !strcmp(sc, "overlap_reg")) {
        char *B = nullptr;
        if (posix_memalign((void **)&B, 4096, 8u << 20)) return 2;
        fill(B, 8u << 20, 2);
        char *reg = B + 4096 + 16;                      // unaligned start, unaligned end: like a malloc'ed array
        printf("  buffer %p, registered [%p, %p)\n", (void *)B, (void *)reg, (void *)(reg + BYTES));
        SAY(hipHostRegister(reg, BYTES, hipHostRegisterDefault));
        if (!strcmp(sc, "adjacent")) {
            char *src = reg + BYTES + 16;               // starts in the registered range's last page
            printf("  pageable source [%p, %p): first page shared with the registered range\n", (void *)src, (void *)(src + BYTES));
            SAY(hipMemcpyAsync(dev, src, BYTES, hipMemcpyHostToDevice, st));
            SAY(hipStreamSynchronize(st));
        } else if (!strcmp(sc, "d2h_adjacent")) {
            char *dst = reg + BYTES + 16;
            SAY(hipMemcpyAsync(dst, dev, BYTES, hipMemcpyDeviceToHost, st));
            SAY(hipStreamSynchronize(st));
        } else if (!strcmp(sc, "before")) {
            char *src = B + 32;                         // 4096 + 16 - 32 bytes before `reg`; ends inside its first page
            SAY(hipMemcpyAsync(dev, src, 4096, hipMemcpyHostToDevice, st));
            SAY(hipStreamSynchronize(st));
        } else {
            char *second = reg + BYTES - 100;           // overlaps the last 100 bytes (and page) of the first range
            SAY(hipHostRegister(second, BYTES, hipHostRegisterDefault));
            SAY(hipHostUnregister(second));
        }
        SAY(hipHostUnregister(reg));
        free(B);
    } else if (!strcmp(sc, "adjacent_heap")) {
        mallopt(M_MMAP_THRESHOLD, 64 << 20);            // what glibc's dynamic threshold does after large arrays were freed
        mallopt(M_TRIM_THRESHOLD, 256 << 20);
        char *out = (char *)malloc(BYTES), *want = (char *)malloc(BYTES);
        fill(out, BYTES, 3); fill(want, BYTES, 4);
        printf("  out [%p, %p)  want [%p, %p)  same page at the joint: %s\n", (void *)out, (void *)(out + BYTES), (void *)want,
               (void *)(want + BYTES), (((uintptr_t)(out + BYTES - 1)) >> 12) == (((uintptr_t)want) >> 12) ? "yes" : "no");
        SAY(hipHostRegister(out, BYTES, hipHostRegisterDefault));
        SAY(hipMemcpyAsync(dev, want, BYTES, hipMemcpyHostToDevice, st));
        SAY(hipStreamSynchronize(st));
        SAY(hipMemcpyAsync(want, dev, BYTES, hipMemcpyDeviceToHost, st));
        SAY(hipStreamSynchronize(st));
        SAY(hipHostUnregister(out));
        free(want); free(out);
    } else if (!strcmp(sc, "stale_reuse") || !strcmp(sc, "stale_reuse_mmap")) {
        // _mmap: a fixed threshold switches glibc's dynamic adjustment off, so the second block is mmapped again and
        // (with nothing else mapped in between) lands on the address range the first one had
        if (!strcmp(sc, "stale_reuse_mmap")) mallopt(M_MMAP_THRESHOLD, 128 << 10);
        char *a = (char *)malloc(BYTES);                // > 128 KiB: its own mmap
        fill(a, BYTES, 5);
        printf("  first block %p\n", (void *)a);
        SAY(hipHostRegister(a, BYTES, hipHostRegisterDefault));
        free(a);                                        // munmap of registered pages; the runtime is not told
        char *b = (char *)malloc(BYTES);
        fill(b, BYTES, 6);
        printf("  second block %p (%s)\n", (void *)b, a == b ? "same address" : "different address");
        SAY(hipMemcpyAsync(dev, b, BYTES, hipMemcpyHostToDevice, st));
        SAY(hipStreamSynchronize(st));
        SAY(hipHostRegister(b, BYTES, hipHostRegisterDefault));
        SAY(hipMemcpyAsync(dev, b, BYTES, hipMemcpyHostToDevice, st));
        SAY(hipStreamSynchronize(st));
        SAY(hipHostUnregister(b));
        free(b);
    } else if (!strcmp(sc, "unreg_freed")) {
        char *a = (char *)malloc(BYTES);
        fill(a, BYTES, 7);
        SAY(hipHostRegister(a, BYTES, hipHostRegisterDefault));
        free(a);
        SAY(hipHostUnregister(a));
    } else {
        printf("unknown scenario\n");
        return 2;
    }
    SAY(hipStreamDestroy(st));
    SAY(hipFree(dev));
    printf("scenario %s: completed\n", sc);
    return 0;
}
