/* nbody_debug.h — TEST HOOKS of libnbody_hip.  NOT part of the product ABI.
 *
 * The shipped library (nbodysim_amd/libnbody_hip.so, `make -C nbodysim_amd/csrc`) does not contain these functions at
 * all: they are compiled only with -DNB_TEST_HOOKS, which the test build does (tests/libnbody_hip_testhooks.so,
 * `make -C nbodysim_amd/csrc hooks`, built by __graft_entry__.build()).  A maintainer binding the library binds
 * include/nbody.h and never sees this file; INTEGRATION.md does not cite it.
 *
 * What is here and why it cannot be a product symbol:
 *   nb_debug_comm_transport   makes the library dlopen a NAMED file in place of librccl (tests/loopback_rccl.hip)
 *   nb_debug_comm_schedule    the step schedule of nb_comm_step as data (host-only checks of call order and counts)
 *   nb_debug_sym_plan         the symmetric planner without a GPU (coverage / balance checks on the CPU)
 *   nb_debug_ticket_seed      sets a handle's item-ticket counters (steps a handle across their 2^32 wrap)
 *   nb_debug_fast_inv_sqrt    the device's two Quake-rsqrt forms on an array (bit check against the reference's grid)
 */
#ifndef NBODY_DEBUG_H
#define NBODY_DEBUG_H

#include "nbody.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* Test hook: load the nccl* entry points from `path` instead of librccl.so.1 (tests/loopback_rccl.hip: an in-process
 * transport that lets several ranks share one device, so nb_comm_step can run with 2 and 4 members on a one-GPU box).
 * Call before anything has loaded the transport; NULL restores the default. */
int      nb_debug_comm_transport(const char *path);

/* The schedule of ONE step as data: what nb_comm_step issues, in order (host-only view; the CPU tests check call order
 * and element counts with it).  kind: NB_OP_*; handle: index into the process's handle list (-1 for the group ops);
 * stream: 0 = the handle's compute stream, 1 = its communication stream; event: NB_EV_* (-1 if none);
 * count: elements (reals) of a collective — per rank for all-gather (send) and reduce-scatter (receive). */
typedef struct nb_comm_op { int32_t kind, handle, stream, event; uint64_t count; } nb_comm_op;
enum { NB_OP_BEGIN = 0, NB_OP_MID = 1, NB_OP_FINISH = 2, NB_OP_RECORD = 3, NB_OP_WAIT = 4, NB_OP_ALLGATHER = 5,
       NB_OP_REDUCE_SCATTER = 6, NB_OP_ALLREDUCE = 7, NB_OP_GROUP_START = 8, NB_OP_GROUP_END = 9 };
enum { NB_EV_POS = 0, NB_EV_AG = 1, NB_EV_ACC = 2, NB_EV_RED = 3 };
int nb_debug_comm_schedule(int protocol, int handles, uint64_t block_reals, uint64_t full_reals, int ag_pending,
                           nb_comm_op *ops_out, size_t cap, size_t *count);

/* Host-only view of the planner (no GPU needed; used by the CPU tests to check that the items of all ranks
 * cover every unordered (tile, chunk) pair exactly once, that the slab ranges are disjoint, and the balance).
 * `tuning` may be NULL (defaults) — only flags, sym_chunks_per_item, sym_late_us, sym_tail and precision are read.
 * items_out receives up to cap items, local ones first, then cross, then late. */
int nb_debug_sym_plan(size_t n, int cus, int rank, int world, const nb_params *tuning,
                      nb_sym_item *items_out, size_t cap, nb_sym_info *info);

/* Test hook for the dynamic work items (NB_FLAG_STATIC_ITEMS above): sets the handle's item-ticket counters, and the host's record of
 * what has been drawn, to `value` — a test steps a handle across the 2^32 wrap of the counters this way.  Waits for the handle's streams. */
int nb_debug_ticket_seed(nb_sim *s, uint32_t value);

/* Quadtree::fast_inv_sqrt (Quadtree.hpp:106-111) exactly as the device kernels evaluate it, on an array of n (even)
 * floats: y_scalar through the scalar form (reference-order kernel), y_packed through the packed form (tiled and
 * symmetric kernels).  Test hook for the bit-exact check against the reference's golden grid. */
int nb_debug_fast_inv_sqrt(const float *x, float *y_scalar, float *y_packed, size_t n);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* NBODY_DEBUG_H */
