#!/usr/bin/env python3
"""tools/host_cost.py — what ONE sharded step costs the HOST in the Python/torch driver.

A one-GPU box cannot run 8 RCCL ranks, but the host-side cost of a step does not depend on the transport: it is
the ctypes calls into the library, the torch.distributed calls that enqueue the collectives (through RCCL itself,
world size 1) and the event records.  This tool builds rank P/2's handle of a P-rank run (default P = 8,
N = 262 144), drives exactly the call sequence of nbodysim_amd.dist.DistributedSimulation.step for each protocol
with the collectives issued on a 1-rank RCCL group over tensors of the real sizes, and reports
    host_enqueue_us   wall time of the Python loop per step while the GPU queue is never empty and never full
                      (the loop is timed over `steps` steps without any synchronisation)
    gpu_ms            wall time per step including the final synchronisation (the GPU's own time for this share)
with HIP-event phase marks off and on.  The physics is wrong (no real exchange); the launches and calls are real.

    python tools/host_cost.py [--world 8] [--n 262144] [--steps 300]
"""
import argparse
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29713")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import nbodysim_amd as nb  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--n", type=int, default=262144)
ap.add_argument("--steps", type=int, default=300)
args = ap.parse_args()
P, n, steps = args.world, args.n, args.steps
rank, blk = P // 2, n // P

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ic = nb.plummer_2d(n, 42)
stream = torch.cuda.Stream()
print(f"host cost of one sharded step: rank {rank} of {P}, N = {n}, {steps} steps per figure; torch {torch.__version__}", flush=True)

for name, kw in (("symmetric", dict(sym_late_us=-1.0)), ("symmetric+late", dict(sym_late_us=40.0)),
                 ("allreduce", dict(shard_allreduce=True)), ("allgather", dict(symmetry=False))):
    replicated = name == "allreduce"
    pos = [torch.zeros((n, 2), dtype=torch.float32, device="cuda") for _ in range(2)]
    acc_full = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
    acc_owned = torch.zeros((blk, 2), dtype=torch.float32, device="cuda")
    rs_out = torch.zeros((n, 2), dtype=torch.float32, device="cuda")      # world 1: reduce-scatter output = input size
    own = dict(i_begin=0, i_count=n) if replicated else dict(i_begin=rank * blk, i_count=blk)
    acc = None if name == "allgather" else ((acc_full.data_ptr(), acc_full.data_ptr()) if replicated else (acc_full.data_ptr(), acc_owned.data_ptr()))
    with nb.Simulation(ic, eps=0.01, shard_rank=rank, shard_world=P, stream=stream.cuda_stream,
                       pos_buffers=(pos[0].data_ptr(), pos[1].data_ptr()), acc_buffers=acc, **own, **kw) as s:
        symmetric = s.shard_protocol == nb._lib.NB_SHARD_SYMMETRIC
        state = {"pending": None, "cur": 0}

        def step(marks):
            def mark():
                if marks is not None:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(stream)
                    marks.append(ev)
            with torch.cuda.stream(stream):
                mark()
                s.step_begin(1e-3)
                mark()
                if replicated:
                    dist.all_reduce(acc_full, op=dist.ReduceOp.SUM)
                    mark()
                    s.step_finish()
                    mark()
                    return
                if state["pending"] is not None:
                    state["pending"].wait()
                    state["pending"] = None
                mark()
                if symmetric:
                    s.step_mid()
                    mark()
                    dist.reduce_scatter_tensor(rs_out, acc_full, op=dist.ReduceOp.SUM)
                    mark()
                s.step_finish()
                mark()
                state["cur"] ^= 1
                # world 1: gather of my own block into a block-sized tensor (same call, same message size per rank)
                blockt = pos[state["cur"]][rank * blk:(rank + 1) * blk]
                state["pending"] = dist.all_gather_into_tensor(blockt, blockt, async_op=True)

        def drain():
            with torch.cuda.stream(stream):
                if state["pending"] is not None:
                    state["pending"].wait()
                    state["pending"] = None
            stream.synchronize()

        for events in (False, True):
            for _ in range(10):
                step(None)
            drain()
            keep = []
            t0 = time.perf_counter()
            for _ in range(steps):
                m = [] if events else None
                step(m)
                if m is not None:
                    keep.append(m)
            t_host = time.perf_counter() - t0
            drain()
            t_all = time.perf_counter() - t0
            print(f"  {name:15s} events={'on ' if events else 'off'}  host_enqueue {t_host / steps * 1e6:7.1f} us/step   "
                  f"gpu {t_all / steps * 1e3:6.3f} ms/step   host/gpu {t_host / t_all * 100:5.1f} %", flush=True)
        # the three library calls alone (no collectives, no torch): what the C ABI itself costs per step
        for _ in range(10):
            s.step_begin(1e-3); s.step_mid(); s.step_finish()
        s.wait()
        t0 = time.perf_counter()
        for _ in range(steps):
            s.step_begin(1e-3); s.step_mid(); s.step_finish()
        t_host = time.perf_counter() - t0
        s.wait()
        t_all = time.perf_counter() - t0
        print(f"  {name:15s} library calls only (ctypes, no collectives): host {t_host / steps * 1e6:7.1f} us/step   gpu {t_all / steps * 1e3:6.3f} ms/step",
              flush=True)
dist.destroy_process_group()
