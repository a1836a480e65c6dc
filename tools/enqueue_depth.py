#!/usr/bin/env python3
"""Host cost of nb_comm_step against the number of steps already in flight (one rank through RCCL, real buffer sizes).

profiles/r03_soak.log showed 1630 us of host time per step over 3000 steps where 200 steps cost 40 us each: the host
runs ahead of the GPU until the HIP queue of a stream is full, then every further enqueue BLOCKS until the device has
retired a packet — from that depth on the host is paced by the device (it does not lose time: it has nothing else to do).
This measures where that happens: steps are enqueued in batches without waiting, and each batch's enqueue time per
step is printed with the depth (steps enqueued minus steps the device has completed, from nb_frame... here: from time).

    python tools/enqueue_depth.py [--n 262144] [--world 8] [--rank 4] [--allreduce]
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import nbodysim_amd as nb  # noqa: E402
from nbodysim_amd.comm import Comm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=262144)
    ap.add_argument("--allreduce", action="store_true")
    ap.add_argument("--batch", type=int, default=50)
    ap.add_argument("--batches", type=int, default=40)
    args = ap.parse_args()
    ic = nb.plummer_2d(args.n, 42)
    with nb.Simulation(ic, eps=0.01, shard_rank=0, shard_world=1, shard_single=True, shard_allreduce=args.allreduce) as s:
        with Comm.all([s]) as comm:
            comm.step(5, 1e-3)
            comm.wait()
            t0 = time.perf_counter()
            comm.step(args.batch, 1e-3)
            comm.wait()
            dev_ms = (time.perf_counter() - t0) / args.batch * 1e3           # device time of one step (the host waited)
            print(f"n={args.n} one rank, protocol {'allreduce' if args.allreduce else 'symmetric'}: device {dev_ms*1e3:.0f} us/step")
            start = time.perf_counter()
            enq = 0
            for b in range(args.batches):
                t0 = time.perf_counter()
                comm.step(args.batch, 1e-3)
                t1 = time.perf_counter()
                enq += args.batch
                done = min(enq, int((t1 - start) * 1e3 / dev_ms))             # steps the device can have retired by now
                print(f"  batch {b:2d}: host {(t1 - t0) / args.batch * 1e6:8.1f} us/step   enqueued {enq:5d}   in flight ~{enq - done:5d}", flush=True)
            comm.wait()
            print(f"  all {enq} steps done after {(time.perf_counter() - start) * 1e3:.1f} ms = {(time.perf_counter() - start) / enq * 1e6:.0f} us/step")


if __name__ == "__main__":
    main()
