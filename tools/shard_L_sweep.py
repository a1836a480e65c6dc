import sys, time
sys.path.insert(0, ".")
import nbodysim_amd as nb
n, steps = 262144, 80
ic = nb.plummer_2d(n, 42)
for parts in (8, 4):
    rank, blk = parts // 2, n // parts
    for proto in ("allreduce", "symmetric"):
        for rep in range(2):
            row = []
            for L in (0, 6, 9, 12, 16, 20, 24, 32):
                own = dict(i_begin=0, i_count=n, shard_allreduce=True) if proto == "allreduce" else dict(i_begin=rank * blk, i_count=blk, sym_late_us=-1.0)
                with nb.Simulation(ic, eps=0.01, shard_rank=rank, shard_world=parts, sym_chunks_per_item=L, **own) as s:
                    def go(k):
                        for _ in range(k):
                            s.step_begin(1e-3); s.step_mid(); s.step_finish()
                    go(5); s.wait()
                    t0 = time.perf_counter(); go(steps); s.wait()
                    t = (time.perf_counter() - t0) / steps * 1e3
                    row.append(f"L={L or s.sym_info()['chunks_per_item']}{'*' if not L else ''}:{t:.3f}")
            print(f"P={parts} {proto:9s}", " ".join(row), flush=True)
