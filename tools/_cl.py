import sys, time
sys.path.insert(0, '/root/repo')
import nbodysim_amd as nb
def run(ic, kb, steps, **kw):
    with nb.Simulation(ic, eps=0.01, **kb, **kw) as s:
        s.advance(20, 1e-3); s.wait()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); s.advance(steps, 1e-3); s.wait(); best = min(best, (time.perf_counter() - t0) / steps)
        info = s.sym_info()
    return best * 1e6, info
for rnd in range(2):
    for name, mk, kb in (("3-D fp32", nb.plummer_3d, dict(dims=3)), ("2-D fp64", nb.plummer_2d, dict(precision="fp64")), ("3-D fp64", nb.plummer_3d, dict(dims=3, precision="fp64"))):
        for n in (16384, 24576, 32768, 49152):
            ic = mk(n, 42)
            base, info = run(ic, kb, 200)
            row = [f"as built L={info['chunks_per_item']} items={info['items']:4d} {base:7.1f}"]
            for L in (1, 2, 3, 4, 6):
                for tail, label in ((None, "none"), ((0.85, 0.94, 0.98), "late")):
                    kw = dict(sym_chunks_per_item=L)
                    if tail is None: kw["guided_tail"] = False
                    else: kw["sym_tail"] = tail
                    us, info = run(ic, kb, 200, **kw)
                    row.append(f"L{L} {label} {info['items']:4d} {us:7.1f}")
            print(f"round {rnd+1} {name} n={n:6d} | " + " | ".join(row), flush=True)
