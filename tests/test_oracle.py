"""CPU: the oracle (oracle/nb_oracle.c) against the golden vectors the compiled
reference produced, and — where oracle/_ref/libnbref.so exists (build container)
— against the live reference.  Bit-exact for the reference-arithmetic (quake)
mode; documented envelopes for the others."""
import json

import numpy as np
import pytest

from conftest import GOLD, ROOT, max_rel

EPS, DT = 0.05, 1e-3


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_manifest_hashes():
    import hashlib
    man = json.loads((GOLD / "manifest.json").read_text())
    for name, h in man["sha256"].items():
        assert hashlib.sha256((GOLD / name).read_bytes()).hexdigest() == h, name


def test_layout_fixture():
    lay = json.loads((GOLD / "layout.json").read_text())
    assert lay == {"sizeof_Body": 64, "alignof_Body": 16, "off_pos": 0, "off_vel": 16, "off_acc": 32,
                   "off_mass": 48, "off_radius": 52, "sizeof_Vec2": 16}


def test_fast_inv_sqrt_bit_exact(gold, nbo):
    y = nbo.fast_inv_sqrt(gold["fast_inv_sqrt_x"])
    g = gold["fast_inv_sqrt_y"]
    same = (bits(y) == bits(g)) | (np.isnan(y) & np.isnan(g))
    assert same.all()
    # the published envelope of the Quake approximation: rel err in [-1.752e-3, +1.3e-7]
    x = gold["fast_inv_sqrt_x"].astype(np.float64)
    ok = (x >= 1.0) & (x < 4.0)
    rel = y[ok].astype(np.float64) * np.sqrt(x[ok]) - 1.0
    assert rel.min() > -1.76e-3 and rel.max() < 2e-7


@pytest.mark.parametrize("name,n,eps", [("ref_direct_acc_1024", 1024, EPS), ("ref_direct_acc_4096", 4096, EPS),
                                        ("ref_direct_acc_eps1_1024", 1024, 1.0)])
def test_accel_quake_bit_exact(gold, nbo, name, n, eps):
    st = nbo.state_from_flat(gold[f"ic_plummer_{n}"])
    ax, ay = nbo.accel_f32(st, eps, nbo.RSQRT_QUAKE)
    assert np.array_equal(bits(ax), bits(gold[name][:, 0]))
    assert np.array_equal(bits(ay), bits(gold[name][:, 1]))


@pytest.mark.parametrize("steps", [1, 10, 100])
def test_ref_direct_steps_bit_exact(gold, nbo, steps):
    st = nbo.state_from_flat(gold["ic_plummer_1024"])
    nbo.step_f32(st, EPS, DT, steps, nbo.RSQRT_QUAKE)
    got = nbo.state_to_flat(st)
    want = gold[f"ref_direct_s{steps}"]
    assert np.array_equal(bits(got), bits(want))


def test_thread_count_does_not_change_bits(gold, nbo):
    st = nbo.state_from_flat(gold["ic_plummer_4096"])
    nbo.set_threads(1)
    a1 = nbo.accel_f32(st, EPS, nbo.RSQRT_QUAKE)
    nbo.set_threads(3)
    a3 = nbo.accel_f32(st, EPS, nbo.RSQRT_QUAKE)
    nbo.set_threads(0)
    assert np.array_equal(bits(a1[0]), bits(a3[0])) and np.array_equal(bits(a1[1]), bits(a3[1]))


def test_i_range_slices_agree(gold, nbo):
    st = nbo.state_from_flat(gold["ic_plummer_1024"])
    full = nbo.accel_f32(st, EPS, nbo.RSQRT_EXACT)
    part = nbo.accel_f32(st, EPS, nbo.RSQRT_EXACT, 100, 777)
    assert np.array_equal(bits(full[0][100:777]), bits(part[0][100:777]))
    assert not part[0][:100].any() and not part[0][777:].any()


def test_envelopes_vs_fp64_direct(gold, nbo):
    """The three distances of SURVEY §8c after 100 steps (N=1024, eps .05, dt 1e-3)."""
    ic = gold["ic_plummer_1024"]
    d = nbo.step_f64(nbo.state_from_flat(ic, np.float64), EPS, DT, 100)
    e = nbo.step_f32(nbo.state_from_flat(ic), EPS, DT, 100, nbo.RSQRT_EXACT)
    pos64 = np.stack([d["x"], d["y"]], 1)
    vel64 = np.stack([d["vx"], d["vy"]], 1)
    # fp32 exact-rsqrt direct sits ~2e-6 from fp64: the 1e-5 target is attainable
    assert max_rel(np.stack([e["x"], e["y"]], 1), pos64) < 1e-5
    assert max_rel(np.stack([e["vx"], e["vy"]], 1), vel64) < 1e-5
    # the reference's own arithmetic (Quake) is ~2e-4 / 7e-3 away — outside 1e-5, as documented
    q = gold["ref_direct_s100"]
    assert 1e-5 < max_rel(q[:, 0:2], pos64) < 1e-3
    assert 1e-4 < max_rel(q[:, 2:4], vel64) < 5e-2
    # and the production Barnes-Hut step is ~1e-2 away
    s = gold["ref_step_s100"]
    assert 1e-3 < max_rel(s[:, 0:2], pos64) < 1e-1


def test_energy_matches_numpy(gold, nbo):
    ic = gold["ic_plummer_1024"].astype(np.float64)
    st = nbo.state_from_flat(ic, np.float64)
    k, u = nbo.energy(st, EPS)
    x, y, m = ic[:, 0], ic[:, 1], ic[:, 6]
    dx, dy = x[:, None] - x[None, :], y[:, None] - y[None, :]
    inv = 1.0 / np.sqrt(dx * dx + dy * dy + EPS * EPS)
    np.fill_diagonal(inv, 0.0)
    u_np = -0.5 * np.sum(m[:, None] * m[None, :] * inv)
    k_np = 0.5 * np.sum(m * (ic[:, 2] ** 2 + ic[:, 3] ** 2))
    assert abs(k - k_np) < 1e-13 * abs(k_np) + 1e-15
    assert abs(u - u_np) < 1e-12 * abs(u_np)


def test_extras_inactive_on_unit_scale(gold, nbo):
    """Velocity clamp / soft boundary (Simulation.hpp:133-155) do nothing at unit scale."""
    a = nbo.step_f32(nbo.state_from_flat(gold["ic_plummer_1024"]), EPS, DT, 5, nbo.RSQRT_QUAKE, 0)
    b = nbo.step_f32(nbo.state_from_flat(gold["ic_plummer_1024"]), EPS, DT, 5, nbo.RSQRT_QUAKE, 1)
    assert np.array_equal(bits(nbo.state_to_flat(a)), bits(nbo.state_to_flat(b)))


def test_extras_active_far_out(nbo):
    n = 8
    st = {k: np.zeros(n, np.float32) for k in nbo.FIELDS}
    st["x"][:] = np.linspace(7e4, 1.2e5, n)
    st["vx"][:] = 2000.0
    st["m"][:] = 1.0
    before = st["vx"].copy()
    nbo.lib().nbo_kick_drift_f32(n, st["x"], st["y"], st["vx"], st["vy"], st["ax"], st["ay"], 0.01, 1)
    assert (np.abs(st["vx"]) <= 1000.0 + 1e-3).all() and (st["vx"] < before).all()
    # beyond the soft boundary (80 000) the damping also applied
    assert st["vx"][-1] < 1000.0 * 0.9996


# ---- live reference (build container only) -----------------------------------
def test_live_reference_agrees_with_fixtures(gold, nbo):
    if not nbo.have_ref():
        pytest.skip("oracle/_ref/libnbref.so not present (reference cannot travel)")
    ref = nbo.ref()
    assert nbo.ref_layout()["sizeof_Body"] == 64
    a = np.ascontiguousarray(gold["ic_plummer_1024"].copy())
    ref.ref_direct_step(a, a.shape[0], EPS, DT, 10, 0)
    assert np.array_equal(bits(a), bits(gold["ref_direct_s10"]))
    # random (non-Plummer) inputs: restatement == reference, bit for bit
    rng = np.random.default_rng(5)
    f = np.zeros((333, 8), np.float32)
    f[:, 0:2] = rng.normal(size=(333, 2)) * 3
    f[:, 2:4] = rng.normal(size=(333, 2))
    f[:, 6] = rng.uniform(0.1, 2.0, 333)
    f[7, 0:2] = f[8, 0:2]  # a coincident pair exercises the r_sq > 0 guard
    g = np.ascontiguousarray(f.copy())
    ref.ref_direct_step(g, g.shape[0], 0.3, 0.01, 7, 0)
    st = nbo.step_f32(nbo.state_from_flat(f), 0.3, 0.01, 7, nbo.RSQRT_QUAKE)
    assert np.array_equal(bits(nbo.state_to_flat(st)), bits(g))
    # eps = 0 keeps the guard meaningful
    g0 = np.ascontiguousarray(f.copy())
    ref.ref_direct_acc(g0, g0.shape[0], 0.0)
    ax, ay = nbo.accel_f32(nbo.state_from_flat(f), 0.0, nbo.RSQRT_QUAKE)
    assert np.array_equal(bits(ax), bits(g0[:, 4])) and np.array_equal(bits(ay), bits(g0[:, 5]))


@pytest.mark.parametrize("steps", [1, 4])
def test_extras_bit_exact_vs_real_reference_step(gold, nbo, steps):
    """Velocity clamp and soft boundary (Simulation.hpp:133-155) against the reference's own step():
    massless bodies feel no force, so step() reduces to kick(0) + clamp + boundary + drift."""
    st = nbo.state_from_flat(gold["ic_extras_512"])
    for _ in range(steps):
        nbo.lib().nbo_kick_drift_f32(512, st["x"], st["y"], st["vx"], st["vy"], st["ax"], st["ay"], 0.01, 1)
    got = nbo.state_to_flat(st)
    want = gold[f"ref_step_extras_s{steps}"]
    assert np.array_equal(bits(got[:, 0:4]), bits(want[:, 0:4]))
    # the fixture really exercises both branches
    ic = gold["ic_extras_512"]
    assert (np.hypot(ic[:, 2], ic[:, 3]) > 1000).sum() > 50 and (np.hypot(ic[:, 0], ic[:, 1]) > 8e4).sum() > 100


def test_reference_terms_summed_in_double_against_numpy(nbo):
    """nbo.accel_f32_terms_acc64 (the reference's fp32 per-pair terms, Quadtree.hpp:136-143, accumulated in double — used
    at N = 262 144 to separate a kernel's error from the reference's own fp32 summation noise): the same terms built
    with numpy float32 arithmetic and summed in float64, value by value; and the reference's fp32 running sum stays
    within its expected rounding of it."""
    import nbodysim_amd as nb
    n = 700
    ic = nb.plummer_2d(n, 9)
    ic["pos"][3] = ic["pos"][4]                                   # a coincident pair: the r_sq > 0 guard
    st = nbo.state_from_bodies(ic)
    eps2 = np.float32(0.05) * np.float32(0.05)
    x, y, m = st["x"], st["y"], st["m"]
    for mode in (nbo.RSQRT_QUAKE, nbo.RSQRT_EXACT):
        ax, ay = nbo.accel_f32_terms_acc64(st, 0.05, mode)
        rx = (x[None, :] - x[:, None]).astype(np.float32)
        ry = (y[None, :] - y[:, None]).astype(np.float32)
        r_sq = (rx * rx + ry * ry).astype(np.float32)
        t = (r_sq + eps2).astype(np.float32)
        if mode == nbo.RSQRT_QUAKE:
            inv = nbo.fast_inv_sqrt(t.reshape(-1)).reshape(n, n)
        else:
            inv = (np.float32(1.0) / np.sqrt(t)).astype(np.float32)
        inv3 = ((inv * inv).astype(np.float32) * inv).astype(np.float32)
        s = (m[None, :] * inv3).astype(np.float32)
        cx = np.where(r_sq > 0, (rx * s).astype(np.float32), np.float32(0)).astype(np.float64)
        cy = np.where(r_sq > 0, (ry * s).astype(np.float32), np.float32(0)).astype(np.float64)
        want_x, want_y = cx.sum(1), cy.sum(1)
        assert np.max(np.abs(ax - want_x)) <= 1e-12 * np.max(np.abs(want_x)) and np.max(np.abs(ay - want_y)) <= 1e-12 * np.max(np.abs(want_y))
        bx, by = nbo.accel_f32(st, 0.05, mode)
        assert np.max(np.abs(bx - ax)) < 1e-5 * np.max(np.abs(ax))


def test_device_expf_algorithm_is_libms_bit_for_bit(tmp_path):
    """The soft boundary of Simulation::iterate calls std::exp on a float (Simulation.hpp:147) = glibc's expf.  The device restates
    glibc's algorithm (expf_libm in nbodysim_amd/csrc/nb_kernels.hip.h: table of 2^(i/32), cubic in double, one rounding to float).
    Here the SAME statements and the SAME table — extracted from that header, so the pin cannot drift from the product — are compiled
    by gcc, with and without FMA contraction of the polynomial, and compared with this machine's libm over 2.5e7 floats covering the
    whole finite range (strided; the full 2.4e8-value sweep ran once in round 6: 0 mismatches either way)."""
    import re
    import subprocess
    hdr = (ROOT / "nbodysim_amd" / "csrc" / "nb_kernels.hip.h").read_text()
    tab = re.search(r"__device__ static const uint64_t EXPF_TAB\[32\] = \{(.*?)\};", hdr, re.S).group(1)
    body = re.search(r"__device__ __forceinline__ float expf_libm\(float x\)\n\{(.*?)\n\}", hdr, re.S).group(1)
    body = (body.replace("#pragma clang fp contract(off)", "").replace("__builtin_inff()", "INFINITY")
                .replace("(uint64_t)__double_as_longlong(kd)", "asu64(kd)").replace("__longlong_as_double((long long)t)", "asd(t)")
                .replace("__builtin_fma", "MADD"))
    src = """
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static const uint64_t EXPF_TAB[32] = {%s};
static inline uint64_t asu64(double d){uint64_t u;memcpy(&u,&d,8);return u;}
static inline double asd(uint64_t u){double d;memcpy(&d,&u,8);return d;}
#ifdef USE_FMA
#define MADD(a,b,c) fma((a),(b),(c))
#else
#define MADD(a,b,c) ((a)*(b)+(c))
#endif
static float expf_port(float x)
{%s
}
int main(void)
{
    unsigned long long bad = 0, n = 0;
    for (uint32_t u = 0; u < 0xff800000u; u += 173) {            /* both signs, up to the infinities */
        if ((u & 0x7f800000u) == 0x7f800000u) continue;          /* inf / nan patterns: not what the boundary feeds it */
        float x, a, b; memcpy(&x, &u, 4);
        a = expf(x); b = expf_port(x);
        if (memcmp(&a, &b, 4)) { if (bad < 5) printf("x=%%a libm=%%a port=%%a\\n", x, a, b); ++bad; }
        ++n;
    }
    printf("%%llu values, %%llu mismatches\\n", n, bad);
    return bad != 0;
}
""" % (tab, body)
    c = tmp_path / "expf_pin.c"
    c.write_text(src)
    for flags in ([], ["-DUSE_FMA", "-mfma"]):
        exe = tmp_path / ("pin" + ("_fma" if flags else ""))
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", *flags, "-o", str(exe), str(c), "-lm"], check=True, capture_output=True)
        r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-400:]
