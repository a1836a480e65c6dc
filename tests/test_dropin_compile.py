"""CPU (container only: needs /root/reference): the drop-in claim of SURVEY §8(b), checked by the compiler.

The reference's one caller of the hot path is Nbodysim/source/main.cpp (simulation_thread, :612-635; it also
declares `std::vector<Node> SHARED_QUADTREE` :41, copies `simulation->quadtree.nodes` :626 and uses Vec2's
operators :108-154).  With nbodysim_amd/host FIRST on the include path, `#include "Simulation.hpp"` (:22) resolves
to the adaptor, which pulls the reference's own Vec2.hpp / Body.hpp / Node.hpp from further down the path.  The
UNMODIFIED main.cpp must then pass `g++ -std=c++20 -fsyntax-only` (no raylib library is needed for that: its
headers are vendored in the reference tree).  Nothing of the reference is copied into the repo."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/Nbodysim")
pytestmark = pytest.mark.skipif(not (REF / "source" / "main.cpp").exists(), reason="the reference tree is not present on this box")

INC = ["-I", str(ROOT / "nbodysim_amd" / "host"), "-I", str(ROOT / "include"), "-I", str(REF / "headers")]
# -msse4.1: the reference's Vec2.hpp uses _mm_dp_ps on x86-64 (Vec2.hpp:207); its own build is arm64/NEON
FLAGS = ["-std=c++20", "-msse4.1", "-w"]


def test_unmodified_reference_main_cpp_compiles_against_the_adaptor():
    src = REF / "source" / "main.cpp"
    r = subprocess.run(["g++", *FLAGS, "-fsyntax-only", *INC, str(src)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    deps = subprocess.run(["g++", *FLAGS, "-M", *INC, str(src)], capture_output=True, text=True, timeout=300).stdout.split()
    used = {Path(d).name: d for d in deps if d.endswith((".hpp", ".h"))}
    assert used["Simulation.hpp"] == str(ROOT / "nbodysim_amd" / "host" / "Simulation.hpp")      # the adaptor, not the reference's
    assert used["nbody.h"] == str(ROOT / "include" / "nbody.h")
    for own in ("Vec2.hpp", "Body.hpp", "Node.hpp"):                                            # the reference's own types
        assert used[own] == str(REF / "headers" / own)
    assert "Quadtree.hpp" not in used                                                            # no tree: direct sum


def test_adaptor_surface_matches_what_the_caller_uses(tmp_path):
    """The members main.cpp touches have the reference's types; links against the product library; without a GPU the
    constructor throws (there is no CPU path) instead of computing."""
    probe = tmp_path / "probe.cpp"
    probe.write_text(r'''
#include "raylib.h"
#include "Simulation.hpp"
#include "Vec2.hpp"
#include <cstdio>
#include <memory>
#include <type_traits>
std::atomic<float> SIMULATION_DT{0.01f};                      // main.cpp:39
std::vector<Body> SHARED_BODIES;                              // main.cpp:40
std::vector<Node> SHARED_QUADTREE;                            // main.cpp:41
static_assert(std::is_same_v<decltype(Simulation::bodies), std::vector<Body>>);
static_assert(std::is_same_v<decltype(std::declval<Simulation>().quadtree.nodes), std::vector<Node>>);
static_assert(std::is_same_v<decltype(Simulation::frame), size_t>);
static_assert(sizeof(Body) == 64 && sizeof(Vec2) == 16);
int main() {
    Vec2 a(3.0f, 4.0f), b(1.0f, 1.0f);
    const Vec2 c = (a - b) * 2.0f;                            // the reference's own operators (main.cpp:108-154)
    std::printf("mag_sq=%g c=(%g,%g)\n", a.mag_sq(), c.x, c.y);
    try {
        auto simulation = std::make_shared<Simulation>();     // main.cpp:657
        simulation->step();                                   // main.cpp:621
        SHARED_BODIES = simulation->bodies;                   // main.cpp:625
        SHARED_QUADTREE = simulation->quadtree.nodes;         // main.cpp:626
        std::printf("stepped frame=%zu bodies=%zu nodes=%zu\n", simulation->frame, SHARED_BODIES.size(), SHARED_QUADTREE.size());
    } catch (const std::exception &e) {
        std::printf("refused: %s\n", e.what());
    }
    return 0;
}
''')
    exe = tmp_path / "probe"
    lib = ROOT / "nbodysim_amd"
    r = subprocess.run(["g++", *FLAGS, "-O1", *INC, str(probe), "-o", str(exe), f"-L{lib}", "-lnbody_hip",
                        f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-pthread"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mag_sq=25 c=(4,6)" in out.stdout
    assert ("refused: nb_create: nb_create: no HIP device" in out.stdout) or ("stepped frame=1 bodies=25000 nodes=0" in out.stdout), out.stdout


def test_adaptor_stands_alone_without_the_reference_headers(tmp_path):
    """Without the reference on the include path the adaptor declares layout-identical minimal types."""
    src = ROOT / "nbodysim_amd" / "host" / "sim_thread_example.cpp"
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", str(ROOT / "include"), str(src)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
