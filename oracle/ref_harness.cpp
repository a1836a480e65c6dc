// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" shim around the *unmodified* reference headers, which are
// included BY PATH from /root/reference (never copied).  Built by
// oracle/Makefile into oracle/_ref/libnbref.so with
//
//   g++ -std=c++20 -O3 -msse4.1 -ffp-contract=off -include bit -include cstdint
//       -I/root/reference/Nbodysim/headers -shared -fPIC
//
// (flags explained in oracle/README.md; no -ffast-math, so results are
// reproducible).  Three oracles are exposed:
//
//   REF-STEP    the reference's production Simulation::step()  (Barnes-Hut
//               theta=1 + collide), Simulation.hpp:67-75
//   REF-DIRECT  the reference's own Quadtree::acc() pairwise leaf loop
//               (Quadtree.hpp:133-147) driven as a direct O(N^2) sum through a
//               single-leaf tree, followed by the kick/drift lines of
//               Simulation::iterate (Simulation.hpp:129-131,160-163)
//   plus scalar probes (fast_inv_sqrt, sizeof/offsetof Body).
//
// Only used to (a) generate tests/golden/* and (b) validate oracle/nb_oracle.c
// in this container.  It does not exist on the GPU box unless the prebuilt .so
// travelled there.
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstring>
#include <future>
#include <vector>

#include "Simulation.hpp"  // pulls Body.hpp, Quadtree.hpp, Vec2.hpp (+ raylib.h for Vector2)

// The reference declares this extern (Simulation.hpp:16) and defines it in
// main.cpp:39, which we must not compile (GUI).
std::atomic<float> SIMULATION_DT{0.01f};

namespace {

// Flat, padding-free view of a Body used on the C side of the shim:
// x, y, vx, vy, ax, ay, mass, radius.
constexpr int NF = 8;

std::vector<Body> from_flat(const float *f, size_t n)
{
    std::vector<Body> b(n);
    for (size_t i = 0; i < n; ++i)
    {
        const float *p = f + i * NF;
        b[i].pos = Vec2(p[0], p[1]);
        b[i].vel = Vec2(p[2], p[3]);
        b[i].acc = Vec2(p[4], p[5]);
        b[i].mass = p[6];
        b[i].radius = p[7];
    }
    return b;
}

void to_flat(const std::vector<Body> &b, float *f)
{
    for (size_t i = 0; i < b.size(); ++i)
    {
        float *p = f + i * NF;
        p[0] = b[i].pos.x; p[1] = b[i].pos.y;
        p[2] = b[i].vel.x; p[3] = b[i].vel.y;
        p[4] = b[i].acc.x; p[5] = b[i].acc.y;
        p[6] = b[i].mass;  p[7] = b[i].radius;
    }
}

// Make the reference's tree a single leaf that owns every body, so that
// Quadtree::acc falls into its pairwise leaf loop for all j.
void make_single_leaf(Quadtree &q, const std::vector<Body> &b)
{
    q.t_sq = 0.0f;  // "size^2 < d^2 * 0" is never true -> the cell is never accepted
    q.clear(Quad::new_containing(b));
    q.nodes[0].bodies = Range(0, b.size());
}

void direct_acc(Quadtree &q, std::vector<Body> &b)
{
    make_single_leaf(q, b);
    for (size_t i = 0; i < b.size(); ++i)
        b[i].acc = q.acc(b[i].pos, b);
}

} // namespace

extern "C" {

// layout[0]=sizeof(Body) alignof, offsets pos vel acc mass radius, sizeof(Vec2)
void ref_layout(size_t *out)
{
    out[0] = sizeof(Body);
    out[1] = alignof(Body);
    out[2] = offsetof(Body, pos);
    out[3] = offsetof(Body, vel);
    out[4] = offsetof(Body, acc);
    out[5] = offsetof(Body, mass);
    out[6] = offsetof(Body, radius);
    out[7] = sizeof(Vec2);
}

// Quadtree::fast_inv_sqrt (Quadtree.hpp:106-111) on an array.
void ref_fast_inv_sqrt(const float *x, float *y, size_t n)
{
    Quadtree q(1.0f, 1.0f, 16);
    for (size_t i = 0; i < n; ++i)
        y[i] = q.fast_inv_sqrt(x[i]);
}

// REF-DIRECT single force evaluation; writes acc into the flat records.
void ref_direct_acc(float *flat, size_t n, float eps)
{
    std::vector<Body> b = from_flat(flat, n);
    Quadtree q(1.0f, eps, 16);
    direct_acc(q, b);
    to_flat(b, flat);
}

// Wall time (seconds) of the reference's own pairwise loop — Quadtree::acc through the single-leaf tree — for the
// i-particles [i_begin, i_end) against all n bodies, fanned over `threads` std::async tasks on contiguous i-chunks,
// the way Simulation::attract() fans its i-loop (Simulation.hpp:180-208).  This is the "reference's own CPU loop"
// figure of bench.py's cpu_baseline (kind "reference"); the accelerations are written back for a cross-check.
double ref_direct_acc_timed(float *flat, size_t n, float eps, size_t i_begin, size_t i_end, int threads)
{
    std::vector<Body> b = from_flat(flat, n);
    Quadtree q(1.0f, eps, 16);
    make_single_leaf(q, b);
    if (i_end > n) i_end = n;
    if (i_begin >= i_end) return 0.0;
    if (threads < 1) threads = 1;
    const size_t count = i_end - i_begin, chunk = (count + (size_t)threads - 1) / (size_t)threads;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::future<void>> tasks;
    for (size_t start = i_begin; start < i_end; start += chunk)
    {
        const size_t stop = start + chunk < i_end ? start + chunk : i_end;
        tasks.push_back(std::async(std::launch::async, [&q, &b, start, stop]() {
            for (size_t i = start; i < stop; ++i)
                b[i].acc = q.acc(b[i].pos, b);
        }));
    }
    for (auto &t : tasks) t.get();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    to_flat(b, flat);
    return secs;
}

// REF-DIRECT nsteps of kick-drift.  use_body_update=1 integrates through the
// reference's own Body::update (Body.hpp:34-38) instead of the restated
// iterate() lines; both must give identical bits (checked by make_golden.py).
void ref_direct_step(float *flat, size_t n, float eps, float dt, int nsteps, int use_body_update)
{
    std::vector<Body> b = from_flat(flat, n);
    Quadtree q(1.0f, eps, 16);
    for (int s = 0; s < nsteps; ++s)
    {
        direct_acc(q, b);
        if (use_body_update)
        {
            for (auto &p : b) p.update(dt);
        }
        else
        {
            for (size_t i = 0; i < n; ++i)  // Simulation.hpp:129-131
            {
                b[i].vel.x += b[i].acc.x * dt;
                b[i].vel.y += b[i].acc.y * dt;
            }
            for (size_t i = 0; i < n; ++i)  // Simulation.hpp:160-163
            {
                b[i].pos.x += b[i].vel.x * dt;
                b[i].pos.y += b[i].vel.y * dt;
            }
        }
    }
    to_flat(b, flat);
}

// REF-STEP: the reference's production step() on caller-supplied bodies.
// Returns the frame counter after the run.
size_t ref_step(float *flat, size_t n, float eps, float dt, int nsteps)
{
    Simulation s;                 // ctor builds its default 25 000-body ICs; replaced below
    s.bodies = from_flat(flat, n);
    s.quadtree.e_sq = eps * eps;  // public member, Quadtree.hpp:12
    SIMULATION_DT.store(dt);
    for (int k = 0; k < nsteps; ++k)
        s.step();
    to_flat(s.bodies, flat);
    return s.frame;
}

// The reference's default initial conditions (Simulation() ctor ->
// uniform_disc(25000), Simulation.hpp:58-65,347-603).  out must hold 25000*8.
size_t ref_default_ics(float *out, size_t cap)
{
    Simulation s;
    if (s.bodies.size() > cap) return s.bodies.size();
    to_flat(s.bodies, out);
    return s.bodies.size();
}

} // extern "C"
