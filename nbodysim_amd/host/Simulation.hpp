// Simulation.hpp — C++ adaptor with the reference's `Simulation` surface over the C ABI.
//
// Reference surface being replaced (Nbodysim/headers/Simulation.hpp:49-75):
//   class Simulation { public: float dt; size_t frame; std::vector<Body> bodies; Quadtree quadtree;
//                      Simulation(); void step(); }
//   extern std::atomic<float> SIMULATION_DT;                     Simulation.hpp:16
//
// HOW IT DROPS IN.  Put this directory BEFORE the reference's `headers/` on the include path:
//     g++ -std=c++20 -I nbodysim_amd/host -I include -I Nbodysim/headers  Nbodysim/source/main.cpp ...
// `#include "Simulation.hpp"` (main.cpp:22) then finds this file, and this file uses the reference's OWN
// `Vec2.hpp`, `Body.hpp` and `Node.hpp` (found further down the same include path), so `Vec2` keeps its
// operators and `mag_sq` (main.cpp:108-154), `#include "Vec2.hpp"` (main.cpp:23) is a harmless repeat,
// `std::vector<Node> SHARED_QUADTREE` (main.cpp:41) has its `Node`, and `simulation->quadtree.nodes`
// (main.cpp:626) is this adaptor's always-empty node list (the force is a direct sum: there is no tree to
// draw).  The UNMODIFIED main.cpp compiles against this header: tests/test_dropin_compile.py runs
// `g++ -std=c++20 -fsyntax-only` on it.  Without the reference's headers on the path (stand-alone use, e.g.
// sim_thread_example.cpp) minimal layout-identical `Vec2` / `Body` / `Node` are declared here instead.
//
// `Simulation()` starts, like the reference's, from uniform_disc(25000) with epsilon = 1 and the velocity
// clamp + soft boundary of iterate() switched on; a second constructor takes any initial bodies.
// Differences, all deliberate: the force is the direct O(N^2) sum, not Barnes-Hut, and
// collide() (Simulation.hpp:72,216-346) is NOT run — bodies with radius > 0 pass through each other, so a run
// from `Simulation()` follows the reference's gravity + clamp + boundary, not its collisions.
#pragma once
#include <atomic>
#include <cmath>
#include <cstddef>
#include <future>
#include <mutex>
#include <numbers>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "nbody.h"

#if defined(__has_include)
#if __has_include("Vec2.hpp") && __has_include("Body.hpp") && __has_include("Node.hpp")
#define NB_ADAPTOR_REFERENCE_TYPES 1
#endif
#endif

#ifdef NB_ADAPTOR_REFERENCE_TYPES
// the reference's own types (Vec2.hpp:17-357, Body.hpp:6-107, Node.hpp:10-53), by include path — never copied
#include "Body.hpp"
#include "Node.hpp"
#include "Vec2.hpp"
#else
struct alignas(16) Vec2 {                                   // Vec2.hpp:17-20
    float x, y;
    Vec2() noexcept = default;
    constexpr Vec2(float x_, float y_) noexcept : x(x_), y(y_) {}
    static constexpr Vec2 zero() noexcept { return Vec2(0.0f, 0.0f); }
};

struct alignas(16) Body {                                   // Body.hpp:6-17
    Vec2 pos, vel, acc;
    float mass, radius;
    Body() = default;
    Body(Vec2 p, Vec2 v, float m, float r) : pos(p), vel(v), acc(Vec2::zero()), mass(m), radius(r) {}
};

struct Node {};                                             // tree node: nothing to show for a direct sum
#endif

extern std::atomic<float> SIMULATION_DT;  // defined by the application, as in main.cpp:39

static_assert(sizeof(Vec2) == sizeof(nb_vec2) && alignof(Vec2) == 16, "Vec2 layout must match the C ABI (Vec2.hpp:17)");
static_assert(sizeof(Body) == sizeof(nb_body) && offsetof(Body, vel) == offsetof(nb_body, vel) &&
              offsetof(Body, acc) == offsetof(nb_body, acc) && offsetof(Body, mass) == offsetof(nb_body, mass) &&
              offsetof(Body, radius) == offsetof(nb_body, radius), "Body layout must match the C ABI (Body.hpp:6)");

// Stand-in for the reference's `Quadtree quadtree` member (Simulation.hpp:55): the public fields its caller and a
// harness touch.  `nodes` stays empty — main.cpp:626 copies it, main.cpp:737 draws nothing for an empty list.
struct DirectSumQuadtree {
    float t_sq = 1.0f;      // Quadtree.hpp:11: theta^2 — unused by a direct sum
    float e_sq = 1.0f;      // Quadtree.hpp:12: epsilon^2 — the softening this handle was created with
    std::vector<Node> nodes;
};

class Simulation {
public:
    float dt = 0.0f;            // unused by the reference too (Simulation.hpp:52)
    size_t frame = 0;
    std::vector<Body> bodies;
    DirectSumQuadtree quadtree;

    // Simulation.hpp:58-65 — the reference's own start (its n, epsilon, ICs; iterate()'s extras on).
    Simulation() : Simulation(uniform_disc(25000), 1.0f, reference_params()) {}

    // Simulation.hpp:347-603, bit-identical bodies (nb_default_ics).
    static std::vector<Body> uniform_disc(size_t n)
    {
        std::vector<Body> b(n);
        check(nb_default_ics(reinterpret_cast<nb_body *>(b.data()), n), "nb_default_ics");
        return b;
    }

    explicit Simulation(std::vector<Body> initial, float epsilon = 1.0f, const nb_params *overrides = nullptr)
        : bodies(std::move(initial))
    {
        nb_params p;
        if (overrides) p = *overrides; else nb_params_default(&p);
        p.eps = epsilon;
        p.dt = SIMULATION_DT.load();
        quadtree.e_sq = epsilon * epsilon;
        sim_ = nb_create(reinterpret_cast<const nb_body *>(bodies.data()), bodies.size(), &p);
        if (!sim_) throw std::runtime_error(std::string("nb_create: ") + nb_last_error());
    }
    ~Simulation()
    {
        if (snapshot_in_flight_) (void)nb_snapshot_wait(sim_);
        nb_destroy(sim_);
    }
    Simulation(const Simulation &) = delete;
    Simulation &operator=(const Simulation &) = delete;

    // HOST MEMORY.  `bodies` is an ordinary std::vector, as in the reference (the caller copies it by value,
    // main.cpp:625); its storage is the heap's, may move whenever the caller assigns or swaps the vector, and shares
    // its first and last page with other heap data.  The adaptor therefore never page-locks it (no hipHostRegister of
    // memory it does not own page-wise, nothing to unregister after a reallocation): nb_sync / nb_snapshot_wait fill it
    // from the library's own page-locked staging buffer, the host copy pipelined against the DMA.

    // Simulation.hpp:67-75 — on return `bodies` is coherent (pos, vel, acc, mass, radius).
    // `bodies` is public in the reference and stays public here, but the device state has a fixed size: a caller
    // that resized the vector gets an exception, not an out-of-bounds write (host edits go through upload()).
    void step()
    {
        check_size();
        if (snapshot_in_flight_) { check(nb_snapshot_wait(sim_), "nb_snapshot_wait"); snapshot_in_flight_ = false; }
        const float current_dt = SIMULATION_DT.load();
        check(nb_step(sim_, current_dt, 1), "nb_step");
        check(nb_sync(sim_, reinterpret_cast<nb_body *>(bodies.data())), "nb_sync");
        ++frame;
    }

    // Pipelined form of step() for a viewer that can draw one frame late: advances one step and returns with `bodies`
    // holding the state after the PREVIOUS call's step.  The device-to-host copy of frame k (16.8 MB at N = 262 144)
    // runs on a copy stream into the library's page-locked staging buffer while the force of step k + 1 computes
    // (nb_snapshot_begin); nb_snapshot_wait then moves it into a PRIVATE vector — also while step k + 1 is running,
    // because that step was enqueued first — which is swapped with `bodies` (O(1)).  The destination of a snapshot in
    // flight is never the public vector, whose storage the caller may move at any time.  Call sync() to catch up.
    void step_overlapped()
    {
        check_size();
        if (back_.size() != bodies.size()) {
            if (snapshot_in_flight_) { check(nb_snapshot_wait(sim_), "nb_snapshot_wait"); snapshot_in_flight_ = false; }
            back_.assign(bodies.size(), Body());
        }
        check(nb_step(sim_, SIMULATION_DT.load(), 1), "nb_step");                                   // enqueue step k + 1
        if (snapshot_in_flight_) {
            check(nb_snapshot_wait(sim_), "nb_snapshot_wait");                                      // frame k -> back_ (GPU busy with k + 1)
            bodies.swap(back_);
        }
        check(nb_snapshot_begin(sim_, reinterpret_cast<nb_body *>(back_.data())), "nb_snapshot_begin");   // frame k + 1 follows it
        snapshot_in_flight_ = true;
        ++frame;
    }

    // Throughput form: k steps without refreshing `bodies` (call sync() when a snapshot is wanted).
    void advance(int k)
    {
        check(nb_step(sim_, SIMULATION_DT.load(), k), "nb_step");
        frame += (size_t)k;
    }
    void sync()
    {
        check_size();
        if (snapshot_in_flight_) { check(nb_snapshot_wait(sim_), "nb_snapshot_wait"); snapshot_in_flight_ = false; }
        check(nb_sync(sim_, reinterpret_cast<nb_body *>(bodies.data())), "nb_sync");
    }
    // After editing `bodies` in place on the host (same count).
    void upload() { check_size(); check(nb_upload(sim_, reinterpret_cast<const nb_body *>(bodies.data())), "nb_upload"); }
    nb_sim *handle() { return sim_; }

private:
    static const nb_params *reference_params()
    {
        static nb_params p;
        nb_params_default(&p);
        p.extras = NB_EXTRA_VCLAMP | NB_EXTRA_BOUNDARY;     // Simulation.hpp:133-155
        return &p;
    }
    static void check(int rc, const char *what)
    {
        if (rc != NB_OK) throw std::runtime_error(std::string(what) + ": " + nb_last_error());
    }
    void check_size() const
    {
        if (bodies.size() != nb_count(sim_))
            throw std::length_error("Simulation::bodies was resized (" + std::to_string(bodies.size()) + " != " +
                                    std::to_string(nb_count(sim_)) + "): the device state has a fixed body count");
    }
    nb_sim *sim_ = nullptr;
    std::vector<Body> back_;            // step_overlapped(): private destination of the snapshot in flight
    bool snapshot_in_flight_ = false;
};
