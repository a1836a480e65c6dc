// Simulation.hpp — C++ adaptor with the reference's `Simulation` surface over the C ABI.
//
// Reference surface being mirrored (Nbodysim/headers):
//   struct alignas(16) Vec2 { float x, y; }                      Vec2.hpp:17-20
//   struct alignas(16) Body { Vec2 pos, vel, acc; float mass, radius; }   Body.hpp:6-13
//   class Simulation { public: float dt; size_t frame; std::vector<Body> bodies;
//                      Simulation(); void step(); }              Simulation.hpp:49-75
//   extern std::atomic<float> SIMULATION_DT;                     Simulation.hpp:16
//
// A caller written against the reference — `simulation->step();` then copying
// `simulation->bodies` (main.cpp:621-627) — compiles against this header
// unchanged, with the O(N^2) force + kick/drift running on the MI355X.
// `Simulation()` starts, like the reference's, from uniform_disc(25000) with
// epsilon = 1 and the velocity clamp + soft boundary of iterate() switched on;
// a second constructor takes any initial bodies.  Differences, all deliberate
// (DESIGN.md §boundary): there is no `quadtree` member (direct sum instead of
// Barnes-Hut) and collide() is not run (not gravity).
#pragma once
#include <atomic>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

#include "nbody.h"

extern std::atomic<float> SIMULATION_DT;  // defined by the application, as in main.cpp:39

struct alignas(16) Vec2 {
    float x, y;
    Vec2() noexcept = default;
    constexpr Vec2(float x_, float y_) noexcept : x(x_), y(y_) {}
    static constexpr Vec2 zero() noexcept { return Vec2(0.0f, 0.0f); }
};

struct alignas(16) Body {
    Vec2 pos, vel, acc;
    float mass, radius;
    Body() = default;
    Body(Vec2 p, Vec2 v, float m, float r) : pos(p), vel(v), acc(Vec2::zero()), mass(m), radius(r) {}
};

static_assert(sizeof(Vec2) == sizeof(nb_vec2) && sizeof(Body) == sizeof(nb_body), "layout must match the C ABI");

class Simulation {
public:
    float dt = 0.0f;            // unused by the reference too (Simulation.hpp:52)
    size_t frame = 0;
    std::vector<Body> bodies;

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
        sim_ = nb_create(reinterpret_cast<const nb_body *>(bodies.data()), bodies.size(), &p);
        if (!sim_) throw std::runtime_error(std::string("nb_create: ") + nb_last_error());
        pin();
    }
    ~Simulation() { unpin(); nb_destroy(sim_); }
    Simulation(const Simulation &) = delete;
    Simulation &operator=(const Simulation &) = delete;

    // Simulation.hpp:67-75 — on return `bodies` is coherent (pos, vel, acc, mass, radius).
    void step()
    {
        if (bodies.data() != pinned_) pin();   // the caller resized / replaced the vector
        const float current_dt = SIMULATION_DT.load();
        check(nb_step(sim_, current_dt, 1), "nb_step");
        check(nb_sync(sim_, reinterpret_cast<nb_body *>(bodies.data())), "nb_sync");
        ++frame;
    }

    // Throughput form: k steps without refreshing `bodies` (call sync() when a snapshot is wanted).
    void advance(int k)
    {
        check(nb_step(sim_, SIMULATION_DT.load(), k), "nb_step");
        frame += (size_t)k;
    }
    void sync() { check(nb_sync(sim_, reinterpret_cast<nb_body *>(bodies.data())), "nb_sync"); }
    // After editing `bodies` on the host (e.g. the GUI's SPAWN_QUEUE, main.cpp:43).
    void upload() { check(nb_upload(sim_, reinterpret_cast<const nb_body *>(bodies.data())), "nb_upload"); }
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
    // page-lock the vector's storage so nb_sync DMAs straight into it
    void pin()
    {
        unpin();
        if (!bodies.empty() && nb_host_register(bodies.data(), bodies.size() * sizeof(Body)) == NB_OK) pinned_ = bodies.data();
    }
    void unpin()
    {
        if (pinned_) { nb_host_unregister(pinned_); pinned_ = nullptr; }
    }
    nb_sim *sim_ = nullptr;
    Body *pinned_ = nullptr;
};
