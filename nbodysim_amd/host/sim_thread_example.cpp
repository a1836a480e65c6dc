// sim_thread_example.cpp — the reference's simulation_thread pattern
// (main.cpp:612-635) compiled against the adaptor: step(), lock, copy bodies.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "Simulation.hpp"

std::atomic<float> SIMULATION_DT{0.01f};   // main.cpp:39
std::mutex UPDATE_LOCK;                    // main.cpp:38
std::vector<Body> SHARED_BODIES;           // main.cpp:40

static void simulation_thread(std::shared_ptr<Simulation> simulation, int frames)
{
    for (int f = 0; f < frames; ++f) {
        simulation->step();
        {
            std::lock_guard<std::mutex> lock(UPDATE_LOCK);
            SHARED_BODIES = simulation->bodies;
        }
    }
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "reference") {
        // exactly what main.cpp does: default-constructed Simulation (25 000-body disc, eps = 1), dt = 0.01
        auto simulation = std::make_shared<Simulation>();
        simulation_thread(simulation, argc > 2 ? atoi(argv[2]) : 10);
        double px = 0.0, py = 0.0;
        for (const Body &b : SHARED_BODIES) { px += (double)b.mass * b.vel.x; py += (double)b.mass * b.vel.y; }
        std::printf("frame=%zu bodies=%zu body1=(%.9g, %.9g) last=(%.9g, %.9g) p=(%.9g, %.9g)\n", simulation->frame,
                    SHARED_BODIES.size(), SHARED_BODIES[1].pos.x, SHARED_BODIES[1].pos.y,
                    SHARED_BODIES.back().pos.x, SHARED_BODIES.back().pos.y, px, py);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "overlap") {
        // the caller's loop (step, lock, copy bodies) in its blocking form and with the pipelined snapshot:
        // PCIe-inclusive ms per frame of each, and the frames themselves (the pipelined one runs one frame late)
        const size_t n = argc > 2 ? (size_t)atol(argv[2]) : 262144;
        const int frames = argc > 3 ? atoi(argv[3]) : 40;
        SIMULATION_DT.store(1e-3f);
        double ms[2] = {0.0, 0.0};
        std::vector<Body> last[2];
        for (int mode = 0; mode < 2; ++mode) {
            std::vector<Body> init(n);
            if (nb_plummer_2d(reinterpret_cast<nb_body *>(init.data()), n, 42) != NB_OK) return 1;
            Simulation simulation(std::move(init), 0.01f);
            simulation.advance(3);
            simulation.sync();
            const auto t0 = std::chrono::steady_clock::now();
            for (int f = 0; f < frames + mode; ++f) {       // the pipelined loop needs one more call to deliver frame `frames`
                if (mode == 0) simulation.step(); else simulation.step_overlapped();
                std::lock_guard<std::mutex> lock(UPDATE_LOCK);
                SHARED_BODIES = simulation.bodies;
            }
            ms[mode] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (frames + mode);
            last[mode] = SHARED_BODIES;
        }
        size_t diff = 0;
        for (size_t i = 0; i < n; ++i)
            diff += last[0][i].pos.x != last[1][i].pos.x || last[0][i].pos.y != last[1][i].pos.y || last[0][i].vel.x != last[1][i].vel.x;
        std::printf("n=%zu frames=%d blocking=%.3f ms/frame overlapped=%.3f ms/frame differing_bodies=%zu\n", n, frames, ms[0], ms[1], diff);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "frames") {
        // What a maintainer who adopts the adaptor sees per frame of the reference caller's loop (main.cpp:621-627), at the
        // reference's OWN workload by default (Simulation(): 25 000 bodies, eps = 1, dt = 0.01, clamp + boundary; Simulation.hpp:58-65)
        // or on n Plummer bodies: ms per frame of
        //   step_copy        step(); lock; SHARED_BODIES = bodies      (the caller's loop, blocking nb_sync)
        //   step             step() alone                              (what the adaptor costs without the caller's own vector copy)
        //   overlapped_copy  step_overlapped(); lock; copy             (pipelined snapshot, one frame late)
        //   step_wait        nb_step(1) + nb_wait                      (one host round trip per step, nothing copied)
        //   resident         advance(frames) + one sync()              (the device's own rate)
        const bool reference = argc <= 2 || std::string(argv[2]) == "reference";
        const size_t n = reference ? 25000 : (size_t)atol(argv[2]);
        const int frames = argc > 3 ? atoi(argv[3]) : 200;
        if (!reference) SIMULATION_DT.store(1e-3f);
        //   positions        nb_step(1) + nb_sync_positions            (a viewer that draws positions only: 8 B per body instead of 64)
        const char *names[6] = {"step_copy", "step", "overlapped_copy", "step_wait", "resident", "positions"};
        double ms[6] = {0, 0, 0, 0, 0, 0};
        std::vector<float> xy(2 * n);
        for (int mode = 0; mode < 6; ++mode) {
            std::unique_ptr<Simulation> sim;
            if (reference) sim.reset(new Simulation());
            else {
                std::vector<Body> init(n);
                if (nb_plummer_2d(reinterpret_cast<nb_body *>(init.data()), n, 42) != NB_OK) return 1;
                sim.reset(new Simulation(std::move(init), 0.01f));
            }
            sim->advance(5);
            sim->sync();
            SHARED_BODIES = sim->bodies;
            const auto t0 = std::chrono::steady_clock::now();
            if (mode == 4) { sim->advance(frames); sim->sync(); }
            else for (int f = 0; f < frames; ++f) {
                if (mode == 0 || mode == 1) sim->step();
                else if (mode == 2) sim->step_overlapped();
                else if (mode == 5) { sim->advance(1); if (nb_sync_positions(sim->handle(), xy.data()) != NB_OK) return 1; }
                else { sim->advance(1); if (nb_wait(sim->handle()) != NB_OK) return 1; }
                if (mode == 0 || mode == 2) { std::lock_guard<std::mutex> lock(UPDATE_LOCK); SHARED_BODIES = sim->bodies; }
            }
            ms[mode] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
        }
        std::printf("{\"workload\": \"%s\", \"n\": %zu, \"frames\": %d, \"bytes_per_frame\": %zu", reference ? "Simulation() default start" : "plummer_2d", n, frames, n * sizeof(Body));
        for (int k = 0; k < 6; ++k) std::printf(", \"%s_ms\": %.4f", names[k], ms[k]);
        std::printf("}\n");
        return 0;
    }
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 4096;
    std::vector<Body> init(n);
    if (nb_plummer_2d(reinterpret_cast<nb_body *>(init.data()), n, 42) != NB_OK) return 1;
    SIMULATION_DT.store(1e-3f);
    auto simulation = std::make_shared<Simulation>(std::move(init), 0.05f);
    simulation_thread(simulation, 10);
    std::printf("frame=%zu bodies=%zu body0=(%.6f, %.6f)\n", simulation->frame, SHARED_BODIES.size(),
                SHARED_BODIES[0].pos.x, SHARED_BODIES[0].pos.y);
    return 0;
}
