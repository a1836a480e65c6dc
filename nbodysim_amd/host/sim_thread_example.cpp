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
