// The env(action) loop of the reference (src/env.jl:91-121 inside src/data.jl:22-27) written against the C ABI alone
// (include/waves_amd.h): what a host in a compiled language -- the Julia shim of INTEGRATION.md -- pays per action, without
// the Python mirror's interpreter time.  A TIMING example, not a parity test: coordinates, the triple-ring design and the
// random actions are built here in plain C++ (the parity tests live in tests/ and go through the same entry points).
//
//   host_loop [grid 700] [actions 40] [in_flight 1|2] [state 0|1] [pause_us 0]
//     in_flight 1: every action is ended before the next one is begun (a policy that looks at the wave state);
//     state 1:     wv_observation (state(env), 128x128x4) in front of every action;
//     pause_us:    the host spins this long in front of every action (a policy that thinks).
//   prints one JSON line: ms per action, Mcell-updates/s, the launch-level figures of wv_get_timing.
//
// build:  make -C waves.jl_amd/csrc example      (g++, links libwaves_amd.so; no HIP headers needed on this side)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/waves_amd.h"

#define CK(call)                                                                        \
    do {                                                                                \
        const int rc_ = (call);                                                         \
        if (rc_ != 0) {                                                                 \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, wv_last_error(ctx));          \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static unsigned long long g_rng = 0x9E3779B97F4A7C15ull;
static float urand()  // [0, 1)
{
    g_rng = g_rng * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((g_rng >> 40) * (1.0 / 16777216.0));
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 700;
    const int actions = argc > 2 ? atoi(argv[2]) : 40;
    const int in_flight = argc > 3 ? atoi(argv[3]) : 1;
    const int with_state = argc > 4 ? atoi(argv[4]) : 0;
    const double pause_us = argc > 5 ? atof(argv[5]) : 0.0;
    const int steps = 100, warm = 5;
    const float dt = 1e-5f;
    wv_ctx *ctx = nullptr;

    // TwoDim(15f0, n): x = y = range(-15, 15, n)
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) x[i] = (float)(-15.0 + 30.0 * i / (n - 1));
    wv_config cfg{n, n, 1531.0f, dt, 2.0f, 20000.0f, 0, WV_IMPL_AUTO};
    CK(wv_create(&cfg, x.data(), x.data(), &ctx));
    const float mu[2] = {-10.0f, 2.5f}, sigma[1] = {0.3f}, amp[1] = {1.0f};
    CK(wv_set_gaussian_source(ctx, 1, mu, sigma, amp, 1000.0f));
    CK(wv_reset(ctx));

    // build_triple_ring_design_space (src/designs.jl:353-365): three hexagon rings around (5, 0) + the core, radii in [0.2, 1]
    const int M = 19;
    std::vector<float> pos(2 * M), c(M), r0(M), r1(M);
    const double ring_r[3] = {3.5, 4.75, 6.0}, ring_rot[3] = {0.0, M_PI / 6.0, 0.0};
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 6; ++j) {
            const double a = j * 2.0 * M_PI / 6.0 + ring_rot[k];
            pos[6 * k + j] = (float)(ring_r[k] * cos(a) + 5.0);      // column-major (M, 2): all x ...
            pos[M + 6 * k + j] = (float)(ring_r[k] * sin(a));        // ... then all y
            c[6 * k + j] = 3.0f * 344.0f;
            r0[6 * k + j] = 0.2f + 0.8f * urand();
        }
    pos[18] = 5.0f, pos[M + 18] = 0.0f, c[18] = 3.0f * 344.0f, r0[18] = 2.0f;  // the core: fixed
    const float scale = 250.0f * dt * (float)steps;  // action_space: action_speed * dt * integration_steps (src/env.jl:143-145)

    std::vector<float> tspan(steps + 1), sig(3 * (steps + 1)), obs(128 * 128 * 4);
    double checksum = 0.0;
    int step0 = 0, begun = 0, ended = 0;
    auto begin = [&]() -> int {
        if (pause_us > 0.0) {
            const auto p0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - p0).count() < pause_us) {}
        }
        if (with_state) CK(wv_observation(ctx, 128, 128, obs.data()));                 // s = state(env)
        for (int j = 0; j < 18; ++j) {                                                 // a = policy(s); design_space(design, a)
            const float a = scale * (2.0f * urand() - 1.0f);
            r1[j] = fminf(fmaxf(r0[j] + a, 0.2f), 1.0f);
        }
        r1[18] = r0[18];
        for (int s = 0; s <= steps; ++s) tspan[s] = (float)((double)(step0 + s) * (double)dt);
        CK(wv_set_design(ctx, M, pos.data(), r0.data(), c.data(), pos.data(), r1.data(), c.data(), tspan[0], tspan[steps]));
        CK(wv_integrate_begin(ctx, tspan.data(), steps, 1, 1, 0));
        r0.swap(r1);
        step0 += steps;
        ++begun;
        return 0;
    };
    auto end = [&]() -> int {
        CK(wv_integrate_end(ctx, sig.data(), nullptr, nullptr));
        checksum += sig[3 * steps + 2];
        ++ended;
        return 0;
    };
    auto run = [&](int count) -> int {
        const int target = ended + count;
        while (ended < target) {
            while (begun < target && begun - ended < in_flight)
                if (begin()) return 1;
            if (end()) return 1;
        }
        return 0;
    };
    if (run(warm)) return 1;
    CK(wv_synchronize(ctx));  // the warm-up's launch leaves: the timed region pays for its own, like bench.py's
    const auto t0 = std::chrono::steady_clock::now();
    if (run(actions)) return 1;
    CK(wv_synchronize(ctx));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    wv_timing t{};
    CK(wv_get_timing(ctx, &t));
    const double per = ms / actions, cells = (double)n * n * steps;
    printf("{\"host\": \"C++ over the C ABI (examples/host_loop.cpp)\", \"grid\": %d, \"actions\": %d, \"in_flight\": %d, \"state_before_every_action\": %d, "
           "\"ms_per_action\": %.4f, \"Mcell_updates_per_s\": %.1f, \"whole_job_frac_of_8TBs_at_104B\": %.4f, \"resident\": %d, \"gave_up\": %d, "
           "\"last_launch_ms\": %.4f, \"last_launch_jobs\": %d, \"signal_checksum\": %.6g}\n",
           n, actions, in_flight, with_state, per, cells / per / 1e3, 104.0 * cells / (per * 1e-3) / 8e12, t.resident, t.gave_up, t.launch_ms, t.launch_jobs,
           checksum);
    CK(wv_destroy(ctx));
    return 0;
}
