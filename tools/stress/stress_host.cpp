// Stress of the resident launch's job protocol from a FAST host (C++ over the C ABI: calls follow each other within
// microseconds, which the Python mirror never does).  A random mix of calls -- one action at a time, two in flight, pipelines of
// three, state(env), synchronize, reset, the state leaving and coming back -- is run twice per seed on a fresh context: once with
// random host pauses of 0-60 us and a short idle limit of the launch, once undisturbed.  Every trace, observation and the final
// frames must be the same bytes; a context that loses the resident kernel (give-up) is a failure too.
//   stress_host [grid 320] [first_seed 0] [n_seeds 10] [ops 200] [need_resident 1] [second run on the single-step kernels 0] [two threads 0]
// build: make -C waves.jl_amd/csrc stress
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/waves_amd.h"

static thread_local unsigned long long g_rng;
static unsigned urand32()
{
    g_rng = g_rng * 6364136223846793005ull + 1442695040888963407ull;
    return (unsigned)(g_rng >> 33);
}
static float urand() { return (float)((urand32() >> 7) * (1.0 / 16777216.0)); }
static void spin(double us)
{
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < us) {}
}

#define CK(call)                                                                                 \
    do {                                                                                         \
        const int rc_ = (call);                                                                  \
        if (rc_ != 0) {                                                                          \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, wv_last_error(ctx));                   \
            return false;                                                                        \
        }                                                                                        \
    } while (0)

struct Result {
    std::vector<float> out;
    std::vector<size_t> op_at;   // out.size() at the start of every operation
    std::vector<int> op_kind, op_steps;
    int gave_up = 0, resident = 0;
};

static bool run(int n, unsigned long long seed, int ops, bool jitter, Result &res)
{
    // the pattern and the actions come from `seed` alone; the pauses from a generator of their own
    g_rng = seed * 2654435761ull + 12345ull;
    unsigned long long jit = seed ^ 0xabcdef12345ull;
    auto jrand = [&]() { jit = jit * 6364136223846793005ull + 1442695040888963407ull; return (double)(jit >> 40) / 16777216.0; };
    const int max_steps = 40, M = 19;
    int steps = 30;
    const float dt = 1e-5f;
    wv_ctx *ctx = nullptr;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) x[i] = (float)(-15.0 + 30.0 * i / (n - 1));
    wv_config cfg{n, n, 1531.0f, dt, 2.0f, 20000.0f, 0, WV_IMPL_AUTO};
    CK(wv_create(&cfg, x.data(), x.data(), &ctx));
    const float mu[2] = {-10.0f, 2.5f}, sigma[1] = {0.3f}, amp[1] = {1.0f};
    CK(wv_set_gaussian_source(ctx, 1, mu, sigma, amp, 1000.0f));
    CK(wv_reset(ctx));
    std::vector<float> pos(2 * M), c(M), r0(M), r1(M);
    const double ring_r[3] = {3.5, 4.75, 6.0}, ring_rot[3] = {0.0, M_PI / 6.0, 0.0};
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 6; ++j) {
            const double a = j * 2.0 * M_PI / 6.0 + ring_rot[k];
            pos[6 * k + j] = (float)(ring_r[k] * cos(a) + 5.0);
            pos[M + 6 * k + j] = (float)(ring_r[k] * sin(a));
            c[6 * k + j] = 3.0f * 344.0f;
            r0[6 * k + j] = 0.2f + 0.8f * urand();
        }
    pos[18] = 5.0f, pos[M + 18] = 0.0f, c[18] = 3.0f * 344.0f, r0[18] = 2.0f;
    const float scale = 250.0f * dt * 30.0f;
    std::vector<float> tspan(max_steps + 1), sig(3 * (max_steps + 1)), obs(128 * 128 * 4), state((size_t)12 * n * n);
    std::vector<float> ut((size_t)(max_steps + 1) * n * n), ui((size_t)(max_steps + 1) * n * n);
    int step0 = 0, pending = 0, fields = 0;
    std::vector<int> pend_steps, pend_fields;
    auto pause = [&]() { if (jitter) spin(60.0 * jrand()); };
    auto begin = [&]() -> bool {
        for (int j = 0; j < 18; ++j) r1[j] = fminf(fmaxf(r0[j] + scale * (2.0f * urand() - 1.0f), 0.2f), 1.0f);
        r1[18] = r0[18];
        for (int s = 0; s <= steps; ++s) tspan[s] = (float)((double)(step0 + s) * (double)dt);
        CK(wv_set_design(ctx, M, pos.data(), r0.data(), c.data(), pos.data(), r1.data(), c.data(), tspan[0], tspan[steps]));
        CK(wv_integrate_begin(ctx, tspan.data(), steps, 1, 1, fields));
        r0.swap(r1);
        step0 += steps;
        ++pending;
        pend_steps.push_back(steps);
        pend_fields.push_back(fields);
        return true;
    };
    auto end = [&]() -> bool {
        const int ns = pend_steps.front(), wf = pend_fields.front();
        pend_steps.erase(pend_steps.begin());
        pend_fields.erase(pend_fields.begin());
        if (wf == 2) {
            const float *vt = nullptr, *vi = nullptr;
            int planes = 0;
            CK(wv_integrate_end_view(ctx, sig.data(), &vt, &vi, &planes));
            res.out.push_back((float)planes);
            res.out.insert(res.out.end(), vt + (size_t)(planes - 1) * n * n, vt + (size_t)planes * n * n);
            res.out.insert(res.out.end(), vi + (size_t)(planes / 2) * n * n, vi + (size_t)(planes / 2) * n * n + n);
        } else if (wf == 1) {
            CK(wv_integrate_end(ctx, sig.data(), ut.data(), ui.data()));
            res.out.insert(res.out.end(), ut.begin() + (size_t)ns * n * n, ut.begin() + (size_t)(ns + 1) * n * n);
            res.out.insert(res.out.end(), ui.begin() + (size_t)(ns / 2) * n * n, ui.begin() + (size_t)(ns / 2) * n * n + n);
        } else {
            CK(wv_integrate_end(ctx, sig.data(), nullptr, nullptr));
        }
        res.out.insert(res.out.end(), sig.begin(), sig.begin() + 3 * (ns + 1));
        wv_timing t{};
        CK(wv_get_timing(ctx, &t));
        res.gave_up += t.gave_up;
        res.resident = t.resident;
        --pending;
        return true;
    };
    for (int op = 0; op < ops; ++op) {
        const unsigned kind = urand32() % 30;
        steps = 20 + 10 * (int)(urand32() % 3);   // 20 (the first frame is the initial state: needs the stream), 30, 40
        fields = 0;
        res.op_at.push_back(res.out.size());
        res.op_kind.push_back((int)kind);
        res.op_steps.push_back(steps);
        pause();
        if (kind < 5) {
            if (!begin()) return false;
            pause();
            if (!end()) return false;
        } else if (kind < 9) {
            if (!begin() || !begin()) return false;
            pause();
            if (!end()) return false;
            pause();
            if (!end()) return false;
        } else if (kind < 11) {  // a pipeline: never fewer than one call in flight for a while
            if (!begin()) return false;
            for (int k = 0; k < 4; ++k) {
                if (!begin()) return false;
                pause();
                if (!end()) return false;
            }
            if (!end()) return false;
        } else if (kind < 13) {
            CK(wv_observation(ctx, 128, 128, obs.data()));
            res.out.insert(res.out.end(), obs.begin(), obs.begin() + 4096);
            res.out.insert(res.out.end(), obs.end() - 4096, obs.end());
        } else if (kind == 13) {
            CK(wv_synchronize(ctx));
        } else if (kind == 14) {
            CK(wv_reset(ctx));
        } else if (kind == 15) {
            CK(wv_get_state(ctx, state.data()));
            res.out.push_back(state[(size_t)n * (n / 2) + n / 3]);
            CK(wv_set_state(ctx, state.data()));
        } else if (kind == 16) {  // an action that returns its trajectories
            fields = 1;
            if (!begin()) return false;
            pause();
            if (!end()) return false;
        } else if (kind == 17) {  // streamed trajectories, two such calls in flight
            fields = 2;
            if (!begin() || !begin()) return false;
            pause();
            if (!end()) return false;
            pause();
            if (!end()) return false;
        } else if (kind == 18) {  // the right-hand side between two actions
            CK(wv_get_state(ctx, state.data()));
            std::vector<float> k12((size_t)12 * n * n);
            CK(wv_rhs(ctx, state.data(), (float)(step0 * (double)dt), k12.data()));
            res.out.insert(res.out.end(), k12.begin() + (size_t)n * (n / 2), k12.begin() + (size_t)n * (n / 2) + n);
        } else if (kind == 19) {  // profiling mode for one action
            CK(wv_set_profiling(ctx, 1));
            if (!begin()) return false;
            if (!end()) return false;
            CK(wv_set_profiling(ctx, 0));
        } else if (kind == 20 || kind == 21) {  // three actions as ONE call (kind 21: every action's frames kept, observed afterwards)
            const int na = 3, sps = kind == 21 && steps == 20 ? 30 : steps;
            std::vector<float> dsg((size_t)(na + 1) * M * 4), tt(2 * na), ts((size_t)na * (sps + 1)), sg((size_t)3 * (na * sps + 1));
            for (int a = 0; a <= na; ++a) {
                if (a > 0) {
                    for (int j = 0; j < 18; ++j) r1[j] = fminf(fmaxf(r0[j] + scale * (2.0f * urand() - 1.0f), 0.2f), 1.0f);
                    r1[18] = r0[18];
                    r0.swap(r1);
                }
                for (int m = 0; m < M; ++m) {
                    float *d = &dsg[((size_t)a * M + m) * 4];
                    d[0] = pos[m], d[1] = pos[M + m], d[2] = r0[m], d[3] = c[m];
                }
            }
            for (int a = 0; a < na; ++a) {
                for (int q = 0; q <= sps; ++q) ts[(size_t)a * (sps + 1) + q] = (float)((double)(step0 + a * sps + q) * (double)dt);
                tt[2 * a] = ts[(size_t)a * (sps + 1)], tt[2 * a + 1] = ts[(size_t)a * (sps + 1) + sps];
            }
            CK(wv_set_design_sequence(ctx, na, sps, M, dsg.data(), tt.data()));
            CK(wv_integrate_begin(ctx, ts.data(), na * sps, kind == 21 ? 2 : 1, 1, 0));
            pause();
            CK(wv_integrate_end(ctx, sg.data(), nullptr, nullptr));
            res.out.insert(res.out.end(), sg.begin(), sg.end());
            step0 += na * sps;
            if (kind == 21)
                for (int a = 0; a < na; ++a) {
                    CK(wv_observation_action(ctx, a, 128, 128, obs.data()));
                    res.out.insert(res.out.end(), obs.begin() + 8192, obs.begin() + 8192 + 2048);
                }
        } else if (kind == 22) {  // the source moves
            const float mu2[2] = {-10.0f, -8.0f + 16.0f * urand()};
            CK(wv_set_gaussian_source(ctx, 1, mu2, sigma, amp, 1000.0f));
            if (!begin()) return false;
            if (!end()) return false;
        } else if (kind == 23) {  // the coefficient fields of the current design / source
            std::vector<float> f((size_t)n * n);
            CK(wv_speed_field(ctx, (float)(step0 * (double)dt), f.data()));
            res.out.insert(res.out.end(), f.begin() + (size_t)n * (n / 2), f.begin() + (size_t)n * (n / 2) + n);
            CK(wv_source_field(ctx, (float)(step0 * (double)dt), f.data()));
            res.out.insert(res.out.end(), f.begin() + (size_t)n * (n / 3), f.begin() + (size_t)n * (n / 3) + n);
        } else if (kind == 24) {  // env.wave leaves and comes back
            std::vector<float> wv((size_t)12 * n * n * 3);
            CK(wv_get_frames(ctx, wv.data()));
            res.out.insert(res.out.end(), wv.begin() + (size_t)n * (n / 2), wv.begin() + (size_t)n * (n / 2) + n);
            CK(wv_set_frames(ctx, wv.data()));
        } else if (kind == 25) {  // somebody holds the raw pointer of env.wave over an action
            void *dp = nullptr;
            size_t bytes = 0;
            CK(wv_device_frames(ctx, &dp, &bytes));
            if (!begin()) return false;
            if (!end()) return false;
            CK(wv_observation(ctx, 128, 128, obs.data()));
            res.out.insert(res.out.end(), obs.begin(), obs.begin() + 2048);
            CK(wv_release_device_frames(ctx));
        } else if (kind == 26) {  // strided trajectories
            CK(wv_set_trajectory_stride(ctx, 5));
            fields = 1;
            if (!begin()) return false;
            {
                const int ns = pend_steps.front();
                pend_steps.clear();
                pend_fields.clear();
                CK(wv_integrate_end(ctx, sig.data(), ut.data(), ui.data()));
                --pending;
                const int planes = ns / 5 + 1;
                res.out.insert(res.out.end(), ut.begin() + (size_t)(planes - 1) * n * n, ut.begin() + (size_t)planes * n * n);
                res.out.insert(res.out.end(), sig.begin(), sig.begin() + 3 * (ns + 1));
            }
            CK(wv_set_trajectory_stride(ctx, 1));
        } else if (kind == 28 || kind == 29) {  // a SECOND context on the device for a while (this one then leaves the resident kernel
                                                // to nobody: single-step kernels on both), stepped in turn with this one, then destroyed
            wv_ctx *other = nullptr;
            const int n2 = 96 + 32 * (int)(urand32() % 3);
            std::vector<float> x2(n2);
            for (int i = 0; i < n2; ++i) x2[i] = (float)(-15.0 + 30.0 * i / (n2 - 1));
            wv_config cfg2{n2, n2, 1531.0f, dt, 2.0f, 20000.0f, 0, WV_IMPL_AUTO};
            if (wv_create(&cfg2, x2.data(), x2.data(), &other) != 0) {
                fprintf(stderr, "second context: %s\n", wv_last_error(nullptr));
                return false;
            }
            std::vector<float> ts2(21), sg2(63);
            bool ok2 = wv_set_gaussian_source(other, 1, mu, sigma, amp, 1000.0f) == 0;
            for (int a = 0; a < 3 && ok2; ++a) {
                for (int q = 0; q <= 20; ++q) ts2[q] = (float)((double)(a * 20 + q) * (double)dt);
                ok2 = wv_set_design(other, M, pos.data(), r0.data(), c.data(), pos.data(), r0.data(), c.data(), ts2[0], ts2[20]) == 0 &&
                      wv_integrate_begin(other, ts2.data(), 20, 1, 1, 0) == 0;
                if (!begin()) return false;     // (this context's action while the other's is in flight)
                if (kind == 29 && !begin()) return false;
                ok2 = ok2 && wv_integrate_end(other, sg2.data(), nullptr, nullptr) == 0;
                res.out.insert(res.out.end(), sg2.begin(), sg2.end());
                pause();
                if (!end()) return false;
                if (kind == 29 && !end()) return false;
            }
            if (!ok2) {
                fprintf(stderr, "second context: %s\n", wv_last_error(other));
                return false;
            }
            if (wv_destroy(other) != 0) return false;
        } else {                  // an action without a design (NoDesign: C(t) = c0), then the design again
            for (int s2 = 0; s2 <= steps; ++s2) tspan[s2] = (float)((double)(step0 + s2) * (double)dt);
            CK(wv_set_design(ctx, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, tspan[0], tspan[steps]));
            CK(wv_integrate_begin(ctx, tspan.data(), steps, 1, 1, 0));
            CK(wv_integrate_end(ctx, sig.data(), nullptr, nullptr));
            res.out.insert(res.out.end(), sig.begin(), sig.begin() + 3 * (steps + 1));
            step0 += steps;
        }
    }
    steps = 30;   // one more plain action, alone on the device: the context must be (back) on the resident kernel
    fields = 0;
    if (!begin() || !end()) return false;
    CK(wv_get_state(ctx, state.data()));
    res.out.insert(res.out.end(), state.begin(), state.begin() + (size_t)n * n);
    CK(wv_destroy(ctx));
    return true;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 320;
    const int first = argc > 2 ? atoi(argv[2]) : 0, nseeds = argc > 3 ? atoi(argv[3]) : 10, ops = argc > 4 ? atoi(argv[4]) : 200;
    const bool need_resident = !(argc > 5 && atoi(argv[5]) == 0);  // 0: grids beyond the resident kernel (single-step kernels)
    const bool cross = argc > 6 && atoi(argv[6]) != 0;             // 1: the undisturbed run on the single-step kernels (resident vs single-step)
    const bool threads = argc > 7 && atoi(argv[7]) != 0;           // 1: seeds s and s + 100000 at once, a context each on a thread of
                                                                   //    its own (contexts are per thread: include/waves_amd.h), against
                                                                   //    the same two run one after the other
    if (threads) {
        int badt = 0;
        for (int s = first; s < first + nseeds; ++s) {
            Result a0, a1, b0, b1;
            bool ok0 = false, ok1 = false;
            std::thread t0([&] { ok0 = run(n, (unsigned long long)s, ops, true, a0); });
            std::thread t1([&] { ok1 = run(n, (unsigned long long)s + 100000ull, ops, true, a1); });
            t0.join();
            t1.join();
            const bool okb = run(n, (unsigned long long)s, ops, false, b0) && run(n, (unsigned long long)s + 100000ull, ops, false, b1);
            const bool same = ok0 && ok1 && okb && a0.out.size() == b0.out.size() && a1.out.size() == b1.out.size() &&
                              memcmp(a0.out.data(), b0.out.data(), a0.out.size() * sizeof(float)) == 0 &&
                              memcmp(a1.out.data(), b1.out.data(), a1.out.size() * sizeof(float)) == 0;
            printf("seeds %d and %d on two threads: %zu + %zu values: %s\n", s, s + 100000, a0.out.size(), a1.out.size(), same ? "same bytes" : "FAILED");
            fflush(stdout);
            badt += same ? 0 : 1;
        }
        printf(badt ? "FAILED\n" : "PASS\n");
        return badt ? 1 : 0;
    }
    const int idles[6] = {10, 20, 35, 50, 80, 1000};
    int bad = 0;
    for (int s = first; s < first + nseeds; ++s) {
        char buf[32];
        snprintf(buf, sizeof buf, "%d", idles[s % 6]);
        setenv("WAVES_AMD_IDLE_US", buf, 1);
        Result a, b;
        const bool oka = run(n, (unsigned long long)s, ops, true, a);
        unsetenv("WAVES_AMD_IDLE_US");
        if (cross) setenv("WAVES_AMD_FUSED_RESIDENT", "0", 1);
        const bool okb = run(n, (unsigned long long)s, ops, false, b);
        if (cross) unsetenv("WAVES_AMD_FUSED_RESIDENT");
        const bool same = oka && okb && a.out.size() == b.out.size() && memcmp(a.out.data(), b.out.data(), a.out.size() * sizeof(float)) == 0;
        const bool ok = same && a.gave_up == 0 && b.gave_up == 0 && (!need_resident || (a.resident && (cross || b.resident)));
        size_t firstdiff = 0;
        if (oka && okb && !same)
            for (; firstdiff < a.out.size() && firstdiff < b.out.size(); ++firstdiff)
                if (memcmp(&a.out[firstdiff], &b.out[firstdiff], 4) != 0) break;
        printf("seed %d idle %s us: %zu values, give-ups %d/%d, resident %d/%d: %s", s, buf, a.out.size(), a.gave_up, b.gave_up, a.resident, b.resident,
               ok ? "same bytes\n" : "FAILED");
        if (!ok) {
            size_t k = 0;
            while (k + 1 < a.op_at.size() && a.op_at[k + 1] <= firstdiff) ++k;
            printf(" (first difference at value %zu: operation %zu of kind %d with %d steps, offset %zu in it; kinds before: %d %d %d; %g vs %g)\n", firstdiff, k,
                   a.op_kind.empty() ? -1 : a.op_kind[k], a.op_steps.empty() ? 0 : a.op_steps[k], a.op_at.empty() ? 0 : firstdiff - a.op_at[k],
                   k > 0 ? a.op_kind[k - 1] : -1, k > 1 ? a.op_kind[k - 2] : -1, k > 2 ? a.op_kind[k - 3] : -1,
                   firstdiff < a.out.size() ? a.out[firstdiff] : 0.0f, firstdiff < b.out.size() ? b.out[firstdiff] : 0.0f);
        }
        fflush(stdout);
        bad += ok ? 0 : 1;
    }
    printf(bad ? "FAILED\n" : "PASS\n");
    return bad ? 1 : 0;
}
