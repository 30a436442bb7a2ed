// What the launch order of a call costs the host (plan_build_cyl + plan_pair_order, fused_plan.h) at 700^2 with the triple ring:
// the part of wv_integrate_begin that a call without host-built tables still pays (fused_try_resident, dev).
//   g++ -O2 -std=c++17 -I waves.jl_amd/csrc tools/micro/plan_cost.cpp -o /tmp/plan_cost && /tmp/plan_cost
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fused_plan.h"
using namespace wv;

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 700, M = 19;
    std::vector<float> x(n), sx(n, 0.0f);
    for (int i = 0; i < n; ++i) x[i] = (float)(-15.0 + 30.0 * i / (n - 1));
    for (int i = 0; i < n; ++i) {
        const double d = std::min(x[i] + 15.0, 15.0 - x[i]);
        sx[i] = d < 2.0 ? (float)(20000.0 * (2.0 - d) / 2.0) : 0.0f;
    }
    HostPlan pl;
    if (!plan_build_tiles(pl, n, n, 32, 24, 16, x.data(), x.data(), sx.data(), sx.data(), true, true)) return 1;
    printf("%zu tiles\n", pl.tiles.size());
    std::vector<Cyl> ends(2 * M);
    const double ring_r[3] = {3.5, 4.75, 6.0}, rot[3] = {0.0, M_PI / 6.0, 0.0};
    unsigned long long rng = 12345;
    auto urand = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (double)(rng >> 40) / 16777216.0; };
    std::vector<int> idx;
    double best = 1e9, sum = 0.0;
    const int reps = 200;
    for (int it = 0; it < reps; ++it) {
        for (int e = 0; e < 2; ++e)
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < 6; ++j) {
                    const double a = j * M_PI / 3.0 + rot[k], r = 0.2 + 0.8 * urand();
                    ends[e * M + 6 * k + j] = Cyl{(float)(ring_r[k] * cos(a) + 5.0), (float)(ring_r[k] * sin(a)), (float)(r * r), 1032.0f};
                }
        ends[18] = ends[M + 18] = Cyl{5.0f, 0.0f, 4.0f, 1032.0f};
        const auto t0 = std::chrono::steady_clock::now();
        plan_build_cyl(pl, x.data(), x.data(), ends.data(), M, 2, idx, true, 256, 0, 1);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        best = std::min(best, us);
        sum += us;
    }
    printf("plan_build_cyl + pair order: best %.1f us, mean %.1f us\n", best, sum / reps);
    best = 1e9;
    for (int it = 0; it < reps; ++it) {
        const auto t0 = std::chrono::steady_clock::now();
        plan_build_cyl(pl, x.data(), x.data(), ends.data(), 0, 2, idx, true, 256, 0, 1);
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    {
        double bc = 1e9, bp = 1e9;
        for (int it = 0; it < reps; ++it) {
            auto t0 = std::chrono::steady_clock::now();
            plan_build_cyl(pl, x.data(), x.data(), ends.data(), M, 2, idx, false, 0, 0, 1);
            auto t1 = std::chrono::steady_clock::now();
            plan_pair_order(pl, 256);
            auto t2 = std::chrono::steady_clock::now();
            bc = std::min(bc, std::chrono::duration<double, std::micro>(t1 - t0).count());
            bp = std::min(bp, std::chrono::duration<double, std::micro>(t2 - t1).count());
        }
        printf("culling alone: best %.1f us; pair order alone: best %.1f us\n", bc, bp);
    }
    printf("without cylinders (copy of the base tiles + pair order): best %.1f us\n", best);
    return 0;
}
