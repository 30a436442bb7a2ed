// CPU emulation of the fused Runge-Kutta step kernel (TEST INFRASTRUCTURE).
//
// Runs the very phase functions the HIP kernel runs (waves.jl_amd/csrc/fused_body.h) in host loops over the thread
// ids -- a barrier is simply the end of a loop -- and compares the result bit for bit with the CPU oracle
// (oracle/waves_oracle.c, linked in).  Built with -fsanitize=address,undefined so that an out-of-range LDS or global
// index of the kernel's tile/halo logic is a hard failure here, on the CPU, before anything is launched on a GPU.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../waves.jl_amd/csrc/fused_plan.h"

extern "C" {
void wo_build_pml_profile(int n, const float *xs, float width, float scale, float *out);
void wo_design_at(int M, const float *d0, const float *d1, float ti, float tf, float t, float *out);
float wo_source_factor(float t, float freq);
int wo_integrate(int nx, int ny, const float *x, const float *y, const float *sx, const float *sy, float c0, float dt,
                 float *state, const float *tspan, int nsteps, const float *G, float freq, int M, const float *d0,
                 const float *d1, float ti, float tf, double *esum, float *frames, const int *frame_steps, int nframes,
                 int nthreads);
}

using namespace wv;

static F2 g_lds_raw[lds_elems(8 * 6)];  // deliberately NOT cleared between tiles: stale contents must never matter

template <bool PML, bool EDGE, int NW, int RPT>
static void run_tile(const FusedParams &p, const TileDesc &t, double esum[3])
{
    constexpr int NT = NW * 64;
    const FusedLds lds = lds_view(g_lds_raw, NW * RPT);
    std::vector<FusedRegs<PML, RPT>> regs(NT);
    for (int tid = 0; tid < NT; ++tid) fused_load<PML, EDGE, NW, RPT>(p, t, tid, regs[tid]);
#define STAGE(S)                                                                                            \
    for (int tid = 0; tid < NT; ++tid) fused_publish<PML, EDGE, NW, RPT, S>(p, t, tid, lds, regs[tid]);     \
    for (int tid = 0; tid < NT; ++tid) fused_compute<PML, EDGE, NW, RPT, S>(p, t, tid, lds, regs[tid]);
    STAGE(1) STAGE(2) STAGE(3) STAGE(4)
#undef STAGE
    for (int tid = 0; tid < NT; ++tid) {
        float e[3];
        fused_store<PML, EDGE, NW, RPT>(p, t, tid, regs[tid], e);
        for (int c = 0; c < 3; ++c) esum[c] += (double)e[c];
    }
}

template <int NW, int RF, int RP>
static void run_step(const FusedParams &p, const HostPlan &pl, double esum[3])
{
    for (const TileDesc &t : pl.tiles) {
        switch (t.variant) {
            case VAR_FAST: run_tile<false, false, NW, RF>(p, t, esum); break;
            case VAR_MID: run_tile<true, false, NW, RP>(p, t, esum); break;
            default: run_tile<true, true, NW, RP>(p, t, esum); break;
        }
    }
}

static unsigned long long rng_state = 88172645463325252ull;
static double urand()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}

struct Case {
    const char *name;
    int n;
    int nsteps;
    int NW, RF, RP;  // waves per block; rows per thread of FAST / MID+GEN tiles
    float pml_width, pml_scale;
    int M;           // cylinders
    int source;      // 0/1
    int aux;         // 1: random non-zero auxiliary fields everywhere (forces MID tiles); 0: aux zero -> FAST tiles
    int force_all;   // 1: natural launch order instead of the XCD-aware one
};

static int run_case(const Case &cs)
{
    const int n = cs.n, nsteps = cs.nsteps;
    const size_t P = (size_t)n * n, N = 12 * P;
    std::vector<float> x(n), sx(n);
    const float gs = 15.0f;
    for (int i = 0; i < n; ++i) x[i] = (float)(-gs + (2.0 * gs) * i / (n - 1));
    wo_build_pml_profile(n, x.data(), cs.pml_width, cs.pml_scale, sx.data());
    const float c0 = 1531.0f, dt = 1e-5f, freq = 1000.0f;
    // state: smooth-ish random, aux fields per case
    std::vector<float> u0(N, 0.0f);
    for (int f = 0; f < 12; ++f) {
        const bool is_aux = (f % 6) >= 3;
        for (size_t q = 0; q < P; ++q) {
            const int i = (int)(q % n), j = (int)(q / n);
            float v = (float)(0.1 * (urand() - 0.5));
            if (is_aux) {
                const bool in_pml = sx[i] != 0.0f || sx[j] != 0.0f;
                if (!cs.aux && !in_pml) v = 0.0f;
            }
            u0[(size_t)f * P + q] = v;
        }
    }
    std::vector<float> G(P, 0.0f);
    if (cs.source)
        for (size_t q = 0; q < P; ++q) {
            const float xx = x[q % n] - 1.0f, yy = x[q / n] + 0.5f;
            G[q] = (float)exp(-(xx * xx + yy * yy) / 2.0);
        }
    const int M = cs.M;
    std::vector<float> d0(4 * (size_t)(M ? M : 1)), d1(4 * (size_t)(M ? M : 1));
    for (int m = 0; m < M; ++m) {
        d0[4 * m + 0] = (float)(20.0 * (urand() - 0.5));
        d0[4 * m + 1] = (float)(20.0 * (urand() - 0.5));
        d0[4 * m + 2] = (float)(0.5 + 2.0 * urand());
        d0[4 * m + 3] = (float)(600.0 + 2000.0 * urand());
        d1[4 * m + 0] = d0[4 * m + 0] + (float)(0.5 * (urand() - 0.5));
        d1[4 * m + 1] = d0[4 * m + 1] + (float)(0.5 * (urand() - 0.5));
        d1[4 * m + 2] = d0[4 * m + 2] + (float)(0.5 * (urand() - 0.5));
        d1[4 * m + 3] = d0[4 * m + 3];
    }
    std::vector<float> tspan(nsteps + 1);
    const float t0 = 0.002f;
    for (int s = 0; s <= nsteps; ++s) tspan[s] = (float)((double)t0 + (double)s * 1e-5);
    const float ti = tspan[0], tf = tspan[nsteps];

    // ---- oracle
    std::vector<float> ref = u0;
    std::vector<double> eref(3 * (size_t)(nsteps + 1));
    wo_integrate(n, n, x.data(), x.data(), sx.data(), sx.data(), c0, dt, ref.data(), tspan.data(), nsteps,
                 cs.source ? G.data() : nullptr, freq, M, d0.data(), d1.data(), ti, tf, eref.data(), nullptr, nullptr, 0, 4);

    // ---- emulated fused kernel: tables exactly as api.hip builds them
    const float hdt = 0.5f * dt;
    std::vector<Cyl> table(3 * (size_t)nsteps * (M ? M : 1));
    std::vector<float> sfac(3 * (size_t)nsteps, 0.0f), tmp(4 * (size_t)(M ? M : 1));
    for (int s = 0; s < nsteps; ++s) {
        const float t = tspan[s];
        const float tq[3] = {t, t + hdt, t + dt};
        for (int q = 0; q < 3; ++q) {
            if (cs.source) sfac[3 * s + q] = wo_source_factor(tq[q], freq);
            if (M) {
                wo_design_at(M, d0.data(), d1.data(), ti, tf, tq[q], tmp.data());
                for (int m = 0; m < M; ++m) {
                    Cyl c{tmp[4 * m], tmp[4 * m + 1], tmp[4 * m + 2] * tmp[4 * m + 2], tmp[4 * m + 3]};
                    table[(size_t)(3 * s + q) * M + m] = c;
                }
            }
        }
    }
    HostPlan pl;
    if (!plan_build_tiles(pl, n, n, cs.NW * cs.RF, cs.NW * cs.RP, x.data(), x.data(), sx.data(), sx.data(), cs.aux == 0,
                          cs.force_all == 0)) {
        printf("%-28s plan_build_tiles failed\n", cs.name);
        return 1;
    }
    std::vector<int> idx;
    plan_build_cyl(pl, x.data(), x.data(), table.data(), M, 3 * nsteps, idx);

    std::vector<float> bufA = u0, bufB(N, 0.0f);
    if (cs.aux) {  // a dirty output buffer must not matter when no FAST tile exists
        for (size_t q = 0; q < N; ++q) bufB[q] = 123.0f;
    }
    float *cur = bufA.data(), *nxt = bufB.data();
    double emax = 0.0;
    for (int s = 0; s < nsteps; ++s) {
        FusedParams p{};
        p.nx = n; p.ny = n; p.P = P;
        const float delta = (x[n - 1] - x[0]) / (float)(n - 1), two_d = 2.0f * delta;
        p.ops = Ops{-1.0f / two_d, 1.0f / two_d, -3.0f / two_d, 4.0f / two_d, -1.0f / two_d, 1.0f / two_d, -4.0f / two_d, 3.0f / two_d};
        p.x = x.data(); p.y = x.data(); p.sx = sx.data(); p.sy = sx.data();
        p.c0 = c0; p.c0sq = c0 * c0;
        p.u = cur; p.out = nxt; p.G = cs.source ? G.data() : nullptr;
        p.sfac[0] = sfac[3 * s]; p.sfac[1] = sfac[3 * s + 1]; p.sfac[2] = sfac[3 * s + 2];
        p.cyl = table.data() + (size_t)(3 * s) * M; p.M = M;
        p.dt = dt; p.hdt = hdt;
        p.tiles = pl.tiles.data(); p.cyl_idx = idx.data();
        p.epart = nullptr; p.traj_tot = nullptr; p.traj_inc = nullptr;
        double es[3] = {0, 0, 0};
        const int key = cs.NW * 100 + cs.RF * 10 + cs.RP;
        if (key == 842) run_step<8, 4, 2>(p, pl, es);
        else if (key == 832) run_step<8, 3, 2>(p, pl, es);
        else if (key == 833) run_step<8, 3, 3>(p, pl, es);
        else if (key == 822) run_step<8, 2, 2>(p, pl, es);
        else if (key == 844) run_step<8, 4, 4>(p, pl, es);
        else if (key == 464) run_step<4, 6, 4>(p, pl, es);
        else { printf("unsupported NW/RF/RP\n"); return 1; }
        for (int c = 0; c < 3; ++c) {
            const double r = eref[3 * (size_t)(s + 1) + c];
            const double rel = fabs(es[c] - r) / (fabs(eref[3 * (size_t)(s + 1)]) + 1e-300);
            if (rel > emax) emax = rel;
        }
        float *tsw = cur; cur = nxt; nxt = tsw;
    }
    // compare (bit-exact up to the sign of zero)
    size_t bad = 0, first = 0;
    for (size_t q = 0; q < N; ++q)
        if (!(cur[q] == ref[q])) {
            if (!bad) first = q;
            ++bad;
        }
    double umax = 0;
    for (size_t q = 0; q < P; ++q) umax = fmax(umax, fabs(ref[q]));
    printf("%-28s n=%4d steps=%3d NW,RF,RP=%d,%d,%d tiles FAST/MID/GEN=%d/%d/%d  culled-list=%zu  max|U|=%.3g  energy rel=%.1e  %s",
           cs.name, n, nsteps, cs.NW, cs.RF, cs.RP, pl.count[0], pl.count[1], pl.count[2], idx.size(), umax, emax,
           bad ? "MISMATCH" : "bit-exact\n");
    if (bad) {
        const size_t f = first / P, q = first % P;
        printf(" (%zu cells; first: field %zu, i=%zu, j=%zu: got %.9g want %.9g)\n", bad, f, q % n, q / n, cur[first], ref[first]);
    }
    return (bad || emax > 1e-6) ? 1 : 0;
}

int main(int argc, char **argv)
{
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    std::vector<Case> cases = {
        {"default tiles, design+src", 160, 6, 8, 4, 2, 2.0f, 20000.0f, 6, 1, 0, 0},
        {"all-mid (aux everywhere)", 96, 5, 8, 4, 2, 2.0f, 20000.0f, 4, 1, 1, 0},
        {"aux everywhere, 260", 260, 3, 8, 4, 2, 2.0f, 20000.0f, 4, 1, 1, 0},
        {"no pml (scale 0), no design", 130, 5, 8, 4, 2, 1.0f, 0.0f, 0, 1, 0, 0},
        {"RF,RP=3,2", 231, 4, 8, 3, 2, 2.0f, 20000.0f, 5, 1, 0, 0},
        {"RF,RP=3,3 natural order", 200, 4, 8, 3, 3, 2.0f, 20000.0f, 5, 0, 0, 1},
        {"RF,RP=2,2", 131, 4, 8, 2, 2, 2.0f, 20000.0f, 5, 1, 0, 0},
        {"RF,RP=4,4", 150, 4, 8, 4, 4, 2.0f, 20000.0f, 3, 1, 0, 0},
        {"NW=4 RF,RP=6,4", 117, 4, 4, 6, 4, 3.0f, 20000.0f, 3, 1, 0, 0},
        {"tiny grid 8", 8, 3, 8, 4, 2, 2.0f, 20000.0f, 1, 1, 0, 0},
        {"grid 57 (two strips)", 57, 4, 8, 4, 2, 2.0f, 20000.0f, 2, 1, 0, 0},
        {"many cylinders", 180, 3, 8, 4, 2, 2.0f, 20000.0f, 40, 1, 0, 0},
    };
    if (!quick) {
        cases.push_back({"config-2 like 700, 3 steps", 700, 3, 8, 4, 2, 2.0f, 20000.0f, 19, 1, 0, 0});
        cases.push_back({"wide pml 4.0 at 300", 300, 5, 8, 4, 2, 4.0f, 20000.0f, 8, 1, 0, 0});
    }
    int fails = 0;
    for (const Case &c : cases) fails += run_case(c);
    printf("%s (%d failing case%s)\n", fails ? "FAIL" : "PASS", fails, fails == 1 ? "" : "s");
    return fails ? 1 : 0;
}
