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

#include <atomic>
#include <thread>
#include <vector>

#define WV_XCH_DEBUG 1
static thread_local int wv_xch_debug = 0;
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

// (thread_local: the job-protocol emulation runs every block of a launch as a thread of its own)
static thread_local F2 g_lds_raw[lds_elems(8 * 6)];  // deliberately NOT cleared between tiles: stale contents must never matter

static thread_local bool g_probe_only = false;  // resident emulation: only poll the halo, do not run the step
static int g_maxdrift = 0;
static thread_local bool g_not_ready = false;  // set by a resident tile whose halo words have not all arrived
static thread_local const int *g_cull = nullptr;  // resident emulation: the tile's cylinder list as the DEVICE culls it, or nullptr
static int g_force_all = 0;  // 1: every tile runs the F_ALL instantiation (must give the same bits as the specialised ones)

// Registers of one tile (type-erased: the register struct depends on the variant), kept from step to step in the
// emulation of k_steps_resident.
struct TileMem {
    std::vector<char> regs, cx;
};
enum { MODE_SINGLE = 0, MODE_RESIDENT_FIRST = 1, MODE_RESIDENT_NEXT = 2 };

template <int AUX, int FL, int NW, int RPT, int RYMAX>
static void run_tile(const FusedParams &p, const StepIO &io, const TileDesc &t, double esum[3], TileMem &mem, int mode)
{
    constexpr int NT = NW * 64;
    using Regs = FusedRegs<AUX, RPT>;
    const FusedLds lds = lds_view(g_lds_raw, NW * RPT, RYMAX);
    if (mode != MODE_RESIDENT_NEXT) {
        mem.regs.assign(NT * sizeof(Regs), (char)0x5a);  // garbage, like fresh registers
        mem.cx.assign(NT * sizeof(TileCtx), (char)0x5a);
    }
    Regs *regs = reinterpret_cast<Regs *>(mem.regs.data());
    TileCtx *cx = reinterpret_cast<TileCtx *>(mem.cx.data());
    if (mode == MODE_SINGLE) {
        for (int tid = 0; tid < NT; ++tid) fused_load<AUX, FL, NW, RPT>(p, io, t, tid, lds, cx[tid], regs[tid]);
    } else {  // the phase sequence of run_tile_resident
        if (mode == MODE_RESIDENT_FIRST) {
            for (int tid = 0; tid < NT; ++tid) {
                fused_tile_init<AUX, FL, NW, RPT>(p, t, tid, cx[tid], regs[tid], g_cull);
                fused_load_state<AUX, NW, RPT>(p, io.u, t, tid, regs[tid]);
            }
        } else {
            // the poll of the halo words of the previous step: all threads must find their tags, else the tile is not
            // runnable yet (nothing but r.u is touched by a failed poll)
            bool ok = true;
            for (int tid = 0; tid < NT; ++tid)
                ok = fused_xch_load<AUX, NW, RPT>(p, p.tag_base + (unsigned)io.step, t, tid, regs[tid]) && ok;
            if (!ok) {
                g_not_ready = true;
                return;
            }
            if (g_probe_only) return;
        }
        for (int tid = 0; tid < NT; ++tid) {
            fused_step_init<FL>(p, io.step, cx[tid]);
            fused_cyl_times<FL>(p, io.step, cx[tid]);
            // (on the device the fetch of a step > 0 is issued before the halo poll of the previous step and committed
            // after it; the tile's LDS image is private to the tile, so the emulation can do both here)
            fused_cyl_commit<FL>(t, tid, lds, cx[tid], fused_cyl_fetch<AUX, FL, RPT>(p, io.step, t, tid, cx[tid], regs[tid]));
        }
    }
    for (int tid = 0; tid < NT; ++tid) fused_publish<AUX, FL, NW, RPT, 1>(p, t, tid, lds, cx[tid], regs[tid]);
    for (int tid = 0; tid < NT; ++tid) fused_speed<AUX, FL, NW, RPT>(p, t, tid, lds, cx[tid], regs[tid]);
    // Between two barriers the waves of a block run in any order and at any relative speed, while the 64 lanes of one
    // wave run in lock-step.  The emulation runs whole waves one after the other in a tile- and stage-dependent order:
    // a wave's compute(S) for all its lanes (neighbour lanes are read as the DPP shifts would), then -- with no barrier
    // in between unless the variant has one -- its publish(S+1), so a wave that "runs ahead" overwrites whatever it is
    // allowed to overwrite before the slower waves have read the current stage.
#define STAGE(S)                                                                                                      \
    for (int k = 0; k < NW; ++k) {                                                                                    \
        const int w = (5 * k + 3 * S + t.slot) % NW;                                                                  \
        for (int l = 0; l < 64; ++l)                                                                                  \
            fused_compute<AUX, FL, NW, RPT, S>(p, t, w * 64 + l, lds, cx[w * 64 + l], regs[w * 64 + l], &regs[w * 64]); \
        if (!((FL & F_EDGE) && !lds_side_double(NW * RPT, RYMAX)))                                                    \
            for (int l = 0; l < 64; ++l)                                                                              \
                fused_publish<AUX, FL, NW, RPT, S + 1>(p, t, w * 64 + l, lds, cx[w * 64 + l], regs[w * 64 + l]);      \
    }                                                                                                                 \
    if ((FL & F_EDGE) && !lds_side_double(NW * RPT, RYMAX))                                                           \
        for (int tid = 0; tid < NT; ++tid) fused_publish<AUX, FL, NW, RPT, S + 1>(p, t, tid, lds, cx[tid], regs[tid]);
    STAGE(1) STAGE(2) STAGE(3)
#undef STAGE
    for (int k = 0; k < NW; ++k) {
        const int w = (3 * k + t.slot) % NW;
        for (int l = 0; l < 64; ++l)
            fused_compute<AUX, FL, NW, RPT, 4>(p, t, w * 64 + l, lds, cx[w * 64 + l], regs[w * 64 + l], &regs[w * 64]);
    }
    if (mode != MODE_SINGLE && io.step + 1 != p.nsteps)
        for (int tid = 0; tid < NT; ++tid)
            fused_xch_store<AUX, NW, RPT>(p, p.tag_base + (unsigned)(io.step + 1), t, tid, regs[tid]);
    for (int tid = 0; tid < NT; ++tid) {
        float e[3];
        fused_store<AUX, NW, RPT>(p, io, t, tid, regs[tid], e);
        for (int c = 0; c < 3; ++c) esum[c] += (double)e[c];
    }
}

// the variant dispatch of k_step_fused / k_steps_resident for one tile
template <int NW, int RF, int RB, int RP>
static void run_one(const FusedParams &p, const StepIO &io, const TileDesc &t, double esum[3], TileMem &mem, int mode)
{
    constexpr int RMAX = RF > RB ? (RF > RP ? RF : RP) : (RB > RP ? RB : RP);
    constexpr int RYMAX = NW * RMAX;
    static_assert(RYMAX <= 48, "g_lds_raw too small");
#define RUN(A, F, R) run_tile<A, F, NW, R, RYMAX>(p, io, t, esum, mem, mode)
    if (g_force_all) {  // every tile through the most general body of its field set: same bits expected
        if (t.aux == AUX_NONE) RUN(AUX_NONE, F_ALL, RF);
        else if (t.aux == AUX_PX) RUN(AUX_PX, F_ALL, RB);
        else if (t.aux == AUX_PY) RUN(AUX_PY, F_ALL, RB);
        else RUN(AUX_ALL, F_ALL, RP);
        return;
    }
    const int fl = tile_flags(p, t);
    const int fe = fl & F_EDGE;
    const bool cyl = (fl & F_CYL) != 0;
    if (t.aux == AUX_NONE) {  // never a boundary tile (fused_plan.h)
        if (fl == 0) RUN(AUX_NONE, 0, RF);
        else if (fl == F_SRC) RUN(AUX_NONE, F_SRC, RF);
        else RUN(AUX_NONE, F_CYL | F_SRC, RF);
    } else if (t.aux == AUX_PX) {
        const bool src = (fl & F_SRC) != 0;  // (the source rarely reaches the PML: its own variants without it)
        if (!cyl && !src && (fe & ~F_EL) == 0) RUN(AUX_PX, F_EL, RB);  // left strip, or a PML strip off the boundary
        else if (!cyl && !src && (fe & ~F_ER) == 0) RUN(AUX_PX, F_ER, RB);
        else if (!cyl && (fe & ~F_EL) == 0) RUN(AUX_PX, F_EL | F_SRC, RB);
        else if (!cyl && (fe & ~F_ER) == 0) RUN(AUX_PX, F_ER | F_SRC, RB);
        else RUN(AUX_PX, F_ALL, RB);
    } else if (t.aux == AUX_PY) {
        if (fl == 0) RUN(AUX_PY, 0, RB);
        else if (!cyl && (fl & F_SRC) == 0 && (fe & ~F_ET) == 0) RUN(AUX_PY, F_ET, RB);
        else if (!cyl && (fl & F_SRC) == 0 && (fe & ~F_EB) == 0) RUN(AUX_PY, F_EB, RB);
        else if (!cyl && (fe & ~F_ET) == 0) RUN(AUX_PY, F_ET | F_SRC, RB);
        else if (!cyl && (fe & ~F_EB) == 0) RUN(AUX_PY, F_EB | F_SRC, RB);
        else RUN(AUX_PY, F_ALL, RB);
    } else {
        if (!cyl && (fl & F_SRC) == 0) RUN(AUX_ALL, F_EDGE, RP);
        else if (!cyl) RUN(AUX_ALL, F_EDGE | F_SRC, RP);
        else RUN(AUX_ALL, F_ALL, RP);
    }
#undef RUN
}

static bool run_dispatch(int key, const FusedParams &p, const StepIO &io, const TileDesc &t, double esum[3], TileMem &mem, int mode)
{
    if (key == 8432) run_one<8, 4, 3, 2>(p, io, t, esum, mem, mode);
    else if (key == 8332) run_one<8, 3, 3, 2>(p, io, t, esum, mem, mode);
    else if (key == 8322) run_one<8, 3, 2, 2>(p, io, t, esum, mem, mode);
    else if (key == 8222) run_one<8, 2, 2, 2>(p, io, t, esum, mem, mode);
    else if (key == 8333) run_one<8, 3, 3, 3>(p, io, t, esum, mem, mode);
    else if (key == 8422) run_one<8, 4, 2, 2>(p, io, t, esum, mem, mode);
    else if (key == 4644) run_one<4, 6, 4, 4>(p, io, t, esum, mem, mode);
    else return false;
    return true;
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
    int NW, RF, RB, RP;  // waves per block; rows per thread of AUX_NONE / AUX_PX+PY / AUX_ALL tiles
    float pml_width, pml_scale;
    int M;           // cylinders
    int source;      // 0/1
    int aux;         // 1: random non-zero auxiliary fields everywhere (forces AUX_ALL tiles); 0: only where they may be
                     // non-zero (Psi_x where sigma_x != 0, ...) -> reduced field sets
    int force_all;   // 1: natural launch order instead of the XCD-aware one AND every tile through the F_ALL body
    int src_mode = 1;  // 1: per-tile source flags computed; 0: src_flags = nullptr (assume non-zero everywhere)
    int resident = 0;  // 1: the protocol of k_steps_resident -- tiles keep their registers from step to step, re-read only
                       // their halo, and run in a random order constrained by nothing but the neighbour flags
    int devcyl = 0;    // 1 (resident only): no cylinder table and no culled lists from the host -- the tiles evaluate the
                       // DesignInterpolator themselves (design_cyl) and cull themselves (device_cull_keep), as FusedParams::dsg /
                       // dev_cull make k_steps_resident do
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
            if (is_aux && !cs.aux) {
                const int k = f % 6;  // 3: Psi_x, 4: Psi_y, 5: Omega
                const bool zx = sx[i] == 0.0f, zy = sx[j] == 0.0f;
                if ((k == 3 && zx) || (k == 4 && zy) || (k == 5 && (zx || zy))) v = 0.0f;
            }
            u0[(size_t)f * P + q] = v;
        }
    }
    std::vector<float> G(P, 0.0f);
    if (cs.source)
        for (size_t q = 0; q < P; ++q) {
            const float xx = x[q % n] - 1.0f, yy = x[q / n] + 0.5f;
            G[q] = (float)exp(-(xx * xx + yy * yy) / 0.18);  // underflows to exactly 0 ~4.4 units away, like the env's source
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
    g_force_all = cs.force_all;
    if (!plan_build_tiles(pl, n, n, cs.NW * cs.RF, cs.NW * cs.RB, cs.NW * cs.RP, x.data(), x.data(), sx.data(), sx.data(),
                          cs.aux == 0, cs.force_all == 0)) {
        printf("%-28s plan_build_tiles failed\n", cs.name);
        return 1;
    }
    std::vector<int> idx;
    plan_build_cyl(pl, x.data(), x.data(), table.data(), M, 3 * nsteps, idx);
    {   // the culling from the rows of the earliest / latest stage time only (what api.hip passes) must give the same lists
        HostPlan pl2 = pl;
        std::vector<int> idx2;
        plan_build_cyl(pl2, x.data(), x.data(), table.data(), M, 3 * nsteps, idx2, true, 0, 0, 3 * nsteps - 1);
        bool same = idx2 == idx && pl2.tiles.size() == pl.tiles.size();
        for (size_t k = 0; same && k < pl.tiles.size(); ++k)
            same = pl.tiles[k].slot == pl2.tiles[k].slot && pl.tiles[k].cyl_begin == pl2.tiles[k].cyl_begin &&
                   pl.tiles[k].cyl_count == pl2.tiles[k].cyl_count;
        if (!same) {
            printf("%-28s culling from the two extreme rows differs from the full scan\n", cs.name);
            return 1;
        }
    }

    // per-tile source flags exactly as k_src_flags computes them
    std::vector<unsigned char> flags(pl.tiles.size(), 0);
    for (const TileDesc &t : pl.tiles)
        for (int gy = t.y0 - FT_H; gy < t.y0 + t.oy + FT_H; ++gy)
            for (int gx = t.x0 - FT_H; gx < t.x0 + t.ox + FT_H; ++gx)
                if (gx >= 0 && gx < n && gy >= 0 && gy < n && G[(size_t)gy * n + gx] != 0.0f) flags[t.slot] = 1;
    std::vector<float> bufA = u0, bufB(N, 0.0f);
    if (cs.aux) {  // a dirty output buffer must not matter when no FAST tile exists
        for (size_t q = 0; q < N; ++q) bufB[q] = 123.0f;
    }
    FusedParams p{};
    p.nx = n; p.ny = n; p.P = (unsigned)P;
    const float delta = (x[n - 1] - x[0]) / (float)(n - 1), two_d = 2.0f * delta;
    p.ops = Ops{-1.0f / two_d, 1.0f / two_d, -3.0f / two_d, 4.0f / two_d, -1.0f / two_d, 1.0f / two_d, -4.0f / two_d, 3.0f / two_d};
    p.x = x.data(); p.y = x.data(); p.sx = sx.data(); p.sy = sx.data();
    p.c0 = c0; p.c0sq = c0 * c0;
    p.G = cs.source ? G.data() : nullptr;
    p.src_flags = (cs.source && cs.src_mode) ? flags.data() : nullptr;
    p.sfac_tab = sfac.data(); p.cyl_tab = table.data(); p.M = M; p.tile_offset = 0;
    p.dt = dt; p.hdt = hdt;
    p.tiles = pl.tiles.data(); p.cyl_idx = idx.data();
    std::vector<unsigned long long> xch(cs.resident ? (size_t)2 * XCH_PLANES * 2 * P : 2, 0ull);
    p.xch = xch.data();
    p.xch_bytes = (unsigned)(xch.size() * sizeof(unsigned long long));
    p.tag_base = 4094;  // arbitrary; even + odd tags both occur
    p.reduced = cs.aux == 0;
    const int key = cs.NW * 1000 + cs.RF * 100 + cs.RB * 10 + cs.RP;
    // step table: ping-pong between the two buffers, exactly two (the protocol must make that safe)
    std::vector<StepIO> steps(nsteps);
    for (int s = 0; s < nsteps; ++s)
        steps[s] = StepIO{(s & 1) ? bufB.data() : bufA.data(), (s & 1) ? bufA.data() : bufB.data(), nullptr, nullptr, nullptr, s, 0};
    p.steps = steps.data();
    p.nsteps = nsteps;
    std::vector<double> es(3 * (size_t)nsteps, 0.0);
    std::vector<TileMem> mem(pl.tiles.size());
    const size_t nt = pl.tiles.size();
    JobDesign dsg{};
    std::vector<std::vector<int>> culls(nt);
    if (cs.devcyl) {
        if (M < 1 || M > FT_MAXCYL) { printf("%-28s devcyl needs 1 <= M <= %d\n", cs.name, FT_MAXCYL); return 1; }
        dsg.M = M; dsg.ti = ti; dsg.tf = tf;
        design_slopes(dsg, d0.data(), d1.data());
        p.dsg = &dsg; p.tspan = tspan.data(); p.dev_cull = 1;
        p.cull_t_lo = tspan[0]; p.cull_t_hi = tspan[nsteps - 1] + dt;
        // the device's evaluation of the interpolator gives the host table's bits, its culling the host's lists
        for (int s = 0; s < nsteps; ++s)
            for (int q = 0; q < 3; ++q)
                for (int m = 0; m < M; ++m) {
                    const float tq = q == 0 ? tspan[s] : (q == 1 ? tspan[s] + hdt : tspan[s] + dt);
                    const Cyl a = design_cyl(dsg, m, tq), b = table[(size_t)(3 * s + q) * M + m];
                    if (memcmp(&a, &b, sizeof(Cyl)) != 0) { printf("%-28s design_cyl differs from the host table (step %d stage %d cyl %d)\n", cs.name, s, q, m); return 1; }
                }
        for (TileDesc &t : pl.tiles) {
            std::vector<int> &c = culls[t.slot];
            c.push_back(0);
            for (int m = 0; m < M; ++m)
                if (device_cull_keep(p, t, m)) c.push_back(m);
            c[0] = (int)c.size() - 1;
            bool same = c[0] == t.cyl_count;
            for (int k = 0; same && k < t.cyl_count; ++k) same = c[1 + k] == idx[t.cyl_begin + k];
            if (!same) { printf("%-28s device culling differs from the host's list (tile slot %d: %d vs %d cylinders)\n", cs.name, t.slot, c[0], t.cyl_count); return 1; }
        }
        p.cyl_tab = nullptr;   // neither the table nor the host's lists are there for the tiles to read
        p.cyl_idx = nullptr;
    }
    if (!cs.resident) {
        for (int s = 0; s < nsteps; ++s) {
            p.io = steps[s];
            for (const TileDesc &t : pl.tiles)
                if (!run_dispatch(key, p, p.io, t, &es[3 * (size_t)s], mem[t.slot], MODE_SINGLE)) { printf("unsupported NW/RF/RB/RP\n"); return 1; }
        }
    } else {
        // done[slot] = steps the tile has completed.  Tiles are tried in random order; a tile runs its next step when all
        // the halo words it polls carry the expected tag -- nothing else orders them, exactly as on the device -- so they
        // drift as far apart as the protocol lets them.  The state buffers ping-pong between two allocations and are
        // never read back by a resident tile; the exchange buffer has one copy per tag parity.
        std::vector<int> done(nt, 0);
        size_t remaining = nt * (size_t)nsteps;
        int maxdrift = 0;
        while (remaining) {
            std::vector<int> order(nt);
            for (size_t i = 0; i < nt; ++i) order[i] = (int)i;
            for (size_t k = 0; k + 1 < nt; ++k) {
                const size_t j = k + (size_t)(urand() * (double)(nt - k));
                std::swap(order[k], order[j < nt ? j : nt - 1]);
            }
            size_t progressed = 0;
            for (size_t k = 0; k < nt; ++k) {
                if (urand() < 0.5) continue;  // leave some runnable tiles behind on purpose
                const TileDesc &t = pl.tiles[order[k]];
                const int s = done[t.slot];
                if (s >= nsteps) continue;
                g_not_ready = false;
                g_cull = cs.devcyl ? culls[t.slot].data() : nullptr;
                double es_tile[3] = {0, 0, 0};
                if (!run_dispatch(key, p, steps[s], t, es_tile, mem[t.slot], s == 0 ? MODE_RESIDENT_FIRST : MODE_RESIDENT_NEXT)) {
                    printf("unsupported NW/RF/RB/RP\n");
                    return 1;
                }
                if (g_not_ready) continue;
                for (int c = 0; c < 3; ++c) es[3 * (size_t)s + c] += es_tile[c];
                done[t.slot] = s + 1;
                --remaining;
                ++progressed;
            }
            int lo = nsteps, hi = 0;
            for (int d : done) { lo = d < lo ? d : lo; hi = d > hi ? d : hi; }
            maxdrift = hi - lo > maxdrift ? hi - lo : maxdrift;
            if (!progressed) {
                bool any = false;  // nobody advanced: fine if the dice skipped every runnable tile, a deadlock otherwise
                for (size_t k = 0; k < nt && !any; ++k) {
                    const TileDesc &t = pl.tiles[k];
                    const int s = done[t.slot];
                    if (s >= nsteps) continue;
                    if (s == 0) { any = true; break; }
                    std::vector<char> save = mem[t.slot].regs;
                    g_not_ready = false;
                    TileMem probe = mem[t.slot];
                    double dummy[3] = {0, 0, 0};
                    // probing must not run the step: use a copy and a poll-only pass
                    g_probe_only = true;
                    wv_xch_debug = 3;
                    run_dispatch(key, p, steps[s], t, dummy, probe, MODE_RESIDENT_NEXT);
                    g_probe_only = false;
                    any = !g_not_ready;
                }
                if (!any) { printf("%-28s resident protocol deadlocked\n", cs.name); return 1; }
            }
        }
        g_maxdrift = maxdrift;
    }
    double emax = 0.0;
    for (int s = 0; s < nsteps; ++s)
        for (int c = 0; c < 3; ++c) {
            const double r = eref[3 * (size_t)(s + 1) + c];
            const double rel = fabs(es[3 * (size_t)s + c] - r) / (fabs(eref[3 * (size_t)(s + 1)]) + 1e-300);
            if (rel > emax) emax = rel;
        }
    const float *cur = (nsteps & 1) ? bufB.data() : bufA.data();
    // compare (bit-exact up to the sign of zero)
    size_t bad = 0, first = 0;
    for (size_t q = 0; q < N; ++q)
        if (!(cur[q] == ref[q])) {
            if (!bad) first = q;
            ++bad;
        }
    double umax = 0;
    for (size_t q = 0; q < P; ++q) umax = fmax(umax, fabs(ref[q]));
    int nedge = 0;
    for (const TileDesc &t : pl.tiles) nedge += t.edge ? 1 : 0;
    printf("%-28s n=%4d steps=%3d NW,RF,RB,RP=%d,%d,%d,%d tiles NONE/PX/PY/ALL=%d/%d/%d/%d edge=%d  culled-list=%zu  max|U|=%.3g  energy rel=%.1e  %s",
           cs.name, n, nsteps, cs.NW, cs.RF, cs.RB, cs.RP, pl.count[0], pl.count[1], pl.count[2], pl.count[3], nedge, idx.size(), umax, emax,
           bad ? "MISMATCH" : (cs.resident ? "bit-exact (resident)" : "bit-exact\n"));
    if (cs.resident && !bad) printf(" max drift %d steps\n", g_maxdrift);
    if (bad) {
        const size_t f = first / P, q = first % P;
        printf(" (%zu cells; first: field %zu, i=%zu, j=%zu: got %.9g want %.9g)\n", bad, f, q % n, q / n, cur[first], ref[first]);
    }
    return (bad || emax > 1e-6) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The JOB protocol of k_steps_resident (fused_body.h "jobs", kernels_fused.hip): one launch serves several wv_integrate
// calls.  Every block of the launch is a thread here and runs the kernel's job loop with the SAME protocol functions the
// device runs (job_leader_fetch / job_try_fetch / job_wait_go / job_barrier: compiled for the host they use C++ atomics),
// around the emulated tile steps; another thread plays the host API: it describes jobs in the mailbox, rings the bell
// late, early (so that the leader's look-ahead under barrier B finds the next job) or not at all (idle limit), tells the
// launch to leave, or forces a give-up.  Checked: every job's final state bit for bit against the oracle although the
// tile -> block mapping changes from job to job (the state travels through memory between jobs), the traces, that the
// initial condition of an abandoned job is intact, and that the launch DRAINS -- every block thread ends -- on every path.
static std::atomic<unsigned long long> g_vclock{0};
unsigned long long wv::emu_job_clock() { return g_vclock.fetch_add(1, std::memory_order_relaxed); }
void wv::emu_job_pause() { std::this_thread::yield(); }

struct EmuLaunch {
    JobArgs args;
    int key;
    std::vector<float> *epart;     // [job & 1][step][tile][3]: the tiles' partial sums (what the device keeps in d_epart)
    std::vector<double> *rows;     // [job & 1][(nsteps + 1)][3]: the trace the blocks reduce ("pinned host memory")
    std::atomic<int> running{0};
};

static void emu_block(EmuLaunch *E, int b)
{
    const JobArgs &a = E->args;
    unsigned seq = a.first_seq, pref_seq = 0;
    int pref_cmd = 0;
    for (;;) {
        int cmd;
        if (b == 0) {
            cmd = pref_seq == seq ? pref_cmd : job_leader_fetch(a, seq, 0);
            if (cmd == JOB_RUN) job_st_sys64(&a.back->t_begin[seq & 1u], job_clock());
        } else {
            cmd = job_wait_go(a, seq);
        }
        if (cmd != JOB_RUN) break;
        const FusedParams &p = a.ctl->jobs[seq & 1u].p;
        TileDesc t = p.tiles[b];
        int cull[1 + FT_MAXCYL] = {0};
        if (p.dev_cull) {  // the tile finds its cylinders itself (kernel body of k_steps_resident)
            for (int m = 0; m < p.M; ++m)
                if (device_cull_keep(p, t, m)) cull[1 + cull[0]++] = m;
            t.cyl_count = cull[0];
        }
        g_cull = p.dev_cull ? cull : nullptr;
        TileMem mem;
        bool ok = true;
        for (int s = 0; s < p.nsteps && ok; ++s) {
            double es[3] = {0, 0, 0};
            for (int polls = 0;; ++polls) {
                g_not_ready = false;
                run_dispatch(E->key, p, p.steps[s], t, es, mem, s == 0 ? MODE_RESIDENT_FIRST : MODE_RESIDENT_NEXT);
                if (!g_not_ready) break;
                if (polls >= p.max_polls || job_ld_agent(reinterpret_cast<const unsigned *>(p.abort)) != 0u) {
                    job_st_agent(reinterpret_cast<unsigned *>(p.abort), 1u);
                    ok = false;
                    break;
                }
                job_pause();
            }
            if (ok)
                for (int c = 0; c < 3; ++c) job_st_agentf(&(*E->epart)[(((size_t)(seq & 1u) * (p.nsteps + 1) + s + 1) * p.ntiles + t.slot) * 3 + c], (float)es[c]);
        }
        if (!ok) {
            job_st_sys(&a.back->exit_seq[a.launch], seq);
            job_st_sys(&a.back->status[a.launch], JOBS_EXIT_ABORT);
            break;
        }
        if (b == 0 && !p.last) {  // the leader's second wave looks for the next job under barrier B
            const int c2 = job_try_fetch(a, seq + 1u, 0);
            pref_seq = c2 != 0 ? seq + 1u : 0u;
            pref_cmd = c2;
        }
        if (!job_barrier(job_flags(a.ctl, 1, p.ntiles), p.ntiles, b, seq, p.max_polls, p.abort, 0)) {
            job_st_agent(reinterpret_cast<unsigned *>(p.abort), 2u);
            job_st_sys(&a.back->status[a.launch], JOBS_EXIT_ABORT);
            break;
        }
        const int nrows = p.nsteps + 1;
        for (int row = b; row < nrows; row += p.ntiles) {
            if (row == 0) continue;  // (the initial state's energies: not part of this emulation)
            for (int c = 0; c < 3; ++c) {
                double sum = 0.0;
                for (int k = 0; k < p.ntiles; ++k) sum += (double)job_ld_agentf(&(*E->epart)[(((size_t)(seq & 1u) * nrows + row) * p.ntiles + k) * 3 + c]);
                (*E->rows)[((size_t)(seq & 1u) * nrows + row) * 3 + c] = sum;
            }
        }
        if (b < nrows) job_st_sys(&a.back->rowdone[b], seq);
        if (b == 0) {
            job_st_sys64(&a.back->t_end[seq & 1u], job_clock());
            if (p.last) job_st_sys(&a.back->status[a.launch], JOBS_EXIT_LAST);
            job_st_sys(&a.back->done, seq);
        }
        if (p.last) break;
        ++seq;
    }
    E->running.fetch_sub(1);
}

static int run_jobs(const char *name, int scenario)
{
    const int n = 131, nsteps = 5, njobs = 4, NW = 8, RF = 2, RB = 2, RP = 2, M = 5;
    const size_t P = (size_t)n * n, N = 12 * P;
    std::vector<float> x(n), sx(n);
    for (int i = 0; i < n; ++i) x[i] = (float)(-15.0 + 30.0 * i / (n - 1));
    wo_build_pml_profile(n, x.data(), 2.0f, 20000.0f, sx.data());
    const float c0 = 1531.0f, dt = 1e-5f, freq = 1000.0f, hdt = 0.5f * dt;
    std::vector<float> G(P, 0.0f);
    for (size_t q = 0; q < P; ++q) {
        const float xx = x[q % n] - 1.0f, yy = x[q / n] + 0.5f;
        G[q] = (float)exp(-(xx * xx + yy * yy) / 0.18);
    }
    // state buffers: frames are not captured here; the last frame alternates between two buffers (wv_ctx::cur2)
    std::vector<float> buf[2], scratch[2];
    buf[0].assign(N, 0.0f);
    buf[1].assign(N, 0.0f);
    scratch[0].assign(N, 0.0f);
    scratch[1].assign(N, 0.0f);
    for (int f = 0; f < 12; ++f)
        for (size_t q = 0; q < P; ++q) {
            const int i = (int)(q % n), j = (int)(q / n), k = f % 6;
            float v = (float)(0.1 * (urand() - 0.5));
            const bool zx = sx[i] == 0.0f, zy = sx[j] == 0.0f;
            if ((k == 3 && zx) || (k == 4 && zy) || (k == 5 && (zx || zy))) v = 0.0f;
            buf[0][(size_t)f * P + q] = v;
        }
    std::vector<float> ref = buf[0];
    HostPlan pl;
    g_force_all = 0;
    if (!plan_build_tiles(pl, n, n, NW * RF, NW * RB, NW * RP, x.data(), x.data(), sx.data(), sx.data(), true, true)) return 1;
    const int nt = (int)pl.tiles.size();
    std::vector<unsigned char> flags(nt, 0);
    for (const TileDesc &t : pl.tiles)
        for (int gy = t.y0 - FT_H; gy < t.y0 + t.oy + FT_H; ++gy)
            for (int gx = t.x0 - FT_H; gx < t.x0 + t.ox + FT_H; ++gx)
                if (gx >= 0 && gx < n && gy >= 0 && gy < n && G[(size_t)gy * n + gx] != 0.0f) flags[t.slot] = 1;
    // the launch's shared memory
    static JobMail mail;   // (static: tens of kilobytes)
    static JobBack back;
    memset((void *)&mail, 0, sizeof(mail));
    memset((void *)&back, 0, sizeof(back));
    std::vector<unsigned long long> ctlmem((job_ctl_bytes(nt) + 7) / 8, 0ull);
    JobCtl *ctl = reinterpret_cast<JobCtl *>(ctlmem.data());
    std::vector<unsigned long long> xch((size_t)2 * XCH_PLANES * 2 * P, 0ull);
    int abort_word[4] = {0, 0, 0, 0};
    std::vector<float> epart((size_t)2 * (nsteps + 1) * nt * 3, 0.0f);
    std::vector<double> rows((size_t)2 * (nsteps + 1) * 3, 0.0);
    // per-job tables (two slots, like the library)
    std::vector<Cyl> table[2];
    std::vector<float> sfac[2];
    std::vector<TileDesc> tiles[2];
    std::vector<int> idx[2];
    std::vector<StepIO> steps[2];
    std::vector<std::thread> th;
    EmuLaunch E;
    E.key = NW * 1000 + RF * 100 + RB * 10 + RP;
    E.epart = &epart;
    E.rows = &rows;
    auto launch = [&](unsigned first, int l) {
        memset((void *)ctl->go, 0, sizeof(ctl->go));
        back.status[l] = JOBS_RUNNING;
        E.args = JobArgs{&mail, &back, ctl, first, scenario == 2 ? 200000u : 100000000u, nt, l};
        E.running = nt;
        for (int b = 0; b < nt; ++b) th.emplace_back(emu_block, &E, b);
    };
    auto join = [&]() {
        for (std::thread &t : th) t.join();
        th.clear();
    };
    unsigned tag_base = 4094, seq = 0;
    int cur = 0, fails = 0, launch_idx = 0, idle_exits = 0, dev_jobs = 0;
    const TileDesc *launch_tiles = nullptr;
    bool alive = false;
    float t_now = 0.002f;
    std::vector<float> want_ic;
    for (int j = 0; j < njobs; ++j) {
        const int slot = j & 1;
        // a fresh design per job (the culling and the launch order change with it), the oracle's result
        std::vector<float> d0(4 * (size_t)M), d1(4 * (size_t)M), tspan(nsteps + 1);
        for (int m = 0; m < M; ++m) {
            d0[4 * m + 0] = (float)(20.0 * (urand() - 0.5));
            d0[4 * m + 1] = (float)(20.0 * (urand() - 0.5));
            d0[4 * m + 2] = (float)(0.5 + 2.0 * urand());
            d0[4 * m + 3] = (float)(600.0 + 2000.0 * urand());
            for (int k = 0; k < 4; ++k) d1[4 * m + k] = d0[4 * m + k] + (k < 3 ? (float)(0.5 * (urand() - 0.5)) : 0.0f);
        }
        for (int s = 0; s <= nsteps; ++s) tspan[s] = (float)((double)t_now + (double)s * 1e-5);
        t_now = tspan[nsteps];
        std::vector<double> eref(3 * (size_t)(nsteps + 1));
        want_ic = ref;
        wo_integrate(n, n, x.data(), x.data(), sx.data(), sx.data(), c0, dt, ref.data(), tspan.data(), nsteps, G.data(), freq, M, d0.data(),
                     d1.data(), tspan[0], tspan[nsteps], eref.data(), nullptr, nullptr, 0, 4);
        table[slot].assign(3 * (size_t)nsteps * M, Cyl{});
        sfac[slot].assign(3 * (size_t)nsteps, 0.0f);
        std::vector<float> tmp(4 * (size_t)M);
        for (int s = 0; s < nsteps; ++s) {
            const float tq[3] = {tspan[s], tspan[s] + hdt, tspan[s] + dt};
            for (int q = 0; q < 3; ++q) {
                sfac[slot][3 * s + q] = wo_source_factor(tq[q], freq);
                wo_design_at(M, d0.data(), d1.data(), tspan[0], tspan[nsteps], tq[q], tmp.data());
                for (int m = 0; m < M; ++m) table[slot][(size_t)(3 * s + q) * M + m] = Cyl{tmp[4 * m], tmp[4 * m + 1], tmp[4 * m + 2] * tmp[4 * m + 2], tmp[4 * m + 3]};
            }
        }
        plan_build_cyl(pl, x.data(), x.data(), table[slot].data(), M, 3 * nsteps, idx[slot], true, nt / 2 + 1);  // (pair order: a per-job permutation)
        tiles[slot] = pl.tiles;
        for (size_t k = 0; k + 1 < tiles[slot].size(); ++k)  // ... and a random one on top: block b's tile differs from job to job
            std::swap(tiles[slot][k], tiles[slot][k + (size_t)(urand() * (double)(tiles[slot].size() - k))]);
        steps[slot].resize(nsteps);
        float *in = buf[cur].data();
        for (int s = 0; s < nsteps; ++s) {
            float *out = s + 1 == nsteps ? buf[cur ^ 1].data() : scratch[s & 1].data();
            steps[slot][s] = StepIO{in, s + 1 == nsteps ? out : nullptr, nullptr, nullptr, nullptr, s, 0};
            in = out;
        }
        FusedParams p{};
        p.nx = n; p.ny = n; p.P = (unsigned)P;
        const float delta = (x[n - 1] - x[0]) / (float)(n - 1), two_d = 2.0f * delta;
        p.ops = Ops{-1.0f / two_d, 1.0f / two_d, -3.0f / two_d, 4.0f / two_d, -1.0f / two_d, 1.0f / two_d, -4.0f / two_d, 3.0f / two_d};
        p.x = x.data(); p.y = x.data(); p.sx = sx.data(); p.sy = sx.data();
        p.c0 = c0; p.c0sq = c0 * c0;
        p.G = G.data(); p.src_flags = flags.data();
        p.sfac_tab = sfac[slot].data(); p.cyl_tab = table[slot].data(); p.M = M;
        p.dt = dt; p.hdt = hdt;
        p.tiles = tiles[slot].data(); p.cyl_idx = idx[slot].data();
        p.steps = steps[slot].data(); p.nsteps = nsteps;
        p.xch = xch.data(); p.xch_bytes = (unsigned)(xch.size() * 8);
        p.tag_base = tag_base; p.reduced = 1;
        p.max_polls = (scenario == 3 && j == 2) ? 1 : (1 << 26);
        p.abort = abort_word;
        p.seq = seq + 1; p.cmd = JOB_RUN; p.last = (scenario == 1 && j == 1) ? 1 : 0;
        p.ntiles = nt; p.ctl = ctl;
        // the API side: describe, ring (scenario 2: the launch has been left waiting past its idle limit before job 2)
        if (scenario == 2 && j == 2 && alive) {
            while (__atomic_load_n(&back.status[launch_idx], __ATOMIC_ACQUIRE) == JOBS_RUNNING) std::this_thread::yield();
            join();
            alive = false;
            if (back.status[launch_idx] != JOBS_EXIT_IDLE) { printf("%-28s expected an idle exit\n", name); return 1; }
            ++idle_exits;
        }
        if (alive && __atomic_load_n(&back.status[launch_idx], __ATOMIC_ACQUIRE) != JOBS_RUNNING) {  // (it left meanwhile: fused_try_resident)
            join();
            alive = false;
            ++idle_exits;
        }
        ++seq;
        JobDesc &desc = mail.desc[seq & 1u];
        if (alive && scenario != 1) {
            // a job for a launch that is already there travels without host tables (FusedDevTables): the tiles evaluate the
            // interpolator and cull themselves, the tile table is the one of the launch's first job
            JobDesc *dj = &ctl->jobs[seq & 1u];
            desc.dsg.M = M; desc.dsg.ti = tspan[0]; desc.dsg.tf = tspan[nsteps];
            design_slopes(desc.dsg, d0.data(), d1.data());
            memcpy(desc.tspan, tspan.data(), (size_t)(nsteps + 1) * sizeof(float));
            memcpy(desc.sfac, sfac[slot].data(), 3 * (size_t)nsteps * sizeof(float));
            p.dsg = &dj->dsg; p.tspan = dj->tspan; p.sfac_tab = dj->sfac;
            p.cyl_tab = nullptr; p.cyl_idx = nullptr; p.dev_cull = 1;
            p.cull_t_lo = tspan[0]; p.cull_t_hi = tspan[nsteps - 1] + dt;
            p.tiles = launch_tiles;
            ++dev_jobs;
        } else {
            launch_tiles = p.tiles;
        }
        desc.p = p;
        __atomic_store_n(&mail.bell, seq, __ATOMIC_RELEASE);
        if (!alive) {
            launch_idx ^= 1;
            launch(seq, launch_idx);
            alive = true;
        }
        tag_base += (unsigned)nsteps;
        // wait for the job as fused_job_wait does
        bool gave_up = false;
        for (;;) {
            bool done = job_reached(__atomic_load_n(&back.done, __ATOMIC_ACQUIRE), seq);
            for (int k = 0; done && k < std::min(nsteps + 1, nt); ++k) done = job_reached(__atomic_load_n(&back.rowdone[k], __ATOMIC_ACQUIRE), seq);
            if (done) break;
            if (E.running.load() == 0) {
                bool d2 = job_reached(__atomic_load_n(&back.done, __ATOMIC_ACQUIRE), seq);
                if (d2) continue;
                gave_up = back.status[launch_idx] == JOBS_EXIT_ABORT;
                if (gave_up) break;
                // it left on its idle limit before it saw this job (the emulated host can be slow): the description and the
                // bell are still there -- a new launch takes over from the first job that is not done (fused_job_wait)
                join();
                ++idle_exits;
                launch_idx ^= 1;
                launch(__atomic_load_n(&back.done, __ATOMIC_ACQUIRE) + 1, launch_idx);
            }
            std::this_thread::yield();
        }
        if (gave_up || !job_reached(back.done, seq)) {
            join();
            alive = false;
            if (scenario != 3 || j != 2) { printf("%-28s job %d was not completed (status %u)\n", name, j, back.status[launch_idx]); return 1; }
            // the give-up: every block has ended (join returned), and the job's initial condition is intact
            size_t bad = 0;
            for (size_t q = 0; q < N; ++q) bad += buf[cur][q] == want_ic[q] ? 0 : 1;
            printf("%-28s job %d given up as forced, launch drained, initial condition %s\n", name, j, bad ? "DAMAGED" : "intact");
            return bad ? 1 : 0;
        }
        cur ^= 1;
        size_t bad = 0;
        for (size_t q = 0; q < N; ++q) bad += buf[cur][q] == ref[q] ? 0 : 1;
        double emax = 0.0;
        for (int s = 1; s <= nsteps; ++s)
            for (int c = 0; c < 3; ++c) {
                const double r = eref[3 * (size_t)s + c];
                emax = fmax(emax, fabs(rows[((size_t)(seq & 1u) * (nsteps + 1) + s) * 3 + c] - r) / (fabs(eref[3 * (size_t)s]) + 1e-300));
            }
        if (bad || emax > 1e-6) {
            printf("%-28s job %d: %zu cells differ, energy rel %.1e\n", name, j, bad, emax);
            ++fails;
        }
        if (p.last) {  // the launch ended with this job
            join();
            alive = false;
            if (back.status[launch_idx] != JOBS_EXIT_LAST) { printf("%-28s expected the launch to end with job %d\n", name, j); return 1; }
        }
    }
    if (alive) {  // tell it to leave (fused_retire) and see that it does
        ++seq;
        FusedParams d{};
        d.seq = seq;
        d.cmd = JOB_EXIT;
        mail.desc[seq & 1u].p = d;
        __atomic_store_n(&mail.bell, seq, __ATOMIC_RELEASE);
        join();
        if (back.status[launch_idx] != JOBS_EXIT_TOLD) { printf("%-28s expected the launch to leave when told (status %u)\n", name, back.status[launch_idx]); return 1; }
    }
    if (scenario == 2 && idle_exits == 0) { printf("%-28s no idle exit happened\n", name); return 1; }
    printf("%-28s %d jobs x %d steps on %d block threads, tile -> block mapping changed per job, %d idle exit(s), %d job(s) without host tables: %s\n",
           name, njobs, nsteps, nt, idle_exits, dev_jobs, fails ? "MISMATCH" : "bit-exact (jobs)");
    return fails;
}

// plan_pair_order's per-call code (only the tiles with cylinders are sorted) against its generic code: the same launch order,
// tile for tile, over grids whose tiles pair up on `cus` compute units and random designs
static int check_pair_order(int n, int cus, int M, int rounds)
{
    std::vector<float> x(n), sx(n, 0.0f);
    for (int i = 0; i < n; ++i) x[i] = (float)(-15.0 + 30.0 * i / (n - 1));
    for (int i = 0; i < n; ++i) {
        const double d = std::min(x[i] + 15.0, 15.0 - (double)x[i]);
        sx[i] = d < 2.0 ? (float)(20000.0 * (2.0 - d) / 2.0) : 0.0f;
    }
    HostPlan pa, pb;
    if (!plan_build_tiles(pa, n, n, 32, 24, 16, x.data(), x.data(), sx.data(), sx.data(), true, true)) return 1;
    pb = pa;
    const int nt = (int)pa.tiles.size();
    int bad = 0;
    std::vector<Cyl> ends((size_t)2 * M);
    std::vector<int> ia, ib;
    for (int r = 0; r < rounds; ++r) {
        const int m_now = r % 5 == 4 ? 0 : M;  // (also: no design at all)
        for (int e = 0; e < 2; ++e)
            for (int m = 0; m < M; ++m) {
                const double rad = 0.2 + 1.8 * urand();
                ends[(size_t)e * M + m] = Cyl{(float)(-12.0 + 24.0 * urand()), (float)(-12.0 + 24.0 * urand()), (float)(rad * rad), 1000.0f};
            }
        g_plan_pair_generic = 0;
        plan_build_cyl(pa, x.data(), x.data(), ends.data(), m_now, 2, ia, true, cus, 0, 1);
        g_plan_pair_generic = 1;
        plan_build_cyl(pb, x.data(), x.data(), ends.data(), m_now, 2, ib, true, cus, 0, 1);
        g_plan_pair_generic = 0;
        if (pa.tiles.size() != pb.tiles.size() || memcmp(pa.tiles.data(), pb.tiles.data(), pa.tiles.size() * sizeof(TileDesc)) != 0 || ia != ib) ++bad;
    }
    printf("%-36s %s (%d tiles on %d CUs, %d designs)\n", "pair order: per-call code = generic", bad ? "MISMATCH" : "ok", nt, cus, rounds);
    return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    std::vector<Case> cases = {
        {"default tiles, design+src", 160, 6, 8, 4, 3, 2, 2.0f, 20000.0f, 6, 1, 0, 0},
        {"aux everywhere (all AUX_ALL)", 96, 5, 8, 4, 3, 2, 2.0f, 20000.0f, 4, 1, 1, 0},
        {"aux everywhere, 260", 260, 3, 8, 4, 3, 2, 2.0f, 20000.0f, 4, 1, 1, 0},
        {"no pml (scale 0), no design", 130, 5, 8, 4, 3, 2, 1.0f, 0.0f, 0, 1, 0, 0},
        {"no src flags", 231, 4, 8, 4, 3, 2, 2.0f, 20000.0f, 5, 1, 0, 0, 0},
        {"RF,RB,RP=3,3,2", 231, 4, 8, 3, 3, 2, 2.0f, 20000.0f, 5, 1, 0, 0},
        {"3,3,3 natural order", 200, 4, 8, 3, 3, 3, 2.0f, 20000.0f, 5, 0, 0, 1},
        {"2,2,2", 131, 4, 8, 2, 2, 2, 2.0f, 20000.0f, 5, 1, 0, 0},
        {"4,2,2", 150, 4, 8, 4, 2, 2, 2.0f, 20000.0f, 3, 1, 0, 0},
        {"NW=4 6,4,4", 117, 4, 4, 6, 4, 4, 3.0f, 20000.0f, 3, 1, 0, 0},
        {"tiny grid 8", 8, 3, 8, 4, 3, 2, 2.0f, 20000.0f, 1, 1, 0, 0},
        {"tiny grid 9, no pml", 9, 3, 8, 4, 3, 2, 2.0f, 0.0f, 1, 1, 0, 0},
        {"grid 57 (two strips)", 57, 4, 8, 4, 3, 2, 2.0f, 20000.0f, 2, 1, 0, 0},
        {"many cylinders", 180, 3, 8, 4, 3, 2, 2.0f, 20000.0f, 40, 1, 0, 0},
        {"resident: design+src", 160, 7, 8, 4, 3, 2, 2.0f, 20000.0f, 6, 1, 0, 0, 1, 1},
        {"resident: aux everywhere", 96, 6, 8, 4, 3, 2, 2.0f, 20000.0f, 4, 1, 1, 0, 1, 1},
        {"resident: no pml, no design", 130, 6, 8, 4, 3, 2, 1.0f, 0.0f, 0, 1, 0, 0, 1, 1},
        {"resident: 2,2,2 many tiles", 131, 9, 8, 2, 2, 2, 2.0f, 20000.0f, 5, 1, 0, 0, 1, 1},
        {"resident: tiny grid 9", 9, 5, 8, 4, 3, 2, 2.0f, 0.0f, 1, 1, 0, 0, 1, 1},
        {"resident: grid 57", 57, 6, 8, 4, 3, 2, 2.0f, 20000.0f, 2, 1, 0, 0, 1, 1},
        {"resident: F_ALL bodies", 150, 5, 8, 4, 3, 2, 2.0f, 20000.0f, 3, 1, 0, 1, 1, 1},
        {"resident: device cylinders", 160, 7, 8, 4, 3, 2, 2.0f, 20000.0f, 6, 1, 0, 0, 1, 1, 1},
        {"resident: device cyl, 2,2,2", 131, 6, 8, 2, 2, 2, 2.0f, 20000.0f, 19, 1, 0, 0, 1, 1, 1},
    };
    if (!quick) {
        cases.push_back({"config-2 like 700, 3 steps", 700, 3, 8, 4, 3, 2, 2.0f, 20000.0f, 19, 1, 0, 0});
        cases.push_back({"wide pml 4.0 at 300", 300, 5, 8, 4, 3, 2, 4.0f, 20000.0f, 8, 1, 0, 0});
        cases.push_back({"thin pml 0.5 at 300", 300, 5, 8, 4, 3, 2, 0.5f, 20000.0f, 8, 1, 0, 0});
        cases.push_back({"600 cylinders (global-list path)", 200, 3, 8, 4, 3, 2, 2.0f, 20000.0f, 600, 1, 0, 0});
        cases.push_back({"resident: config-2 like 700", 700, 4, 8, 4, 3, 2, 2.0f, 20000.0f, 19, 1, 0, 0, 1, 1});
        cases.push_back({"resident: 600 cylinders", 200, 4, 8, 4, 3, 2, 2.0f, 20000.0f, 600, 1, 0, 0, 1, 1});
        cases.push_back({"resident: device cyl, 700", 700, 3, 8, 4, 3, 2, 2.0f, 20000.0f, 19, 1, 0, 0, 1, 1, 1});
    }
    int fails = 0;
    for (const Case &c : cases) fails += run_case(c);
    fails += check_pair_order(700, 256, 19, quick ? 20 : 200);
    fails += check_pair_order(700, 240, 5, quick ? 10 : 60);    // (another split into tiles alone / pairs)
    fails += check_pair_order(420, 100, 30, quick ? 10 : 60);
    fails += run_jobs("jobs: plain sequence", 0);
    fails += run_jobs("jobs: a launch that ends with a job", 1);
    fails += run_jobs("jobs: idle limit, new launch", 2);
    fails += run_jobs("jobs: forced give-up drains", 3);
    printf("%s (%d failing case%s)\n", fails ? "FAIL" : "PASS", fails, fails == 1 ? "" : "s");
    return fails ? 1 : 0;
}
