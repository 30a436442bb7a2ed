// Fused Runge-Kutta step: all four stages of one integration step for one tile, written as barrier-separated PHASES
// that are plain inline functions of (thread id, per-thread register struct, LDS image).  The HIP kernel
// (kernels_fused.hip) runs the phases with __syncthreads() between them; tests/cpu_emu runs the very same functions in
// host loops over the thread ids, which is how the tile/halo/index logic is checked (and ASan-ed) on a machine without
// a GPU.  Reference arithmetic: src/dynamics.jl:9-16 (runge_kutta) around src/dynamics.jl:151-188 (acoustic_dynamics
// for the total and the incident wave set), src/designs.jl:99-116 (speed), src/sources.jl:67-69 (source).
//
// Geometry of a tile
//   region  : 64 columns (one wavefront: lane = x, 256-B coalesced rows) x RY = NW*RPT rows
//   outputs : the region minus a halo of 4 cells on every side (one cell per RK stage): <= 56 x (RY-8) cells
//   thread  : lane l of wave w owns the RPT cells (l, w + NW*r), r = 0..RPT-1 -- rows dealt round-robin so that the
//             shrinking set of rows a stage still needs is spread evenly over the waves
//   stage s : k_s is formed on the region shrunk by s cells; its inputs y_s are needed one cell further out
//   LDS     : the three fields whose NEIGHBOURS a stage reads (W = U + f, Vx, Vy), as (total, incident) pairs so one
//             ds_read_b64 serves both wave sets; Psi_x, Psi_y, Omega, u and the RK accumulator never leave registers
//
// Variants (block-uniform, chosen per tile on the host)
//   FAST <PML=0, EDGE=0> : sigma_x = sigma_y = 0 over the region, region strictly inside the domain, auxiliary fields
//                          zero there (they then stay exactly zero: d(Psi) = b*sigma*(..) = 0).  Six fields instead of
//                          twelve; every dropped term is an exact "+0" / "-0*x" of the reference expression.
//   MID  <PML=1, EDGE=0> : full equations, region strictly inside the domain.
//   GEN  <PML=1, EDGE=1> : full equations, one-sided boundary stencils and the Dirichlet mask.  Valid for any tile.
//   Non-EDGE variants publish P = cp*v instead of v:  cm*v[i-1] + cp*v[i+1]  ==  (cp*v[i+1]) - (cp*v[i-1])  bit for
//   bit because cm == -cp exactly (both are +-1/(2D) rounded) and a + (-b) == a - b; one multiply per cell and field
//   instead of two per derivative.
#pragma once
#include "types.h"

namespace wv {

constexpr int FT_X = 64;         // region width
constexpr int FT_H = 4;          // halo = number of RK stages
constexpr int FT_LX = FT_X + 2;  // LDS row length: one guard cell each side, so lanes 0 / 63 read x-1 / x+1 unbranched

enum : int { VAR_FAST = 0, VAR_MID = 1, VAR_GEN = 2 };

struct TileDesc {
    int x0, y0;      // first output cell (global indices)
    int ox, oy;      // output extent, ox <= 56, oy <= RY - 8
    int variant;     // VAR_*
    int cyl_begin;   // slice of FusedParams::cyl_idx with the cylinders that can touch the region ...
    int cyl_count;   // ... or -1: test all M cylinders
    int slot;        // row of the energy-partial array (natural tile order: the reduction order never depends on
                     // the launch order)
};

struct FusedParams {
    int nx, ny;
    size_t P;
    Ops ops;
    const float *x, *y, *sx, *sy;
    float c0, c0sq;
    const float *u;   // state at the start of the step
    float *out;       // state at the end of the step
    const float *G;   // source shape or nullptr (NoSource)
    float sfac[3];    // sin(2f0*pi*t*freq) at t, t + dt/2, t + dt
    const Cyl *cyl;   // 3 rows of M cylinders: stage times t, t + dt/2, t + dt
    int M;
    float dt, hdt;
    const TileDesc *tiles;
    const int *cyl_idx;
    float *epart;     // [ntiles][3] or nullptr
    float *traj_tot;  // optional copies of the new U_tot / U_inc planes
    float *traj_inc;
    unsigned long long *stamps;  // diagnostic: [ntiles][16] shader-clock stamps per phase, or nullptr (normal runs)
};

// LDS image of one tile: three (RY + 2) x FT_LX arrays of (total, incident) pairs carved out of one raw buffer (the
// kernel instantiates variants with different RY over the same allocation).
struct FusedLds {
    F2 *W, *Vx, *Vy;
};
constexpr int lds_elems(int RY) { return 3 * (RY + 2) * FT_LX; }
WV_HD FusedLds lds_view(F2 *raw, int RY)
{
    return FusedLds{raw, raw + (RY + 2) * FT_LX, raw + 2 * (RY + 2) * FT_LX};
}

WV_HD int lds_at(int lx, int ly) { return (ly + 1) * FT_LX + (lx + 1); }

template <bool PML, int RPT>
struct FusedRegs {
    static constexpr int NS = PML ? 6 : 3;  // fields per wave set: U, Vx, Vy [, Psi_x, Psi_y, Omega]
    float u[RPT][2][NS];    // state at the start of the step
    float acc[RPT][2][NS];  // k1 + 2k2 + 2k3 (+ k4)
    float y[RPT][2][NS];    // input of the current stage; after stage 4 the new state
    float g[RPT];           // source shape at the cell
    float b[RPT];           // c^2 of the total set at the current stage time
    float sy[RPT];          // sigma_y of the row
    float ys[RPT];          // y coordinate of the row
    float sx, xs;           // sigma_x / x coordinate of the column
};

WV_HD int stage_q(int S) { return S == 1 ? 0 : (S == 4 ? 2 : 1); }  // which of the three stage times a stage uses

// speed(design, grid, c0) at one cell from the tile's culled cylinder list (culled cylinders would add an exact 0).
// src/designs.jl:99-116.  No FMA may be formed here (-ffp-contract=off).
WV_HD float tile_speed(const FusedParams &p, const TileDesc &t, int q, float x, float y)
{
    const Cyl *row = p.cyl + (size_t)q * p.M;
    const int n = t.cyl_count < 0 ? p.M : t.cyl_count;
    int count = 0;
    float cd = 0.0f;
    for (int k = 0; k < n; ++k) {
        const int m = t.cyl_count < 0 ? k : p.cyl_idx[t.cyl_begin + k];
        const Cyl c = row[m];
        const float ddx = x - c.px;
        const float ddy = y - c.py;
        const float d2 = ddx * ddx + ddy * ddy;
        const bool in = d2 < c.r2;
        count += in ? 1 : 0;
        cd = cd + (in ? c.c : 0.0f);
    }
    const float C0 = count == 0 ? p.c0 : 0.0f;
    return C0 + cd;
}

// ---- phase 0: global -> registers ---------------------------------------------------------------------------
template <bool PML, bool EDGE, int NW, int RPT>
WV_HD void fused_load(const FusedParams &p, const TileDesc &t, int tid, FusedRegs<PML, RPT> &r)
{
    constexpr int NS = FusedRegs<PML, RPT>::NS;
    const int lane = tid & 63, w = tid >> 6;
    const int gx = t.x0 - FT_H + lane;
    const bool inx = gx >= 0 && gx < p.nx;
    const int cgx = gx < 0 ? 0 : (gx >= p.nx ? p.nx - 1 : gx);
    r.xs = p.x[cgx];
    r.sx = PML ? p.sx[cgx] : 0.0f;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
        const int gy = t.y0 - FT_H + ly;
        const bool in = inx && gy >= 0 && gy < p.ny && ly < t.oy + 2 * FT_H;
        const int cgy = gy < 0 ? 0 : (gy >= p.ny ? p.ny - 1 : gy);
        const size_t id = (size_t)cgy * p.nx + cgx;
        r.ys[rr] = p.y[cgy];
        r.sy[rr] = PML ? p.sy[cgy] : 0.0f;
        r.g[rr] = (p.G && in) ? p.G[id] : 0.0f;
        r.b[rr] = p.c0sq;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const float v = in ? p.u[(size_t)(6 * s + j) * p.P + id] : 0.0f;
                r.u[rr][s][j] = v;
                r.y[rr][s][j] = v;
                r.acc[rr][s][j] = 0.0f;
            }
    }
}

// ---- phase "publish": the stage input's stencil fields -> LDS ------------------------------------------------
template <bool PML, bool EDGE, int NW, int RPT, int S>
WV_HD void fused_publish(const FusedParams &p, const TileDesc &t, int tid, const FusedLds &lds,
                         const FusedRegs<PML, RPT> &r)
{
    const int lane = tid & 63, w = tid >> 6;
    const int rows = t.oy + 2 * FT_H;
    const float sf = p.sfac[stage_q(S)];
    const float cp = p.ops.cp;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
        if (ly < S - 1 || ly >= rows - (S - 1)) continue;  // y_S is only needed on the region shrunk by S-1
        const float f = r.g[rr] * sf;                      // shape .* sin(...)      src/sources.jl:67-69
        const float wt = r.y[rr][0][0] + f;                // U .+ f                 src/dynamics.jl:166-167
        const float wi = r.y[rr][1][0] + f;
        const int i = lds_at(lane, ly);
        if (EDGE) {
            lds.W[i] = F2{wt, wi};
            lds.Vx[i] = F2{r.y[rr][0][1], r.y[rr][1][1]};
            lds.Vy[i] = F2{r.y[rr][0][2], r.y[rr][1][2]};
        } else {
            lds.W[i] = F2{cp * wt, cp * wi};
            lds.Vx[i] = F2{cp * r.y[rr][0][1], cp * r.y[rr][1][1]};
            lds.Vy[i] = F2{cp * r.y[rr][0][2], cp * r.y[rr][1][2]};
        }
    }
}

// `grad * v` along one LDS axis at an edge-aware cell (src/operators.jl:10-22,45-46); st = index stride of the axis
WV_HD F2 edge_deriv(const Ops &o, const F2 *v, int i, int st, int g, int n)
{
    F2 d;
    if (g == 0) {
        d.x = (o.f0 * v[i].x + o.f1 * v[i + st].x) + o.f2 * v[i + 2 * st].x;
        d.y = (o.f0 * v[i].y + o.f1 * v[i + st].y) + o.f2 * v[i + 2 * st].y;
    } else if (g == n - 1) {
        d.x = (o.b0 * v[i - 2 * st].x + o.b1 * v[i - st].x) + o.b2 * v[i].x;
        d.y = (o.b0 * v[i - 2 * st].y + o.b1 * v[i - st].y) + o.b2 * v[i].y;
    } else {
        d.x = o.cm * v[i - st].x + o.cp * v[i + st].x;
        d.y = o.cm * v[i - st].y + o.cp * v[i + st].y;
    }
    return d;
}

// ---- phase "compute": k_S from the LDS image, then the RK update of the registers ----------------------------
template <bool PML, bool EDGE, int NW, int RPT, int S>
WV_HD void fused_compute(const FusedParams &p, const TileDesc &t, int tid, const FusedLds &lds,
                         FusedRegs<PML, RPT> &r)
{
    constexpr int NS = FusedRegs<PML, RPT>::NS;
    const int lane = tid & 63, w = tid >> 6;
    const int rows = t.oy + 2 * FT_H;
    const int gx = t.x0 - FT_H + lane;
    const bool has_cyl = p.M > 0 && t.cyl_count != 0;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
        if (ly < S || ly >= rows - S) continue;  // k_S is only needed on the region shrunk by S
        const int gy = t.y0 - FT_H + ly;
        if (gy < 0 || gy >= p.ny) continue;
        const int i = lds_at(lane, ly);
        F2 Ux, Uy, Vxx, Vyy;
        if (EDGE) {
            // lanes outside the domain only produce values nobody reads; keep their LDS indices inside the row
            const int gxe = (gx < 0 || gx >= p.nx) ? 1 : gx;
            Ux = edge_deriv(p.ops, lds.W, i, 1, gxe, p.nx);
            Uy = edge_deriv(p.ops, lds.W, i, FT_LX, gy, p.ny);
            Vxx = edge_deriv(p.ops, lds.Vx, i, 1, gxe, p.nx);
            Vyy = edge_deriv(p.ops, lds.Vy, i, FT_LX, gy, p.ny);
        } else {
            const F2 Wl = lds.W[i - 1], Wr = lds.W[i + 1], Wd = lds.W[i - FT_LX], Wu = lds.W[i + FT_LX];
            const F2 Xl = lds.Vx[i - 1], Xr = lds.Vx[i + 1], Yd = lds.Vy[i - FT_LX], Yu = lds.Vy[i + FT_LX];
            Ux = F2{Wr.x - Wl.x, Wr.y - Wl.y};
            Uy = F2{Wu.x - Wd.x, Wu.y - Wd.y};
            Vxx = F2{Xr.x - Xl.x, Xr.y - Xl.y};
            Vyy = F2{Yu.x - Yd.x, Yu.y - Yd.y};
        }
        if (S != 3 && has_cyl) {  // stage 3 shares t + dt/2 with stage 2
            const float c = tile_speed(p, t, stage_q(S), r.xs, r.ys[rr]);  // C(t)   src/env.jl:99
            r.b[rr] = c * c;                                               // c .^ 2 src/dynamics.jl:159
        }
        float k[2][NS];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float b = s == 0 ? r.b[rr] : p.c0sq;
            const float ux = s == 0 ? Ux.x : Ux.y, uy = s == 0 ? Uy.x : Uy.y;
            const float vxx = s == 0 ? Vxx.x : Vxx.y, vyy = s == 0 ? Vyy.x : Vyy.y;
            if (PML) {
                const float sx = r.sx, sy = r.sy[rr];
                const float U = r.y[rr][s][0];
                // src/dynamics.jl:169-174, left to right as written
                float dU = (((b * (vxx + vyy) + r.y[rr][s][3]) + r.y[rr][s][4]) - (sx + sy) * U) - r.y[rr][s][5];
                if (EDGE) {
                    const float bcv = (gx <= 0 || gy == 0 || gx >= p.nx - 1 || gy == p.ny - 1) ? 0.0f : 1.0f;
                    dU = bcv * dU;  // bc .* dU   src/dynamics.jl:176, src/dims.jl:117-124
                }
                k[s][0] = dU;
                k[s][1] = ux - sx * r.y[rr][s][1];
                k[s][2] = uy - sy * r.y[rr][s][2];
                k[s][3] = (b * sx) * vyy;
                k[s][4] = (b * sy) * vxx;
                k[s][5] = (sx * sy) * U;
            } else {
                k[s][0] = b * (vxx + vyy);
                k[s][1] = ux;
                k[s][2] = uy;
            }
        }
        // src/dynamics.jl:9-16 and :41
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const float kk = k[s][j];
                if (S == 1) {
                    r.acc[rr][s][j] = kk;
                    r.y[rr][s][j] = r.u[rr][s][j] + p.hdt * kk;   // u .+ 0.5f0*dt*k1
                } else if (S == 2) {
                    r.acc[rr][s][j] = __builtin_fmaf(2.0f, kk, r.acc[rr][s][j]);  // 2*k exact: == acc + 2*k
                    r.y[rr][s][j] = r.u[rr][s][j] + p.hdt * kk;   // u .+ 0.5f0*dt*k2
                } else if (S == 3) {
                    r.acc[rr][s][j] = __builtin_fmaf(2.0f, kk, r.acc[rr][s][j]);
                    r.y[rr][s][j] = r.u[rr][s][j] + p.dt * kk;    // u .+ dt*k3
                } else {
                    const float du = ((1.0f / 6.0f) * (r.acc[rr][s][j] + kk)) * p.dt;  // (1/6f0*(...))*dt
                    r.y[rr][s][j] = r.u[rr][s][j] + du;                                 // _u .+ du
                }
            }
    }
}

// ---- last phase: registers -> global, energy terms of src/env.jl:105-111 --------------------------------------
template <bool PML, bool EDGE, int NW, int RPT>
WV_HD void fused_store(const FusedParams &p, const TileDesc &t, int tid, const FusedRegs<PML, RPT> &r, float e[3])
{
    constexpr int NS = FusedRegs<PML, RPT>::NS;
    const int lane = tid & 63, w = tid >> 6;
    e[0] = e[1] = e[2] = 0.0f;
    if (lane < FT_H || lane >= FT_H + t.ox) return;
    const int gx = t.x0 - FT_H + lane;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
        if (ly < FT_H || ly >= FT_H + t.oy) continue;
        const int gy = t.y0 - FT_H + ly;
        const size_t id = (size_t)gy * p.nx + gx;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < NS; ++j) p.out[(size_t)(6 * s + j) * p.P + id] = r.y[rr][s][j];
        const float ut = r.y[rr][0][0], ui = r.y[rr][1][0], us = ut - ui;
        e[0] += ut * ut;
        e[1] += ui * ui;
        e[2] += us * us;
        if (p.traj_tot) p.traj_tot[id] = ut;
        if (p.traj_inc) p.traj_inc[id] = ui;
    }
}

}  // namespace wv
