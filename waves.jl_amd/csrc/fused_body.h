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
//   x nbrs  : lane = x, so the x-neighbours of a cell sit in the adjacent lanes of the same wave: they are fetched with
//             DPP wave shifts (v_mov/v_sub .. wave_shr:1 / wave_shl:1), never through LDS
//   LDS     : only the two fields whose Y-neighbours a stage reads (W = U + f and Vy), as (total, incident) pairs so one
//             ds_read_b64 serves both wave sets, double-buffered so that a stage's results are published for the next
//             stage while slower waves still read the current one (one barrier per stage); everything else (u, the RK
//             accumulator, Vx, the PML auxiliaries) stays in registers for the whole step
//
// Field sets (template parameter AUX, block-uniform, chosen per tile on the host).  A PML auxiliary field whose
// damping coefficient is zero over the tile's region stays exactly zero when it starts at zero
// (dPsi_x = b*sigma_x*Vyy, dPsi_y = b*sigma_y*Vxx, dOmega = sigma_x*sigma_y*U), so a tile only carries the fields that
// can be non-zero there; every term dropped from src/dynamics.jl:169-174 is an exact "+0", "-0*x" or "sigma+0":
//   AUX_NONE  sigma_x = sigma_y = 0 : U, Vx, Vy                      (6 of 12 planes)   the interior: ~70 % of 700^2
//   AUX_PX    sigma_y = 0           : U, Vx, Vy, Psi_x               (8 planes)         left / right PML bands
//   AUX_PY    sigma_x = 0           : U, Vx, Vy, Psi_y               (8 planes)         bottom / top PML bands
//   AUX_ALL                         : all six                        (12 planes)        corners; always valid
//
// Derivatives.  LDS holds P = cp*v, not v:  cm*v[i-1] + cp*v[i+1]  ==  (cp*v[i+1]) - (cp*v[i-1])  bit for bit, because
// cm == -cp exactly (both are +-1/(2D) rounded) and a + (-b) == a - b: one multiply per cell and field instead of two
// per derivative.  Tiles that touch the domain boundary additionally publish the RAW values of the three cells next
// to that boundary into small side buffers, from which the boundary cell forms the reference's one-sided stencil
// (src/operators.jl:14-15) and then applies the Dirichlet mask (src/dims.jl:117-124).
#pragma once
#include <stdint.h>

#include "types.h"

namespace wv {

constexpr int FT_X = 64;         // region width
constexpr int FT_H = 4;          // halo = number of RK stages
constexpr int FT_LX = FT_X;      // LDS row length (x-neighbours never come from LDS, so no guard columns)

enum : int { AUX_NONE = 0, AUX_PX = 1, AUX_PY = 2, AUX_ALL = 3 };
enum : int { EDGE_L = 1, EDGE_R = 2, EDGE_T = 4, EDGE_B = 8 };  // region contains gx = 0 / nx-1 / gy = 0 / ny-1

struct TileDesc {
    int x0, y0;      // first output cell (global indices)
    int ox, oy;      // output extent, ox <= 56, oy <= RY - 8
    int aux;         // AUX_*
    int edge;        // EDGE_* bits
    int cyl_begin;   // slice of FusedParams::cyl_idx with the cylinders that can touch the region ...
    int cyl_count;   // ... or -1: test all M cylinders
    int slot;        // natural tile index: row of the energy-partial array and of src_flags (the reduction order
                     // never depends on the launch order)
    int pad;
};

// DesignInterpolator(initial, final, ti, tf) as the data it is made of (src/designs.jl:274-292): M <= FT_MAXCYL cylinders as
// {px, py, r, c}.  With it (FusedParams::dsg) the tiles evaluate their cylinders at the stage times themselves -- design_cyl
// below, the host's design_at (api.hip) operation for operation -- instead of reading them from a [nsteps][3][M] table the
// host has built and uploaded for every call (91 KB per 100-step action at M = 19).
constexpr int FT_MAXCYL = 32;    // cylinders of one tile staged in LDS (more: read from global memory)
struct JobDesign {
    int M;
    float ti, tf;
    int pad;
    alignas(16) float d0[FT_MAXCYL * 4];     // v_i
    alignas(16) float slope[FT_MAXCYL * 4];  // (v_f + (-1f0*v_i)) * (1f0/Dt), formed by design_slopes (on the host, once per call)
};

// DesignInterpolator call at one time for one cylinder: src/designs.jl:287-292 with the algebra of :47-53, per scalar
// component  v_i + ((v_f + (-1f0*v_i)) * (1f0/Dt)) * (clamp(t, ti, tf) - ti),  then r .^ 2 (src/designs.jl:102).  The same
// operations in the same order as the host's design_at (api.hip), fp32, no contraction, correctly rounded division: the
// same bits.
// (the time-independent half of the formula, once per call and cylinder instead of once per step, stage and tile)
WV_HD void design_slopes(JobDesign &d, const float *d0, const float *d1)
{
    float dt = d.tf - d.ti;
    dt = dt > 0.0f ? dt : 1.0f;
    const float inv_dt = 1.0f / dt;
    for (int k = 0; k < 4 * d.M; ++k) {
        const float vi = d0[k], vf = d1[k];
        const float dy = vf + (-1.0f * vi);
        d.d0[k] = vi;
        d.slope[k] = dy * inv_dt;
    }
}
WV_HD float design_tau(const JobDesign &d, float t)
{
    const float tc = t < d.ti ? d.ti : (t > d.tf ? d.tf : t);
    return tc - d.ti;
}
WV_HD Cyl design_cyl_tau(const JobDesign &d, int m, float tau)
{
    const Cyl a = reinterpret_cast<const Cyl *>(d.d0)[m], k = reinterpret_cast<const Cyl *>(d.slope)[m];  // (one 16-byte load each)
    const float r = a.r2 + k.r2 * tau;  // (third component of a design entry: the radius)
    return Cyl{a.px + k.px * tau, a.py + k.py * tau, r * r, a.c + k.c * tau};
}
WV_HD Cyl design_cyl(const JobDesign &d, int m, float t) { return design_cyl_tau(d, m, design_tau(d, t)); }

// What differs between the steps of one wv_integrate call.
struct StepIO {
    const float *u;   // state at the start of the step (12 planes)
    float *out;       // state at the end of the step (12 planes, != u); k_steps_resident: nullptr = not wanted in memory
    float *epart;     // per-tile energy partials [ntiles][3] or nullptr
    float *traj_tot;  // optional copies of the new U_tot / U_inc planes
    float *traj_inc;
    int step;         // integration step of this call, 0-based: row of sfac_tab / cyl_tab
    int pad;
};

struct FusedParams {
    int nx, ny;
    unsigned P;       // nx*ny (12*P < 2^31, checked at create: 32-bit element offsets everywhere)
    Ops ops;
    const float *x, *y, *sx, *sy;
    float c0, c0sq;
    const float *G;   // source shape or nullptr (NoSource)
    const unsigned char *src_flags;  // per tile slot: shape != 0 somewhere in the region (nullptr: assume yes)
    // Per-step scalars live in device tables indexed by the step, so that the kernel arguments of a given step are the
    // same for every wv_integrate call of the same shape (a captured hipGraph can then be replayed unchanged).
    const float *sfac_tab;  // [nsteps][3]  sin(2f0*pi*t*freq) at t, t + dt/2, t + dt
    const Cyl *cyl_tab;     // [nsteps][3][M] cylinders at the three stage times
    int M;
    float dt, hdt;
    const TileDesc *tiles;
    int tile_offset;  // first tile of this launch (a launch covers one band of tiles)
    const int *cyl_idx;
    StepIO io;        // k_step_fused: the one step of this launch
    // k_steps_resident: all steps of the call in one launch, every tile resident for the whole call
    const StepIO *steps;       // [nsteps]; steps[s].u == steps[s-1].out
    int nsteps;
    unsigned long long *xch;   // halo exchange: [2 parities][4 planes][P cells] 16-byte granules {3 values, tag}, see fused_xch_*
    unsigned xch_bytes;        // its size
    unsigned tag_base;         // the border cells of step s carry the tag tag_base + s + 1
    int *abort;                // set when a wait gave up: every tile then leaves the kernel
    int reduced;               // the tiles use reduced field sets (auxiliary fields are zero outside the PML)
    int max_polls;             // a wave gives up (and the launch drains) after this many polls of one halo
    unsigned long long *stamps;  // diagnostic: [ntiles][16] shader-clock stamps per phase, or nullptr (normal runs)
    // k_steps_resident as a JOB of the action-outliving launch (see "jobs" below); all zero / nullptr for k_step_fused
    unsigned seq;              // number of this job (1, 2, ...: never 0)
    int cmd;                   // JOB_RUN, or JOB_EXIT: the host retires the launch
    int last;                  // the launch ends after this job (profiling, diagnostics, contexts that may not idle on the device)
    int ntiles;                // blocks of the launch == tiles of the plan
    // the second pass of the energy sums (k_energy_final's arithmetic, done by the tiles themselves after the job's last step)
    const float *ef_row0;      // partial sums of the initial state [ntiles][3] (the previous job's last row, or k_energy_partial's)
    const float *ef_epart;     // [nsteps + 1][ntiles][3]; row 0 unused
    float *ef_signal;          // pinned HOST memory [(nsteps + 1)][3], or nullptr: no trace wanted
    float ef_dOmega;
    int pad2;
    struct JobCtl *ctl;        // the launch's control block in device memory (flag arrays of the barriers)
    struct JobBack *back;      // diagnostic (WAVES_AMD_JOBLOG): where the leader tile stamps the phases of the job, or nullptr
    // cylinders evaluated by the tiles themselves (cyl_tab == nullptr then): the interpolator and the tabulated times
    const JobDesign *dsg;
    const float *tspan;        // [nsteps + 1]
    int dev_cull;              // the tiles also find out themselves which cylinders can reach them (device_cull_keep; cyl_idx unused)
    float cull_t_lo, cull_t_hi;  // the earliest / the latest stage time of the call
    // state(env) of the frames this job leaves (src/env.jl:132-137), produced by the job itself behind its end-of-job barrier
    // (the last JOB_OBS_BLOCKS blocks of the launch order, k_observation's arithmetic: common.h obs_pixel): a caller that looks
    // at the state in front of every action then needs no kernel of its own for it.  ob_out: PINNED HOST memory
    // [4][ob_ry][ob_rx] or nullptr (not wanted); ob_f0..2: the three frames as they will be after the job, ob_G the source shape or nullptr.
    const float *ob_f0, *ob_f1, *ob_f2, *ob_G;
    float *ob_out;
    int ob_rx, ob_ry;
};
constexpr int JOB_OBS_BLOCKS = 128;  // blocks that share the observation of a job (65 536 elements at 128 x 128: one per thread)

// ---- jobs: the resident launch that outlives the action ---------------------------------------------------------------
// One launch of k_steps_resident serves a SEQUENCE of wv_integrate calls ("jobs").  The host describes job `seq` in pinned
// memory (JobMail::desc[seq & 1]) and rings JobMail::bell = seq with plain CPU stores -- no HIP call; block 0 (the leader)
// polls the bell, copies the description into device memory (JobCtl::jobs[seq & 1]) and releases the other blocks through
// JobCtl::go[seq & 1].  Every block then runs the job exactly as a single launch would (its tile of THIS job: the tile
// table, the culled cylinder lists and the launch order may differ from job to job, so the state travels through memory
// between jobs: final-state stores of job k, the end-of-job barrier, state loads of job k + 1), joins the end-of-job
// barrier, reduces its share of the energy-trace rows into pinned host memory and reports through JobBack.  What a job
// saves over a launch: the start-up ramp of ~490 workgroups (38 us at 700^2), the kernel-to-kernel gap (41 us) and every
// HIP API call on the host's path.  The launch leaves when told to (JOB_EXIT / FusedParams::last), when a tile gives up
// (abort word), or when no bell has rung for `idle_ticks` -- it never waits unboundedly for the host.
enum : int { JOB_RUN = 1, JOB_EXIT = 2 };
constexpr int JOB_MAX_TILES = 2048;  // (grids of the resident kernel: at most the block slots of the device, 512 on MI355X)
enum : unsigned { JOBS_RUNNING = 0, JOBS_EXIT_TOLD = 1, JOBS_EXIT_IDLE = 2, JOBS_EXIT_ABORT = 3, JOBS_EXIT_LAST = 4 };

constexpr int JOB_MAXSTEPS = 256;  // steps of a call whose per-step tables travel inside the job description
struct JobDesc {                 // everything a job reads that differs from call to call and is small
    FusedParams p;
    // the tables of a call whose tiles evaluate and cull their cylinders themselves (p.dsg / p.tspan / p.sfac_tab then point at
    // the device copy of these): no table is built, uploaded or waited for on the host's path
    JobDesign dsg;
    float tspan[JOB_MAXSTEPS + 1];
    float sfac[3 * JOB_MAXSTEPS];
};
struct JobMail {                 // pinned host memory; written by the host, read by the leader block
    unsigned bell;               // number of the newest job described (monotonic)
    unsigned pad[15];
    JobDesc desc[2];             // desc[seq & 1] describes job `seq` once bell >= seq
};
struct JobBack {                 // pinned host memory; written by the device
    unsigned done;               // number of the newest job whose outputs (state, frames, trace rows) are complete
    unsigned status[2];          // [launch]: JOBS_*: why that launch has left (JOBS_RUNNING while it has not)
    unsigned exit_seq[2];        // [launch]: the job it was waiting for / working on when it left
    unsigned pad[11];
    unsigned long long t_begin[2], t_end[2];  // [seq & 1]: 100 MHz device clock when the leader saw the job / saw it complete
    unsigned long long phase[2][8];           // diagnostic (FusedParams::back): the leader tile's clock at the phases of the job
    unsigned rowdone[JOB_MAX_TILES];          // [block]: number of the newest job whose trace rows of this block are in host memory
};
struct JobGo {
    unsigned seq, cmd;           // written as ONE 8-byte word
};
struct JobCtl {                  // device memory
    JobGo go[2];                 // [seq & 1]
    unsigned pad[12];
    JobDesc jobs[2];             // the leader's copies of JobMail::desc
    // followed by flag arrays of `ntiles` words each (job_flags): 0 = A "reached the end of the last step" (nobody has given
    // up: the final state may now replace the initial condition), 1 = B "all stores of the job have left"
};
WV_HD unsigned *job_flags(JobCtl *c, int which, int ntiles) { return reinterpret_cast<unsigned *>(c + 1) + (size_t)which * (size_t)((ntiles + 63) & ~63); }
constexpr size_t job_ctl_bytes(int ntiles) { return sizeof(JobCtl) + 2 * (size_t)((ntiles + 63) & ~63) * sizeof(unsigned); }

// kernel arguments of k_steps_resident
struct JobArgs {
    const JobMail *mail;
    JobBack *back;
    JobCtl *ctl;
    unsigned first_seq;          // the first job this launch serves
    unsigned idle_ticks;         // the leader leaves after this many 100 MHz ticks without a new bell
    int ntiles;
    int launch;                  // which of the two status words of JobBack is this launch's
};

// LDS image of one tile, carved out of one raw buffer (the kernel instantiates field sets with different RY over the
// same allocation): two buffers of two (RY + 2) x FT_LX arrays of (total, incident) pairs (one guard row above and
// below), then the boundary side buffers and the staged cylinders.
struct FusedLds {
    F2 *W[2], *Vy[2];    // [stage parity]: stage S reads buffer (S-1)&1 and publishes stage S+1 into buffer S&1
    F2 *XL[2], *XR[2];   // [parity][row][3 cells][W, Vx]   raw values at gx = 0,1,2 / nx-3,nx-2,nx-1
    F2 *YT[2], *YB[2];   // [parity][3 rows][lane][W, Vy]   raw values at gy = 0,1,2 / ny-3,ny-2,ny-1
    Cyl *cyl;            // [3 stage times][cyl_count] the tile's culled cylinders
};
constexpr int lds_main_elems(int RY) { return 2 * (RY + 2) * FT_LX; }  // one buffer
constexpr int lds_side_elems(int RYMAX) { return 2 * RYMAX * 6 + 2 * 3 * FT_X * 2; }
constexpr int lds_elems(int RYMAX) { return 2 * lds_main_elems(RYMAX) + lds_side_elems(RYMAX) + 3 * FT_MAXCYL * 2; }
// Boundary tiles are never the tallest ones, so their two (smaller) main buffers leave room for a second set of side
// buffers inside the same allocation: the side copies are then double-buffered like the main arrays and a boundary
// tile needs no extra barrier.  (When a configuration gives boundary tiles the full height, they fall back to one set
// plus a barrier between the two halves of a stage.)
constexpr bool lds_side_double(int RY, int RYMAX) { return 2 * lds_main_elems(RY) + lds_side_elems(RYMAX) <= 2 * lds_main_elems(RYMAX); }
WV_HD FusedLds lds_view(F2 *raw, int RY, int RYMAX)
{
    F2 *b1 = raw + lds_main_elems(RY);
    F2 *side0 = raw + 2 * lds_main_elems(RYMAX);
    F2 *side1 = lds_side_double(RY, RYMAX) ? raw + 2 * lds_main_elems(RY) : side0;
    FusedLds l;
    l.W[0] = raw;
    l.Vy[0] = raw + (RY + 2) * FT_LX;
    l.W[1] = b1;
    l.Vy[1] = b1 + (RY + 2) * FT_LX;
    F2 *sd[2] = {side0, side1};
    for (int k = 0; k < 2; ++k) {
        l.XL[k] = sd[k];
        l.XR[k] = sd[k] + RYMAX * 6;
        l.YT[k] = sd[k] + 2 * RYMAX * 6;
        l.YB[k] = sd[k] + 2 * RYMAX * 6 + 3 * FT_X * 2;
    }
    l.cyl = reinterpret_cast<Cyl *>(side0 + lds_side_elems(RYMAX));
    return l;
}

WV_HD int lds_at(int lx, int ly) { return (ly + 1) * FT_LX + lx; }

constexpr int aux_ns(int AUX) { return AUX == AUX_NONE ? 3 : (AUX == AUX_ALL ? 6 : 4); }
// state plane (within one wave set) of local field j
constexpr int aux_plane(int AUX, int j) { return j < 3 ? j : (AUX == AUX_PX ? 3 : (AUX == AUX_PY ? 4 : j)); }

template <int AUX, int RPT>
struct FusedRegs {
    static constexpr int NS = aux_ns(AUX);  // fields per wave set: U, Vx, Vy [, Psi_x | Psi_y | Psi_x, Psi_y, Omega]
    float u[RPT][2][NS];    // state at the start of the step
    float acc[RPT][2][NS];  // k1 + 2k2 + 2k3
    float y[RPT][2][NS];    // input of stages 2..4; after stage 4 the new state
    float g[RPT];           // source shape at the cell
    float px[RPT][4];       // cp*W (total, incident), cp*Vx (total, incident) of the current stage input: what the
                            // x-neighbour lanes read through DPP
    float sx;               // sigma_x of the column
    float xs;               // x coordinate of the column (tiles with cylinders only)
    float bsq[3][RPT];      // c^2 of the total set at the three stage times of the step (variants with F_CYL only)
    int cidx;               // threads 0 .. 3*cyl_count-1 stage the tile's cylinders in LDS: table column of this one
};

WV_HD int stage_q(int S) { return S == 1 ? 0 : (S == 4 ? 2 : 1); }  // which of the three stage times a stage uses

// Compile-time feature flags of a tile body (template parameter FL).  A flag that is compiled in but not needed by the
// tile at hand is harmless (every use is still guarded by the tile's run-time data); a flag that is NOT compiled in
// removes the code -- and its scalar branches -- altogether.  The kernel picks the smallest instantiated superset.
enum : int { F_EL = 1, F_ER = 2, F_ET = 4, F_EB = 8, F_EDGE = 15, F_CYL = 16, F_SRC = 32, F_ALL = 63 };  // F_E* == EDGE_*

// Block-uniform values fetched once per tile (they live in SGPRs): re-reading them from memory in every phase costs a
// dependent load per phase on a path whose length is what bounds the kernel.
struct TileCtx {
    int step;      // row of the per-step tables
    float sf[3];   // sin(2f0*pi*t*freq) at the three stage times of this step (0 without a source)
    bool has_src;  // the source shape is non-zero somewhere in this tile's region
    bool has_cyl;  // at least one cylinder can reach this tile's region
    bool cyl_lds;  // ... and the culled list is staged in LDS
    float tau[3];  // tiles that evaluate their cylinders themselves: clamp(t, ti, tf) - ti at the three stage times of the step
                   // whose cylinders are fetched next (fused_cyl_times: formed early in the step before, from scalar loads
                   // whose latency the step hides -- formed next to the fetch they delayed every halo poll by ~0.5 us)
};

WV_HD bool tile_has_src(const FusedParams &p, const TileDesc &t)
{
    return p.G != nullptr && (p.src_flags == nullptr || p.src_flags[t.slot] != 0);
}

WV_HD int tile_flags(const FusedParams &p, const TileDesc &t)
{
    return (t.edge & F_EDGE) | ((p.M > 0 && t.cyl_count != 0) ? F_CYL : 0) | (tile_has_src(p, t) ? F_SRC : 0);
}

// speed(design, grid, c0) at one cell from the tile's culled cylinder list (culled cylinders would add an exact 0).
// src/designs.jl:99-116.  No FMA may be formed here (-ffp-contract=off).  The list is read from the LDS copy made by
// fused_load (one broadcast ds_read_b128 per cylinder) or, for tiles with more than FT_MAXCYL cylinders, from global.
WV_HD void speed_accum(const Cyl c, float x, float y, int &count, float &cd)
{
    const float ddx = x - c.px;
    const float ddy = y - c.py;
    const float d2 = ddx * ddx + ddy * ddy;
    const bool in = d2 < c.r2;
    count += in ? 1 : 0;
    cd = cd + (in ? c.c : 0.0f);
}

// c^2 of the total set for the RPT rows of one thread at the three stage times of a step.  The loop over the tile's
// cylinders is the OUTER loop and handles the three stage times together (three broadcast LDS reads in flight per
// iteration instead of one per row and time); each (time, row) still accumulates its cylinders in ascending order,
// exactly like sum(mask .* c, dims = 3).
// Two loops, one per home of the list: as one loop with the choice inside, the compiler merged the LDS and the global
// pointer into one generic pointer -- flat loads, and a wait for EVERY outstanding memory access in each iteration.
// (Measured and not kept, round 3: ONE squared distance per cell for the three stage times of a cylinder whose centre does not
// move -- a third fewer instructions per cylinder, bit-exact -- and the rows' y as scalar loads: -0.3 % at 700^2, but +2.7 % and
// +1.3 % at 256^2, where the wave-uniform test and the scalar loads' latency lengthen a step that is nothing but latency.
// What cylinders cost at 700^2 is not this arithmetic at all (skipping it altogether: -0.2 %) but the balance of the CU pairs,
// DESIGN 6 "what the cylinders cost".)
template <int NW, int RPT>
WV_HD void tile_speed_sq(const FusedParams &p, const TileDesc &t, const TileCtx &cx, const FusedLds &lds, int w, float x,
                         float bsq[3][RPT])
{
    int count[3][RPT];
    float cd[3][RPT], ys[RPT];
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int gy = t.y0 - FT_H + w + NW * rr;
        ys[rr] = p.y[gy < 0 ? 0 : (gy >= p.ny ? p.ny - 1 : gy)];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            count[q][rr] = 0;
            cd[q][rr] = 0.0f;
        }
    }
    const int n = t.cyl_count < 0 ? p.M : t.cyl_count;
    if (cx.cyl_lds) {
        for (int k = 0; k < n; ++k) {
            Cyl c[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) c[q] = lds.cyl[q * t.cyl_count + k];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int rr = 0; rr < RPT; ++rr) speed_accum(c[q], x, ys[rr], count[q][rr], cd[q][rr]);
        }
    } else {
        for (int k = 0; k < n; ++k) {
            Cyl c[3];
            const int col = t.cyl_count < 0 ? k : p.cyl_idx[t.cyl_begin + k];
#pragma unroll
            for (int q = 0; q < 3; ++q) c[q] = p.cyl_tab[(size_t)(3 * cx.step + q) * p.M + col];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int rr = 0; rr < RPT; ++rr) speed_accum(c[q], x, ys[rr], count[q][rr], cd[q][rr]);
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
            const float c = (count[q][rr] == 0 ? p.c0 : 0.0f) + cd[q][rr];  // C(t)   src/env.jl:99, src/designs.jl:110-116
            bsq[q][rr] = c * c;                                                // c .^ 2 src/dynamics.jl:159
        }
}

// Value of `v` in the lane to the left / right of this one (the cell at x-1 / x+1 of the same row).  On the device `nb`
// is the thread's own register struct and the value moves by a DPP wave shift (lane 0 / 63 receive 0: those lanes are
// halo cells whose results nobody reads); in the CPU emulation `nb` points at lane 0 of the wave's 64 register structs.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float wv_dpp_from_left(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float wv_dpp_from_right(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
#endif

template <class R>
WV_HD float px_left(const R *nb, int lane, int rr, int k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return wv_dpp_from_left(nb->px[rr][k]);
#else
    return lane > 0 ? nb[lane - 1].px[rr][k] : 0.0f;
#endif
}
template <class R>
WV_HD float px_right(const R *nb, int lane, int rr, int k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return wv_dpp_from_right(nb->px[rr][k]);
#else
    return lane < 63 ? nb[lane + 1].px[rr][k] : 0.0f;
#endif
}

// index of the thread's wave in its block.  Wave-uniform by construction, and said so (WV_UNIFORM_ROWS): the row tests of a
// wave (ly = w + NW * rr against the stage's row range) then are scalar compares and branches instead of v_cmp +
// exec-mask regions -- whose "row skipped" side needs a value for everything the region assigns
WV_HD int wv_wave_of(int tid)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(WV_UNIFORM_ROWS)
    return __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    return tid >> 6;
#endif
}
// ---- phase 0a: what a tile keeps for all its steps -------------------------------------------------------------
// Which cylinders can reach a tile's region at ANY stage time of the call: the device's version of plan_build_cyl
// (fused_plan.h), formula for formula and in double like it, so that host and device arrive at the same lists.  The
// interpolation is linear in a clamped time and rounded monotonically, so every component takes its extremes at the
// earliest and the latest stage time.  Conservative (margins far above fp32 round-off): a cylinder that is dropped has
// mask == false at every cell of the region, so dropping it is exact.
WV_HD bool device_cull_keep(const FusedParams &p, const TileDesc &t, int m)
{
    const Cyl a = design_cyl(*p.dsg, m, p.cull_t_lo), b = design_cyl(*p.dsg, m, p.cull_t_hi);
    const bool ok = a.px - a.px == 0.0f && a.py - a.py == 0.0f && a.r2 - a.r2 == 0.0f && b.px - b.px == 0.0f && b.py - b.py == 0.0f &&
                    b.r2 - b.r2 == 0.0f;  // all finite
    if (!ok) return true;
    const double pxmin = a.px < b.px ? (double)a.px : (double)b.px, pxmax = a.px < b.px ? (double)b.px : (double)a.px;
    const double pymin = a.py < b.py ? (double)a.py : (double)b.py, pymax = a.py < b.py ? (double)b.py : (double)a.py;
    double r2 = 0.0;
    r2 = (double)a.r2 > r2 ? (double)a.r2 : r2;
    r2 = (double)b.r2 > r2 ? (double)b.r2 : r2;
    const double rad = __builtin_sqrt(r2) * (1.0 + 1e-5);
    const double mx = 1e-4 * (1.0 + __builtin_fabs(pxmin) + __builtin_fabs(pxmax) + rad);
    const double my = 1e-4 * (1.0 + __builtin_fabs(pymin) + __builtin_fabs(pymax) + rad);
    const double rr = rad + (mx > my ? mx : my);
    const double bx0 = pxmin - rr, bx1 = pxmax + rr, by0 = pymin - rr, by1 = pymax + rr;
    const int rx0 = t.x0 - FT_H > 0 ? t.x0 - FT_H : 0, rx1 = t.x0 + t.ox + FT_H - 1 < p.nx - 1 ? t.x0 + t.ox + FT_H - 1 : p.nx - 1;
    const int ry0 = t.y0 - FT_H > 0 ? t.y0 - FT_H : 0, ry1 = t.y0 + t.oy + FT_H - 1 < p.ny - 1 ? t.y0 + t.oy + FT_H - 1 : p.ny - 1;
    const double xa = p.x[rx0], xb = p.x[rx1], ya = p.y[ry0], yb = p.y[ry1];
    if (bx1 < xa || bx0 > xb || by1 < ya || by0 > yb) return false;
    const double cx0 = bx0 + rr, cx1 = bx1 - rr, cy0 = by0 + rr, cy1 = by1 - rr;  // box of the possible centres
    double ddx = xa - cx1 > cx0 - xb ? xa - cx1 : cx0 - xb;
    ddx = ddx > 0.0 ? ddx : 0.0;
    double ddy = ya - cy1 > cy0 - yb ? ya - cy1 : cy0 - yb;
    ddy = ddy > 0.0 ? ddy : 0.0;
    return !(ddx * ddx + ddy * ddy > rr * rr);
}

// cull_lds: the tile's cylinder list as the device found it ([0] count, [1 ...] indices), or nullptr: the host's list
template <int AUX, int FL, int NW, int RPT>
WV_HD void fused_tile_init(const FusedParams &p, const TileDesc &t, int tid, TileCtx &cx, FusedRegs<AUX, RPT> &r,
                           const int *cull_lds = nullptr)
{
    const int lane = tid & 63, w = wv_wave_of(tid);
    const int gx = t.x0 - FT_H + lane;
    const bool inx = gx >= 0 && gx < p.nx;
    const int cgx = gx < 0 ? 0 : (gx >= p.nx ? p.nx - 1 : gx);
    cx.has_src = (FL & F_SRC) ? tile_has_src(p, t) : false;
    cx.has_cyl = (FL & F_CYL) ? (p.M > 0 && t.cyl_count != 0) : false;
    cx.cyl_lds = cx.has_cyl && t.cyl_count > 0 && t.cyl_count <= FT_MAXCYL;
    r.sx = (AUX == AUX_PX || AUX == AUX_ALL) ? p.sx[cgx] : 0.0f;
    r.xs = ((FL & F_CYL) && cx.has_cyl) ? p.x[cgx] : 0.0f;
    r.cidx = ((FL & F_CYL) && cx.cyl_lds && tid < 3 * t.cyl_count)
                 ? (cull_lds ? cull_lds[1 + tid % t.cyl_count] : p.cyl_idx[t.cyl_begin + tid % t.cyl_count]) : 0;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
        const int gy = t.y0 - FT_H + ly;
        const bool in = inx && gy >= 0 && gy < p.ny && ly < t.oy + 2 * FT_H;
        const int cgy = gy < 0 ? 0 : (gy >= p.ny ? p.ny - 1 : gy);
        const unsigned id = (unsigned)cgy * (unsigned)p.nx + (unsigned)cgx;
        r.g[rr] = ((FL & F_SRC) && cx.has_src && in) ? p.G[id] : 0.0f;
    }
}

// ---- phase 0b: the per-step scalars -----------------------------------------------------------------------------
// The tile's culled cylinders at the three stage times of a step go to LDS (threads 0 .. 3*cyl_count-1, one cylinder
// each), in two halves so that a resident tile can have the global load of the NEXT step's cylinders in flight while it
// waits for its halo: fused_cyl_fetch (global -> register) ... fused_cyl_commit (register -> LDS).  The LDS copy is
// read by fused_speed only, right after the first barrier of a step; the commit for the next step may therefore happen
// any time after that barrier and needs one barrier before the next fused_speed.
template <int AUX, int FL, int RPT>
WV_HD Cyl fused_cyl_fetch(const FusedParams &p, int step, const TileDesc &t, int tid, const TileCtx &cx,
                          const FusedRegs<AUX, RPT> &r)
{
#ifdef WV_CYL_NOFETCH  // (timing experiment only: wrong results)
    return Cyl{0.0f, 0.0f, 0.0f, 0.0f};
#endif
    if ((FL & F_CYL) && cx.cyl_lds && tid < 3 * t.cyl_count) {  // block-uniform up to the thread test
        if (p.dsg) {  // (block-uniform) the tile evaluates the interpolator itself: cx.tau was prepared for this `step`
            const int q = tid / t.cyl_count;
            return design_cyl_tau(*p.dsg, r.cidx, q == 0 ? cx.tau[0] : (q == 1 ? cx.tau[1] : cx.tau[2]));
        }
        return p.cyl_tab[(size_t)(3 * step + tid / t.cyl_count) * p.M + r.cidx];
    }
    return Cyl{0.0f, 0.0f, 0.0f, 0.0f};
}
// stage times t, t + 0.5f0*dt, t + dt of the tabulated time of `step` (src/dynamics.jl:10-13), clamped and shifted as the
// interpolator does: what fused_cyl_fetch(step) of a tile that evaluates its cylinders itself needs
template <int FL>
WV_HD void fused_cyl_times(const FusedParams &p, int step, TileCtx &cx)
{
    if (!(FL & F_CYL) || !p.dsg || step >= p.nsteps) return;  // block-uniform
    const float t0 = p.tspan[step];
    cx.tau[0] = design_tau(*p.dsg, t0);
    cx.tau[1] = design_tau(*p.dsg, t0 + p.hdt);
    cx.tau[2] = design_tau(*p.dsg, t0 + p.dt);
}
template <int FL>
WV_HD void fused_cyl_commit(const TileDesc &t, int tid, const FusedLds &lds, const TileCtx &cx, const Cyl &c)
{
    if ((FL & F_CYL) && cx.cyl_lds && tid < 3 * t.cyl_count) lds.cyl[tid] = c;
}
template <int FL>
WV_HD void fused_step_init(const FusedParams &p, int step, TileCtx &cx)
{
    cx.step = step;
#pragma unroll
    for (int q = 0; q < 3; ++q) cx.sf[q] = (cx.has_src && p.sfac_tab) ? p.sfac_tab[3 * step + q] : 0.0f;
}

// ---- phase 0c: global -> registers --------------------------------------------------------------------------------
template <int AUX, int NW, int RPT>
WV_HD void fused_load_state(const FusedParams &p, const float *u, const TileDesc &t, int tid, FusedRegs<AUX, RPT> &r)
{
    constexpr int NS = aux_ns(AUX);
    const int lane = tid & 63, w = wv_wave_of(tid);
    const int gx = t.x0 - FT_H + lane;
    const bool inx = gx >= 0 && gx < p.nx;
    const int cgx = gx < 0 ? 0 : (gx >= p.nx ? p.nx - 1 : gx);
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
        const int gy = t.y0 - FT_H + ly;
        const bool in = inx && gy >= 0 && gy < p.ny && ly < t.oy + 2 * FT_H;
        const int cgy = gy < 0 ? 0 : (gy >= p.ny ? p.ny - 1 : gy);
        const unsigned id = (unsigned)cgy * (unsigned)p.nx + (unsigned)cgx;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const float *plane = u + (size_t)(6 * s + aux_plane(AUX, j)) * p.P;
                r.u[rr][s][j] = in ? plane[id] : 0.0f;
            }
    }
}

// ---- halo exchange of k_steps_resident ----------------------------------------------------------------------------
// A resident tile keeps its own output cells in registers from step to step; what it needs from outside is the halo
// ring, i.e. the outermost FT_H cells of the neighbouring tiles' outputs.  Those travel through p.xch as 16-byte
// GRANULES {three field values, step tag}, each written by ONE agent-scope (sc1, write-through) 16-byte store and read by
// ONE agent-scope 16-byte load: a reader that finds the expected tag has the three values -- no flag, no acknowledgement
// to wait for, no ordering between different granules needed.  One buffer per tag parity: a tile can only write the
// border of step s+2 after it has read its neighbours' borders of step s+1, which they wrote after reading this tile's
// border of step s -- so nobody still needs the granule that is being overwritten.
//
// What this relies on: a 16-byte-aligned 16-byte access of one lane is never observed torn (tag of one store, values of
// another).  The ISA does not promise it (only naturally aligned accesses up to 8 bytes are atomics the compiler itself
// emits); it holds on gfx950 because such an access lies within one 32-byte sector of one 64-byte request on every hop
// (vector L1 -> L2 -> fabric -> memory side), and requests are not split below a sector: tools/micro/tear16.hip looks for
// a torn granule under load from all XCDs (1.5e11 checked loads, none torn: profiles/r02/tear16_b128.txt), and
// tests/test_gpu_parity.py runs it (wv_selftest_granules) on the device the tests run on.
//
// Why granules and not one tagged word per value: three values per tag instead of one -- 25 % fewer bytes and a third
// fewer vector-memory instructions in the exchange burst that every tile issues at the same moment of a step.
// (Measured and NOT done: letting the idle neighbour lanes of an x-border row carry the incident granule, with the values
// crossing the 4 lanes by ds_bpermute -- one instruction per such row instead of two -- cost 4 %: the two dependent
// LDS-crossbar round trips sit on the critical path of the exchange, the saved instructions do not.)
//
// Layout: [tag parity (2)][granule plane (4)][cell (P)] 16-byte granules.  Planes 0 / 1: {U, Vx, Vy} of the total /
// incident set.  Planes 2 / 3 depend on the CELL's damping class (both sides derive it from sigma_x, sigma_y at the
// cell, so writer and reader agree whatever field sets their tiles carry):
//   class 0  sigma_x == 0, sigma_y == 0 : nothing (the auxiliary fields are the exact zeros the field-set invariant guarantees)
//   class 1  sigma_x != 0 only          : plane 2 = {Psi_x total, Psi_x incident, -}
//   class 2  sigma_y != 0 only          : plane 2 = {Psi_y total, Psi_y incident, -}
//   class 3  both (or p.reduced == 0)   : plane 2 = {Psi_x, Psi_y, Omega} total, plane 3 = the same of the incident set
// (On the device a value crosses into / out of the 128-bit word through an explicit register copy: otherwise the register
// coalescer makes the long-lived state value a sub-register of the word's aligned register tuple, and the kernel spills.)
WV_HD unsigned xch_copy(unsigned v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned o;
    asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(v));
    return o;
#else
    return v;
#endif
}
// keeps the compiler from hoisting the per-granule address arithmetic out of the poll loop -- formed next to the load it
// folds into the instruction's scalar-base + 32-bit-offset addressing
WV_HD unsigned xch_opaque(unsigned off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(off));
#endif
    return off;
}
struct XchG {
    unsigned a, b, c, t;  // three value bit patterns, tag
};
constexpr int XCH_PLANES = 4;
// a byte offset no exchange buffer reaches (num_records < 2^31): the buffer unit returns zeros for such a lane and drops its
// stores, without a memory access
constexpr unsigned XCH_OOB = 0x80000000u;
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned int wv_u4 __attribute__((ext_vector_type(4)));
// raw buffer resource over the exchange buffer (gfx9 word 3: 32-bit data format, raw addressing); accesses carry the
// agent-scope (sc1) cache policy: stores write through to memory, loads are served from there
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xch_rsrc(const FusedParams &p)
{
    return __builtin_amdgcn_make_buffer_rsrc(p.xch, 0, (int)p.xch_bytes, 0x00020000);
}
#endif
// byte offset of granule plane g (0..3) in the buffer of tag parity `par` (block-uniform); planes are P*16 bytes apart
WV_HD unsigned xch_plane_offset(const FusedParams &p, unsigned par, int g) { return (par * (unsigned)XCH_PLANES + (unsigned)g) * (p.P * 16u); }
WV_HD int xch_class(const FusedParams &p, bool lx, bool ly) { return p.reduced ? ((lx ? 1 : 0) | (ly ? 2 : 0)) : 3; }
WV_HD void xch_put(const FusedParams &p, unsigned base, unsigned off, unsigned a, unsigned b, unsigned c, unsigned tag)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // The three values are copied into the word's registers by hand (see xch_copy).  Wait states on both sides of the
    // copies, because the compiler does not look into inline assembly -- and because its own model of this hazard says
    // "none when the store has an SGPR soffset", which gfx950 does not honour for these sc1 stores:
    //   * before: the registers may be the data registers of the 16-byte store issued just before (the previous granule).
    //     A buffer store of more than 8 bytes reads its data some cycles after issue; overwriting them at once made lanes
    //     12-15 of every 16 of the PREVIOUS store carry the NEW values (found with tools/debug_res.py: the total-set
    //     granule arrived with the incident set's values in those lanes);
    //   * after: a VALU write of a VGPR needs a wait state before a store of more than 8 bytes reads it.
    unsigned o0, o1, o2;
    asm volatile("s_nop 3\n\tv_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5\n\ts_nop 1"
                 : "=&v"(o0), "=&v"(o1), "=&v"(o2)
                 : "v"(a), "v"(b), "v"(c));
    const wv_u4 w = {o0, o1, o2, tag};
#ifdef WV_XCH_PLAINSTORE  // (timing experiment only: the stores stay in the XCD's L2)
    __builtin_amdgcn_raw_buffer_store_b128(w, xch_rsrc(p), (int)off, (int)base, 0);
#else
    __builtin_amdgcn_raw_buffer_store_b128(w, xch_rsrc(p), (int)off, (int)base, 16);  // aux 16: sc1
#endif
#else
    unsigned *q = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(p.xch) + base + off);
    q[0] = a;
    q[1] = b;
    q[2] = c;
    q[3] = tag;
#endif
}
WV_HD void xch_putf(const FusedParams &p, unsigned base, unsigned off, float a, float b, float c, unsigned tag)
{
    xch_put(p, base, off, __builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c), tag);
}
WV_HD XchG xch_get(const FusedParams &p, unsigned base, unsigned off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const wv_u4 w = __builtin_amdgcn_raw_buffer_load_b128(xch_rsrc(p), (int)off, (int)base, 16);
    return XchG{w.x, w.y, w.z, w.w};
#else
    if (off >= XCH_OOB) return XchG{0u, 0u, 0u, 0u};  // (what the buffer unit does with an out-of-range lane)
    const unsigned *q = reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(p.xch) + base + off);
    return XchG{q[0], q[1], q[2], q[3]};
#endif
}
// auxiliary field k (0 Psi_x, 1 Psi_y, 2 Omega) of one wave set of a cell, as the tile's field set holds it (0: not carried)
template <int AUX>
WV_HD float xch_aux_of(const float (&y)[aux_ns(AUX)], int k)
{
    if (AUX == AUX_PX) return k == 0 ? y[aux_ns(AUX) - 1] : 0.0f;
    if (AUX == AUX_PY) return k == 1 ? y[aux_ns(AUX) - 1] : 0.0f;
    if (AUX == AUX_ALL) return y[aux_ns(AUX) - 3 + k];
    return 0.0f;
}

// Between two steps of a resident tile only the state is carried: y (the new state) up to the halo poll, u after it.
// The stage code writes acc / px / y under row conditions, which makes them look live around the whole step loop to the
// register allocator (a path that skips the write reaches the next read).  Giving them an arbitrary value here ends
// those live ranges without creating new ones (a frozen undefined value needs no register: zeros would have to be carried
// around the loop), so that the halo poll -- which keeps all of a thread's exchange granules in flight -- has the registers.
WV_HD float wv_any(float v)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WV_NO_ANY)
    return __builtin_nondeterministic_value(v);
#else
    return v * 0.0f;
#endif
}
// The same with an uninitialised read: `undef` to the optimiser, which folds every merge with it to the other side (the
// frozen form above is resolved to 0.0f, which then has to be materialised: a `v_mov_b32 v, 0` per register and step).
// Nothing ever branches on such a value or lets it reach an output: the rows / lanes it stands in are the ones no later
// stage reads.
WV_HD float wv_undef(float v)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WV_NO_ANY)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
    float x;
    (void)v;
    return x;
#pragma clang diagnostic pop
#else
    return v * 0.0f;
#endif
}
// A register with arbitrary content and no instruction behind it (an empty asm "defines" it): the other side of a merge
// with an assignment made under a row test.  Unlike the two above the optimiser cannot resolve it to a constant, and its
// live range starts here -- so it is placed right in front of the test.
WV_HD float wv_fresh()
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WV_NO_ANY)
    float x;
    asm volatile("" : "=v"(x));
    return x;
#else
    return 0.0f;
#endif
}
#ifndef WV_UNDEF_MASK
#define WV_UNDEF_MASK 18  // bsq and y: free; px, acc, u: the undef form raises register pressure (spills)
#endif
template <int AUX, int RPT>
WV_HD void fused_end_step(FusedRegs<AUX, RPT> &r)  // after stage 4 (y is still needed)
{
    constexpr int NS = aux_ns(AUX);
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) r.px[rr][k] = (WV_UNDEF_MASK & 1) ? wv_undef(r.px[rr][k]) : wv_any(r.px[rr][k]);
#pragma unroll
        for (int q = 0; q < 3; ++q) r.bsq[q][rr] = (WV_UNDEF_MASK & 2) ? wv_undef(r.bsq[q][rr]) : wv_any(r.bsq[q][rr]);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                r.acc[rr][s][j] = (WV_UNDEF_MASK & 4) ? wv_undef(r.acc[rr][s][j]) : wv_any(r.acc[rr][s][j]);
                r.u[rr][s][j] = (WV_UNDEF_MASK & 8) ? wv_undef(r.u[rr][s][j]) : wv_any(r.u[rr][s][j]);
            }
    }
}
template <int AUX, int RPT>
WV_HD void fused_end_poll(FusedRegs<AUX, RPT> &r)  // after the halo poll (u holds the state)
{
    constexpr int NS = aux_ns(AUX);
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < NS; ++j) r.y[rr][s][j] = (WV_UNDEF_MASK & 16) ? wv_undef(r.y[rr][s][j]) : wv_any(r.y[rr][s][j]);
}

// after stage 4: the tile's output cells within FT_H of the edge of its output rectangle -> exchange buffer.
// Rows of the top / bottom border: every own lane stores the granules of its cell.  Other own rows: only the 4 + 4
// x-border cells go out.
template <int AUX, int NW, int RPT>
WV_HD void fused_xch_store(const FusedParams &p, unsigned tag, const TileDesc &t, int tid, const FusedRegs<AUX, RPT> &r)
{
    constexpr int NS = aux_ns(AUX);
    constexpr bool HAS_SY = AUX == AUX_PY || AUX == AUX_ALL;
    const int lane = tid & 63, w = wv_wave_of(tid);
    const int gx = t.x0 - FT_H + lane;
    const bool ownx = lane >= FT_H && lane < FT_H + t.ox;
    const bool edgex = lane < 2 * FT_H || lane >= t.ox;
    const bool lx = r.sx != 0.0f;
    const unsigned PS = p.P * 16u;
    const unsigned base = xch_plane_offset(p, tag & 1u, 0);
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
        if (ly < FT_H || ly >= FT_H + t.oy) continue;
        const bool rowb = ly < 2 * FT_H || ly >= t.oy;
        const int gy = t.y0 - FT_H + ly;
        const unsigned row = (unsigned)gy * (unsigned)p.nx;
        const unsigned off = (row + (unsigned)gx) * 16u;  // byte offset in a plane: 16*P < 2^32
        const float(&y)[2][NS] = r.y[rr];
        const bool mine = ownx && (rowb || edgex);  // (plain mask algebra: a select between the two masks became a v_cndmask chain)
#ifdef WV_XCH_NOSTORE  // (timing experiment only)
        (void)off; (void)mine; (void)base; (void)PS; (void)lx;
        continue;
#else
        if (mine) {
            xch_putf(p, base, off, y[0][0], y[0][1], y[0][2], tag);
            xch_putf(p, base + PS, off, y[1][0], y[1][1], y[1][2], tag);
        }
        if (NS > 3) {
            const bool lyy = HAS_SY ? p.sy[gy] != 0.0f : false;
            const int cls = xch_class(p, lx, lyy);
            if (mine && cls == 3) {
                xch_putf(p, base + 2 * PS, off, xch_aux_of<AUX>(y[0], 0), xch_aux_of<AUX>(y[0], 1), xch_aux_of<AUX>(y[0], 2), tag);
                xch_putf(p, base + 3 * PS, off, xch_aux_of<AUX>(y[1], 0), xch_aux_of<AUX>(y[1], 1), xch_aux_of<AUX>(y[1], 2), tag);
            } else if (mine && cls != 0) {  // (selects, not a run-time index: the state must stay in registers)
                const float vt = cls == 1 ? xch_aux_of<AUX>(y[0], 0) : xch_aux_of<AUX>(y[0], 1);
                const float vi = cls == 1 ? xch_aux_of<AUX>(y[1], 0) : xch_aux_of<AUX>(y[1], 1);
                xch_putf(p, base + 2 * PS, off, vt, vi, 0.0f, tag);
            }
        }
#endif
    }
}

// Before the next step: u <- y on the tile's own cells, u <- the neighbours' border values on the halo ring.  Returns
// false when some granule of this thread's halo cells does not carry `tag` yet (the caller polls: nothing but r.u has
// been modified, and every call rewrites all of it).  Halo rows: every lane loads the granules of its cell.  Own rows:
// only the 4 + 4 halo lanes load anything.
template <int AUX, int NW, int RPT>
WV_HD bool fused_xch_load(const FusedParams &p, unsigned tag, const TileDesc &t, int tid, FusedRegs<AUX, RPT> &r)
{
    constexpr int NS = aux_ns(AUX);
    constexpr int NG = NS == 3 ? 2 : (AUX == AUX_ALL ? 4 : 3);  // granules a halo cell of this field set may need
    const int lane = tid & 63, w = wv_wave_of(tid);
    const int gx = t.x0 - FT_H + lane;
    const bool inx = gx >= 0 && gx < p.nx && lane < t.ox + 2 * FT_H;  // (columns beyond the region belong to nobody's ring)
    const int cgx = gx < 0 ? 0 : (gx >= p.nx ? p.nx - 1 : gx);
    const bool ownx = lane >= FT_H && lane < FT_H + t.ox;
    const bool lx = r.sx != 0.0f;
    const unsigned PS = p.P * 16u;
    const unsigned base = xch_plane_offset(p, tag & 1u, 0);
    // ---- first all the loads, branch-free: a lane that needs nothing from a row addresses beyond the buffer's
    // num_records (XCH_OOB) -- the buffer unit answers such a lane with zeros without touching memory -- so every
    // granule of the halo is in flight before the first tag is looked at (with the loads under row conditions the
    // compiler waited for each row's granules before issuing the next row's: four round trips instead of one), and
    // a cell that is not read needs no zero of its own.  Then the tag checks: a read that comes too early is
    // abandoned before any of the selects below.
    XchG g[RPT][NG];
    int cls[RPT];
    bool needs[RPT];
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
        const int gy = t.y0 - FT_H + ly;
        const bool in_row = gy >= 0 && gy < p.ny && ly < t.oy + 2 * FT_H;
        const int cgy = gy < 0 ? 0 : (gy >= p.ny ? p.ny - 1 : gy);
        const bool own = ownx && ly >= FT_H && ly < FT_H + t.oy;
#ifdef WV_XCH_NOLOAD  // (timing experiment only: no halo is read at all)
        const bool need = false && inx && in_row;
#else
        const bool need = inx && in_row && !own;
#endif
        needs[rr] = need;
        const unsigned off = ((unsigned)cgy * (unsigned)p.nx + (unsigned)cgx) * 16u;
        // With reduced field sets the owner of a halo cell carries -- and sends -- an auxiliary field only where that
        // field can be non-zero (class of the cell); anywhere else its value is the exact zero the field-set invariant
        // guarantees.  AUX_PX / AUX_PY tiles only exist with reduced field sets, and their region has sigma == 0 along the
        // other axis: their one auxiliary field arrives in a class 1 / class 2 granule.
        cls[rr] = 0;
        if (AUX == AUX_PX) cls[rr] = lx ? 1 : 0;
        if (AUX == AUX_PY) cls[rr] = p.sy[cgy] != 0.0f ? 2 : 0;
        if (AUX == AUX_ALL) cls[rr] = xch_class(p, lx, p.sy[cgy] != 0.0f);
        const unsigned o = xch_opaque(need ? off : XCH_OOB);
        g[rr][0] = xch_get(p, base, o);
        g[rr][1] = xch_get(p, base + PS, o);
        if (NG > 2) g[rr][2] = xch_get(p, base + 2 * PS, xch_opaque(need && cls[rr] != 0 ? off : XCH_OOB));
        if (NG > 3) g[rr][3] = xch_get(p, base + 3 * PS, xch_opaque(need && cls[rr] == 3 ? off : XCH_OOB));
    }
    bool ok = true;
#ifndef WV_XCH_NOWAIT  // (timing experiment only: results are wrong without the check)
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        bool have = g[rr][0].t == tag && g[rr][1].t == tag;
        if (NG > 2) have = have && (cls[rr] == 0 || g[rr][2].t == tag);
        if (NG > 3) have = have && (cls[rr] != 3 || g[rr][3].t == tag);
        ok = ok && (have || !needs[rr]);
#ifdef WV_XCH_DEBUG
        if (needs[rr] && !have && wv_xch_debug)
            printf("  slot %d aux %d x0 %d y0 %d ox %d oy %d: lane %d rr %d has tags %u %u want %u\n", t.slot, t.aux, t.x0, t.y0, t.ox, t.oy, lane, rr, g[rr][0].t, g[rr][1].t, tag), wv_xch_debug--;
#endif
    }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    if (!__all(ok)) return false;   // wave-uniform: the caller polls again
#else
    if (!ok) return false;
#endif
    // ---- then the new state of the step: own cells keep what stage 4 left in y, halo cells take the granules' values,
    // everything else (outside the domain, beyond the region) the zeros of the out-of-range reads: one select per value
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
        const bool own = ownx && ly >= FT_H && ly < FT_H + t.oy;
        float at[3] = {0.0f, 0.0f, 0.0f}, ai[3] = {0.0f, 0.0f, 0.0f};  // Psi_x, Psi_y, Omega of the halo cell
        if (NG == 3) {
            at[AUX == AUX_PX ? 0 : 1] = __builtin_bit_cast(float, g[rr][2].a);
            ai[AUX == AUX_PX ? 0 : 1] = __builtin_bit_cast(float, g[rr][2].b);
        } else if (NG == 4) {
            // (selects, not run-time indices: everything stays in registers)
            const int c = cls[rr];
            const XchG &h2 = g[rr][2], &h3 = g[rr][NG - 1];
            const float a2 = __builtin_bit_cast(float, h2.a), b2 = __builtin_bit_cast(float, h2.b);
            const float c2 = __builtin_bit_cast(float, h2.c);
            const float a3 = __builtin_bit_cast(float, h3.a), b3 = __builtin_bit_cast(float, h3.b);
            const float c3 = __builtin_bit_cast(float, h3.c);
            at[0] = (c & 1) ? a2 : 0.0f;
            ai[0] = c == 3 ? a3 : (c == 1 ? b2 : 0.0f);
            at[1] = c == 3 ? b2 : (c == 2 ? a2 : 0.0f);
            ai[1] = c == 3 ? b3 : (c == 2 ? b2 : 0.0f);
            at[2] = c == 3 ? c2 : 0.0f;
            ai[2] = c == 3 ? c3 : 0.0f;
        }
        const unsigned v0[3] = {g[rr][0].a, g[rr][0].b, g[rr][0].c}, v1[3] = {g[rr][1].a, g[rr][1].b, g[rr][1].c};
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            float ht, hi;
            if (j < 3) {
                ht = __builtin_bit_cast(float, v0[j]);
                hi = __builtin_bit_cast(float, v1[j]);
            } else {
                const int k = aux_plane(AUX, j) - 3;
                ht = at[k];
                hi = ai[k];
            }
            r.u[rr][0][j] = own ? r.y[rr][0][j] : ht;
            r.u[rr][1][j] = own ? r.y[rr][1][j] : hi;
        }
    }
    return true;
}

// Cheap look at the halo before the full read: ONE granule per lane (the incident-set granule of the lane's first halo
// cell -- written after the total-set one).  A full read that comes too early costs every lane all its loads and the
// register shuffling behind them a second time; spinning on this single load until it carries the tag costs next to
// nothing.  Purely a timing device: fused_xch_load still checks every granule.
template <int AUX, int NW, int RPT>
WV_HD bool fused_xch_probe(const FusedParams &p, unsigned tag, const TileDesc &t, int tid)
{
    const int lane = tid & 63, w = wv_wave_of(tid);
    const int gx = t.x0 - FT_H + lane;
    const bool inx = gx >= 0 && gx < p.nx && lane < t.ox + 2 * FT_H;
    const bool ownx = lane >= FT_H && lane < FT_H + t.ox;
    bool have = false;
    unsigned off = 0;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
        const int gy = t.y0 - FT_H + ly;
        const bool in_row = gy >= 0 && gy < p.ny && ly < t.oy + 2 * FT_H;
        const bool own = ownx && ly >= FT_H && ly < FT_H + t.oy;
        const bool need = inx && in_row && !own;
        if (need && !have) {
            have = true;
            off = ((unsigned)gy * (unsigned)p.nx + (unsigned)gx) * 16u;
        }
    }
    unsigned tg = tag;
    if (have) tg = xch_get(p, xch_plane_offset(p, tag & 1u, 1), xch_opaque(off)).t;
    return tg == tag;
}

// phases 0a-0c of a single-step launch
template <int AUX, int FL, int NW, int RPT>
WV_HD void fused_load(const FusedParams &p, const StepIO &io, const TileDesc &t, int tid, const FusedLds &lds, TileCtx &cx,
                      FusedRegs<AUX, RPT> &r)
{
    fused_tile_init<AUX, FL, NW, RPT>(p, t, tid, cx, r);
    fused_step_init<FL>(p, io.step, cx);
    fused_cyl_times<FL>(p, io.step, cx);
    fused_cyl_commit<FL>(t, tid, lds, cx, fused_cyl_fetch<AUX, FL, RPT>(p, io.step, t, tid, cx, r));
    fused_load_state<AUX, NW, RPT>(p, io.u, t, tid, r);
}

// ---- phase "publish": the stage input's stencil fields -> LDS ------------------------------------------------
// (stage S's inputs go to LDS buffer (S-1)&1; the x-neighbour values stay in the thread's px registers)
template <int AUX, int FL, int NW, int RPT, int S>
WV_HD void fused_publish(const FusedParams &p, const TileDesc &t, int tid, const FusedLds &lds, const TileCtx &cx,
                         FusedRegs<AUX, RPT> &r)
{
    constexpr int BUF = (S - 1) & 1;
    const int lane = tid & 63, w = wv_wave_of(tid);
    const int rows = t.oy + 2 * FT_H;
    const float sf = cx.sf[stage_q(S)];
    const float cp = p.ops.cp;
    const int gx = t.x0 - FT_H + lane;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
        const float(&yin)[2][aux_ns(AUX)] = S == 1 ? r.u[rr] : r.y[rr];
        float wt = yin[0][0], wi = yin[1][0];
        if (FL & F_SRC) {  // tiles whose region misses the source's support carry g == 0: U + (+-0) == U exactly
            const float f = r.g[rr] * sf;  // shape .* sin(...)      src/sources.jl:67-69
            wt = wt + f;                   // U .+ f                 src/dynamics.jl:166-167
            wi = wi + f;
        }
        const int i = lds_at(lane, ly);
        r.px[rr][0] = cp * wt;
        r.px[rr][1] = cp * wi;
        r.px[rr][2] = cp * yin[0][1];
        r.px[rr][3] = cp * yin[1][1];
        // (the four products of a row the stage does not need are formed all the same: assigned under the row test, px
        // would need a value on the other side of it -- a register move per product and step -- and the waves that skip
        // rows wait at the stage barrier for those that do not anyway)
        if (ly < S - 1 || ly >= rows - (S - 1)) continue;  // y_S is only needed on the region shrunk by S-1
        lds.W[BUF][i] = F2{r.px[rr][0], r.px[rr][1]};
        lds.Vy[BUF][i] = F2{cp * yin[0][2], cp * yin[1][2]};
        if (FL & F_EDGE) {  // raw copies of the three cells next to a domain boundary (sides compiled in per variant)
            const int gy = t.y0 - FT_H + ly;
            if ((FL & F_EL) && (t.edge & EDGE_L) && gx >= 0 && gx < 3) {
                lds.XL[BUF][(ly * 3 + gx) * 2 + 0] = F2{wt, wi};
                lds.XL[BUF][(ly * 3 + gx) * 2 + 1] = F2{yin[0][1], yin[1][1]};
            }
            if ((FL & F_ER) && (t.edge & EDGE_R) && gx >= p.nx - 3 && gx < p.nx) {
                lds.XR[BUF][(ly * 3 + (gx - (p.nx - 3))) * 2 + 0] = F2{wt, wi};
                lds.XR[BUF][(ly * 3 + (gx - (p.nx - 3))) * 2 + 1] = F2{yin[0][1], yin[1][1]};
            }
            if ((FL & F_ET) && (t.edge & EDGE_T) && gy >= 0 && gy < 3) {
                lds.YT[BUF][(gy * FT_X + lane) * 2 + 0] = F2{wt, wi};
                lds.YT[BUF][(gy * FT_X + lane) * 2 + 1] = F2{yin[0][2], yin[1][2]};
            }
            if ((FL & F_EB) && (t.edge & EDGE_B) && gy >= p.ny - 3 && gy < p.ny) {
                lds.YB[BUF][((gy - (p.ny - 3)) * FT_X + lane) * 2 + 0] = F2{wt, wi};
                lds.YB[BUF][((gy - (p.ny - 3)) * FT_X + lane) * 2 + 1] = F2{yin[0][2], yin[1][2]};
            }
        }
    }
}

// ---- phase "speed": c^2 at the three stage times of the step, once per step --------------------------------------
// Runs right after the first barrier of the step (the cylinder copy in LDS is complete, and of the thread's state only
// u and px are live, so the cylinder loop has the registers); stages 2 and 3 share one evaluation (t + dt/2).
template <int AUX, int FL, int NW, int RPT>
WV_HD void fused_speed(const FusedParams &p, const TileDesc &t, int tid, const FusedLds &lds, const TileCtx &cx,
                       FusedRegs<AUX, RPT> &r)
{
    if (!(FL & F_CYL)) return;
    const int w = wv_wave_of(tid);
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) r.bsq[q][rr] = p.c0sq;
#ifndef WV_CYL_NOSPEED  // (timing experiment only: wrong results)
    if (cx.has_cyl) tile_speed_sq<NW, RPT>(p, t, cx, lds, w, r.xs, r.bsq);  // block-uniform
#endif
}

// one-sided rows of `grad` (src/operators.jl:14-15) on three raw (total, incident) pairs, ascending column order
WV_HD F2 one_sided(float c0, float c1, float c2, F2 v0, F2 v1, F2 v2)
{
    return F2{(c0 * v0.x + c1 * v1.x) + c2 * v2.x, (c0 * v0.y + c1 * v1.y) + c2 * v2.y};
}

// ---- phase "compute": k_S from the LDS image, then the RK update of the registers ----------------------------
// `nb`: see px_left / px_right.
template <int AUX, int FL, int NW, int RPT, int S>
WV_HD void fused_compute(const FusedParams &p, const TileDesc &t, int tid, const FusedLds &lds, const TileCtx &cx,
                         FusedRegs<AUX, RPT> &r, const FusedRegs<AUX, RPT> *nb)
{
    constexpr int NS = aux_ns(AUX);
    constexpr int BUF = (S - 1) & 1;
    constexpr bool HAS_SX = AUX == AUX_PX || AUX == AUX_ALL;
    constexpr bool HAS_SY = AUX == AUX_PY || AUX == AUX_ALL;
    const int lane = tid & 63, w = wv_wave_of(tid);
    const int rows = t.oy + 2 * FT_H;
    const int gx = t.x0 - FT_H + lane;
    const Ops &o = p.ops;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
#ifndef WV_NO_FRESH_ACC  // (default since the end of round 2: 24 fewer v_mov_b32 v, 0 per wave and step)
        if (S == 1) {  // (k1 opens the accumulator: nothing of the previous step is carried into the row test)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < NS; ++j) r.acc[rr][s][j] = wv_fresh();
        }
#endif
        if (ly < S || ly >= rows - S) continue;  // k_S is only needed on the region shrunk by S
        const int gy = t.y0 - FT_H + ly;
        if ((FL & (F_ET | F_EB)) && (gy < 0 || gy >= p.ny)) continue;
        const int i = lds_at(lane, ly);
        const F2 Wd = lds.W[BUF][i - FT_LX], Wu = lds.W[BUF][i + FT_LX];
        const F2 Yd = lds.Vy[BUF][i - FT_LX], Yu = lds.Vy[BUF][i + FT_LX];
        F2 Ux = F2{px_right(nb, lane, rr, 0) - px_left(nb, lane, rr, 0), px_right(nb, lane, rr, 1) - px_left(nb, lane, rr, 1)};
        F2 Uy = F2{Wu.x - Wd.x, Wu.y - Wd.y};
        F2 Vxx = F2{px_right(nb, lane, rr, 2) - px_left(nb, lane, rr, 2), px_right(nb, lane, rr, 3) - px_left(nb, lane, rr, 3)};
        F2 Vyy = F2{Yu.x - Yd.x, Yu.y - Yd.y};
        bool border = false;
        if (FL & F_EDGE) {  // one-sided stencils and the Dirichlet mask, for the sides this variant was compiled with
            // (x sides: every lane forms the one-sided value -- a broadcast read of the row's six raw values -- and the
            // boundary lane selects it: no lane-divergent branch, so the rows of a thread still interleave; the branch
            // with one active lane cost the same issue slots and serialised them)
            if ((FL & F_EL) && (t.edge & EDGE_L)) {
                const F2 *v = lds.XL[BUF] + ly * 6;
#ifndef WV_EDGE_SELECT  // (measured: selects instead of the one-lane branch are 2.5 % slower)
                if (gx == 0) {
                    Ux = one_sided(o.f0, o.f1, o.f2, v[0], v[2], v[4]);
                    Vxx = one_sided(o.f0, o.f1, o.f2, v[1], v[3], v[5]);
                }
#else
                const F2 a = one_sided(o.f0, o.f1, o.f2, v[0], v[2], v[4]), b = one_sided(o.f0, o.f1, o.f2, v[1], v[3], v[5]);
                Ux = gx == 0 ? a : Ux;
                Vxx = gx == 0 ? b : Vxx;
#endif
                border = border || gx <= 0;
            }
            if ((FL & F_ER) && (t.edge & EDGE_R)) {
                const F2 *v = lds.XR[BUF] + ly * 6;
#ifndef WV_EDGE_SELECT  // (measured: selects instead of the one-lane branch are 2.5 % slower)
                if (gx == p.nx - 1) {
                    Ux = one_sided(o.b0, o.b1, o.b2, v[0], v[2], v[4]);
                    Vxx = one_sided(o.b0, o.b1, o.b2, v[1], v[3], v[5]);
                }
#else
                const F2 a = one_sided(o.b0, o.b1, o.b2, v[0], v[2], v[4]), b = one_sided(o.b0, o.b1, o.b2, v[1], v[3], v[5]);
                Ux = gx == p.nx - 1 ? a : Ux;
                Vxx = gx == p.nx - 1 ? b : Vxx;
#endif
                border = border || gx >= p.nx - 1;
            }
            if ((FL & F_ET) && (t.edge & EDGE_T) && gy == 0) {
                const F2 *v = lds.YT[BUF] + lane * 2;
                Uy = one_sided(o.f0, o.f1, o.f2, v[0], v[2 * FT_X], v[4 * FT_X]);
                Vyy = one_sided(o.f0, o.f1, o.f2, v[1], v[2 * FT_X + 1], v[4 * FT_X + 1]);
                border = true;
            }
            if ((FL & F_EB) && (t.edge & EDGE_B) && gy == p.ny - 1) {
                const F2 *v = lds.YB[BUF] + lane * 2;
                Uy = one_sided(o.b0, o.b1, o.b2, v[0], v[2 * FT_X], v[4 * FT_X]);
                Vyy = one_sided(o.b0, o.b1, o.b2, v[1], v[2 * FT_X + 1], v[4 * FT_X + 1]);
                border = true;
            }
        }
        const float(&yin)[2][NS] = S == 1 ? r.u[rr] : r.y[rr];
        const float sx = r.sx;
        const float sy = HAS_SY ? p.sy[gy] : 0.0f;
        float k[2][NS];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float b = (s == 0 && (FL & F_CYL)) ? r.bsq[stage_q(S)][rr] : p.c0sq;
            const float ux = s == 0 ? Ux.x : Ux.y, uy = s == 0 ? Uy.x : Uy.y;
            const float vxx = s == 0 ? Vxx.x : Vxx.y, vyy = s == 0 ? Vyy.x : Vyy.y;
            const float U = yin[s][0];
            // src/dynamics.jl:169-174, left to right as written, with the exact zeros of the field set dropped
            float dU = b * (vxx + vyy);
            if (AUX == AUX_PX) dU = (dU + yin[s][3]) - sx * U;
            if (AUX == AUX_PY) dU = (dU + yin[s][3]) - sy * U;
            if (AUX == AUX_ALL) dU = (((dU + yin[s][3]) + yin[s][4]) - (sx + sy) * U) - yin[s][5];
            if ((FL & F_EDGE) && border) dU = 0.0f * dU;  // bc .* dU   src/dynamics.jl:176, src/dims.jl:117-124
            k[s][0] = dU;
            k[s][1] = HAS_SX ? ux - sx * yin[s][1] : ux;
            k[s][2] = HAS_SY ? uy - sy * yin[s][2] : uy;
            if (AUX == AUX_PX) k[s][3] = (b * sx) * vyy;
            if (AUX == AUX_PY) k[s][3] = (b * sy) * vxx;
            if (AUX == AUX_ALL) {
                k[s][3] = (b * sx) * vyy;
                k[s][4] = (b * sy) * vxx;
                k[s][5] = (sx * sy) * U;
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

// Output-state store.  Agent-scope write-through (sc1) by default: the new state is not re-read by this kernel, and streaming it out
// while other tiles still compute shortens the end-of-kernel L2 write-back (measured at 700^2: nt +4 %, sc1 +7 % over plain).
// WV_STORE_MODE (build-time knob for A/B runs): 0 plain, 1 non-temporal, 2 agent-scope write-through (sc1).
#ifndef WV_STORE_MODE
#define WV_STORE_MODE 2
#endif
WV_HD void store_out(float *ptr, float v)
{
#if defined(__HIP_DEVICE_COMPILE__) && WV_STORE_MODE == 1
    __builtin_nontemporal_store(v, ptr);
#elif defined(__HIP_DEVICE_COMPILE__) && WV_STORE_MODE == 2
    __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *ptr = v;
#endif
}

// ---- last phase: registers -> global, energy terms of src/env.jl:105-111 --------------------------------------
template <int AUX, int NW, int RPT>
WV_HD void fused_store(const FusedParams &p, const StepIO &io, const TileDesc &t, int tid, const FusedRegs<AUX, RPT> &r,
                       float e[3])
{
    constexpr int NS = aux_ns(AUX);
    const int lane = tid & 63, w = wv_wave_of(tid);
    e[0] = e[1] = e[2] = 0.0f;
    // The step's output pointers ONCE, in front of the rows: read through `io` inside the row loop they were fetched again
    // for every row (the output stores in between might alias the table, as far as the compiler knows) -- three dependent
    // scalar-memory round trips per row and wave, ~0.5 us per step of pure waiting.
    float *const io_out = io.out, *const io_tt = io.traj_tot, *const io_ti = io.traj_inc;
    const unsigned nx_ = (unsigned)p.nx;
    if (lane < FT_H || lane >= FT_H + t.ox) return;
    const int gx = t.x0 - FT_H + lane;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int ly = w + NW * rr;
        if (ly < FT_H || ly >= FT_H + t.oy) continue;
        const int gy = t.y0 - FT_H + ly;
        const unsigned id = (unsigned)gy * nx_ + (unsigned)gx;
        if (io_out) {  // block-uniform: a resident tile only writes the states somebody reads (frames, the last step)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    float *plane = io_out + (size_t)(6 * s + aux_plane(AUX, j)) * p.P;
                    store_out(plane + id, r.y[rr][s][j]);
                }
        }
        // (the energy sums are this project's own arithmetic -- the reference leaves their order open, src/env.jl:105-111 --
        // so the fused multiply-add is used here: 4 instead of 7 instructions per cell)
        const float ut = r.y[rr][0][0], ui = r.y[rr][1][0], us = ut - ui;
        e[0] = __builtin_fmaf(ut, ut, e[0]);
        e[1] = __builtin_fmaf(ui, ui, e[1]);
        e[2] = __builtin_fmaf(us, us, e[2]);
        if (io_tt) io_tt[id] = ut;
        if (io_ti) io_ti[id] = ui;
    }
}

// ---- job protocol primitives (device: one WAVE executes them, lanes in parallel; CPU emulation: one thread, lane == 0) ----
// Scopes: "agent" = other blocks of the launch (sc1: served from L2 / memory, never from a CU's L1), "sys" = the host.
#if defined(__HIPCC__)
#define WV_SCOPE_AGENT __HIP_MEMORY_SCOPE_AGENT
#define WV_SCOPE_SYS __HIP_MEMORY_SCOPE_SYSTEM
#else
#define WV_SCOPE_AGENT 0
#define WV_SCOPE_SYS 0
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define WV_JOB_LD(p, scope) __hip_atomic_load(p, __ATOMIC_RELAXED, scope)
#define WV_JOB_ST(p, v, scope) __hip_atomic_store(p, v, __ATOMIC_RELAXED, scope)
constexpr int JOB_LANES = 64;
#else
#define WV_JOB_LD(p, scope) __atomic_load_n(p, __ATOMIC_ACQUIRE)
#define WV_JOB_ST(p, v, scope) __atomic_store_n(p, v, __ATOMIC_RELEASE)
constexpr int JOB_LANES = 1;
unsigned long long emu_job_clock();  // provided by the CPU emulation (a virtual clock it can move at will) ...
void emu_job_pause();                // ... and its yield to the other emulated blocks; never referenced by the library itself
#endif
WV_HD unsigned job_ld_agent(const unsigned *p) { return WV_JOB_LD(p, WV_SCOPE_AGENT); }
WV_HD void job_st_agent(unsigned *p, unsigned v) { WV_JOB_ST(p, v, WV_SCOPE_AGENT); }
WV_HD unsigned long long job_ld_agent64(const unsigned long long *p) { return WV_JOB_LD(p, WV_SCOPE_AGENT); }
WV_HD void job_st_agent64(unsigned long long *p, unsigned long long v) { WV_JOB_ST(p, v, WV_SCOPE_AGENT); }
WV_HD unsigned job_ld_sys(const unsigned *p) { return WV_JOB_LD(p, WV_SCOPE_SYS); }
WV_HD void job_st_sys(unsigned *p, unsigned v) { WV_JOB_ST(p, v, WV_SCOPE_SYS); }
WV_HD void job_st_sys64(unsigned long long *p, unsigned long long v) { WV_JOB_ST(p, v, WV_SCOPE_SYS); }
WV_HD void job_st_sysf(float *p, float v) { WV_JOB_ST(reinterpret_cast<unsigned *>(p), __builtin_bit_cast(unsigned, v), WV_SCOPE_SYS); }
WV_HD float job_ld_agentf(const float *p) { return __builtin_bit_cast(float, WV_JOB_LD(reinterpret_cast<const unsigned *>(p), WV_SCOPE_AGENT)); }
WV_HD void job_st_agentf(float *p, float v) { WV_JOB_ST(reinterpret_cast<unsigned *>(p), __builtin_bit_cast(unsigned, v), WV_SCOPE_AGENT); }
WV_HD unsigned long long job_clock()  // 100 MHz
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_s_memrealtime();
#else
    return emu_job_clock();
#endif
}
WV_HD void job_drain()  // my stores have left / my loads are back
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
WV_HD void job_pause()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_sleep(4);
#else
    emu_job_pause();
#endif
}
WV_HD bool job_all(bool v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __all(v) != 0;
#else
    return v;
#endif
}
WV_HD bool job_any(bool v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __any(v) != 0;
#else
    return v;
#endif
}
WV_HD bool job_reached(unsigned have, unsigned want) { return (int)(have - want) >= 0; }  // (numbers wrap; launches do not live 2^31 jobs)

// Leader, one wave, non-blocking: if job `seq` has been rung, bring its description over and release the other blocks
// (they look at go[seq & 1] only once they are through with job seq - 1, so this may happen while that job still runs).
// Returns the job's command, or 0 when nothing has been rung yet.
WV_HD int job_try_fetch(const JobArgs &a, unsigned seq, int lane)
{
    const unsigned par = seq & 1u;
    if (!job_reached(job_ld_sys(&a.mail->bell), seq)) return 0;
    // the description, dword by dword: host memory -> device memory (the parameters, and the per-step tables only when the
    // job carries them: ~90 resp. ~1 400 dwords, all loads of a lane in flight together)
    const unsigned *src = reinterpret_cast<const unsigned *>(&a.mail->desc[par]);
    unsigned *dst = reinterpret_cast<unsigned *>(&a.ctl->jobs[par]);
    constexpr int NP4 = (int)(sizeof(FusedParams) / 4), NW4 = (int)(sizeof(JobDesc) / 4);
    for (int k = lane; k < NP4; k += JOB_LANES) job_st_agent(dst + k, job_ld_sys(src + k));
    job_drain();
    const unsigned got = job_ld_agent(&a.ctl->jobs[par].p.seq);  // (a description that is not the one rung: treat as "leave")
    const int cmd = (got == seq && job_ld_agent(reinterpret_cast<const unsigned *>(&a.ctl->jobs[par].p.cmd)) == (unsigned)JOB_RUN) ? JOB_RUN : JOB_EXIT;
    if (cmd == JOB_RUN && job_ld_agent(reinterpret_cast<const unsigned *>(&a.ctl->jobs[par].p.dev_cull)) != 0u) {
        // (of the per-step tables only the rows of this call's steps)
        const int ns = (int)job_ld_agent(reinterpret_cast<const unsigned *>(&a.ctl->jobs[par].p.nsteps));
        constexpr int T4 = (int)(offsetof(JobDesc, tspan) / 4), S4 = (int)(offsetof(JobDesc, sfac) / 4);
        const int t_end = T4 + ns + 1 < S4 ? T4 + ns + 1 : S4, s_end = S4 + 3 * ns < NW4 ? S4 + 3 * ns : NW4;
        for (int k = NP4 + lane; k < t_end; k += JOB_LANES) job_st_agent(dst + k, job_ld_sys(src + k));
        for (int k = S4 + lane; k < s_end; k += JOB_LANES) job_st_agent(dst + k, job_ld_sys(src + k));
        job_drain();
    }
    if (lane == 0) {
        if (cmd != JOB_RUN) {
            job_st_sys(&a.back->exit_seq[a.launch], seq);
            job_st_sys(&a.back->status[a.launch], JOBS_EXIT_TOLD);
            job_drain();
        }
        job_st_agent64(reinterpret_cast<unsigned long long *>(&a.ctl->go[par]), (unsigned long long)seq | ((unsigned long long)(unsigned)cmd << 32));
    }
    return cmd;
}

// Leader, one wave: wait for job `seq` (bounded by idle_ticks: then back->status says JOBS_EXIT_IDLE, the others are told
// to leave, and the host must not count on this launch for the job).  Returns the job's command.
WV_HD int job_leader_fetch(const JobArgs &a, unsigned seq, int lane)
{
    const unsigned long long t0 = job_clock();
    for (;;) {
        const int cmd = job_try_fetch(a, seq, lane);
        if (cmd != 0) return cmd;
        if (job_clock() - t0 > (unsigned long long)a.idle_ticks) break;
        job_pause();
    }
    if (lane == 0) {
        job_st_sys(&a.back->exit_seq[a.launch], seq);
        job_st_sys(&a.back->status[a.launch], JOBS_EXIT_IDLE);
        job_drain();
        job_st_agent64(reinterpret_cast<unsigned long long *>(&a.ctl->go[seq & 1u]), (unsigned long long)seq | ((unsigned long long)(unsigned)JOB_EXIT << 32));
    }
    return JOB_EXIT;
}

// Everybody (one wave per block): the command of job `seq`.  Bounded: a leader that never answers (it cannot: it leaves
// through the same word) is treated as "leave".
WV_HD int job_wait_go(const JobArgs &a, unsigned seq)
{
    const unsigned par = seq & 1u;
    const unsigned long long t0 = job_clock();
    for (;;) {
        const unsigned long long g = job_ld_agent64(reinterpret_cast<const unsigned long long *>(&a.ctl->go[par]));
        if ((unsigned)g == seq) return (int)(unsigned)(g >> 32);
        if (job_clock() - t0 > 16ull * (unsigned long long)a.idle_ticks + 100000ull) return JOB_EXIT;
        job_pause();
    }
}

// Grid barrier through a flag array: flags[b] = seq says "block b has arrived"; returns when every block has (true), or when
// the abort word is set / max_polls looks did not suffice (false).  One wave; the caller has made sure (barrier + drain)
// that what the arrival stands for has happened in EVERY wave of the block.
WV_HD bool job_barrier(unsigned *flags, int ntiles, int b, unsigned seq, int max_polls, int *abort, int lane)
{
    if (lane == 0) job_st_agent(flags + b, seq);
    for (int polls = 0; polls < max_polls; ++polls) {
        bool ok = true;
        for (int k = lane; k < ntiles; k += JOB_LANES) ok = ok && job_reached(job_ld_agent(flags + k), seq);
        if (job_all(ok)) return true;
        if (job_any(job_ld_agent(reinterpret_cast<const unsigned *>(abort)) != 0u)) return false;
        job_pause();
    }
    return false;
}

}  // namespace wv
