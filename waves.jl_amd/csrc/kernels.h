// Launch wrappers of the HIP kernels (implemented in kernels_staged.hip / kernels_aux.hip / kernels_fused.hip).
#pragma once
#include "common.h"

namespace wv {

// Everything a Runge-Kutta stage needs that does not change during one wv_integrate call.
struct Grid {
    int nx, ny;
    size_t P;          // nx*ny
    Ops ops;
    const float *x;    // device, nx
    const float *y;    // device, ny
    const float *sx;   // device, nx   sigma_x profile (src/pml.jl:21-29)
    const float *sy;   // device, ny   sigma_y = sigma_x' (src/dynamics.jl:161-162)
    float c0;
    float c0sq;        // dyn.c0 .^ 2 (scalar b of the incident set, src/dynamics.jl:159,186)
};

struct StageIO {
    const float *yin;  // stage input (12 planes)
    const float *u;    // state at the start of the step (12 planes)
    float *acc;        // running k1 + 2k2 + 2k3 (12 planes)
    float *out;        // next stage input (stages 0..2), new state (stage 3) or k (RHS only)
    const float *G;    // source shape or nullptr (NoSource)
    float sfac;        // sin(2f0*pi*t*freq) for this stage time
    const Cyl *cyl;    // M cylinders at this stage time (device) or nullptr
    int M;
    float a;           // 0.5f0*dt (stages 0,1) or dt (stage 2)
    float dt;
    float *epart;      // stage 3: per-block energy partials [nblocks][3] or nullptr
    float *traj_tot;   // stage 3: optional copy of U_tot / U_inc planes
    float *traj_inc;
};

// stage: 0..3 = RK stages of src/dynamics.jl:9-16; 4 = RHS only (out = k).
void launch_stage(const Grid &g, const StageIO &io, int stage, hipStream_t s);
int staged_energy_blocks(const Grid &g);

// aux kernels
void launch_energy_partial(const Grid &g, const float *state, float *epart, int nblocks, hipStream_t s);
// signal[r][c] = float(sum_b epart[r][b][c]) * dOmega for r in [0, nrows); row 0 is read from row0 (which may be
// epart itself); signal may be pinned host memory; *flag_dst = *flag_src (both may be nullptr)
void launch_energy_final(const float *row0, const float *epart, int nrows, int nblocks, float dOmega, float *signal,
                         const int *flag_src, int *flag_dst, hipStream_t s);
// 16-byte granule atomicity self-test (see k_selftest_granules); out[0] += granules checked, out[1] += torn ones
void launch_selftest_granules(unsigned char *buf, unsigned bytes, int iters, int nwriters, unsigned stride,
                              unsigned long long *out, hipStream_t s);
void launch_speed_field(const Grid &g, const Cyl *cyl, int M, float *out, hipStream_t s);
void launch_gaussian(const Grid &g, int K, const float *mu, const float *sigma, const float *a, float *out,
                     hipStream_t s);
void launch_scale(const float *in, float f, float *out, size_t n, hipStream_t s);
void launch_gradient(const Grid &g, int axis, const float *u, float *out, hipStream_t s);
void launch_copy_planes(const float *state, size_t P, float *tot, float *inc, hipStream_t s);
// (rx, ry, 4) observation of state(env): U_tot of the three frames + the source shape (nullptr: zeros), resized
void launch_observation(const Grid &g, const float *f0, const float *f1, const float *f2, const float *G, int rx, int ry,
                        float *out, hipStream_t s);

// batched 1-D latent dynamics (kernels_latent.hip); all pointers are device memory, layouts column-major as in Julia
struct LatentArgs {
    int n, B, K, steps;      // cells (<= 1024), batch, knots of C, integration steps
    Ops ops;                 // gradient(dim.x)
    float c0, dt, hdt, pml_scale;
    const float *X;          // (K, B)        knots of C = LinearInterpolation(X, Y)
    const float *Y;          // (n, K, B)
    const float *shape;      // (n, B)        F = Source(shape, freq)
    const float *PML;        // (n, B)
    const float *sfac;       // (steps, 3, B) sin(2f0*pi*t*freq) at the three stage times of every step
    const float *t;          // (steps + 1, B) -- stored as [step][batch]
    const float *z0;         // (n, 4, B)
    float *z;                // (n, 4, B, steps + 1)
};
void launch_latent(const LatentArgs &a, hipStream_t s);

// reverse sweep of adjoint_sensitivity (src/dynamics.jl:97-121) over the same dynamics
struct LatentAdjArgs {
    LatentArgs f;            // sizes, operators, theta; f.sfac holds (steps + 1, 3, B) -- the sweep also visits the last saved time;
                             // f.z = the forward solution (read), f.z0 unused
    const float *adj;        // (n, 4, B, steps + 1)  dL/dz
    float *gz0;              // (n, 4, B)
    float *gY;               // (n, K, B)   zero on entry
    float *gshape;           // (n, B)
    float *gPML;             // (n, B)
};
void launch_latent_adjoint(const LatentAdjArgs &a, hipStream_t s);

}  // namespace wv
