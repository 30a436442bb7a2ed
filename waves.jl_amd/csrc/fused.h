// Host-side interface of the fused Runge-Kutta step kernel (kernels_fused.hip): all four stages of one
// integration step (src/dynamics.jl:9-16 around src/dynamics.jl:151-188) in ONE launch.
#pragma once
#include "kernels.h"

namespace wv {

struct FusedPlan;

struct FusedStep {
    const float *u;     // state at the start of the step (12 planes)
    float *out;         // state at the end of the step (12 planes, != u)
    const float *G;     // source shape or nullptr
    float sfac[3];      // source time factor at t, t + dt/2, t + dt
    int table_row;      // row of the cylinder table holding stage time t (rows +1, +2: t + dt/2, t + dt)
    float dt;
    float *epart;       // per-block energy partials [fused_energy_blocks][3] or nullptr
    float *traj_tot;    // optional copies of the new U_tot / U_inc planes
    float *traj_inc;
};

FusedPlan *fused_create(const Grid &g, const float *x_host, const float *y_host, const float *sx_host,
                        const float *sy_host);
void fused_destroy(FusedPlan *p);
void fused_set_pml(FusedPlan *p, const float *sx_host, const float *sy_host);
// The caller replaced the state: the "auxiliary fields are zero outside the PML" fast-path precondition is unknown.
void fused_state_changed(FusedPlan *p);
void fused_state_zeroed(FusedPlan *p);
// number of tiles == rows of the energy-partial array; valid after fused_prepare (the decomposition depends on whether
// the 6-field fast path applies).  fused_generation changes whenever the decomposition does.
int fused_energy_blocks(FusedPlan *p);
int fused_generation(const FusedPlan *p);
void fused_dump_stamps(FusedPlan *p, hipStream_t s);  // diagnostic, no-op unless WAVES_AMD_STAMPS is set
// Called once per wv_integrate before the first step: d_table = device cylinder table (rows x M), h_table its host copy.
// frames = env.wave (3 states, the last one is the initial condition), scratch0/1 the two ping-pong states.
// G = device source shape or nullptr (NoSource).
int fused_prepare(FusedPlan *p, float *frames, float *scratch0, float *scratch1, bool capture, const float *G,
                  const Cyl *d_table, const Cyl *h_table, int M, int rows, hipStream_t s);
void fused_source_changed(FusedPlan *p);  // the source shape was replaced
void fused_launch(FusedPlan *p, const FusedStep &st, hipStream_t s);
void fused_variant_counts(const FusedPlan *p, int out[4]);  // tiles per field set: NONE, PX, PY, ALL

}  // namespace wv
