// Host-side interface of the fused Runge-Kutta step kernel (kernels_fused.hip): all four stages of one
// integration step (src/dynamics.jl:9-16 around src/dynamics.jl:151-188) in ONE launch.
#pragma once
#include "kernels.h"

namespace wv {

struct FusedPlan;

// What differs between the steps of one wv_integrate call (everything else is per-call).
struct FusedStep {
    const float *u;     // state at the start of the step (12 planes)
    float *out;         // state at the end of the step (12 planes, != u)
    bool keep;          // somebody reads `out` after the call (a captured frame, the final state): the resident kernel,
                        // whose tiles carry the state in registers, skips the store of every other step
    float *epart;       // per-tile energy partials [fused_energy_blocks][3] or nullptr
    float *traj_tot;    // optional copies of the new U_tot / U_inc planes
    float *traj_inc;
};

// Per-call arguments shared by all steps.
struct FusedCall {
    const float *G;         // device source shape or nullptr (NoSource)
    const float *d_sfac;    // device [nsteps][3] source time factors (nullptr when G is)
    float dt;
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
// after fused_prepare: the tiles use reduced field sets, i.e. every buffer a step writes to must hold zeros in the planes a
// tile does not write
bool fused_reduced(const FusedPlan *p);
void fused_dump_stamps(FusedPlan *p, hipStream_t s);  // diagnostic, no-op unless WAVES_AMD_STAMPS is set
// (defer_wait: a resident launch owns `s` -- nothing is enqueued there, fused_try_resident waits for the uploads itself)
// Called once per wv_integrate before the first step: d_table = device cylinder table (rows x M), h_table its host copy.
// frames = env.wave (3 states, the last one is the initial condition), scratch0/1 the two ping-pong states.
// G = device source shape or nullptr (NoSource).
// row_lo / row_hi: rows of h_table with the earliest / latest stage time (see plan_build_cyl), or -1.
// slot (0 / 1): which of the two sets of per-call device tables this call uses -- two calls may be in flight (the host
// prepares call k+1 while call k runs), so everything a call's kernels read that differs from call to call exists twice.
// The tile / cylinder-index tables are uploaded on `up` (a copy stream); `s` then waits for everything enqueued on `up`
// so far, so the caller's own uploads on `up` (enqueued before this call) are covered by the same wait.
// ic / out2: the state the call starts from and the buffer its final state goes to (one of two that alternate: out2_idx
// says which, for the "holds zeros where reduced tiles do not write" bookkeeping).
int fused_prepare(FusedPlan *p, int slot, float *frames, float *ic, float *out2, int out2_idx, float *scratch0, float *scratch1, bool capture, const float *G,
                  const Cyl *d_table, const Cyl *h_table, int M, int rows, hipStream_t s, hipStream_t up, int row_lo = -1,
                  int row_hi = -1, bool defer_wait = false);
void fused_scratch_dirty(FusedPlan *p);  // somebody else wrote into the scratch states (wv_rhs)
void fused_source_changed(FusedPlan *p);  // the source shape was replaced
// one step, eagerly, as a single launch over all tiles (profiling mode brackets these with events)
void fused_launch(FusedPlan *p, int slot, const FusedCall &call, int step, const FusedStep &st, hipStream_t s);
constexpr int kDevTablesMaxCyl = 32;     // == FT_MAXCYL (fused_body.h)
constexpr int kDevTablesMaxSteps = 256;  // == JOB_MAXSTEPS
// The small per-call tables of a call whose tiles evaluate their cylinders themselves (host memory, copied by the call).
struct FusedDevTables {
    int M;                 // 1 .. FT_MAXCYL cylinders
    const float *d0, *d1;  // [M][4] {px, py, r, c}: DesignInterpolator(initial, final, ti, tf)
    float ti, tf;
    const float *tspan;    // [nsteps + 1], nsteps <= JOB_MAXSTEPS
    const float *sfac;     // [nsteps][3] or nullptr (NoSource)
    float t_lo, t_hi;      // the earliest / latest stage time of the call
};
// Where the energy trace of a call goes when the resident kernel produces it (the tiles do k_energy_final's second pass
// themselves): row0 = partial sums of the initial state [blocks][3], epart = [nsteps + 1][blocks][3] (row s = after step s),
// signal = PINNED HOST memory [(nsteps + 1)][3] or nullptr (no trace wanted).
struct FusedEnergy {
    const float *row0;
    const float *epart;
    float *signal;
    float dOmega;
};
// state(env) of the frames a call leaves, produced by the resident kernel itself at the end of the job (FusedParams::ob_*):
// out = PINNED HOST memory [4][ry][rx] or nullptr (not wanted), f0..f2 = where the three frames will be, G = source shape or nullptr.
struct FusedObs {
    const float *f0, *f1, *f2, *G;
    float *out;
    int rx, ry;
};
// all steps of a call: one JOB of the resident kernel when the tiles fit the device at once (fused_body.h, "jobs": the
// launch may outlive the call and serve the next ones), else a cached hipGraph of single-step kernel nodes
// (WAVES_AMD_FUSED_GRAPH=0: eager launches).  `up`: the stream the call's tables were uploaded on.  keep: the resident
// launch may stay on the device after this call.  The events bracket the single-step launches (they are not used by the
// resident path, whose durations come from the launch's own events and the kernel's clock stamps).  Returns 0 on success.
int fused_run(FusedPlan *p, int slot, const FusedCall &call, const FusedStep *steps, int nsteps, hipStream_t s, hipStream_t up,
              const FusedEnergy &ef, bool keep, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
              const FusedDevTables *dev = nullptr, const FusedObs *ob = nullptr);
// Resident path only.  Returns 0 when the job was handed over, -1 when this call cannot take that path (more tiles than
// the device holds at once, a single step, ...; the caller then launches step by step), 1 on a HIP error.
int fused_try_resident(FusedPlan *p, int slot, const FusedCall &call, const FusedStep *steps, int nsteps, hipStream_t s,
                       hipStream_t up, const FusedEnergy &ef, bool keep, const FusedDevTables *dev = nullptr,
                       const FusedObs *ob = nullptr);
// With `dev` (only for a call that is handed to a resident launch which is already there: fused_persist_alive, and
// fused_dev_tables_ok) the tiles evaluate the DesignInterpolator and cull their cylinders themselves; the caller then has
// built no cylinder table, uploaded nothing and called fused_prepare_light instead of fused_prepare.  Returns 3 when the
// launch has left meanwhile: nothing has happened, the caller prepares the call the ordinary way.
bool fused_dev_tables_ok(FusedPlan *p);
void fused_prepare_light(FusedPlan *p, int slot);
// Wait for the slot's resident call: 0 done (also when the slot has none), 2 the resident kernel gave the call up -- nothing
// has been written over its initial condition; run it again with fused_rerun_steps after fused_gave_up --, 1 HIP error.
int fused_job_wait(FusedPlan *p, int slot, hipStream_t s);
void fused_gave_up(FusedPlan *p, hipStream_t s);
int fused_rerun_steps(FusedPlan *p, int slot, const FusedCall &call, const FusedStep *steps, int nsteps, hipStream_t s,
                      hipEvent_t ev_start, hipEvent_t ev_stop);
// A resident launch is on the device waiting for (or working on) jobs: nothing may be enqueued on the context's stream
// until fused_retire has made it leave (it would wait out the launch's idle limit).  fused_needs_stream: would preparing
// such a call enqueue anything there?
bool fused_persist_alive(FusedPlan *p);
int fused_retire(FusedPlan *p);
bool fused_needs_stream(FusedPlan *p, bool capture, const float *G, int out2_idx);
void fused_allow_persist(FusedPlan *p, bool allow);  // per call: false ends the launch with the call
bool fused_obs_beside_launch(FusedPlan *p);          // a launch is waiting and leaves room for a small kernel on another stream
double fused_last_job_ms(const FusedPlan *p);        // in-kernel duration of the resident call waited for last
int fused_job_times(FusedPlan *p, double *ms, int cap);  // ... of the (last `cap`) resident calls since the previous query; clears
void fused_launch_stats(const FusedPlan *p, double *ms, int *jobs);  // HIP-event duration and jobs of the launch that ended last
// device word the resident kernel sets when it gives up / pinned host word of this slot the caller has it copied to
const int *fused_abort_src(const FusedPlan *p);
int *fused_abort_dst(FusedPlan *p, int slot);
bool fused_last_resident(const FusedPlan *p);  // the last fused_run took the resident path
void fused_allow_resident(FusedPlan *p, bool allow);  // per call: false keeps this call on the single-step kernels
void fused_variant_counts(const FusedPlan *p, int out[4]);  // tiles per field set: NONE, PX, PY, ALL

}  // namespace wv
