/* waves_amd.h -- C ABI of the MI355X-native WaveEnv integrator (libwaves_amd.so).
 *
 * Drop-in boundary for the hot path of gladisor/Waves.jl (reference citations are
 * relative to the reference checkout, e.g. src/dynamics.jl:37-53).  The reference's
 * interface is a set of Julia callable structs whose coefficient inputs are opaque
 * closures (theta = [C, F], src/env.jl:99,102).  Closures cannot cross a C ABI, so the
 * boundary is drawn at the parametric data those closures are built from:
 *
 *   C(t) = speed(DesignInterpolator(initial, final, ti, tf)(t), grid, c0)   -> wv_set_design
 *   F(t) = shape .* sin(2f0*pi*t*freq)                                      -> wv_set_source_*
 *
 * Conventions
 *   - every function returns an int status (WV_OK == 0); no C++ exception crosses the ABI;
 *     wv_last_error() gives the message of the last failure.
 *   - all pointer arguments are CALLER-OWNED HOST memory, copied during the call and never
 *     retained; outputs go to caller-allocated buffers of the documented size; NULL means
 *     "not wanted" where documented.
 *   - arrays are fp32 in the reference's memory layout: Julia (x, y, field[, frame])
 *     column-major, i.e. x contiguous, then y, then field.  A "state" is 12*nx*ny floats in
 *     the field order of src/dynamics.jl:179-188: U,Vx,Vy,Psix,Psiy,Omega (total), then the
 *     same six for the incident wave.
 *   - one ctx = one device + one HIP stream; calls on a ctx are not re-entrant; different
 *     ctxs may be driven from different threads/processes.
 *   - the library has NO CPU fallback: with no usable gfx950 device wv_create fails with
 *     WV_ERR_NO_DEVICE.
 */
#ifndef WAVES_AMD_H
#define WAVES_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WV_ABI_VERSION 3
#define WV_NFIELDS 12 /* src/dynamics.jl:185-187 */
#define WV_NFRAMES 3  /* src/env.jl:54,116 */
#define WV_FRAMESKIP 10 /* src/env.jl:90 */

typedef struct wv_ctx wv_ctx;

enum wv_status {
    WV_OK = 0,
    WV_ERR_INVALID = 1,   /* bad argument (the reference would throw an AssertionError/BoundsError/MethodError) */
    WV_ERR_HIP = 2,       /* a HIP runtime call failed */
    WV_ERR_NO_DEVICE = 3, /* no usable GPU: there is no CPU fallback */
    WV_ERR_NOMEM = 4,
    WV_ERR_STATE = 5      /* call sequence error (e.g. wv_integrate_end without _begin) */
};

/* which integrator implementation runs wv_integrate */
enum wv_impl {
    WV_IMPL_AUTO = 0,   /* fastest verified path */
    WV_IMPL_STAGED = 1, /* one kernel per Runge-Kutta stage (simple, the in-library cross-check) */
    WV_IMPL_FUSED = 2   /* all four RK stages of a step fused in one LDS-tiled kernel */
};

/* Replaces the constructor arguments of
 *   AcousticDynamics(dim, c0, pml_width, pml_scale)   src/dynamics.jl:141-149
 *   Integrator(runge_kutta, dyn, dt)                  src/dynamics.jl:18-22
 * as WaveEnv's ctor builds them (src/env.jl:37-67). */
typedef struct wv_config {
    int nx, ny;      /* size(dim): length(dim.x), length(dim.y)   src/dims.jl:70-72 */
    float c0;        /* ambient wave speed (WATER = 1531f0)         src/designs.jl:13 */
    float dt;        /* Integrator.dt (1f-5)                        src/env.jl:47 */
    float pml_width; /* 2f0                                         src/env.jl:43 */
    float pml_scale; /* 20000f0                                     src/env.jl:44 */
    int device;      /* HIP device ordinal (replaces Flux.device!(n), scripts/data.jl:33) */
    int impl;        /* enum wv_impl */
} wv_config;

/* timing of the last wv_integrate on this ctx, measured with HIP events on the ctx's stream */
typedef struct wv_timing {
    double total_ms;        /* first enqueue -> last kernel of the call */
    double step_kernel_ms;  /* duration of the integrator launch(es): fused path, the events around the resident launch
                             * or the whole chain of single-step launches; profiling mode: the sum of the per-launch
                             * brackets; staged path without profiling: 0 */
    int step_kernel_launches;
    int steps;
    int impl;               /* implementation that ran */
    int resident;           /* 1: all steps ran as ONE job of the resident kernel (step_kernel_launches == 1); total_ms and
                             * step_kernel_ms are then the kernel's own clock stamps around the job (100 MHz device clock: job
                             * seen by the launch -> state, frames and trace complete), because a resident launch may serve
                             * many calls and HIP events only see the launch */
    int gave_up;            /* 1: the resident kernel abandoned the call (a tile waited in vain for a neighbour: the device is
                             * shared with somebody else's kernels); the call was then run again, transparently, by the
                             * single-step kernels from its untouched initial condition, and the context stays on them */
    double launch_ms;       /* the resident LAUNCH that ended last on this ctx: its duration by HIP events ... */
    int launch_jobs;        /* ... and the number of wv_integrate calls it served (0: none has ended yet) */
} wv_timing;

int wv_abi_version(void);
/* message of the last failure on this ctx; with ctx == NULL, of the last failure on this thread that had no ctx
 * (wv_create).  Replaces Julia exceptions (src/env.jl:52 @assert, MethodErrors). */
const char *wv_last_error(const wv_ctx *ctx);
int wv_device_count(int *count);

/* AcousticDynamics + Integrator construction.  x[nx], y[ny] are dim.x / dim.y (src/dims.jl:56-60) passed verbatim:
 * the library derives the gradient coefficients (src/operators.jl:10-22 -- uses x only, like build_gradient(dim)),
 * the PML profile (src/pml.jl:21-29 -- sigma_y is the transposed x profile, src/dynamics.jl:161-162, so nx must
 * equal ny exactly as in the reference), the Dirichlet mask (src/dims.jl:117-124) and dx*dy (src/dims.jl:126-127). */
int wv_create(const wv_config *cfg, const float *x, const float *y, wv_ctx **out);
int wv_destroy(wv_ctx *ctx);

/* dyn.pml as the 1-D profiles it is made of: sigma_x[i] (nx) and sigma_y[j] (ny).  src/pml.jl:21-29 */
int wv_get_pml(wv_ctx *ctx, float *sigma_x, float *sigma_y);
int wv_set_pml(wv_ctx *ctx, const float *sigma_x, const float *sigma_y);
/* dOmega = get_dx(dim) * get_dy(dim), the factor of the energy traces.  src/env.jl:108 */
int wv_get_cell_area(wv_ctx *ctx, float *dOmega);

/* env.wave (nx, ny, 12, 3).  src/env.jl:17,54.  The integrator's initial condition is the LAST frame
 * (env.wave[:, :, :, end], src/env.jl:102). */
int wv_set_frames(wv_ctx *ctx, const float *wave /* 12*nx*ny*3 */);
int wv_get_frames(wv_ctx *ctx, float *wave /* 12*nx*ny*3 */);
/* last frame only == the current state `ui` of Integrator(ui, tspan, theta).  src/dynamics.jl:37 */
int wv_set_state(wv_ctx *ctx, const float *u /* 12*nx*ny */);
int wv_get_state(wv_ctx *ctx, float *u /* 12*nx*ny */);
/* `env.wave *= 0f0` of reset!.  src/env.jl:81-88 (design and source re-randomisation are host-side policy). */
int wv_reset(wv_ctx *ctx);

/* F = Source(shape, freq) (src/sources.jl:10-23); shape == NULL selects NoSource (src/sources.jl:7-8). */
int wv_set_source_shape(wv_ctx *ctx, const float *shape /* nx*ny or NULL */, float freq);
/* F = RandomPosGaussianSource after reset!: shape = build_normal(grid, mu, sigma, a) built on the device.
 * mu is K x 2 in Julia's column-major layout (mu[k], then mu[K + k]).  src/sources.jl:41-51, src/utils.jl:12-18 */
int wv_set_gaussian_source(wv_ctx *ctx, int K, const float *mu, const float *sigma, const float *a, float freq);
int wv_get_source_shape(wv_ctx *ctx, float *shape /* nx*ny */);

/* The observation of RLBase.state(env), src/env.jl:132-137:
 *     x = imresize(cat(env.wave[:, :, 1, :], env.source.shape, dims = 3), env.resolution)
 * out is (rx, ry, 4) column-major: U_tot of the three frames of env.wave, then the source shape (zeros for NoSource),
 * each resized on the device from (nx, ny) to (rx, ry); 1 <= rx <= nx, 1 <= ry <= ny (the reference asserts
 * size(dim) .> resolution, src/env.jl:52).  imresize belongs to Images.jl (third-party, unpinned): the rule implemented
 * is stated in kernels_aux.hip (k_observation); the test-suite's CPU restatement of it is imresize_linear.
 * Needs no pending integrate call.  A waiting resident launch is not disturbed: after the first call at a resolution the
 * following wv_integrate calls produce the observation of the frames they leave themselves (same arithmetic, into pinned
 * memory; this call then is a copy), and one they have not produced is computed beside the launch on another stream. */
int wv_observation(wv_ctx *ctx, int rx, int ry, float *out /* rx*ry*4 */);

/* C = t -> speed(DesignInterpolator(initial, final, ti, tf)(t), grid, c0).  src/env.jl:95-99, src/designs.jl:274-292.
 * A design is M cylinders (a Cloak is passed stacked: config cylinders then the core, src/designs.jl:228,133-138):
 * pos is M x 2 column-major (all x, then all y, like Julia's Matrix), r and c have M entries.  M == 0 is NoDesign
 * (C(t) = c0, src/designs.jl:63). */
int wv_set_design(wv_ctx *ctx, int M, const float *pos_initial, const float *r_initial, const float *c_initial,
                  const float *pos_final, const float *r_final, const float *c_final, float ti, float tf);

/* n actions integrated by ONE call: the reference's loop `for a in actions; env(a); end` (src/data.jl:22-27 around
 * src/env.jl:91-121) for a policy that does not read the wave state between actions (RandomDesignPolicy,
 * src/env.jl:151-157).  designs holds n_actions + 1 designs of M cylinders as {px, py, r, c} per cylinder -- design k
 * is the one in force before action k, design k + 1 the one it moves to -- and ti_tf the (ti, tf) of every action's
 * DesignInterpolator (src/env.jl:95-98).  The sequence describes the NEXT wv_integrate / wv_integrate_begin call and only
 * that one: its nsteps must be n_actions * steps_per_action, its tspan holds every action's own tspan in a row
 * (n_actions x (steps_per_action + 1) values: step s of action k starts at tspan[k * (steps_per_action + 1) + s]), its
 * signal has n_actions * steps_per_action + 1 rows (row k * steps_per_action is the last row of action k - 1 and the first of
 * action k), capture_frames == 1 keeps the three frames of the LAST action (env.wave), capture_frames == 2 those of
 * every action (steps_per_action > 20; such a call is not overlapped with another one; wv_observation_action /
 * wv_get_frames_action read them until the next integrate call), trajectories are not available.  Afterwards the
 * context's design is the last action's interpolator.  One launch instead of n: the start-up of a launch and the gap
 * between two launches (together ~8 % of a 100-step action at 700^2) are paid once. */
int wv_set_design_sequence(wv_ctx *ctx, int n_actions, int steps_per_action, int M, const float *designs /* (n_actions+1)*M*4 */,
                           const float *ti_tf /* n_actions*2 */);

/* env.wave / state(env) as they were after action `action` (0-based) of the last wv_integrate call that ran a design
 * sequence with capture_frames == 2: what `s = state(env)` in front of action + 1 sees in the reference's rollout loop
 * (src/data.jl:23, src/env.jl:132-137).  The last action's are wv_get_frames / wv_observation themselves. */
int wv_observation_action(wv_ctx *ctx, int action, int rx, int ry, float *out /* rx*ry*4 */);
int wv_get_frames_action(wv_ctx *ctx, int action, float *wave /* 12*nx*ny*3 */);

/* The closures evaluated at one time (for tests / drop-in of speed() and the source call). */
int wv_speed_field(wv_ctx *ctx, float t, float *out /* nx*ny */);   /* src/designs.jl:110-116 via :287-292 */
int wv_source_field(wv_ctx *ctx, float t, float *out /* nx*ny */);  /* src/sources.jl:67-69 */

/* d/dx (axis 0: `grad * u`) or d/dy (axis 1: `(grad * u')'`) of one (nx, ny) plane.  src/operators.jl:45-46 */
int wv_gradient(wv_ctx *ctx, int axis, const float *u /* nx*ny */, float *out /* nx*ny */);
/* (dyn::AcousticDynamics{TwoDim})(x, t, theta): k = f(x, t).  src/dynamics.jl:179-188 */
int wv_rhs(wv_ctx *ctx, const float *x /* 12*nx*ny */, float t, float *k /* 12*nx*ny */);

/* sol = iter(env.wave[:,:,:,end], tspan, [C, F]) + the reductions of src/env.jl:105-116.
 *   tspan[nsteps+1]   the tabulated times (src/dynamics.jl:5-7); step i uses tspan[i], not an accumulated t.
 *   capture_frames    != 0: env.wave <- states at nsteps-20, nsteps-10, nsteps (src/env.jl:116; needs nsteps >= 20,
 *                     else WV_ERR_INVALID like Julia's BoundsError); == 0: only the last frame is replaced.
 *   signal            (nsteps+1) x 3 row-major [tot, inc, sc] energies x dOmega at every saved time incl. the initial
 *                     one (src/env.jl:105-114), or NULL.
 *   u_tot, u_inc      (nsteps+1) planes of nx*ny (sol[:,:,1,:], sol[:,:,7,:], src/env.jl:105-106,120) or NULL.
 * Synchronous: outputs are ready on return. */
int wv_integrate(wv_ctx *ctx, const float *tspan, int nsteps, int capture_frames, float *signal, float *u_tot,
                 float *u_inc);
/* Trajectories for rendering (render! / build_interpolator, src/plot.jl:24-45, scripts/mpc.jl:65-108) do not need every
 * saved time: with stride k the u_tot / u_inc outputs of wv_integrate hold the saved times 0, k, 2k, ... <= nsteps, i.e.
 * nsteps / k + 1 planes each (k = 1, the default, is the reference's sol[:, :, 1|7, :] of src/env.jl:113,120: all
 * nsteps + 1).  The 396 MB device-to-host copy of a 700^2 action shrinks by k; signal and frames are unaffected. */
int wv_set_trajectory_stride(wv_ctx *ctx, int stride);

/* The same split in two: _begin enqueues all device work and returns, _end waits and copies the outputs.  Exactly one
 * _end per _begin.  Two uses: several ctxs on one device overlap (begin on all, then end on all); and ONE ctx may have
 * two calls in flight -- begin(k+1) may be called before end(k), the host then prepares call k+1 (coefficient tables,
 * tile culling, uploads on a copy stream) while call k runs, and only the device state is sequentially dependent (the
 * reference's loop `env(policy(env))`, src/data.jl:22-27, with a policy that does not read the wave state).  _end ends
 * the OLDEST pending call.  A second call is refused (WV_ERR_STATE) when either call returns trajectories or profiling
 * is on; wv_set_design is the only setter allowed while a call is pending (it describes the NEXT call). */
int wv_integrate_begin(wv_ctx *ctx, const float *tspan, int nsteps, int capture_frames, int want_signal,
                       int want_fields);
int wv_integrate_end(wv_ctx *ctx, float *signal, float *u_tot, float *u_inc);
/* Calls begun and not yet ended (0, 1 or 2).  An _end that fails with an argument / sequence error leaves its call pending;
 * one that fails after the device work was waited for has ended it: a binding that mirrors the queue (the Python one does)
 * asks here instead of guessing.  No reference counterpart (the reference's step is synchronous, src/env.jl:91-121). */
int wv_pending(wv_ctx *ctx, int *count);
/* Trajectory streaming (render! / build_interpolator, src/plot.jl:24-45, need u_tot / u_inc of src/env.jl:120 on the
 * host): with want_fields == 2 in _begin the planes of a call are copied to pinned host memory on a copy stream as soon
 * as its kernels have finished -- i.e. while the NEXT call (begun before this one is ended) computes -- instead of being
 * fetched by a blocking copy inside _end (want_fields == 1).  wv_integrate_end_view ends the oldest pending call and returns pointers to the planes
 * ((nsteps / stride + 1) planes of nx*ny each, see wv_set_trajectory_stride) in library-owned pinned memory: valid until
 * the second wv_integrate_begin after this call.  wv_integrate_end with u_tot / u_inc also works (one more host copy).
 * Two streamed calls may be in flight. */
int wv_integrate_end_view(wv_ctx *ctx, float *signal, const float **u_tot, const float **u_inc, int *planes);

/* measurement and plumbing */
int wv_set_profiling(wv_ctx *ctx, int on); /* bracket every step kernel with HIP events (slower; for roofline) */
int wv_get_timing(wv_ctx *ctx, wv_timing *out);
/* Durations (ms, the resident kernel's own clock: wv_timing.step_kernel_ms) of the resident calls ended since the previous
 * query, oldest first, at most `cap` (the newest ones); *n = how many were written.  Lets a benchmark loop collect per-call
 * times without a wv_get_timing call inside the timed region. */
int wv_get_call_times(wv_ctx *ctx, double *ms, int cap, int *n);
int wv_set_stream(wv_ctx *ctx, void *hip_stream); /* run on a caller-owned hipStream_t (NULL: back to the ctx's own) */
int wv_synchronize(wv_ctx *ctx);
/* raw device pointer of env.wave (12*nx*ny*3 floats) for zero-copy interop (RCCL, torch); valid until wv_destroy.
 * The caller may WRITE through it between integrate calls: from this call until wv_release_device_frames every
 * wv_integrate looks at the state afresh (the reduced-field-set precondition of the step kernels, the initial energies
 * of the trace) instead of relying on what it left there itself. */
int wv_device_frames(wv_ctx *ctx, void **dptr, size_t *bytes);
int wv_release_device_frames(wv_ctx *ctx);
/* Self-test of the one hardware property the resident step kernel relies on beyond the ISA's promises: a 16-byte-aligned
 * 16-byte agent-scope buffer access (the {3 values, tag} granule of its halo exchange, csrc/fused_body.h) is never observed
 * torn.  Writers on all XCDs rewrite self-describing granules for `iters` rounds while readers check them; returns the
 * number of granules checked and the number found torn (expected: 0).  No reference counterpart. */
int wv_selftest_granules(wv_ctx *ctx, int iters, unsigned long long *checked, unsigned long long *torn);
/* raw device pointer of the source shape (nx*ny floats) */
int wv_device_source_shape(wv_ctx *ctx, void **dptr, size_t *bytes);

/* ---- the batched 1-D latent dynamics of the surrogate models -------------------------------------------
 * z = iter(z0, t, [C, F, PML]) with iter = Integrator(runge_kutta, AcousticDynamics(latent_dim, c0, pml_width, pml_scale), dt)
 * (src/model/acoustic_energy_model.jl:89-107, 121-124; dynamics src/dynamics.jl:190-222; RK4 :9-16; Integrator :37-49):
 *   x[n]              latent_dim.x (OneDim, src/dims.jl:48-50); n <= 1024
 *   X (K, B), Y (n, K, B)   C = LinearInterpolation(X, Y), evaluated as linear_interp (src/utils.jl:69-98)
 *   shape (n, B), freq      F = Source(shape, freq) called with the time vector (src/sources.jl:21-23)
 *   PML (n, B)              sigma = dyn.pml[[1]] .* PML  (src/dynamics.jl:192-193, build_pml(::OneDim) src/pml.jl:6-15)
 *   z0 (n, 4, B), t (steps + 1, B)   initial fields [U_tot, V_tot, U_inc, V_inc] and the tabulated times
 *   z (n, 4, B, steps + 1)  every state (the reference's `cat(ui, ...; dims = 4)`)
 * All arrays column-major (Julia layout), caller-owned host memory.  Stateless: no ctx. */
typedef struct wv_latent_config {
    int n, batch, knots, steps;
    float c0, dt, pml_width, pml_scale, freq;
    int device;
} wv_latent_config;
int wv_latent_integrate(const wv_latent_config *cfg, const float *x, const float *X, const float *Y, const float *shape,
                        const float *PML, const float *z0, const float *t, float *z);

/* adjoint_sensitivity(iter, z, t, theta, dL_dz)  (src/dynamics.jl:97-121) and with it the pullback of
 * `Flux.ChainRulesCore.rrule(iter::Integrator, z0, t, theta)` (:123-128): the reverse sweep over every saved time -- the
 * last one included, as the reference writes it -- of the vector-Jacobian product of one runge_kutta call.
 *   z, adj (n, 4, B, steps + 1)   the solution wv_latent_integrate returned and dL/dz
 *   gz0 (n, 4, B)                 dL/dz0
 *   gY (n, K, B), gshape (n, B), gPML (n, B)   the parts of dL/dtheta the reference's models train through this call:
 *                                 C.Y (LinearInterpolation trains Y only, src/utils.jl:94), F.shape, PML.  (Zygote also
 *                                 returns cotangents for C.X and F.freq; nothing upstream consumes them.)
 * The products are written out by hand (the reference gets them from Zygote): same mathematics, summation order this
 * library's own. */
int wv_latent_adjoint(const wv_latent_config *cfg, const float *x, const float *X, const float *Y, const float *shape,
                      const float *PML, const float *z, const float *t, const float *adj, float *gz0, float *gY,
                      float *gshape, float *gPML);

#ifdef __cplusplus
}
#endif
#endif /* WAVES_AMD_H */
