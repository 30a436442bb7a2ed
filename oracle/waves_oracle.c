/* CPU ORACLE (C restatement) -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * Plain-C restatement of the WaveEnv integrator hot path of gladisor/Waves.jl,
 * written from the reference's source text.  Same arithmetic, operation order
 * and rounding as oracle/waves_oracle.py (tests/test_golden.py and tests/cpu_emu checks the two
 * bit-for-bit); it exists so that 700^2 / 2048^2 cases finish in seconds and as
 * the `cpu_baseline` ("port") leg of bench.py.
 *
 * PARITY STATUS: parity unpinned except for the gradient operator (the only
 * test the reference has: test/operators.jl:4-30).  See oracle/waves_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  Build: `make -C oracle` (gcc -O2 -ffp-contract=off: the
 * reference's CPU broadcast never fuses a*b+c).
 *
 * Memory layout: Julia (x, y, field) column-major == plane f at f*nx*ny,
 * row j at j*nx, x contiguous.
 *
 * Reference citations are relative to /root/reference/.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int nx, ny;
    float cm, cp;          /* operators.jl:14-21  -1/(2D), +1/(2D) */
    float fwd[3], bwd[3];  /* [-3,4,-1]/(2D), [1,-4,3]/(2D)         */
    const float *sx;       /* sigma profile along x (pml.jl:21-29)  */
    const float *sy;       /* sigma_y = sigma_x' -> same profile indexed by j (dynamics.jl:161-162) */
} wo_ops;

/* src/operators.jl:10-22: every coefficient is coef/(2*Delta) in fp32. */
static void wo_make_ops(wo_ops *o, int nx, int ny, const float *x, const float *sx, const float *sy)
{
    float delta = (x[nx - 1] - x[0]) / (float)(nx - 1);
    float two_d = 2.0f * delta;
    o->nx = nx; o->ny = ny;
    o->cm = -1.0f / two_d;
    o->cp = 1.0f / two_d;
    o->fwd[0] = -3.0f / two_d; o->fwd[1] = 4.0f / two_d; o->fwd[2] = -1.0f / two_d;
    o->bwd[0] = 1.0f / two_d;  o->bwd[1] = -4.0f / two_d; o->bwd[2] = 3.0f / two_d;
    o->sx = sx; o->sy = sy;
}

/* `grad * u` along a strided line (operators.jl:45-46): SparseArrays accumulates
 * ascending columns from zero, products rounded separately. */
static inline float wo_d(const wo_ops *o, const float *u, int i, int n, int stride)
{
    if (i == 0)
        return (o->fwd[0] * u[0] + o->fwd[1] * u[stride]) + o->fwd[2] * u[2 * stride];
    if (i == n - 1)
        return (o->bwd[0] * u[(n - 3) * stride] + o->bwd[1] * u[(n - 2) * stride]) + o->bwd[2] * u[(n - 1) * stride];
    return o->cm * u[(i - 1) * stride] + o->cp * u[(i + 1) * stride];
}

/* src/pml.jl:21-29 -> 1-D profile of length n. */
void wo_build_pml_profile(int n, const float *xs, float width, float scale, float *out)
{
    float x0 = fabsf(xs[0]);
    float pml_start = x0 - width;
    float mn = INFINITY;
    for (int i = 0; i < n; ++i) {
        float a = fabsf(xs[i]);
        if (a > pml_start && a < mn) mn = a;
    }
    for (int i = 0; i < n; ++i) {
        float a = fabsf(xs[i]);
        float v = 0.0f;
        if (a > pml_start) v = (a - mn) / width;
        out[i] = ((v * v) * v) * scale;
    }
}

/* src/designs.jl:287-292 with the design algebra of :47-53, :80-82 for ONE scalar
 * component: v_i + ((v_f + (-1*v_i)) * (1/Dt)) * (clamp(t,ti,tf) - ti). */
static inline float wo_interp1(float vi, float vf, float inv_dt, float tau)
{
    float dy = vf + (-1.0f * vi);
    return vi + (dy * inv_dt) * tau;
}

/* design at time t: out[m*4 + {0,1,2,3}] = px, py, r, c */
void wo_design_at(int M, const float *d0, const float *d1, float ti, float tf, float t, float *out)
{
    float dt = tf - ti;
    dt = dt > 0.0f ? dt : 1.0f;
    float inv_dt = 1.0f / dt;
    float tc = t < ti ? ti : (t > tf ? tf : t);
    float tau = tc - ti;
    for (int k = 0; k < 4 * M; ++k) out[k] = wo_interp1(d0[k], d1[k], inv_dt, tau);
}

/* src/designs.jl:99-116: c = c0*[no mask] + sum_m mask_m*c_m (ascending m). */
void wo_speed_field(int nx, int ny, const float *x, const float *y, int M, const float *cyl /*M*4*/, float c0,
                    float *out)
{
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ny; ++j) {
        for (int i = 0; i < nx; ++i) {
            int count = 0;
            float cd = 0.0f;
            for (int m = 0; m < M; ++m) {
                float ddx = x[i] - cyl[4 * m + 0];
                float ddy = y[j] - cyl[4 * m + 1];
                float r = cyl[4 * m + 2];
                float d2 = ddx * ddx + ddy * ddy;
                int in = d2 < r * r;
                count += in;
                cd = cd + (in ? cyl[4 * m + 3] : 0.0f);
            }
            float C0 = count == 0 ? c0 : 0.0f;
            out[(size_t)j * nx + i] = C0 + cd;
        }
    }
}

/* src/sources.jl:21-22,67-69: sin(((2f0*pi)*t)*freq), fp32 argument, accurately rounded sin. */
float wo_source_factor(float t, float freq)
{
    float two_pi = 2.0f * (float)M_PI;
    float arg = (two_pi * t) * freq;
    return (float)sin((double)arg);
}

/* src/dynamics.jl:151-177 for one wave set (6 planes at xin), writing 6 planes at kout.
 * c is a field (cfield != NULL) or the scalar cs.  G may be NULL (NoSource: f = 0f0). */
static void wo_acoustic_dynamics(const wo_ops *o, const float *xin, const float *cfield, float cs, const float *G,
                                 float sfac, float *kout)
{
    const int nx = o->nx, ny = o->ny;
    const size_t P = (size_t)nx * ny;
    const float *U = xin, *Vx = xin + P, *Vy = xin + 2 * P, *Px = xin + 3 * P, *Py = xin + 4 * P, *Om = xin + 5 * P;
#pragma omp parallel
    {
        float *Wcol = (float *)malloc(sizeof(float) * 3 * (size_t)nx);
#pragma omp for schedule(static)
        for (int j = 0; j < ny; ++j) {
            /* rows of W = U + f needed for the y stencil at row j */
            int jlo = j == 0 ? 0 : (j == ny - 1 ? ny - 3 : j - 1);
            for (int r = 0; r < 3; ++r) {
                int jj = jlo + r;
                for (int i = 0; i < nx; ++i) {
                    size_t id = (size_t)jj * nx + i;
                    float f = G ? G[id] * sfac : 0.0f;
                    Wcol[r * nx + i] = U[id] + f;
                }
            }
            const float *Wm = Wcol, *W0 = Wcol + nx, *Wp = Wcol + 2 * nx; /* rows jlo..jlo+2 */
            const float *Wrow = (j == 0) ? Wm : (j == ny - 1 ? Wp : W0);  /* row j itself  */
            for (int i = 0; i < nx; ++i) {
                size_t id = (size_t)j * nx + i;
                float c = cfield ? cfield[id] : cs;
                float b = c * c;
                float sx = o->sx[i], sy = o->sy[j];
                float Vxx = wo_d(o, Vx + (size_t)j * nx, i, nx, 1);
                float Vyy = wo_d(o, Vy + i, j, ny, nx);
                float Ux = wo_d(o, Wrow, i, nx, 1);
                float Uy;
                if (j == 0)
                    Uy = (o->fwd[0] * Wm[i] + o->fwd[1] * W0[i]) + o->fwd[2] * Wp[i];
                else if (j == ny - 1)
                    Uy = (o->bwd[0] * Wm[i] + o->bwd[1] * W0[i]) + o->bwd[2] * Wp[i];
                else
                    Uy = o->cm * Wm[i] + o->cp * Wp[i];
                float u = U[id];
                float dU = (((b * (Vxx + Vyy) + Px[id]) + Py[id]) - (sx + sy) * u) - Om[id];
                float bc = (i == 0 || j == 0 || i == nx - 1 || j == ny - 1) ? 0.0f : 1.0f;
                kout[id] = bc * dU;
                kout[P + id] = Ux - sx * Vx[id];
                kout[2 * P + id] = Uy - sy * Vy[id];
                kout[3 * P + id] = (b * sx) * Vyy;
                kout[4 * P + id] = (b * sy) * Vxx;
                kout[5 * P + id] = (sx * sy) * u;
            }
        }
        free(Wcol);
    }
}

/* src/dynamics.jl:179-188: the AcousticDynamics{TwoDim} call: 12 planes in, 12 out. */
void wo_rhs(int nx, int ny, const float *x, const float *sx, const float *sy, float c0, const float *state,
            const float *cfield /* NULL -> scalar c0 for the total set too */, const float *G, float sfac, float *kout)
{
    wo_ops o;
    wo_make_ops(&o, nx, ny, x, sx, sy);
    size_t P = (size_t)nx * ny;
    wo_acoustic_dynamics(&o, state, cfield, c0, G, sfac, kout);
    wo_acoustic_dynamics(&o, state + 6 * P, NULL, c0, G, sfac, kout + 6 * P);
}

/* `grad * u` along x (axis 0) or y (axis 1) of an (nx, ny) plane: operators.jl:45-46. */
void wo_gradient(int nx, int ny, const float *x, int axis, const float *u, float *out)
{
    wo_ops o;
    wo_make_ops(&o, nx, ny, x, NULL, NULL);
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i)
            out[(size_t)j * nx + i] = axis == 0 ? wo_d(&o, u + (size_t)j * nx, i, nx, 1) : wo_d(&o, u + i, j, ny, nx);
}

/* src/env.jl:105-111 raw sums for one state: [sum U_tot^2, sum U_inc^2, sum (U_tot-U_inc)^2],
 * squares in fp32, accumulated in double (the reference's sum order is unspecified). */
static void wo_energy(size_t P, const float *state, double out[3])
{
    const float *ut = state, *ui = state + 6 * P;
    double a = 0, b = 0, c = 0;
    for (size_t k = 0; k < P; ++k) {
        float t = ut[k], i = ui[k], s = t - i;
        a += (double)(t * t);
        b += (double)(i * i);
        c += (double)(s * s);
    }
    out[0] = a; out[1] = b; out[2] = c;
}

/* Integrator (dynamics.jl:37-53) with runge_kutta (dynamics.jl:9-16), the C/F closures of
 * env.jl:99-102, and the energy sums of env.jl:105-111.
 *   state   : 12*nx*ny, in/out (u at tspan[0] -> u at tspan[nsteps])
 *   design  : d0/d1 = M*4 floats (px,py,r,c) of DesignInterpolator.initial/.final; M = 0 -> scalar c0
 *   G       : source shape or NULL; freq
 *   esum    : (nsteps+1)*3 doubles (raw sums, not yet scaled by dx*dy) or NULL
 *   frames  : nframes*12*nx*ny floats receiving the state after step frame_steps[k] (0 = initial) or NULL
 * Returns 0. */
int wo_integrate(int nx, int ny, const float *x, const float *y, const float *sx, const float *sy, float c0, float dt,
                 float *state, const float *tspan, int nsteps, const float *G, float freq, int M, const float *d0,
                 const float *d1, float ti, float tf, double *esum, float *frames, const int *frame_steps,
                 int nframes, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    wo_ops o;
    wo_make_ops(&o, nx, ny, x, sx, sy);
    const size_t P = (size_t)nx * ny, N = 12 * P;
    float *k = (float *)malloc(sizeof(float) * N);
    float *acc = (float *)malloc(sizeof(float) * N);
    float *ys = (float *)malloc(sizeof(float) * N);
    float *cf = M > 0 ? (float *)malloc(sizeof(float) * P) : NULL;
    float *cyl = M > 0 ? (float *)malloc(sizeof(float) * 4 * (size_t)M) : NULL;
    if (!k || !acc || !ys || (M > 0 && (!cf || !cyl))) return 1;

    const float hdt = 0.5f * dt;
    const float sixth = 1.0f / 6.0f;

    if (esum) wo_energy(P, state, esum);
    for (int f = 0; f < nframes; ++f)
        if (frame_steps[f] == 0) memcpy(frames + (size_t)f * N, state, sizeof(float) * N);

    for (int s = 0; s < nsteps; ++s) {
        const float t = tspan[s];
        const float tq[4] = {t, t + hdt, t + hdt, t + dt};
        const float aq[3] = {hdt, hdt, dt};
        const float *yin = state;
        for (int q = 0; q < 4; ++q) {
            if (M > 0 && q != 2) { /* stages 2 and 3 share t + dt/2 */
                wo_design_at(M, d0, d1, ti, tf, tq[q], cyl);
                wo_speed_field(nx, ny, x, y, M, cyl, c0, cf);
            }
            float sf = G ? wo_source_factor(tq[q], freq) : 0.0f;
            wo_acoustic_dynamics(&o, yin, cf, c0, G, sf, k);
            wo_acoustic_dynamics(&o, yin + 6 * P, NULL, c0, G, sf, k + 6 * P);
            if (q == 0) {
#pragma omp parallel for schedule(static)
                for (size_t e = 0; e < N; ++e) { acc[e] = k[e]; ys[e] = state[e] + aq[0] * k[e]; }
            } else if (q < 3) {
                const float a = aq[q];
#pragma omp parallel for schedule(static)
                for (size_t e = 0; e < N; ++e) {
                    float kk = k[e];
                    acc[e] = acc[e] + 2.0f * kk;
                    /* yin == ys here: k was fully formed above, so in-place is safe */
                    ys[e] = state[e] + a * kk;
                }
            } else {
#pragma omp parallel for schedule(static)
                for (size_t e = 0; e < N; ++e) {
                    float du = (sixth * (acc[e] + k[e])) * dt;
                    state[e] = state[e] + du;
                }
            }
            yin = ys;
        }
        if (esum) wo_energy(P, state, esum + 3 * (size_t)(s + 1));
        for (int f = 0; f < nframes; ++f)
            if (frame_steps[f] == s + 1) memcpy(frames + (size_t)f * N, state, sizeof(float) * N);
    }
    free(k); free(acc); free(ys); free(cf); free(cyl);
    return 0;
}

int wo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
