// Batched 1-D latent dynamics: `AcousticDynamics{OneDim}` integrated with classical RK4, as the surrogate
// models drive it -- z = iter(z0, t, [C, F, PML]) with C a LinearInterpolation of latent wave-speed fields, F a Source and
// PML a learned damping profile (src/dynamics.jl:190-222, 9-16, 37-49; src/utils.jl:69-98; src/sources.jl:21-23;
// src/model/acoustic_energy_model.jl:89-107).
//
// One block per batch element, one thread per cell (n <= 1024: the reference's scripts use 1024): the four fields of a
// cell, the RK accumulator and the stage input live in registers for the WHOLE integration; what a stage needs from the
// neighbouring cells (V and U + f of both wave sets) goes through LDS.  All steps run in one launch; the only memory
// traffic is the output z (n x 4 x B x (steps + 1), written once) -- HBM-bound by that write, and tiny.
// fp32 in the reference's operation order, no FMA (-ffp-contract=off).
#include "kernels.h"

namespace wv {

namespace {

__global__ __launch_bounds__(1024) void k_latent(LatentArgs a)
{
    __shared__ float sh[4][1024 + 2];
    const int i = threadIdx.x, b = blockIdx.x, n = a.n, B = a.B, K = a.K;
    const bool on = i < n;
    const int ic = on ? i : n - 1;
    float u[4], acc[4], y[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) u[f] = a.z0[(size_t)ic + (size_t)n * (f + 4 * (size_t)b)];
    const float shape = a.shape[(size_t)ic + (size_t)n * b];
    const float sigma = a.pml_scale * a.PML[(size_t)ic + (size_t)n * b];     // pml_scale .* PML
    const float bc = (i == 0 || i == n - 1) ? 0.0f : 1.0f;                   // build_dirichlet(::OneDim)
    const Ops o = a.ops;
    const float c0 = a.c0;
    // (c0 * grad): every coefficient scaled and rounded  (dynamics.jl:207: the generic left fold of `c0 * grad * (U_inc .+ f)`)
    const float scm = c0 * o.cm, scp = c0 * o.cp, sf0 = c0 * o.f0, sf1 = c0 * o.f1, sf2 = c0 * o.f2, sb0 = c0 * o.b0, sb1 = c0 * o.b1,
                sb2 = c0 * o.b2;
    float *out = a.z;
    if (on)
#pragma unroll
        for (int f = 0; f < 4; ++f) out[(size_t)i + (size_t)n * (f + 4 * (size_t)b)] = u[f];
    for (int s = 0; s < a.steps; ++s) {
        const float t0 = a.t[(size_t)s * B + b];
        const float tq[3] = {t0, t0 + a.hdt, t0 + a.dt};
#pragma unroll
        for (int S = 1; S <= 4; ++S) {
            const int q = S == 1 ? 0 : (S == 4 ? 2 : 1);
            const float t = tq[q];
            const float(&yin)[4] = S == 1 ? u : y;
            // f = shape .* sin.(2f0 * pi * t * freq)   (the time factor comes from the host table, like the 2-D path)
            const float fsrc = shape * a.sfac[((size_t)s * 3 + q) * B + b];
            const float wt = yin[0] + fsrc, wi = yin[2] + fsrc;
            __syncthreads();  // the previous stage's reads are done
            sh[0][i + 1] = yin[1];
            sh[1][i + 1] = wt;
            sh[2][i + 1] = yin[3];
            sh[3][i + 1] = wi;
            __syncthreads();
            // c = C(t): linear_interp(X, Y, t), utils.jl:69-86 -- every interval contributes (value .* mask)
            float x0 = 0.0f, y0 = 0.0f, dydx = 0.0f;
            const float xe = a.X[(size_t)(K - 1) + (size_t)K * b];
            for (int k = 0; k + 1 < K; ++k) {
                const float l = a.X[(size_t)k + (size_t)K * b], r = a.X[(size_t)k + 1 + (size_t)K * b];
                const float yl = a.Y[(size_t)ic + (size_t)n * (k + (size_t)K * b)], yr = a.Y[(size_t)ic + (size_t)n * (k + 1 + (size_t)K * b)];
                const float m = ((l <= t && t < r) || (r == xe && xe == t)) ? 1.0f : 0.0f;
                const float slope = (yr - yl) / ((r - t) - (l - t));
                x0 = x0 + l * m;
                y0 = y0 + yl * m;
                dydx = dydx + slope * m;
            }
            const float c = y0 + (t - x0) * dydx;
            float gVt, gWt, gVi, gWi;   // grad * V_tot, grad * (U_tot + f), grad * V_inc, (c0 * grad) * (U_inc + f)
            if (i == 0) {
                gVt = (o.f0 * sh[0][1] + o.f1 * sh[0][2]) + o.f2 * sh[0][3];
                gWt = (o.f0 * sh[1][1] + o.f1 * sh[1][2]) + o.f2 * sh[1][3];
                gVi = (o.f0 * sh[2][1] + o.f1 * sh[2][2]) + o.f2 * sh[2][3];
                gWi = (sf0 * sh[3][1] + sf1 * sh[3][2]) + sf2 * sh[3][3];
            } else if (i >= n - 1) {
                gVt = (o.b0 * sh[0][n - 2] + o.b1 * sh[0][n - 1]) + o.b2 * sh[0][n];
                gWt = (o.b0 * sh[1][n - 2] + o.b1 * sh[1][n - 1]) + o.b2 * sh[1][n];
                gVi = (o.b0 * sh[2][n - 2] + o.b1 * sh[2][n - 1]) + o.b2 * sh[2][n];
                gWi = (sb0 * sh[3][n - 2] + sb1 * sh[3][n - 1]) + sb2 * sh[3][n];
            } else {
                gVt = o.cm * sh[0][i] + o.cp * sh[0][i + 2];
                gWt = o.cm * sh[1][i] + o.cp * sh[1][i + 2];
                gVi = o.cm * sh[2][i] + o.cp * sh[2][i + 2];
                gWi = scm * sh[3][i] + scp * sh[3][i + 2];
            }
            const float ac = c0 * c;
            float k[4];
            k[0] = (ac * gVt - sigma * yin[0]) * bc;    // dU_tot .* bc
            k[1] = ac * gWt - sigma * yin[1];           // dV_tot
            k[2] = (c0 * gVi - sigma * yin[2]) * bc;    // dU_inc .* bc
            k[3] = gWi - sigma * yin[3];                // dV_inc
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (S == 1) {
                    acc[f] = k[f];
                    y[f] = u[f] + a.hdt * k[f];
                } else if (S == 2) {
                    acc[f] = __builtin_fmaf(2.0f, k[f], acc[f]);   // 2*k exact: == acc + 2*k
                    y[f] = u[f] + a.hdt * k[f];
                } else if (S == 3) {
                    acc[f] = __builtin_fmaf(2.0f, k[f], acc[f]);
                    y[f] = u[f] + a.dt * k[f];
                } else {
                    const float du = ((1.0f / 6.0f) * (acc[f] + k[f])) * a.dt;
                    u[f] = u[f] + du;
                }
            }
        }
        if (on)
#pragma unroll
            for (int f = 0; f < 4; ++f) out[(size_t)i + (size_t)n * (f + 4 * ((size_t)b + (size_t)B * (s + 1)))] = u[f];
    }
}


// ---- adjoint_sensitivity (src/dynamics.jl:97-121) ------------------------------------------------------------------
// The reference takes the vector-Jacobian product of one `runge_kutta` call from Zygote at every saved time, last to
// first.  Here the same product is written out: per step the four stage inputs are recomputed (registers), then the four
// stages are pulled back in reverse -- the transposed gradient needs the neighbours' cotangents, which go through LDS like
// the forward stencil -- and the parameter gradients a cell owns (its column of C.Y, its entry of F.shape and of PML) are
// accumulated by the cell's own thread, so nothing is atomic.  Same launch shape as k_latent.
__device__ __forceinline__ float grad_T(const float *w, int i, int n, float cm, float cp, float f0, float f1, float f2, float b0,
                                        float b1, float b2)
{
    // column i of grad: row i-1 (cp) and row i+1 (cm) where those are interior rows, plus the one-sided rows 0 and n-1
    float r = 0.0f;
    if (i >= 2) r = r + cp * w[i];            // w is stored at [cell + 1]: w[i] == cell i-1
    if (i <= n - 3) r = r + cm * w[i + 2];
    if (i < 3) r = r + (i == 0 ? f0 : (i == 1 ? f1 : f2)) * w[1];
    if (i >= n - 3) r = r + (i == n - 3 ? b0 : (i == n - 2 ? b1 : b2)) * w[n];
    return r;
}

__global__ __launch_bounds__(1024) void k_latent_adjoint(LatentAdjArgs aa)
{
    __shared__ float sh[4][1024 + 2];
    const LatentArgs &a = aa.f;
    const int i = threadIdx.x, b = blockIdx.x, n = a.n, B = a.B, K = a.K;
    const bool on = i < n;
    const int ic = on ? i : n - 1;
    const float shape = a.shape[(size_t)ic + (size_t)n * b];
    const float sigma = a.pml_scale * a.PML[(size_t)ic + (size_t)n * b];
    const float bc = (i == 0 || i == n - 1) ? 0.0f : 1.0f;
    const Ops o = a.ops;
    const float c0 = a.c0;
    const float scm = c0 * o.cm, scp = c0 * o.cp, sf0 = c0 * o.f0, sf1 = c0 * o.f1, sf2 = c0 * o.f2, sb0 = c0 * o.b0, sb1 = c0 * o.b1,
                sb2 = c0 * o.b2;
    const float xe = a.X[(size_t)(K - 1) + (size_t)K * b];
    float lam[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // dL_dz0 = adj[:, :, :, end] * 0
    float gsh = 0.0f, gp = 0.0f;
    bool first = true;
    for (int s = a.steps; s >= 0; --s) {
        float ys[4][4];        // the four stage inputs
        float acS[4], gVtS[4], gWtS[4], x0S[4], tS[4], sfS[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const size_t at = (size_t)ic + (size_t)n * (f + 4 * ((size_t)b + (size_t)B * s));
            ys[0][f] = a.z[at];
            lam[f] = lam[f] + aa.adj[at];
        }
        const float t0 = a.t[(size_t)s * B + b];
        const float tq[3] = {t0, t0 + a.hdt, t0 + a.dt};
        // ---- forward: stage inputs and what the pullback of each stage needs
#pragma unroll
        for (int S = 1; S <= 4; ++S) {
            const int q = S == 1 ? 0 : (S == 4 ? 2 : 1);
            const float t = tq[q];
            const float(&yin)[4] = ys[S - 1];
            const float sfac = a.sfac[((size_t)s * 3 + q) * B + b];
            const float fsrc = shape * sfac;
            __syncthreads();
            sh[0][i + 1] = yin[1];
            sh[1][i + 1] = yin[0] + fsrc;
            sh[2][i + 1] = yin[3];
            sh[3][i + 1] = yin[2] + fsrc;
            __syncthreads();
            float x0 = 0.0f, y0 = 0.0f, dydx = 0.0f;
            for (int k = 0; k + 1 < K; ++k) {
                const float l = a.X[(size_t)k + (size_t)K * b], r = a.X[(size_t)k + 1 + (size_t)K * b];
                const float yl = a.Y[(size_t)ic + (size_t)n * (k + (size_t)K * b)], yr = a.Y[(size_t)ic + (size_t)n * (k + 1 + (size_t)K * b)];
                const float m = ((l <= t && t < r) || (r == xe && xe == t)) ? 1.0f : 0.0f;
                const float slope = (yr - yl) / ((r - t) - (l - t));
                x0 = x0 + l * m;
                y0 = y0 + yl * m;
                dydx = dydx + slope * m;
            }
            const float c = y0 + (t - x0) * dydx;
            float gVt, gWt, gVi, gWi;
            if (i == 0) {
                gVt = (o.f0 * sh[0][1] + o.f1 * sh[0][2]) + o.f2 * sh[0][3];
                gWt = (o.f0 * sh[1][1] + o.f1 * sh[1][2]) + o.f2 * sh[1][3];
                gVi = (o.f0 * sh[2][1] + o.f1 * sh[2][2]) + o.f2 * sh[2][3];
                gWi = (sf0 * sh[3][1] + sf1 * sh[3][2]) + sf2 * sh[3][3];
            } else if (i >= n - 1) {
                gVt = (o.b0 * sh[0][n - 2] + o.b1 * sh[0][n - 1]) + o.b2 * sh[0][n];
                gWt = (o.b0 * sh[1][n - 2] + o.b1 * sh[1][n - 1]) + o.b2 * sh[1][n];
                gVi = (o.b0 * sh[2][n - 2] + o.b1 * sh[2][n - 1]) + o.b2 * sh[2][n];
                gWi = (sb0 * sh[3][n - 2] + sb1 * sh[3][n - 1]) + sb2 * sh[3][n];
            } else {
                gVt = o.cm * sh[0][i] + o.cp * sh[0][i + 2];
                gWt = o.cm * sh[1][i] + o.cp * sh[1][i + 2];
                gVi = o.cm * sh[2][i] + o.cp * sh[2][i + 2];
                gWi = scm * sh[3][i] + scp * sh[3][i + 2];
            }
            const float ac = c0 * c;
            acS[S - 1] = ac; gVtS[S - 1] = gVt; gWtS[S - 1] = gWt; x0S[S - 1] = x0; tS[S - 1] = t; sfS[S - 1] = sfac;
            if (S < 4) {
                float k[4];
                k[0] = (ac * gVt - sigma * yin[0]) * bc;
                k[1] = ac * gWt - sigma * yin[1];
                k[2] = (c0 * gVi - sigma * yin[2]) * bc;
                k[3] = gWi - sigma * yin[3];
                const float h = S == 3 ? a.dt : a.hdt;
#pragma unroll
                for (int f = 0; f < 4; ++f) ys[S][f] = ys[0][f] + h * k[f];
            }
        }
        // ---- reverse: du = (1/6 * (k1 + 2 k2 + 2 k3 + k4)) * dt
        float kb[4][4], zb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float sb = (lam[f] * a.dt) * (1.0f / 6.0f);
            kb[0][f] = sb; kb[1][f] = 2.0f * sb; kb[2][f] = 2.0f * sb; kb[3][f] = sb;
        }
        float shb = 0.0f, pb = 0.0f;
#pragma unroll
        for (int S = 4; S >= 1; --S) {
            const float(&q)[4] = kb[S - 1];
            const float(&yin)[4] = ys[S - 1];
            const float ac = acS[S - 1];
            const float g1 = q[0] * bc, q1 = q[1], g3 = q[2] * bc, q3 = q[3];
            __syncthreads();
            sh[0][i + 1] = on ? ac * q1 : 0.0f;
            sh[1][i + 1] = on ? ac * g1 : 0.0f;
            sh[2][i + 1] = on ? c0 * g3 : 0.0f;
            sh[3][i + 1] = on ? q3 : 0.0f;
            __syncthreads();
            const float GT0 = grad_T(sh[0], i, n, o.cm, o.cp, o.f0, o.f1, o.f2, o.b0, o.b1, o.b2);
            const float GT1 = grad_T(sh[1], i, n, o.cm, o.cp, o.f0, o.f1, o.f2, o.b0, o.b1, o.b2);
            const float GT2 = grad_T(sh[2], i, n, o.cm, o.cp, o.f0, o.f1, o.f2, o.b0, o.b1, o.b2);
            const float GT3 = grad_T(sh[3], i, n, scm, scp, sf0, sf1, sf2, sb0, sb1, sb2);
            float yb[4];
            yb[0] = GT0 - sigma * g1;
            yb[1] = GT1 - sigma * q1;
            yb[2] = GT3 - sigma * g3;
            yb[3] = GT2 - sigma * q3;
            const float fbar = GT0 + GT3;
            const float abar = g1 * gVtS[S - 1] + q1 * gWtS[S - 1];
            const float cbar = c0 * abar;
            const float sbar = -(((g1 * yin[0] + q1 * yin[1]) + g3 * yin[2]) + q3 * yin[3]);
            // c = y0 + (t - x0) * dydx: the interval(s) the mask selects
            if (on) {
                const float t = tS[S - 1];
                for (int k = 0; k + 1 < K; ++k) {
                    const float l = a.X[(size_t)k + (size_t)K * b], r = a.X[(size_t)k + 1 + (size_t)K * b];
                    if ((l <= t && t < r) || (r == xe && xe == t)) {
                        const float wr = (cbar * (t - x0S[S - 1])) / ((r - t) - (l - t));
                        float *gl = aa.gY + (size_t)i + (size_t)n * (k + (size_t)K * b), *gr = gl + n;
                        *gr = *gr + wr;
                        *gl = *gl + (cbar - wr);
                    }
                }
            }
            shb = first && S == 4 ? fbar * sfS[S - 1] : shb + fbar * sfS[S - 1];
            pb = first && S == 4 ? a.pml_scale * sbar : pb + a.pml_scale * sbar;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                zb[f] = zb[f] + yb[f];
                if (S > 1) kb[S - 2][f] = kb[S - 2][f] + (S == 4 ? a.dt : a.hdt) * yb[f];
            }
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) lam[f] = lam[f] + zb[f];
        gsh = first ? shb : gsh + shb;
        gp = first ? pb : gp + pb;
        first = false;
    }
    if (on) {
#pragma unroll
        for (int f = 0; f < 4; ++f) aa.gz0[(size_t)i + (size_t)n * (f + 4 * (size_t)b)] = lam[f];
        aa.gshape[(size_t)i + (size_t)n * b] = gsh;
        aa.gPML[(size_t)i + (size_t)n * b] = gp;
    }
}

}  // namespace

void launch_latent(const LatentArgs &a, hipStream_t s) { hipLaunchKernelGGL(k_latent, dim3(a.B), dim3(1024), 0, s, a); }

void launch_latent_adjoint(const LatentAdjArgs &a, hipStream_t s) { hipLaunchKernelGGL(k_latent_adjoint, dim3(a.f.B), dim3(1024), 0, s, a); }

}  // namespace wv
