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

}  // namespace

void launch_latent(const LatentArgs &a, hipStream_t s) { hipLaunchKernelGGL(k_latent, dim3(a.B), dim3(1024), 0, s, a); }

}  // namespace wv
