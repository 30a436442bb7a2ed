// One kernel per Runge-Kutta stage ("staged" implementation).
//
// This is the simple, obviously-correct device path: one thread per cell, neighbours read straight from
// global memory (L1/L2 serve the re-use), the whole right-hand side of src/dynamics.jl:151-188 and the RK4
// update of src/dynamics.jl:9-16 for that stage fused in one pass.  It is the in-library cross-check of the fused
// step kernel (kernels_fused.hip) and serves wv_rhs.  Roofline: HBM/L2-bound, ~848 B per cell-update over the four
// stages -- which is why it is not the fast path.
#include "kernels.h"

namespace wv {

namespace {

constexpr int BX = 64;  // x along the 64 lanes of a wave: 256-B coalesced rows
constexpr int BY = 4;

// dyn(x, t, theta) at one cell: k[0..11].  src/dynamics.jl:151-188.
__device__ __forceinline__ void rhs_cell(const Grid &g, const StageIO &io, int i, int j, float k[kFields])
{
    const int nx = g.nx, ny = g.ny;
    const size_t P = g.P;
    const size_t id = (size_t)j * nx + i;
    const float sx = g.sx[i];
    const float sy = g.sy[j];
    const float sxy = sx + sy;   // (sigma_x .+ sigma_y)
    const float sxsy = sx * sy;  // sigma_x .* sigma_y
    const float bcv = (i == 0 || j == 0 || i == nx - 1 || j == ny - 1) ? 0.0f : 1.0f;  // src/dims.jl:117-124
    const float c = io.M > 0 ? speed_at(g.x[i], g.y[j], io.cyl, io.M, g.c0) : g.c0;    // C(t), src/env.jl:99
    const float btot = c * c;                                                           // c .^ 2
    const float *__restrict__ G = io.G;
    const float sfac = io.sfac;
#pragma unroll
    for (int set = 0; set < 2; ++set) {
        const float *__restrict__ U = io.yin + (size_t)(6 * set) * P;
        const float *__restrict__ Vx = U + P;
        const float *__restrict__ Vy = U + 2 * P;
        const float *__restrict__ Px = U + 3 * P;
        const float *__restrict__ Py = U + 4 * P;
        const float *__restrict__ Om = U + 5 * P;
        const float b = set == 0 ? btot : g.c0sq;
        auto W = [&](size_t q) -> float {  // U .+ f, f = shape .* sin(...) (src/sources.jl:67-69) or 0f0 (NoSource)
            const float f = G ? G[q] * sfac : 0.0f;
            return U[q] + f;
        };
        const float Vxx = deriv(g.ops, i, nx, [&](int ii) { return Vx[(size_t)j * nx + ii]; });
        const float Vyy = deriv(g.ops, j, ny, [&](int jj) { return Vy[(size_t)jj * nx + i]; });
        const float Ux = deriv(g.ops, i, nx, [&](int ii) { return W((size_t)j * nx + ii); });
        const float Uy = deriv(g.ops, j, ny, [&](int jj) { return W((size_t)jj * nx + i); });
        const float u = U[id];
        const float dU = (((b * (Vxx + Vyy) + Px[id]) + Py[id]) - sxy * u) - Om[id];
        k[6 * set + 0] = bcv * dU;
        k[6 * set + 1] = Ux - sx * Vx[id];
        k[6 * set + 2] = Uy - sy * Vy[id];
        k[6 * set + 3] = (b * sx) * Vyy;
        k[6 * set + 4] = (b * sy) * Vxx;
        k[6 * set + 5] = sxsy * u;
    }
}

// STAGE 0..3: RK stages (src/dynamics.jl:9-16).  STAGE 4: out = k (RHS only).
template <int STAGE>
__global__ __launch_bounds__(BX *BY) void k_stage(Grid g, StageIO io)
{
    const int i = blockIdx.x * BX + threadIdx.x;
    const int j = blockIdx.y * BY + threadIdx.y;
    float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f;
    if (i < g.nx && j < g.ny) {
        const size_t P = g.P;
        const size_t id = (size_t)j * g.nx + i;
        float k[kFields];
        rhs_cell(g, io, i, j, k);
        float unew0 = 0.0f, unew6 = 0.0f;
#pragma unroll
        for (int e = 0; e < kFields; ++e) {
            const size_t q = (size_t)e * P + id;
            if (STAGE == 0) {
                io.acc[q] = k[e];
                io.out[q] = io.u[q] + io.a * k[e];  // u .+ 0.5f0*dt*k1
            } else if (STAGE == 1 || STAGE == 2) {
                io.acc[q] = __builtin_fmaf(2.0f, k[e], io.acc[q]);  // 2*k is exact: fma == acc + 2*k bit for bit
                io.out[q] = io.u[q] + io.a * k[e];
            } else if (STAGE == 3) {
                const float du = ((1.0f / 6.0f) * (io.acc[q] + k[e])) * io.dt;  // (1/6f0 * (...)) * dt
                const float un = io.u[q] + du;                                   // _u .+ du, src/dynamics.jl:41
                io.out[q] = un;
                if (e == 0) unew0 = un;
                if (e == 6) unew6 = un;
            } else {
                io.out[q] = k[e];
            }
        }
        if (STAGE == 3) {
            const float usc = unew0 - unew6;  // src/env.jl:107
            e0 = unew0 * unew0;
            e1 = unew6 * unew6;
            e2 = usc * usc;
            if (io.traj_tot) io.traj_tot[id] = unew0;
            if (io.traj_inc) io.traj_inc[id] = unew6;
        }
    }
    if (STAGE == 3 && io.epart) {
        __shared__ float red[3][BY];
        e0 = wave_sum(e0);
        e1 = wave_sum(e1);
        e2 = wave_sum(e2);
        if (threadIdx.x == 0) {
            red[0][threadIdx.y] = e0;
            red[1][threadIdx.y] = e1;
            red[2][threadIdx.y] = e2;
        }
        __syncthreads();
        if (threadIdx.x == 0 && threadIdx.y == 0) {
            const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
            for (int c = 0; c < 3; ++c) io.epart[b * 3 + c] = (red[c][0] + red[c][1]) + (red[c][2] + red[c][3]);
        }
    }
}

dim3 stage_grid(const Grid &g) { return dim3((g.nx + BX - 1) / BX, (g.ny + BY - 1) / BY, 1); }

}  // namespace

int staged_energy_blocks(const Grid &g)
{
    const dim3 gr = stage_grid(g);
    return (int)(gr.x * gr.y);
}

void launch_stage(const Grid &g, const StageIO &io, int stage, hipStream_t s)
{
    const dim3 gr = stage_grid(g), bl(BX, BY, 1);
    switch (stage) {
        case 0: hipLaunchKernelGGL(k_stage<0>, gr, bl, 0, s, g, io); break;
        case 1: hipLaunchKernelGGL(k_stage<1>, gr, bl, 0, s, g, io); break;
        case 2: hipLaunchKernelGGL(k_stage<2>, gr, bl, 0, s, g, io); break;
        case 3: hipLaunchKernelGGL(k_stage<3>, gr, bl, 0, s, g, io); break;
        default: hipLaunchKernelGGL(k_stage<4>, gr, bl, 0, s, g, io); break;
    }
}

}  // namespace wv
