// Small kernels around the integrator: energy reductions (K5), the wave-speed field (K1) and Gaussian source shape
// (K2) as stand-alone passes, the gradient operator, plane copies.  All are single-pass, HBM-bound, coalesced along x.
#include "kernels.h"

namespace wv {

namespace {

constexpr int TPB = 256;

// [sum U_tot^2, sum U_inc^2, sum (U_tot-U_inc)^2] per block.  src/env.jl:105-111 (squares and difference in fp32).
__global__ __launch_bounds__(TPB) void k_energy_partial(const float *__restrict__ ut, const float *__restrict__ ui,
                                                        size_t P, float *__restrict__ epart)
{
    float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f;
    for (size_t q = (size_t)blockIdx.x * TPB + threadIdx.x; q < P; q += (size_t)gridDim.x * TPB) {
        const float t = ut[q], i = ui[q], s = t - i;
        e0 += t * t;
        e1 += i * i;
        e2 += s * s;
    }
    __shared__ float red[3][TPB / kWave];
    e0 = wave_sum(e0);
    e1 = wave_sum(e1);
    e2 = wave_sum(e2);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][w] = e0;
        red[1][w] = e1;
        red[2][w] = e2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            epart[(size_t)blockIdx.x * 3 + c] = (red[c][0] + red[c][1]) + (red[c][2] + red[c][3]);
    }
}

// Deterministic second pass: one block per saved time point sums that point's per-block partials in double, in a
// fixed order, rounds once to fp32 and applies `* dOmega` in fp32 (src/env.jl:108-111).
// Row 0 (the initial state) may live elsewhere: the partial sums the previous call ended on.  The signal may be pinned
// host memory (the trace then needs no copy of its own); block 0 also forwards the give-up word of the resident kernel.
__global__ __launch_bounds__(TPB) void k_energy_final(const float *__restrict__ row0, const float *__restrict__ epart, int nblocks,
                                                      float dOmega, float *__restrict__ signal, const int *__restrict__ flag_src,
                                                      int *__restrict__ flag_dst)
{
    const int r = blockIdx.x;
    const float *p = r == 0 ? row0 : epart + (size_t)r * nblocks * 3;
    if (r == 0 && threadIdx.x == 0 && flag_dst) *flag_dst = flag_src ? *flag_src : 0;
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += TPB) {
        s[0] += (double)p[b * 3 + 0];
        s[1] += (double)p[b * 3 + 1];
        s[2] += (double)p[b * 3 + 2];
    }
    __shared__ double red[3][TPB];
#pragma unroll
    for (int c = 0; c < 3; ++c) red[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int off = TPB / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) signal[r * 3 + threadIdx.x] = (float)red[threadIdx.x][0] * dOmega;
}

// speed(design, grid, c0): src/designs.jl:110-116.
__global__ __launch_bounds__(TPB) void k_speed_field(Grid g, const Cyl *__restrict__ cyl, int M, float *__restrict__ out)
{
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i < g.nx && j < g.ny) out[(size_t)j * g.nx + i] = speed_at(g.x[i], g.y[j], cyl, M, g.c0);
}

// build_normal(grid, mu, sigma, a): src/utils.jl:12-18.  exp is evaluated in double and rounded once (Julia's
// exp(::Float32) is accurate to < 1 ulp); everything else is fp32 in the reference's order.
__global__ __launch_bounds__(TPB) void k_gaussian(Grid g, int K, const float *__restrict__ mu,
                                                  const float *__restrict__ sigma, const float *__restrict__ a,
                                                  float *__restrict__ out)
{
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= g.nx || j >= g.ny) return;
    const float x = g.x[i], y = g.y[j];
    const float two_pi = 6.2831855f;  // 2.0f0 * pi in Float32
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float ddx = x - mu[k];      // mu is K x 2 column-major
        const float ddy = y - mu[K + k];
        const float d2 = ddx * ddx + ddy * ddy;
        const float s2 = sigma[k] * sigma[k];
        const float coef = 1.0f / (two_pi * s2);
        const float e = (float)exp((double)((-d2) / (2.0f * s2)));
        acc = acc + (coef * a[k]) * e;
    }
    out[(size_t)j * g.nx + i] = acc;
}

__global__ __launch_bounds__(TPB) void k_scale(const float *__restrict__ in, float f, float *__restrict__ out, size_t n)
{
    for (size_t q = (size_t)blockIdx.x * TPB + threadIdx.x; q < n; q += (size_t)gridDim.x * TPB) out[q] = in[q] * f;
}

__global__ __launch_bounds__(TPB) void k_gradient(Grid g, int axis, const float *__restrict__ u, float *__restrict__ out)
{
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= g.nx || j >= g.ny) return;
    float d;
    if (axis == 0)
        d = deriv(g.ops, i, g.nx, [&](int ii) { return u[(size_t)j * g.nx + ii]; });
    else
        d = deriv(g.ops, j, g.ny, [&](int jj) { return u[(size_t)jj * g.nx + i]; });
    out[(size_t)j * g.nx + i] = d;
}

__global__ __launch_bounds__(TPB) void k_copy_planes(const float *__restrict__ state, size_t P, float *__restrict__ tot,
                                                     float *__restrict__ inc)
{
    for (size_t q = (size_t)blockIdx.x * TPB + threadIdx.x; q < P; q += (size_t)gridDim.x * TPB) {
        if (tot) tot[q] = state[q];
        if (inc) inc[q] = state[6 * P + q];
    }
}

// state(env): x = imresize(cat(env.wave[:, :, 1, :], env.source.shape, dims = 3), env.resolution)   src/env.jl:132-137.
// imresize is Images.jl / ImageTransformations (third-party, version unpinned, absent here).  Its published rule,
// restated: linear B-spline interpolation of the original (flat beyond the edge pixels), sampled pixel-centre aligned at
// x_o = (n/r)*(i - 0.5) + 0.5 (1-based) per resized axis, evaluated in Float64 and rounded once to Float32; axes that
// keep their size (the 4 channels) are copied.  The summation order is OUR definition (x inside, y outside): parity with
// Julia is unpinned; the tests compare bit for bit with the same rule restated on the CPU.
__global__ __launch_bounds__(TPB) void k_observation(const float *__restrict__ f0, const float *__restrict__ f1,
                                                     const float *__restrict__ f2, const float *__restrict__ G, int nx,
                                                     int ny, int rx, int ry, float *__restrict__ out)
{
    const size_t total = (size_t)rx * ry * 4;
    for (size_t q = (size_t)blockIdx.x * TPB + threadIdx.x; q < total; q += (size_t)gridDim.x * TPB) {
        const int i = (int)(q % rx), j = (int)((q / rx) % ry), ch = (int)(q / ((size_t)rx * ry));
        const float *src = ch == 0 ? f0 : (ch == 1 ? f1 : (ch == 2 ? f2 : G));
        // NoSource: the shape channel is zero
        out[q] = src ? obs_pixel([src](size_t k) { return src[k]; }, nx, ny, rx, ry, i, j) : 0.0f;
    }
}

dim3 grid2d(const Grid &g) { return dim3((g.nx + 63) / 64, (g.ny + 3) / 4, 1); }
int grid1d(size_t n) { size_t b = (n + TPB - 1) / TPB; return (int)(b > 2048 ? 2048 : (b ? b : 1)); }

}  // namespace

// Self-test of what the halo exchange of k_steps_resident relies on: a 16-byte-aligned, 16-byte agent-scope (sc1) buffer
// access is never observed torn.  Writer blocks rewrite granules {x, f(x), ~x, x ^ K} in a tight loop, reader blocks on
// every XCD load them and check that the four dwords belong to one x (tools/micro/tear16.hip is the long-running form).
typedef unsigned int st_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_selftest_granules(unsigned char *buf, unsigned bytes, int iters, int nwriters,
                                                           unsigned stride, unsigned long long *out /* checked, torn */)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)bytes, 0x00020000);
    const unsigned slot = (blockIdx.x % nwriters) * 256u + threadIdx.x;
    const unsigned off = (unsigned)(((unsigned long long)slot * stride) % bytes) & ~15u;
    if ((int)blockIdx.x < nwriters) {
        for (int i = 1; i <= iters; ++i) {
            const unsigned x = (unsigned)i * 977u + slot;
            __builtin_amdgcn_raw_buffer_store_b128(st_u4{x, x * 2654435761u + 1u, ~x, x ^ 0x9e3779b9u}, rs, (int)off, 0, 16);
            if ((i & 7) == 0) __builtin_amdgcn_s_waitcnt(0);
        }
    } else {
        unsigned long long bad = 0, seen = 0;
        for (int i = 0; i < iters; ++i) {
            const st_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);
            if (v.x == 0 && v.y == 0 && v.z == 0 && v.w == 0) continue;  // not written yet
            ++seen;
            if (!(v.y == v.x * 2654435761u + 1u && v.z == ~v.x && v.w == (v.x ^ 0x9e3779b9u))) ++bad;
        }
        atomicAdd(out, seen);
        if (bad) atomicAdd(out + 1, bad);
    }
#endif
}

void launch_selftest_granules(unsigned char *buf, unsigned bytes, int iters, int nwriters, unsigned stride,
                              unsigned long long *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_granules, dim3(nwriters * 4), dim3(256), 0, s, buf, bytes, iters, nwriters, stride, out);
}

void launch_energy_partial(const Grid &g, const float *state, float *epart, int nblocks, hipStream_t s)
{
    hipLaunchKernelGGL(k_energy_partial, dim3(nblocks), dim3(TPB), 0, s, state, state + 6 * g.P, g.P, epart);
}

void launch_energy_final(const float *row0, const float *epart, int nrows, int nblocks, float dOmega, float *signal,
                         const int *flag_src, int *flag_dst, hipStream_t s)
{
    hipLaunchKernelGGL(k_energy_final, dim3(nrows), dim3(TPB), 0, s, row0, epart, nblocks, dOmega, signal, flag_src, flag_dst);
}

void launch_speed_field(const Grid &g, const Cyl *cyl, int M, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_speed_field, grid2d(g), dim3(TPB), 0, s, g, cyl, M, out);
}

void launch_gaussian(const Grid &g, int K, const float *mu, const float *sigma, const float *a, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_gaussian, grid2d(g), dim3(TPB), 0, s, g, K, mu, sigma, a, out);
}

void launch_scale(const float *in, float f, float *out, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_scale, dim3(grid1d(n)), dim3(TPB), 0, s, in, f, out, n);
}

void launch_gradient(const Grid &g, int axis, const float *u, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_gradient, grid2d(g), dim3(TPB), 0, s, g, axis, u, out);
}

void launch_observation(const Grid &g, const float *f0, const float *f1, const float *f2, const float *G, int rx, int ry,
                        float *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_observation, dim3(grid1d((size_t)rx * ry * 4)), dim3(TPB), 0, s, f0, f1, f2, G, g.nx, g.ny, rx, ry,
                       out);
}

void launch_copy_planes(const float *state, size_t P, float *tot, float *inc, hipStream_t s)
{
    hipLaunchKernelGGL(k_copy_planes, dim3(grid1d(P)), dim3(TPB), 0, s, state, P, tot, inc);
}

}  // namespace wv
