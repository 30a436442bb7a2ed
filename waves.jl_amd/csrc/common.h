// Internal definitions shared by the kernels and the C-ABI host layer of libwaves_amd.so.
// Target: gfx950 (MI355X) only.  All floating-point code in this library is compiled with
// -ffp-contract=off: the reference's CPU integrator never fuses a*b+c, and the cylinder mask
// (x-px)^2+(y-py)^2 < r^2 flips cells on 1-ulp differences (hard part 2).
#pragma once
#include <hip/hip_runtime.h>

#include "types.h"

namespace wv {

// `grad * u` for one output element (src/operators.jl:45-46): SparseArrays accumulates the row's
// non-zeros in ascending column order from zero, each product rounded separately.
template <class At>
__device__ __forceinline__ float deriv(const Ops &o, int i, int n, At at)
{
    if (i == 0) return (o.f0 * at(0) + o.f1 * at(1)) + o.f2 * at(2);
    if (i == n - 1) return (o.b0 * at(n - 3) + o.b1 * at(n - 2)) + o.b2 * at(n - 1);
    return o.cm * at(i - 1) + o.cp * at(i + 1);
}

// speed(design, grid, c0) at one cell (src/designs.jl:99-116): c0*[no cylinder covers the cell] + sum_m mask_m*c_m,
// ascending m.  Squares are literal products, the two squares are added, compared with r*r (precomputed on the host
// in fp32).  No FMA may be formed here.
__device__ __forceinline__ float speed_at(float x, float y, const Cyl *__restrict__ cyl, int M, float c0)
{
    int count = 0;
    float cd = 0.0f;
    for (int m = 0; m < M; ++m) {
        const Cyl q = cyl[m];
        const float ddx = x - q.px;
        const float ddy = y - q.py;
        const float d2 = ddx * ddx + ddy * ddy;
        const bool in = d2 < q.r2;
        count += in ? 1 : 0;
        cd = cd + (in ? q.c : 0.0f);
    }
    const float C0 = count == 0 ? c0 : 0.0f;
    return C0 + cd;
}

// One element of state(env) (src/env.jl:132-137): channel `src` resized to rx x ry at pixel (i, j).  The rule is stated at
// k_observation (kernels_aux.hip); one function, because the resident kernel also produces observations itself at the end of
// an action (FusedParams::ob_out) and the two must agree bit for bit.  `ld(k)` reads element k of the channel.
template <class Ld>
__device__ __forceinline__ float obs_pixel(Ld ld, int nx, int ny, int rx, int ry, int i, int j)
{
    const double sx = (double)nx / (double)rx, sy = (double)ny / (double)ry;
    double xo = sx * ((double)(i + 1) - 0.5) + 0.5 - 1.0;  // 0-based coordinate in the original
    double yo = sy * ((double)(j + 1) - 0.5) + 0.5 - 1.0;
    xo = xo < 0.0 ? 0.0 : (xo > (double)(nx - 1) ? (double)(nx - 1) : xo);
    yo = yo < 0.0 ? 0.0 : (yo > (double)(ny - 1) ? (double)(ny - 1) : yo);
    const int i0 = (int)floor(xo), j0 = (int)floor(yo);
    const int i1 = i0 + 1 < nx ? i0 + 1 : nx - 1, j1 = j0 + 1 < ny ? j0 + 1 : ny - 1;
    const double fx = xo - (double)i0, fy = yo - (double)j0;
    const double a00 = ld((size_t)j0 * nx + i0), a10 = ld((size_t)j0 * nx + i1);
    const double a01 = ld((size_t)j1 * nx + i0), a11 = ld((size_t)j1 * nx + i1);
    const double lo = (1.0 - fx) * a00 + fx * a10;
    const double hi = (1.0 - fx) * a01 + fx * a11;
    return (float)((1.0 - fy) * lo + fy * hi);
}

// wave64 sum, returned to every lane.  Six DPP adds (butterfly inside a row of 16 lanes, then row broadcasts) instead of
// six dependent LDS-crossbar shuffles: the sum of a wave is on the critical path between two steps of k_steps_resident.
// The order of the additions is fixed, so results are deterministic (cdna_hip_programming.md Appendix B "Reduction").
__device__ __forceinline__ float wave_sum(float v)
{
#ifdef WV_SHFL_SUM
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return __shfl(v, 0, 64);
#endif
#define WV_DPP_ADD(ctrl, rmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
    WV_DPP_ADD(0xb1, 0xf);   // quad_perm [1,0,3,2]: neighbour
    WV_DPP_ADD(0x4e, 0xf);   // quad_perm [2,3,0,1]: other pair -> every lane holds its quad's sum
    WV_DPP_ADD(0x141, 0xf);  // row_half_mirror: the other quad of the 8 lanes
    WV_DPP_ADD(0x140, 0xf);  // row_mirror: the other half of the 16 lanes -> every lane holds its row's sum
    WV_DPP_ADD(0x142, 0xa);  // row_bcast15 into rows 1 and 3: they hold the sum of 32 lanes
    WV_DPP_ADD(0x143, 0xc);  // row_bcast31 into rows 2 and 3: lane 63 holds the sum of the wave
#undef WV_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

}  // namespace wv
