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

// wave64 sum by shuffles (cdna_hip_programming.md Appendix B "Reduction")
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

}  // namespace wv
