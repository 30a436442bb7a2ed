// Plain C++ types shared by device code, the host layer and the CPU emulation harness of the fused kernel
// (tests/cpu_emu).  No HIP headers here.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define WV_HD __host__ __device__ __forceinline__
#else
#define WV_HD inline
#endif

namespace wv {

constexpr int kFields = 12;  // src/dynamics.jl:185-187
constexpr int kWave = 64;    // CDNA wavefront

// Non-zeros of gradient(x) (src/operators.jl:10-22), each one coef/(2*Delta) rounded on its own.
struct Ops {
    float cm, cp;      // row i:   -1/(2D) at i-1, +1/(2D) at i+1
    float f0, f1, f2;  // row 0:   [-3, 4, -1]/(2D) at 0, 1, 2
    float b0, b1, b2;  // row n-1: [ 1,-4,  3]/(2D) at n-3, n-2, n-1
};

// One cylinder at one stage time: centre, r*r and wave speed (src/designs.jl:99-116).
struct Cyl {
    float px, py, r2, c;
};

struct alignas(8) F2 {
    float x, y;  // (total-field value, incident-field value) of the same quantity at one cell
};

}  // namespace wv
