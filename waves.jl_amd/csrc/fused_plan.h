// Host-side planning of the fused step kernel: tile decomposition, per-tile variant and per-tile cylinder culling.
// Pure C++ (no HIP), shared by kernels_fused.hip and the CPU emulation harness in tests/cpu_emu.
#pragma once
#include <math.h>

#include <algorithm>
#include <vector>

#include "fused_body.h"

namespace wv {

struct HostPlan {
    int nx = 0, ny = 0;
    int RY = 0;                     // region rows = NW * RPT
    std::vector<TileDesc> tiles;    // launch order: expensive variants first, so the cheap tiles fill the tail
    int count[3] = {0, 0, 0};       // tiles per variant
    bool monotonic = true;          // x[] and y[] strictly increasing (needed for bounding-box culling)
};

// n cells in pieces of at most omax, sizes differing by at most one (never a sliver: every piece >= 3 when n >= 8,
// which the one-sided boundary stencil -- it reaches two cells inward -- relies on).
inline void plan_split(int n, int omax, std::vector<int> &start, std::vector<int> &len)
{
    const int pieces = (n + omax - 1) / omax;
    const int base = n / pieces, rem = n % pieces;
    start.clear();
    len.clear();
    int s = 0;
    for (int k = 0; k < pieces; ++k) {
        const int l = base + (k < rem ? 1 : 0);
        start.push_back(s);
        len.push_back(l);
        s += l;
    }
}

// aux_zero: Psi_x, Psi_y, Omega are zero at every cell with sigma_x = sigma_y = 0 (and every state buffer is clean
// there), so tiles wholly inside that zone may run the 6-field FAST variant.
inline bool plan_build_tiles(HostPlan &pl, int nx, int ny, int RY, const float *x, const float *y, const float *sx,
                             const float *sy, bool aux_zero)
{
    pl.nx = nx;
    pl.ny = ny;
    pl.RY = RY;
    pl.tiles.clear();
    pl.count[0] = pl.count[1] = pl.count[2] = 0;
    pl.monotonic = true;
    for (int i = 1; i < nx; ++i)
        if (!(x[i] > x[i - 1])) pl.monotonic = false;
    for (int j = 1; j < ny; ++j)
        if (!(y[j] > y[j - 1])) pl.monotonic = false;
    std::vector<int> xs, xl, ys, yl;
    plan_split(nx, FT_X - 2 * FT_H, xs, xl);
    plan_split(ny, RY - 2 * FT_H, ys, yl);
    for (int l : xl)
        if (l < 3) return false;
    for (int l : yl)
        if (l < 3) return false;
    std::vector<TileDesc> all;
    int slot = 0;
    for (size_t b = 0; b < ys.size(); ++b)
        for (size_t a = 0; a < xs.size(); ++a) {
            TileDesc t{};
            t.x0 = xs[a];
            t.y0 = ys[b];
            t.ox = xl[a];
            t.oy = yl[b];
            t.cyl_begin = 0;
            t.cyl_count = 0;
            t.slot = slot++;
            const int rx0 = t.x0 - FT_H, rx1 = t.x0 + t.ox + FT_H - 1;  // region, inclusive
            const int ry0 = t.y0 - FT_H, ry1 = t.y0 + t.oy + FT_H - 1;
            const bool edge = rx0 <= 0 || ry0 <= 0 || rx1 >= nx - 1 || ry1 >= ny - 1;
            bool pml = false;
            for (int i = std::max(rx0, 0); i <= std::min(rx1, nx - 1); ++i)
                if (sx[i] != 0.0f) pml = true;
            for (int j = std::max(ry0, 0); j <= std::min(ry1, ny - 1); ++j)
                if (sy[j] != 0.0f) pml = true;
            t.variant = edge ? VAR_GEN : ((pml || !aux_zero) ? VAR_MID : VAR_FAST);
            pl.count[t.variant]++;
            all.push_back(t);
        }
    for (int v = VAR_GEN; v >= VAR_FAST; --v)
        for (const TileDesc &t : all)
            if (t.variant == v) pl.tiles.push_back(t);
    return true;
}

// Per-tile list of the cylinders whose disc can reach the tile's region at ANY of the `rows` stage times of this
// integrate call.  Conservative bounding boxes with a safety margin far above fp32 round-off: a culled cylinder must
// have mask == false at every cell of the region, in which case dropping it is exact (it would add +0).
inline void plan_build_cyl(HostPlan &pl, const float *x, const float *y, const Cyl *table, int M, int rows,
                           std::vector<int> &idx)
{
    idx.clear();
    if (M <= 0) {
        for (TileDesc &t : pl.tiles) t.cyl_begin = t.cyl_count = 0;
        return;
    }
    struct Box { double x0, x1, y0, y1; bool ok; };
    std::vector<Box> box(M);
    for (int m = 0; m < M; ++m) {
        double pxmin = INFINITY, pxmax = -INFINITY, pymin = INFINITY, pymax = -INFINITY, r2 = 0.0;
        bool ok = true;
        for (int r = 0; r < rows; ++r) {
            const Cyl &c = table[(size_t)r * M + m];
            if (!isfinite(c.px) || !isfinite(c.py) || !isfinite(c.r2)) ok = false;
            pxmin = std::min(pxmin, (double)c.px);
            pxmax = std::max(pxmax, (double)c.px);
            pymin = std::min(pymin, (double)c.py);
            pymax = std::max(pymax, (double)c.py);
            r2 = std::max(r2, (double)c.r2);
        }
        const double rad = sqrt(r2) * (1.0 + 1e-5);
        const double mx = 1e-4 * (1.0 + fabs(pxmin) + fabs(pxmax) + rad);
        const double my = 1e-4 * (1.0 + fabs(pymin) + fabs(pymax) + rad);
        box[m] = Box{pxmin - rad - mx, pxmax + rad + mx, pymin - rad - my, pymax + rad + my, ok};
    }
    for (TileDesc &t : pl.tiles) {
        if (!pl.monotonic) {
            t.cyl_begin = 0;
            t.cyl_count = -1;
            continue;
        }
        const int rx0 = std::max(t.x0 - FT_H, 0), rx1 = std::min(t.x0 + t.ox + FT_H - 1, pl.nx - 1);
        const int ry0 = std::max(t.y0 - FT_H, 0), ry1 = std::min(t.y0 + t.oy + FT_H - 1, pl.ny - 1);
        const double xa = x[rx0], xb = x[rx1], ya = y[ry0], yb = y[ry1];
        t.cyl_begin = (int)idx.size();
        t.cyl_count = 0;
        for (int m = 0; m < M; ++m) {
            const Box &b = box[m];
            const bool miss = b.ok && (b.x1 < xa || b.x0 > xb || b.y1 < ya || b.y0 > yb);
            if (!miss) {
                idx.push_back(m);
                t.cyl_count++;
            }
        }
    }
}

}  // namespace wv
