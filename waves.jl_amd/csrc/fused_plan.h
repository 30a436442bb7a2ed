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
    int RYF = 0, RYP = 0;           // region rows of FAST tiles / of MID and GEN tiles
    std::vector<TileDesc> tiles;    // launch order (see plan_order)
    int count[3] = {0, 0, 0};       // tiles per variant
    bool monotonic = true;          // x[] and y[] strictly increasing (needed for bounding-box culling)
};

// n cells starting at `first` in pieces of at most omax, sizes differing by at most one (never a sliver: a run of
// >= 5 cells cut into pieces of <= 8 keeps every piece >= 3, which the one-sided boundary stencil -- it reaches two
// cells inward -- relies on).
inline void plan_split(int first, int n, int omax, std::vector<int> &start, std::vector<int> &len)
{
    const int pieces = (n + omax - 1) / omax;
    const int base = n / pieces, rem = n % pieces;
    int s = first;
    for (int k = 0; k < pieces; ++k) {
        const int l = base + (k < rem ? 1 : 0);
        start.push_back(s);
        len.push_back(l);
        s += l;
    }
}

// clean[i]: a FAST tile may own output cell i along this axis -- every cell within the 4-cell halo is strictly inside
// the domain and has sigma == 0 there.
inline std::vector<char> plan_clean_axis(int n, const float *sig)
{
    std::vector<char> c(n, 0);
    for (int i = 0; i < n; ++i) {
        bool ok = i - FT_H >= 1 && i + FT_H <= n - 2;
        for (int k = std::max(i - FT_H, 0); ok && k <= std::min(i + FT_H, n - 1); ++k) ok = sig[k] == 0.0f;
        c[i] = ok ? 1 : 0;
    }
    return c;
}

// Launch order.  Blocks are dealt round-robin over the 8 XCDs (block b and b + 8 share an L2: MI355X_MICROARCH.md,
// "Workgroup dispatch"), so the tiles are cut into 8 spatially contiguous groups of equal estimated cost, and launch
// position i takes the next tile of group i % 8: neighbouring tiles -- which re-read each other's halo rows, and
// re-read next step what they wrote this step -- meet in the same L2.  Inside a group the expensive (12-field) tiles
// go first so the cheap ones fill the tail.  Placement only affects speed, never results.
inline void plan_order(std::vector<TileDesc> &natural, std::vector<TileDesc> &out, bool xcd_aware)
{
    auto cost = [](const TileDesc &t) {
        const double cells = (double)(t.ox + 2 * FT_H) * (t.oy + 2 * FT_H);
        return cells * (t.variant == VAR_FAST ? 1.0 : (t.variant == VAR_MID ? 2.4 : 2.8));
    };
    out.clear();
    const int G = xcd_aware ? 8 : 1;
    double total = 0.0;
    for (const TileDesc &t : natural) total += cost(t);
    std::vector<std::vector<TileDesc>> grp(G);
    double acc = 0.0;
    for (const TileDesc &t : natural) {
        int g = (int)(acc / (total / G + 1e-9));
        if (g >= G) g = G - 1;
        grp[g].push_back(t);
        acc += cost(t);
    }
    for (auto &g : grp)
        std::stable_sort(g.begin(), g.end(), [](const TileDesc &a, const TileDesc &b) { return a.variant > b.variant; });
    std::vector<size_t> pos(G, 0);
    const size_t n = natural.size();
    for (size_t i = 0; out.size() < n; ++i) {
        int g = (int)(i % G);
        if (pos[g] >= grp[g].size()) {  // this XCD's group ran dry: take from the fullest remaining one
            size_t best = 0;
            int bg = -1;
            for (int k = 0; k < G; ++k)
                if (grp[k].size() - pos[k] > best) {
                    best = grp[k].size() - pos[k];
                    bg = k;
                }
            if (bg < 0) break;
            g = bg;
        }
        out.push_back(grp[g][pos[g]++]);
    }
}

// aux_zero: Psi_x, Psi_y, Omega are zero at every cell with sigma_x = sigma_y = 0 (and every state buffer is clean
// there), so tiles wholly inside that zone may run the 6-field FAST variant -- with taller tiles (RYF rows), since a
// FAST thread carries half the state of a MID/GEN thread.  Everything else is cut into RYP-row tiles.
inline bool plan_build_tiles(HostPlan &pl, int nx, int ny, int RYF, int RYP, const float *x, const float *y,
                             const float *sx, const float *sy, bool aux_zero, bool xcd_aware)
{
    pl.nx = nx;
    pl.ny = ny;
    pl.RYF = RYF;
    pl.RYP = RYP;
    pl.tiles.clear();
    pl.count[0] = pl.count[1] = pl.count[2] = 0;
    pl.monotonic = true;
    for (int i = 1; i < nx; ++i)
        if (!(x[i] > x[i - 1])) pl.monotonic = false;
    for (int j = 1; j < ny; ++j)
        if (!(y[j] > y[j - 1])) pl.monotonic = false;
    const int OYF = RYF - 2 * FT_H, OYP = RYP - 2 * FT_H;
    if (OYF < 3 || OYP < 3 || nx < 8 || ny < 8) return false;
    std::vector<int> xs, xl;
    plan_split(0, nx, FT_X - 2 * FT_H, xs, xl);
    for (int l : xl)
        if (l < 3) return false;
    const std::vector<char> cx = plan_clean_axis(nx, sx), cy = plan_clean_axis(ny, sy);
    std::vector<TileDesc> all;
    int slot = 0;
    for (size_t a = 0; a < xs.size(); ++a) {
        bool strip_clean = aux_zero;
        for (int i = xs[a]; i < xs[a] + xl[a]; ++i) strip_clean = strip_clean && cx[i];
        // rows: maximal runs of equal cleanliness, each cut evenly
        std::vector<int> ys, yl;
        std::vector<char> tall;
        int j = 0;
        while (j < ny) {
            int e = j;
            while (e < ny && cy[e] == cy[j]) ++e;
            const bool t = strip_clean && cy[j];
            const size_t before = ys.size();
            plan_split(j, e - j, t ? OYF : OYP, ys, yl);
            tall.insert(tall.end(), ys.size() - before, t ? 1 : 0);
            j = e;
        }
        for (size_t b = 0; b < ys.size(); ++b) {
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
            for (int jj = std::max(ry0, 0); jj <= std::min(ry1, ny - 1); ++jj)
                if (sy[jj] != 0.0f) pml = true;
            t.variant = edge ? VAR_GEN : ((pml || !aux_zero) ? VAR_MID : VAR_FAST);
            if (tall[b] && t.variant != VAR_FAST) return false;                  // cannot happen: see plan_clean_axis
            if (t.oy > (t.variant == VAR_FAST ? OYF : OYP)) return false;
            if ((t.x0 == 0 || t.x0 + t.ox == nx) && t.ox < 3) return false;      // boundary stencil reaches 2 inward
            if ((t.y0 == 0 || t.y0 + t.oy == ny) && t.oy < 3) return false;
            pl.count[t.variant]++;
            all.push_back(t);
        }
    }
    plan_order(all, pl.tiles, xcd_aware);
    return pl.tiles.size() == all.size();
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
