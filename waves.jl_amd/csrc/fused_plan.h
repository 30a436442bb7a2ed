// Host-side planning of the fused step kernel: tile decomposition, per-tile variant and per-tile cylinder culling.
// Pure C++ (no HIP), shared by kernels_fused.hip and the CPU emulation harness in tests/cpu_emu.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "fused_body.h"

namespace wv {

struct HostPlan {
    int nx = 0, ny = 0;
    int RYF = 0, RYB = 0, RYP = 0;  // region rows of AUX_NONE tiles / AUX_PX+AUX_PY tiles / AUX_ALL tiles
    std::vector<TileDesc> tiles;    // launch order: band-major, inside a band see plan_order; refined per call (plan_build_cyl)
    std::vector<TileDesc> base;     // the order plan_build_tiles produced: every per-call refinement starts from it
    std::vector<int> band_begin;    // band b = tiles[band_begin[b] .. band_begin[b+1]); bands are contiguous ranges of
                                    // x-strips, so a band's tiles only read halo cells of the two adjacent bands
    int count[4] = {0, 0, 0, 0};    // tiles per field set
    bool monotonic = true;          // x[] and y[] strictly increasing (needed for bounding-box culling)
    // plan_pair_order's per-tiling tables (built on first use, dropped by plan_build_tiles): the weight order of the tiles when
    // none evaluates a cylinder, and their geometric order -- a call then only sorts the tiles that DO evaluate cylinders
    int pair_static_for = 0;        // the pair_cus the tables were made for (0: none)
    std::vector<double> pair_w0;    // [base position] weight without cylinders
    std::vector<int> pair_ord0;     // base positions by (descending weight, position)
    std::vector<int> pair_geo;      // base positions by (y0, x0)
    std::vector<TileDesc> pair_out; // scratch of a call ...
    std::vector<std::pair<double, int>> pair_cyl;
    std::vector<int> pair_rank, pair_hl, pair_ll;
};

// n cells starting at `first` in pieces of at most omax, sizes differing by at most one (never a sliver: a run of
// >= 3 cells cut into pieces of <= omax >= 8 keeps every piece >= 3, which the one-sided boundary stencil -- it
// reaches two cells inward -- relies on).
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

// zero[i]: sigma == 0 at every cell within the 4-cell halo of output cell i (clipped to the domain), i.e. a tile
// that owns cell i may drop the PML terms of this axis.
inline std::vector<char> plan_axis_zero(int n, const float *sig)
{
    std::vector<char> c(n, 0);
    for (int i = 0; i < n; ++i) {
        bool ok = true;
        for (int k = std::max(i - FT_H, 0); ok && k <= std::min(i + FT_H, n - 1); ++k) ok = sig[k] == 0.0f;
        c[i] = ok ? 1 : 0;
    }
    return c;
}

struct PlanRun {
    int start, len;
    bool zero;
};

// maximal runs of equal `zero`; a run shorter than 3 cells at either end of the axis is merged into its neighbour
// (as non-zero) so that the piece holding a boundary cell is never a sliver
inline std::vector<PlanRun> plan_runs(const std::vector<char> &zero)
{
    std::vector<PlanRun> runs;
    const int n = (int)zero.size();
    int j = 0;
    while (j < n) {
        int e = j;
        while (e < n && zero[e] == zero[j]) ++e;
        runs.push_back(PlanRun{j, e - j, zero[j] != 0});
        j = e;
    }
    // A sigma == 0 run that reaches a domain boundary (no PML there, e.g. pml_scale = 0) is cut so that the pieces
    // holding boundary cells form their own (non-zero class) run: AUX_NONE tiles then never need the boundary code.
    const int carve = 2 * FT_H;
    if (!runs.empty() && runs.front().zero) {
        PlanRun &f = runs.front();
        if (f.len >= 2 * carve + 3) {
            runs.insert(runs.begin(), PlanRun{f.start, carve, false});
            runs[1].start += carve;
            runs[1].len -= carve;
        } else {
            f.zero = false;
        }
    }
    if (!runs.empty() && runs.back().zero) {
        PlanRun &b = runs.back();
        if (b.len >= carve + 3) {
            b.len -= carve;
            runs.push_back(PlanRun{b.start + b.len, carve, false});
        } else {
            b.zero = false;
        }
    }
    while (runs.size() > 1 && runs.front().len < 3) {
        runs[1].start = runs[0].start;
        runs[1].len += runs[0].len;
        runs[1].zero = false;
        runs.erase(runs.begin());
    }
    while (runs.size() > 1 && runs.back().len < 3) {
        runs[runs.size() - 2].len += runs.back().len;
        runs[runs.size() - 2].zero = false;
        runs.pop_back();
    }
    return runs;
}

// Launch order.  Blocks are dealt round-robin over the 8 XCDs (block b and b + 8 share an L2: MI355X_MICROARCH.md,
// "Workgroup dispatch"), so the tiles are cut into 8 spatially contiguous groups of equal estimated cost, and launch
// position i takes the next tile of group i % 8: neighbouring tiles -- which re-read each other's halo rows, and
// re-read next step what they wrote this step -- meet in the same L2.  Inside a group the expensive tiles go first
// so the cheap ones fill the tail.  Placement only affects speed, never results.
inline double plan_tile_cost(const TileDesc &t)
{
    const double cells = (double)(t.ox + 2 * FT_H) * (t.oy + 2 * FT_H);
    return cells * (t.aux == AUX_NONE ? 1.0 : (t.aux == AUX_ALL ? 2.2 : 1.45)) * (t.edge ? 1.1 : 1.0);
}

inline void plan_order(std::vector<TileDesc> &natural, std::vector<TileDesc> &out, bool xcd_aware)
{
    out.clear();
    const int G = xcd_aware ? 8 : 1;
    double total = 0.0;
    for (const TileDesc &t : natural) total += plan_tile_cost(t);
    std::vector<std::vector<TileDesc>> grp(G);
    double acc = 0.0;
    for (const TileDesc &t : natural) {
        int g = (int)(acc / (total / G + 1e-9));
        if (g >= G) g = G - 1;
        grp[g].push_back(t);
        acc += plan_tile_cost(t);
    }
    for (auto &g : grp)
        std::stable_sort(g.begin(), g.end(),
                         [](const TileDesc &a, const TileDesc &b) { return plan_tile_cost(a) > plan_tile_cost(b); });
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

// reduced: the state satisfies "Psi_x = 0 wherever sigma_x = 0, Psi_y = 0 wherever sigma_y = 0, Omega = 0 wherever
// sigma_x*sigma_y = 0" (and every state buffer a step writes to holds zeros in the planes a reduced tile leaves out),
// so tiles may carry reduced field sets -- and be taller, since a thread then carries less state per cell.  Otherwise
// every tile is AUX_ALL.
inline bool plan_build_tiles(HostPlan &pl, int nx, int ny, int RYF, int RYB, int RYP, const float *x, const float *y,
                             const float *sx, const float *sy, bool reduced, bool xcd_aware, int nbands = 1, int oyf_cap = 0,
                             int oy_cap_all = 0)
{
    pl.nx = nx;
    pl.ny = ny;
    pl.RYF = RYF;
    pl.RYB = RYB;
    pl.RYP = RYP;
    pl.tiles.clear();
    pl.pair_static_for = 0;
    for (int &c : pl.count) c = 0;
    pl.monotonic = true;
    for (int i = 1; i < nx; ++i)
        if (!(x[i] > x[i - 1])) pl.monotonic = false;
    for (int j = 1; j < ny; ++j)
        if (!(y[j] > y[j - 1])) pl.monotonic = false;
    // rows a tile of each field set owns at most: what its region holds, or less (WAVES_AMD_FUSED_OY="none,px,py,all":
    // more, smaller tiles -- e.g. to fill every block slot of the device in the resident kernel)
    int OY[4] = {RYF - 2 * FT_H, RYB - 2 * FT_H, RYB - 2 * FT_H, RYP - 2 * FT_H};
    {
        static int cap[4] = {0, 0, 0, 0};
        static bool parsed = false;
        if (!parsed) {
            parsed = true;
            if (const char *e = getenv("WAVES_AMD_FUSED_OY")) (void)sscanf(e, "%d,%d,%d,%d", &cap[0], &cap[1], &cap[2], &cap[3]);
        }
        for (int k = 0; k < 4; ++k)
            if (cap[k] >= 8 && cap[k] < OY[k]) OY[k] = cap[k];
        if (oyf_cap >= 8 && oyf_cap < OY[0]) OY[0] = oyf_cap;
        for (int k = 0; k < 4; ++k)
            if (oy_cap_all >= 8 && oy_cap_all < OY[k]) OY[k] = oy_cap_all;
    }
    if (OY[0] < 8 || OY[1] < 8 || OY[3] < 8 || nx < 8 || ny < 8) return false;
    std::vector<int> xs, xl;
    plan_split(0, nx, FT_X - 2 * FT_H, xs, xl);
    for (int l : xl)
        if (l < 3) return false;
    const std::vector<char> zx = plan_axis_zero(nx, sx), zy = plan_axis_zero(ny, sy);
    const std::vector<PlanRun> runs = plan_runs(zy);
    std::vector<TileDesc> all;
    std::vector<size_t> strip_begin;
    int slot = 0;
    for (size_t a = 0; a < xs.size(); ++a) {
        strip_begin.push_back(all.size());
        bool x_zero = reduced;
        for (int i = xs[a]; i < xs[a] + xl[a]; ++i) x_zero = x_zero && zx[i];
        if (xs[a] - FT_H <= 0 || xs[a] + xl[a] + FT_H - 1 >= nx - 1) x_zero = false;  // boundary strips: never AUX_NONE/PY
        for (const PlanRun &run : runs) {
            const bool y_zero = reduced && run.zero;
            const int aux = x_zero ? (y_zero ? AUX_NONE : AUX_PY) : (y_zero ? AUX_PX : AUX_ALL);
            std::vector<int> ys, yl;
            plan_split(run.start, run.len, OY[aux], ys, yl);
            for (size_t b = 0; b < ys.size(); ++b) {
                TileDesc t{};
                t.x0 = xs[a];
                t.y0 = ys[b];
                t.ox = xl[a];
                t.oy = yl[b];
                t.aux = aux;
                t.slot = slot++;
                const int rx0 = t.x0 - FT_H, rx1 = t.x0 + t.ox + FT_H - 1;  // region, inclusive
                const int ry0 = t.y0 - FT_H, ry1 = t.y0 + t.oy + FT_H - 1;
                t.edge = (rx0 <= 0 ? EDGE_L : 0) | (rx1 >= nx - 1 ? EDGE_R : 0) | (ry0 <= 0 ? EDGE_T : 0) |
                         (ry1 >= ny - 1 ? EDGE_B : 0);
                // the field set must be consistent with the damping profile over the whole region
                bool sx_zero = true, sy_zero = true;
                for (int i = std::max(rx0, 0); i <= std::min(rx1, nx - 1); ++i) sx_zero = sx_zero && sx[i] == 0.0f;
                for (int j = std::max(ry0, 0); j <= std::min(ry1, ny - 1); ++j) sy_zero = sy_zero && sy[j] == 0.0f;
                if ((aux == AUX_NONE || aux == AUX_PY) && !sx_zero) return false;
                if ((aux == AUX_NONE || aux == AUX_PX) && !sy_zero) return false;
                if (aux != AUX_ALL && !reduced) return false;
                if (aux == AUX_NONE && t.edge) return false;  // cannot happen: see plan_runs and the strip test above
                if (t.oy > OY[aux] || t.ox > FT_X - 2 * FT_H) return false;
                if ((t.x0 == 0 || t.x0 + t.ox == nx) && t.ox < 3) return false;  // boundary stencil reaches 2 inward
                if ((t.y0 == 0 || t.y0 + t.oy == ny) && t.oy < 3) return false;
                pl.count[aux]++;
                all.push_back(t);
            }
        }
    }
    strip_begin.push_back(all.size());
    // bands: contiguous strip ranges of (nearly) equal cost
    const int nstrips = (int)xs.size();
    const int B = std::max(1, std::min(nbands, nstrips));
    double total = 0.0;
    for (const TileDesc &t : all) total += plan_tile_cost(t);
    pl.tiles.clear();
    pl.band_begin.assign(1, 0);
    int a = 0;
    double acc = 0.0;
    for (int b = 0; b < B; ++b) {
        std::vector<TileDesc> band, ordered;
        const double target = total * (b + 1) / B;
        while (a < nstrips && (band.empty() || nstrips - a > B - 1 - b)) {
            double c = 0.0;
            for (size_t k = strip_begin[a]; k < strip_begin[a + 1]; ++k) c += plan_tile_cost(all[k]);
            if (!band.empty() && b < B - 1 && acc + 0.5 * c > target) break;
            for (size_t k = strip_begin[a]; k < strip_begin[a + 1]; ++k) band.push_back(all[k]);
            acc += c;
            ++a;
        }
        plan_order(band, ordered, xcd_aware);
        pl.tiles.insert(pl.tiles.end(), ordered.begin(), ordered.end());
        pl.band_begin.push_back((int)pl.tiles.size());
    }
    pl.base = pl.tiles;
    return pl.tiles.size() == all.size() && a == nstrips;
}

// Launch order for the resident kernel (all tiles on the device at once, pair_cus = its CU count C, n <= 2C tiles).
// The hardware gives launch position b the CU b mod C, so positions b and b + C share a CU -- and its four SIMDs -- for
// the whole call, while positions n-C .. C-1 have a CU to themselves.  Every tile advances at the pace of the slowest
// one, so: the heaviest 2C-n tiles get the CUs of their own, the next heaviest are paired with the lightest.
// (Measured on MI355X, 700^2: alone on a CU a corner tile takes ~1.5x and a tile that evaluates cylinders ~1.35x the
// time of a plain interior tile; two tiles sharing a CU take ~1.3x the time of the slower one alone.  Pairing the two
// kinds of PML strip with each other instead of with light interior tiles costs 9 %.)  L2 locality is irrelevant here:
// resident tiles exchange halos through memory.  Placement only affects speed, never results.
inline int g_plan_pair_generic = 0;  // tests: != 0 makes plan_pair_order take its generic code
inline void plan_pair_order(HostPlan &pl, int pair_cus)
{
    const int n = (int)pl.tiles.size();
    if (!(pair_cus > 0 && pl.band_begin.size() == 2 && n > pair_cus && n <= 2 * pair_cus)) return;
    // relative cost of a tile: field set x rows in use x cylinders evaluated x boundary code.  WAVES_AMD_PAIR_W
    // ("px,py,all,cyl1,cylk,edge") overrides the factors (tuning runs).
    static double W[6] = {1.22, 1.4, 2.0, 1.6, 0.2, 1.0};  // (end of round 3: py 1.22 -> 1.4 and all 1.75 -> 2.0 -- every corner tile a CU of its own --: -2.6 % at 700^2)
    static bool parsed = false;
    if (!parsed) {
        parsed = true;
        if (const char *e = getenv("WAVES_AMD_PAIR_W")) {
            double v[6];
            if (sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) == 6)
                for (int k = 0; k < 6; ++k) W[k] = v[k];
        }
    }
    auto weight = [&](const TileDesc &t) {
        const int RY = t.aux == AUX_NONE ? pl.RYF : (t.aux == AUX_ALL ? pl.RYP : pl.RYB);
        const double fill = (double)(t.oy + 2 * FT_H) / RY;
        const double set = t.aux == AUX_NONE ? 1.0 : (t.aux == AUX_ALL ? W[2] : (t.aux == AUX_PX ? W[0] : W[1]));
        const int nc = t.cyl_count < 0 ? 8 : t.cyl_count;
        const double wt = fill * set * (nc > 0 ? W[3] + W[4] * (nc - 1) : 1.0) * (t.edge ? W[5] : 1.0);
        // A strip or corner of the PML never counts lighter than a plain interior tile, however few rows it owns: the lightest
        // tiles become the partners of the heaviest, and two PML strips on one CU are the worst pair there is (700^2: with
        // the PY strips weighed 3 % lighter, i.e. below 1, an action takes 980 us instead of 890 -- measured, round 3).
        return (t.aux != AUX_NONE && wt < 1.02) ? 1.02 : wt;
    };
    const int C = pair_cus, alone = 2 * C - n, pairs = n - C;
    static const bool plain = !getenv("WAVES_AMD_PAIR_SHUFFLE") && !getenv("WAVES_AMD_PAIR_KEYS") && !getenv("WAVES_AMD_PAIR_GENERIC") &&
                              !(getenv("WAVES_AMD_PAIR_FLIP") && atoi(getenv("WAVES_AMD_PAIR_FLIP")) != 0);
    if (plain && !g_plan_pair_generic && pl.tiles.size() == pl.base.size()) {
        // The product's rule (below: weight order, partner rule 5) without sorting all tiles per call -- this runs on the host in
        // front of every call, also of those whose tiles cull their cylinders themselves.  pl.tiles is in base order here (the
        // caller's culling starts from pl.base); only tiles that evaluate cylinders weigh differently from call to call.
        // Same result as the generic code (tests/test_plan_cpu.py compares the two; WAVES_AMD_PAIR_GENERIC=1 forces that code).
        if (pl.pair_static_for != C || (int)pl.pair_w0.size() != n) {
            pl.pair_w0.resize(n);
            pl.pair_ord0.resize(n);
            pl.pair_geo.resize(n);
            for (int i = 0; i < n; ++i) {
                TileDesc t = pl.base[i];
                t.cyl_begin = t.cyl_count = 0;
                pl.pair_w0[i] = weight(t);
                pl.pair_ord0[i] = pl.pair_geo[i] = i;
            }
            std::sort(pl.pair_ord0.begin(), pl.pair_ord0.end(), [&](int a, int b) {
                return pl.pair_w0[a] != pl.pair_w0[b] ? pl.pair_w0[a] > pl.pair_w0[b] : a < b;
            });
            std::sort(pl.pair_geo.begin(), pl.pair_geo.end(), [&](int a, int b) {
                const TileDesc &ta = pl.base[a], &tb = pl.base[b];
                return ta.y0 * 4096 + ta.x0 < tb.y0 * 4096 + tb.x0;
            });
            pl.pair_static_for = C;
        }
        // the tiles with cylinders, heaviest first
        std::vector<std::pair<double, int>> &cyl = pl.pair_cyl;
        cyl.clear();
        for (int i = 0; i < n; ++i)
            if (pl.tiles[i].cyl_count != 0) cyl.push_back({-weight(pl.tiles[i]), i});
        std::sort(cyl.begin(), cyl.end());
        const int nc = (int)cyl.size();
        // merged with the rest: rank[position] in the order (descending weight, position)
        std::vector<int> &rank = pl.pair_rank, &hl = pl.pair_hl, &ll = pl.pair_ll;
        rank.resize(n);
        hl.resize(n);
        ll.resize(n);
        {
            int a = 0, b = 0, r = 0;
            while (r < n) {
                while (a < n && pl.tiles[pl.pair_ord0[a]].cyl_count != 0) ++a;
                const bool take_cyl = b < nc && (a >= n || std::make_pair(cyl[b].first, cyl[b].second) <
                                                               std::make_pair(-pl.pair_w0[pl.pair_ord0[a]], pl.pair_ord0[a]));
                if (take_cyl) rank[cyl[b++].second] = r++;
                else rank[pl.pair_ord0[a++]] = r++;
            }
            std::vector<TileDesc> &out = pl.pair_out;
            out.resize(n);
            // heavy tiles in ascending (y, x) take the light tiles in descending (y, x)
            int nh = 0, nl = 0;
            for (int g = 0; g < n; ++g) {
                const int i = pl.pair_geo[g], r2 = rank[i];
                if (r2 < alone) out[pairs + r2] = pl.tiles[i];
                else if (r2 < alone + pairs) hl[nh++] = i;
                else ll[nl++] = i;
            }
            for (int r2 = 0; r2 < pairs; ++r2) {
                const int h = hl[r2], l = ll[pairs - 1 - r2], i = rank[h] - alone;
                out[i] = pl.tiles[h];
                out[C + i] = pl.tiles[l];
            }
            pl.tiles.swap(out);
            return;
        }
    }
    std::vector<std::pair<double, int>> key(n);  // (−weight, position): ascending sort = heaviest first, ties in order
    for (int i = 0; i < n; ++i) key[i] = {-weight(pl.tiles[i]), i};
    std::sort(key.begin(), key.end());
    const std::vector<TileDesc> src = pl.tiles;
    // which of the two tiles of a CU is launched first (the older block wins the issue arbitration): the heavy one by
    // default; WAVES_AMD_PAIR_FLIP=1 (tuning runs) the light one
    static const bool flip = getenv("WAVES_AMD_PAIR_FLIP") && atoi(getenv("WAVES_AMD_PAIR_FLIP")) != 0;
    // WHICH light tile joins which heavy one, and at which launch positions the pairs sit.  WAVES_AMD_PAIR_SHUFFLE=seed,what
    // (experiments): what = 0 partners in weight order (ties in plan order: rounds 2 and 3 until the last day), 1 random partners,
    // 2 random positions, 3 both, 4 ... 10 structured assignments (below)
    std::vector<int> hpos(pairs), lsel(pairs);
    for (int i = 0; i < pairs; ++i) hpos[i] = lsel[i] = i;
    {
        // Default (what = 5): the heavy tiles in ascending y take the light tiles in DESCENDING y -- found at the end of round 3:
        // which light tile joins which heavy one decides 5 % of an action (random partners 880-893 us, partners in weight
        // order with ties in plan order 864-872, this 840-847; a point reflection does as well, a half-domain shift does not).
        unsigned long long seed = 0;
        int what = 5;
        if (const char *e = getenv("WAVES_AMD_PAIR_SHUFFLE")) (void)sscanf(e, "%llu,%d", &seed, &what);
        auto rnd = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(seed >> 33); };
        const int pos_rule = what >= 100 ? what / 100 : 0;  // (experiments: what = 100 * position rule + partner rule)
        what %= 100;
        if (what == 4 || what == 5 || what == 6 || what == 7) {
            // structured alternatives: heavy tiles in ascending y (4, 5) or x (6, 7) order take the light tiles in the same (4, 6)
            // or the opposite (5, 7) order
            std::vector<int> hi(pairs), li(pairs);
            for (int i = 0; i < pairs; ++i) hi[i] = li[i] = i;
            auto ky = [&](int idx) { const TileDesc &t = src[key[idx].second]; return what < 6 ? t.y0 * 4096 + t.x0 : t.x0 * 4096 + t.y0; };
            std::sort(hi.begin(), hi.end(), [&](int a, int b) { return ky(alone + a) < ky(alone + b); });
            std::sort(li.begin(), li.end(), [&](int a, int b) { return ky(n - 1 - a) < ky(n - 1 - b); });
            for (int r = 0; r < pairs; ++r) lsel[hi[r]] = li[(what & 1) ? pairs - 1 - r : r];
            what = 0;
        }
        if (what == 8 || what == 9 || what == 10 || what == 11 || what == 12) {
            // the light partner nearest to the heavy tile's image under a half-domain shift (8), a point reflection (9), or a
            // half-domain shift in y only (10): greedy, heavy tiles in weight order
            std::vector<char> used(pairs, 0);
            for (int i = 0; i < pairs; ++i) {
                const TileDesc &h = src[key[alone + i].second];
                const double hx = h.x0 + 0.5 * h.ox, hy = h.y0 + 0.5 * h.oy;
                double tx = hx, ty = hy;
                if (what == 8) tx = fmod(hx + 0.5 * pl.nx, (double)pl.nx), ty = fmod(hy + 0.5 * pl.ny, (double)pl.ny);
                if (what == 9) tx = pl.nx - hx, ty = pl.ny - hy;
                if (what == 10) ty = fmod(hy + 0.5 * pl.ny, (double)pl.ny);
                if (what == 11) ty = pl.ny - hy;  // the mirror image about the horizontal axis
                if (what == 12) tx = pl.nx - hx;  // ... about the vertical axis
                int best = -1;
                double bd = 1e300;
                for (int j = 0; j < pairs; ++j) {
                    if (used[j]) continue;
                    const TileDesc &l = src[key[n - 1 - j].second];
                    const double dx = l.x0 + 0.5 * l.ox - tx, dy = l.y0 + 0.5 * l.oy - ty, d = dx * dx + dy * dy;
                    if (d < bd) bd = d, best = j;
                }
                used[best] = 1;
                lsel[i] = best;
            }
            what = 0;
        }
        if (pos_rule) {
            // launch positions of the pairs: by the heavy tile's y (1), x (2), y interleaved from both ends (3), slot (4)
            std::vector<int> ord(pairs);
            for (int i = 0; i < pairs; ++i) ord[i] = i;
            auto kk = [&](int i) {
                const TileDesc &t = src[key[alone + i].second];
                return pos_rule == 2 ? t.x0 * 4096 + t.y0 : (pos_rule == 4 ? t.slot : t.y0 * 4096 + t.x0);
            };
            std::sort(ord.begin(), ord.end(), [&](int a, int b) { return kk(a) < kk(b); });
            if (pos_rule == 3) {
                std::vector<int> o2;
                for (int a = 0, b = pairs - 1; a <= b; ++a, --b) {
                    o2.push_back(ord[a]);
                    if (a != b) o2.push_back(ord[b]);
                }
                ord = o2;
            }
            for (int r = 0; r < pairs; ++r) hpos[ord[r]] = r;
        }
        if (what & 1)
            for (int i = pairs - 1; i > 0; --i) std::swap(lsel[i], lsel[rnd() % (unsigned)(i + 1)]);
        if (what & 2)
            for (int i = pairs - 1; i > 0; --i) std::swap(hpos[i], hpos[rnd() % (unsigned)(i + 1)]);
    }
    if (const char *pf = getenv("WAVES_AMD_PAIR_KEYS")) {
        // (experiment, tools/exp_pair_search.py) one key per tile slot from a file: the heavy tiles in ascending key take the
        // light tiles in descending key -- the default rule is this with key = y * 4096 + x
        if (FILE *f = fopen(pf, "r")) {
            std::vector<double> kv(n, 0.0);
            bool ok = true;
            for (int i = 0; i < n && ok; ++i) ok = fscanf(f, "%lf", &kv[i]) == 1;
            fclose(f);
            if (ok) {
                std::vector<int> hi(pairs), li(pairs);
                for (int i = 0; i < pairs; ++i) hi[i] = li[i] = i;
                auto kh = [&](int i) { return kv[src[key[alone + i].second].slot]; };
                auto kl = [&](int i) { return kv[src[key[n - 1 - i].second].slot]; };
                std::stable_sort(hi.begin(), hi.end(), [&](int x, int y) { return kh(x) < kh(y); });
                std::stable_sort(li.begin(), li.end(), [&](int x, int y) { return kl(x) < kl(y); });
                for (int r = 0; r < pairs; ++r) lsel[hi[r]] = li[pairs - 1 - r];
            }
        }
    }
    for (int i = 0; i < alone; ++i) pl.tiles[pairs + i] = src[key[i].second];
    for (int i = 0; i < pairs; ++i) {
        pl.tiles[flip ? C + hpos[i] : hpos[i]] = src[key[alone + i].second];
        pl.tiles[flip ? hpos[i] : C + hpos[i]] = src[key[n - 1 - lsel[i]].second];
    }
}

// Per-tile list of the cylinders whose disc can reach the tile's region at ANY of the `rows` stage times of this
// integrate call.  Conservative bounding boxes with a safety margin far above fp32 round-off: a culled cylinder must
// have mask == false at every cell of the region, in which case dropping it is exact (it would add +0).
// row_lo / row_hi (optional): the rows of the earliest and of the latest stage time.  The table is a linear interpolation
// in time rounded monotonically (v_i + (k * tau), tau = clamp(t) - t_i), so every component takes its extremes over the
// call at those two rows and the bounding boxes need no scan of all `rows`.
inline void plan_build_cyl(HostPlan &pl, const float *x, const float *y, const Cyl *table, int M, int rows,
                           std::vector<int> &idx, bool resort = true, int pair_cus = 0, int row_lo = -1, int row_hi = -1)
{
    idx.clear();
    pl.tiles = pl.base;
    if (M <= 0) {
        for (TileDesc &t : pl.tiles) t.cyl_begin = t.cyl_count = 0;
        plan_pair_order(pl, pair_cus);
        return;
    }
    struct Box { double x0, x1, y0, y1, r; bool ok; };
    std::vector<Box> box(M);
    for (int m = 0; m < M; ++m) {
        double pxmin = INFINITY, pxmax = -INFINITY, pymin = INFINITY, pymax = -INFINITY, r2 = 0.0;
        bool ok = true;
        const bool ends = row_lo >= 0 && row_hi >= 0 && row_lo < rows && row_hi < rows;
        for (int q = 0; q < (ends ? 2 : rows); ++q) {
            const int r = ends ? (q == 0 ? row_lo : row_hi) : q;
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
        const double rr = rad + std::max(mx, my);  // inflated radius; box = swept centres grown by rr
        box[m] = Box{pxmin - rr, pxmax + rr, pymin - rr, pymax + rr, rr, ok};
    }
    // (the box around ALL the cylinders' boxes: a tile outside it is missed by each of them -- four compares instead of M tests
    // for most tiles of a grid whose design sits in one part of it; this loop is host time in front of every call)
    double ux0 = INFINITY, ux1 = -INFINITY, uy0 = INFINITY, uy1 = -INFINITY;
    bool all_ok = true;
    for (const Box &b : box) {
        all_ok = all_ok && b.ok;
        ux0 = std::min(ux0, b.x0), ux1 = std::max(ux1, b.x1), uy0 = std::min(uy0, b.y0), uy1 = std::max(uy1, b.y1);
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
        if (all_ok && (ux1 < xa || ux0 > xb || uy1 < ya || uy0 > yb)) continue;
        for (int m = 0; m < M; ++m) {
            const Box &b = box[m];
            bool miss = b.ok && (b.x1 < xa || b.x0 > xb || b.y1 < ya || b.y0 > yb);
            if (!miss && b.ok) {
                // the boxes overlap: does the (inflated, swept) disc really reach the rectangle?  Distance from the box
                // of possible centres to the region rectangle, against the largest radius (margins already included in
                // b via rad + m*): conservative, so still exact.
                const double cx0 = b.x0 + b.r, cx1 = b.x1 - b.r, cy0 = b.y0 + b.r, cy1 = b.y1 - b.r;  // centre box
                const double ddx = std::max({xa - cx1, cx0 - xb, 0.0});
                const double ddy = std::max({ya - cy1, cy0 - yb, 0.0});
                miss = ddx * ddx + ddy * ddy > b.r * b.r;
            }
            if (!miss) {
                idx.push_back(m);
                t.cyl_count++;
            }
        }
    }
    // Launch order refinement: inside each XCD group (launch positions congruent modulo 8 -- or all tiles when the order
    // is not XCD-aware) the tiles that will take longest (cylinders to evaluate, more fields) go first: the kernel ends
    // when its slowest tile ends, and launch positions are worth up to ~1 us of head start.
    const int ntile = (int)pl.tiles.size();
    const bool will_pair = pair_cus > 0 && pl.band_begin.size() == 2 && ntile > pair_cus && ntile <= 2 * pair_cus;
    for (size_t b = 0; resort && !will_pair && b + 1 < pl.band_begin.size(); ++b) {
        const int lo = pl.band_begin[b], hi = pl.band_begin[b + 1];
        for (int g = 0; g < 8; ++g) {
            std::vector<TileDesc> grp;
            for (int k = lo + g; k < hi; k += 8) grp.push_back(pl.tiles[k]);
            std::stable_sort(grp.begin(), grp.end(), [](const TileDesc &a, const TileDesc &c) {
                auto w = [](const TileDesc &t) { return plan_tile_cost(t) * (1.0 + 0.12 * (t.cyl_count < 0 ? 8 : t.cyl_count)); };
                return w(a) > w(c);
            });
            size_t q = 0;
            for (int k = lo + g; k < hi; k += 8) pl.tiles[k] = grp[q++];
        }
    }
    plan_pair_order(pl, pair_cus);
}

}  // namespace wv
