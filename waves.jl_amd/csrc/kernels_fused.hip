// Fused Runge-Kutta step kernel: one launch advances the whole grid by one integration step (K1+K2+K3+K4+K5 of
// SURVEY 2.3 in one pass).  The per-tile algorithm lives in fused_body.h (shared with the CPU emulation harness); this
// file holds the __global__ wrapper, the block reduction of the energy terms and the host-side plan.
//
// Roofline: HBM-bound by contract (104 algorithmic bytes per cell-update).  Per launch the kernel reads the
// 12-field state once plus a 4-cell halo ring per tile (x 64/56 in x, x RY/(RY-8) in y) and writes it once; the wave
// speed is evaluated from <= a handful of culled cylinders per tile in registers, so no c-field is ever materialised.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "fused.h"
#include "fused_plan.h"

namespace wv {

namespace {

template <bool PML, bool EDGE, int NW, int RPT>
__device__ __forceinline__ void run_tile(const FusedParams &p, const TileDesc &t, F2 *raw, float e[3])
{
    const int tid = threadIdx.x;
    const FusedLds lds = lds_view(raw, NW * RPT);
    FusedRegs<PML, RPT> r;
    // diagnostic stamps (p.stamps == nullptr in every normal run: one block-uniform branch per phase)
    unsigned long long *st = p.stamps ? p.stamps + (size_t)t.slot * 16 : nullptr;
#define WV_STAMP(k)                                                     \
    if (st && tid == 0) {                                               \
        __builtin_amdgcn_s_waitcnt(0);                                  \
        st[k] = __builtin_amdgcn_s_memtime();                           \
    }
    WV_STAMP(0)
    fused_load<PML, EDGE, NW, RPT>(p, t, tid, r);
#define WV_STAGE(S)                                          \
    fused_publish<PML, EDGE, NW, RPT, S>(p, t, tid, lds, r); \
    __syncthreads();                                         \
    WV_STAMP(2 * S - 1)                                      \
    fused_compute<PML, EDGE, NW, RPT, S>(p, t, tid, lds, r); \
    if (S < 4) __syncthreads();                              \
    WV_STAMP(2 * S)
    WV_STAGE(1)
    WV_STAGE(2)
    WV_STAGE(3)
    WV_STAGE(4)
#undef WV_STAGE
    fused_store<PML, EDGE, NW, RPT>(p, t, tid, r, e);
    WV_STAMP(9)
    if (st && tid == 0) {
        st[10] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID, all 32 bits
        st[11] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        st[12] = __builtin_amdgcn_s_memrealtime();
        st[13] = (unsigned long long)t.variant;
    }
#undef WV_STAMP
}

// RF / RP: rows per thread of the 6-field FAST tiles / of the 12-field MID and GEN tiles (a FAST thread carries half
// the state per cell, so it can own twice the cells at the same register budget).
template <int NW, int RF, int RP>
__global__ __launch_bounds__(NW * 64) void k_step_fused(FusedParams p)
{
    constexpr int RYMAX = NW * (RF > RP ? RF : RP);
    __shared__ F2 raw[lds_elems(RYMAX)];
    __shared__ float red[3][NW];
    const TileDesc t = p.tiles[blockIdx.x];
    float e[3];
    if (t.variant == VAR_FAST)
        run_tile<false, false, NW, RF>(p, t, raw, e);
    else if (t.variant == VAR_MID)
        run_tile<true, false, NW, RP>(p, t, raw, e);
    else
        run_tile<true, true, NW, RP>(p, t, raw, e);
    if (p.epart) {  // block-uniform
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s = wave_sum(e[c]);
            if (lane == 0) red[c][w] = s;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < NW; ++k) s += red[threadIdx.x][k];
            p.epart[(size_t)t.slot * 3 + threadIdx.x] = s;
        }
    }
}

// any non-zero auxiliary field (Psi_x, Psi_y, Omega of either set) at a cell with sigma_x = sigma_y = 0 ?
__global__ __launch_bounds__(256) void k_aux_check(const float *__restrict__ state, int nx, int ny, size_t P,
                                                   const float *__restrict__ sx, const float *__restrict__ sy,
                                                   int *__restrict__ flag)
{
    bool bad = false;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < P; q += (size_t)gridDim.x * 256) {
        const int i = (int)(q % nx), j = (int)(q / nx);
        if (sx[i] == 0.0f && sy[j] == 0.0f) {
#pragma unroll
            for (int f : {3, 4, 5, 9, 10, 11}) bad = bad || (state[(size_t)f * P + q] != 0.0f);
        }
    }
    if (bad) atomicOr(flag, 1);
}

// zero the auxiliary planes at every cell with sigma_x = sigma_y = 0 (makes a buffer a valid FAST-tile target)
__global__ __launch_bounds__(256) void k_aux_clean(float *__restrict__ state, int nx, int ny, size_t P,
                                                   const float *__restrict__ sx, const float *__restrict__ sy)
{
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < P; q += (size_t)gridDim.x * 256) {
        const int i = (int)(q % nx), j = (int)(q / nx);
        if (sx[i] == 0.0f && sy[j] == 0.0f) {
#pragma unroll
            for (int f : {3, 4, 5, 9, 10, 11}) state[(size_t)f * P + q] = 0.0f;
        }
    }
}

}  // namespace

struct FusedPlan {
    Grid g{};
    int NW = 8, RF = 4, RP = 2;   // rows per thread: FAST tiles / MID+GEN tiles
    bool xcd_aware = true;
    std::vector<float> x, y, sx, sy;
    HostPlan hp;
    bool tiles_valid = false;
    int generation = 0;           // bumped whenever the tile decomposition changes
    bool tiles_aux_zero = false;  // the aux_zero value the current tile classification was built with
    int aux_state = 1;            // initial condition: 1 zero outside the PML, 0 not, -1 unknown
    bool scratch_clean = true;    // the two scratch states have zero auxiliary planes outside the PML
    bool frames_clean = true;     // ... and so have frames 0 and 1 (frame 2 is the initial condition itself)
    std::vector<int> idx;
    TileDesc *d_tiles = nullptr;
    size_t tiles_cap = 0;
    int *d_idx = nullptr;
    size_t idx_cap = 0;
    int *d_flag = nullptr;
    const Cyl *d_table = nullptr;
    int M = 0;
    unsigned long long *d_stamps = nullptr;  // diagnostic (WAVES_AMD_STAMPS=<file>)
    size_t stamps_cap = 0;
    const char *stamps_path = nullptr;
};

FusedPlan *fused_create(const Grid &g, const float *x, const float *y, const float *sx, const float *sy)
{
    FusedPlan *p = new (std::nothrow) FusedPlan();
    if (!p) return nullptr;
    p->g = g;
    p->x.assign(x, x + g.nx);
    p->y.assign(y, y + g.ny);
    p->sx.assign(sx, sx + g.nx);
    p->sy.assign(sy, sy + g.ny);
    if (const char *e = getenv("WAVES_AMD_FUSED_TILES")) {  // tuning knob: "RF,RP" in {4,2 | 3,2 | 3,3 | 2,2}
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && ((a == 4 && b == 2) || (a == 3 && b == 2) || (a == 3 && b == 3) || (a == 2 && b == 2))) {
            p->RF = a;
            p->RP = b;
        }
    }
    if (const char *e = getenv("WAVES_AMD_FUSED_XCD")) p->xcd_aware = atoi(e) != 0;
    p->stamps_path = getenv("WAVES_AMD_STAMPS");
    if (hipMalloc((void **)&p->d_flag, sizeof(int)) != hipSuccess) {
        delete p;
        return nullptr;
    }
    return p;
}

void fused_destroy(FusedPlan *p)
{
    if (!p) return;
    if (p->d_tiles) (void)hipFree(p->d_tiles);
    if (p->d_idx) (void)hipFree(p->d_idx);
    if (p->d_flag) (void)hipFree(p->d_flag);
    if (p->d_stamps) (void)hipFree(p->d_stamps);
    delete p;
}

void fused_set_pml(FusedPlan *p, const float *sx, const float *sy)
{
    p->sx.assign(sx, sx + p->g.nx);
    p->sy.assign(sy, sy + p->g.ny);
    p->tiles_valid = false;
    p->aux_state = -1;
    p->scratch_clean = false;
    p->frames_clean = false;
}

void fused_state_changed(FusedPlan *p)
{
    p->aux_state = -1;
    p->frames_clean = false;
}

void fused_state_zeroed(FusedPlan *p)
{
    p->aux_state = 1;
    p->frames_clean = true;
}

static bool ensure_tiles(FusedPlan *p, bool aux_zero)
{
    if (p->tiles_valid && p->tiles_aux_zero == aux_zero) return true;
    if (!plan_build_tiles(p->hp, p->g.nx, p->g.ny, p->NW * p->RF, p->NW * p->RP, p->x.data(), p->y.data(),
                          p->sx.data(), p->sy.data(), aux_zero, p->xcd_aware))
        return false;
    p->tiles_valid = true;
    p->tiles_aux_zero = aux_zero;
    p->generation++;
    return true;
}

int fused_energy_blocks(FusedPlan *p)
{
    if (!p->tiles_valid && !ensure_tiles(p, p->aux_state == 1)) return 0;
    return (int)p->hp.tiles.size();
}

int fused_prepare(FusedPlan *p, float *frames, float *scratch0, float *scratch1, bool capture, const Cyl *d_table,
                  const Cyl *h_table, int M, int rows, hipStream_t s)
{
    const Grid &g = p->g;
    const size_t N = g.P * kFields;
    float *ic = frames + 2 * N;
    if (p->aux_state < 0) {  // the caller replaced the state: look at it once
        int h = 0;
        if (hipMemsetAsync(p->d_flag, 0, sizeof(int), s) != hipSuccess) return 1;
        hipLaunchKernelGGL(k_aux_check, dim3(1024), dim3(256), 0, s, ic, g.nx, g.ny, g.P, g.sx, g.sy, p->d_flag);
        if (hipMemcpyAsync(&h, p->d_flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
        if (hipStreamSynchronize(s) != hipSuccess) return 1;
        p->aux_state = h ? 0 : 1;
    }
    const bool aux_zero = p->aux_state == 1;
    // FAST tiles write only 6 of the 12 planes: every buffer a step writes to must already hold zeros in the others
    if (aux_zero) {
        if (!p->scratch_clean) {
            for (float *b : {scratch0, scratch1})
                hipLaunchKernelGGL(k_aux_clean, dim3(1024), dim3(256), 0, s, b, g.nx, g.ny, g.P, g.sx, g.sy);
            p->scratch_clean = true;
        }
        if (capture && !p->frames_clean) {  // frames 0/1 are only written (and then fully replaced) when capturing
            for (float *b : {frames, frames + N})
                hipLaunchKernelGGL(k_aux_clean, dim3(1024), dim3(256), 0, s, b, g.nx, g.ny, g.P, g.sx, g.sy);
            p->frames_clean = true;
        }
    } else {  // this integrate writes non-zero auxiliaries outside the PML
        p->scratch_clean = false;
        if (capture) p->frames_clean = false;
    }
    if (!ensure_tiles(p, aux_zero)) return 2;
    plan_build_cyl(p->hp, p->x.data(), p->y.data(), h_table, M, rows, p->idx);
    const size_t nt = p->hp.tiles.size();
    if (nt > p->tiles_cap) {
        if (p->d_tiles) (void)hipFree(p->d_tiles);
        p->d_tiles = nullptr;
        p->tiles_cap = 0;
        if (hipMalloc((void **)&p->d_tiles, nt * sizeof(TileDesc)) != hipSuccess) return 1;
        p->tiles_cap = nt;
    }
    const size_t ni = p->idx.size() ? p->idx.size() : 1;
    if (ni > p->idx_cap) {
        if (p->d_idx) (void)hipFree(p->d_idx);
        p->d_idx = nullptr;
        p->idx_cap = 0;
        if (hipMalloc((void **)&p->d_idx, ni * sizeof(int)) != hipSuccess) return 1;
        p->idx_cap = ni;
    }
    if (hipMemcpyAsync(p->d_tiles, p->hp.tiles.data(), nt * sizeof(TileDesc), hipMemcpyHostToDevice, s) != hipSuccess) return 1;
    if (!p->idx.empty() &&
        hipMemcpyAsync(p->d_idx, p->idx.data(), p->idx.size() * sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess)
        return 1;
    // the host vectors are reused by the next prepare: make sure the copies are done with them
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    p->d_table = d_table;
    p->M = M;
    return 0;
}

void fused_launch(FusedPlan *pl, const FusedStep &st, hipStream_t s)
{
    const Grid &g = pl->g;
    FusedParams p{};
    p.nx = g.nx;
    p.ny = g.ny;
    p.P = g.P;
    p.ops = g.ops;
    p.x = g.x;
    p.y = g.y;
    p.sx = g.sx;
    p.sy = g.sy;
    p.c0 = g.c0;
    p.c0sq = g.c0sq;
    p.u = st.u;
    p.out = st.out;
    p.G = st.G;
    p.sfac[0] = st.sfac[0];
    p.sfac[1] = st.sfac[1];
    p.sfac[2] = st.sfac[2];
    p.cyl = pl->d_table + (size_t)st.table_row * pl->M;
    p.M = pl->M;
    p.dt = st.dt;
    p.hdt = 0.5f * st.dt;
    p.tiles = pl->d_tiles;
    p.cyl_idx = pl->d_idx;
    p.epart = st.epart;
    p.traj_tot = st.traj_tot;
    p.traj_inc = st.traj_inc;
    p.stamps = nullptr;
    if (pl->stamps_path) {
        const size_t need = pl->hp.tiles.size() * 16;
        if (need > pl->stamps_cap) {
            if (pl->d_stamps) (void)hipFree(pl->d_stamps);
            pl->d_stamps = nullptr;
            if (hipMalloc((void **)&pl->d_stamps, need * sizeof(unsigned long long)) == hipSuccess) pl->stamps_cap = need;
        }
        p.stamps = pl->d_stamps;
    }
    const dim3 grid((unsigned)pl->hp.tiles.size());
    const int key = pl->RF * 10 + pl->RP;
    switch (key) {
        case 32: hipLaunchKernelGGL((k_step_fused<8, 3, 2>), grid, dim3(512), 0, s, p); break;
        case 33: hipLaunchKernelGGL((k_step_fused<8, 3, 3>), grid, dim3(512), 0, s, p); break;
        case 22: hipLaunchKernelGGL((k_step_fused<8, 2, 2>), grid, dim3(512), 0, s, p); break;
        default: hipLaunchKernelGGL((k_step_fused<8, 4, 2>), grid, dim3(512), 0, s, p); break;
    }
}

int fused_generation(const FusedPlan *p) { return p->generation; }

// diagnostic: write the phase stamps of the LAST launched step, with the tile list, to WAVES_AMD_STAMPS (text)
void fused_dump_stamps(FusedPlan *p, hipStream_t s)
{
    if (!p->stamps_path || !p->d_stamps) return;
    const size_t nt = p->hp.tiles.size();
    std::vector<unsigned long long> h(nt * 16);
    if (hipStreamSynchronize(s) != hipSuccess) return;
    if (hipMemcpy(h.data(), p->d_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
    FILE *f = fopen(p->stamps_path, "w");
    if (!f) return;
    fprintf(f, "# pos slot x0 y0 ox oy variant cyl | t0..t9 xcc hwid realtime variant\n");
    for (size_t i = 0; i < nt; ++i) {
        const TileDesc &t = p->hp.tiles[i];
        fprintf(f, "%zu %d %d %d %d %d %d %d |", i, t.slot, t.x0, t.y0, t.ox, t.oy, t.variant, t.cyl_count);
        for (int k = 0; k < 14; ++k) fprintf(f, " %llu", h[(size_t)t.slot * 16 + k]);
        fprintf(f, "\n");
    }
    fclose(f);
}

void fused_variant_counts(const FusedPlan *p, int out[3])
{
    out[0] = p->hp.count[0];
    out[1] = p->hp.count[1];
    out[2] = p->hp.count[2];
}

}  // namespace wv
