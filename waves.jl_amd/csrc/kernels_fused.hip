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

template <int AUX, int NW, int RPT, int RYMAX>
__device__ __forceinline__ void run_tile(const FusedParams &p, const TileDesc &t, F2 *raw, float e[3])
{
    const int tid = threadIdx.x;
    const FusedLds lds = lds_view(raw, NW * RPT, RYMAX);
    FusedRegs<AUX, RPT> r;
    // diagnostic stamps (p.stamps == nullptr in every normal run: one block-uniform branch per phase)
    unsigned long long *st = p.stamps ? p.stamps + (size_t)t.slot * 16 : nullptr;
#define WV_STAMP(k)                                                     \
    if (st && tid == 0) {                                               \
        __builtin_amdgcn_s_waitcnt(0);                                  \
        st[k] = __builtin_amdgcn_s_memtime();                           \
    }
    WV_STAMP(0)
    if (st && tid == 0) st[14] = __builtin_amdgcn_s_memrealtime();
    fused_load<AUX, NW, RPT>(p, t, tid, lds, r);
#define WV_STAGE(S)                                          \
    fused_publish<AUX, NW, RPT, S>(p, t, tid, lds, r);      \
    __syncthreads();                                         \
    WV_STAMP(2 * S - 1)                                      \
    fused_compute<AUX, NW, RPT, S>(p, t, tid, lds, r);      \
    if (S < 4) __syncthreads();                              \
    WV_STAMP(2 * S)
    WV_STAGE(1)
    WV_STAGE(2)
    WV_STAGE(3)
    WV_STAGE(4)
#undef WV_STAGE
    fused_store<AUX, NW, RPT>(p, t, tid, r, e);
    WV_STAMP(9)
    if (st && tid == 0) {
        st[10] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID, all 32 bits
        st[11] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        st[12] = __builtin_amdgcn_s_memrealtime();
        st[13] = (unsigned long long)(t.aux | (t.edge << 4));
    }
#undef WV_STAMP
}

// RF / RB / RP: rows per thread of the AUX_NONE / AUX_PX+AUX_PY / AUX_ALL tiles (a thread of a reduced field set
// carries less state per cell, so it owns more cells at the same register budget).
template <int NW, int RF, int RB, int RP>
__global__ __launch_bounds__(NW * 64, 4) void k_step_fused(FusedParams p)  // 4 waves/SIMD = 2 blocks per CU: <= 128 VGPRs
{
    constexpr int RMAX = RF > RB ? (RF > RP ? RF : RP) : (RB > RP ? RB : RP);
    constexpr int RYMAX = NW * RMAX;
    __shared__ F2 raw[lds_elems(RYMAX)];
    __shared__ float red[3][NW];
    const TileDesc t = p.tiles[blockIdx.x];
    float e[3];
    if (t.aux == AUX_NONE)
        run_tile<AUX_NONE, NW, RF, RYMAX>(p, t, raw, e);
    else if (t.aux == AUX_PX)
        run_tile<AUX_PX, NW, RB, RYMAX>(p, t, raw, e);
    else if (t.aux == AUX_PY)
        run_tile<AUX_PY, NW, RB, RYMAX>(p, t, raw, e);
    else
        run_tile<AUX_ALL, NW, RP, RYMAX>(p, t, raw, e);
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

// Does the state violate "Psi_x = 0 wherever sigma_x = 0, Psi_y = 0 wherever sigma_y = 0, Omega = 0 wherever
// sigma_x*sigma_y = 0" ?  (the precondition of the reduced field sets)
__global__ __launch_bounds__(256) void k_aux_check(const float *__restrict__ state, int nx, int ny, size_t P,
                                                   const float *__restrict__ sx, const float *__restrict__ sy,
                                                   int *__restrict__ flag)
{
    bool bad = false;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < P; q += (size_t)gridDim.x * 256) {
        const int i = (int)(q % nx), j = (int)(q / nx);
        const bool zx = sx[i] == 0.0f, zy = sy[j] == 0.0f;
        if (zx) bad = bad || state[3 * P + q] != 0.0f || state[9 * P + q] != 0.0f;
        if (zy) bad = bad || state[4 * P + q] != 0.0f || state[10 * P + q] != 0.0f;
        if (zx || zy) bad = bad || state[5 * P + q] != 0.0f || state[11 * P + q] != 0.0f;
    }
    if (bad) atomicOr(flag, 1);
}

// zero the planes a reduced tile would leave unwritten (makes a buffer a valid target of reduced tiles)
__global__ __launch_bounds__(256) void k_aux_clean(float *__restrict__ state, int nx, int ny, size_t P,
                                                   const float *__restrict__ sx, const float *__restrict__ sy)
{
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < P; q += (size_t)gridDim.x * 256) {
        const int i = (int)(q % nx), j = (int)(q / nx);
        const bool zx = sx[i] == 0.0f, zy = sy[j] == 0.0f;
        if (zx) state[3 * P + q] = state[9 * P + q] = 0.0f;
        if (zy) state[4 * P + q] = state[10 * P + q] = 0.0f;
        if (zx || zy) state[5 * P + q] = state[11 * P + q] = 0.0f;
    }
}

// flags[slot] = the source shape is non-zero somewhere in the tile's region (tiles without it skip the G loads and
// the "U .+ f" adds: U + (+-0) == U exactly)
__global__ __launch_bounds__(256) void k_src_flags(const TileDesc *__restrict__ tiles, const float *__restrict__ G,
                                                   int nx, int ny, unsigned char *__restrict__ flags)
{
    const TileDesc t = tiles[blockIdx.x];
    const int w = t.ox + 2 * FT_H, h = t.oy + 2 * FT_H;
    bool any = false;
    for (int q = threadIdx.x; q < w * h; q += 256) {
        const int gx = t.x0 - FT_H + q % w, gy = t.y0 - FT_H + q / w;
        if (gx >= 0 && gx < nx && gy >= 0 && gy < ny) any = any || G[(size_t)gy * nx + gx] != 0.0f;
    }
    __shared__ int s_any;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    if (any) atomicOr(&s_any, 1);
    __syncthreads();
    if (threadIdx.x == 0) flags[t.slot] = s_any ? 1 : 0;
}

}  // namespace

struct FusedPlan {
    Grid g{};
    int NW = 8, RF = 4, RB = 3, RP = 2;  // rows per thread: AUX_NONE / AUX_PX, AUX_PY / AUX_ALL tiles
    bool xcd_aware = true;
    unsigned char *d_src_flags = nullptr;
    size_t src_flags_cap = 0;
    bool src_dirty = true;        // the source shape (or the tiling) changed since the flags were computed
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
    if (const char *e = getenv("WAVES_AMD_FUSED_TILES")) {  // tuning knob: "RF,RB,RP", see fused_launch for the set
        int a = 0, b = 0, c = 0;
        if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3) {
            const int key = a * 100 + b * 10 + c;
            if (key == 432 || key == 332 || key == 322 || key == 222 || key == 333 || key == 422) {
                p->RF = a;
                p->RB = b;
                p->RP = c;
            }
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
    if (p->d_src_flags) (void)hipFree(p->d_src_flags);
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
    if (!plan_build_tiles(p->hp, p->g.nx, p->g.ny, p->NW * p->RF, p->NW * p->RB, p->NW * p->RP, p->x.data(),
                          p->y.data(), p->sx.data(), p->sy.data(), aux_zero, p->xcd_aware))
        return false;
    p->src_dirty = true;
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

void fused_source_changed(FusedPlan *p) { p->src_dirty = true; }

int fused_prepare(FusedPlan *p, float *frames, float *scratch0, float *scratch1, bool capture, const float *G,
                  const Cyl *d_table, const Cyl *h_table, int M, int rows, hipStream_t s)
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
    if (G && p->src_dirty) {
        if (nt > p->src_flags_cap) {
            if (p->d_src_flags) (void)hipFree(p->d_src_flags);
            p->d_src_flags = nullptr;
            p->src_flags_cap = 0;
            if (hipMalloc((void **)&p->d_src_flags, nt) != hipSuccess) return 1;
            p->src_flags_cap = nt;
        }
        hipLaunchKernelGGL(k_src_flags, dim3((unsigned)nt), dim3(256), 0, s, p->d_tiles, G, g.nx, g.ny, p->d_src_flags);
        p->src_dirty = false;
    }
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
    p.P = (unsigned)g.P;
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
    p.src_flags = st.G ? pl->d_src_flags : nullptr;
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
    const int key = pl->RF * 100 + pl->RB * 10 + pl->RP;
    switch (key) {
        case 332: hipLaunchKernelGGL((k_step_fused<8, 3, 3, 2>), grid, dim3(512), 0, s, p); break;
        case 322: hipLaunchKernelGGL((k_step_fused<8, 3, 2, 2>), grid, dim3(512), 0, s, p); break;
        case 222: hipLaunchKernelGGL((k_step_fused<8, 2, 2, 2>), grid, dim3(512), 0, s, p); break;
        case 333: hipLaunchKernelGGL((k_step_fused<8, 3, 3, 3>), grid, dim3(512), 0, s, p); break;
        case 422: hipLaunchKernelGGL((k_step_fused<8, 4, 2, 2>), grid, dim3(512), 0, s, p); break;
        default: hipLaunchKernelGGL((k_step_fused<8, 4, 3, 2>), grid, dim3(512), 0, s, p); break;
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
        fprintf(f, "%zu %d %d %d %d %d %d %d |", i, t.slot, t.x0, t.y0, t.ox, t.oy, t.aux | (t.edge << 4), t.cyl_count);
        for (int k = 0; k < 15; ++k) fprintf(f, " %llu", h[(size_t)t.slot * 16 + k]);
        fprintf(f, "\n");
    }
    fclose(f);
}

void fused_variant_counts(const FusedPlan *p, int out[4])
{
    for (int k = 0; k < 4; ++k) out[k] = p->hp.count[k];
}

}  // namespace wv
