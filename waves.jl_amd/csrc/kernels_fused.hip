// Fused Runge-Kutta step kernel: one launch advances the whole grid by one integration step (K1+K2+K3+K4+K5 of
// SURVEY 2.3 in one pass).  The per-tile algorithm lives in fused_body.h (shared with the CPU emulation harness); this
// file holds the __global__ wrapper, the block reduction of the energy terms and the host-side plan.
//
// Roofline: HBM-bound by contract (104 algorithmic bytes per cell-update).  Per launch the kernel reads the
// 12-field state once plus a 4-cell halo ring per tile (x 64/56 in x, x RY/(RY-8) in y) and writes it once; the wave
// speed is evaluated from <= a handful of culled cylinders per tile in registers, so no c-field is ever materialised.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <vector>

#include <hip/hip_ext.h>

#include "fused.h"
#include "fused_plan.h"

namespace wv {

namespace {

// The step loop must compile like a sequence of single-step bodies: only the tile's registers are carried from one
// step to the next.  Left alone, the compiler hoists every loop-invariant load, index and address out of the loop and
// keeps them live across it (hundreds of spills); so each iteration starts from values the optimiser cannot see through.
// (the pointers keep their address space -- constant: kernarg segment / tables nobody writes during the launch -- so
// that what is loaded through them stays scalar loads, and the pointers found there global rather than flat)
template <class T>
__device__ __forceinline__ const T *opaque(const T *p)
{
    auto q = (const __attribute__((address_space(4))) T *)p;
    asm volatile("" : "+s"(q));
    return (const T *)q;
}
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// a pointer / number that IS the same in every lane, said so to the compiler (it then lives in scalar registers)
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}

// The tile descriptor of this block, through scalar loads (the table is written by the host before the launch and by
// nobody during it; left to itself the compiler fetches it with vector loads -- a memory round trip in front of everything)
__device__ __forceinline__ TileDesc load_tile(const FusedParams &p)
{
    return opaque(p.tiles)[p.tile_offset + (int)blockIdx.x];
}

template <int AUX, int FL, int NW, int RPT, int RYMAX>
__device__ __forceinline__ void run_tile(const FusedParams &p, const TileDesc &t, F2 *raw, float e[3])
{
    const int tid = threadIdx.x;
    const FusedLds lds = lds_view(raw, NW * RPT, RYMAX);
    FusedRegs<AUX, RPT> r;
    TileCtx cx;
    // diagnostic stamps (p.stamps == nullptr in every normal run: one block-uniform branch per phase)
    unsigned long long *st = p.stamps ? p.stamps + (size_t)t.slot * 16 : nullptr;
#define WV_STAMP(k)                                                     \
    if (st && tid == 0) {                                               \
        __builtin_amdgcn_s_waitcnt(0);                                  \
        st[k] = __builtin_amdgcn_s_memtime();                           \
    }
    WV_STAMP(0)
    if (st && tid == 0) st[14] = __builtin_amdgcn_s_memrealtime();
    fused_load<AUX, FL, NW, RPT>(p, p.io, t, tid, lds, cx, r);
    fused_publish<AUX, FL, NW, RPT, 1>(p, t, tid, lds, cx, r);
    __syncthreads();
    fused_speed<AUX, FL, NW, RPT>(p, t, tid, lds, cx, r);
    WV_STAMP(1)
    // stage S: read buffer (S-1)&1 (+ DPP), update the registers, publish stage S+1 into buffer S&1; one barrier per
    // stage (boundary tiles whose side buffers cannot be double-buffered separate the two halves with a barrier).
#define WV_STAGE(S)                                                      \
    fused_compute<AUX, FL, NW, RPT, S>(p, t, tid, lds, cx, r, &r);       \
    if ((FL & F_EDGE) && !lds_side_double(NW * RPT, RYMAX)) __syncthreads();  \
    WV_STAMP(2 * S)                                                      \
    fused_publish<AUX, FL, NW, RPT, S + 1>(p, t, tid, lds, cx, r);       \
    __syncthreads();                                                     \
    WV_STAMP(2 * S + 1)
    WV_STAGE(1)
    WV_STAGE(2)
    WV_STAGE(3)
#undef WV_STAGE
    fused_compute<AUX, FL, NW, RPT, 4>(p, t, tid, lds, cx, r, &r);
    WV_STAMP(8)
    fused_store<AUX, NW, RPT>(p, p.io, t, tid, r, e);
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
__global__ __launch_bounds__(NW * 64, 4) void k_step_fused(FusedParams p_)  // 4 waves/SIMD = 2 blocks per CU: <= 128 VGPRs
{
    // The arguments are read where they are used, straight from the kernarg segment (uniform s_loads of invariant
    // memory), instead of being loaded once at entry: ~50 SGPRs live across the whole variant dispatch cost the
    // variants SGPR spills and made their codegen depend on each other.
#if defined(__HIP_DEVICE_COMPILE__)
    (void)p_;
    const FusedParams &p = *(const FusedParams *)(const void *)__builtin_amdgcn_kernarg_segment_ptr();
#else
    const FusedParams &p = p_;
#endif
    constexpr int RMAX = RF > RB ? (RF > RP ? RF : RP) : (RB > RP ? RB : RP);
    constexpr int RYMAX = NW * RMAX;
    __shared__ F2 raw[lds_elems(RYMAX)];
    __shared__ float red[3][NW];
    const TileDesc t = load_tile(p);
    float e[3];
    // smallest instantiated superset of the features this tile needs (block-uniform dispatch): boundary tiles are
    // specialised per side (left/right strips, bottom/top bands), corners get all four sides
#define RUN(A, F, R) run_tile<A, F, NW, R, RYMAX>(p, t, raw, e)
    const int fl = tile_flags(p, t);
    const int fe = fl & F_EDGE;
    const bool cyl = (fl & F_CYL) != 0;
    if (t.aux == AUX_NONE) {  // never a boundary tile (fused_plan.h)
        if (fl == 0) RUN(AUX_NONE, 0, RF);
        else if (fl == F_SRC) RUN(AUX_NONE, F_SRC, RF);
        else RUN(AUX_NONE, F_CYL | F_SRC, RF);
    } else if (t.aux == AUX_PX) {
        const bool src = (fl & F_SRC) != 0;  // (the source rarely reaches the PML: its own variants without it)
        if (!cyl && !src && (fe & ~F_EL) == 0) RUN(AUX_PX, F_EL, RB);  // left strip, or a PML strip off the boundary
        else if (!cyl && !src && (fe & ~F_ER) == 0) RUN(AUX_PX, F_ER, RB);
        else if (!cyl && (fe & ~F_EL) == 0) RUN(AUX_PX, F_EL | F_SRC, RB);
        else if (!cyl && (fe & ~F_ER) == 0) RUN(AUX_PX, F_ER | F_SRC, RB);
        else RUN(AUX_PX, F_ALL, RB);
    } else if (t.aux == AUX_PY) {
        if (fl == 0) RUN(AUX_PY, 0, RB);
        else if (!cyl && (fl & F_SRC) == 0 && (fe & ~F_ET) == 0) RUN(AUX_PY, F_ET, RB);
        else if (!cyl && (fl & F_SRC) == 0 && (fe & ~F_EB) == 0) RUN(AUX_PY, F_EB, RB);
        else if (!cyl && (fe & ~F_ET) == 0) RUN(AUX_PY, F_ET | F_SRC, RB);
        else if (!cyl && (fe & ~F_EB) == 0) RUN(AUX_PY, F_EB | F_SRC, RB);
        else RUN(AUX_PY, F_ALL, RB);
    } else {
        if (!cyl && (fl & F_SRC) == 0) RUN(AUX_ALL, F_EDGE, RP);
        else if (!cyl) RUN(AUX_ALL, F_EDGE | F_SRC, RP);
        else RUN(AUX_ALL, F_ALL, RP);
    }
#undef RUN
    if (p.io.epart) {  // block-uniform
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
            p.io.epart[(size_t)t.slot * 3 + threadIdx.x] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_steps_resident: ALL steps of one wv_integrate call in one launch.  Every tile is resident for the whole call (the
// launch is cooperative: grid <= the number of blocks the device holds at once) and keeps its own output cells in
// registers from one step to the next.  Per step it still writes its outputs to the state buffer (memory holds the
// complete state of every step: frames, trajectories and the final state are exactly what k_step_fused leaves), but it
// reads nothing back from there: the halo ring comes from the neighbouring tiles through the tagged exchange buffer
// (fused_body.h, fused_xch_*).  There is no grid-wide barrier and no flag: a tile starts step s+1 as soon as the border
// words of step s of its neighbours have arrived, so tiles drift apart by up to one step per tile of distance and the
// memory latencies of some overlap the arithmetic of others on the same CU -- which a sequence of single-step launches,
// all tiles in lock-step, cannot do.
// A wave polls its halo words at most WV_WAIT_POLLS times (seconds; a healthy wait takes microseconds), then raises
// *abort; every wave watches that word while it polls and the block leaves together: the kernel always drains and the
// host reports the failure.
constexpr int WV_WAIT_POLLS = 1 << 20;  // default of FusedParams::max_polls (WAVES_AMD_WAIT_POLLS overrides: tests)

// barrier that orders LDS accesses only (__syncthreads also waits for every outstanding global access)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One JOB of one tile.  Returns false when the job was abandoned (a wave of this or of another block gave up waiting: the
// abort word is set and every block leaves); the job's initial condition is never written (its final state has a buffer of
// its own).
// NOT inlined into the job loop of the kernel (WV_TILE_CALL): as part of that loop's body the very same instructions of a
// step ran 3.5 % slower (the step loop then is a loop inside a loop with seventeen bodies: layout and allocation are shaped
// around all of them); as a function of its own each variant is laid out and allocated like the one-job kernel of round 2.
// The LDS arrays travel as address-space-3 pointers so that their accesses stay ds_ instructions, the job description as a
// pointer that is made uniform again on this side of the call (arguments travel in vector registers).
#ifndef WV_TILE_INLINE
#define WV_TILE_CALL __attribute__((noinline))
#else
#define WV_TILE_CALL __forceinline__
#endif
typedef __attribute__((address_space(3))) F2 *lds_f2_ptr;
typedef __attribute__((address_space(3))) float *lds_f_ptr;
typedef __attribute__((address_space(3))) int *lds_i_ptr;
template <int AUX, int FL, int NW, int RPT, int RYMAX>
__device__ WV_TILE_CALL bool run_tile_resident(const FusedParams *p0_, lds_f2_ptr raw_, lds_f_ptr red_, lds_i_ptr vote_)
{
    const FusedParams *p0 = uniform_ptr(p0_);
    F2 *raw = (F2 *)raw_;
    float(*red)[NW] = (float(*)[NW])(float *)red_;
    int *vote = (int *)vote_;
    const int *cull = vote + 1;  // the tile's cylinder list as the kernel found it (FusedParams::dev_cull): [0] count, [1 ...] indices
    const FusedLds lds = lds_view(raw, NW * RPT, RYMAX);
    FusedRegs<AUX, RPT> r;
    TileCtx cx;
#if defined(WV_PRIO_SECOND)  // (experiment) the block that shares its CU with an older one loses every arbitration: raise it
    if ((int)blockIdx.x >= WV_PRIO_SECOND) __builtin_amdgcn_s_setprio(2);
#endif
#if defined(WV_PRIO_NONE)    // (experiment) interior tiles above their PML partners
    if (AUX == AUX_NONE) __builtin_amdgcn_s_setprio(2);
#endif
    // the tile descriptor once, in scalar registers: a scalar load chain at the top of every step costs 1.3 %
    TileDesc t = load_tile(*opaque(p0));
    const bool dev_cull = opaque(p0)->dev_cull != 0;
    if (dev_cull) t.cyl_count = __builtin_amdgcn_readfirstlane(cull[0]);
#ifndef WV_TID_SPILLED  // (the first version, kept for A/B)
    // The thread index of a step is put together from the wave's index (a scalar) and the lane number: threadIdx.x itself,
    // live around the step loop, was kept in scratch and reloaded -- with a full wait -- at the top of every step.
    const int wave_base = __builtin_amdgcn_readfirstlane((int)threadIdx.x & ~63);
#define WV_TID() (wave_base | (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
#else
#define WV_TID() opaque((int)threadIdx.x)
#endif
    {
        const FusedParams &p = *opaque(p0);
        const int tid = opaque((int)threadIdx.x);
        fused_tile_init<AUX, FL, NW, RPT>(p, t, tid, cx, r, dev_cull ? cull : nullptr);
        fused_cyl_times<FL>(p, 0, cx);
        fused_cyl_commit<FL>(t, tid, lds, cx, fused_cyl_fetch<AUX, FL, RPT>(p, 0, t, tid, cx, r));  // steps[s].step == s
        fused_load_state<AUX, NW, RPT>(p, opaque(p.steps)[0].u, t, tid, r);
        if (tid == 0) *vote = 0;  // (set by a wave that gives up; read after the first barrier of a step)
        if (p.back && blockIdx.x == 0 && tid == 0) {  // diagnostic
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            job_st_sys64(&p.back->phase[p.seq & 1u][1], job_clock());
        }
    }
    for (int s = 0;; ++s) {
        const FusedParams &p = *opaque(p0);
        const int tid = opaque(WV_TID());
        const StepIO &io = opaque(p.steps)[s];
        // diagnostic stamps of ONE step (the middle one); p.stamps == nullptr in every normal run
        unsigned long long *st = (p.stamps && s == p.nsteps / 2) ? p.stamps + (size_t)t.slot * 16 : nullptr;
#define WV_STAMP(k)                                                     \
    if (st && tid == 0) {                                               \
        __builtin_amdgcn_s_waitcnt(0);                                  \
        st[k] = __builtin_amdgcn_s_memrealtime();                       \
    }
        WV_STAMP(0)
        fused_step_init<FL>(p, io.step, cx);
        fused_cyl_times<FL>(p, io.step + 1, cx);  // (for the fetch of the NEXT step's cylinders at the end of this one)
        fused_publish<AUX, FL, NW, RPT, 1>(p, t, tid, lds, cx, r);
        __syncthreads();
        WV_STAMP(7)
        // This barrier is also the one that closes the previous step: its energy terms are complete in `red`, and a
        // wave that gave up waiting for its halo has said so.  (There is no barrier of its own between two steps: a wave
        // whose halo has arrived starts publishing while the others still poll -- stage buffer 0 was last read in stage
        // 3 -- and the cylinder copy in LDS was last read right after this barrier of the previous step.)
        if (s > 0) {
            float *ep = opaque(p.steps)[s - 1].epart;
            if (ep && tid < 3) {
                float v = 0.0f;
#pragma unroll
                for (int k = 0; k < NW; ++k) v += red[tid][k];
#ifdef WV_PLAIN_EPART  // (timing experiment only)
                ep[(size_t)t.slot * 3 + tid] = v;
#else
                job_st_agentf(ep + (size_t)t.slot * 3 + tid, v);  // (write-through: the rows are summed by other blocks of this launch)
#endif
            }
            if (*vote != 0) return false;  // block-uniform: a block that gives up leaves together
        }
        fused_speed<AUX, FL, NW, RPT>(p, t, tid, lds, cx, r);
        WV_STAMP(8)
#define WV_STAGE(S)                                                      \
    fused_compute<AUX, FL, NW, RPT, S>(p, t, tid, lds, cx, r, &r);       \
    if ((FL & F_EDGE) && !lds_side_double(NW * RPT, RYMAX)) __syncthreads();  \
    fused_publish<AUX, FL, NW, RPT, S + 1>(p, t, tid, lds, cx, r);       \
    __syncthreads();
        WV_STAGE(1)
        WV_STAMP(9)
        WV_STAGE(2)
        WV_STAGE(3)
        WV_STAMP(11)
#undef WV_STAGE
        fused_compute<AUX, FL, NW, RPT, 4>(p, t, tid, lds, cx, r, &r);
        WV_STAMP(1)
#ifdef WV_SLEEP_CLASS  // (sensitivity experiment: ~1 us of extra time per step in the tiles of one field set)
        if (AUX == WV_SLEEP_CLASS)
            __builtin_amdgcn_s_sleep(WV_SLEEP_LEN);
#endif
        // The last step ends behind the loop (what happens there -- the job's final stores, its stamps -- is then not part of the
        // loop the register allocator and the scheduler shape around: with such code inside, the steps themselves were 2 % slower).
        if (s + 1 == p.nsteps) break;
        // the border first: the neighbours are waiting for it
        const unsigned tag = p.tag_base + (unsigned)(s + 1);
        fused_xch_store<AUX, NW, RPT>(p, tag, t, tid, r);
        float e[3];
        fused_store<AUX, NW, RPT>(p, io, t, tid, r, e);
        fused_end_step<AUX, RPT>(r);
        WV_STAMP(2)
#ifndef WV_NO_ENERGY  // (timing experiment only)
        if (io.epart) {  // block-uniform
            const int lane = tid & 63, w = wv_wave_of(tid);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = wave_sum(e[c]);
                if (lane == 0) red[c][w] = v;
            }
        }
#endif
        WV_STAMP(3)
        // halo of the next step: every wave polls the words of its own rows
        // (the next step's cylinders travel while the halo is awaited)
        const Cyl nc = fused_cyl_fetch<AUX, FL, RPT>(p, io.step + 1, t, tid, cx, r);
        bool ok = false;
        int polls = 0;
#ifdef WV_XCH_PROBE  // (measured: the extra round trip in front of the full read costs 10 %)
        // spin on one granule per lane first (bounded: the full poll below is what counts, and it watches the abort word)
        for (int k = 0; k < 64; ++k) {
            if (__all(fused_xch_probe<AUX, NW, RPT>(p, tag, t, tid))) break;
            __builtin_amdgcn_s_sleep(1);
        }
#endif
        for (; polls < p.max_polls; ++polls) {
            ok = __all(fused_xch_load<AUX, NW, RPT>(p, tag, t, tid, r));
            if (ok) break;
            const int ab = ((tid & 63) == 0) ? __hip_atomic_load(p.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            if (__any(ab != 0)) break;
#ifdef WV_POLL_BACKOFF
            __builtin_amdgcn_s_sleep(WV_POLL_BACKOFF);
#else
            __builtin_amdgcn_s_sleep(1);
#endif
        }
        if (!ok && (tid & 63) == 0) {
            __hip_atomic_store(p.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *vote = 1;
        }
        fused_cyl_commit<FL>(t, tid, lds, cx, nc);
#ifdef WV_XCH_VERIFY  // (diagnostic: a second read of the halo must return the very same values)
        {
            FusedRegs<AUX, RPT> r2 = r;
            (void)fused_xch_load<AUX, NW, RPT>(p, tag, t, tid, r2);
            int diff = 0;
#pragma unroll
            for (int rr = 0; rr < RPT; ++rr)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < aux_ns(AUX); ++j)
                        diff += __builtin_bit_cast(unsigned, r2.u[rr][q][j]) != __builtin_bit_cast(unsigned, r.u[rr][q][j]) ? 1 : 0;
            if (ok && diff) atomicAdd(p.abort + 1, diff);
        }
#endif
        fused_end_poll<AUX, RPT>(r);
        // (nothing of the next step is scheduled into the poll: it would stretch the live ranges of the exchange words
        // over the publish phase and spill)
        asm volatile("s_nop 0" ::: "memory");
        WV_STAMP(4)
        if (st && tid == 0) {
            st[5] = st[4];
            st[6] = (unsigned long long)polls;
            st[10] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
            st[13] = (unsigned long long)(t.aux | (t.edge << 4));
        }
#undef WV_STAMP
    }
    // ---- the end of the last step: stage 4 has run, the new state is in r.y
    {
        const FusedParams &p = *opaque(p0);
        const int tid = opaque(WV_TID());
        const StepIO &io = opaque(p.steps)[p.nsteps - 1];
        if (p.back && blockIdx.x == 0 && tid == 0) job_st_sys64(&p.back->phase[p.seq & 1u][2], job_clock());  // diagnostic
        // (No barrier in front of this store: the final state goes to a buffer of its own -- wv_ctx::cur2 --, never over the
        // state the job started from.  A job that is given up therefore leaves its initial condition intact, and the host
        // runs the same call again with the single-step kernels.)
        if (p.back && blockIdx.x == 0 && tid == 0) job_st_sys64(&p.back->phase[p.seq & 1u][3], job_clock());  // diagnostic
        float e[3];
        fused_store<AUX, NW, RPT>(p, io, t, tid, r, e);
#ifndef WV_NO_ENERGY  // (timing experiment only)
        if (io.epart) {  // block-uniform
            const int lane = tid & 63, w = wv_wave_of(tid);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = wave_sum(e[c]);
                if (lane == 0) red[c][w] = v;
            }
        }
#endif
        lds_barrier();
        if (io.epart && tid < 3) {
            float v = 0.0f;
#pragma unroll
            for (int k = 0; k < NW; ++k) v += red[tid][k];
            job_st_agentf(io.epart + (size_t)t.slot * 3 + tid, v);
        }
    }
#undef WV_TID
    return true;
}

// Second pass of the energy sums of one saved time, by one block: exactly k_energy_final's arithmetic (256 threads, strided
// double sums in a fixed order, a pairwise tree in LDS, one rounding, then * dOmega in fp32), so that the trace does not depend
// on which kernel produced it.  The partial sums were written write-through by all blocks (barrier B lies in between).
__device__ __forceinline__ void job_energy_row(const FusedParams &p, int row, double *red /* LDS [3][256] */)
{
    const int tid = (int)threadIdx.x;
    const float *src = row == 0 ? p.ef_row0 : p.ef_epart + (size_t)row * (size_t)p.ntiles * 3;
    if (tid < 256) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        for (int b = tid; b < p.ntiles; b += 256) {
            s0 += (double)job_ld_agentf(src + b * 3 + 0);
            s1 += (double)job_ld_agentf(src + b * 3 + 1);
            s2 += (double)job_ld_agentf(src + b * 3 + 2);
        }
        red[tid] = s0;
        red[256 + tid] = s1;
        red[512 + tid] = s2;
    }
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
#pragma unroll
            for (int c = 0; c < 3; ++c) red[c * 256 + tid] += red[c * 256 + tid + off];
        }
        __syncthreads();
    }
    if (tid < 3) job_st_sysf(p.ef_signal + (size_t)row * 3 + tid, (float)red[tid * 256] * p.ef_dOmega);
    __syncthreads();  // (red is reused by the block's next row)
}

// This block's share of state(env) (src/env.jl:132-137) of the frames the job leaves: k_observation's arithmetic (obs_pixel),
// read past this CU's L1 (the frames were written by other blocks of this launch), written straight to pinned host memory.
__device__ __forceinline__ void job_observation(const FusedParams &p, int rb, int nob)
{
    const int rx = p.ob_rx, ry = p.ob_ry;
    const size_t total = (size_t)rx * ry * 4;
    for (size_t q = (size_t)rb * 512 + threadIdx.x; q < total; q += (size_t)nob * 512) {
        const int i = (int)(q % rx), j = (int)((q / rx) % ry), ch = (int)(q / ((size_t)rx * ry));
        const float *src = ch == 0 ? p.ob_f0 : (ch == 1 ? p.ob_f1 : (ch == 2 ? p.ob_f2 : p.ob_G));
        const float v = src ? obs_pixel([src](size_t k) { return job_ld_agentf(src + k); }, p.nx, p.ny, rx, ry, i, j) : 0.0f;
        job_st_sysf(p.ob_out + q, v);
    }
}

template <int NW, int RF, int RB, int RP>
__global__ __launch_bounds__(NW * 64, 4) void k_steps_resident(JobArgs a_)
{
    // The arguments are read where they are used, straight from the kernarg segment, through pointers the optimiser cannot
    // see through: whatever is loop-invariant here (mailbox addresses, flag arrays, ...) would otherwise be hoisted out of the
    // job loop and kept in scalar registers across the tile bodies, which have none to spare.
#if defined(__HIP_DEVICE_COMPILE__)
    (void)a_;
    const JobArgs *const a0 = (const JobArgs *)(const void *)__builtin_amdgcn_kernarg_segment_ptr();
#else
    const JobArgs *const a0 = &a_;
#endif
    constexpr int RMAX = RF > RB ? (RF > RP ? RF : RP) : (RB > RP ? RB : RP);
    constexpr int RYMAX = NW * RMAX;
    __shared__ F2 raw[lds_elems(RYMAX)];
    __shared__ float red[3][NW];
    __shared__ int vote[2 + FT_MAXCYL];  // [0]: a wave of the block gave up waiting for its halo; [1], [2 ...]: the tile's cylinder
                                         // list when the tiles cull themselves (FusedParams::dev_cull): count, indices
    __shared__ unsigned s_job[4];  // the job's command and number (kept here, not in registers, across the tile body); leader
                                   // block: number and command of a job it has fetched ahead (0: none)
    static_assert(sizeof(double) * 3 * 256 <= sizeof(F2) * lds_elems(RYMAX), "the energy rows are summed in the tile's LDS image");
    if (threadIdx.x == 0) {
        s_job[1] = opaque(a0)->first_seq;
        s_job[2] = 0;
    }
    __syncthreads();
    for (;;) {
        // ---- the job: block 0 asks the host, everybody else asks block 0
        if (threadIdx.x < 64) {
            const JobArgs &a = *opaque(a0);
            const unsigned seq = __builtin_amdgcn_readfirstlane(s_job[1]);
            int cmd;
            if (blockIdx.x == 0) {
                cmd = s_job[2] == seq ? (int)s_job[3] : job_leader_fetch(a, seq, (int)threadIdx.x);  // (fetched ahead: see below)
                if (cmd == JOB_RUN && threadIdx.x == 0) job_st_sys64(&a.back->t_begin[seq & 1u], job_clock());
            } else {
                cmd = job_wait_go(a, seq);
            }
            // what the job reads was written while this launch was running (the leader's copy of the description, the host's
            // uploads, other blocks' final state of the previous job): nothing of it may come from this CU's vector L1 or
            // from the scalar cache
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __builtin_amdgcn_s_dcache_inv();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) s_job[0] = (unsigned)cmd;
        }
        __syncthreads();
        if (s_job[0] != (unsigned)JOB_RUN) return;
        if (blockIdx.x == 0 && threadIdx.x == 0) {  // diagnostic
            const JobArgs &a = *opaque(a0);
            const unsigned sq = s_job[1];
            if (a.ctl->jobs[sq & 1u].p.back) job_st_sys64(&a.back->phase[sq & 1u][0], job_clock());
        }
        bool ok = true;
        {
            const FusedParams *pj = uniform_ptr(&opaque(a0)->ctl->jobs[__builtin_amdgcn_readfirstlane(s_job[1]) & 1u].p);
            const FusedParams &p = *opaque(pj);
            TileDesc t = load_tile(p);
            if (p.dev_cull) {  // (block-uniform) which cylinders can reach this tile during the call: one lane per cylinder
                if (threadIdx.x < 64) {
                    const int m = (int)threadIdx.x;
                    const bool keep = m < p.M && device_cull_keep(p, t, m);
                    const unsigned long long mask = __ballot(keep);
                    if (keep) vote[2 + __popcll(mask & ((1ull << m) - 1ull))] = m;  // ascending, like the host's lists
                    if (m == 0) vote[1] = __popcll(mask);
                }
                __syncthreads();
                t.cyl_count = __builtin_amdgcn_readfirstlane(vote[1]);
            }
#define RUN(A, F, R) ok = run_tile_resident<A, F, NW, R, RYMAX>(pj, (lds_f2_ptr)raw, (lds_f_ptr)&red[0][0], (lds_i_ptr)vote)
            const int fl = tile_flags(p, t);
            const int fe = fl & F_EDGE;
            const bool cyl = (fl & F_CYL) != 0;
            if (t.aux == AUX_NONE) {
                if (fl == 0) RUN(AUX_NONE, 0, RF);
                else if (fl == F_SRC) RUN(AUX_NONE, F_SRC, RF);
                else RUN(AUX_NONE, F_CYL | F_SRC, RF);
            } else if (t.aux == AUX_PX) {
                const bool src = (fl & F_SRC) != 0;
                if (!cyl && !src && (fe & ~F_EL) == 0) RUN(AUX_PX, F_EL, RB);
                else if (!cyl && !src && (fe & ~F_ER) == 0) RUN(AUX_PX, F_ER, RB);
                else if (!cyl && (fe & ~F_EL) == 0) RUN(AUX_PX, F_EL | F_SRC, RB);
                else if (!cyl && (fe & ~F_ER) == 0) RUN(AUX_PX, F_ER | F_SRC, RB);
                else RUN(AUX_PX, F_ALL, RB);
            } else if (t.aux == AUX_PY) {
                if (fl == 0) RUN(AUX_PY, 0, RB);
                else if (!cyl && (fl & F_SRC) == 0 && (fe & ~F_ET) == 0) RUN(AUX_PY, F_ET, RB);
                else if (!cyl && (fl & F_SRC) == 0 && (fe & ~F_EB) == 0) RUN(AUX_PY, F_EB, RB);
                else if (!cyl && (fe & ~F_ET) == 0) RUN(AUX_PY, F_ET | F_SRC, RB);
                else if (!cyl && (fe & ~F_EB) == 0) RUN(AUX_PY, F_EB | F_SRC, RB);
                else RUN(AUX_PY, F_ALL, RB);
            } else {
                if (!cyl && (fl & F_SRC) == 0) RUN(AUX_ALL, F_EDGE, RP);
                else if (!cyl) RUN(AUX_ALL, F_EDGE | F_SRC, RP);
                else RUN(AUX_ALL, F_ALL, RP);
            }
#undef RUN
        }
        // ---- the end of the job (everything re-read: nothing is carried in registers across the tile body)
        const JobArgs &a = *opaque(a0);
        const unsigned seq = __builtin_amdgcn_readfirstlane(s_job[1]);
        const FusedParams &p = *opaque(uniform_ptr(&a.ctl->jobs[seq & 1u].p));
        const int b = (int)blockIdx.x;
        if (!ok) {  // abandoned (block-uniform): tell the host why the launch is gone
            if (threadIdx.x == 0) {
                job_st_sys(&a.back->exit_seq[a.launch], seq);
                job_st_sys(&a.back->status[a.launch], JOBS_EXIT_ABORT);
            }
            return;
        }
        // barrier B: every store of the job (final state, frames, trajectories, partial sums) of every block has left
        job_drain();
        __syncthreads();
        if (p.back && b == 0 && threadIdx.x == 0) job_st_sys64(&p.back->phase[seq & 1u][4], job_clock());  // diagnostic
        // The leader block looks for the NEXT job while its first wave waits at barrier B (the second wave has nothing else to
        // do): the description's trip over PCIe (3 round trips, ~5 us) then lies under this job's end instead of between two jobs.
        if (b == 0 && threadIdx.x >= 64 && threadIdx.x < 128 && !p.last) {
            const int cmd = job_try_fetch(a, seq + 1u, (int)threadIdx.x - 64);
            if (threadIdx.x == 64) {
                s_job[2] = cmd != 0 ? seq + 1u : 0u;
                s_job[3] = (unsigned)cmd;
            }
        }
        if (threadIdx.x < 64) {
            const bool okb = job_barrier(job_flags(a.ctl, 1, p.ntiles), p.ntiles, b, seq, p.max_polls, p.abort, (int)threadIdx.x);
            if (threadIdx.x == 0) vote[0] = okb ? 0 : 1;
        }
        __syncthreads();
        if (vote[0] != 0) {  // (cannot happen short of a fault: a block that got here has nothing left to wait for but the others' last stores)
            if (threadIdx.x == 0) {
                __hip_atomic_store(p.abort, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                job_st_sys(&a.back->exit_seq[a.launch], seq);
                job_st_sys(&a.back->status[a.launch], JOBS_EXIT_ABORT);
            }
            return;
        }
        if (p.back && b == 0 && threadIdx.x == 0) job_st_sys64(&p.back->phase[seq & 1u][5], job_clock());  // diagnostic
        // the energy trace: the blocks sum the rows straight into pinned host memory -- counted from the END of the launch order
        // (block ntiles - 1 takes rows 0, ntiles, ...): the last positions hold the lightest tiles (plan_pair_order), which reach
        // this point first and are not the ones their neighbours wait for at the next call's first exchange
        const int nrows = p.ef_signal ? p.nsteps + 1 : 0;
        const int rb = p.ntiles - 1 - b;
        for (int row = rb; row < nrows; row += p.ntiles) job_energy_row(p, row, reinterpret_cast<double *>(raw));
        // state(env) of the frames this job leaves, when asked for: the same blocks, one element per thread and pass
        const int nob = p.ob_out ? (p.ntiles < JOB_OBS_BLOCKS ? p.ntiles : JOB_OBS_BLOCKS) : 0;
        if (rb < nob) job_observation(p, rb, nob);
        if (rb < nrows || rb < nob) {  // my rows are in host memory (the host looks at these words, nobody on the device waits for them)
            job_drain();
            __syncthreads();
            if (threadIdx.x == 0) job_st_sys(&a.back->rowdone[rb], seq);
        }
        // the leader reports: state, frames and trajectories of the job are complete (barrier B)
        if (b == 0 && threadIdx.x == 0) {
            job_st_sys64(&a.back->t_end[seq & 1u], job_clock());
            if (p.last) {
                job_st_sys(&a.back->exit_seq[a.launch], seq);
                job_st_sys(&a.back->status[a.launch], JOBS_EXIT_LAST);
            }
            job_drain();
            job_st_sys(&a.back->done, seq);
            job_drain();
        }
        if (p.last) return;
#ifdef WV_ONE_JOB  // (timing experiment only: no loop around the tile bodies)
        return;
#endif
        if (threadIdx.x == 0) s_job[1] = seq + 1;
        __syncthreads();
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

struct FusedPlan;
// Plans that may have a launch waiting on the device.  A process that ends without destroying its contexts must not pull the
// mailbox (pinned host memory the leader block polls) from under a running kernel: at exit every launch that stays is told to
// leave -- with plain stores, no HIP call (the runtime may already be shutting down) -- and given a moment to say that it has.
namespace {
struct LiveLaunches {
    std::mutex mu;
    std::vector<FusedPlan *> plans;
    ~LiveLaunches();
};
LiveLaunches g_live;
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
    bool frames_clean = true;     // ... and so have frames 0 and 1 ...
    bool f2_clean[2] = {true, true};  // ... and the two buffers the last frame alternates between
    std::vector<int> idx;
    // per-call tables, one set per slot (two calls may be in flight): device copies and pinned staging
    int cur = 0;                  // slot of the call being prepared / launched
    TileDesc *d_tiles[2] = {nullptr, nullptr};
    TileDesc *h_tiles[2] = {nullptr, nullptr};
    // the launch order of a job WITHOUT host-built tables (fused_try_resident, dev): pinned host memory the blocks read their
    // tile from directly, one buffer per job parity
    TileDesc *h_order[2] = {nullptr, nullptr};
    size_t order_cap[2] = {0, 0};
    std::vector<Cyl> order_ends;             // the call's design at its earliest and latest stage time, [2][M]
    size_t tiles_cap[2] = {0, 0};
    int *d_idx[2] = {nullptr, nullptr};
    int *h_idx[2] = {nullptr, nullptr};
    size_t idx_cap[2] = {0, 0};
    hipEvent_t up_ev[2] = {nullptr, nullptr};
    int *d_flag = nullptr;
    const Cyl *d_table = nullptr;
    int M = 0;
    int nbands = 1;               // tile bands per step in graph mode (WAVES_AMD_FUSED_BANDS)
    bool use_graph = true;        // WAVES_AMD_FUSED_GRAPH=0 disables
    hipGraph_t graph[2] = {nullptr, nullptr};
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};
    std::vector<char> graph_key[2];  // everything the cached graph's kernel arguments were built from
    unsigned long long *d_stamps = nullptr;  // diagnostic (WAVES_AMD_STAMPS=<file>)
    size_t stamps_cap = 0;
    const char *stamps_path = nullptr;
    // k_steps_resident (all steps of a call in one cooperative launch)
    bool use_resident = true;     // WAVES_AMD_FUSED_RESIDENT=0 disables
    bool granules_checked = false;  // granules_ok() has been asked on behalf of this plan
    bool allow_resident = true;   // set per call by the owner (false: other contexts share the device)
    int max_polls = WV_WAIT_POLLS;
    int resident_capacity = -1;   // blocks of k_steps_resident the device holds at once (-1: not asked yet)
    int cu_count = 0;
    StepIO *d_steps[2] = {nullptr, nullptr};
    size_t steps_cap[2] = {0, 0};
    std::vector<StepIO> h_steps[2];  // what d_steps holds
    unsigned long long *d_xch = nullptr;  // tagged halo exchange buffer: [2 parities][4 planes][P] granules of 16 bytes, see fused_xch_*
    unsigned tag_base = 0;        // tags handed out so far (the buffer never holds a tag above it)
    unsigned tag_base_init = 0;   // diagnostic: first tag base after the buffer is created
    int *d_abort = nullptr;
    int *h_abort = nullptr;       // pinned copies [2] (one per slot), valid after the call's last event
    bool abort_pending[2] = {false, false};  // a resident launch is in flight (or finished) whose verdict has not been looked at
    bool last_resident = false;   // the last fused_run took the single-launch path
    // the action-outliving launch (fused_body.h, "jobs")
    bool persist = true;          // WAVES_AMD_PERSIST=0: every resident launch ends with its job
    bool allow_persist = true;    // set per call by the owner (profiling, caller-owned streams, shared devices: false)
    unsigned idle_us = 1000;      // WAVES_AMD_IDLE_US: the launch leaves after this long without a new job (at most: see idle_cur)
    // How long the NEXT launch waits: a launch that waits occupies nearly every block slot of the device, so whatever else the
    // caller runs on it between two actions (a policy network, a planner: scripts/mpc.jl) crawls until the launch leaves.  The
    // host sees which it was: a call that finds the launch gone says the waiting was wasted -- a quarter as long next time (not
    // below 50 us, where a limit just above the last gap is tried now and then: a merely slow host gets its launch back that
    // way); one that finds it waiting lets the limit grow back, half as much again per call (never above idle_us).  WAVES_AMD_IDLE_ADAPT=0: always idle_us.  (700^2, a small torch MLP on state(env) between the actions: 2.24 ms
    // per action with a fixed 1 ms, 1.39 ms adaptive, tools/exp_gpu_policy.py.)
    unsigned idle_cur = 1000;
    bool idle_adapt = true, idle_probe = false;
    unsigned idle_backoff = 1, idle_exits = 0;
    std::chrono::steady_clock::time_point t_last_done{};  // when the host saw the newest job complete
    bool have_last_done = false;
    JobMail *mail = nullptr;      // pinned host memory
    JobBack *back = nullptr;      // pinned host memory
    JobCtl *d_ctl = nullptr;
    const TileDesc *launch_tiles = nullptr;  // the (device) tile table the newest job with a host-built table was given:
                                             // what a later call WITHOUT a table of its own (FusedDevTables) reads.  A copy of
                                             // its own, alternating between two buffers: the per-slot table it was copied from is
                                             // overwritten by the slot's next call -- possibly while such a job is still
                                             // starting (a compiled host begins the next call microseconds later: its tiles
                                             // then read two different launch orders, a tile is run twice, another not at
                                             // all, and the launch gives up)
    TileDesc *d_snap[2] = {nullptr, nullptr};
    size_t snap_cap[2] = {0, 0};
    int snap_w = -1;                         // the copy this call's fused_prepare has just filled (-1: none)
    int ctl_tiles = 0;            // tiles the flag arrays behind d_ctl were sized for
    unsigned seq = 0;             // jobs described so far (job numbers start at 1)
    // Launches: at most two are known at a time -- the newest, and the one before it while it still runs (a launch that ends
    // with its job, FusedParams::last, is simply followed by the next call's launch in stream order; only a launch that STAYS
    // has to be told to leave before anything else may use the stream).
    struct Launch {
        bool alive = false;       // started and not yet seen to have ended (its stop event not yet taken)
        bool stays = false;       // the newest job it was given lets it stay (FusedParams::last == 0)
        int ntiles = 0;
        unsigned first = 0;       // the first job it serves
        hipStream_t stream = nullptr;
        hipEvent_t start = nullptr, stop = nullptr;
    } L[2];
    int cur_l = 0;                // the newest launch
    hipEvent_t p_up = nullptr;
    int job_launch[2] = {0, 0};       // per slot: the launch its resident call was given to
    int job_rows[2] = {0, 0};         // per slot: blocks of that call that write trace rows to host memory
    unsigned job_seq[2] = {0, 0};     // per slot: the job of the slot's resident call that has not been waited for (0: none)
    bool job_keep[2] = {false, false};  // ... and whether the launch was asked to stay after it
    double last_job_ms = 0.0;     // in-kernel duration of the job waited for last
    double last_launch_ms = 0.0;  // duration (HIP events) and jobs of the launch that ended last
    int last_launch_jobs = 0;
    long n_launches = 0, n_jobs = 0, n_handed = 0;  // diagnostics (WAVES_AMD_HOSTPROF): launches, jobs, jobs handed to a launch that was there
    std::vector<double> job_ms;   // in-kernel duration of the jobs waited for since the last fused_job_stats reset
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
    if (const char *e = getenv("WAVES_AMD_FUSED_TILES")) {  // tuning knob: "RF,RB,RP" in {4,3,2 | 2,2,2}
        int a = 0, b = 0, c = 0;
        if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3) {
            const int key = a * 100 + b * 10 + c;
            if (key == 432 || key == 222) {
                p->RF = a;
                p->RB = b;
                p->RP = c;
            }
        }
    }
    if (const char *e = getenv("WAVES_AMD_FUSED_XCD")) p->xcd_aware = atoi(e) != 0;
    p->stamps_path = getenv("WAVES_AMD_STAMPS");
    if (const char *e = getenv("WAVES_AMD_FUSED_BANDS")) p->nbands = atoi(e) > 0 ? atoi(e) : 1;
    if (const char *e = getenv("WAVES_AMD_FUSED_GRAPH")) p->use_graph = atoi(e) != 0;
    if (const char *e = getenv("WAVES_AMD_FUSED_RESIDENT")) p->use_resident = atoi(e) != 0;
    if (const char *e = getenv("WAVES_AMD_TAG_BASE")) p->tag_base_init = (unsigned)strtoul(e, nullptr, 0);  // tests: wrap
    if (const char *e = getenv("WAVES_AMD_PERSIST")) p->persist = atoi(e) != 0;
    if (const char *e = getenv("WAVES_AMD_IDLE_US")) p->idle_us = (unsigned)std::max(0, atoi(e));
    if (p->idle_us == 0) p->persist = false;
    p->idle_cur = p->idle_us;
    if (const char *e = getenv("WAVES_AMD_IDLE_ADAPT")) p->idle_adapt = atoi(e) != 0;
    if (hipMalloc((void **)&p->d_flag, sizeof(int)) != hipSuccess || hipMalloc((void **)&p->d_abort, 4 * sizeof(int)) != hipSuccess ||
        hipHostMalloc((void **)&p->h_abort, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&p->up_ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->up_ev[1], hipEventDisableTiming) != hipSuccess ||
        hipMemset(p->d_abort, 0, 4 * sizeof(int)) != hipSuccess) {
        fused_destroy(p);
        return nullptr;
    }
    p->h_abort[0] = p->h_abort[1] = 0;
    {
        std::lock_guard<std::mutex> g(g_live.mu);
        g_live.plans.push_back(p);
    }
    return p;
}

void fused_destroy(FusedPlan *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_live.mu);
        g_live.plans.erase(std::remove(g_live.plans.begin(), g_live.plans.end(), p), g_live.plans.end());
    }
    (void)fused_retire(p);
    if (getenv("WAVES_AMD_HOSTPROF") && p->n_jobs)
        fprintf(stderr, "[waves_amd hostprof] resident launches %ld, jobs %ld, of them handed to a launch that was waiting: %ld\n",
                p->n_launches, p->n_jobs, p->n_handed);
    if (p->mail) (void)hipHostFree(p->mail);
    if (p->back) (void)hipHostFree(p->back);
    if (p->d_ctl) (void)hipFree(p->d_ctl);
    for (hipEvent_t e : {p->L[0].start, p->L[0].stop, p->L[1].start, p->L[1].stop, p->p_up})
        if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; ++k) {
        if (p->d_tiles[k]) (void)hipFree(p->d_tiles[k]);
        if (p->d_snap[k]) (void)hipFree(p->d_snap[k]);
        if (p->h_order[k]) (void)hipHostFree(p->h_order[k]);
        if (p->h_tiles[k]) (void)hipHostFree(p->h_tiles[k]);
        if (p->d_idx[k]) (void)hipFree(p->d_idx[k]);
        if (p->h_idx[k]) (void)hipHostFree(p->h_idx[k]);
        if (p->d_steps[k]) (void)hipFree(p->d_steps[k]);
        if (p->up_ev[k]) (void)hipEventDestroy(p->up_ev[k]);
        if (p->graph_exec[k]) (void)hipGraphExecDestroy(p->graph_exec[k]);
        if (p->graph[k]) (void)hipGraphDestroy(p->graph[k]);
    }
    if (p->d_flag) (void)hipFree(p->d_flag);
    if (p->d_stamps) (void)hipFree(p->d_stamps);
    if (p->d_src_flags) (void)hipFree(p->d_src_flags);
    if (p->d_xch) (void)hipFree(p->d_xch);
    if (p->d_abort) (void)hipFree(p->d_abort);
    if (p->h_abort) (void)hipHostFree(p->h_abort);
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
    p->f2_clean[0] = p->f2_clean[1] = false;
}

void fused_state_changed(FusedPlan *p)
{
    p->aux_state = -1;
    p->frames_clean = false;
    p->f2_clean[0] = p->f2_clean[1] = false;
}

void fused_state_zeroed(FusedPlan *p)  // (all of env.wave in its home buffers)
{
    p->aux_state = 1;
    p->frames_clean = true;
    p->f2_clean[0] = true;
}

static int device_slots(FusedPlan *pl);
static int resident_capacity(FusedPlan *pl);

static bool ensure_tiles(FusedPlan *p, bool aux_zero)
{
    if (p->tiles_valid && p->tiles_aux_zero == aux_zero) return true;
    auto build = [&](int oyf_cap, int oy_cap_all) {
        return plan_build_tiles(p->hp, p->g.nx, p->g.ny, p->NW * p->RF, p->NW * p->RB, p->NW * p->RP, p->x.data(), p->y.data(),
                                p->sx.data(), p->sy.data(), aux_zero, p->xcd_aware, p->nbands, oyf_cap, oy_cap_all);
    };
    if (!build(0, 0)) return false;
    // Resident kernel, small grids (fewer tiles than half the device's block slots): the lowest tiles that still fit,
    // every field set alike -- a step is a chain of latencies there, and more, smaller tiles shorten the arithmetic link
    // and put more CUs to work (256^2: 26 -> 90 tiles, +25 %; 384^2 +22 %; 500^2 +11 %).
    // Larger grids keep the tallest tiles the registers hold.  (Round 2 first also lowered the interior tiles of grids
    // that nearly fill the slots by up to two rows -- 700^2: 467 -> 489 tiles, -2 % at the time; with the cheaper halo
    // read of the final kernel the taller tiles win again: 700^2 +1.2 %, 600^2 +7 %.  WAVES_AMD_FUSED_AUTOTILE=2 brings
    // that rule back for experiments, =0 switches both off.)
    // (decided from the device alone, not from whether this call may run resident: both step kernels use the same tiles,
    // so their energy partial sums -- and with them the traces -- stay bit-identical)
    static const int autotile = getenv("WAVES_AMD_FUSED_AUTOTILE") ? atoi(getenv("WAVES_AMD_FUSED_AUTOTILE")) : 1;
    const int cap = (p->nbands == 1 && autotile != 0) ? device_slots(p) : 0;
    const int n0 = (int)p->hp.tiles.size();
    if (cap > 0 && n0 <= cap) {
        const int oyf = p->NW * p->RF - 2 * FT_H;
        int best_f = 0, best_all = 0, best_n = n0;
        if (2 * n0 < cap) {
            for (int c = oyf - 1; c >= 8; --c) {
                if (!build(0, c)) break;
                const int n = (int)p->hp.tiles.size();
                if (n * 100 > cap * 96) break;
                if (n > best_n) {
                    best_all = c;
                    best_n = n;
                }
            }
        } else if (autotile == 2) {
            for (int d = 1; d <= 2 && oyf - d >= 8; ++d) {
                if (!build(oyf - d, 0)) break;
                const int n = (int)p->hp.tiles.size();
                if (n > best_n && n * 100 <= cap * 96) {
                    best_f = oyf - d;
                    best_n = n;
                }
            }
        }
        if (!build(best_f, best_all)) return false;
    }
    p->src_dirty = true;
    p->tiles_valid = true;
    p->tiles_aux_zero = aux_zero;
    p->launch_tiles = nullptr;  // (a table of the previous tiling)
    p->generation++;
    return true;
}

int fused_energy_blocks(FusedPlan *p)
{
    if (!p->tiles_valid && !ensure_tiles(p, p->aux_state == 1)) return 0;
    return (int)p->hp.tiles.size();
}

void fused_source_changed(FusedPlan *p) { p->src_dirty = true; }

// May a call skip fused_prepare (no culling, no tile upload) and let the tiles cull themselves?  Only with coordinates the
// bounding-box culling is valid for, and a tile table from an earlier call of this tiling.
bool fused_dev_tables_ok(FusedPlan *p) { return p->tiles_valid && p->hp.monotonic && p->launch_tiles != nullptr; }
void fused_prepare_light(FusedPlan *p, int slot) { p->cur = slot; }

void fused_scratch_dirty(FusedPlan *p) { p->scratch_clean = false; }

// device + pinned host buffer pair of `need` elements (grown generously: a new device pointer is a new kernel argument,
// i.e. a re-instantiation of the cached hipGraph, ~40 ms)
template <class T>
static bool ensure_pair(T **d, T **h, size_t *cap, size_t need, size_t want)
{
    if (need <= *cap) return true;
    if (*d) (void)hipFree(*d);
    if (*h) (void)hipHostFree(*h);
    *d = nullptr;
    *h = nullptr;
    *cap = 0;
    if (hipMalloc((void **)d, want * sizeof(T)) != hipSuccess) return false;
    if (hipHostMalloc((void **)h, want * sizeof(T), hipHostMallocDefault) != hipSuccess) return false;
    *cap = want;
    return true;
}

int fused_prepare(FusedPlan *p, int slot, float *frames, float *ic, float *out2, int out2_idx, float *scratch0, float *scratch1, bool capture, const float *G,
                  const Cyl *d_table, const Cyl *h_table, int M, int rows, hipStream_t s, hipStream_t up, int row_lo, int row_hi,
                  bool defer_wait)
{
    const Grid &g = p->g;
    const size_t N = g.P * kFields;
    p->cur = slot;
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
        if (!p->f2_clean[out2_idx]) {
            hipLaunchKernelGGL(k_aux_clean, dim3(1024), dim3(256), 0, s, out2, g.nx, g.ny, g.P, g.sx, g.sy);
            p->f2_clean[out2_idx] = true;
        }
        p->f2_clean[out2_idx ^ 1] = true;  // (the initial condition itself: it satisfies the invariant, as just checked)
    } else {  // this integrate writes non-zero auxiliaries outside the PML
        p->scratch_clean = false;
        if (capture) p->frames_clean = false;
        p->f2_clean[out2_idx] = false;
    }
    if (!ensure_tiles(p, aux_zero)) return 2;
    static const bool resort = !(getenv("WAVES_AMD_FUSED_RESORT") && atoi(getenv("WAVES_AMD_FUSED_RESORT")) == 0);
    // (tiles that fit the device at once will run resident: order them for that kernel)
    static const bool pairing = !(getenv("WAVES_AMD_FUSED_PAIRING") && atoi(getenv("WAVES_AMD_FUSED_PAIRING")) == 0);
    const bool fits = (int)p->hp.tiles.size() <= resident_capacity(p);
    plan_build_cyl(p->hp, p->x.data(), p->y.data(), h_table, M, rows, p->idx, resort, (fits && pairing) ? p->cu_count : 0,
                   row_lo, row_hi);
    const size_t nt = p->hp.tiles.size();
    const size_t ni = p->idx.size() ? p->idx.size() : 1;
    if (const char *dump = getenv("WAVES_AMD_PLAN_DUMP")) {  // diagnostic: the launch order of this call (tools/plan_balance.py)
        if (FILE *f = fopen(dump, "w")) {
            fprintf(f, "# cus %d slots %d RYF %d RYB %d RYP %d | pos x0 y0 ox oy aux edge cyl slot\n", p->cu_count, device_slots(p), p->hp.RYF, p->hp.RYB, p->hp.RYP);
            for (size_t i = 0; i < nt; ++i) {
                const TileDesc &t = p->hp.tiles[i];
                fprintf(f, "%zu %d %d %d %d %d %d %d %d\n", i, t.x0, t.y0, t.ox, t.oy, t.aux, t.edge, t.cyl_count, t.slot);
            }
            fclose(f);
        }
    }
    // (the slot's buffers were last read by the call before the previous one, which the caller has ended)
    if (!ensure_pair(&p->d_tiles[slot], &p->h_tiles[slot], &p->tiles_cap[slot], nt, nt)) return 1;
    if (!ensure_pair(&p->d_idx[slot], &p->h_idx[slot], &p->idx_cap[slot], ni, std::max(2 * ni, nt * 16))) return 1;
    memcpy(p->h_tiles[slot], p->hp.tiles.data(), nt * sizeof(TileDesc));
    if (!p->idx.empty()) memcpy(p->h_idx[slot], p->idx.data(), p->idx.size() * sizeof(int));
    if (hipMemcpyAsync(p->d_tiles[slot], p->h_tiles[slot], nt * sizeof(TileDesc), hipMemcpyHostToDevice, up) != hipSuccess) return 1;
    {   // ... and the copy for later calls without a table: into the buffer that is NOT the current launch_tiles (a pending job may be reading that)
        const int w = (p->launch_tiles != nullptr && p->launch_tiles == p->d_snap[0]) ? 1 : 0;
        if (nt > p->snap_cap[w]) {
            if (p->d_snap[w]) (void)hipFree(p->d_snap[w]);
            p->d_snap[w] = nullptr;
            p->snap_cap[w] = 0;
            if (hipMalloc((void **)&p->d_snap[w], nt * sizeof(TileDesc)) != hipSuccess) return 1;
            p->snap_cap[w] = nt;
        }
        if (hipMemcpyAsync(p->d_snap[w], p->h_tiles[slot], nt * sizeof(TileDesc), hipMemcpyHostToDevice, up) != hipSuccess) return 1;
        p->snap_w = w;
    }
    if (!p->idx.empty() &&
        hipMemcpyAsync(p->d_idx[slot], p->h_idx[slot], p->idx.size() * sizeof(int), hipMemcpyHostToDevice, up) != hipSuccess)
        return 1;
    // everything uploaded so far (the caller's coefficient tables included) before anything of this call runs on s
    // (a resident launch that is waiting for jobs owns `s`: fused_try_resident then waits for the copy stream on the host)
    if (up != s && !defer_wait && (hipEventRecord(p->up_ev[slot], up) != hipSuccess || hipStreamWaitEvent(s, p->up_ev[slot], 0) != hipSuccess)) return 1;
    if (G && p->src_dirty) {
        if (nt > p->src_flags_cap) {
            if (p->d_src_flags) (void)hipFree(p->d_src_flags);
            p->d_src_flags = nullptr;
            p->src_flags_cap = 0;
            if (hipMalloc((void **)&p->d_src_flags, nt) != hipSuccess) return 1;
            p->src_flags_cap = nt;
        }
        hipLaunchKernelGGL(k_src_flags, dim3((unsigned)nt), dim3(256), 0, s, p->d_tiles[slot], G, g.nx, g.ny, p->d_src_flags);
        p->src_dirty = false;
    }
    p->d_table = d_table;
    p->M = M;
    return 0;
}

static FusedParams make_params(FusedPlan *pl, const FusedCall &call, int step, const FusedStep &st)
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
    p.io.u = st.u;
    p.io.out = st.out;
    p.io.epart = st.epart;
    p.io.traj_tot = st.traj_tot;
    p.io.traj_inc = st.traj_inc;
    p.io.step = step;
    p.G = call.G;
    p.src_flags = call.G ? pl->d_src_flags : nullptr;
    p.sfac_tab = call.G ? call.d_sfac : nullptr;
    p.cyl_tab = pl->d_table;
    p.M = pl->M;
    p.dt = call.dt;
    p.hdt = 0.5f * call.dt;
    p.tiles = pl->d_tiles[pl->cur];
    p.tile_offset = 0;
    p.cyl_idx = pl->d_idx[pl->cur];
    p.steps = nullptr;
    p.nsteps = 0;
    p.xch = nullptr;
    p.xch_bytes = 0;
    p.tag_base = 0;
    p.reduced = 0;
    p.max_polls = 0;
    p.abort = nullptr;
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
    return p;
}

static const void *kernel_ptr(const FusedPlan *pl)
{
    switch (pl->RF * 100 + pl->RB * 10 + pl->RP) {
        case 222: return (const void *)k_step_fused<8, 2, 2, 2>;
        default: return (const void *)k_step_fused<8, 4, 3, 2>;
    }
}

void fused_launch(FusedPlan *pl, int slot, const FusedCall &call, int step, const FusedStep &st, hipStream_t s)
{
    pl->cur = slot;
    FusedParams p = make_params(pl, call, step, st);
    void *args[1] = {&p};
    // WAVES_AMD_FUSED_PADLDS (diagnostic): extra dynamic LDS per block, to force one block per CU in occupancy studies
    static const size_t pad = getenv("WAVES_AMD_FUSED_PADLDS") ? (size_t)atoi(getenv("WAVES_AMD_FUSED_PADLDS")) : 0;
    (void)hipLaunchKernel(kernel_ptr(pl), dim3((unsigned)pl->hp.tiles.size()), dim3(512), args, pad, s);
}

template <class T>
static void key_put(std::vector<char> &k, const T &v)
{
    const char *b = reinterpret_cast<const char *>(&v);
    k.insert(k.end(), b, b + sizeof(T));
}

static const void *resident_ptr(const FusedPlan *pl)
{
    switch (pl->RF * 100 + pl->RB * 10 + pl->RP) {
        case 222: return (const void *)k_steps_resident<8, 2, 2, 2>;
        default: return (const void *)k_steps_resident<8, 4, 3, 2>;
    }
}

// blocks of k_steps_resident the device holds at once, whether or not this context may use that kernel right now
static int device_slots(FusedPlan *pl)
{
    if (pl->resident_capacity < 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, resident_ptr(pl), 512, 0) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            (void)hipGetLastError();
            pl->resident_capacity = 0;
        } else {
            pl->resident_capacity = (prop.cooperativeLaunch ? per_cu * prop.multiProcessorCount : 0);
            pl->cu_count = prop.multiProcessorCount;
        }
    }
    return pl->resident_capacity;
}

// The one hardware property the resident kernel's halo exchange relies on beyond the ISA's promises -- a 16-byte-aligned
// 16-byte agent-scope access is never observed torn (fused_body.h, fused_xch_*) -- is checked ONCE per process and device,
// before the first resident launch (a short run of wv_selftest_granules' experiment, ~2 ms): on a device where a granule
// is ever seen torn the resident kernel is not used at all.  WAVES_AMD_SELFTEST=0 skips the check.
static bool granules_ok(hipStream_t s)
{
    static std::mutex mu;
    static int verdict[64] = {0};  // 0 not asked, 1 fine, -1 torn (or the check itself failed)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> g(mu);
    int &v = verdict[dev & 63];
    if (v != 0) return v > 0;
    if (getenv("WAVES_AMD_SELFTEST") && atoi(getenv("WAVES_AMD_SELFTEST")) == 0) {
        v = 1;
        return true;
    }
    const unsigned bytes = 8u << 20;
    unsigned char *buf = nullptr;
    unsigned long long *out = nullptr, h[2] = {0, 0};
    hipError_t e = hipMalloc((void **)&buf, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&out, sizeof(h));
    if (e == hipSuccess) e = hipMemsetAsync(out, 0, sizeof(h), s);
    const unsigned strides[2] = {16u, 64u};
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipMemsetAsync(buf, 0, bytes, s);
        if (e == hipSuccess) launch_selftest_granules(buf, bytes, 3000, 64, strides[k], out, s);
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, out, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (buf) (void)hipFree(buf);
    if (out) (void)hipFree(out);
    if (e != hipSuccess) (void)hipGetLastError();
    // (no granule seen -- readers that ran before any writer, as under a profiler that serialises workgroups -- is
    // inconclusive, not a failure)
    v = (e == hipSuccess && h[1] == 0) ? 1 : -1;
    if (v < 0)
        fprintf(stderr, "[waves_amd] 16-byte granule self-test: %llu of %llu granules torn (status %d): the resident step kernel is not used "
                        "on this device\n", h[1], h[0], (int)e);
    return v > 0;
}

// ... and 0 when the resident path is not available to this call
static int resident_capacity(FusedPlan *pl)
{
    if (!pl->use_resident || !pl->allow_resident || pl->nbands != 1) return 0;
    return device_slots(pl);
}

static_assert(kDevTablesMaxCyl == FT_MAXCYL && kDevTablesMaxSteps == JOB_MAXSTEPS, "fused.h and fused_body.h disagree");


// ---- the action-outliving launch: host side ----------------------------------------------------------------------------
LiveLaunches::~LiveLaunches()
{
    std::lock_guard<std::mutex> g(mu);
    for (FusedPlan *pl : plans) {
        if (!pl->mail || !pl->back) continue;
        const int l = pl->cur_l;
        if (!pl->L[l].alive || !pl->L[l].stays || __atomic_load_n(&pl->back->status[l], __ATOMIC_ACQUIRE) != JOBS_RUNNING) continue;
        pl->seq++;
        FusedParams &d = pl->mail->desc[pl->seq & 1u].p;
        d = FusedParams{};
        d.seq = pl->seq;
        d.cmd = JOB_EXIT;
        __atomic_store_n(&pl->mail->bell, pl->seq, __ATOMIC_RELEASE);
        const auto t0 = std::chrono::steady_clock::now();
        while (__atomic_load_n(&pl->back->status[l], __ATOMIC_ACQUIRE) == JOBS_RUNNING &&
               std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(50)) {}
    }
}

static bool jobs_ensure(FusedPlan *pl, int ntiles)
{
    if (!pl->mail) {
        if (hipHostMalloc((void **)&pl->mail, sizeof(JobMail), hipHostMallocDefault) != hipSuccess) return false;
        memset((void *)pl->mail, 0, sizeof(JobMail));
    }
    if (!pl->back) {
        if (hipHostMalloc((void **)&pl->back, sizeof(JobBack), hipHostMallocDefault) != hipSuccess) return false;
        memset((void *)pl->back, 0, sizeof(JobBack));
    }
    if (!pl->p_up) {
        for (FusedPlan::Launch &l : pl->L)
            if (hipEventCreate(&l.start) != hipSuccess || hipEventCreate(&l.stop) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&pl->p_up, hipEventDisableTiming) != hipSuccess) return false;
    }
    if (!pl->d_ctl || ntiles > pl->ctl_tiles) {  // (never while a launch is alive: the caller waits for them when the tiling changes)
        if (pl->d_ctl) (void)hipFree(pl->d_ctl);
        pl->d_ctl = nullptr;
        const int cap = std::max(ntiles, 1024);
        if (hipMalloc((void **)&pl->d_ctl, job_ctl_bytes(cap)) != hipSuccess) return false;
        if (hipMemset(pl->d_ctl, 0, job_ctl_bytes(cap)) != hipSuccess) return false;
        pl->ctl_tiles = cap;
    }
    return true;
}

static const bool g_trace = getenv("WAVES_AMD_TRACE") != nullptr;  // diagnostic: the host's decisions of the job protocol, on stderr
#define WV_TRACE(...)                        \
    do {                                     \
        if (g_trace) {                       \
            fprintf(stderr, "[wv trace] ");  \
            fprintf(stderr, __VA_ARGS__);    \
            fprintf(stderr, "\n");           \
        }                                    \
    } while (0)
static unsigned back_status(FusedPlan *pl, int l) { return __atomic_load_n(&pl->back->status[l], __ATOMIC_ACQUIRE); }
static unsigned back_done(FusedPlan *pl) { return __atomic_load_n(&pl->back->done, __ATOMIC_ACQUIRE); }

// The newest launch is on the device, stays there after its jobs, and has not said that it is leaving: it owns the stream
// (anything enqueued there would wait out its idle limit) and takes further jobs.
bool fused_persist_alive(FusedPlan *pl)
{
    const FusedPlan::Launch &l = pl->L[pl->cur_l];
    return l.alive && l.stays && back_status(pl, pl->cur_l) == JOBS_RUNNING;
}

// launch l has ended (its stop event has fired): bookkeeping
static void jobs_reap(FusedPlan *pl, int l)
{
    FusedPlan::Launch &L = pl->L[l];
    if (!L.alive) return;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, L.start, L.stop) == hipSuccess) {
        pl->last_launch_ms = ms;
        // the jobs it completed: from its first job to the last one done before the next launch's first (or to `done`)
        const unsigned upto = (l != pl->cur_l && pl->L[pl->cur_l].alive) ? pl->L[pl->cur_l].first - 1 : back_done(pl);
        pl->last_launch_jobs = job_reached(upto, L.first) ? (int)(upto - L.first + 1) : 0;
    } else {
        (void)hipGetLastError();
    }
    L.alive = false;
}

// wait (polling first: a launch that has said it is leaving is microseconds from its end) until launch l has ended
static int jobs_join(FusedPlan *pl, int l)
{
    FusedPlan::Launch &L = pl->L[l];
    if (!L.alive) return 0;
    hipError_t e;
    unsigned n = 0;
    while ((e = hipEventQuery(L.stop)) == hipErrorNotReady && ++n < (1u << 20)) {}
    if (e == hipErrorNotReady) e = hipEventSynchronize(L.stop);
    jobs_reap(pl, l);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 1;
    }
    return 0;
}

// Make a launch that stays leave (after the jobs it has been given) and wait until every launch has ended.  Everything that
// wants the context's stream for itself goes through here first.
int fused_retire(FusedPlan *pl)
{
    if (!pl) return 0;
    if (fused_persist_alive(pl)) {
        pl->seq++;
        WV_TRACE("retire: EXIT as job %u (done %u)", pl->seq, back_done(pl));
        FusedParams &d = pl->mail->desc[pl->seq & 1u].p;
        d = FusedParams{};
        d.seq = pl->seq;
        d.cmd = JOB_EXIT;
        __atomic_store_n(&pl->mail->bell, pl->seq, __ATOMIC_RELEASE);
    }
    int rc = jobs_join(pl, pl->cur_l ^ 1);
    rc |= jobs_join(pl, pl->cur_l);
    return rc;
}

// Would fused_prepare / fused_try_resident enqueue anything on the context's stream for such a call?  (Then a launch that
// stays has to be retired first: see fused_retire.)
bool fused_needs_stream(FusedPlan *pl, bool capture, const float *G, int out2_idx)
{
    if (pl->aux_state < 0) return true;
    const bool aux_zero = pl->aux_state == 1;
    if (aux_zero && (!pl->scratch_clean || (capture && !pl->frames_clean) || !pl->f2_clean[out2_idx])) return true;
    if (!pl->tiles_valid || pl->tiles_aux_zero != aux_zero) return true;
    if (G && pl->src_dirty) return true;
    if (!pl->d_xch || pl->tag_base > 0xFFFF0000u - (1u << 21)) return true;
    return false;
}

// a new launch, behind whatever is on `s` (an earlier launch that ends with its job included)
static int jobs_launch(FusedPlan *pl, unsigned first_seq, int ntiles, hipStream_t s)
{
    const int l = pl->cur_l ^ 1;
    WV_TRACE("launch record %d first job %u tiles %d", l, first_seq, ntiles);
    if (jobs_join(pl, l) != 0) return 1;  // (the launch before the previous one: long gone)
    // (leftovers of an earlier launch in the go words -- "job n: leave" -- must not be taken for this launch's answer)
    if (hipMemsetAsync(pl->d_ctl->go, 0, sizeof(pl->d_ctl->go), s) != hipSuccess) return 1;
    __atomic_store_n(&pl->back->status[l], (unsigned)JOBS_RUNNING, __ATOMIC_RELEASE);
    JobArgs a{};
    a.mail = pl->mail;
    a.back = pl->back;
    a.ctl = pl->d_ctl;
    a.first_seq = first_seq;
    const unsigned long long ticks = (unsigned long long)(pl->idle_adapt ? pl->idle_cur : pl->idle_us) * 100ull;
    a.idle_ticks = (unsigned)std::min<unsigned long long>(ticks, 0x7fffffffull);
    a.ntiles = ntiles;
    a.launch = l;
    void *args[1] = {&a};
    // A plain launch: the grid is at most the number of blocks the device holds at once (resident_capacity, from the
    // occupancy query -- the same number a cooperative launch would check it against), and an ordinary launch has the same
    // residency.  hipLaunchCooperativeKernel (WAVES_AMD_COOP=1) additionally serialises the launch against the work of
    // every other stream of the process.  The launch carries its two events itself (hipExtLaunchKernel: they take the
    // kernel's own start and end timestamps).
    static const bool coop = getenv("WAVES_AMD_COOP") && atoi(getenv("WAVES_AMD_COOP")) != 0;
    FusedPlan::Launch &L = pl->L[l];
    hipError_t e;
    if (coop) {
        (void)hipEventRecord(L.start, s);
        e = hipLaunchCooperativeKernel(resident_ptr(pl), dim3((unsigned)ntiles), dim3(512), args, 0, s);
        if (e == hipSuccess) (void)hipEventRecord(L.stop, s);
    } else {
        e = hipExtLaunchKernel(resident_ptr(pl), dim3((unsigned)ntiles), dim3(512), args, 0, s, L.start, L.stop, 0);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    L.alive = true;
    L.stays = false;  // (the caller says)
    L.ntiles = ntiles;
    L.first = first_seq;
    L.stream = s;
    pl->cur_l = l;
    pl->n_launches++;
    return 0;
}

// `up`: the stream the caller's uploads of this call went to (its tables must have arrived before the job may start);
// `ef`: where the trace goes.  keep: the launch may stay on the device after the job (see FusedPlan::persist).
int fused_try_resident(FusedPlan *pl, int slot, const FusedCall &call, const FusedStep *steps, int nsteps, hipStream_t s,
                       hipStream_t up, const FusedEnergy &ef, bool keep, const FusedDevTables *dev, const FusedObs *ob)
{
    pl->cur = slot;
    const size_t nt = pl->hp.tiles.size();
    if (nsteps < 2 || (int)nt > resident_capacity(pl) || (int)nt > JOB_MAX_TILES) return dev ? 3 : -1;
    bool alive = fused_persist_alive(pl);
    if (pl->idle_adapt && pl->have_last_done) {  // (see FusedPlan::idle_cur)
        const double gap = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - pl->t_last_done).count();
        const unsigned lo = std::min(50u, pl->idle_us);
        if (alive) {  // found waiting: the limit may grow back, half as much again per call
            pl->idle_cur = std::min(pl->idle_us, pl->idle_cur + pl->idle_cur / 2u + 1u);
            pl->idle_probe = false;
            pl->idle_backoff = 1;
            pl->idle_exits = 0;
        } else if (pl->L[pl->cur_l].alive && back_status(pl, pl->cur_l) == JOBS_EXIT_IDLE) {
            // It waited out its limit in vain.  How long the caller WOULD have taken cannot be told from the gap: work of the
            // caller's on this device only gets going once the launch has left, so the gap is "limit + that work" -- or the
            // host was simply slow (then a limit above the gap would have caught the call).  So: shrink; and once at the floor,
            // try now and then whether a limit above the last gap catches the next call (a probe that fails is dropped at once,
            // and the next one comes after twice as many exits).
            if (pl->idle_probe) {
                pl->idle_probe = false;
                pl->idle_cur = lo;
                pl->idle_backoff = std::min(64u, pl->idle_backoff * 2u);
                pl->idle_exits = 0;
            } else if (pl->idle_cur > lo) {
                pl->idle_cur = std::max(lo, pl->idle_cur / 4u);
            } else if (++pl->idle_exits >= pl->idle_backoff && 2.0 * gap < (double)pl->idle_us) {
                pl->idle_cur = std::max(lo, (unsigned)(2.0 * gap));
                pl->idle_probe = true;
            }
        }
        pl->have_last_done = false;  // (one verdict per completed job)
    }
    if (!alive && !pl->granules_checked) {  // (before this plan's first resident launch; the stream is free then)
        pl->granules_checked = true;
        if (!granules_ok(s)) {
            pl->use_resident = false;
            return dev ? 3 : -1;
        }
    }
    // (a new tiling: new grid; no copy stream: the uploads would queue behind the launch they are meant for)
    if (alive && ((int)nt != pl->L[pl->cur_l].ntiles || s != pl->L[pl->cur_l].stream || up == s)) {
        if (fused_retire(pl)) return 1;
        alive = false;
    }
    if (!alive && pl->L[pl->cur_l].alive && pl->L[pl->cur_l].stays) {  // it has said it is leaving (idle limit): take its end
        if (jobs_join(pl, pl->cur_l)) return 1;
    }
    if (!pl->d_ctl || (int)nt > pl->ctl_tiles) {
        if (fused_retire(pl)) return 1;  // (nothing alive may hold the control block that is about to be replaced)
    }
    if (!jobs_ensure(pl, (int)nt)) {
        fprintf(stderr, "[waves_amd] the resident step kernel is not used by this context: its control blocks could not be allocated (%s)\n",
                hipGetErrorString(hipGetLastError()));
        pl->resident_capacity = 0;
        return dev ? 3 : -1;
    }
    // exchange buffer: zeroed once (tag 0 is never expected); tags only grow, so words of earlier calls -- or of an
    // earlier tile decomposition -- can never be mistaken for the ones a step waits for
    const size_t xwords = (size_t)2 * XCH_PLANES * 2 * pl->g.P;  // 8-byte words: 2 parities x 4 planes x P granules of 16 bytes
    if (!pl->d_xch || pl->tag_base > 0xFFFF0000u - (unsigned)nsteps) {
        if (alive) {  // (fused_needs_stream says so beforehand; a call that got here all the same takes the long way)
            if (fused_retire(pl)) return 1;
            alive = false;
        }
        if (!pl->d_xch && hipMalloc((void **)&pl->d_xch, xwords * sizeof(unsigned long long)) != hipSuccess) {
            (void)hipGetLastError();
            pl->resident_capacity = 0;
            return -1;
        }
        if (hipMemsetAsync(pl->d_xch, 0, xwords * sizeof(unsigned long long), s) != hipSuccess) return 1;
        pl->tag_base = pl->tag_base_init;
        pl->tag_base_init = 0;
    }
    // the step table (unchanged from call to call in a rollout: uploaded only when it differs)
    std::vector<StepIO> tab((size_t)nsteps);
    for (int i = 0; i < nsteps; ++i) {
        tab[i] = StepIO{steps[i].u, (steps[i].keep || i + 1 == nsteps) ? steps[i].out : nullptr, steps[i].epart,
                        steps[i].traj_tot, steps[i].traj_inc, i, 0};
        if (i > 0 && steps[i].u != steps[i - 1].out) return dev ? 3 : -1;
    }
    const bool same = tab.size() == pl->h_steps[slot].size() &&
                      memcmp(tab.data(), pl->h_steps[slot].data(), tab.size() * sizeof(StepIO)) == 0;
    if (!same) {
        if (tab.size() > pl->steps_cap[slot]) {
            if (pl->d_steps[slot]) (void)hipFree(pl->d_steps[slot]);
            pl->d_steps[slot] = nullptr;
            pl->steps_cap[slot] = 0;
            pl->h_steps[slot].clear();
            if (hipMalloc((void **)&pl->d_steps[slot], tab.size() * sizeof(StepIO)) != hipSuccess) return 1;
            pl->steps_cap[slot] = tab.size();
        }
        // (rare: the table only changes with the call's shape.  The slot's previous call has been ended by the caller;
        // the host vector is not touched again before the slot's next call.  On the copy stream: a live launch owns `s`.)
        pl->h_steps[slot] = tab;
        if (hipMemcpyAsync(pl->d_steps[slot], pl->h_steps[slot].data(), tab.size() * sizeof(StepIO), hipMemcpyHostToDevice, up) != hipSuccess)
            return 1;
    }
    FusedParams p = make_params(pl, call, 0, steps[0]);
    p.steps = pl->d_steps[slot];
    p.nsteps = nsteps;
    p.xch = pl->d_xch;
    p.xch_bytes = (unsigned)(xwords * sizeof(unsigned long long));
    p.tag_base = pl->tag_base;
    p.reduced = pl->tiles_aux_zero ? 1 : 0;
    const char *mp = getenv("WAVES_AMD_WAIT_POLLS");  // diagnostic, read per call (tests force a give-up with it)
    p.max_polls = (mp && atoi(mp) > 0) ? atoi(mp) : pl->max_polls;
    p.abort = pl->d_abort;
    p.seq = pl->seq + 1;
    p.cmd = JOB_RUN;
    p.last = (keep && pl->persist && pl->allow_persist && !pl->stamps_path) ? 0 : 1;
    p.ntiles = (int)nt;
    p.ef_row0 = ef.row0;
    p.ef_epart = ef.epart;
    p.ef_signal = ef.signal;
    p.ef_dOmega = ef.dOmega;
    if (ob && ob->out) {
        p.ob_f0 = ob->f0, p.ob_f1 = ob->f1, p.ob_f2 = ob->f2, p.ob_G = ob->G;
        p.ob_out = ob->out;
        p.ob_rx = ob->rx, p.ob_ry = ob->ry;
    }
    p.ctl = pl->d_ctl;
    static const bool joblog_ = getenv("WAVES_AMD_JOBLOG") != nullptr;
    p.back = joblog_ ? pl->back : nullptr;
    JobDesc &desc = pl->mail->desc[p.seq & 1u];
    if (dev) {
        // The tiles evaluate and cull their cylinders themselves and find the call's small tables in the job description:
        // nothing was built, uploaded or has to be waited for.  The tile table is the one the launch's first job was given
        // (the launch order is a matter of speed only; it stays as that job's culling shaped it).
        if (!pl->launch_tiles || dev->M < 1 || dev->M > FT_MAXCYL || nsteps > JOB_MAXSTEPS) return 1;  // (the caller checked)
        JobDesc *dj = &pl->d_ctl->jobs[p.seq & 1u];  // (device address)
        desc.dsg.M = dev->M;
        desc.dsg.ti = dev->ti;
        desc.dsg.tf = dev->tf;
        design_slopes(desc.dsg, dev->d0, dev->d1);
        memcpy(desc.tspan, dev->tspan, (size_t)(nsteps + 1) * sizeof(float));
        if (dev->sfac) memcpy(desc.sfac, dev->sfac, 3 * (size_t)nsteps * sizeof(float));
        p.dsg = &dj->dsg;
        p.tspan = dj->tspan;
        p.sfac_tab = call.G ? dj->sfac : nullptr;
        p.cyl_tab = nullptr;
        p.cyl_idx = nullptr;
        p.M = dev->M;
        p.dev_cull = 1;
        p.cull_t_lo = dev->t_lo;
        p.cull_t_hi = dev->t_hi;
        p.tiles = pl->launch_tiles;
        // The launch ORDER is made for this call's design (which two tiles share a CU, fused_plan.h plan_pair_order): the order an
        // earlier call left behind goes stale as the design walks away from that call's -- 700^2, random radii, one action at a
        // time: jobs of 868 us right after a call with tables, 915-955 us five actions later, 840-850 with this.  The culling
        // of the call's two end designs and the sorts are host work in front of the bell (WAVES_AMD_DEV_ORDER=0: the old table);
        // the table travels in pinned memory that every block reads its own 40 bytes of.
        const char *dev_order_env = getenv("WAVES_AMD_DEV_ORDER");  // (read per call: the tests compare the two in one process)
        const bool dev_order = !(dev_order_env && atoi(dev_order_env) == 0);
        static const bool pairing = !(getenv("WAVES_AMD_FUSED_PAIRING") && atoi(getenv("WAVES_AMD_FUSED_PAIRING")) == 0);
        if (dev_order && pairing && pl->hp.monotonic && (int)nt > pl->cu_count && (int)nt <= 2 * pl->cu_count) {
            const int k = (int)(p.seq & 1u);
            if (nt > pl->order_cap[k]) {
                if (pl->h_order[k]) (void)hipHostFree(pl->h_order[k]);
                pl->h_order[k] = nullptr;
                pl->order_cap[k] = 0;
                if (hipHostMalloc((void **)&pl->h_order[k], nt * sizeof(TileDesc), hipHostMallocDefault) != hipSuccess) return 1;
                pl->order_cap[k] = nt;
            }
            pl->order_ends.resize((size_t)2 * dev->M);
            for (int m = 0; m < dev->M; ++m) {
                pl->order_ends[m] = design_cyl(desc.dsg, m, dev->t_lo);
                pl->order_ends[(size_t)dev->M + m] = design_cyl(desc.dsg, m, dev->t_hi);
            }
            plan_build_cyl(pl->hp, pl->x.data(), pl->y.data(), pl->order_ends.data(), dev->M, 2, pl->idx, true, pl->cu_count, 0, 1);
            memcpy(pl->h_order[k], pl->hp.tiles.data(), nt * sizeof(TileDesc));
            p.tiles = pl->h_order[k];
        }
    } else if (pl->snap_w >= 0) {
        pl->launch_tiles = pl->d_snap[pl->snap_w];
        pl->snap_w = -1;
    }
    // Everything the job reads must be in device memory before the bell rings.  A launch that is already there cannot be
    // made to wait by the stream, so the host waits for the copy stream itself (the uploads are ~100 KB: they are long
    // done when the previous job ends, and in the two-in-flight rhythm this wait sits under that job).
    if (alive && (!dev || !same)) {
        if (hipEventRecord(pl->p_up, up) != hipSuccess) return 1;
        hipError_t q;
        while ((q = hipEventQuery(pl->p_up)) == hipErrorNotReady) {}
        if (q != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
        if (!fused_persist_alive(pl)) {  // it left meanwhile (idle limit)
            if (jobs_join(pl, pl->cur_l)) return 1;
            alive = false;
        }
    }
    if (!alive && up != s) {  // the new launch waits in stream order
        if (hipEventRecord(pl->p_up, up) != hipSuccess || hipStreamWaitEvent(s, pl->p_up, 0) != hipSuccess) return 1;
    }
    pl->seq++;
    desc.p = p;
    __atomic_store_n(&pl->mail->bell, pl->seq, __ATOMIC_RELEASE);
    WV_TRACE("ring job %u slot %d alive %d dev %d last %d tiles %p steps %p same %d done %u", pl->seq, slot, (int)alive, dev ? 1 : 0, p.last, (const void *)p.tiles,
             (const void *)p.steps, (int)same, back_done(pl));
    if (!alive) {
        // The new launch starts with this job -- unless the call before it is still pending and was rung just as the previous
        // launch left on its idle limit without taking it (two in flight; found by the soak test: begun with this job, the new
        // launch skipped the other one, whose wv_integrate_end then took this job's completion for its own).  A pending call
        // that a launch is still WORKING on (one that ends with its job) is not meant: that launch says so (exit_seq / status).
        unsigned first = pl->seq;
        {
            const unsigned other = pl->job_seq[slot ^ 1];
            const int lo = pl->cur_l;
            if (other != 0 && other + 1u == pl->seq && back_status(pl, lo) == JOBS_EXIT_IDLE &&
                __atomic_load_n(&pl->back->exit_seq[lo], __ATOMIC_ACQUIRE) == other && !job_reached(back_done(pl), other))
                first = other;
        }
        const int rc = jobs_launch(pl, first, (int)nt, s);
        if (rc != 0) {
            pl->seq--;  // (nobody has seen the description)
            __atomic_store_n(&pl->mail->bell, pl->seq, __ATOMIC_RELEASE);
            if (rc < 0) {  // do not try again
                fprintf(stderr, "[waves_amd] the resident step kernel is not used by this context: its launch was refused\n");
                pl->resident_capacity = 0;
            }
            return rc;
        }
    }
    pl->L[pl->cur_l].stays = p.last == 0;
    pl->tag_base += (unsigned)nsteps;
    pl->n_jobs++;
    pl->n_handed += alive ? 1 : 0;
    pl->job_seq[slot] = pl->seq;
    pl->job_launch[slot] = pl->cur_l;
    pl->job_keep[slot] = p.last == 0;
    // (blocks that report through JobBack::rowdone: the ones with trace rows and the ones with a share of the observation)
    pl->job_rows[slot] = std::max(ef.signal ? std::min(nsteps + 1, (int)nt) : 0, (ob && ob->out) ? std::min(JOB_OBS_BLOCKS, (int)nt) : 0);
    pl->abort_pending[slot] = true;
    return 0;
}

// Wait for the slot's resident call.  0: done; 2: the launch gave the call up -- its initial condition is intact (the final
// state has a buffer of its own) and the caller runs it again with the single-step kernels (fused_rerun_steps); 1: HIP
// error.  A launch that left on its idle limit before it saw the job is started again here.
int fused_job_wait(FusedPlan *pl, int slot, hipStream_t s)
{
    const unsigned want = pl->job_seq[slot];
    if (!want) return 0;
    pl->job_seq[slot] = 0;
    pl->abort_pending[slot] = false;
    const int writers = pl->job_rows[slot];  // blocks that put trace rows into host memory (0: no trace wanted)
    pl->job_rows[slot] = 0;
    int l = pl->job_launch[slot];
    auto complete = [&]() {
        if (!job_reached(back_done(pl), want)) return false;
        for (int k = writers - 1; k >= 0; --k)
            if (!job_reached(__atomic_load_n(&pl->back->rowdone[k], __ATOMIC_ACQUIRE), want)) return false;
        return true;
    };
    unsigned spins = 0;
    for (;;) {
        if (complete()) break;
        if ((++spins & 1023u) != 0) continue;
        // not done yet: is the launch still there?
        FusedPlan::Launch &L = pl->L[l];
        const bool ended = !L.alive || hipEventQuery(L.stop) == hipSuccess;
        if (!ended) {
            (void)hipGetLastError();  // (hipErrorNotReady)
            continue;
        }
        if (complete()) break;
        const unsigned why = back_status(pl, l);
        WV_TRACE("wait job %u: launch record %d has ended, status %u exit_seq %u done %u", want, l, why, pl->back->exit_seq[l], back_done(pl));
        jobs_reap(pl, l);
        if (why == JOBS_EXIT_ABORT) return 2;
        // it left on its idle limit (or was told to) without having seen this job: the description and the bell are still
        // there -- a new launch takes over from the first job that is not done
        if (l == pl->cur_l) {
            const int rc = jobs_launch(pl, back_done(pl) + 1, L.ntiles, s);
            if (rc != 0) return 1;
            pl->L[pl->cur_l].stays = pl->job_keep[slot];
        }
        l = pl->cur_l;  // (or: a newer launch exists already and will get to it)
    }
    // in-kernel duration of the job (the leader's 100 MHz stamps)
    const unsigned par = want & 1u;
    const unsigned long long t0 = pl->back->t_begin[par], t1 = pl->back->t_end[par];
    const double ms = t1 > t0 ? (double)(t1 - t0) * 1e-5 : 0.0;
    static const bool joblog = getenv("WAVES_AMD_JOBLOG") != nullptr;  // diagnostic: the device clock of every job
    if (joblog) {
        static unsigned long long prev_end = 0;
        fprintf(stderr, "[waves_amd job] seq %u begin %llu end %llu: %.2f us, %.2f us after the previous job's end\n", want, t0, t1,
                (double)(t1 - t0) * 0.01, prev_end ? (double)((long long)(t0 - prev_end)) * 0.01 : 0.0);
        prev_end = t1;
        const unsigned long long *ph = pl->back->phase[par];
        fprintf(stderr, "    phases (us after begin): go seen by all %.2f | state loaded %.2f | steps done %.2f | stores drained %.2f | "
                        "barrier B %.2f | end %.2f\n", (double)(long long)(ph[0] - t0) * 0.01, (double)(long long)(ph[1] - t0) * 0.01,
                (double)(long long)(ph[2] - t0) * 0.01, (double)(long long)(ph[4] - t0) * 0.01, (double)(long long)(ph[5] - t0) * 0.01,
                (double)(long long)(t1 - t0) * 0.01);
    }
    pl->last_job_ms = ms;
    pl->t_last_done = std::chrono::steady_clock::now();
    pl->have_last_done = true;
    if (pl->job_ms.size() < (size_t)1 << 16) pl->job_ms.push_back(ms);
    // a launch that ended with this job: take its stop event (it is microseconds away) so that its duration is known
    if (!pl->job_keep[slot] && pl->L[l].alive && !pl->L[l].stays && jobs_join(pl, l) != 0) return 1;
    return 0;
}

// after a give-up: the protocol is reset so that the context stays usable, and this context stays on the single-step
// kernels from now on: whatever kept a tile from running (another process's kernels on the same device, most likely) may
// well still be there at the next call
void fused_gave_up(FusedPlan *pl, hipStream_t s)
{
    (void)jobs_join(pl, pl->cur_l ^ 1);
    (void)jobs_join(pl, pl->cur_l);
    (void)hipMemsetAsync(pl->d_abort, 0, sizeof(int), s);
    // (rare and worth a line: from here on every call of this context pays for a launch per step)
    fprintf(stderr, "[waves_amd] a resident launch could not make progress (a tile waited in vain for a neighbour: is the device shared with "
                    "other kernels?); the call was run again by the single-step kernels, which this context keeps using\n");
    pl->use_resident = false;
    for (int k = 0; k < 2; ++k) {
        pl->job_seq[k] = 0;
        pl->abort_pending[k] = false;
    }
}

const int *fused_abort_src(const FusedPlan *p) { return p->d_abort; }
int *fused_abort_dst(FusedPlan *p, int slot) { return p->h_abort + slot; }

double fused_last_job_ms(const FusedPlan *p) { return p->last_job_ms; }
int fused_job_times(FusedPlan *p, double *ms, int cap)
{
    const int n = (int)std::min<size_t>(p->job_ms.size(), (size_t)std::max(cap, 0));
    for (int k = 0; k < n; ++k) ms[k] = p->job_ms[p->job_ms.size() - (size_t)n + (size_t)k];
    p->job_ms.clear();
    return n;
}
void fused_launch_stats(const FusedPlan *p, double *ms, int *jobs)
{
    *ms = p->last_launch_ms;
    *jobs = p->last_launch_jobs;
}
void fused_allow_persist(FusedPlan *p, bool allow) { p->allow_persist = allow; }
// A launch is waiting for jobs and leaves at least 16 block slots of the device free: a small kernel on another stream
// finds room beside it (WAVES_AMD_OBS_BESIDE=0: never)
bool fused_obs_beside_launch(FusedPlan *p)
{
    const char *e = getenv("WAVES_AMD_OBS_BESIDE");  // (read per call: tests switch it)
    if ((e && atoi(e) == 0) || !fused_persist_alive(p)) return false;
    return device_slots(p) - p->L[p->cur_l].ntiles >= 16;
}

static int fused_run_steps(FusedPlan *pl, int slot, const FusedCall &call, const FusedStep *steps, int nsteps, hipStream_t s, hipEvent_t ev_start);

int fused_run(FusedPlan *pl, int slot, const FusedCall &call, const FusedStep *steps, int nsteps, hipStream_t s, hipStream_t up,
              const FusedEnergy &ef, bool keep, hipEvent_t ev_start, hipEvent_t ev_stop, const FusedDevTables *dev, const FusedObs *ob)
{
    const int rr = fused_try_resident(pl, slot, call, steps, nsteps, s, up, ef, keep, dev, ob);
    if (dev && rr != 0) return rr > 0 ? rr : 1;  // (a call without host tables has no other way to run)
    pl->cur = slot;
    pl->last_resident = rr == 0;
    if (rr >= 0) return rr;
    const int rc = fused_run_steps(pl, slot, call, steps, nsteps, s, ev_start);
    if (ev_stop && hipEventRecord(ev_stop, s) != hipSuccess) return 1;
    return rc;
}

// the same call once more with the single-step kernels (after fused_job_wait returned 2)
int fused_rerun_steps(FusedPlan *pl, int slot, const FusedCall &call, const FusedStep *steps, int nsteps, hipStream_t s,
                      hipEvent_t ev_start, hipEvent_t ev_stop)
{
    pl->cur = slot;
    const int rc = fused_run_steps(pl, slot, call, steps, nsteps, s, ev_start);
    if (ev_stop && hipEventRecord(ev_stop, s) != hipSuccess) return 1;
    return rc;
}

// the single-step kernels of a call (one launch per step, or the cached hipGraph of them), behind ev_start when given
static int fused_run_steps(FusedPlan *pl, int slot, const FusedCall &call, const FusedStep *steps, int nsteps, hipStream_t s, hipEvent_t ev_start)
{
    if (ev_start && hipEventRecord(ev_start, s) != hipSuccess) return 1;
    if (!pl->use_graph || pl->stamps_path) {
        for (int i = 0; i < nsteps; ++i) fused_launch(pl, slot, call, i, steps[i], s);
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
    // the graph's kernel arguments are functions of exactly these values: rebuild only when one of them changes
    std::vector<char> key;
    key_put(key, pl->generation);
    key_put(key, nsteps);
    key_put(key, call);
    key_put(key, pl->d_table);
    key_put(key, pl->M);
    key_put(key, pl->d_tiles[slot]);
    key_put(key, pl->d_idx[slot]);
    key_put(key, pl->d_src_flags);
    key_put(key, pl->nbands);
    for (int i = 0; i < nsteps; ++i) key_put(key, steps[i]);
    if (!pl->graph_exec[slot] || key != pl->graph_key[slot]) {
        if (getenv("WAVES_AMD_HOSTPROF")) fprintf(stderr, "[waves_amd] (re)building the step graph of plan %p\n", (void *)pl);
        if (pl->graph_exec[slot]) (void)hipGraphExecDestroy(pl->graph_exec[slot]);
        if (pl->graph[slot]) (void)hipGraphDestroy(pl->graph[slot]);
        pl->graph_exec[slot] = nullptr;
        pl->graph[slot] = nullptr;
        pl->graph_key[slot].clear();
        if (hipGraphCreate(&pl->graph[slot], 0) != hipSuccess) return 1;
        const int B = (int)pl->hp.band_begin.size() - 1;
        std::vector<hipGraphNode_t> prev(B), cur(B);
        const void *fn = kernel_ptr(pl);
        for (int i = 0; i < nsteps; ++i) {
            for (int b = 0; b < B; ++b) {
                FusedParams p = make_params(pl, call, i, steps[i]);
                p.tile_offset = pl->hp.band_begin[b];
                void *args[1] = {&p};
                hipKernelNodeParams kp{};
                kp.func = const_cast<void *>(fn);
                kp.gridDim = dim3((unsigned)(pl->hp.band_begin[b + 1] - pl->hp.band_begin[b]));
                kp.blockDim = dim3(512);
                kp.sharedMemBytes = 0;
                kp.kernelParams = args;
                kp.extra = nullptr;
                // a band's tiles read halo cells of the adjacent bands only, and overwrite a buffer whose last readers
                // were those same three kernels of the previous step
                hipGraphNode_t deps[3];
                int nd = 0;
                if (i > 0)
                    for (int d = b - 1; d <= b + 1; ++d)
                        if (d >= 0 && d < B) deps[nd++] = prev[d];
                if (hipGraphAddKernelNode(&cur[b], pl->graph[slot], nd ? deps : nullptr, nd, &kp) != hipSuccess) return 1;
            }
            prev = cur;
        }
        if (hipGraphInstantiate(&pl->graph_exec[slot], pl->graph[slot], nullptr, nullptr, 0) != hipSuccess) {
            pl->graph_exec[slot] = nullptr;
            return 1;
        }
        pl->graph_key[slot] = key;
    }
    return hipGraphLaunch(pl->graph_exec[slot], s) == hipSuccess ? 0 : 1;
}

int fused_generation(const FusedPlan *p) { return p->generation; }
bool fused_reduced(const FusedPlan *p) { return p->tiles_aux_zero; }
void fused_allow_resident(FusedPlan *p, bool allow) { p->allow_resident = allow; }
bool fused_last_resident(const FusedPlan *p) { return p->last_resident; }

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
