// C-ABI host layer of libwaves_amd.so (include/waves_amd.h).  Owns the device memory of one environment, builds the
// per-stage coefficient tables on the host exactly as the reference's closures would evaluate them, and enqueues the
// integrator kernels on the ctx's HIP stream.  No CPU compute fallback exists: every numerical result comes from a
// gfx950 kernel.
#include "../../include/waves_amd.h"

#include <math.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "fused.h"
#include "kernels.h"

using namespace wv;

struct wv_ctx {
    wv_config cfg{};
    int nx = 0, ny = 0;
    size_t P = 0, N = 0;  // plane, state (12 planes)
    hipStream_t own_stream = nullptr, stream = nullptr;

    std::vector<float> x, y, sx, sy;
    float dOmega = 0.0f;
    Grid grid{};

    float *d_x = nullptr, *d_y = nullptr, *d_sx = nullptr, *d_sy = nullptr;
    float *d_frames = nullptr;           // env.wave: 3 states ...
    float *d_f2alt = nullptr;            // ... whose LAST one lives alternately there ("home") and here: a call reads its initial
    int cur2 = 0;                        // condition from one and writes its final state to the other (0: home, 1: d_f2alt), so
                                         // that a call the resident kernel gives up can simply be run again
    float *d_scratch[2] = {nullptr, nullptr};
    float *d_yA = nullptr, *d_yB = nullptr, *d_acc = nullptr;  // staged implementation only
    float *d_G = nullptr;                // source shape
    bool has_source = false;
    float freq = 0.0f;
    float *d_plane[2] = {nullptr, nullptr};

    // DesignInterpolator(initial, final, ti, tf), cylinders stacked: px, py, r, c per cylinder
    int M = 0;
    std::vector<float> d0, d1;
    float ti = 0.0f, tf = 0.0f;
    // wv_set_design_sequence: the designs of n actions integrated by the NEXT call (consumed by it)
    int seq_n = 0, seq_steps = 0;
    std::vector<float> seq_d;   // (seq_n + 1) x M x {px, py, r, c}
    std::vector<float> seq_t;   // seq_n x {ti, tf}

    Cyl *d_cyl = nullptr;      // one stage time: wv_speed_field / wv_rhs
    size_t cyl_cap = 0;
    std::vector<Cyl> h_cyl;
    float *d_traj = nullptr;
    int traj_stride = 1;       // wv_set_trajectory_stride: every traj_stride-th saved time goes to u_tot / u_inc
    size_t traj_cap = 0;
    float *d_small = nullptr;  // gaussian parameters
    size_t small_cap = 0;
    float *d_obs = nullptr;    // wv_observation output
    size_t obs_cap = 0;
    float *h_obs = nullptr;    // ... in pinned host memory, written by the kernel itself when it runs beside a waiting launch
    size_t h_obs_cap = 0;
    // state(env) produced by the resident job itself (FusedParams::ob_out): once wv_observation has been asked for a resolution,
    // the following calls deliver the observation of the frames they leave into their slot's pinned buffer
    int obs_auto_rx = 0, obs_auto_ry = 0;  // the resolution to produce (0: nobody has asked, or nobody has used the last ones)
    int obs_slot = -1;                     // slot whose h_obs holds state(env) of the CURRENT frames, or -1
    int obs_unused = 0;                    // observations produced since the last one that was asked for
    // capture_frames == 2 of a design sequence: the three frames of every action but the last (those are env.wave itself)
    float *d_seq_frames = nullptr;
    size_t seq_frames_cap = 0;
    int seq_frames_actions = 0;     // actions of the last such call (0: none): wv_observation_action / wv_get_frames_action
    bool seq_frames_clean = false;  // the buffer holds zeros wherever a reduced-field-set tile does not write
    std::vector<FusedStep> fsteps;
    bool counted = false;  // this ctx is included in g_live_ctx
    int elast_generation = -1;
    bool elast_valid = false;  // elast still describes the current state (so row 1 of a step == last row of the previous)
    const float *elast = nullptr;  // per-block energy partials of the state the last integrate ended on (inside a slot's d_epart)
    int elast_blocks = 0;
    bool frames_exposed = false;  // wv_device_frames handed the state out: the caller may write it at any time

    // Up to two integrate calls are in flight (the host prepares call k+1 while call k runs on the device): everything a
    // call's device work reads or writes that differs from call to call exists once per slot; calls alternate slots.
    struct Slot {
        Cyl *d_cyl = nullptr, *h_cyl = nullptr;      // [nsteps][3][M] cylinders at the stage times (device / pinned staging)
        size_t cyl_cap = 0;
        float *d_sfac = nullptr, *h_sfac = nullptr;  // [nsteps][3] source time factors
        size_t sfac_cap = 0;
        float *d_epart = nullptr;                    // [nsteps + 1][nblocks][3] energy partials
        size_t epart_cap = 0;
        float *h_signal = nullptr;                   // pinned: the device writes the energy trace straight into it
        size_t signal_cap = 0;
        float *h_traj = nullptr;                     // the pinned buffer (one of wv_ctx::h_stream) this call's planes are copied to
        float *d_traj = nullptr;                     // streamed calls: the slot's own device planes (the other slot's are being copied)
        size_t traj_cap = 0;
        hipEvent_t copy_ev = nullptr;                // the copy of the planes to h_traj has finished
        hipEvent_t ev1 = nullptr;                    // after the last device work of the call
        float *h_obs = nullptr;                      // pinned: state(env) of the frames this call leaves, written by its job
        size_t h_obs_cap = 0;
        int obs_rx = 0, obs_ry = 0;                  // ... at this resolution (0: this call produces none)
        bool capture_all = false;                    // capture_frames == 2
        std::vector<hipEvent_t> kev;                 // kev[0] / kev[1] bracket the integrator launch(es); more when profiling
        // what running the call once more needs (after the resident kernel gave it up)
        std::vector<FusedStep> fsteps;
        FusedCall fcall{};
        bool dev_mode = false;                       // the call ran without host-built cylinder tables (FusedDevTables) ...
        std::vector<float> in_tspan, in_d0, in_d1;   // ... these are its inputs
        float in_ti = 0.0f, in_tf = 0.0f;
        int in_M = 0, in_rows[2] = {-1, -1};
        bool in_capture = false;
        const float *row0 = nullptr;
        int nblocks = 0;
        bool pending = false;
        int nsteps = 0, planes = 0, impl = 0;
        bool want_signal = false, want_fields = false, streamed = false, bracketed = false, resident = false, gave_up = false;
        int prof_launches = 0, prof_events = 0;
    } slot[2];
    int next_slot = 0;   // slot of the next wv_integrate_begin
    int n_pending = 0;   // calls begun and not ended (the oldest is slot[(next_slot + 2 - n_pending) % 2])
    // streamed trajectories (want_fields == 2): three pinned buffers used in turn, so that the planes of a call stay
    // readable while the next two calls are begun (two may be in flight)
    float *h_stream[3] = {nullptr, nullptr, nullptr};
    size_t stream_cap[3] = {0, 0, 0};
    unsigned stream_seq = 0;
    const float *last_view_tot = nullptr, *last_view_inc = nullptr;  // streamed trajectories of the call ended last
    int last_view_planes = 0;
    hipStream_t up_stream = nullptr;  // uploads of the per-call tables, overlapped with the previous call's kernels
    hipEvent_t obs_ev = nullptr;        // wv_observation beside a waiting resident launch (on up_stream)
    hipStream_t down_stream = nullptr;  // device-to-host copies of streamed trajectories (its own stream: the next call's
                                        // kernels wait for everything on up_stream, and must not wait for these)
    hipEvent_t up_ev = nullptr;

    FusedPlan *fused = nullptr;

    bool profiling = false;
    wv_timing timing{};

    std::string err;
};

static thread_local std::string g_err;

static int fail(wv_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_err = msg;
    return code;
}

#define HIPCHK(c, expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail((c), WV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

#define CHECK_CTX(c)                                   \
    do {                                               \
        if (!(c)) return fail(nullptr, WV_ERR_INVALID, "ctx is NULL"); \
        HIPCHK((c), hipSetDevice((c)->cfg.device));    \
    } while (0)

// Entry points that use the context's stream (or let the caller touch the device state) first make a resident launch that
// is waiting for further wv_integrate calls leave: work enqueued behind it would wait out its idle limit.
#define QUIET(c)                                                                                                  \
    do {                                                                                                          \
        if ((c)->fused && fused_retire((c)->fused) != 0) return fail((c), WV_ERR_HIP, "the resident launch did not leave"); \
    } while (0)

// ---- host-side restatement of the closures' scalar work (fp32, reference order, no FMA) -----------------------

// src/operators.jl:10-22
static Ops make_ops(const std::vector<float> &x)
{
    const int n = (int)x.size();
    const float delta = (x[n - 1] - x[0]) / (float)(n - 1);
    const float two_d = 2.0f * delta;
    Ops o;
    o.cm = -1.0f / two_d;
    o.cp = 1.0f / two_d;
    o.f0 = -3.0f / two_d; o.f1 = 4.0f / two_d; o.f2 = -1.0f / two_d;
    o.b0 = 1.0f / two_d;  o.b1 = -4.0f / two_d; o.b2 = 3.0f / two_d;
    return o;
}

// src/pml.jl:21-29 (1-D profile of the field that `repeat` tiles along y)
static std::vector<float> make_pml(const std::vector<float> &xs, float width, float scale)
{
    const size_t n = xs.size();
    std::vector<float> out(n);
    const float pml_start = fabsf(xs[0]) - width;
    float mn = INFINITY;
    for (size_t i = 0; i < n; ++i) {
        const float a = fabsf(xs[i]);
        if (a > pml_start && a < mn) mn = a;
    }
    for (size_t i = 0; i < n; ++i) {
        const float a = fabsf(xs[i]);
        float v = 0.0f;
        if (a > pml_start) v = (a - mn) / width;
        out[i] = ((v * v) * v) * scale;
    }
    return out;
}

// Flux.mean(diff(x)), src/dims.jl:126-127 (double accumulation, rounded once)
static float mean_diff(const std::vector<float> &x)
{
    double s = 0.0;
    for (size_t i = 1; i < x.size(); ++i) s += (double)(x[i] - x[i - 1]);
    return (float)(s / (double)(x.size() - 1));
}

// DesignInterpolator call, src/designs.jl:287-292 with the algebra of :47-53: per scalar component
//   v_i + ((v_f + (-1f0*v_i)) * (1f0/Dt)) * (clamp(t, ti, tf) - ti)
static void design_at(int M, const float *d0, const float *d1, float ti, float tf, float t, Cyl *out)
{
    float dt = tf - ti;
    dt = dt > 0.0f ? dt : 1.0f;
    const float inv_dt = 1.0f / dt;
    const float tc = t < ti ? ti : (t > tf ? tf : t);
    const float tau = tc - ti;
    for (int m = 0; m < M; ++m) {
        float v[4];
        for (int k = 0; k < 4; ++k) {
            const float vi = d0[4 * m + k], vf = d1[4 * m + k];
            const float dy = vf + (-1.0f * vi);
            v[k] = vi + (dy * inv_dt) * tau;
        }
        out[m].px = v[0];
        out[m].py = v[1];
        out[m].r2 = v[2] * v[2];  // `r .^ 2`, src/designs.jl:102
        out[m].c = v[3];
    }
}
static void design_at(const wv_ctx *c, float t, Cyl *out) { design_at(c->M, c->d0.data(), c->d1.data(), c->ti, c->tf, t, out); }

// sin.(2.0f0 * pi * t * freq): ((2f0*pi)*t)*freq in fp32, accurately rounded sin.  src/sources.jl:21-22,67-69
static float source_factor(float t, float freq)
{
    const float two_pi = 6.2831855f;
    const float arg = (two_pi * t) * freq;
    return (float)sin((double)arg);
}

template <class T>
static int ensure(wv_ctx *c, T **p, size_t *cap, size_t need)
{
    if (need <= *cap) return WV_OK;
    if (*p) HIPCHK(c, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    HIPCHK(c, hipMalloc((void **)p, need * sizeof(T)));
    *cap = need;
    return WV_OK;
}

// device buffer + pinned staging buffer of `need` elements
template <class T>
static int ensure_pinned(wv_ctx *c, T **d, T **h, size_t *cap, size_t need)
{
    if (need <= *cap) return WV_OK;
    if (*d) HIPCHK(c, hipFree(*d));
    if (*h) HIPCHK(c, hipHostFree(*h));
    *d = nullptr;
    *h = nullptr;
    *cap = 0;
    HIPCHK(c, hipMalloc((void **)d, need * sizeof(T)));
    HIPCHK(c, hipHostMalloc((void **)h, need * sizeof(T), hipHostMallocDefault));
    *cap = need;
    return WV_OK;
}

static int wait_event(wv_ctx *c, hipEvent_t ev);
static float *frame(wv_ctx *c, int k) { return (k == 2 && c->cur2) ? c->d_f2alt : c->d_frames + (size_t)k * c->N; }
static float *other2(wv_ctx *c) { return c->cur2 ? c->d_frames + 2 * c->N : c->d_f2alt; }  // where the next call's final state goes
// env.wave as ONE array again (raw-pointer access, wv_set_frames): the last frame back to its home
static int frames_home(wv_ctx *c)
{
    if (c->cur2) {
        hipError_t e = hipMemcpyAsync(c->d_frames + 2 * c->N, c->d_f2alt, c->N * sizeof(float), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            c->err = std::string("frames_home: ") + hipGetErrorString(e);
            return WV_ERR_HIP;
        }
        c->cur2 = 0;
        fused_state_changed(c->fused);  // (which buffer holds what has changed: look again)
        c->elast_valid = false;
    }
    return WV_OK;
}

// ---- ABI -------------------------------------------------------------------------------------------------

extern "C" {

int wv_abi_version(void) { return WV_ABI_VERSION; }

const char *wv_last_error(const wv_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int wv_device_count(int *count)
{
    if (!count) return fail(nullptr, WV_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(nullptr, WV_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return WV_OK;
}

// Contexts alive per device.  The resident step kernel owns the whole device for the duration of a call (cooperative
// launch, every tile resident), which is the fastest way to run ONE environment; several environments stepped
// concurrently on one GPU are better served by the single-step kernels, whose launches interleave on the streams
// (measured at 700^2, 4 envs: 48 vs 22-37 Gcell-updates/s) -- so the resident path is taken only by a context that has
// the device to itself within this process.
static std::atomic<int> g_live_ctx[64];

int wv_destroy(wv_ctx *c)
{
    if (!c) return WV_OK;
    if (c->counted) g_live_ctx[c->cfg.device & 63]--;
    (void)hipSetDevice(c->cfg.device);
    if (c->fused) (void)fused_retire(c->fused);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    float *bufs[] = {c->d_x, c->d_y, c->d_sx, c->d_sy, c->d_frames, c->d_f2alt, c->d_scratch[0], c->d_scratch[1], c->d_yA, c->d_yB,
                     c->d_acc, c->d_G, c->d_plane[0], c->d_plane[1], c->d_traj, c->d_small, c->d_obs, c->d_seq_frames};
    for (float *b : bufs)
        if (b) (void)hipFree(b);
    if (c->d_cyl) (void)hipFree(c->d_cyl);
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    if (c->down_stream) (void)hipStreamSynchronize(c->down_stream);
    for (wv_ctx::Slot &q : c->slot) {
        if (q.d_cyl) (void)hipFree(q.d_cyl);
        if (q.h_cyl) (void)hipHostFree(q.h_cyl);
        if (q.d_sfac) (void)hipFree(q.d_sfac);
        if (q.h_sfac) (void)hipHostFree(q.h_sfac);
        if (q.d_epart) (void)hipFree(q.d_epart);
        if (q.h_signal) (void)hipHostFree(q.h_signal);
        if (q.d_traj) (void)hipFree(q.d_traj);
        if (q.copy_ev) (void)hipEventDestroy(q.copy_ev);
        for (hipEvent_t e : q.kev) (void)hipEventDestroy(e);
        if (q.ev1) (void)hipEventDestroy(q.ev1);
    }
    for (float *b : c->h_stream)
        if (b) (void)hipHostFree(b);
    if (c->h_obs) (void)hipHostFree(c->h_obs);
    for (wv_ctx::Slot &q : c->slot)
        if (q.h_obs) (void)hipHostFree(q.h_obs);
    if (c->fused) fused_destroy(c->fused);
    if (c->up_ev) (void)hipEventDestroy(c->up_ev);
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
    if (c->obs_ev) (void)hipEventDestroy(c->obs_ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return WV_OK;
}

int wv_create(const wv_config *cfg, const float *x, const float *y, wv_ctx **out)
{
    if (!cfg || !x || !y || !out) return fail(nullptr, WV_ERR_INVALID, "wv_create: NULL argument");
    *out = nullptr;
    if (cfg->nx < 8 || cfg->ny < 8) return fail(nullptr, WV_ERR_INVALID, "wv_create: nx and ny must be >= 8");
    if (cfg->nx != cfg->ny)
        return fail(nullptr, WV_ERR_INVALID,
                    "wv_create: nx must equal ny (the reference uses the x gradient matrix and the transposed x PML "
                    "profile for y: src/dynamics.jl:146,161-162)");
    if (!(cfg->dt > 0.0f)) return fail(nullptr, WV_ERR_INVALID, "wv_create: dt must be > 0");
    if ((size_t)cfg->nx * cfg->ny * kFields >= ((size_t)1 << 31))
        return fail(nullptr, WV_ERR_INVALID, "wv_create: 12*nx*ny must be < 2^31 (32-bit element offsets in the kernels)");
    if (cfg->impl < WV_IMPL_AUTO || cfg->impl > WV_IMPL_FUSED) return fail(nullptr, WV_ERR_INVALID, "wv_create: bad impl");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, WV_ERR_NO_DEVICE,
                    std::string("wv_create: no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0") +
                        "); libwaves_amd has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, WV_ERR_INVALID, "wv_create: device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(nullptr, WV_ERR_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, WV_ERR_NO_DEVICE,
                    std::string("wv_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");

    wv_ctx *c = new (std::nothrow) wv_ctx();
    if (!c) return fail(nullptr, WV_ERR_NOMEM, "wv_create: out of host memory");
    c->cfg = *cfg;
    c->nx = cfg->nx;
    c->ny = cfg->ny;
    c->P = (size_t)cfg->nx * cfg->ny;
    c->N = c->P * kFields;
    c->x.assign(x, x + c->nx);
    c->y.assign(y, y + c->ny);
    c->sx = make_pml(c->x, cfg->pml_width, cfg->pml_scale);
    c->sy = c->sx;  // sigma_y = sigma_x' : the x profile indexed by j
    c->dOmega = mean_diff(c->x) * mean_diff(c->y);

#define CK(expr)                                                                                         \
    do {                                                                                                 \
        hipError_t e2_ = (expr);                                                                         \
        if (e2_ != hipSuccess) {                                                                         \
            std::string m_ = std::string(#expr) + ": " + hipGetErrorString(e2_);                         \
            wv_destroy(c);                                                                               \
            return fail(nullptr, e2_ == hipErrorOutOfMemory ? WV_ERR_NOMEM : WV_ERR_HIP, m_);            \
        }                                                                                                \
    } while (0)

    CK(hipSetDevice(cfg->device));
    CK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    CK(hipEventCreateWithFlags(&c->up_ev, hipEventDisableTiming));
    CK(hipEventCreate(&c->slot[0].ev1));
    CK(hipEventCreate(&c->slot[1].ev1));
    CK(hipEventCreateWithFlags(&c->slot[0].copy_ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&c->slot[1].copy_ev, hipEventDisableTiming));
    CK(hipMalloc((void **)&c->d_x, c->nx * sizeof(float)));
    CK(hipMalloc((void **)&c->d_y, c->ny * sizeof(float)));
    CK(hipMalloc((void **)&c->d_sx, c->nx * sizeof(float)));
    CK(hipMalloc((void **)&c->d_sy, c->ny * sizeof(float)));
    CK(hipMalloc((void **)&c->d_frames, 3 * c->N * sizeof(float)));
    CK(hipMalloc((void **)&c->d_f2alt, c->N * sizeof(float)));
    CK(hipMalloc((void **)&c->d_scratch[0], c->N * sizeof(float)));
    CK(hipMalloc((void **)&c->d_scratch[1], c->N * sizeof(float)));
    CK(hipMalloc((void **)&c->d_yA, c->N * sizeof(float)));
    CK(hipMalloc((void **)&c->d_yB, c->N * sizeof(float)));
    CK(hipMalloc((void **)&c->d_acc, c->N * sizeof(float)));
    CK(hipMalloc((void **)&c->d_G, c->P * sizeof(float)));
    CK(hipMalloc((void **)&c->d_plane[0], c->P * sizeof(float)));
    CK(hipMalloc((void **)&c->d_plane[1], c->P * sizeof(float)));
    CK(hipMemcpy(c->d_x, c->x.data(), c->nx * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(c->d_y, c->y.data(), c->ny * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(c->d_sx, c->sx.data(), c->nx * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(c->d_sy, c->sy.data(), c->ny * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemset(c->d_frames, 0, 3 * c->N * sizeof(float)));
    CK(hipMemset(c->d_f2alt, 0, c->N * sizeof(float)));
    // the reduced field sets of the fused kernel rely on unwritten planes of every output buffer holding zeros
    CK(hipMemset(c->d_scratch[0], 0, c->N * sizeof(float)));
    CK(hipMemset(c->d_scratch[1], 0, c->N * sizeof(float)));
    CK(hipMemset(c->d_G, 0, c->P * sizeof(float)));
#undef CK

    Grid &g = c->grid;
    g.nx = c->nx;
    g.ny = c->ny;
    g.P = c->P;
    g.ops = make_ops(c->x);
    g.x = c->d_x;
    g.y = c->d_y;
    g.sx = c->d_sx;
    g.sy = c->d_sy;
    g.c0 = cfg->c0;
    g.c0sq = cfg->c0 * cfg->c0;

    c->fused = fused_create(g, c->x.data(), c->y.data(), c->sx.data(), c->sy.data());
    if (!c->fused) {
        wv_destroy(c);
        return fail(nullptr, WV_ERR_HIP, "wv_create: fused plan allocation failed");
    }
    c->counted = true;
    g_live_ctx[cfg->device & 63]++;
    *out = c;
    return WV_OK;
}

int wv_get_pml(wv_ctx *c, float *sigma_x, float *sigma_y)
{
    CHECK_CTX(c);
    if (sigma_x) memcpy(sigma_x, c->sx.data(), c->nx * sizeof(float));
    if (sigma_y) memcpy(sigma_y, c->sy.data(), c->ny * sizeof(float));
    return WV_OK;
}

int wv_set_pml(wv_ctx *c, const float *sigma_x, const float *sigma_y)
{
    CHECK_CTX(c);
    QUIET(c);
    if (!sigma_x || !sigma_y) return fail(c, WV_ERR_INVALID, "wv_set_pml: NULL profile");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_set_pml: an integrate is pending");
    c->sx.assign(sigma_x, sigma_x + c->nx);
    c->sy.assign(sigma_y, sigma_y + c->ny);
    HIPCHK(c, hipMemcpy(c->d_sx, c->sx.data(), c->nx * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_sy, c->sy.data(), c->ny * sizeof(float), hipMemcpyHostToDevice));
    fused_set_pml(c->fused, c->sx.data(), c->sy.data());
    return WV_OK;
}

int wv_get_cell_area(wv_ctx *c, float *dOmega)
{
    CHECK_CTX(c);
    if (!dOmega) return fail(c, WV_ERR_INVALID, "dOmega is NULL");
    *dOmega = c->dOmega;
    return WV_OK;
}

int wv_set_frames(wv_ctx *c, const float *wave)
{
    CHECK_CTX(c);
    c->obs_slot = -1;  // (what state(env) shows changes)
    QUIET(c);
    if (!wave) return fail(c, WV_ERR_INVALID, "wv_set_frames: NULL");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_set_frames: an integrate is pending");
    c->cur2 = 0;
    HIPCHK(c, hipMemcpyAsync(c->d_frames, wave, 3 * c->N * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fused_state_changed(c->fused);
    c->elast_valid = false;
    return WV_OK;
}

int wv_get_frames(wv_ctx *c, float *wave)
{
    CHECK_CTX(c);
    QUIET(c);
    if (!wave) return fail(c, WV_ERR_INVALID, "wv_get_frames: NULL");
    HIPCHK(c, hipMemcpyAsync(wave, c->d_frames, 2 * c->N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(wave + 2 * c->N, frame(c, 2), c->N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WV_OK;
}

int wv_set_state(wv_ctx *c, const float *u)
{
    CHECK_CTX(c);
    c->obs_slot = -1;  // (what state(env) shows changes)
    QUIET(c);
    if (!u) return fail(c, WV_ERR_INVALID, "wv_set_state: NULL");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_set_state: an integrate is pending");
    HIPCHK(c, hipMemcpyAsync(frame(c, 2), u, c->N * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fused_state_changed(c->fused);
    c->elast_valid = false;
    return WV_OK;
}

int wv_get_state(wv_ctx *c, float *u)
{
    CHECK_CTX(c);
    QUIET(c);
    if (!u) return fail(c, WV_ERR_INVALID, "wv_get_state: NULL");
    HIPCHK(c, hipMemcpyAsync(u, frame(c, 2), c->N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WV_OK;
}

int wv_reset(wv_ctx *c)
{
    CHECK_CTX(c);
    c->obs_slot = -1;  // (what state(env) shows changes)
    QUIET(c);
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_reset: an integrate is pending");
    c->cur2 = 0;
    HIPCHK(c, hipMemsetAsync(c->d_frames, 0, 3 * c->N * sizeof(float), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fused_state_zeroed(c->fused);
    c->elast_valid = false;
    return WV_OK;
}

int wv_set_source_shape(wv_ctx *c, const float *shape, float freq)
{
    CHECK_CTX(c);
    c->obs_slot = -1;  // (what state(env) shows changes)
    QUIET(c);
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_set_source_shape: an integrate is pending");
    c->has_source = shape != nullptr;
    c->freq = freq;
    if (shape) {
        HIPCHK(c, hipMemcpyAsync(c->d_G, shape, c->P * sizeof(float), hipMemcpyHostToDevice, c->stream));
    } else {
        HIPCHK(c, hipMemsetAsync(c->d_G, 0, c->P * sizeof(float), c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fused_source_changed(c->fused);
    return WV_OK;
}

int wv_set_gaussian_source(wv_ctx *c, int K, const float *mu, const float *sigma, const float *a, float freq)
{
    CHECK_CTX(c);
    c->obs_slot = -1;  // (what state(env) shows changes)
    QUIET(c);
    if (K < 1 || !mu || !sigma || !a) return fail(c, WV_ERR_INVALID, "wv_set_gaussian_source: bad arguments");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_set_gaussian_source: an integrate is pending");
    int rc = ensure(c, &c->d_small, &c->small_cap, (size_t)4 * K);
    if (rc) return rc;
    std::vector<float> h(4 * (size_t)K);
    memcpy(h.data(), mu, 2 * K * sizeof(float));
    memcpy(h.data() + 2 * K, sigma, K * sizeof(float));
    memcpy(h.data() + 3 * K, a, K * sizeof(float));
    HIPCHK(c, hipMemcpy(c->d_small, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    launch_gaussian(c->grid, K, c->d_small, c->d_small + 2 * K, c->d_small + 3 * K, c->d_G, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->has_source = true;
    c->freq = freq;
    fused_source_changed(c->fused);
    return WV_OK;
}

int wv_get_source_shape(wv_ctx *c, float *shape)
{
    CHECK_CTX(c);
    QUIET(c);
    if (!shape) return fail(c, WV_ERR_INVALID, "wv_get_source_shape: NULL");
    HIPCHK(c, hipMemcpyAsync(shape, c->d_G, c->P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WV_OK;
}

int wv_observation(wv_ctx *c, int rx, int ry, float *out)
{
    CHECK_CTX(c);
    if (!out) return fail(c, WV_ERR_INVALID, "wv_observation: NULL");
    if (rx < 1 || ry < 1 || rx > c->nx || ry > c->ny)
        return fail(c, WV_ERR_INVALID, "wv_observation: resolution must be within 1 .. grid size (src/env.jl:52)");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_observation: an integrate is pending");
    const size_t n = (size_t)rx * ry * 4;
    // The call that ended last has produced this very observation itself (k_steps_resident, FusedParams::ob_out): it is in
    // pinned memory already.
    const char *oij = getenv("WAVES_AMD_OBS_IN_JOB");  // (read per call: tests switch it)
    const bool obs_in_job = !(oij && atoi(oij) == 0);
    if (c->obs_slot >= 0 && !c->frames_exposed && c->slot[c->obs_slot].obs_rx == rx && c->slot[c->obs_slot].obs_ry == ry) {
        memcpy(out, c->slot[c->obs_slot].h_obs, n * sizeof(float));
        c->obs_unused = 0;
        return WV_OK;
    }
    if (obs_in_job) {  // from now on the calls produce it (until eight in a row have gone unused)
        c->obs_auto_rx = rx;
        c->obs_auto_ry = ry;
        c->obs_unused = 0;
    }
    // state(env) in front of every action (src/data.jl:23, scripts/mpc.jl:83-85) must not cost the rollout its resident launch:
    // while one waits on the context's stream -- every call it was given has been ended, so the frames are complete in
    // memory -- the resize kernel runs on a stream of its own, on block slots the launch leaves free (it never takes them
    // all: fused_obs_beside_launch).  Otherwise the launch is asked to leave first, like for every other user of the stream.
    // (on the copy stream, idle at this point: one more stream of its own pushed the process over its hardware queues and
    // cost every action 0.3 ms)
    const bool beside = c->fused && n <= c->h_obs_cap && c->up_stream && c->obs_ev && fused_obs_beside_launch(c->fused);
    if (beside) {
        // (the kernel writes the 256 KB straight into pinned host memory: a copy on another stream beside the spinning launch
        // takes the runtime hundreds of microseconds, measured)
        launch_observation(c->grid, frame(c, 0), frame(c, 1), frame(c, 2), c->has_source ? c->d_G : nullptr, rx, ry, c->h_obs, c->up_stream);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(c->obs_ev, c->up_stream));
        const int rcw = wait_event(c, c->obs_ev);  // (polled: the wake-up of a blocking wait costs more than the kernel)
        if (rcw) return rcw;
        memcpy(out, c->h_obs, n * sizeof(float));
        return WV_OK;
    }
    QUIET(c);
    int rc = ensure(c, &c->d_obs, &c->obs_cap, n);  // (may free device memory: never beside a launch)
    if (rc) return rc;
    if (n > c->h_obs_cap) {  // ... and what the next call needs to run beside a launch
        if (c->h_obs) (void)hipHostFree(c->h_obs);
        c->h_obs = nullptr;
        c->h_obs_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_obs, n * sizeof(float), hipHostMallocDefault));
        c->h_obs_cap = n;
    }
    for (wv_ctx::Slot &q : c->slot) {  // (and the calls' own buffers: allocated here, where no launch is on the device)
        if (!obs_in_job || n <= q.h_obs_cap) continue;
        if (q.h_obs) (void)hipHostFree(q.h_obs);
        q.h_obs = nullptr;
        q.h_obs_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&q.h_obs, n * sizeof(float), hipHostMallocDefault));
        q.h_obs_cap = n;
    }
    if (!c->obs_ev) HIPCHK(c, hipEventCreateWithFlags(&c->obs_ev, hipEventDisableTiming));
    launch_observation(c->grid, frame(c, 0), frame(c, 1), frame(c, 2), c->has_source ? c->d_G : nullptr, rx, ry, c->d_obs, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->d_obs, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WV_OK;
}

// the three frames of action `action` of the last call that kept every action's frames (the last action's are env.wave)
static bool action_frames(wv_ctx *c, int action, const float *f[3])
{
    if (action < 0 || action >= c->seq_frames_actions) return false;
    if (action == c->seq_frames_actions - 1) {
        for (int k = 0; k < 3; ++k) f[k] = frame(c, k);
    } else {
        for (int k = 0; k < 3; ++k) f[k] = c->d_seq_frames + ((size_t)action * 3 + k) * c->N;
    }
    return true;
}

int wv_observation_action(wv_ctx *c, int action, int rx, int ry, float *out)
{
    CHECK_CTX(c);
    QUIET(c);
    if (!out) return fail(c, WV_ERR_INVALID, "wv_observation_action: NULL");
    if (rx < 1 || ry < 1 || rx > c->nx || ry > c->ny)
        return fail(c, WV_ERR_INVALID, "wv_observation_action: resolution must be within 1 .. grid size (src/env.jl:52)");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_observation_action: an integrate is pending");
    const float *f[3];
    if (!action_frames(c, action, f)) return fail(c, WV_ERR_INVALID, "wv_observation_action: no such action in the last sequence call with capture_frames == 2");
    const size_t n = (size_t)rx * ry * 4;
    int rc = ensure(c, &c->d_obs, &c->obs_cap, n);
    if (rc) return rc;
    launch_observation(c->grid, f[0], f[1], f[2], c->has_source ? c->d_G : nullptr, rx, ry, c->d_obs, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->d_obs, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WV_OK;
}

int wv_get_frames_action(wv_ctx *c, int action, float *wave)
{
    CHECK_CTX(c);
    QUIET(c);
    if (!wave) return fail(c, WV_ERR_INVALID, "wv_get_frames_action: NULL");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_get_frames_action: an integrate is pending");
    const float *f[3];
    if (!action_frames(c, action, f)) return fail(c, WV_ERR_INVALID, "wv_get_frames_action: no such action in the last sequence call with capture_frames == 2");
    for (int k = 0; k < 3; ++k) HIPCHK(c, hipMemcpyAsync(wave + (size_t)k * c->N, f[k], c->N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WV_OK;
}

int wv_set_design(wv_ctx *c, int M, const float *pos_i, const float *r_i, const float *c_i, const float *pos_f,
                  const float *r_f, const float *c_f, float ti, float tf)
{
    CHECK_CTX(c);
    if (M < 0 || M > 4096) return fail(c, WV_ERR_INVALID, "wv_set_design: M out of range [0, 4096]");
    if (M > 0 && (!pos_i || !r_i || !c_i || !pos_f || !r_f || !c_f)) return fail(c, WV_ERR_INVALID, "wv_set_design: NULL array");
    c->M = M;
    c->seq_n = 0;  // (a plain design replaces a pending sequence)
    c->ti = ti;
    c->tf = tf;
    c->d0.resize(4 * (size_t)M);
    c->d1.resize(4 * (size_t)M);
    for (int m = 0; m < M; ++m) {
        c->d0[4 * m + 0] = pos_i[m];
        c->d0[4 * m + 1] = pos_i[M + m];
        c->d0[4 * m + 2] = r_i[m];
        c->d0[4 * m + 3] = c_i[m];
        c->d1[4 * m + 0] = pos_f[m];
        c->d1[4 * m + 1] = pos_f[M + m];
        c->d1[4 * m + 2] = r_f[m];
        c->d1[4 * m + 3] = c_f[m];
    }
    return WV_OK;
}

int wv_set_design_sequence(wv_ctx *c, int n_actions, int steps_per_action, int M, const float *designs, const float *ti_tf)
{
    CHECK_CTX(c);
    if (n_actions < 1 || steps_per_action < 1) return fail(c, WV_ERR_INVALID, "wv_set_design_sequence: n_actions and steps_per_action must be >= 1");
    if (M < 1 || M > 4096) return fail(c, WV_ERR_INVALID, "wv_set_design_sequence: M out of range [1, 4096]");
    if (!designs || !ti_tf) return fail(c, WV_ERR_INVALID, "wv_set_design_sequence: NULL array");
    if ((long long)n_actions * steps_per_action > (1 << 20)) return fail(c, WV_ERR_INVALID, "wv_set_design_sequence: more than 2^20 steps");
    c->M = M;
    c->seq_n = n_actions;
    c->seq_steps = steps_per_action;
    c->seq_d.assign(designs, designs + (size_t)(n_actions + 1) * M * 4);
    c->seq_t.assign(ti_tf, ti_tf + 2 * (size_t)n_actions);
    // what a later call without a sequence sees: the last action's interpolator (as after n wv_set_design calls)
    c->d0.assign(c->seq_d.end() - 2 * (size_t)M * 4, c->seq_d.end() - (size_t)M * 4);
    c->d1.assign(c->seq_d.end() - (size_t)M * 4, c->seq_d.end());
    c->ti = c->seq_t[2 * (size_t)(n_actions - 1)];
    c->tf = c->seq_t[2 * (size_t)(n_actions - 1) + 1];
    return WV_OK;
}

int wv_speed_field(wv_ctx *c, float t, float *out)
{
    CHECK_CTX(c);
    QUIET(c);
    if (!out) return fail(c, WV_ERR_INVALID, "wv_speed_field: NULL");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_speed_field: an integrate is pending");
    int rc = ensure(c, &c->d_cyl, &c->cyl_cap, (size_t)(c->M > 0 ? c->M : 1));
    if (rc) return rc;
    c->h_cyl.resize(c->M > 0 ? c->M : 1);
    design_at(c, t, c->h_cyl.data());
    if (c->M > 0) HIPCHK(c, hipMemcpy(c->d_cyl, c->h_cyl.data(), c->M * sizeof(Cyl), hipMemcpyHostToDevice));
    launch_speed_field(c->grid, c->d_cyl, c->M, c->d_plane[0], c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->d_plane[0], c->P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WV_OK;
}

int wv_source_field(wv_ctx *c, float t, float *out)
{
    CHECK_CTX(c);
    QUIET(c);
    if (!out) return fail(c, WV_ERR_INVALID, "wv_source_field: NULL");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_source_field: an integrate is pending");
    if (!c->has_source) {  // NoSource returns the scalar 0f0 (src/sources.jl:8)
        memset(out, 0, c->P * sizeof(float));
        return WV_OK;
    }
    launch_scale(c->d_G, source_factor(t, c->freq), c->d_plane[0], c->P, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->d_plane[0], c->P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WV_OK;
}

int wv_gradient(wv_ctx *c, int axis, const float *u, float *out)
{
    CHECK_CTX(c);
    QUIET(c);
    if (!u || !out || (axis != 0 && axis != 1)) return fail(c, WV_ERR_INVALID, "wv_gradient: bad arguments");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_gradient: an integrate is pending");
    HIPCHK(c, hipMemcpyAsync(c->d_plane[0], u, c->P * sizeof(float), hipMemcpyHostToDevice, c->stream));
    launch_gradient(c->grid, axis, c->d_plane[0], c->d_plane[1], c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->d_plane[1], c->P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WV_OK;
}

int wv_rhs(wv_ctx *c, const float *x, float t, float *k)
{
    CHECK_CTX(c);
    QUIET(c);
    if (!x || !k) return fail(c, WV_ERR_INVALID, "wv_rhs: NULL");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_rhs: an integrate is pending");
    int rc = ensure(c, &c->d_cyl, &c->cyl_cap, (size_t)(c->M > 0 ? c->M : 1));
    if (rc) return rc;
    c->h_cyl.resize(c->M > 0 ? c->M : 1);
    design_at(c, t, c->h_cyl.data());
    if (c->M > 0) HIPCHK(c, hipMemcpy(c->d_cyl, c->h_cyl.data(), c->M * sizeof(Cyl), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpyAsync(c->d_scratch[0], x, c->N * sizeof(float), hipMemcpyHostToDevice, c->stream));
    fused_scratch_dirty(c->fused);  // the reduced field sets rely on zero auxiliary planes in the scratch states
    StageIO io{};
    io.yin = c->d_scratch[0];
    io.u = c->d_scratch[0];
    io.acc = c->d_acc;
    io.out = c->d_scratch[1];
    io.G = c->has_source ? c->d_G : nullptr;
    io.sfac = c->has_source ? source_factor(t, c->freq) : 0.0f;
    io.cyl = c->d_cyl;
    io.M = c->M;
    io.a = 0.0f;
    io.dt = c->cfg.dt;
    launch_stage(c->grid, io, 4, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(k, c->d_scratch[1], c->N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WV_OK;
}

// diagnostic (WAVES_AMD_HOSTPROF=1): where the host time of wv_integrate_begin goes, printed at exit
namespace {
// (contexts may be driven from different threads: with the diagnostic off -- every normal run -- nothing here is written;
// with it on, the accumulators are shared under a mutex and the lap clock is the calling thread's own)
struct HostProf {
    const bool on = getenv("WAVES_AMD_HOSTPROF") != nullptr;
    const bool each = on && atoi(getenv("WAVES_AMD_HOSTPROF")) >= 2;  // =2: every call's sections as they happen
    std::mutex mu;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long calls = 0;
    bool warmed = false;
    static std::chrono::steady_clock::time_point &clock()
    {
        static thread_local std::chrono::steady_clock::time_point t;
        return t;
    }
    void start()
    {
        if (!on) return;
        std::lock_guard<std::mutex> g(mu);
        ++calls;
        if (!warmed && calls == 6) {  // the first calls allocate / load code: not representative
            for (double &a : acc) a = 0.0;
            calls = 1;
            warmed = true;
        }
        clock() = std::chrono::steady_clock::now();
    }
    void lap(int k)
    {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(n - clock()).count();
        {
            std::lock_guard<std::mutex> g(mu);
            acc[k] += us;
        }
        if (each) fprintf(stderr, "[waves_amd hostprof] call %ld section %d: %.1f us\n", calls, k, us);
        if (us > 3000.0) fprintf(stderr, "[waves_amd hostprof] section %d of wv_integrate_begin took %.1f ms\n", k, us / 1000.0);
        clock() = n;
    }
    ~HostProf()
    {
        if (on && calls)
            fprintf(stderr, "[waves_amd hostprof] per integrate_begin (us): tables %.1f | uploads %.1f | prepare(plan+cull) %.1f | "
                            "buffers %.1f | enqueue aux %.1f | launch %.1f | tail %.1f  (%ld calls)\n",
                    acc[0] / calls, acc[1] / calls, acc[2] / calls, acc[3] / calls, acc[4] / calls, acc[5] / calls, acc[6] / calls, calls);
    }
};
HostProf g_hostprof;
}  // namespace

int wv_integrate_begin(wv_ctx *c, const float *tspan, int nsteps, int capture, int want_signal, int want_fields)
{
    CHECK_CTX(c);
    g_hostprof.start();
    // Two calls may be in flight: the second one is prepared (tables, culling, uploads on the copy stream) and enqueued
    // while the first one runs; only the device state is sequentially dependent, and the stream orders that.
    if (c->n_pending >= 2) return fail(c, WV_ERR_STATE, "wv_integrate_begin: two integrates are already pending");
    if (c->n_pending == 1) {
        const wv_ctx::Slot &o = c->slot[(c->next_slot + 1) % 2];
        if (want_fields == 1 || (o.want_fields && !o.streamed) || c->profiling)
            return fail(c, WV_ERR_STATE, "wv_integrate_begin: previous integrate not ended (a second call may only be enqueued "
                                         "when neither call returns trajectories through the device buffer and profiling is off)");
    }
    if (!tspan || nsteps < 1) return fail(c, WV_ERR_INVALID, "wv_integrate: tspan NULL or nsteps < 1");
    const int seq_n = c->seq_n, sps = c->seq_steps;  // a design sequence describes this call (and only this one)
    c->seq_n = 0;
    if (seq_n > 0 && (long long)seq_n * sps != nsteps)
        return fail(c, WV_ERR_INVALID, "wv_integrate: nsteps must be n_actions * steps_per_action of wv_set_design_sequence");
    if (seq_n > 0 && want_fields) return fail(c, WV_ERR_INVALID, "wv_integrate: a design sequence returns no trajectories");
    const bool capture_all = capture == 2;  // the frames of EVERY action of a sequence
    if (capture_all && (seq_n < 1 || sps <= 2 * WV_FRAMESKIP))
        return fail(c, WV_ERR_INVALID, "wv_integrate: capture_frames == 2 needs a design sequence with more than 20 steps per action");
    if (capture_all && c->n_pending)
        return fail(c, WV_ERR_STATE, "wv_integrate_begin: a call that keeps every action's frames is not overlapped with another");
    if (c->n_pending == 1 && c->slot[(c->next_slot + 1) % 2].capture_all)
        return fail(c, WV_ERR_STATE, "wv_integrate_begin: the pending call keeps every action's frames: end it first");
    if (capture && nsteps < 2 * WV_FRAMESKIP)
        return fail(c, WV_ERR_INVALID,
                    "wv_integrate: capture_frames needs nsteps >= 20 (sol[:, :, :, end-20:10:end], src/env.jl:116)");
    const int M = c->M;
    const float dt = c->cfg.dt;
    const float hdt = 0.5f * dt;
    const int impl = c->cfg.impl == WV_IMPL_STAGED ? WV_IMPL_STAGED : WV_IMPL_FUSED;  // AUTO -> fused
    const int si = c->next_slot;
    wv_ctx::Slot &q = c->slot[si];
    c->obs_slot = -1;  // (the frames are about to change)
    q.obs_rx = q.obs_ry = 0;  // (set again below when this call's job produces the observation: not by every kind of call)
    // The copy stream is created when first needed, and only by a context that has its device to itself: HIP multiplexes
    // the streams of a process over a few hardware queues, and with several environments per GPU (one stream each) extra
    // streams make kernels of different environments queue behind each other (8 envs: 45 -> 35 Gcell-updates/s).
    const bool shared = g_live_ctx[c->cfg.device & 63] > 1;
    if (!shared && !c->up_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    hipStream_t st = c->stream, up = (shared || !c->up_stream) ? st : c->up_stream;

    // per-stage coefficient tables: stage times t, t + 0.5f0*dt, t + dt (src/dynamics.jl:10-13), built in pinned memory
    // (the slot's buffers were last used by the call before the previous one, which has been ended)
    const size_t ncyl = 3 * (size_t)nsteps * (M > 0 ? M : 1), nsf = 3 * (size_t)nsteps;
    int rc = ensure_pinned(c, &q.d_cyl, &q.h_cyl, &q.cyl_cap, ncyl);
    if (rc) return rc;
    rc = ensure_pinned(c, &q.d_sfac, &q.h_sfac, &q.sfac_cap, nsf);
    if (rc) return rc;
    int row_lo = 0, row_hi = 0;  // rows of the earliest / latest stage time (bounding boxes of the cylinder culling)
    float t_lo = INFINITY, t_hi = -INFINITY;
    bool t_ok = true;
    // the source's time factors (a double-precision sin each) and the stage-time range; the cylinders only when asked for
    auto build_tables = [&](bool with_cylinders) {
        t_lo = INFINITY;
        t_hi = -INFINITY;
        t_ok = true;
        for (int s = 0; s < nsteps; ++s) {
            // (a sequence: every action brings its own tspan of sps + 1 entries and its own interpolator)
            const int act = seq_n > 0 ? s / sps : 0;
            const float t = seq_n > 0 ? tspan[(size_t)act * (sps + 1) + (s - act * sps)] : tspan[s];
            const float tq[3] = {t, t + hdt, t + dt};
            for (int k = 0; k < 3; ++k) {
                q.h_sfac[3 * s + k] = c->has_source ? source_factor(tq[k], c->freq) : 0.0f;
                if (with_cylinders && M > 0 && seq_n > 0)
                    design_at(M, c->seq_d.data() + (size_t)act * M * 4, c->seq_d.data() + (size_t)(act + 1) * M * 4, c->seq_t[2 * (size_t)act],
                              c->seq_t[2 * (size_t)act + 1], tq[k], q.h_cyl + (size_t)(3 * s + k) * M);
                else if (with_cylinders && M > 0)
                    design_at(c, tq[k], q.h_cyl + (size_t)(3 * s + k) * M);
                t_ok = t_ok && isfinite(tq[k]);
                if (tq[k] < t_lo) { t_lo = tq[k]; row_lo = 3 * s + k; }
                if (tq[k] > t_hi) { t_hi = tq[k]; row_hi = 3 * s + k; }
            }
        }
        if (!t_ok) row_lo = row_hi = -1;  // (a NaN time: let the culling look at every row)
        if (seq_n > 0) row_lo = row_hi = -1;  // (a sequence is not ONE monotone interpolation: every row)
    };
    if (c->frames_exposed) {  // somebody holds the raw pointer of env.wave: assume it was written
        fused_state_changed(c->fused);
        c->elast_valid = false;
    }
    // A resident launch that is still on the device (waiting for this very call, ideally) owns the stream: the call may be
    // handed to it only if nothing has to run on the stream first.  Otherwise the launch is told to leave, and the call
    // starts a new one behind its prerequisites.
    static const bool force_res = getenv("WAVES_AMD_FORCE_RESIDENT") && atoi(getenv("WAVES_AMD_FORCE_RESIDENT")) != 0;
    const bool may_stay = impl == WV_IMPL_FUSED && !c->profiling && !shared && c->stream == c->own_stream && want_fields == 0 &&
                          !capture_all && !c->frames_exposed;
    bool launch_there = impl == WV_IMPL_FUSED && fused_persist_alive(c->fused);
    const bool plan_needs_stream = c->fused && fused_needs_stream(c->fused, capture != 0, c->has_source ? c->d_G : nullptr, c->cur2 ^ 1);
    if (c->fused) {
        const bool needs = !may_stay || (want_signal && !c->elast_valid) || (capture && nsteps == 2 * WV_FRAMESKIP) || plan_needs_stream;
        if (needs || impl != WV_IMPL_FUSED) {
            if (fused_retire(c->fused) != 0) return fail(c, WV_ERR_HIP, "wv_integrate_begin: the resident launch did not leave");
            launch_there = false;
        }
    }
    // A call for the resident kernel with a design small enough for the tiles to evaluate and cull themselves
    // (k_steps_resident, FusedParams::dsg / dev_cull) needs nothing from the host but the source's time factors: no cylinder
    // table (91 KB per action at 700^2), no culling, no upload, no wait for a copy stream -- whether a launch is already
    // there or a new one starts with this call (the first upload after an idle spell takes the copy engine ~0.4 ms).  The
    // tile table is the one an earlier call of this tiling left on the device.
    static const bool dev_tables_on = !(getenv("WAVES_AMD_DEV_TABLES") && atoi(getenv("WAVES_AMD_DEV_TABLES")) == 0);
    // Only for a call that is not begun under another one's device work: there the host's table building, culling and
    // uploads are hidden anyway, and the tiles' own evaluation costs the small grids more than it saves (256^2: +5 % per call).
    static const bool dev_tables_always = getenv("WAVES_AMD_DEV_TABLES") && atoi(getenv("WAVES_AMD_DEV_TABLES")) == 2;
    bool dev_mode = dev_tables_on && (c->n_pending == 0 || dev_tables_always) && impl == WV_IMPL_FUSED && !c->profiling && !plan_needs_stream && seq_n == 0 && M >= 1 &&
                    M <= kDevTablesMaxCyl && nsteps >= 2 && nsteps <= kDevTablesMaxSteps && fused_dev_tables_ok(c->fused);
    build_tables(!dev_mode);
    if (dev_mode && !t_ok) {
        dev_mode = false;
        build_tables(true);
    }
    g_hostprof.lap(0);
    // everything a call's kernels need from the host when the tiles do NOT help themselves
    auto host_prepare = [&](bool there) -> int {
        if (M > 0) HIPCHK(c, hipMemcpyAsync(q.d_cyl, q.h_cyl, ncyl * sizeof(Cyl), hipMemcpyHostToDevice, up));
        HIPCHK(c, hipMemcpyAsync(q.d_sfac, q.h_sfac, nsf * sizeof(float), hipMemcpyHostToDevice, up));
        g_hostprof.lap(1);
        if (impl == WV_IMPL_FUSED) {
            const int r2 = fused_prepare(c->fused, si, c->d_frames, frame(c, 2), other2(c), c->cur2 ^ 1, c->d_scratch[0], c->d_scratch[1],
                                         capture != 0, c->has_source ? c->d_G : nullptr, q.d_cyl, M > 0 ? q.h_cyl : nullptr, M, 3 * nsteps, st,
                                         up, row_lo, row_hi, there);
            if (r2) return fail(c, r2 == 2 ? WV_ERR_INVALID : WV_ERR_HIP, "fused_prepare failed");
            if (fused_generation(c->fused) != c->elast_generation) c->elast_valid = false;
            c->elast_generation = fused_generation(c->fused);
        } else if (up != st) {
            HIPCHK(c, hipEventRecord(c->up_ev, up));
            HIPCHK(c, hipStreamWaitEvent(st, c->up_ev, 0));
        }
        return WV_OK;
    };
    if (impl == WV_IMPL_FUSED) {
        // (WAVES_AMD_FORCE_RESIDENT=1: experiments with several co-resident resident kernels -- the caller answers for
        // the sum of their tiles fitting the device's block slots)
        fused_allow_resident(c->fused, force_res || g_live_ctx[c->cfg.device & 63] <= 1);
        fused_allow_persist(c->fused, may_stay);
    }
    if (dev_mode) {
        fused_prepare_light(c->fused, si);
        g_hostprof.lap(1);
    } else {
        rc = host_prepare(launch_there);
        if (rc) return rc;
    }
    g_hostprof.lap(2);
    const int nblocks = impl == WV_IMPL_STAGED ? staged_energy_blocks(c->grid) : fused_energy_blocks(c->fused);
    if (want_signal) {
        rc = ensure(c, &q.d_epart, &q.epart_cap, (size_t)(nsteps + 1) * nblocks * 3);
        if (rc) return rc;
        const size_t ns = (size_t)(nsteps + 1) * 3;
        if (ns > q.signal_cap) {
            if (q.h_signal) (void)hipHostFree(q.h_signal);
            q.h_signal = nullptr;
            q.signal_cap = 0;
            HIPCHK(c, hipHostMalloc((void **)&q.h_signal, ns * sizeof(float), hipHostMallocDefault));
            q.signal_cap = ns;
        }
        if (c->elast_blocks != nblocks) c->elast_valid = false;
    }
    const int tstride = c->traj_stride;
    const int nplanes = nsteps / tstride + 1;  // saved times 0, stride, 2*stride, ... <= nsteps
    const bool streamed = want_fields == 2;
    if (streamed) {  // the planes go to the slot's device buffer and from there, on the copy stream, to pinned host memory
                     // while the NEXT call's kernels run (writing them to host memory from the step kernel itself was
                     // measured: the PCIe bursts stall the tiles, +65 %)
        const size_t need = (size_t)2 * nplanes * c->P;
        const int b = (int)(c->stream_seq++ % 3u);
        if (need > c->stream_cap[b]) {
            if (c->h_stream[b]) (void)hipHostFree(c->h_stream[b]);
            c->h_stream[b] = nullptr;
            c->stream_cap[b] = 0;
            HIPCHK(c, hipHostMalloc((void **)&c->h_stream[b], need * sizeof(float), hipHostMallocDefault));
            c->stream_cap[b] = need;
        }
        q.h_traj = c->h_stream[b];
        rc = ensure(c, &q.d_traj, &q.traj_cap, need);
        if (rc) return rc;
    } else if (want_fields) {
        rc = ensure(c, &c->d_traj, &c->traj_cap, (size_t)2 * nplanes * c->P);
        if (rc) return rc;
    }
    {  // one pair of events always (it brackets the integrator launch(es) of the call), one pair per step when profiling
        const size_t want = c->profiling ? 2 * (size_t)nsteps : 2;
        while (q.kev.size() < want) {
            hipEvent_t ev;
            HIPCHK(c, hipEventCreate(&ev));
            q.kev.push_back(ev);
        }
    }

    g_hostprof.lap(3);
    float *traj = streamed ? q.d_traj : c->d_traj;
    float *tt = want_fields ? traj : nullptr;                                    // u_tot planes
    float *ti_ = want_fields ? traj + (size_t)nplanes * c->P : nullptr;             // u_inc planes

    float *cur = frame(c, 2);
    // energies of the initial state: the very partial sums the previous call ended on when the state is unchanged (the
    // reference sums the same array in both places, src/env.jl:105-111), else a fresh reduction
    const float *row0 = nullptr;
    if (want_signal) {
        if (c->elast_valid) {
            row0 = c->elast;
        } else {
            launch_energy_partial(c->grid, cur, q.d_epart, nblocks, st);
            row0 = q.d_epart;
        }
    }
    if (capture_all) {
        const size_t need = (size_t)(seq_n - 1) * 3 * c->N;
        const size_t had = c->seq_frames_cap;
        if (need > 0) {
            rc = ensure(c, &c->d_seq_frames, &c->seq_frames_cap, need);
            if (rc) return rc;
        }
        if (c->seq_frames_cap != had) c->seq_frames_clean = false;
        const bool reduced = impl == WV_IMPL_FUSED && fused_reduced(c->fused);
        if (need > 0 && reduced && !c->seq_frames_clean) {  // (tiles with reduced field sets write 6 or 8 of the 12 planes)
            HIPCHK(c, hipMemsetAsync(c->d_seq_frames, 0, c->seq_frames_cap * sizeof(float), st));
            c->seq_frames_clean = true;
        }
        if (!reduced && impl == WV_IMPL_FUSED) c->seq_frames_clean = false;
    }
    c->seq_frames_actions = 0;  // (set when the call has been enqueued)
    if (want_fields) launch_copy_planes(cur, c->P, tt, ti_, st);
    if (capture && nsteps == 2 * WV_FRAMESKIP)
        HIPCHK(c, hipMemcpyAsync(frame(c, 0), cur, c->N * sizeof(float), hipMemcpyDeviceToDevice, st));

    const FusedCall fcall{c->has_source ? c->d_G : nullptr, c->has_source ? q.d_sfac : nullptr, dt};
    const bool staged_prof = c->profiling && impl == WV_IMPL_STAGED;
    if (impl == WV_IMPL_STAGED && !c->profiling) HIPCHK(c, hipEventRecord(q.kev[0], st));
    c->fsteps.clear();
    for (int s = 1; s <= nsteps; ++s) {
        float *out;
        // (capture_all: action a = (s - 1) / sps ends at step (a + 1) * sps; its frames go to their own buffers)
        const int act_s = capture_all ? (s - 1) / sps : 0;
        const int rel = capture_all ? s - act_s * sps : 0;
        float *const fa = (capture_all && act_s < seq_n - 1) ? c->d_seq_frames + (size_t)act_s * 3 * c->N : nullptr;
        if (fa && rel == sps - 2 * WV_FRAMESKIP) out = fa;
        else if (fa && rel == sps - WV_FRAMESKIP) out = fa + c->N;
        else if (fa && rel == sps) out = fa + 2 * c->N;
        else if (capture && s == nsteps - 2 * WV_FRAMESKIP) out = frame(c, 0);
        else if (capture && s == nsteps - WV_FRAMESKIP) out = frame(c, 1);
        else if (s == nsteps) out = other2(c);  // (never the buffer the call started from: see wv_ctx::cur2)
        else out = (cur == c->d_scratch[0]) ? c->d_scratch[1] : c->d_scratch[0];
        const Cyl *cyl_s = q.d_cyl + (size_t)(3 * (s - 1)) * (M > 0 ? M : 0);
        const float *sf = q.h_sfac + 3 * (size_t)(s - 1);
        float *ep = want_signal ? q.d_epart + (size_t)s * nblocks * 3 : nullptr;
        const bool keep_t = s % tstride == 0;
        float *tts = (tt && keep_t) ? tt + (size_t)(s / tstride) * c->P : nullptr;
        float *tis = (ti_ && keep_t) ? ti_ + (size_t)(s / tstride) * c->P : nullptr;
        const float *G = c->has_source ? c->d_G : nullptr;
        if (staged_prof) HIPCHK(c, hipEventRecord(q.kev[2 * (s - 1)], st));
        if (impl == WV_IMPL_STAGED) {
            StageIO io{};
            io.u = cur;
            io.acc = c->d_acc;
            io.G = G;
            io.M = M;
            io.dt = dt;
            // k1
            io.yin = cur; io.out = c->d_yA; io.sfac = sf[0]; io.cyl = cyl_s; io.a = hdt;
            launch_stage(c->grid, io, 0, st);
            // k2
            io.yin = c->d_yA; io.out = c->d_yB; io.sfac = sf[1]; io.cyl = cyl_s + M; io.a = hdt;
            launch_stage(c->grid, io, 1, st);
            // k3
            io.yin = c->d_yB; io.out = c->d_yA; io.sfac = sf[1]; io.cyl = cyl_s + M; io.a = dt;
            launch_stage(c->grid, io, 2, st);
            // k4 + update + energies
            io.yin = c->d_yA; io.out = out; io.sfac = sf[2]; io.cyl = cyl_s + 2 * M; io.a = dt;
            io.epart = ep; io.traj_tot = tts; io.traj_inc = tis;
            launch_stage(c->grid, io, 3, st);
        } else {
            FusedStep fs{};
            fs.u = cur;
            fs.out = out;
            fs.keep = out != c->d_scratch[0] && out != c->d_scratch[1];  // a frame of env.wave
            fs.epart = ep;
            fs.traj_tot = tts;
            fs.traj_inc = tis;
            c->fsteps.push_back(fs);  // launched together below
        }
        if (staged_prof) HIPCHK(c, hipEventRecord(q.kev[2 * (s - 1) + 1], st));
        cur = out;
    }
    g_hostprof.lap(4);
    q.prof_launches = nsteps * (impl == WV_IMPL_STAGED ? 4 : 1);
    q.prof_events = nsteps;
    q.bracketed = false;
    q.resident = false;
    if (impl == WV_IMPL_STAGED && !c->profiling) {
        HIPCHK(c, hipEventRecord(q.kev[1], st));  // (whole chain of stage kernels: total_ms of the call)
    } else if (impl == WV_IMPL_FUSED && !c->profiling) {
        // The resident path hands the call to the resident launch as a job (which does the second pass of the energy sums
        // itself, straight into q.h_signal, and is timed by its own clock stamps); the single-step path puts its launches
        // between two events.
        const FusedEnergy ef{row0, q.d_epart, want_signal ? q.h_signal : nullptr, c->dOmega};
        // state(env) of the frames this call leaves, by the job itself, when the caller has been asking for it
        const size_t nob = (size_t)c->obs_auto_rx * c->obs_auto_ry * 4;
        const bool with_obs = nob > 0 && nob <= q.h_obs_cap && seq_n == 0 && !c->frames_exposed;
        const FusedObs ob{frame(c, 0), frame(c, 1), other2(c), c->has_source ? c->d_G : nullptr, with_obs ? q.h_obs : nullptr,
                          c->obs_auto_rx, c->obs_auto_ry};
        q.obs_rx = with_obs ? c->obs_auto_rx : 0;
        q.obs_ry = with_obs ? c->obs_auto_ry : 0;
        int fr = -1;
        if (dev_mode) {
            const FusedDevTables dev{M, c->d0.data(), c->d1.data(), c->ti, c->tf, tspan, c->has_source ? q.h_sfac : nullptr, t_lo, t_hi};
            fr = fused_run(c->fused, si, fcall, c->fsteps.data(), (int)c->fsteps.size(), st, up, ef, may_stay, q.kev[0], q.kev[1], &dev, &ob);
            if (fr == 3) {  // not a call for the resident kernel after all (more tiles than the device holds, shared device, ...)
                dev_mode = false;
                build_tables(true);
                rc = host_prepare(false);
                if (rc) return rc;
            }
        }
        if (!dev_mode) fr = fused_run(c->fused, si, fcall, c->fsteps.data(), (int)c->fsteps.size(), st, up, ef, may_stay, q.kev[0], q.kev[1], nullptr, &ob);
        if (fr != 0) return fail(c, WV_ERR_HIP, std::string("fused_run failed: ") + hipGetErrorString(hipGetLastError()));
        q.resident = fused_last_resident(c->fused);
        if (!q.resident) q.obs_rx = q.obs_ry = 0;  // (the single-step kernels produce no observation)
        q.bracketed = !q.resident;
        if (q.resident) q.prof_launches = 1;
    } else if (impl == WV_IMPL_FUSED) {
        // profiling: the resident launch (it ends with the call) bracketed by one pair of events, else every step by its own pair
        HIPCHK(c, hipEventRecord(q.kev[0], st));
        const FusedEnergy ef{row0, q.d_epart, want_signal ? q.h_signal : nullptr, c->dOmega};
        const int rr = fused_try_resident(c->fused, si, fcall, c->fsteps.data(), (int)c->fsteps.size(), st, up, ef, false);
        if (rr > 0) return fail(c, WV_ERR_HIP, std::string("fused_try_resident failed: ") + hipGetErrorString(hipGetLastError()));
        if (rr == 0) {
            HIPCHK(c, hipEventRecord(q.kev[1], st));
            q.prof_launches = 1;
            q.prof_events = 1;
            q.resident = true;
        } else {
            for (int s = 0; s < nsteps; ++s) {
                HIPCHK(c, hipEventRecord(q.kev[2 * s], st));
                fused_launch(c->fused, si, fcall, s, c->fsteps[s], st);
                HIPCHK(c, hipEventRecord(q.kev[2 * s + 1], st));
            }
        }
    }
    g_hostprof.lap(5);
    HIPCHK(c, hipGetLastError());
    q.fsteps = c->fsteps;
    q.fcall = fcall;
    q.row0 = row0;
    q.nblocks = nblocks;
    q.dev_mode = dev_mode;
    if (dev_mode) {  // (a re-run on the single-step kernels has to build what this call did without)
        q.in_tspan.assign(tspan, tspan + nsteps + 1);
        q.in_d0 = c->d0;
        q.in_d1 = c->d1;
        q.in_ti = c->ti;
        q.in_tf = c->tf;
        q.in_M = M;
        q.in_capture = capture != 0;
        q.in_rows[0] = row_lo;
        q.in_rows[1] = row_hi;
    }
    // (nothing may be enqueued behind a resident launch that stays on the device: it would wait out the launch's idle limit)
    const bool stream_free = !(impl == WV_IMPL_FUSED && fused_persist_alive(c->fused));
    if (want_signal) {
        // second pass of the reductions (single-step and staged paths; the resident kernel has done it), written straight
        // into pinned host memory: wv_integrate_end only waits for the call's last event and copies 1.2 KB host to host
        if (!q.resident) launch_energy_final(row0, q.d_epart, nsteps + 1, nblocks, c->dOmega, q.h_signal, nullptr, nullptr, st);
        c->elast = q.d_epart + (size_t)nsteps * nblocks * 3;
        c->elast_blocks = nblocks;
    }
    c->elast_valid = want_signal != 0;
    if (stream_free) HIPCHK(c, hipEventRecord(q.ev1, st));
    if (streamed) {
        if (!c->down_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
        HIPCHK(c, hipStreamWaitEvent(c->down_stream, q.ev1, 0));
        HIPCHK(c, hipMemcpyAsync(q.h_traj, q.d_traj, (size_t)2 * nplanes * c->P * sizeof(float), hipMemcpyDeviceToHost,
                                 c->down_stream));
        HIPCHK(c, hipEventRecord(q.copy_ev, c->down_stream));
    }
    HIPCHK(c, hipGetLastError());

    if (c->frames_exposed) {
        // somebody holds the raw pointer of env.wave (wv_device_frames): the last frame has to stay where that pointer says,
        // so the final state is copied home behind the call (23.5 MB at 700^2 per call, on this path only)
        HIPCHK(c, hipMemcpyAsync(c->d_frames + 2 * c->N, c->d_f2alt, c->N * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIPCHK(c, hipEventRecord(q.ev1, st));
    } else {
        c->cur2 ^= 1;  // env.wave[:, :, :, end] is what this call leaves behind (stream order)
    }
    q.pending = true;
    q.capture_all = capture_all;
    if (capture_all) c->seq_frames_actions = seq_n;
    q.nsteps = nsteps;
    q.want_signal = want_signal != 0;
    q.want_fields = want_fields != 0;
    q.streamed = streamed;
    q.planes = nplanes;
    q.impl = impl;
    c->n_pending++;
    c->next_slot = (si + 1) % 2;
    g_hostprof.lap(6);
    return WV_OK;
}

// The resident kernel gave up the oldest pending call `si` (and with it every call behind it: the launch has left).  Nothing has
// been written over the initial condition (the final state goes to the other of two buffers, wv_ctx::cur2), so the very same call -- same tables, same
// buffers -- runs again on the single-step kernels, and so does a second pending call that was meant for the same launch.
static int rerun_after_give_up(wv_ctx *c, int si)
{
    hipStream_t st = c->stream;
    fused_gave_up(c->fused, st);
    const int order[2] = {si, (si + 1) % 2};
    for (int k = 0; k < 2; ++k) {
        wv_ctx::Slot &q = c->slot[order[k]];
        if (!q.pending || !q.resident) continue;
        if (q.dev_mode) {  // the call was described to the tiles by its interpolator alone: the tables the single-step kernels read
            const int M = q.in_M, n = q.nsteps;
            const float dt = c->cfg.dt, hdt = 0.5f * dt;
            for (int s = 0; s < n; ++s) {
                const float tq[3] = {q.in_tspan[s], q.in_tspan[s] + hdt, q.in_tspan[s] + dt};
                for (int k = 0; k < 3; ++k)
                    design_at(M, q.in_d0.data(), q.in_d1.data(), q.in_ti, q.in_tf, tq[k], q.h_cyl + (size_t)(3 * s + k) * M);
            }
            HIPCHK(c, hipMemcpyAsync(q.d_cyl, q.h_cyl, 3 * (size_t)n * M * sizeof(Cyl), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(q.d_sfac, q.h_sfac, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
            // (the buffers of this call are in q.fsteps; the plan only needs the tables, the culled lists and the tile order)
            if (fused_prepare(c->fused, order[k], c->d_frames, const_cast<float *>(q.fsteps.front().u), q.fsteps.back().out, 0, c->d_scratch[0],
                              c->d_scratch[1], q.in_capture, c->has_source ? c->d_G : nullptr, q.d_cyl, q.h_cyl, M, 3 * n, st, st, q.in_rows[0],
                              q.in_rows[1], false) != 0)
                return fail(c, WV_ERR_HIP, "fused_prepare failed while a given-up call was prepared for the single-step kernels");
            q.fcall.d_sfac = c->has_source ? q.d_sfac : nullptr;
            q.dev_mode = false;
        }
        if (fused_rerun_steps(c->fused, order[k], q.fcall, q.fsteps.data(), (int)q.fsteps.size(), st, q.kev[0], q.kev[1]) != 0)
            return fail(c, WV_ERR_HIP, std::string("the single-step kernels failed after a resident give-up: ") + hipGetErrorString(hipGetLastError()));
        if (q.want_signal) launch_energy_final(q.row0, q.d_epart, q.nsteps + 1, q.nblocks, c->dOmega, q.h_signal, nullptr, nullptr, st);
        HIPCHK(c, hipEventRecord(q.ev1, st));
        if (q.streamed) {  // (the copy enqueued with the call took what the abandoned launch had left there)
            HIPCHK(c, hipStreamWaitEvent(c->down_stream, q.ev1, 0));
            HIPCHK(c, hipMemcpyAsync(q.h_traj, q.d_traj, (size_t)2 * q.planes * c->P * sizeof(float), hipMemcpyDeviceToHost, c->down_stream));
            HIPCHK(c, hipEventRecord(q.copy_ev, c->down_stream));
        }
        q.resident = false;
        q.bracketed = true;
        q.prof_launches = q.nsteps;
        q.prof_events = q.nsteps;
        q.gave_up = true;
    }
    return WV_OK;
}

// poll an event for a while before falling back to a blocking wait (the wake-up of a blocking wait costs more than the
// rest of wv_integrate_end)
static int wait_event(wv_ctx *c, hipEvent_t ev)
{
    static const bool spin = !(getenv("WAVES_AMD_SPIN") && atoi(getenv("WAVES_AMD_SPIN")) == 0);
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t e = hipEventQuery(ev);
            if (e == hipSuccess) return WV_OK;
            if (e != hipErrorNotReady) {
                (void)hipGetLastError();
                break;
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
    }
    HIPCHK(c, hipEventSynchronize(ev));
    return WV_OK;
}

int wv_integrate_end(wv_ctx *c, float *signal, float *u_tot, float *u_inc)
{
    CHECK_CTX(c);
    if (!c->n_pending) return fail(c, WV_ERR_STATE, "wv_integrate_end: no integrate pending");
    const int si = (c->next_slot + 2 - c->n_pending) % 2;  // the oldest pending call
    wv_ctx::Slot &q = c->slot[si];
    // (argument errors leave the call pending: its device work and the buffers it uses stay protected)
    if (signal && !q.want_signal) return fail(c, WV_ERR_STATE, "wv_integrate_end: signal was not requested in _begin");
    if ((u_tot || u_inc) && !q.want_fields) return fail(c, WV_ERR_STATE, "wv_integrate_end: fields were not requested in _begin");
    const int n = q.nsteps;
    hipStream_t st = c->stream;
    const size_t tp = (size_t)q.planes * c->P;
    double job_ms = -1.0;
    if (q.impl == WV_IMPL_FUSED && q.resident) {
        // the job's completion word in pinned memory (no HIP call on this path when the launch stays on the device)
        const int jr = fused_job_wait(c->fused, si, st);
        if (jr == 1) return fail(c, WV_ERR_HIP, std::string("waiting for the resident launch failed: ") + hipGetErrorString(hipGetLastError()));
        if (jr == 2) {
            const int rc = rerun_after_give_up(c, si);
            if (rc) return rc;
        } else {
            job_ms = fused_last_job_ms(c->fused);
        }
    }
    if (u_tot && !q.streamed) HIPCHK(c, hipMemcpyAsync(u_tot, c->d_traj, tp * sizeof(float), hipMemcpyDeviceToHost, st));
    if (u_inc && !q.streamed) HIPCHK(c, hipMemcpyAsync(u_inc, c->d_traj + tp, tp * sizeof(float), hipMemcpyDeviceToHost, st));
    if ((u_tot || u_inc) && !q.streamed) {
        HIPCHK(c, hipStreamSynchronize(st));
    } else if (job_ms < 0.0) {
        // everything the caller gets is already on its way (or here): wait for the call's last event
        const int rc = wait_event(c, q.ev1);
        if (rc) return rc;
    }
    if (q.streamed) HIPCHK(c, hipEventSynchronize(q.copy_ev));
    if (c->frames_exposed && job_ms >= 0.0) {  // (the copy of the final state to where the raw pointer expects it)
        const int rc = wait_event(c, q.ev1);
        if (rc) return rc;
    }
    q.pending = false;
    c->n_pending--;
    // state(env) of the frames this call leaves is in the slot's pinned buffer when its job produced it (not after a give-up:
    // the single-step kernels ran the call then); it is the CURRENT one only while no other call is pending
    if (q.obs_rx > 0 && job_ms >= 0.0 && c->n_pending == 0) {
        c->obs_slot = si;
        if (++c->obs_unused > 8) c->obs_auto_rx = c->obs_auto_ry = 0;  // nobody reads them any more: stop producing them
    } else {
        c->obs_slot = -1;
    }
    if (signal) memcpy(signal, q.h_signal, (size_t)(n + 1) * 3 * sizeof(float));
    if (q.streamed) {  // (the planes are already in host memory; callers that can read them in place use wv_integrate_end_view)
        if (u_tot) memcpy(u_tot, q.h_traj, tp * sizeof(float));
        if (u_inc) memcpy(u_inc, q.h_traj + tp, tp * sizeof(float));
    }
    c->last_view_tot = q.streamed ? q.h_traj : nullptr;
    c->last_view_inc = q.streamed ? q.h_traj + tp : nullptr;
    c->last_view_planes = q.streamed ? q.planes : 0;
    if (q.impl == WV_IMPL_FUSED && c->n_pending == 0 && !fused_persist_alive(c->fused)) fused_dump_stamps(c->fused, st);
    c->timing = wv_timing{};
    c->timing.steps = n;
    c->timing.impl = q.impl;
    c->timing.resident = q.resident ? 1 : 0;
    c->timing.gave_up = q.gave_up ? 1 : 0;
    q.gave_up = false;
    if (q.resident && job_ms >= 0.0) {
        // the leader tile's clock stamps: job seen -> outputs complete (in a launch that stays, HIP events see only the launch)
        c->timing.total_ms = job_ms;
        c->timing.step_kernel_ms = job_ms;
        c->timing.step_kernel_launches = 1;
    } else {
        float ms = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&ms, q.kev[0], q.ev1));
        c->timing.total_ms = ms;  // first integrator launch -> last device work of the call
    }
    if (q.bracketed) {
        float k = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&k, q.kev[0], q.kev[1]));
        c->timing.step_kernel_ms = k;
        c->timing.step_kernel_launches = q.prof_launches;
    }
    if (c->profiling) {
        // (a resident call is known to be complete from the kernel's own completion word, a moment before the events behind
        // the launch have fired)
        if (q.prof_events > 0) HIPCHK(c, hipEventSynchronize(q.kev[2 * (q.prof_events - 1) + 1]));
        double sum = 0.0;
        for (int s = 0; s < q.prof_events; ++s) {
            float k = 0.0f;
            HIPCHK(c, hipEventElapsedTime(&k, q.kev[2 * s], q.kev[2 * s + 1]));
            sum += k;
        }
        c->timing.step_kernel_ms = sum;
        c->timing.step_kernel_launches = q.prof_launches;
    }
    fused_launch_stats(c->fused, &c->timing.launch_ms, &c->timing.launch_jobs);
    return WV_OK;
}

int wv_integrate_end_view(wv_ctx *c, float *signal, const float **u_tot, const float **u_inc, int *planes)
{
    int rc = wv_integrate_end(c, signal, nullptr, nullptr);
    if (rc) return rc;
    if (!c->last_view_tot) return fail(c, WV_ERR_STATE, "wv_integrate_end_view: the call was not begun with want_fields == 2");
    if (u_tot) *u_tot = c->last_view_tot;
    if (u_inc) *u_inc = c->last_view_inc;
    if (planes) *planes = c->last_view_planes;
    return WV_OK;
}

int wv_integrate(wv_ctx *c, const float *tspan, int nsteps, int capture, float *signal, float *u_tot, float *u_inc)
{
    int rc = wv_integrate_begin(c, tspan, nsteps, capture, signal != nullptr, (u_tot || u_inc) ? 1 : 0);
    if (rc) return rc;
    return wv_integrate_end(c, signal, u_tot, u_inc);
}

int wv_pending(wv_ctx *c, int *count)
{
    if (!c) return fail(nullptr, WV_ERR_INVALID, "ctx is NULL");
    if (!count) return fail(c, WV_ERR_INVALID, "wv_pending: NULL");
    *count = c->n_pending;
    return WV_OK;
}

int wv_set_trajectory_stride(wv_ctx *c, int stride)
{
    CHECK_CTX(c);
    if (stride < 1) return fail(c, WV_ERR_INVALID, "wv_set_trajectory_stride: stride must be >= 1");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_set_trajectory_stride: an integrate is pending");
    c->traj_stride = stride;
    return WV_OK;
}

int wv_set_profiling(wv_ctx *c, int on)
{
    CHECK_CTX(c);
    c->profiling = on != 0;
    return WV_OK;
}

int wv_get_timing(wv_ctx *c, wv_timing *out)
{
    CHECK_CTX(c);
    if (!out) return fail(c, WV_ERR_INVALID, "wv_get_timing: NULL");
    fused_launch_stats(c->fused, &c->timing.launch_ms, &c->timing.launch_jobs);  // (a launch may have ended since the last call: wv_synchronize)
    *out = c->timing;
    return WV_OK;
}

int wv_get_call_times(wv_ctx *c, double *ms, int cap, int *n)
{
    if (!c) return fail(nullptr, WV_ERR_INVALID, "ctx is NULL");
    if (!n || (cap > 0 && !ms) || cap < 0) return fail(c, WV_ERR_INVALID, "wv_get_call_times: bad arguments");
    *n = fused_job_times(c->fused, ms, cap);
    return WV_OK;
}

int wv_set_stream(wv_ctx *c, void *hip_stream)
{
    CHECK_CTX(c);
    QUIET(c);
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_set_stream: an integrate is pending");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return WV_OK;
}

int wv_synchronize(wv_ctx *c)
{
    CHECK_CTX(c);
    QUIET(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WV_OK;
}

int wv_device_frames(wv_ctx *c, void **dptr, size_t *bytes)
{
    CHECK_CTX(c);
    c->obs_slot = -1;  // (what state(env) shows changes)
    QUIET(c);
    if (!dptr) return fail(c, WV_ERR_INVALID, "wv_device_frames: NULL");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_device_frames: an integrate is pending");
    {
        const int rc = frames_home(c);
        if (rc) return rc;
    }
    *dptr = c->d_frames;
    if (bytes) *bytes = 3 * c->N * sizeof(float);
    // the caller may write through the pointer at any time from now on: until wv_release_device_frames every integrate
    // looks at the state afresh (reduced-field-set precondition, initial energies)
    c->frames_exposed = true;
    fused_state_changed(c->fused);
    c->elast_valid = false;
    return WV_OK;
}

int wv_release_device_frames(wv_ctx *c)
{
    CHECK_CTX(c);
    c->obs_slot = -1;  // (what state(env) shows changes)
    QUIET(c);
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_release_device_frames: an integrate is pending");
    c->frames_exposed = false;
    fused_state_changed(c->fused);  // (whatever was written before the release is looked at once more)
    c->elast_valid = false;
    return WV_OK;
}

int wv_selftest_granules(wv_ctx *c, int iters, unsigned long long *checked, unsigned long long *torn)
{
    CHECK_CTX(c);
    QUIET(c);
    if (iters < 1 || !checked || !torn) return fail(c, WV_ERR_INVALID, "wv_selftest_granules: bad arguments");
    if (c->n_pending) return fail(c, WV_ERR_STATE, "wv_selftest_granules: an integrate is pending");
    const unsigned bytes = 32u << 20;
    unsigned char *buf = nullptr;
    unsigned long long *out = nullptr, h[2] = {0, 0};
    HIPCHK(c, hipMalloc((void **)&buf, bytes));
    if (hipMalloc((void **)&out, 2 * sizeof(unsigned long long)) != hipSuccess) {
        (void)hipFree(buf);
        return fail(c, WV_ERR_NOMEM, "wv_selftest_granules: out of device memory");
    }
    hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(unsigned long long), c->stream);
    // dense granules, one granule per 64-byte line, one granule per row of this grid (the x-border pattern)
    const unsigned strides[3] = {16u, 64u, (unsigned)c->nx * 16u};
    for (int k = 0; k < 3 && e == hipSuccess; ++k) {
        e = hipMemsetAsync(buf, 0, bytes, c->stream);
        if (e == hipSuccess) launch_selftest_granules(buf, bytes, iters, 128, strides[k], out, c->stream);
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, out, sizeof(h), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(buf);
    (void)hipFree(out);
    if (e != hipSuccess) return fail(c, WV_ERR_HIP, std::string("wv_selftest_granules: ") + hipGetErrorString(e));
    *checked = h[0];
    *torn = h[1];
    return WV_OK;
}

// build_pml(dim::OneDim, width, scale)[1], src/pml.jl:6-15 (what `dyn.pml[[1]]` is at src/dynamics.jl:192)
static float pml_1d_first(const float *x, int n, float width, float scale)
{
    const float a0 = fabsf(x[0]), an = fabsf(x[n - 1]);
    const float start = (a0 < an ? a0 : an) - width;
    float v = (a0 - start > 0.0f ? a0 - start : 0.0f) / width;
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    return ((v * v) * v) * scale;
}

int wv_latent_integrate(const wv_latent_config *cfg, const float *x, const float *X, const float *Y, const float *shape,
                        const float *PML, const float *z0, const float *t, float *z)
{
    if (!cfg || !x || !X || !Y || !shape || !PML || !z0 || !t || !z) return fail(nullptr, WV_ERR_INVALID, "wv_latent_integrate: NULL argument");
    const int n = cfg->n, B = cfg->batch, K = cfg->knots, steps = cfg->steps;
    if (n < 3 || n > 1024) return fail(nullptr, WV_ERR_INVALID, "wv_latent_integrate: 3 <= n <= 1024 (one thread per cell)");
    if (B < 1 || K < 2 || steps < 1 || !(cfg->dt > 0.0f)) return fail(nullptr, WV_ERR_INVALID, "wv_latent_integrate: bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, WV_ERR_NO_DEVICE, "wv_latent_integrate: no HIP device; libwaves_amd has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, WV_ERR_INVALID, "wv_latent_integrate: device ordinal out of range");
    HIPCHK(nullptr, hipSetDevice(cfg->device));
    const size_t nB = (size_t)n * B, nz0 = nB * 4, nz = nz0 * (steps + 1), nt = (size_t)(steps + 1) * B;
    // host tables: time factors of the source at the three stage times (fp32 argument, accurately rounded sin), times by step
    std::vector<float> sf((size_t)steps * 3 * B), ts(nt);
    const float hdt = 0.5f * cfg->dt;
    for (int s = 0; s <= steps; ++s)
        for (int b = 0; b < B; ++b) ts[(size_t)s * B + b] = t[(size_t)s + (size_t)(steps + 1) * b];   // t is (steps + 1, B) column-major
    for (int s = 0; s < steps; ++s)
        for (int b = 0; b < B; ++b) {
            const float t0 = ts[(size_t)s * B + b];
            const float tq[3] = {t0, t0 + hdt, t0 + cfg->dt};
            for (int q = 0; q < 3; ++q) sf[((size_t)s * 3 + q) * B + b] = source_factor(tq[q], cfg->freq);
        }
    const size_t sizes[8] = {(size_t)K * B, nB * K, nB, nB, sf.size(), nt, nz0, nz};
    size_t off[9] = {0};
    for (int k = 0; k < 8; ++k) off[k + 1] = off[k] + ((sizes[k] + 63) & ~(size_t)63);
    float *d = nullptr;
    if (hipMalloc((void **)&d, off[8] * sizeof(float)) != hipSuccess) return fail(nullptr, WV_ERR_NOMEM, "wv_latent_integrate: out of device memory");
    const float *src[7] = {X, Y, shape, PML, sf.data(), ts.data(), z0};
    hipError_t e = hipSuccess;
    for (int k = 0; k < 7 && e == hipSuccess; ++k) e = hipMemcpy(d + off[k], src[k], sizes[k] * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        LatentArgs a{};
        a.n = n; a.B = B; a.K = K; a.steps = steps;
        a.ops = make_ops(std::vector<float>(x, x + n));
        a.c0 = cfg->c0; a.dt = cfg->dt; a.hdt = hdt;
        a.pml_scale = pml_1d_first(x, n, cfg->pml_width, cfg->pml_scale);
        a.X = d + off[0]; a.Y = d + off[1]; a.shape = d + off[2]; a.PML = d + off[3]; a.sfac = d + off[4]; a.t = d + off[5];
        a.z0 = d + off[6]; a.z = d + off[7];
        launch_latent(a, nullptr);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(z, a.z, nz * sizeof(float), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (e != hipSuccess) return fail(nullptr, WV_ERR_HIP, std::string("wv_latent_integrate: ") + hipGetErrorString(e));
    return WV_OK;
}

int wv_latent_adjoint(const wv_latent_config *cfg, const float *x, const float *X, const float *Y, const float *shape,
                      const float *PML, const float *z, const float *t, const float *adj, float *gz0, float *gY,
                      float *gshape, float *gPML)
{
    if (!cfg || !x || !X || !Y || !shape || !PML || !z || !t || !adj || !gz0 || !gY || !gshape || !gPML)
        return fail(nullptr, WV_ERR_INVALID, "wv_latent_adjoint: NULL argument");
    const int n = cfg->n, B = cfg->batch, K = cfg->knots, steps = cfg->steps;
    if (n < 3 || n > 1024) return fail(nullptr, WV_ERR_INVALID, "wv_latent_adjoint: 3 <= n <= 1024 (one thread per cell)");
    if (B < 1 || K < 2 || steps < 1 || !(cfg->dt > 0.0f)) return fail(nullptr, WV_ERR_INVALID, "wv_latent_adjoint: bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, WV_ERR_NO_DEVICE, "wv_latent_adjoint: no HIP device; libwaves_amd has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, WV_ERR_INVALID, "wv_latent_adjoint: device ordinal out of range");
    HIPCHK(nullptr, hipSetDevice(cfg->device));
    const size_t nB = (size_t)n * B, nz0 = nB * 4, nz = nz0 * (steps + 1), nt = (size_t)(steps + 1) * B;
    // the sweep applies the step's pullback at EVERY saved time (src/dynamics.jl:101): steps + 1 rows of time factors
    std::vector<float> sf((size_t)(steps + 1) * 3 * B), ts(nt);
    const float hdt = 0.5f * cfg->dt;
    for (int s = 0; s <= steps; ++s)
        for (int b = 0; b < B; ++b) {
            const float t0 = t[(size_t)s + (size_t)(steps + 1) * b];
            ts[(size_t)s * B + b] = t0;
            const float tq[3] = {t0, t0 + hdt, t0 + cfg->dt};
            for (int q = 0; q < 3; ++q) sf[((size_t)s * 3 + q) * B + b] = source_factor(tq[q], cfg->freq);
        }
    // inputs X Y shape PML sf ts z adj | outputs gz0 gY gshape gPML
    const size_t sizes[12] = {(size_t)K * B, nB * K, nB, nB, sf.size(), nt, nz, nz, nz0, nB * K, nB, nB};
    size_t off[13] = {0};
    for (int k = 0; k < 12; ++k) off[k + 1] = off[k] + ((sizes[k] + 63) & ~(size_t)63);
    float *d = nullptr;
    if (hipMalloc((void **)&d, off[12] * sizeof(float)) != hipSuccess) return fail(nullptr, WV_ERR_NOMEM, "wv_latent_adjoint: out of device memory");
    const float *src[8] = {X, Y, shape, PML, sf.data(), ts.data(), z, adj};
    hipError_t e = hipSuccess;
    for (int k = 0; k < 8 && e == hipSuccess; ++k) e = hipMemcpy(d + off[k], src[k], sizes[k] * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d + off[9], 0, sizes[9] * sizeof(float));
    if (e == hipSuccess) {
        LatentAdjArgs a{};
        a.f.n = n; a.f.B = B; a.f.K = K; a.f.steps = steps;
        a.f.ops = make_ops(std::vector<float>(x, x + n));
        a.f.c0 = cfg->c0; a.f.dt = cfg->dt; a.f.hdt = hdt;
        a.f.pml_scale = pml_1d_first(x, n, cfg->pml_width, cfg->pml_scale);
        a.f.X = d + off[0]; a.f.Y = d + off[1]; a.f.shape = d + off[2]; a.f.PML = d + off[3]; a.f.sfac = d + off[4]; a.f.t = d + off[5];
        a.f.z0 = nullptr; a.f.z = d + off[6];
        a.adj = d + off[7]; a.gz0 = d + off[8]; a.gY = d + off[9]; a.gshape = d + off[10]; a.gPML = d + off[11];
        launch_latent_adjoint(a, nullptr);
        e = hipGetLastError();
        float *dst[4] = {gz0, gY, gshape, gPML};
        for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipMemcpy(dst[k], d + off[8 + k], sizes[8 + k] * sizeof(float), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (e != hipSuccess) return fail(nullptr, WV_ERR_HIP, std::string("wv_latent_adjoint: ") + hipGetErrorString(e));
    return WV_OK;
}

int wv_device_source_shape(wv_ctx *c, void **dptr, size_t *bytes)
{
    CHECK_CTX(c);
    QUIET(c);
    if (!dptr) return fail(c, WV_ERR_INVALID, "wv_device_source_shape: NULL");
    *dptr = c->d_G;
    if (bytes) *bytes = c->P * sizeof(float);
    fused_source_changed(c->fused);  // the caller may write through the pointer
    return WV_OK;
}

}  // extern "C"
