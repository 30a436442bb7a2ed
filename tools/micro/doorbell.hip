// Micro-measurement for the action-outliving resident kernel (DESIGN 5 "episode-resident kernel"): how long does a job
// take to reach a grid of resident workgroups through a doorbell in pinned host memory, and the completion word to come back?
//   host store (pinned, coherent) -> leader block's system-scope poll -> device "go" word (agent scope) -> every block ->
//   counter -> last block's system-scope store to pinned host memory -> host poll
// Also: does a small kernel / an async copy on ANOTHER stream make progress while the resident grid spins (489 of 512 block
// slots taken, as at 700^2)?   hipcc --offload-arch=gfx950 -O3 -o doorbell doorbell.hip && ./doorbell
// Every device loop is bounded by s_memrealtime (100 MHz): the grid drains whatever the host does.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                           \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

struct Host {          // pinned, coherent
    volatile unsigned bell;      // host -> device: number of jobs rung so far
    unsigned pad0[15];
    volatile unsigned done;      // device -> host: jobs completed
    volatile unsigned exited;    // device -> host: the leader gave up waiting (idle limit)
    unsigned pad1[14];
    volatile unsigned long long t_seen[64], t_done[64];  // device clock stamps of the first jobs
};

__global__ __launch_bounds__(512, 4) void k_resident(Host *h, unsigned *go, unsigned *cnt, int njobs, int work_sleep,
                                                      unsigned long long idle_ticks)
{
    extern __shared__ float lds[];
    lds[threadIdx.x] = 0.0f;
    __shared__ unsigned cmd;
    for (int j = 0; j < njobs; ++j) {
        if (threadIdx.x == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned c = 0;
            if (blockIdx.x == 0) {
                for (;;) {
                    const unsigned b = __hip_atomic_load(&h->bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (b >= (unsigned)(j + 1)) { c = 1; break; }
                    if (__builtin_amdgcn_s_memrealtime() - t0 > idle_ticks) { c = 2; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                if (j < 64) h->t_seen[j] = __builtin_amdgcn_s_memrealtime();
                __hip_atomic_store(&go[j], c, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if (c == 2) __hip_atomic_store(&h->exited, (unsigned)(j + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                for (;;) {
                    c = __hip_atomic_load(&go[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (c) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 4 * idle_ticks) { c = 2; break; }  // (the leader died?)
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            cmd = c;
        }
        __syncthreads();
        if (cmd == 2) return;
        for (int k = 0; k < work_sleep; ++k) __builtin_amdgcn_s_sleep(64);  // "the job"
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned n = __hip_atomic_fetch_add(&cnt[j], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (n + 1 == gridDim.x) {
                if (j < 64) h->t_done[j] = __builtin_amdgcn_s_memrealtime();
                __hip_atomic_store(&h->done, (unsigned)(j + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

__global__ void k_small(unsigned *out) { atomicAdd(out, 1u); }

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 489;
    const int njobs = argc > 2 ? atoi(argv[2]) : 200;
    const int work = argc > 3 ? atoi(argv[3]) : 0;       // s_sleep(64) units (~0.4 us each at 2.4 GHz... measured below)
    const double idle_ms = argc > 4 ? atof(argv[4]) : 20.0;
    Host *h = nullptr;
    CK(hipHostMalloc((void **)&h, sizeof(Host), hipHostMallocDefault));
    memset((void *)h, 0, sizeof(Host));
    unsigned *go, *cnt, *small;
    CK(hipMalloc((void **)&go, njobs * sizeof(unsigned)));
    CK(hipMalloc((void **)&cnt, njobs * sizeof(unsigned)));
    CK(hipMalloc((void **)&small, sizeof(unsigned)));
    CK(hipMemset(go, 0, njobs * sizeof(unsigned)));
    CK(hipMemset(cnt, 0, njobs * sizeof(unsigned)));
    CK(hipMemset(small, 0, sizeof(unsigned)));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_resident, 512, 78 * 1024));
    printf("blocks %d  jobs %d  work %d  idle limit %.1f ms  occupancy %d blocks/CU\n", blocks, njobs, work, idle_ms, per_cu);
    const unsigned long long idle_ticks = (unsigned long long)(idle_ms * 1e5);  // 100 MHz
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_resident, dim3(blocks), dim3(512), 78 * 1024, s, h, go, cnt, njobs, work, idle_ticks);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, s));
    std::vector<double> rtt;
    float *hsrc = nullptr, *ddst = nullptr;
    CK(hipHostMalloc((void **)&hsrc, 100 << 10, hipHostMallocDefault));
    CK(hipMalloc((void **)&ddst, 100 << 10));
    double t_small = -1, t_copy = -1;
    bool dead = false;
    for (int j = 0; j < njobs && !dead; ++j) {
        if (j == njobs / 2) {  // something on another stream while the grid spins
            double a = now_us();
            hipLaunchKernelGGL(k_small, dim3(4), dim3(64), 0, s2, small);
            CK(hipStreamSynchronize(s2));
            t_small = now_us() - a;
            a = now_us();
            CK(hipMemcpyAsync(ddst, hsrc, 100 << 10, hipMemcpyHostToDevice, s2));
            CK(hipStreamSynchronize(s2));
            t_copy = now_us() - a;
        }
        // (a little host "think time" so that the grid is really idle-polling when the bell rings)
        const double w0 = now_us();
        while (now_us() - w0 < 30.0) {}
        const double a = now_us();
        std::atomic_thread_fence(std::memory_order_release);
        h->bell = (unsigned)(j + 1);
        std::atomic_thread_fence(std::memory_order_seq_cst);
        for (;;) {
            if (h->done >= (unsigned)(j + 1)) break;
            if (h->exited) { dead = true; break; }
            if (now_us() - a > 2e6) { dead = true; fprintf(stderr, "host: no answer to job %d after 2 s\n", j); break; }
        }
        rtt.push_back(now_us() - a);
    }
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::sort(rtt.begin(), rtt.end());
    printf("kernel %.3f ms, exited word %u, done %u\n", ms, h->exited, h->done);
    if (!rtt.empty())
        printf("round trip host->grid->host (us): min %.2f  median %.2f  p90 %.2f  max %.2f   (job itself: see device stamps)\n",
               rtt[0], rtt[rtt.size() / 2], rtt[rtt.size() * 9 / 10], rtt.back());
    double dsum = 0;
    int dn = 0;
    for (int j = 1; j < 64 && j < njobs; ++j)
        if (h->t_done[j] > h->t_seen[j]) { dsum += (double)(h->t_done[j] - h->t_seen[j]) / 100.0; ++dn; }
    if (dn) printf("device: leader saw the bell -> last block counted, mean %.2f us over %d jobs\n", dsum / dn, dn);
    printf("other stream while the grid spins: small kernel %.1f us, 100 KB H2D copy %.1f us\n", t_small, t_copy);
    return 0;
}
