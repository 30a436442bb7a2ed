// Stress test (diagnostic): are 16-byte agent-scope (sc1) buffer accesses to 16-byte-aligned addresses ever observed
// torn on gfx950?  Writer blocks rewrite granules {x, f(x), ~x, x ^ K} in a tight loop, reader blocks -- on every XCD,
// L1-warm, while other blocks stream -- load them and check that the four dwords belong to one x.
//   hipcc --offload-arch=gfx950 -O3 -o tear16 tear16.hip && ./tear16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u4 make(unsigned x) { return u4{x, x * 2654435761u + 1u, ~x, x ^ 0x9e3779b9u}; }
__device__ __forceinline__ bool good(u4 v) { return v.y == v.x * 2654435761u + 1u && v.z == ~v.x && v.w == (v.x ^ 0x9e3779b9u); }

// NG granules per lane group; block b < nwriters writes, the others read; stride: distance between a lane's granules
__global__ __launch_bounds__(256) void k(unsigned char *buf, unsigned bytes, int iters, int nwriters, unsigned lane_stride,
                                         unsigned long long *bad, unsigned long long *seen, unsigned *example)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)bytes, 0x00020000);
    const unsigned slot = (blockIdx.x % nwriters) * 256u + threadIdx.x;  // writer w and readers w, w + nwriters, ... share granules
    const unsigned off = (unsigned)(((unsigned long long)slot * lane_stride) % bytes) & ~15u;
    if ((int)blockIdx.x < nwriters) {
        for (int i = 1; i <= iters; ++i) {
            __builtin_amdgcn_raw_buffer_store_b128(make((unsigned)i * 977u + slot), rs, (int)off, 0, 16);
            if ((i & 7) == 0) __builtin_amdgcn_s_waitcnt(0);
        }
    } else {
        unsigned long long nb = 0, changes = 0;
        unsigned last = 0;
        for (int i = 0; i < iters; ++i) {
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);
            if (v.x == 0 && v.y == 0 && v.z == 0 && v.w == 0) continue;  // not written yet
            if (!good(v)) {
                ++nb;
                example[0] = v.x; example[1] = v.y; example[2] = v.z; example[3] = v.w;
            }
            if (v.x != last) ++changes;
            last = v.x;
        }
        if (nb) atomicAdd(bad, nb);
        atomicAdd(seen, changes);
    }
}

int main()
{
    const unsigned bytes = 64u << 20;
    unsigned char *buf;
    unsigned long long *bad, *seen;
    unsigned *ex;
    hipMalloc(&buf, bytes);
    hipMalloc(&bad, 8);
    hipMalloc(&seen, 8);
    hipMalloc(&ex, 16);
    for (unsigned stride : {16u, 64u, 11200u}) {  // dense granules, one per 64-B line, one per grid row
        for (int nw : {64, 256}) {
            hipMemset(buf, 0, bytes);
            hipMemset(bad, 0, 8);
            hipMemset(seen, 0, 8);
            hipMemset(ex, 0, 16);
            const int iters = 200000;
            hipLaunchKernelGGL(k, dim3(nw * 4), dim3(256), 0, 0, buf, bytes, iters, nw, stride, bad, seen, ex);
            hipError_t e = hipDeviceSynchronize();
            unsigned long long hb = 0, hs = 0;
            unsigned hex[4];
            hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
            hipMemcpy(&hs, seen, 8, hipMemcpyDeviceToHost);
            hipMemcpy(hex, ex, 16, hipMemcpyDeviceToHost);
            printf("stride %5u B, %3d writer blocks, %d reader blocks: %llu loads checked, %llu distinct values observed, torn %llu"
                   " (example %08x %08x %08x %08x)  [%s]\n", stride, nw, nw * 3, (unsigned long long)nw * 3 * 256 * iters, hs, hb,
                   hex[0], hex[1], hex[2], hex[3], hipGetErrorString(e));
        }
    }
    return 0;
}
