// Microbenchmark (diagnostic): issue cost of buffer loads / stores in the shape the halo exchange of k_steps_resident
// uses them -- NI independent instructions per wave, all issued back to back, one wait at the end of a round --
// as a function of access width, active lanes and cache policy.  512-thread blocks, grid = 467 (two blocks per CU).
//   hipcc --offload-arch=gfx950 -O3 -o xch_rate2 xch_rate2.hip && ./xch_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

template <int W>
struct Word;
template <> struct Word<16> { typedef u4 T; };
template <> struct Word<8> { typedef u2 T; };
template <> struct Word<4> { typedef unsigned T; };

template <int W, int AUX>
__device__ __forceinline__ typename Word<W>::T ld(__amdgpu_buffer_rsrc_t rs, unsigned off)
{
    if constexpr (W == 16) return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, AUX);
    else if constexpr (W == 8) return __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, AUX);
    else return __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, AUX);
}
template <int W, int AUX>
__device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned a, unsigned b)
{
    if constexpr (W == 16) __builtin_amdgcn_raw_buffer_store_b128(u4{a, b, a, b}, rs, (int)off, 0, AUX);
    else if constexpr (W == 8) __builtin_amdgcn_raw_buffer_store_b64(u2{a, b}, rs, (int)off, 0, AUX);
    else __builtin_amdgcn_raw_buffer_store_b32(a, rs, (int)off, 0, AUX);
}
__device__ __forceinline__ unsigned fold(u4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
__device__ __forceinline__ unsigned fold(u2 v) { return v.x ^ v.y; }
__device__ __forceinline__ unsigned fold(unsigned v) { return v; }

// NL loads and NS stores per wave and round; ACT active lanes (the first ACT lanes of the wave)
template <int W, int AUX, int NL, int NS, int ACT>
__global__ __launch_bounds__(512, 4) void k(unsigned char *buf, unsigned bytes, int rounds, unsigned long long *out, unsigned *sink)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)bytes, 0x00020000);
    // like a tile of the 700^2 grid: rows 11200 B apart, planes 8 MB apart
    const unsigned base = (blockIdx.x % 13) * 56u * 16u + (blockIdx.x / 13) * 24u * 11200u + lane * (unsigned)W;
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        if (lane < ACT) {
#pragma unroll
            for (int i = 0; i < NS; ++i)
                st<W, AUX>(rs, base + (unsigned)(w + 8 * (i % 4)) * 11200u + (unsigned)(i / 4) * 0x800000u, acc, (unsigned)r);
            typename Word<W>::T v[NL > 0 ? NL : 1];
#pragma unroll
            for (int i = 0; i < NL; ++i)
                v[i] = ld<W, AUX>(rs, base + (unsigned)(w + 8 * (i % 4)) * 11200u + (unsigned)(i / 4) * 0x800000u + 0x4000000u);
#pragma unroll
            for (int i = 0; i < NL; ++i) acc ^= fold(v[i]);
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0xdeadbeef) *sink = acc;
}

unsigned char *buf;
unsigned long long *out;
unsigned *sink;
const unsigned bytes = 0x8000000u;

template <int W, int AUX, int NL, int NS, int ACT>
void run(const char *name, int grid)
{
    const int rounds = 200;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<W, AUX, NL, NS, ACT>), dim3(grid), dim3(512), 0, 0, buf, bytes, rounds, out, sink);
        (void)hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
    double s = 0, mx = 0;
    for (auto v : h) { s += v; if (v > mx) mx = v; }
    printf("  grid %3d  %-40s W=%2d aux=%2d loads=%2d stores=%2d lanes=%2d : mean %6.0f ns/round  max %6.0f\n", grid, name, W, AUX, NL, NS, ACT,
           s / grid * 10.0 / rounds, mx * 10.0 / rounds);
}

int main()
{
    (void)hipMalloc(&buf, bytes);
    (void)hipMemset(buf, 0, bytes);
    (void)hipMalloc(&out, 512 * 8);
    (void)hipMalloc(&sink, 4);
    for (int grid : {467, 8}) {
        run<16, 16, 1, 0, 64>("loads", grid);
        run<16, 16, 2, 0, 64>("loads", grid);
        run<16, 16, 4, 0, 64>("loads", grid);
        run<16, 16, 8, 0, 64>("loads", grid);
        run<16, 16, 12, 0, 64>("loads", grid);
        run<16, 16, 16, 0, 64>("loads", grid);
        run<16, 16, 4, 0, 8>("loads 8 lanes", grid);
        run<16, 16, 8, 0, 8>("loads 8 lanes", grid);
        run<16, 16, 12, 0, 8>("loads 8 lanes", grid);
        run<16, 16, 16, 0, 8>("loads 8 lanes", grid);
        run<8, 16, 12, 0, 64>("loads b64", grid);
        run<8, 16, 24, 0, 64>("loads b64", grid);
        run<8, 16, 12, 0, 8>("loads b64 8 lanes", grid);
        run<4, 16, 12, 0, 64>("loads b32", grid);
        run<4, 16, 24, 0, 64>("loads b32", grid);
        run<4, 16, 24, 0, 8>("loads b32 8 lanes", grid);
        run<16, 0, 12, 0, 64>("loads plain", grid);
        run<16, 0, 12, 0, 8>("loads plain 8 lanes", grid);
        run<16, 16, 0, 1, 64>("stores", grid);
        run<16, 16, 0, 4, 64>("stores", grid);
        run<16, 16, 0, 8, 64>("stores", grid);
        run<16, 16, 0, 12, 64>("stores", grid);
        run<16, 16, 0, 12, 8>("stores 8 lanes", grid);
        run<16, 16, 0, 8, 8>("stores 8 lanes", grid);
        run<8, 16, 0, 12, 64>("stores b64", grid);
        run<4, 16, 0, 12, 64>("stores b32", grid);
        run<4, 16, 0, 24, 64>("stores b32", grid);
        run<16, 0, 0, 12, 64>("stores plain", grid);
        run<16, 0, 0, 12, 8>("stores plain 8 lanes", grid);
        run<16, 16, 12, 12, 64>("both", grid);
        run<16, 16, 12, 12, 8>("both 8 lanes", grid);
        run<16, 16, 8, 8, 64>("both", grid);
        run<16, 16, 8, 8, 8>("both 8 lanes", grid);
        run<16, 16, 4, 4, 64>("both", grid);
    }
    return 0;
}
