// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths the integrator uses (MI355X_MICROARCH.md,
// HBM section: FETCH_SIZE is exact/half/uncalibrated depending on the access width -- calibrate on a known byte count
// in your own pattern).  Reads / writes a 1 GiB buffer (>> 256 MiB Infinity Cache) with 4 B and 16 B per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void read_dword(const float *__restrict__ in, float *__restrict__ out, size_t n)
{
    float acc = 0.0f;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) acc += in[q];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void read_dwordx4(const float4 *__restrict__ in, float *__restrict__ out, size_t n4)
{
    float acc = 0.0f;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[q];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void write_dword(float *__restrict__ out, size_t n)
{
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) out[q] = 1.0f;
}
__global__ void write_dwordx4(float4 *__restrict__ out, size_t n4)
{
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x)
        out[q] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / 4;
    float *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read_dword, dim3(2048), dim3(256), 0, 0, a, b, n);
        hipLaunchKernelGGL(read_dwordx4, dim3(2048), dim3(256), 0, 0, (const float4 *)a, b, n / 4);
        hipLaunchKernelGGL(write_dword, dim3(2048), dim3(256), 0, 0, b, n);
        hipLaunchKernelGGL(write_dwordx4, dim3(2048), dim3(256), 0, 0, (float4 *)b, n / 4);
    }
    hipDeviceSynchronize();
    printf("each kernel moves %zu bytes\n", bytes);
    return 0;
}
