// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this repo uses:
// a 1 GiB streaming copy with 4-byte-per-lane loads (the stencil tile fill) and with 16-byte-per-lane
// loads (the pointwise kernels).  Run under rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy_dword(const float* __restrict__ a, float* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void copy_dwordx4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    const size_t n = 256u << 20;  // floats = 1 GiB
    float *a, *b;
    if (hipMalloc(&a, n * 4) != hipSuccess || hipMalloc(&b, n * 4) != hipSuccess) return 1;
    (void)hipMemset(a, 1, n * 4);
    hipLaunchKernelGGL(copy_dword, dim3(8192), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(copy_dwordx4, dim3(8192), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4);
    (void)hipDeviceSynchronize();
    printf("copied 2 x %zu bytes\n", n * 4);
    return 0;
}
