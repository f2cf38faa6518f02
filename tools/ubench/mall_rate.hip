// Streaming rates of one MI355X as a function of the footprint: buffers that stay in the 256 MB Infinity Cache between launches
// against buffers far beyond it; copy (1 read : 1 write), read-only, write-only, and 1 read : 2 writes (the shape of FFT pass 1).
// hipcc --offload-arch=gfx950 -O3 -o mall_rate mall_rate.hip && ./mall_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ s, float4* __restrict__ d, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ s, float4* __restrict__ d, long long n) {
    const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
    float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (i + 256 * u < n) {
            const float4 v = s[i + 256 * u];
            a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
        }
    if (a.x + a.y + a.z + a.w == 1.2345e30f) d[0] = a;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ d, long long n, float v) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void k_r1w2(const float4* __restrict__ s, float4* __restrict__ d, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float4 v = s[i];
        d[2 * i] = v, d[2 * i + 1] = v;
    }
}
template <class F>
static float timeit(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) launch();
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 9; ++r) {
        (void)hipEventRecord(e0, 0);
        launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}
int main() {
    const long long cap = 3LL << 30;
    float4 *a, *b;
    if (hipMalloc(&a, cap) != hipSuccess || hipMalloc(&b, cap) != hipSuccess) return 1;
    (void)hipMemset(a, 0, cap), (void)hipMemset(b, 0, cap);
    printf("%10s %12s %12s %12s %12s   (GB/s of bytes read + written; REP launches back to back per timing)\n", "MB/buffer", "copy", "read", "write", "1r:2w");
    for (long long mb : {16LL, 32LL, 48LL, 64LL, 96LL, 128LL, 192LL, 384LL, 1024LL}) {
        const long long bytes = mb << 20, n = bytes / 16;
        const int rep = (int)(2048 / mb) > 1 ? (int)(2048 / mb) : 1;
        const unsigned g = (unsigned)((n + 255) / 256), g4 = (unsigned)((n + 1023) / 1024);
        const float tc = timeit([&] { for (int r = 0; r < rep; ++r) k_copy<<<g, 256>>>(a, b, n); }) / rep;
        const float tr = timeit([&] { for (int r = 0; r < rep; ++r) k_read<<<g4, 256>>>(a, b, n); }) / rep;
        const float tw = timeit([&] { for (int r = 0; r < rep; ++r) k_write<<<g, 256>>>(b, n, 1.f); }) / rep;
        const float t12 = timeit([&] { for (int r = 0; r < rep; ++r) k_r1w2<<<g, 256>>>(a, b, n); }) / rep;
        printf("%10lld %12.0f %12.0f %12.0f %12.0f\n", mb, 2.0 * bytes / tc / 1e6, 1.0 * bytes / tr / 1e6, 1.0 * bytes / tw / 1e6, 3.0 * bytes / t12 / 1e6);
    }
    return 0;
}
