// What the TAIL's memory skeleton can reach: planar fp32 (3 planes) -> interleaved HWC fp32, 100 MP, for several tile shapes
// (threads own 4 consecutive pixels x Q rows, like tail_kernel).  hipcc --offload-arch=gfx950 -O3 -o tile_copy tile_copy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BX, int BY, int Q>
__global__ __launch_bounds__(BX* BY) void tile_copy(const float* __restrict__ src, float* __restrict__ dst, int W, int H, long long plane) {
    const int tx = threadIdx.x % BX, ty = threadIdx.x / BX;
    const int gx = (blockIdx.x * BX + tx) * 4;
    float4 r[Q], g[Q], b[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int gy = (blockIdx.y * BY + ty) * Q + q;
        const long long o = (long long)gy * W + gx;
        r[q] = *reinterpret_cast<const float4*>(src + o);
        g[q] = *reinterpret_cast<const float4*>(src + plane + o);
        b[q] = *reinterpret_cast<const float4*>(src + 2 * plane + o);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int gy = (blockIdx.y * BY + ty) * Q + q;
        float4* o4 = reinterpret_cast<float4*>(dst + ((long long)gy * W + gx) * 3);
        o4[0] = make_float4(r[q].x, g[q].x, b[q].x, r[q].y);
        o4[1] = make_float4(g[q].y, b[q].y, r[q].z, g[q].z);
        o4[2] = make_float4(b[q].z, r[q].w, g[q].w, b[q].w);
    }
}
// The same copy with every wave store instruction writing whole 256-byte runs: a lane's 3 float4 (48-byte lane stride in memory) go
// through a wave-private LDS row image first (ds_write_b128 x 3 at the 48-byte stride, ds_read_b128 x 3 at a 16-byte lane stride),
// so that instruction j of a row's 16 lanes covers bytes [256 j, 256 j + 256) of the row segment instead of every third float4 of
// all 768 ("wavefront shuffles" for the interleaved store).  BX = 16 only (the tail's tile).
template <int BY, int Q>
__global__ __launch_bounds__(16 * BY) void tile_copy_lds(const float* __restrict__ src, float* __restrict__ dst, int W, int H, long long plane) {
    constexpr int BX = 16;
    __shared__ float4 stage[BY * 48];  // one 768-byte row image per (wave row) = per 16 lanes
    const int tx = threadIdx.x % BX, ty = threadIdx.x / BX;
    const int gx = (blockIdx.x * BX + tx) * 4;
    float4 r[Q], g[Q], b[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int gy = (blockIdx.y * BY + ty) * Q + q;
        const long long o = (long long)gy * W + gx;
        r[q] = *reinterpret_cast<const float4*>(src + o);
        g[q] = *reinterpret_cast<const float4*>(src + plane + o);
        b[q] = *reinterpret_cast<const float4*>(src + 2 * plane + o);
    }
    float4* row = stage + ty * 48;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int gy = (blockIdx.y * BY + ty) * Q + q;
        row[3 * tx + 0] = make_float4(r[q].x, g[q].x, b[q].x, r[q].y);
        row[3 * tx + 1] = make_float4(g[q].y, b[q].y, r[q].z, g[q].z);
        row[3 * tx + 2] = make_float4(b[q].z, r[q].w, g[q].w, b[q].w);
        __builtin_amdgcn_wave_barrier();
        float4* o4 = reinterpret_cast<float4*>(dst + ((long long)gy * W + (blockIdx.x * BX) * 4) * 3);
        const float4 v0 = row[tx], v1 = row[16 + tx], v2 = row[32 + tx];
        __builtin_amdgcn_wave_barrier();
        o4[tx] = v0;
        o4[16 + tx] = v1;
        o4[32 + tx] = v2;
    }
}

// The mirror image for an interleaved SOURCE (the front kernel's HWC loads): direct 3 x float4 at the 48-byte lane stride against
// whole 256-byte runs through LDS; planar out.
template <int BY, int Q, bool LDS>
__global__ __launch_bounds__(16 * BY) void tile_copy_in(const float* __restrict__ src, float* __restrict__ dst, int W, int H, long long plane) {
    constexpr int BX = 16;
    __shared__ float4 stage[BY * 48];
    const int tx = threadIdx.x % BX, ty = threadIdx.x / BX;
    const int gx = (blockIdx.x * BX + tx) * 4;
    float4* row = stage + ty * 48;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int gy = (blockIdx.y * BY + ty) * Q + q;
        float4 a0, a1, a2;
        if (LDS) {
            const float4* i4 = reinterpret_cast<const float4*>(src + ((long long)gy * W + (blockIdx.x * BX) * 4) * 3);
            const float4 v0 = i4[tx], v1 = i4[16 + tx], v2 = i4[32 + tx];
            row[tx] = v0, row[16 + tx] = v1, row[32 + tx] = v2;
            __builtin_amdgcn_wave_barrier();
            a0 = row[3 * tx], a1 = row[3 * tx + 1], a2 = row[3 * tx + 2];
            __builtin_amdgcn_wave_barrier();
        } else {
            const float4* i4 = reinterpret_cast<const float4*>(src + ((long long)gy * W + gx) * 3);
            a0 = i4[0], a1 = i4[1], a2 = i4[2];
        }
        const long long o = (long long)gy * W + gx;
        *reinterpret_cast<float4*>(dst + o) = make_float4(a0.x, a0.w, a1.z, a2.y);
        *reinterpret_cast<float4*>(dst + plane + o) = make_float4(a0.y, a1.x, a1.w, a2.z);
        *reinterpret_cast<float4*>(dst + 2 * plane + o) = make_float4(a0.z, a1.y, a2.x, a2.w);
    }
}

// plain float4 copy (the linear reference)
__global__ __launch_bounds__(256) void linear_copy(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

template <class F>
static void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 7; ++r) {
        (void)hipEventRecord(e0, 0);
        launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-34s %7.3f ms  %7.1f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9);
}
#define RUN(BX, BY, Q) timeit("tile " #BX "x4 wide, " #BY "x" #Q " rows", 2.0 * bytes, [&] { \
    hipLaunchKernelGGL((tile_copy<BX, BY, Q>), dim3(W / (4 * BX), H / (BY * Q)), dim3(BX * BY), 0, 0, a, b, W, H, (long long)W * H); })
int main() {
    const int W = 12288, H = 8192;
    const double bytes = (double)W * H * 12;
    float *a, *b;
    if (hipMalloc(&a, (size_t)bytes) != hipSuccess || hipMalloc(&b, (size_t)bytes) != hipSuccess) return 1;
    (void)hipMemset(a, 1, (size_t)bytes);
    RUN(16, 32, 2);   // the tail's 64 x 64 tile, 512 threads
    RUN(32, 16, 2);   // 128 x 32
    RUN(64, 8, 2);    // 256 x 16
    RUN(64, 8, 1);    // 256 x 8, 1 row per thread
    RUN(64, 4, 1);    // 256 x 4, 256 threads
    RUN(16, 16, 4);   // 64 x 64, 256 threads, 4 rows per thread
    RUN(16, 32, 1);   // 64 x 32
    RUN(256, 1, 1);   // 1024 x 1
    RUN(128, 4, 1);   // 512 x 4
#define RUNL(BY, Q) timeit("tile 16x4 wide, " #BY "x" #Q " rows, stores through LDS", 2.0 * bytes, [&] { \
    hipLaunchKernelGGL((tile_copy_lds<BY, Q>), dim3(W / 64, H / (BY * Q)), dim3(16 * BY), 0, 0, a, b, W, H, (long long)W * H); })
    RUNL(32, 2);
    RUNL(16, 4);
    RUNL(32, 1);
    RUNL(16, 2);
#define RUNI(BY, Q, L) timeit("HWC in: tile 16x4 wide, " #BY "x" #Q " rows, loads " #L, 2.0 * bytes, [&] { \
    hipLaunchKernelGGL((tile_copy_in<BY, Q, L>), dim3(W / 64, H / (BY * Q)), dim3(16 * BY), 0, 0, a, b, W, H, (long long)W * H); })
    RUNI(32, 2, false);
    RUNI(32, 2, true);
    RUNI(16, 4, false);
    RUNI(16, 4, true);
    timeit("linear float4 copy", 2.0 * bytes, [&] {
        hipLaunchKernelGGL(linear_copy, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, (const float4*)a, (float4*)b, (long long)(bytes / 16)); });
    return 0;
}
