// Micro-benchmark: fp32 VALU FMA issue rate on gfx950 -- v_fma_f32 vs v_pk_fma_f32, with VGPR or
// SGPR multiplicand, at 1/2/4 waves per SIMD.  Development aid (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float sw) {
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    float a[16];
    float2v p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = i; p[i] = (float2v){(float)i, (float)i + 0.5f}; }
    float2v x2 = {x, x + 1.f}, y2 = {y, y};
    float2v sw2 = {sw, sw * 2.f};
    float2v yy[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) yy[i] = (float2v){y + i, y - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(x2), "v"(y2));
                if (MODE == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sw), "v"(y));
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(x2), "v"(y2));
                if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "s"(sw2), "v"(y2));
                if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(sw2), "v"(y2));
                if (MODE == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(sw2), "v"(yy[i & 7]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu) {
    const int iters = 20000;
    const int blocks = 256 * blocks_per_cu;
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.5f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double fma_per_instr = (MODE == 1 || MODE >= 3) ? 2.0 : 1.0;
    const double flops = 2.0 * fma_per_instr * 64.0 * (double)iters * 64.0 * (blocks * 4.0);
    printf("%-34s waves/SIMD=%d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, best, flops / best / 1e9);
    hipFree(out);
}

int main() {
    for (int b : {1, 2, 4}) {
        run<0>("v_fma_f32 (vgpr,vgpr)", b);
        run<2>("v_fma_f32 (sgpr,vgpr)", b);
        run<1>("v_pk_fma_f32", b);
        run<3>("v_pk_fma_f32 op_sel_hi bcast", b);
        run<4>("v_pk_fma_f32 sgpr-pair bcast-lo", b);
        run<5>("v_pk_fma_f32 sgpr-pair, v bcast", b);
        run<6>("v_pk_fma_f32 sgpr-pair, 8 v srcs", b);
    }
    return 0;
}
