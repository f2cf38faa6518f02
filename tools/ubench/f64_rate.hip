// Micro-benchmark: fp64 VALU issue rates on gfx950 (v_add_f64, v_mul_f64, v_fma_f64, a 3:1:1 mix like the FFT column pass's)
// with 16 independent accumulators per lane at 1 / 2 / 3 / 4 waves per SIMD.  Development aid, not part of the product.
// hipcc --offload-arch=gfx950 -O3 -o f64_rate f64_rate.hip && ./f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0000001;
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                if (MODE == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(y));
                if (MODE == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                if (MODE == 3) {  // 3 adds : 1 mul : 1 fma over 5 consecutive accumulators
                    const int m = (r * 16 + i) % 5;
                    if (m < 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                    else if (m == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(y));
                    else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                }
                if (MODE == 4) asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15]));  // a DFT-like web
                if (MODE == 5) asm volatile("v_fma_f64 %0, %1, 1.0, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15]));  // the same adds as FMAs
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name) {
    const int iters = 4000;
    printf("%-44s", name);
    for (int wpc : {1, 2, 3, 4}) {
        const int blocks = 256 * wpc;
        double* out;
        (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
        k<MODE><<<blocks, 256>>>(out, 10);
        (void)hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0, 0);
            k<MODE><<<blocks, 256>>>(out, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        // wave-instructions per second per SIMD, as a fraction of one per 4 clocks at 2.4 GHz
        const double instr = (double)blocks * 4 /*waves*/ * iters * 64.0, per_simd = instr / 1024.0 / (best * 1e-3);
        printf("  %6.2f G/s/SIMD (%4.2f)", per_simd / 1e9, per_simd / 0.6e9);
        (void)hipFree(out);
    }
    printf("\n");
}
int main() {
    printf("fp64 wave-instructions per second per SIMD at 1 / 2 / 3 / 4 waves per SIMD (in brackets: fraction of one per 4 clocks at 2.4 GHz)\n");
    run<0>("v_add_f64, 16 independent");
    run<1>("v_mul_f64, 16 independent");
    run<2>("v_fma_f64, 16 independent");
    run<3>("3 add : 1 mul : 1 fma");
    run<4>("v_add_f64 web (dst = a[i+1] + a[i+2])");
    run<5>("v_fma_f64 x, 1.0, y web");
    return 0;
}
