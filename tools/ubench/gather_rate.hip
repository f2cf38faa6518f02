// gather_rate.hip -- how fast can a CU gather 16-byte LUT texels?  (MI355X; build: hipcc -O3 --offload-arch=gfx950)
// Each lane of every wave does ITERS dependent-address-free gathers from a table of `texels` float4 entries:
//   mode 0: every lane a random texel (all 64 lanes of an instruction in different cache lines)
//   mode 1: lane quads read 4 neighbouring texels (one 64-byte run per quad: 16 runs per instruction)
//   mode 2: every lane the same texel (broadcast)
//   mode 3: random texels of a table held in LDS (ds_read_b128)
//   mode 4: random 8-byte gathers (global)
//   mode 5: mode 0 with nontemporal loads (global_load_dwordx4 ... nt)
//   mode 6: random 4-byte gathers (global_load_dword): is the rate per LANE-ADDRESS or per byte?
//   mode 7: a texel as three 4-byte gathers from three planes of `texels` floats (what a planar LUT would cost: counted as ONE
//           gather per three loads, to compare with mode 0 directly)
//   mode 8: LDS 4-byte gathers (ds_read_b32)
// Round 4: the tetrahedral 3-D LUT's access pattern itself, texel-major against a CELL-MAJOR copy:
//   cell_kernel<false>: a pixel = 4 corners of one cell of an n^3 texel table (c000, c100, c110, c111: 4 different 128-byte lines)
//   cell_kernel<true>:  a pixel = 4 of the 8 corners of one cell of a cell-major copy (8 x 16 B = ONE 128-byte line per cell;
//                       (n - 1)^3 x 128 B: 4 MB for n = 33 against 575 KB)
// each with cells drawn uniformly from the table and from a hot set of 300 cells (the benchmark frame's densities are strongly
// correlated and cover a few hundred cells).
// Prints wave-instructions per microsecond per CU and lanes per clock per CU (at the reported clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned rnd(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}

template <int MODE>
__global__ __launch_bounds__(512) void gather_kernel(const float4* __restrict__ table, unsigned texels, int iters, float* out) {
    extern __shared__ float4 lds[];
    const int tid = threadIdx.x;
    if (MODE == 3 || MODE == 8) {
        for (unsigned i = tid; i < texels; i += blockDim.x) lds[i] = table[i];
        __syncthreads();
    }
    unsigned s = (blockIdx.x * blockDim.x + tid) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; it += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned idx = rnd(s) % texels;
            if (MODE == 1) idx = ((idx & ~3u) + (tid & 3)) % texels;
            if (MODE == 2) idx = __builtin_amdgcn_readfirstlane(idx);
            if (MODE == 3)
                v[k] = lds[idx];
            else if (MODE == 4) {
                const float2 t = reinterpret_cast<const float2*>(table)[idx * 2];
                v[k] = make_float4(t.x, t.y, 0.f, 0.f);
            } else if (MODE == 6) {
                v[k] = make_float4(reinterpret_cast<const float*>(table)[idx], 0.f, 0.f, 0.f);
            } else if (MODE == 7) {
                const float* p = reinterpret_cast<const float*>(table);
                v[k] = make_float4(p[idx], p[texels + idx], p[2 * texels + idx], 0.f);
            } else if (MODE == 8) {
                v[k] = make_float4(reinterpret_cast<const float*>(lds)[idx], 0.f, 0.f, 0.f);
            } else if (MODE == 5) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(table) + idx);
                v[k] = make_float4(t.x, t.y, t.z, t.w);
            } else
                v[k] = table[idx];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    if (acc == 1.2345e30f) out[0] = acc;
}

template <bool CELL_MAJOR>
__global__ __launch_bounds__(512) void cell_kernel(const float4* __restrict__ table, int n, unsigned hot, int iters, float* out) {
    const int tid = threadIdx.x;
    unsigned s = (blockIdx.x * blockDim.x + tid) * 2654435761u + 12345u;
    const unsigned cells = (unsigned)(n - 1) * (n - 1) * (n - 1);
    float acc = 0.f;
    for (int it = 0; it < iters; it += 2) {  // two pixels' gathers in flight together, like the tail kernel
        float4 v[8];
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            unsigned cell = rnd(s) % (hot ? hot : cells);
            if (hot) cell = (cell * 2654435761u) % cells;  // the hot cells are scattered over the table
            const unsigned r = cell / ((n - 1) * (n - 1)), g = (cell / (n - 1)) % (n - 1), b = cell % (n - 1);
            if (CELL_MAJOR) {
                const float4* c = table + (size_t)cell * 8;
                v[4 * px] = c[0], v[4 * px + 1] = c[4], v[4 * px + 2] = c[6], v[4 * px + 3] = c[7];
            } else {
                const float4* c = table + ((size_t)r * n + g) * n + b;
                v[4 * px] = c[0], v[4 * px + 1] = c[(size_t)n * n], v[4 * px + 2] = c[(size_t)n * n + n], v[4 * px + 3] = c[(size_t)n * n + n + 1];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    if (acc == 1.2345e30f) out[0] = acc;
}

template <bool CELL_MAJOR>
void run_cells(const char* name, const float4* table, int n, unsigned hot, float* out) {
    const int iters = 256, blocks = 256 * 4, threads = 512;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(cell_kernel<CELL_MAJOR>, dim3(blocks), dim3(threads), 0, 0, table, n, hot, iters, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double px = (double)blocks * threads * iters;
    printf("%-58s %7.3f ms  %7.2f G pixels/s  = %5.2f lane-gathers/clk/CU at 2.1 GHz  (24 MP: %.3f ms)\n", name, ms, px / ms * 1e-6,
           4.0 * px / (ms * 1e-3) / 256 / 2.1e9, 24e6 / (px / ms));
}

template <int MODE>
void run(const char* name, const float4* table, unsigned texels, float* out) {
    const int iters = 512, blocks = 256 * 4, threads = 512;
    const size_t lds = (MODE == 3 || MODE == 8) ? texels * sizeof(float4) : 0;
    hipFuncSetAttribute(reinterpret_cast<const void*>(gather_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(gather_kernel<MODE>, dim3(blocks), dim3(threads), lds, 0, table, texels, iters, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double lanes = (double)blocks * threads * iters;
    printf("%-46s table %7.0f KB: %7.3f ms  %8.2f G lane-gathers/s  = %5.2f lanes/clk/CU at 2.1 GHz\n", name, texels * 16.0 / 1024, ms,
           lanes / ms * 1e-6, lanes / (ms * 1e-3) / 256 / 2.1e9);
}

int main() {
    const unsigned sizes[] = {4096, 35937, 1u << 20};  // 64 KB (2-D LUT), 575 KB (33^3 3-D LUT), 16 MB
    float* out;
    hipMalloc(&out, 64);
    for (unsigned texels : sizes) {
        std::vector<float4> h(texels, make_float4(1, 2, 3, 4));
        float4* d;
        hipMalloc(&d, texels * sizeof(float4));
        hipMemcpy(d, h.data(), texels * sizeof(float4), hipMemcpyHostToDevice);
        run<0>("global 16 B, random texel per lane", d, texels, out);
        run<1>("global 16 B, quads read 4 neighbouring texels", d, texels, out);
        run<2>("global 16 B, one texel per wave (broadcast)", d, texels, out);
        run<4>("global 8 B, random per lane", d, texels, out);
        run<5>("global 16 B, random, nontemporal (nt)", d, texels, out);
        run<6>("global 4 B, random per lane", d, texels, out);
        run<7>("global 3 x 4 B from three planes = one texel", d, texels, out);
        if (texels * 16 <= 128 * 1024) run<8>("LDS 4 B (ds_read_b32), random per lane", d, texels, out);
        if (texels * 16 <= 128 * 1024) run<3>("LDS 16 B (ds_read_b128), random texel per lane", d, texels, out);
        hipFree(d);
    }
    {   // the tetrahedral pattern: texel-major 33^3 against a cell-major copy of it
        const int n = 33;
        const size_t texels = (size_t)n * n * n, cell_floats4 = (size_t)(n - 1) * (n - 1) * (n - 1) * 8;
        std::vector<float4> h(std::max(texels, cell_floats4), make_float4(1, 2, 3, 4));
        float4 *dt, *dc;
        hipMalloc(&dt, texels * sizeof(float4));
        hipMalloc(&dc, cell_floats4 * sizeof(float4));
        hipMemcpy(dt, h.data(), texels * sizeof(float4), hipMemcpyHostToDevice);
        hipMemcpy(dc, h.data(), cell_floats4 * sizeof(float4), hipMemcpyHostToDevice);
        run_cells<false>("texel-major 33^3 (575 KB), 4 corners, cells uniform", dt, n, 0, out);
        run_cells<true>("cell-major 32^3 x 128 B (4 MB), 4 of 8 corners, uniform", dc, n, 0, out);
        run_cells<false>("texel-major 33^3, 4 corners, 300 hot cells", dt, n, 300, out);
        run_cells<true>("cell-major, 4 of 8 corners, 300 hot cells", dc, n, 300, out);
        hipFree(dt);
        hipFree(dc);
    }
    return 0;
}
