// r2f_fft2d.hip -- overlap-save FFT correlation with the whole window ON CHIP: the form of a stencil whose tap box fits a
// 128 x 128 window with room to spare (the MTF stencil S5, 35 x 35 at 100 MP: 94 x 92 valid outputs per window).
//
// The three-pass form of r2f_fft.hip sends every window through a global scratch image four times (203 B per pixel of the
// 250 the 100 MP frame moves); it exists because a 256 x 512 complex128 window is 2 MB.  A 128 x 128 window is
// 256 KB -- exactly what 1 024 threads hold in 16 complex128 registers each.  So here one workgroup (16 waves, one per CU)
// keeps a PAIR of windows of one channel (real + imaginary part of one complex image, as in r2f_fft.hip) in registers from
// the input load to the output store, and LDS (152 of the 160 KB) is only the exchange medium between the register DFTs:
//
//   HBM: window floats in (once, coalesced, reflect-101 at the frame edges)            12 B/px/channel x overlap, L2-served
//        valid outputs out (once, coalesced)
//   no scratch image, no rounding between the passes (the scratch of the three-pass MTF is complex64: this form is MORE exact)
//
// A 128-point line transform is DFT-16 (registers) -> twiddles -> exchange inside the line's 8 lanes -> 2 x DFT-8 (registers):
//   n = n1 + 8 n2, f = 16 f1 + f2:   X[16 f1 + f2] = sum_n1 W8^(n1 f1) [ W128^(n1 f2) sum_n2 W16^(n2 f2) x[n1 + 8 n2] ]
// Layout A (before a line transform): the thread (line, n1) holds n2 = 0..15.  Layout B (after it): the thread (line, j) holds
// f2 in {j, j + 8}, f1 = 0..7 -- i.e. f = j + 8 (2 f1 + h): read as i1 + 8 i2 that IS layout A of the inverse transform, so
// the column pass goes forward, multiplies by the kernel spectrum and comes back without an exchange in between.
// Per window pair: rows forward, 2-D exchange (row threads -> column threads), columns forward, x spectrum, columns inverse,
// 2-D exchange back, rows inverse: four exchanges inside lines (wave-private tiles, no block barrier) and two across the block.
//
// LDS layouts, each conflict-free for ds_write_b64 (groups of 16 lanes) and ds_read_b64 (groups of 32 lanes):
//   line tiles   16 x 8 doubles at pitch 9, one tile per line slot (tid / 8) at stride 152 doubles
//   2-D image    128 rows at pitch 132 doubles; a wave's 8 row-threads own rows 2 apart, so that neighbouring lines of a lane
//                group sit a multiple of 8 doubles apart (132 x 2 = 8 mod 32)
//   fp32 staging two 128 x 128 float images at pitch 132 for the coalesced loads and stores
// Real parts, then imaginary parts go through the same bytes.
#include "r2f_launch.h"
#include "r2f_fft_math.h"

#include "../../include/r2f.h"

#ifndef R2F_FFT2D_EXP
#define R2F_FFT2D_EXP 0  // development switch: 1 no window loads, 2 no spectrum loads, 4 no output stores, 8 no line transforms, 16 no 2-D exchanges
#endif

namespace r2f {

namespace {

constexpr int kW = kFft2dN;      // 128: window side
constexpr int kThreads = 512, kSlices = 2;  // two 16-element slices of the window per thread
constexpr int kPitch2D = 132;    // doubles
constexpr int kLinePitch = 9, kLineStride = 152;
constexpr int kPitchF = 132;     // floats
constexpr size_t kLdsBytes = (size_t)kW * kLineStride * sizeof(double);  // 155 648: the line tiles are the largest layout
static_assert(kLdsBytes >= (size_t)kW * kPitch2D * sizeof(double) && kLdsBytes >= (size_t)2 * kW * kPitchF * sizeof(float), "LDS layouts");

// 8-point DFT of v[OFF .. OFF + 7], natural order in and out
template <bool INV, int OFF>
__device__ __forceinline__ void dft8(cplx (&v)[16]) {
    constexpr double h = 0.70710678118654752440;
    dft4<INV>(v[OFF], v[OFF + 2], v[OFF + 4], v[OFF + 6]);      // E[0..3] at OFF + 0, 2, 4, 6
    dft4<INV>(v[OFF + 1], v[OFF + 3], v[OFF + 5], v[OFF + 7]);  // O[0..3] at OFF + 1, 3, 5, 7
    const cplx o0 = v[OFF + 1];
    const cplx o1 = ctw<INV>(v[OFF + 3], make_double2(h, -h));
    const cplx o2 = INV ? make_double2(-v[OFF + 5].y, v[OFF + 5].x) : make_double2(v[OFF + 5].y, -v[OFF + 5].x);  // -+ i
    const cplx o3 = ctw<INV>(v[OFF + 7], make_double2(-h, -h));
    const cplx e0 = v[OFF], e1 = v[OFF + 2], e2 = v[OFF + 4], e3 = v[OFF + 6];
    v[OFF + 0] = cadd(e0, o0), v[OFF + 4] = csub(e0, o0);
    v[OFF + 1] = cadd(e1, o1), v[OFF + 5] = csub(e1, o1);
    v[OFF + 2] = cadd(e2, o2), v[OFF + 6] = csub(e2, o2);
    v[OFF + 3] = cadd(e3, o3), v[OFF + 7] = csub(e3, o3);
}

// exp(-2 pi i x / 128), x < 8
__device__ __forceinline__ cplx w128(int x) {
    constexpr double c[8] = {1.0, 0.99879545620517239271, 0.99518472667219688624, 0.98917650996478097345, 0.98078528040323044913,
                             0.97003125319454399260, 0.95694033573220886494, 0.94154406518302077841};
    constexpr double s[8] = {0.0, 0.04906767432741801425, 0.09801714032956060199, 0.14673047445536175166, 0.19509032201612826785,
                             0.24298017990326388995, 0.29028467725446236764, 0.33688985339222005069};
    return make_double2(c[x], -s[x]);
}

// Exchange inside a line (8 lanes, x = lane & 7, a wave-private tile): in v[p] = the value with first-stage frequency p of this
// lane's n1 = x; out v[k + 8 h] = the value of lane n1 = k at p = x + 8 h.
__device__ __forceinline__ void line_exchange(cplx (&v)[16], double* tile, int x) {
    double* wr = tile + x;
    const double* rd = tile + x * kLinePitch;
#pragma unroll
    for (int p = 0; p < 16; ++p) wr[p * kLinePitch] = v[p].x;
    __builtin_amdgcn_wave_barrier();
    double re[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) re[m] = rd[(m >> 3) * 8 * kLinePitch + (m & 7)];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 16; ++p) wr[p * kLinePitch] = v[p].y;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = make_double2(re[m], rd[(m >> 3) * 8 * kLinePitch + (m & 7)]);
    __builtin_amdgcn_wave_barrier();
}

// One 128-point line transform on layout A -> layout B (see the file comment)
template <bool INV>
__device__ __forceinline__ void line_fft(cplx (&v)[16], double* tile, int x) {
    if (R2F_FFT2D_EXP & 8) return;
    dft16<INV>(v);
    twiddle_powers<INV>(v, w128(x));
    line_exchange(v, tile, x);
    dft8<INV, 0>(v);
    dft8<INV, 8>(v);
}

// Layout B of a forward transform re-read as layout A of the inverse one: i2 = 2 f1 + h sits in register f1 + 8 h
__device__ __forceinline__ void b_to_a(cplx (&v)[16]) {
    cplx u[16];
#pragma unroll
    for (int i2 = 0; i2 < 16; ++i2) u[i2] = v[(i2 >> 1) + 8 * (i2 & 1)];
#pragma unroll
    for (int i2 = 0; i2 < 16; ++i2) v[i2] = u[i2];
}

__device__ __forceinline__ bool window_of2(const FftConvArgs& a, int t, int& wy, int& wx) {
    if (t >= a.ntiles) return false;
    wy = a.y0 + (t / a.gx) * a.vy - a.ay;
    wx = (t % a.gx) * a.vx - a.ax;
    return true;
}

// MODE 0: correlate the window pair `blockIdx.x` of the launch; MODE 1: spectrum of the zero-padded kernel image a.src (raw)
//
// 512 threads, each holding TWO 16-element slices of the window (virtual thread ids tid and tid + 512): a 1 024-thread block with one
// slice per thread has 128 VGPRs per thread for 64 of window data, and hipcc's scheduler -- which hoists the fifteen twiddle
// powers, the spectrum loads and the exchange reads of the next step above the arithmetic of the current one -- spilled 40 to 80
// of them whatever barriers it was given.  At 8 waves per CU the budget is 256: the two slices are independent work the scheduler
// can interleave, and nothing spills.
template <int MODE, bool EPI>
__global__ __launch_bounds__(kThreads) void fft2d_kernel(const FftConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    float* imgA = reinterpret_cast<float*>(sm);
    float* imgB = imgA + kW * kPitchF;
    const int tid = threadIdx.x, x = tid & 7;
    const int gp = a.pair0 + blockIdx.x, ci = gp / a.ppc, pc = gp - ci * a.ppc;
    int wyA = 0, wxA = 0, wyB = 0, wxB = 0;
    const bool hasA = MODE == 1 ? true : window_of2(a, 2 * pc, wyA, wxA);
    const bool hasB = MODE == 1 ? false : window_of2(a, 2 * pc + 1, wyB, wxB);
    // per slice s (virtual thread tid + 512 s, virtual wave w): the row held in the row layouts, (w / 2) * 16 + (w & 1) + 2 li;
    // the column (frequency) held in the column layouts = the line tile slot, virtual tid / 8; the bases of the 2-D image: element
    // (row, x + c) at by_row[c], element (x + r, col) at by_col[r * kPitch2D] (rows 64 .. 127 past the 16-bit immediate:
    // by_col + 64 rows).  Made from a fresh read of the thread id wherever they are used: carried through the whole kernel they
    // are what the allocator spills first.
    auto row_of = [&](int s) {
        const int vt = fresh_tid() + kThreads * s, w = vt >> 6;
        return (w >> 1) * 16 + (w & 1) + 2 * ((vt & 63) >> 3);
    };
    auto tile_of = [&](int s) { return sm + ((fresh_tid() + kThreads * s) >> 3) * kLineStride; };
    auto by_row_of = [&](int s) { return sm + row_of(s) * kPitch2D + (fresh_tid() & 7); };
    auto by_col_of = [&](int s) { return sm + (fresh_tid() & 7) * kPitch2D + ((fresh_tid() + kThreads * s) >> 3); };
    // ---- window floats -> fp32 staging images (coalesced: 128 consecutive lanes read one row)
    {
        const float* src = a.src.data + (long long)a.chan[ci] * a.src.plane_stride;
        const int c = tid & 127, r0 = tid >> 7;  // rows r0 + 4 it
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float f[32];
#pragma unroll
            for (int it = 0; it < 32; ++it) f[it] = 0.f;
            const int wy = half ? wyB : wyA, wx = half ? wxB : wxA;
            if (MODE == 1) {
                if (half == 0) {
#pragma unroll
                    for (int it = 0; it < 32; ++it) f[it] = src[(unsigned)((r0 + 4 * it) * kW + c)];
                }
            } else if (R2F_FFT2D_EXP & 1) {
#pragma unroll
                for (int it = 0; it < 32; ++it) f[it] = 0.25f * (float)(it + c);
            } else if (half ? hasB : hasA) {
                // 32-bit element offsets from the (wave-uniform) plane base: one address register per load, not a 64-bit pair
                const unsigned sx = (unsigned)reflect101(wx + c, a.W);
#pragma unroll
                for (int it = 0; it < 32; ++it) {
                    const int sy = clampi(reflect101(wy + r0 + 4 * it, a.H_global) - a.src.gy0, 0, a.src.rows - 1);
                    f[it] = src[(unsigned)sy * (unsigned)a.W + sx];
                }
            }
            float* img = (half ? imgB : imgA) + r0 * kPitchF + c;  // (one base + immediate offsets, like every LDS access below)
#pragma unroll
            for (int it = 0; it < 32; ++it) img[it * 4 * kPitchF] = f[it];
        }
    }
    __syncthreads();
    cplx v[kSlices][16];  // rows, layout A: element (row, x + 8 q) in v[s][q]
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        const float* pa = imgA + row_of(s) * kPitchF + x;
#pragma unroll
        for (int q = 0; q < 16; ++q) v[s][q] = make_double2((double)pa[8 * q], (double)pa[kW * kPitchF + 8 * q]);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        line_fft<false>(v[s], tile_of(s), x);
        __builtin_amdgcn_sched_barrier(0);  // one slice's transform at a time: interleaved, the two overflow the 256 VGPRs
    }  // v[f1 + 8 h]: row frequency 16 f1 + x + 8 h
    // ---- 2-D exchange: row threads -> column threads (layout A along the rows: thread (col, r1 = x) holds rows x + 8 q)
    double re[kSlices][16];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        double* const br = by_row_of(s);
#pragma unroll
        for (int m = 0; m < 16; ++m) br[16 * (m & 7) + 8 * (m >> 3)] = v[s][m].x;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        const double* const bc = by_col_of(s);
#pragma unroll
        for (int q = 0; q < 16; ++q) re[s][q] = (q < 8 ? bc : bc + 64 * kPitch2D)[8 * (q & 7) * kPitch2D];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        double* const br = by_row_of(s);
#pragma unroll
        for (int m = 0; m < 16; ++m) br[16 * (m & 7) + 8 * (m >> 3)] = v[s][m].y;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        const double* const bc = by_col_of(s);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[s][q] = make_double2(re[s][q], (q < 8 ? bc : bc + 64 * kPitch2D)[8 * (q & 7) * kPitch2D]);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        line_fft<false>(v[s], tile_of(s), x);
        __builtin_amdgcn_sched_barrier(0);  // one slice's transform at a time: interleaved, the two overflow the 256 VGPRs
    }  // v[f1 + 8 h]: 2-D frequency (16 f1 + x + 8 h, col)
    if (MODE == 1) {  // conj of the kernel's spectrum, in the order the correlation reads it
#pragma unroll
        for (int s = 0; s < kSlices; ++s)
#pragma unroll
            for (int m = 0; m < 16; ++m) a.kf_out[m * (kThreads * kSlices) + tid + kThreads * s] = make_double2(v[s][m].x, -v[s][m].y);
        return;
    }
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        const cplx* kf = a.kfs[ci] + tid + kThreads * s;
#pragma unroll
        for (int m = 0; m < 16; ++m) v[s][m] = cmul(v[s][m], (R2F_FFT2D_EXP & 2) ? make_double2(0.5, 0.25 * m) : kf[m * (kThreads * kSlices)]);
        b_to_a(v[s]);
    }
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        line_fft<true>(v[s], tile_of(s), x);
        __builtin_amdgcn_sched_barrier(0);
    }  // v[r1 + 8 h]: row 16 r1 + x + 8 h of column-frequency col
    // ---- 2-D exchange back: column threads -> row threads (layout A along the columns: thread (row, i1 = x) holds x + 8 i2)
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        double* const bc = by_col_of(s);
#pragma unroll
        for (int m = 0; m < 16; ++m) ((m & 7) < 4 ? bc : bc + 64 * kPitch2D)[(16 * (m & 3) + 8 * (m >> 3)) * kPitch2D] = v[s][m].x;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        const double* const br = by_row_of(s);
#pragma unroll
        for (int q = 0; q < 16; ++q) re[s][q] = br[8 * q];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        double* const bc = by_col_of(s);
#pragma unroll
        for (int m = 0; m < 16; ++m) ((m & 7) < 4 ? bc : bc + 64 * kPitch2D)[(16 * (m & 3) + 8 * (m >> 3)) * kPitch2D] = v[s][m].y;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        const double* const br = by_row_of(s);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[s][q] = make_double2(re[s][q], br[8 * q]);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kSlices; ++s) {
        line_fft<true>(v[s], tile_of(s), x);
        __builtin_amdgcn_sched_barrier(0);
    }  // v[k1 + 8 h]: output (row, 16 k1 + x + 8 h); .x window A, .y window B
    // ---- outputs -> fp32 staging images -> coalesced stores of the valid region
    __syncthreads();
    {
        const double scale = 1.0 / ((double)kW * kW);
        const int ch = a.chan[ci];
#pragma unroll
        for (int s = 0; s < kSlices; ++s) {
            float* po = imgA + row_of(s) * kPitchF + x;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                float oa = (float)(v[s][m].x * scale), ob = (float)(v[s][m].y * scale);
                if (EPI) {
                    float o2[2] = {log10_fast(oa, a.log_eps), log10_fast(ob, a.log_eps)};
                    curve_eval_batch<2, 1>(a.curve.cells, a.curve, ch, o2);
                    oa = o2[0], ob = o2[1];
                }
                po[16 * (m & 7) + 8 * (m >> 3)] = oa;
                po[kW * kPitchF + 16 * (m & 7) + 8 * (m >> 3)] = ob;
            }
        }
    }
    __syncthreads();
    float* dplane = a.dst.data + (long long)a.chan[ci] * a.dst.plane_stride;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (!(half ? hasB : hasA) || ((R2F_FFT2D_EXP & 4) && a.vy > 0)) continue;
        const int wy = half ? wyB : wyA, wx = half ? wxB : wxA;
        const float* img = half ? imgB : imgA;
        const int c_end = min(a.vx, a.W - (wx + a.ax));  // valid output columns inside the frame
        if (a.vec4) {  // window origins, valid widths and the frame width are multiples of 4: float4 rows
            const int c4 = (tid & 31) * 4, r0 = tid >> 5;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = r0 + 16 * it, gy = wy + a.ay + r;
                if (r < a.vy && gy < a.y1 && c4 < c_end)
                    *reinterpret_cast<float4*>(dplane + (long long)(gy - a.dst.gy0) * a.W + wx + a.ax + c4) =
                        *reinterpret_cast<const float4*>(img + r * kPitchF + c4);
            }
        } else {
            const int c = tid & 127, r0 = tid >> 7;
#pragma unroll
            for (int it = 0; it < 32; ++it) {
                const int r = r0 + 4 * it, gy = wy + a.ay + r;
                if (r < a.vy && gy < a.y1 && c < c_end) dplane[(long long)(gy - a.dst.gy0) * a.W + wx + a.ax + c] = img[r * kPitchF + c];
            }
        }
    }
}

}  // namespace

hipError_t fft2d_init_attributes() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft2d_kernel<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft2d_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(fft2d_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
}

// mode 0: a.npairs window pairs (pair0 ..) of the launch; mode 1: the kernel spectrum (a.src = the padded kernel image, raw)
hipError_t launch_fft2d(const FftConvArgs& a, int mode, hipStream_t s) {
    if (mode == 1)
        hipLaunchKernelGGL((fft2d_kernel<1, false>), dim3(1), dim3(kThreads), kLdsBytes, s, a);
    else if (a.epilogue == 1)
        hipLaunchKernelGGL((fft2d_kernel<0, true>), dim3(a.npairs), dim3(kThreads), kLdsBytes, s, a);
    else
        hipLaunchKernelGGL((fft2d_kernel<0, false>), dim3(a.npairs), dim3(kThreads), kLdsBytes, s, a);
    return hipGetLastError();
}

}  // namespace r2f
