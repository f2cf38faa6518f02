// Overlap-save FFT form of a large per-channel stencil (S2 halation), fp64.
//
// The reference's own CPU path runs cv.filter2D's DFT branch for kernels larger than 11 x 11; in fp32 that carries an error
// relative to the tile's energy (a shadow pixel next to a specular highlight misses the 1e-5 bar), in fp64 it is exact to
// ~1e-13 and costs ~400 flops per pixel instead of the 2 x 5 721 of the direct sum.  MI355X's fp64 vector rate (78 TFLOP/s)
// makes that the cheaper exact algorithm for the 87 x 87 disc.
//
// Unit of work: a PAIR of 256 x 256 input windows of one channel packed as real + imaginary part of one complex image (the
// kernel is real, so the correlation of the complex image is the pair of correlations -- no Hermitian bookkeeping at all).
// Window (ty, tx) produces the outputs [ty, ty + 256 - kh] x [tx, tx + 256 - kw] from the input rows ty - ay .. + 255.
//
//   pass 1  rows:     load (reflect-101 on the global frame), 256 forward FFTs along x, store TRANSPOSED   S1[k'][r]
//   pass 2  columns:  256 forward FFTs along r, multiply by conj(K^)[k'][r'], 256 inverse FFTs, store row-major S2[r][k']
//   pass 3  rows:     256 inverse FFTs along k', scale 2^-16, crop to the valid outputs, [log + density curve], store
//
// One wave transforms one 256-point line in LDS: radix-4 decimation in frequency forward (natural in, base-4
// digit-reversed out), radix-4 decimation in time backward (digit-reversed in, natural out), so no reordering pass exists;
// the kernel's spectrum is produced by the same two forward passes and therefore sits in the same digit-reversed layout.
#include "r2f_launch.h"

#include "../../include/r2f.h"

namespace r2f {

typedef double2 cplx;

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) {  // a * conj(b)
    return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}

constexpr int kN = kFftN;            // 256
constexpr int kLine = kN + 1;        // LDS line pitch in complex elements: +1 keeps the transposing accesses conflict-free
#ifndef R2F_FFT_LINES
#define R2F_FFT_LINES 8
#endif
#ifndef R2F_FFT_THREADS
#define R2F_FFT_THREADS 512
#endif
constexpr int kLines = R2F_FFT_LINES;  // lines per workgroup
constexpr int kFftThreads = R2F_FFT_THREADS;  // one wave per line at a time

// A lane's twiddles for the four radix-4 stages (stage s works on spans of L = 64 >> 2s): W^e, W^2e, W^3e with
// e = (lane mod L) * 64 / L.  They depend on the lane only, so they live in registers for every line a wave transforms.
struct LaneTwiddles {
    cplx w1[3], w2[3], w3[3];  // the last stage (L = 1) has e = 0: no twiddle at all
};

__device__ __forceinline__ LaneTwiddles lane_twiddles(const cplx* tw_global, int lane) {
    LaneTwiddles t;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int L = 64 >> (2 * s), e = (lane & (L - 1)) * (64 / L);
        t.w1[s] = tw_global[e];
        t.w2[s] = tw_global[2 * e];
        t.w3[s] = tw_global[3 * e];
    }
    return t;
}

// Forward: natural order in, digit-reversed out.  x: one LDS line.
__device__ __forceinline__ void fft256_forward(cplx* x, const LaneTwiddles& t, int lane) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int L = 64 >> (2 * s);
        const int j = lane & (L - 1), base = ((lane / L) * 4 * L) + j;
        const cplx x0 = x[base], x1 = x[base + L], x2 = x[base + 2 * L], x3 = x[base + 3 * L];
        const cplx a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d0 = csub(x1, x3);
        const cplx d = make_double2(d0.y, -d0.x);  // (x1 - x3) * (-i)
        x[base] = cadd(a, c);
        if (s < 3) {
            x[base + L] = cmul(cadd(b, d), t.w1[s]);
            x[base + 2 * L] = cmul(csub(a, c), t.w2[s]);
            x[base + 3 * L] = cmul(csub(b, d), t.w3[s]);
        } else {
            x[base + L] = cadd(b, d);
            x[base + 2 * L] = csub(a, c);
            x[base + 3 * L] = csub(b, d);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Backward (unnormalised: 256 x the inverse): digit-reversed in, natural out.
__device__ __forceinline__ void fft256_backward(cplx* x, const LaneTwiddles& t, int lane) {
#pragma unroll
    for (int s = 3; s >= 0; --s) {
        const int L = 64 >> (2 * s);
        const int j = lane & (L - 1), base = ((lane / L) * 4 * L) + j;
        cplx x0 = x[base], x1 = x[base + L], x2 = x[base + 2 * L], x3 = x[base + 3 * L];
        if (s < 3) x1 = cmulc(x1, t.w1[s]), x2 = cmulc(x2, t.w2[s]), x3 = cmulc(x3, t.w3[s]);
        const cplx a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d0 = csub(x1, x3);
        const cplx d = make_double2(-d0.y, d0.x);  // (x1 - x3) * (+i)
        x[base] = cadd(a, c);
        x[base + L] = cadd(b, d);
        x[base + 2 * L] = csub(a, c);
        x[base + 3 * L] = csub(b, d);
        __builtin_amdgcn_wave_barrier();
    }
}

// window origin (first input row / column) and validity of window `t` of the launch
__device__ __forceinline__ bool window_of(const FftConvArgs& a, int t, int& wy, int& wx) {
    if (t >= a.ntiles) return false;
    const int ty = a.y0 + (t / a.gx) * a.vy, tx = (t % a.gx) * a.vx;
    wy = ty - a.ay;
    wx = tx - a.ax;
    return true;
}

// ---------------------------------------------------------------------------------------------------- pass 1
__global__ __launch_bounds__(kFftThreads) void fft_rows_fwd_kernel(const FftConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) double2 fsm[];
    cplx* lines = fsm;
    const LaneTwiddles tw = lane_twiddles(a.tw, threadIdx.x & 63);
    const int pair = blockIdx.y, r0 = blockIdx.x * kLines;
    int wyA = 0, wxA = 0, wyB = 0, wxB = 0;
    const bool hasA = window_of(a, 2 * (a.pair0 + pair), wyA, wxA), hasB = window_of(a, 2 * (a.pair0 + pair) + 1, wyB, wxB);
    const float* src = a.src.data + (long long)a.ch * a.src.plane_stride;
    // load: one thread per column, the rows split over the thread groups
    const int c = threadIdx.x % kN;
    for (int i = threadIdx.x / kN; i < kLines; i += kFftThreads / kN) {
        const int r = r0 + i;
        double re = 0.0, im = 0.0;
        if (a.raw) {  // the zero-padded kernel image itself: a plain 256 x 256 plane, no reflection
            re = (double)src[(long long)r * kN + c];
        } else {
            if (hasA) {
                int sy = reflect101(wyA + r, a.H_global) - a.src.gy0;
                sy = clampi(sy, 0, a.src.rows - 1);  // rows outside the buffer only feed discarded outputs
                re = (double)src[(long long)sy * a.W + reflect101(wxA + c, a.W)];
            }
            if (hasB) {
                int sy = reflect101(wyB + r, a.H_global) - a.src.gy0;
                sy = clampi(sy, 0, a.src.rows - 1);
                im = (double)src[(long long)sy * a.W + reflect101(wxB + c, a.W)];
            }
        }
        lines[i * kLine + c] = make_double2(re, im);
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = wave; i < kLines; i += kFftThreads / 64) fft256_forward(lines + i * kLine, tw, lane);
    __syncthreads();
    // transposed store: S1[pair][k][r0 .. r0 + kLines): kLines lanes cover the rows of one k (contiguous bytes)
    cplx* s1 = a.s1 + (long long)pair * kN * kN;
    const int rr = threadIdx.x % kLines;
    for (int k = threadIdx.x / kLines; k < kN; k += kFftThreads / kLines) s1[(long long)k * kN + r0 + rr] = lines[rr * kLine + k];
}

// ---------------------------------------------------------------------------------------------------- pass 2
// mode 0: forward along r, multiply by the kernel spectrum, backward, store row-major S2[r][k]
// mode 1: forward along r only and store the conjugate: this IS the kernel spectrum (input = the padded kernel image)
__global__ __launch_bounds__(kFftThreads) void fft_cols_kernel(const FftConvArgs a, const int mode) {
    extern __shared__ __attribute__((aligned(16))) double2 fsm[];
    cplx* lines = fsm;
    const LaneTwiddles tw = lane_twiddles(a.tw, threadIdx.x & 63);
    const int pair = blockIdx.y, k0 = blockIdx.x * kLines;
    const cplx* s1 = a.s1 + (long long)pair * kN * kN;
    for (int i = threadIdx.x / kN; i < kLines; i += kFftThreads / kN)
        lines[i * kLine + threadIdx.x % kN] = s1[(long long)(k0 + i) * kN + threadIdx.x % kN];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = wave; i < kLines; i += kFftThreads / 64) {
        cplx* x = lines + i * kLine;
        fft256_forward(x, tw, lane);
        if (mode == 1) continue;
        const cplx* kf = a.kf + (long long)(k0 + i) * kN;
#pragma unroll
        for (int q = 0; q < 4; ++q) x[lane + 64 * q] = cmul(x[lane + 64 * q], kf[lane + 64 * q]);
        __builtin_amdgcn_wave_barrier();
        fft256_backward(x, tw, lane);
    }
    __syncthreads();
    if (mode == 1) {
        cplx* kf = a.kf_out;
        for (int i = threadIdx.x / kN; i < kLines; i += kFftThreads / kN) {
            const cplx v = lines[i * kLine + threadIdx.x % kN];
            kf[(long long)(k0 + i) * kN + threadIdx.x % kN] = make_double2(v.x, -v.y);
        }
        return;
    }
    cplx* s2 = a.s2 + (long long)pair * kN * kN;
    const int kk = threadIdx.x % kLines;
    for (int r = threadIdx.x / kLines; r < a.vy; r += kFftThreads / kLines)  // pass 3 never reads the rows past the valid outputs
        s2[(long long)r * kN + k0 + kk] = lines[kk * kLine + r];
}

// ---------------------------------------------------------------------------------------------------- pass 3
__global__ __launch_bounds__(kFftThreads) void fft_rows_inv_kernel(const FftConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) double2 fsm[];
    cplx* lines = fsm;
    const LaneTwiddles tw = lane_twiddles(a.tw, threadIdx.x & 63);
    const int pair = blockIdx.y, r0 = blockIdx.x * kLines;
    const cplx* s2 = a.s2 + (long long)pair * kN * kN;
    for (int i = threadIdx.x / kN; i < kLines; i += kFftThreads / kN)
        if (r0 + i < a.vy) lines[i * kLine + threadIdx.x % kN] = s2[(long long)(r0 + i) * kN + threadIdx.x % kN];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = wave; i < kLines; i += kFftThreads / 64) fft256_backward(lines + i * kLine, tw, lane);
    __syncthreads();
    // outputs: window row r0 + i, column c -> pixel (wy + ay + r, wx + ax + c) for r <= 256 - kh, c <= 256 - kw
    const int c = threadIdx.x % kN;
    if (c >= a.vx) return;
    float* dplane = a.dst.data + (long long)a.ch * a.dst.plane_stride;
    {
        const int half = threadIdx.x / kN;  // thread group 0 stores window A (real parts), group 1 window B (imaginary parts)
        int wy, wx;
        if (!window_of(a, 2 * (a.pair0 + pair) + half, wy, wx)) return;
        const int gx = wx + a.ax + c;
        if (gx >= a.W) return;
        for (int i = 0; i < kLines; ++i) {
            const int r = r0 + i;
            if (r >= a.vy) break;
            const int gy = wy + a.ay + r;
            if (gy >= a.y1) break;
            const cplx v = lines[i * kLine + c];
            float o = (float)((half ? v.y : v.x) * (1.0 / 65536.0));
            if (a.epilogue == 1) o = log_curve(a.curve, a.ch, o, a.log_eps);
            dplane[(long long)(gy - a.dst.gy0) * a.W + gx] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- launchers
static size_t fft_lds_bytes() { return (size_t)(kLines * kLine) * sizeof(cplx); }

hipError_t fft_init_attributes() {
    hipError_t e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft_rows_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fft_lds_bytes());
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft_cols_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fft_lds_bytes());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(fft_rows_inv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fft_lds_bytes());
}

hipError_t launch_fft_rows_fwd(const FftConvArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(fft_rows_fwd_kernel, dim3(kN / kLines, a.npairs), dim3(kFftThreads), fft_lds_bytes(), s, a);
    return hipGetLastError();
}

hipError_t launch_fft_cols(const FftConvArgs& a, int mode, hipStream_t s) {
    hipLaunchKernelGGL(fft_cols_kernel, dim3(kN / kLines, a.npairs), dim3(kFftThreads), fft_lds_bytes(), s, a, mode);
    return hipGetLastError();
}

hipError_t launch_fft_rows_inv(const FftConvArgs& a, hipStream_t s) {
    const int blocks = (a.vy + kLines - 1) / kLines;  // rows beyond the valid outputs are never stored
    hipLaunchKernelGGL(fft_rows_inv_kernel, dim3(blocks, a.npairs), dim3(kFftThreads), fft_lds_bytes(), s, a);
    return hipGetLastError();
}

}  // namespace r2f
