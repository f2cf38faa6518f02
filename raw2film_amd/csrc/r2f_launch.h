// r2f_launch.h -- host-callable launchers implemented in r2f_kernels.hip.
#pragma once

#include <cstdint>
#include <tuple>
#include <type_traits>
#include "r2f_device.h"

struct r2f_blit;

namespace r2f {

// Kernel launches report THEIR OWN status.  hipLaunchKernelGGL + hipGetLastError() would hand back (and consume) whatever error was
// pending on the calling thread -- one of PyTorch's or RCCL's, left there by an abandoned capture, say -- as if this library had
// produced it.  hipLaunchKernel returns the status of the launch itself and leaves the thread's last-error slot alone
// unless this launch fails; the first failure of a launcher is parked in a thread-local until the launcher returns it.
inline thread_local hipError_t tl_launch_error = hipSuccess;
template <typename... KArgs, typename... Args>
inline void launch_k(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t s, const Args&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "one value per kernel parameter");
    std::tuple<std::remove_cv_t<KArgs>...> vals{static_cast<std::remove_cv_t<KArgs>>(args)...};
    void* ptrs[sizeof...(KArgs)];
    int i = 0;
    std::apply([&](auto&... v) { ((ptrs[i++] = static_cast<void*>(&v)), ...); }, vals);
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, s);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // the error this launch produced is reported through the return value, not left pending
        if (tl_launch_error == hipSuccess) tl_launch_error = e;
    }
}
inline hipError_t take_launch_status() {
    const hipError_t e = tl_launch_error;
    tl_launch_error = hipSuccess;
    return e;
}

struct FrontArgs {
    const void* in;
    int in_layout;  // R2F_LAYOUT_*
    int in_gy0, in_rows;
    int upto;  // R2F_UPTO_*
    DevPlanes dst;
    float* out_f32;
    uint8_t* out_u8;
    int out_gy0;
    int y0, y1, W, H_global;
    int use_matrix;
    Mat3 mat;
    DevLut2D lut2d;
    DevCurve curve;
    DevLut3D lut3d;
    float log_eps;
    float lut3d_scale;
    int lut3d_mode;
    int vec;  // 1: W % 4 == 0 and all bases 16-byte aligned -> float4 paths
    int blocks_per_cu;  // curve-in-LDS variant: workgroups in the grid per CU (each copies the curve cells once, then walks rows)
    int fast;           // 1: the fused S0..S8 pass may take the specialised kernel of r2f_front.hip when it is eligible
    // upto = EXPOSURE, fast kernel only: channels in finish_mask get finish_w[c] * exposure -> log -> curve and are written to
    // `finish_dst` (density planes) instead of `dst`
    int finish_mask;
    float finish_w[3];
    DevPlanes finish_dst;
    // fast kernel, upto = EXPOSURE: min / max |.| of the exposure samples written to `dst` for the channels in track_mask (the ones
    // the halation's FFT passes read) are accumulated into this record's tile grid (blk = nullptr: not tracked)
    RangeRecord track;
    int track_mask;
    int gx;  // fast kernel: tile columns of the frame (ceil(W / 256)), set by its launcher
};
bool front_fast_eligible(const FrontArgs& a);
hipError_t launch_front_fast(const FrontArgs& a, hipStream_t s);
hipError_t front_fast_init_attributes();

struct StencilArgs {
    DevStencil st[3];
    DevPlanes src, dst;
    int y0, y1, W, H_global;
    int epilogue;  // 0 = none, 1 = log + density curve (halation)
    DevCurve curve;
    float log_eps;
    int nchan;
    int vec;
    int chan[3];       // blockIdx.z -> channel (a launch may cover a subset: the others take the FFT form)
    int xcd_remap;     // 0 none, 1 arithmetic (contiguous run per XCD), 2 table `order`
    const int* order;  // xcd_remap == 2: tile index (bx + gx * by) of every linear workgroup id of one channel
    int ablate;  // profiling aid: 1 = skip the tile fill, 2 = skip the accumulation (results invalid)
    // fixed_r = R in 1..11: every channel of the launch is a square mirror-symmetric (2 R + 1)^2 stencil laid out as stencil_fixed
    // expects (variant 0 only); fixed_w: 3 channels x (2 R + 4) x (R + 1) x 2 weight pairs
    int fixed_r;
    const float* fixed_w;
};

// S7: the blurred low-res highlight map and how to up-sample it (ndimage.zoom(order=1) + edge pad, effects.py:381-388)
struct BurnUp {
    const float* map;  // h_lo x w_lo, nullptr = burn off
    int h_lo, w_lo;
    int h_up, w_up;    // h_lo * cell, w_lo * cell: extent of the zoomed map; beyond it the edge is repeated
    double zy, zx;     // (h_lo - 1) / (h_up - 1), (w_lo - 1) / (w_up - 1) in double, like scipy's zoom
    float strength;
};

struct BurnSumsArgs {
    DevPlanes src;
    float* cell_sums;
    int y0, y1, W, H_global;
    int h_lo, w_lo;
};

struct BurnMapArgs {
    const float* cell_sums;
    float* map;
    float* scratch;
    int h_lo, w_lo;
    float d_ref;
    double w[13];  // scipy's gaussian kernel for sigma = 3, truncate = 2
};

struct TailArgs {
    DevPlanes src;
    float* out_f32;
    uint8_t* out_u8;
    int out_gy0;
    int y0, y1, W, H_global;
    int grain;  // 0/1
    int to_planes;  // 1: stop after grain + clip and write density planes `dst` (S7 needs the whole grained frame);
                    // 2: write the grain field itself (K_g * noise, no image involved) to `dst`
    DevPlanes dst;
    BurnUp burn;
    int mono;
    const FrameParams* frame;  // device-resident: the grain seed of this render (r2f_write_frame_params)
    DevStencil gk[3];  // grain stencil, common geometry for the 3 channels
    DevCurve grain_lut;
    DevLut3D lut3d;
    float lut3d_scale;
    int lut3d_mode;
    int vec;
    int cells_in_lds, cells_off;  // set by the launcher: grain-LUT cells copied to LDS at float offset cells_off
    DevPlanes gfield;             // grain == 0 && has_gfield: a precomputed grain field is applied pointwise (lut3d_kernel)
    int has_gfield;
    // fixed_r = R in 1..9: the grain stencil is a square mirror-symmetric (2 R + 1)^2 box laid out as stencil_fixed expects;
    // fixed_w: 3 channels x (2 R + 2) x (R + 1) weight pairs; fixed_same: the three channels share their taps
    int fixed_r, fixed_same;
    const float* fixed_w;
    // sep = 1 (with fixed_r = R): every channel's stencil is u v^T to fp32 rounding, v mirror symmetric: two 1-D passes.
    // sep_u[c][i], i <= 2 R: taps along y; sep_v[c][j], j <= R: the left half of the taps along x
    int sep;
    float sep_u[3][19], sep_v[3][10];
};

constexpr int kChromaMaxTaps = 63;
struct ChromaArgs {
    const void* in;  // pass 1 input image
    int in_layout, in_gy0, in_rows;
    DevPlanes src;  // pass 2 input planes
    DevPlanes dst;
    int y0, y1, W, H_global;
    int radius;  // taps = 2*radius + 1
    int vec;
    float w[kChromaMaxTaps];
};

struct ResizeArgs {
    const void* in;
    int in_layout, H, W;
    DevPlanes dst;
    int out_h, out_w;
};

// Pre-path free rotation: dst(y, x) = bilinear sample of `in` at M * (x + ox, y + oy, 1), zero outside (cv.warpAffine,
// INTER_LINEAR, BORDER_CONSTANT); only the window the rotate() crop keeps is produced.
struct WarpArgs {
    const void* in;
    int in_layout, H, W;
    DevPlanes dst;
    int out_h, out_w, oy, ox;
    float m[6];  // dst -> src, row-major 2 x 3, rounded from double like OpenCV's float kernels
};

// Post-path up-scale of the uint8 result: cv.resize(..., INTER_LANCZOS4) with host-built fixed-point tables.
struct LanczosArgs {
    const uint8_t* src;  // (H, W, 3)
    uint8_t* dst;        // (out_h, out_w, 3)
    int H, W, out_h, out_w;
    const int* xofs;     // out_w: source column of tap 3
    const short* xcoef;  // out_w x 8
    const int* yofs;     // out_h
    const short* ycoef;  // out_h x 8
};

// A channel whose stencil has a single tap (the halation's identity plane): dst = epilogue(w * src) over rows [y0, y1).
struct TapArgs {
    DevPlanes src, dst;
    int ch, y0, y1, W;
    float w;
    int epilogue;
    DevCurve curve;
    float log_eps;
    int vec;
};
hipError_t launch_single_tap(const TapArgs& a, hipStream_t s);

// Overlap-save FFT form of a large stencil (r2f_fft.hip): one channel, windows [pair0*2, (pair0+npairs)*2) of the launch.
constexpr int kFftN = 256;
struct FftConvArgs {
    DevPlanes src, dst;
    int nch, chan[3];         // channels of this launch (planes of src / dst, curve channels); pairs are numbered channel-major
    int ppc;                  // window pairs per channel
    int y0, y1, W, H_global;  // output rows [y0, y1) of the global frame
    int ay, ax;               // anchor inside the cropped kernel box
    int ny, nx;               // window rows: 256 or 512; columns: 256, 512 or 1024
    int vy, vx;               // valid outputs per window: ny - kh + 1, nx - kw + 1 (vx rounded down to a multiple of 4)
    int gx, ntiles;           // windows per row of windows, windows in total
    int pair0, npairs;
    int raw;                  // 1: src is the zero-padded ny x nx kernel image itself (kernel-spectrum build)
    const double2* tw;        // exp(-2 pi i k / 256), k < 256
    const double2* tw512;     // exp(-2 pi i k / 512), k < 256 (the 512-point lines use k < 32)
    const double2* tw1024;    // exp(-2 pi i k / 1024), k < 64
    const double2* kfs[3];    // per launch channel: conj of the kernel's 2-D spectrum (scratch layout); with `kreal` the same
                              // layout holds one double per element (read through a const double*)
    double2* kf_out;          // pass 2, mode 1
    int kreal;                // 1: the kernel is centrally symmetric and was laid out with its anchor on the window origin, so its
                              // spectrum is REAL: one multiply per component, half the spectrum bytes (pass 2), and the valid
                              // outputs of a window start at row / column (oy, ox) = the anchor instead of (0, 0)
    int oy, ox;               // first scratch row / column that holds a valid output (0, 0 unless kreal)
    int cols_walk;            // 1: pass 2 of 256-row windows with a real spectrum walks the launch's pairs per column block
    int cols_slots;           // ... on a grid of (up to) this many resident workgroups (2 per CU)
    double2* s1;              // npairs x ny x nx scratch images, transformed in place (layout: sidx in r2f_fft.hip)
    int s32;                  // scratch element: 0 complex128, 1 complex64 (half the bytes; the arithmetic stays fp64), 2 the 12-byte form,
                              // 3 chosen ON THE DEVICE per window pair between 0 and 2 from the range of the pair's samples (dyn_flags)
    // s32 == 3: one flag per pair-in-channel (index gp % ppc), written by fft_decide_kernel ahead of the call's launches from the
    // exposure-range tiles: != 0 -> the pair's scratch image holds 12-byte elements (max |x| <= bound * max(min x, floor) over the
    // pair's two windows), 0 -> complex128.  The pairs' images are 16 bytes per element apart either way.
    const int* dyn_flags;
    int epilogue;
    DevCurve curve;
    float log_eps;
    int vec4;                 // 1: vx % 4 == 0, W % 4 == 0 and dst planes 16-byte aligned -> float4 stores in pass 3
    int epi_lds;              // pass 3 with the epilogue: 1 = the channel's curve cells may be copied to LDS (A/B knob)
    int epi_lds_off;          // set by the launcher: offset (in doubles) of those cells in the dynamic LDS, 0 = gather from global memory
};
hipError_t fft_init_attributes();
hipError_t launch_fft_rows_fwd(const FftConvArgs& a, hipStream_t s);
hipError_t launch_fft_cols(const FftConvArgs& a, int mode, hipStream_t s);
hipError_t launch_fft_rows_inv(const FftConvArgs& a, hipStream_t s);
hipError_t launch_fft_decide(const FftConvArgs& a, const RangeRecord& rec, float bound, float floor_, int* flags, hipStream_t s);

struct NoiseArgs {
    uint32_t* hash;
    float* noise;
    int y0, y1, W;
    const FrameParams* frame;  // device-resident seed, like TailArgs
    int mono;
};

// tile geometry of the stencil variants (threads = BX*BY, tile = 4*BX x Q*BY)
struct StencilVariant {
    int id;
    int BX, BY, Q;
    int TW() const { return 4 * BX; }
    int TH() const { return Q * BY; }
};
constexpr int kNumStencilVariants = 3;
extern const StencilVariant kStencilVariants[kNumStencilVariants];
#ifndef R2F_TAIL_BX
#define R2F_TAIL_BX 16
#endif
#ifndef R2F_TAIL_BY
#define R2F_TAIL_BY 32
#endif
#ifndef R2F_TAIL_Q
#define R2F_TAIL_Q 2
#endif
constexpr int kTailBX = R2F_TAIL_BX, kTailBY = R2F_TAIL_BY, kTailQ = R2F_TAIL_Q;  // grain/tail tile 64 x 64, 512 threads
constexpr size_t kMaxLds = 160 * 1024;

size_t stencil_lds_bytes(const StencilVariant& v, const DevStencil* st, int nchan);
size_t tail_lds_bytes(const DevStencil* gk, int mono, int cells_in_lds = 0, int grain_m = 2);

hipError_t init_kernel_attributes();
hipError_t launch_front(const FrontArgs& a, hipStream_t s);
hipError_t launch_stencil(const StencilArgs& a, int variant, hipStream_t s);
hipError_t launch_tail(const TailArgs& a, hipStream_t s);
hipError_t launch_warp_affine(const WarpArgs& a, hipStream_t s);
hipError_t launch_lanczos4_u8(const LanczosArgs& a, hipStream_t s);
hipError_t launch_noise(const NoiseArgs& a, hipStream_t s);
// the per-render write of the context's FrameParams block (and the reset of the record's tiles) ahead of a frame's launches
// mode: 0 seed only, 1 seed + range reset, 2 range reset only, 3 range made unusable (frame_params_kernel)
hipError_t launch_frame_params(const RangeRecord& rec, const FrameParams& v, int mode, hipStream_t s);
// the range of rows [y0, y1) and [y2, y3) of the planes in `mask` merged into the record's tiles (r2f_stage_exposure_range)
hipError_t launch_exposure_range(const DevPlanes& src, int y0, int y1, int y2, int y3, int W, int mask, const RangeRecord& rec, hipStream_t s);

// Caller-side histogram (utils.py:145-165): per-channel counts of an interleaved uint8 image; counts[3][256] is zeroed first.
hipError_t launch_histogram_u8(const uint8_t* image, long long n_bytes, uint32_t* counts, hipStream_t s);
hipError_t launch_burn_sums(const BurnSumsArgs& a, hipStream_t s);
// r2f_post.hip
hipError_t launch_resize_area_u8(const uint8_t* src, int H, int W, uint8_t* dst, int out_h, int out_w, hipStream_t s);
hipError_t launch_stream_copy(const void* src, void* dst, long long bytes, hipStream_t s);
hipError_t launch_decode_u16(const uint16_t* src, long long n, int ch, float divisor, float factor, float* dst, hipStream_t s);
hipError_t launch_lanczos4_f32(const void* in, int in_layout, int H, int W, const DevPlanes& dst, int out_h, int out_w, const int* xofs,
                               const float* xcoef, const int* yofs, const float* ycoef, hipStream_t s);
hipError_t launch_blit_rgba8(const float* src, int H, int W, uint8_t* dst, int dst_h, int dst_w, const ::r2f_blit& t, hipStream_t s);
hipError_t launch_histogram_render(const uint32_t* counts, const uint8_t* mix_rgba, int height, uint8_t* image, uint8_t* target, int th,
                                   int tw, hipStream_t s);
hipError_t launch_burn_map(const BurnMapArgs& a, hipStream_t s);
hipError_t launch_chroma_h(const ChromaArgs& a, hipStream_t s);
hipError_t launch_chroma_v(const ChromaArgs& a, hipStream_t s);
hipError_t launch_resize_area(const ResizeArgs& a, hipStream_t s);

}  // namespace r2f
