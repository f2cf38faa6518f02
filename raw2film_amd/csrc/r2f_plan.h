// r2f_plan.h -- the host-side planners of libr2f_hip.so, free of HIP: everything r2f_api.hip decides on the CPU before it uploads
// a table or launches a kernel (tap boxes and their device entry lists, FFT window shapes and batch sizes, tile orders,
// LANCZOS4 / Gaussian tables, curve cells, workspace sizes).  Plain C++ so that the same translation unit also builds with
// `g++ -fsanitize=address,undefined` into the fuzz harness of tests/test_plan_sanitizers.py (GPU-side sanitizers are not available on
// this pool; this is the part of the library that can run under one).  Nothing here touches the device.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace r2f {
namespace plan {

// ------------------------------------------------------------------------------------------------ stencil taps
// A stencil as handed to r2f_set_kernel: (kh, kw, kc) row-major floats, kc in {1, 3}; channel c reads plane kc == 1 ? 0 : c.
struct Taps {
    const float* k;
    int kh, kw, kc;
    float at(int i, int j, int c) const { return k[((size_t)i * kw + j) * kc + (kc == 1 ? 0 : c)]; }
};

// Bounding box of channel c's non-zero taps {i_lo, i_hi, j_lo, j_hi}; an all-zero plane keeps its centre tap.
void tap_box(const Taps& t, int c, int box[4]);
// Is channel c a single tap at the anchor (kh / 2, kw / 2)?  (*w = its weight)
bool single_tap_channel(const Taps& t, int c, float* w);
// Channel c's box {i_lo, i_hi, j_lo, j_hi}: odd width >= 9, centred on the anchor column, left-right mirror symmetric bit for bit?
bool mirror_symmetric(const Taps& t, int c, const int box[4]);

// Device form of one channel (mirrors DevStencil's geometry fields; the pointers are filled in by the uploader).
struct StencilGeom {
    int kh = 0, kw = 0, kw_pad = 0, RS = 0, ay = 0, ax = 0, sym = 0;
    int n_phases = 0, n_rowsteps = 0, n_entries = 0, mask_first_or = 0, mask_last_or = 0, max_lds_rows = 0;
};
// The entry list stencil_accumulate<Q> consumes (layout: r2f_device.h, DevStencil).
struct StreamHost {
    std::vector<float> w;
    std::vector<int> rowinfo, phases;  // rowinfo: 4 ints per non-empty row step (+ 2 dummy records); phases: 4 ints per phase + terminator
    int n_phases = 0, n_rowsteps = 0, n_entries = 0, max_lds_rows = 0, mask_first_or = 0, mask_last_or = 0;
};
// Channel c of `t` cropped to `box` (the caller may have widened it to a box common to all channels) for a tile of TW x TH
// outputs, Q rows per lane and an LDS budget in bytes (0 = the whole stencil height in one phase).  allow_sym: pair mirrored taps
// when the channel allows it (force_sym_off overrides a shared-geometry decision).  Returns 0, or -3 (R2F_ETOOLARGE) when a
// single row step does not fit the budget.
int plan_stencil_channel(const Taps& t, int c, const int box[4], bool sym, int Q, int TW, int TH, size_t lds_budget, StencilGeom* g,
                         StreamHost* sh);

constexpr int fixed_ax(int R) { return R < 4 ? R : (R <= 6 ? 6 : (R <= 10 ? 10 : 14)); }  // = fixed_stencil_ax of r2f_device.h
// R if the channels `chans` all fill the same square (2 R + 1)^2 box around the anchor, 1 <= R <= max_r, mirror symmetric, and
// their device geometry `geom[c]` is what stencil_fixed expects; else 0.
int fixed_stencil_radius(const Taps& t, const StencilGeom* geom, const int* chans, int nch, int max_r);
// stencil_fixed<R, Q>'s weight table for the three channels ([(2 R + Q)][R + 1][Q / 2] pairs each); *same: identical tables.
std::vector<float> fixed_stencil_weights(const Taps& t, int R, int Q, bool* same);
// Is every channel's (2 R + 1)^2 box K = u v^T to 6e-7 of its largest tap?  Fills u[c][0 .. 2 R], v[c][0 .. R] (left half).
bool separable_taps(const Taps& t, int R, float u[3][19], float v[3][10]);

// Source rows [lo, hi) a stencil reaching `above` / `below` rows needs for outputs [y0, y1) of an H-row frame (reflect-101).
void stencil_source_rows(int y0, int y1, int above, int below, int H, int* lo, int* hi);

// ------------------------------------------------------------------------------------------------ FFT form
struct FftOptions {
    int window = 0;        // columns forced (0 = choose)
    int window_max = 512;  // widest the automatic choice may take
    int window_rows = 0;   // rows forced (0 = choose)
    int batch_mib = 192;   // scratch in flight, MiB (a 256 x 256 complex128 pair = 1 MiB)
    int streams = 2;
    int even = 1;          // equal numbers of launch triples per stream
};
constexpr int kFftMaxTaps = 400;
// Window shape {ny, nx} for a bh x bw tap box on a W x H frame (the one whose three passes move the fewest scratch bytes).
// rows_hint > 0: the rows of THIS call (a row shard) -- the window rows are then chosen for the shard, not for the whole frame.
bool fft_window(const FftOptions& o, int bh, int bw, int W, int H, bool s32, int* ny, int* nx);
struct FftBatches {
    int ny = 0, nx = 0, vy = 0, vx = 0, gx = 0, ntiles = 0, ppc = 0, pairs = 0;
    int nstreams = 1, batch = 0, launches = 0;
    size_t img_bytes = 0, scratch_bytes = 0;
};
// Tiling of output rows [y0, y1) x W columns into window pairs and their batches; elem_bytes: 16 (complex128), 8 or 12.
FftBatches fft_batches(const FftOptions& o, int ny, int nx, int bh, int bw, int W, int y0, int y1, int nch, int elem_bytes);

// ------------------------------------------------------------------------------------------------ tables
// Tile order of a gx x gy grid of stencil workgroups: entry i = tile index of linear workgroup id i (XCD-contiguous runs walked
// in bands of tile columns; band = 0: automatic).
std::vector<int> tile_order(int gx, int gy, int band);
// cv::interpolateLanczos4 weights for fractional position x.
void lanczos4_coeffs(float x, float* coeffs8);
int lanczos4_table_u8(int ssize, int dsize, int* ofs, short* coef);
int lanczos4_table_f32(int ssize, int dsize, int* ofs, float* coef);
// gaussian_kernel_1d(2 size + 1, 0.3 ((taps - 1) / 2 - 1) + 0.8) normalised by its float32 pairwise sum (effects.py:421-435).
constexpr int kChromaMaxTaps = 63;
bool chroma_weights(int size, float* w);
// scipy's gaussian kernel for sigma = 3, truncate = 2 (13 taps, doubles).
void burn_weights(double* w13);

// One cell {xp[i], xp[i + 1], fp[i], slope[i]} per interval and channel of a (4, m) curve table; slopes in double like np.interp.
struct CurveCells {
    std::vector<float> cells;  // 3 x (m - 1) x 4
    float x0 = 0, x1 = 0, inv_step = 0, f_first[3] = {0, 0, 0}, f_last[3] = {0, 0, 0};
    int m = 0, near = 0;
};
// 0, or -1 when the table is unusable (m < 2, xp decreasing).
int curve_cells(const float* lut4xm, int m, CurveCells* out);

// Plane sets + burn scratch of r2f_render, in floats.
size_t plane_set_floats(int H, int W);
bool burn_geometry(int burn_cell, int H, int W, int* h_lo, int* w_lo);
size_t workspace_floats(unsigned flags, int burn_cell, int H, int W);

}  // namespace plan
}  // namespace r2f
