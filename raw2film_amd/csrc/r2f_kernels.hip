// r2f_kernels.hip -- gfx950 kernels of the film-emulation render path and their launchers.
//
// Kernel inventory (pass graph mirrors gpu_processor.py:1763-1862, fused where it is free):
//   front_kernel      S0 3x3 + S1 2-D LUT [+ S3 log + S4 curve [+ S8 3-D LUT + S9 u8]]        HBM-bound
//   stencil_kernel    S2 halation (+S3+S4 epilogue) / S5 MTF: LDS-tiled direct stencil         fp32-VALU-bound
//   tail_kernel       S6 hash noise -> LDS, grain stencil, grain LUT, clip [, S8, S9]          VALU/LDS
//   lut3d_kernel      [S7 burn subtract +] S8 + S9 from density planes                         HBM-bound
//   burn_sums/_map    S7: INTER_AREA cell sums of the green density; clip + Gaussian on the map
//   chroma_h/_v       pre-path chroma NR: xyY + separable Gaussian on the chromaticity planes  HBM-bound
//   resize_area       pre-path INTER_AREA down-scale to the preview resolution
//   warp_affine       pre-path free rotation (cv.warpAffine, INTER_LINEAR, zero border)
//   lanczos4_u8       post-path up-scale of the uint8 result (cv.resize INTER_LANCZOS4): the way back from max_scale
//   noise_kernel      S6a test entry (hash + Gaussian field)
//   histogram_u8      caller-side RGB histogram counts of the uint8 output (utils.generate_histogram, histogram.wgsl pass 1)
#include <algorithm>

#include "r2f_launch.h"

#include "../../include/r2f.h"

#ifndef R2F_CURVE_BATCH3
#define R2F_CURVE_BATCH3 1  // A/B switch: a pixel's three curve evaluations as one batch (see front_kernel)
#endif

namespace r2f {

const StencilVariant kStencilVariants[kNumStencilVariants] = {
    {0, 32, 16, 4},  // 512 threads, tile 128 x 64, 4x4 outputs per lane
    {1, 16, 8, 4},   // 128 threads, tile 64 x 32: fallback for very wide stencils
    {2, 32, 8, 4},   // 256 threads, tile 128 x 32: never auto-selected first; occupancy experiments (2 waves per SIMD)
};

// ------------------------------------------------------------------------------ output
__device__ __forceinline__ uint32_t to_u8(float v) {
    // cpu_processor.py:407: (image * 255).astype(uint8) -- truncation
    float s = fminf(fmaxf(v * 255.0f, 0.0f), 255.0f);
    return (uint32_t)s;
}

// Write `nv` (1..4) consecutive pixels of row `orow` starting at column x, interleaved HWC.
__device__ __forceinline__ void emit_hwc(float* out_f32, uint8_t* out_u8, long long orow, int x, int W, int nv, bool vec,
                                         const float (&r)[4], const float (&g)[4], const float (&b)[4]) {
    const long long base = (orow * W + x) * 3;
    if (out_f32) {
        float* o = out_f32 + base;
        if (vec && nv == 4) {
            float4* o4 = reinterpret_cast<float4*>(o);
            o4[0] = make_float4(r[0], g[0], b[0], r[1]);
            o4[1] = make_float4(g[1], b[1], r[2], g[2]);
            o4[2] = make_float4(b[2], r[3], g[3], b[3]);
        } else {
            for (int p = 0; p < nv; ++p) {
                o[3 * p + 0] = r[p];
                o[3 * p + 1] = g[p];
                o[3 * p + 2] = b[p];
            }
        }
    }
    if (out_u8) {
        uint8_t* o = out_u8 + base;
        if (vec && nv == 4) {
            uint32_t q[12];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                q[3 * p + 0] = to_u8(r[p]);
                q[3 * p + 1] = to_u8(g[p]);
                q[3 * p + 2] = to_u8(b[p]);
            }
            uint32_t* o32 = reinterpret_cast<uint32_t*>(o);
            o32[0] = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
            o32[1] = q[4] | (q[5] << 8) | (q[6] << 16) | (q[7] << 24);
            o32[2] = q[8] | (q[9] << 8) | (q[10] << 16) | (q[11] << 24);
        } else {
            for (int p = 0; p < nv; ++p) {
                o[3 * p + 0] = (uint8_t)to_u8(r[p]);
                o[3 * p + 1] = (uint8_t)to_u8(g[p]);
                o[3 * p + 2] = (uint8_t)to_u8(b[p]);
            }
        }
    }
}

// NT: non-temporal loads (ld4_stream) -- for planes that are read exactly once and written by another kernel (the tail's densities:
// 0.95 against 0.99 ms at 100 MP, profiles/r04_tail_overlap_probe.txt; the pointwise kernels' own plane reads lose with them)
template <bool NT = false>
__device__ __forceinline__ void load_planes4(const DevPlanes& pl, int gy, int x, int W, int nv, bool vec, float (&r)[4],
                                             float (&g)[4], float (&b)[4]) {
    const float* p0 = pl.data + (long long)(gy - pl.gy0) * W + x;
    const float* p1 = p0 + pl.plane_stride;
    const float* p2 = p1 + pl.plane_stride;
    if (vec && nv == 4) {
        const float4 a = NT ? ld4_stream(p0) : *reinterpret_cast<const float4*>(p0);
        const float4 c = NT ? ld4_stream(p1) : *reinterpret_cast<const float4*>(p1);
        const float4 d = NT ? ld4_stream(p2) : *reinterpret_cast<const float4*>(p2);
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
        g[0] = c.x; g[1] = c.y; g[2] = c.z; g[3] = c.w;
        b[0] = d.x; b[1] = d.y; b[2] = d.z; b[3] = d.w;
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool ok = p < nv;
            r[p] = ok ? p0[p] : 0.f;
            g[p] = ok ? p1[p] : 0.f;
            b[p] = ok ? p2[p] : 0.f;
        }
    }
}

__device__ __forceinline__ void store_planes4(const DevPlanes& pl, int gy, int x, int W, int nv, bool vec,
                                              const float (&r)[4], const float (&g)[4], const float (&b)[4]) {
    float* p0 = pl.data + (long long)(gy - pl.gy0) * W + x;
    float* p1 = p0 + pl.plane_stride;
    float* p2 = p1 + pl.plane_stride;
    if (vec && nv == 4) {
        *reinterpret_cast<float4*>(p0) = make_float4(r[0], r[1], r[2], r[3]);
        *reinterpret_cast<float4*>(p1) = make_float4(g[0], g[1], g[2], g[3]);
        *reinterpret_cast<float4*>(p2) = make_float4(b[0], b[1], b[2], b[3]);
    } else {
        for (int p = 0; p < nv; ++p) {
            p0[p] = r[p];
            p1[p] = g[p];
            p2[p] = b[p];
        }
    }
}

__device__ __forceinline__ void apply_lut3d(const DevLut3D& L, float scale, int mode, float& r, float& g, float& b) {
    if (mode == 0)
        apply_lut3d_tetra(L, scale * (float)(L.n - 1), r, g, b);
    else
        apply_lut3d_trilinear(L, scale, r, g, b);
}

// ------------------------------------------------------------------------------ front
// One lane = 4 consecutive pixels of one row.  Block (64, BYF); each block walks down the frame in
// steps of gridDim.y * BYF rows.  LDSC: the density curve (3 x (m-1) float4 cells, 48 KB at m = 1024)
// is copied to LDS once per block, so the three data-dependent curve look-ups per pixel are LDS reads
// instead of scattered 16-byte global gathers (which made the fused LUT-only pass gather-bound).
constexpr int kFrontBY = 8;

template <bool LDSC>
__global__ __launch_bounds__(64 * kFrontBY) void front_kernel(const FrontArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 cells_lds[];
    if (LDSC) {
        const int n = 3 * (a.curve.m - 1);
        for (int i = threadIdx.y * 64 + threadIdx.x; i < n; i += 64 * kFrontBY) cells_lds[i] = a.curve.cells[i];
        __syncthreads();
    }
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (x >= a.W) return;
    const int W = a.W;
    const int nv = min(4, W - x);
    const bool vec = a.vec != 0;
    const float* in = static_cast<const float*>(a.in);
    for (int gy = a.y0 + blockIdx.y * kFrontBY + threadIdx.y; gy < a.y1; gy += gridDim.y * kFrontBY) {
        const long long irow = gy - a.in_gy0;
        float r[4], g[4], b[4];
        if (a.in_layout == R2F_LAYOUT_CHW) {
            DevPlanes pl;
            pl.data = const_cast<float*>(in);
            pl.plane_stride = (long long)a.in_rows * W;
            pl.gy0 = a.in_gy0;
            pl.rows = a.in_rows;
            load_planes4(pl, gy, x, W, nv, vec, r, g, b);
        } else if (a.in_layout == R2F_LAYOUT_HWC3) {
            const float* p = in + (irow * W + x) * 3;
            if (vec && nv == 4) {
                const float4* p4 = reinterpret_cast<const float4*>(p);
                const float4 v0 = p4[0], v1 = p4[1], v2 = p4[2];
                r[0] = v0.x; g[0] = v0.y; b[0] = v0.z;
                r[1] = v0.w; g[1] = v1.x; b[1] = v1.y;
                r[2] = v1.z; g[2] = v1.w; b[2] = v2.x;
                r[3] = v2.y; g[3] = v2.z; b[3] = v2.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool ok = q < nv;
                    r[q] = ok ? p[3 * q + 0] : 0.f;
                    g[q] = ok ? p[3 * q + 1] : 0.f;
                    b[q] = ok ? p[3 * q + 2] : 0.f;
                }
            }
        } else {  // HWC4, alpha ignored (gpu_processor.py:765)
            const float4* p4 = reinterpret_cast<const float4*>(in + (irow * W + x) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nv) {
                    const float4 v = p4[q];
                    r[q] = v.x; g[q] = v.y; b[q] = v.z;
                } else {
                    r[q] = g[q] = b[q] = 0.f;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (a.use_matrix) apply_matrix(a.mat, r[q], g[q], b[q]);
            apply_lut2d(a.lut2d, r[q], g[q], b[q]);
            if (a.upto >= R2F_UPTO_DENSITY) {
                if (R2F_CURVE_BATCH3) {
                    // the pixel's three evaluations together: three gathers in flight, and the corrective gathers only if some
                    // lane of the wave sits on a breakpoint (curve_eval_batch)
                    float v3[3] = {log10_fast(r[q], a.log_eps), log10_fast(g[q], a.log_eps), log10_fast(b[q], a.log_eps)};
                    if (LDSC)
                        curve_eval_batch<3, 3>((const float4*)cells_lds, a.curve, 0, v3);
                    else
                        curve_eval_batch<3, 3>(a.curve.cells, a.curve, 0, v3);
                    r[q] = v3[0], g[q] = v3[1], b[q] = v3[2];
                } else if (LDSC) {
                    r[q] = curve_eval_at(cells_lds, a.curve, 0, log10_fast(r[q], a.log_eps));
                    g[q] = curve_eval_at(cells_lds, a.curve, 1, log10_fast(g[q], a.log_eps));
                    b[q] = curve_eval_at(cells_lds, a.curve, 2, log10_fast(b[q], a.log_eps));
                } else {
                    r[q] = log_curve(a.curve, 0, r[q], a.log_eps);
                    g[q] = log_curve(a.curve, 1, g[q], a.log_eps);
                    b[q] = log_curve(a.curve, 2, b[q], a.log_eps);
                }
            }
            if (a.upto == R2F_UPTO_OUTPUT) apply_lut3d(a.lut3d, a.lut3d_scale, a.lut3d_mode, r[q], g[q], b[q]);
        }
        if (a.upto == R2F_UPTO_OUTPUT)
            emit_hwc(a.out_f32, a.out_u8, gy - a.out_gy0, x, W, nv, vec, r, g, b);
        else
            store_planes4(a.dst, gy, x, W, nv, vec, r, g, b);
    }
}

// ------------------------------------------------------------------------------ lut3d
struct Lut3dArgs {
    DevPlanes src;
    float* out_f32;
    uint8_t* out_u8;
    int out_gy0, y0, y1, W;
    DevLut3D lut3d;
    float lut3d_scale;
    int lut3d_mode;
    int vec;
    BurnUp burn;
    DevPlanes gfield;  // has_gfield: the grain field K_g * noise computed ahead of time (tail_kernel, to_planes = 2)
    int has_gfield;
    DevCurve grain_lut;
};

// S7: the highlight map at pixel (gy, x): ndimage.zoom(map, cell, order=1) = linear interpolation at the
// coordinate o * (in - 1) / (out - 1), evaluated in double like SciPy does, then edge padding to the frame
// (effects.py:381-388).  SciPy's default mode="constant" returns 0 for a coordinate beyond the last sample,
// and o * ((in-1)/(out-1)) can exceed in-1 by one ulp at o = out-1 (e.g. 223 * (31/223) > 31): the reference's
// last zoomed column / row is then 0, and so is the edge padding copied from it.  Reproduced here on purpose.
__device__ __forceinline__ float burn_sample(const BurnUp& bu, int gy, int x) {
    const double cy = (double)min(gy, bu.h_up - 1) * bu.zy, cx = (double)min(x, bu.w_up - 1) * bu.zx;
    if (cy > (double)(bu.h_lo - 1) || cx > (double)(bu.w_lo - 1)) return 0.f;
    const int y0 = (int)cy, x0 = (int)cx;
    const int y1 = min(y0 + 1, bu.h_lo - 1), x1 = min(x0 + 1, bu.w_lo - 1);
    const float ty = (float)(cy - (double)y0), tx = (float)(cx - (double)x0);
    const float a = bu.map[y0 * bu.w_lo + x0], b = bu.map[y0 * bu.w_lo + x1];
    const float c = bu.map[y1 * bu.w_lo + x0], d = bu.map[y1 * bu.w_lo + x1];
    const float top = a + tx * (b - a), bot = c + tx * (d - c);
    return top + ty * (bot - top);
}

__global__ __launch_bounds__(256) void lut3d_kernel(const Lut3dArgs a) {
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int gy = a.y0 + blockIdx.y * 4 + threadIdx.y;
    if (x >= a.W || gy >= a.y1) return;
    const int nv = min(4, a.W - x);
    float r[4], g[4], b[4];
    load_planes4(a.src, gy, x, a.W, nv, a.vec != 0, r, g, b);
    if (a.has_gfield) {  // S6c grain.wgsl:78-89 + clip cpu_processor.py:397, with the field from the side stream
        float gr[4], gg[4], gb[4];
        load_planes4(a.gfield, gy, x, a.W, nv, a.vec != 0, gr, gg, gb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // product and sum rounded separately, like NumPy's image + grain * factor (and like the fused tail kernel)
            r[q] = fmaxf(__fadd_rn(r[q], __fmul_rn(gr[q], curve_eval(a.grain_lut, 0, r[q]))), 0.f);
            g[q] = fmaxf(__fadd_rn(g[q], __fmul_rn(gg[q], curve_eval(a.grain_lut, 1, g[q]))), 0.f);
            b[q] = fmaxf(__fadd_rn(b[q], __fmul_rn(gb[q], curve_eval(a.grain_lut, 2, b[q]))), 0.f);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (a.burn.map) {  // S7: subtract the up-sampled highlight map from all three channels, clip at 0
            const float v = a.burn.strength * burn_sample(a.burn, gy, x + q);
            r[q] = fmaxf(r[q] - v, 0.f);
            g[q] = fmaxf(g[q] - v, 0.f);
            b[q] = fmaxf(b[q] - v, 0.f);
        }
        apply_lut3d(a.lut3d, a.lut3d_scale, a.lut3d_mode, r[q], g[q], b[q]);
    }
    emit_hwc(a.out_f32, a.out_u8, gy - a.out_gy0, x, a.W, nv, a.vec != 0, r, g, b);
}

// ------------------------------------------------------------------------------ stencil
// XCD-aware block order: consecutive linear ids are dealt round-robin to the 8 XCDs, so give
// each XCD a contiguous run of tiles (neighbouring tiles share halo columns/rows in that L2).
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// Fill rows x RS floats of an LDS tile from a source plane with BORDER_REFLECT_101 on the
// global frame.  Tile element (r, c) <-> global (ty0 + r, tx0 + c).  Columns >= cols_valid are
// zero (they only meet zero padding taps).  One wave per tile row, lanes along x (coalesced);
// each lane first issues all its loads for KR rows, then writes them to LDS, so several global
// round trips are in flight per wave instead of one.
template <int NT>
__device__ __forceinline__ void fill_tile_reflect(float* lds, int RS, int rows, int cols_valid, const float* src,
                                                  int src_gy0, int src_rows, int W, int H_global, int ty0, int tx0) {
    constexpr int NW = NT / 64, KR = 4, KC = 4;
    const int tid = fresh_tid();
    const int wave = tid >> 6, lane = tid & 63;
    for (int cb = 0; cb < RS; cb += 64 * KC) {  // column blocks of 256 (one pass for stencils up to 128 taps wide)
        int sx[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) sx[k] = reflect101(tx0 + cb + lane + 64 * k, W);
        for (int r0 = wave * KR; r0 < rows; r0 += NW * KR) {
            float v[KR][KC];
#pragma unroll
            for (int i = 0; i < KR; ++i) {
                int sy = reflect101(ty0 + r0 + i, H_global) - src_gy0;
                sy = clampi(sy, 0, src_rows - 1);  // only rows feeding discarded outputs can fall outside
                const float* srow = src + (long long)sy * W;
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const int c = cb + lane + 64 * k;
                    v[i][k] = (c < cols_valid && r0 + i < rows) ? srow[sx[k]] : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < KR; ++i) {
                if (r0 + i < rows) {
                    float* drow = lds + (r0 + i) * RS;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        const int c = cb + lane + 64 * k;
                        if (c < RS) drow[c] = v[i][k];
                    }
                }
            }
        }
    }
}

// amdgpu_waves_per_eu(4): two 512-thread workgroups per CU (4 waves per SIMD) is the operating point the LDS budget is
// built around, so the register allocator must stay within 128 VGPRs.
// FR > 0: every channel of the launch is a square mirror-symmetric (2 FR + 1)^2 stencil in one LDS phase: the fully unrolled
// stencil_fixed instead of the entry list.
template <int BX, int BY, int Q, int EPI, int FR = 0>
__global__ __launch_bounds__(BX* BY) __attribute__((amdgpu_waves_per_eu(4, 8))) void stencil_kernel(const StencilArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = BX * BY, TW = 4 * BX, TH = Q * BY;
    int bx = blockIdx.x, by = blockIdx.y;
    const int ch = a.chan[blockIdx.z];
    if (a.xcd_remap) {
        // remap inside the channel's own 2-D tile grid: channels differ wildly in cost (the blue
        // halation plane is the identity), so a remap across channels would unbalance the XCDs
        const int nwg = gridDim.x * gridDim.y, lin = bx + gridDim.x * by;
        const int id = a.xcd_remap == 2 ? a.order[lin] : xcd_remap(lin, nwg);
        bx = id % gridDim.x;
        by = id / gridDim.x;
    }
    const DevStencil st = a.st[ch];
    const int tile_x0 = bx * TW, tile_y0 = a.y0 + by * TH;
    const int RS = st.RS;
    const float* src = a.src.data + (long long)ch * a.src.plane_stride;
    const int tx = threadIdx.x % BX, ty = threadIdx.x / BX;
    const float* lds_lane = smem + ty * Q * RS + 4 * tx;
    float2v acc[Q / 2][4];
#pragma unroll
    for (int j = 0; j < Q / 2; ++j)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[j][p] = (float2v){0.f, 0.f};

    if (FR > 0) {
        constexpr int R = FR > 0 ? FR : 1;
        if (a.ablate != 1)
            fill_tile_reflect<NT>(smem, RS, TH + 2 * R, TW + st.kw - 1, src, a.src.gy0, a.src.rows, a.W, a.H_global, tile_y0 - st.ay,
                                  tile_x0 - st.ax);
        __syncthreads();
        if (a.ablate != 2)
            stencil_fixed<R, Q>(lds_lane, RS, (const float2v R2F_CONSTANT*)a.fixed_w + ch * (2 * R + Q) * (R + 1) * (Q / 2), acc);
    }
    const int R2F_CONSTANT* phases = (const int R2F_CONSTANT*)st.phases;
    for (int ph = 0; ph < (FR > 0 ? 0 : st.n_phases); ++ph) {
        const int m0 = phases[4 * ph], lds_rows = phases[4 * ph + 1];
        const int row_begin = phases[4 * ph + 2], e0 = phases[4 * ph + 3];
        const int row_end = phases[4 * ph + 6];
        if (ph > 0) __syncthreads();  // all lanes are done reading the previous phase's rows
        if (a.ablate != 1)
            fill_tile_reflect<NT>(smem, RS, lds_rows, TW + st.kw - 1, src, a.src.gy0, a.src.rows, a.W, a.H_global,
                                  tile_y0 - st.ay + m0, tile_x0 - st.ax);
        __syncthreads();
        if (a.ablate != 2) {
            if (st.sym)  // per channel, wave-uniform
                stencil_accumulate_sym<Q, true>(lds_lane, st, row_begin, row_end, e0, acc);
            else
                stencil_accumulate<Q>(lds_lane, st, row_begin, row_end, e0, acc);
        }
    }

    const int tid_out = fresh_tid();  // (the output coordinates are made here, not carried through the loops above)
    const int gx = tile_x0 + 4 * (tid_out % BX);
    if (gx >= a.W) return;
    const int nv = min(4, a.W - gx);
    float* dplane = a.dst.data + (long long)ch * a.dst.plane_stride;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int gy = tile_y0 + (tid_out / BX) * Q + q;
        if (gy >= a.y1) break;
        float v[4] = {acc[q / 2][0][q & 1], acc[q / 2][1][q & 1], acc[q / 2][2][q & 1], acc[q / 2][3][q & 1]};
        if (EPI == 1) {
#pragma unroll
            for (int p = 0; p < 4; ++p) v[p] = log10_fast(v[p], a.log_eps);
            curve_eval_batch<4, 1>(a.curve.cells, a.curve, ch, v);
        }
        float* d = dplane + (long long)(gy - a.dst.gy0) * a.W + gx;
        if (a.vec && nv == 4) {
            *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int p = 0; p < nv; ++p) d[p] = v[p];
        }
    }
}

// ------------------------------------------------------------------------------ single-tap stencil channel
// The blue plane of the colour-stock halation is the identity (effects.py:248-263: f_b = 0): one tap of weight 1 at the
// anchor.  Pointwise, 4 B + 4 B per pixel, instead of the tiled kernel's LDS round trip.
__global__ __launch_bounds__(256) void single_tap_kernel(const TapArgs a) {
    const int gy = a.y0 + blockIdx.y;
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x >= a.W) return;
    const float* s = a.src.data + (long long)a.ch * a.src.plane_stride + (long long)(gy - a.src.gy0) * a.W + x;
    float* d = a.dst.data + (long long)a.ch * a.dst.plane_stride + (long long)(gy - a.dst.gy0) * a.W + x;
    const int nv = min(4, a.W - x);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.vec && nv == 4) {
        const float4 t = *reinterpret_cast<const float4*>(s);
        v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
    } else {
        for (int p = 0; p < nv; ++p) v[p] = s[p];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) v[p] = a.w * v[p];
    if (a.epilogue == 1) {
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = log10_fast(v[p], a.log_eps);
        curve_eval_batch<4, 1>(a.curve.cells, a.curve, a.ch, v);
    }
    if (a.vec && nv == 4) {
        *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int p = 0; p < nv; ++p) d[p] = v[p];
    }
}

// ------------------------------------------------------------------------------ tail (grain)
// FR > 0: the fully unrolled (2 FR + 1)^2 grain stencil (stencil_fixed: square mirror-symmetric kernels up to 19 x 19);
// FR = 0: the generic entry list, every tap column of every entry (run-time masks, i.e. branches, cost more than the
// padding columns they skip in a stencil of two entries per row step).
// SEP (with FR = R > 0): every channel's grain stencil is separable, K = u v^T to fp32 rounding (checked on the host; any
// Gaussian-like kernel is): the noise planes are filtered along x IN PLACE (every thread first reads the inputs of its row
// segments, barrier, then stores), then along y out of LDS -- 2 (2 R + 1) taps per pixel instead of (2 R + 1)^2.
template <int FR, bool SEP = false>
__global__ __launch_bounds__(kTailBX* kTailBY) void tail_kernel(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = kTailBX * kTailBY, TW = 4 * kTailBX, TH = kTailQ * kTailBY, Q = kTailQ;
    const int tile_x0 = blockIdx.x * TW, tile_y0 = a.y0 + blockIdx.y * TH;
    const DevStencil g0 = a.gk[0];
    const int RS = g0.RS;
    const int rows = TH + g0.kh - 1;
    const int plane_sz = rows * RS + 16;  // + slack: the prefetch of the dummy entry reads past the last row
    const int cols_valid = TW + g0.kw - 1;
    const bool mono = a.mono != 0;
    const uint32_t seed = a.frame->seed;  // wave-uniform: one scalar load
    // The lane's own density samples (2 rows x 4 pixels x 3 planes) are requested NOW: the kernel's only reads from HBM then
    // travel while the noise is hashed and filtered, instead of starting after it (by ablation the 0.44 ms of memory time
    // used to add to the 0.7 ms of compute; neither the compiler nor the hardware moves a load across the barriers below).
    const int ptx = threadIdx.x % kTailBX, pty = threadIdx.x / kTailBX;
    const int pgx = tile_x0 + 4 * ptx;
    float pre_r[Q][4], pre_g[Q][4], pre_b[Q][4];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int gy = tile_y0 + pty * Q + q;
#pragma unroll
        for (int p = 0; p < 4; ++p) pre_r[q][p] = pre_g[q][p] = pre_b[q][p] = 0.f;
        if (a.to_planes != 2 && pgx < a.W && gy < a.y1) load_planes4<true>(a.src, gy, pgx, a.W, min(4, a.W - pgx), a.vec != 0, pre_r[q], pre_g[q], pre_b[q]);
    }
#ifndef R2F_TAIL_LUT_BATCH
#define R2F_TAIL_LUT_BATCH 2  // pixels of a lane whose 3-D LUT gathers are in flight together (1, 2 or 4)
#endif
#ifndef R2F_TAIL_EXP
#define R2F_TAIL_EXP 0  // development switch (tools/ablate_stencil.py): bit 0 no noise generation, 1 no grain stencil, 2 no LUTs
#endif
    // S6a: hash noise for the tile + halo, straight into LDS.  Coordinates are clamped to the
    // frame like the shader's texture reads (grain.wgsl:63-75); the hash sees GLOBAL coordinates.
    // (row, column) of element idx advance with it: one division per thread, not one per element
    // Only the cols_valid columns that taps can reach are hashed (the padding columns up to RS are zeroed below): every lane
    // of a wave then does a hash, instead of the lanes that fall on padding idling through it.
    const int step_r = NT / cols_valid, step_c = NT - step_r * cols_valid;
    int r = (int)threadIdx.x / cols_valid, c = (int)threadIdx.x - r * cols_valid;
    for (int idx = threadIdx.x; idx < ((R2F_TAIL_EXP & 1) ? 0 : rows * cols_valid); idx += NT, r += step_r, c += step_c) {
        if (c >= cols_valid) c -= cols_valid, ++r;
        float nr, ng, nb;
        const int sx = clampi(tile_x0 - g0.ax + c, 0, a.W - 1);
        const int sy = clampi(tile_y0 - g0.ay + r, 0, a.H_global - 1);
        gaussian_noise((uint32_t)sx, (uint32_t)sy, seed, mono, nr, ng, nb);
        const int at = r * RS + c;
        smem[at] = nr;
        if (!mono) {
            smem[plane_sz + at] = ng;
            smem[2 * plane_sz + at] = nb;
        }
    }
    {
        const int padc = RS - cols_valid;  // < 16
        for (int idx = threadIdx.x; idx < rows * padc; idx += NT) {
            const int rr = idx / padc, at = rr * RS + cols_valid + (idx - rr * padc);
            smem[at] = 0.f;
            if (!mono) smem[plane_sz + at] = 0.f, smem[2 * plane_sz + at] = 0.f;
        }
        if (threadIdx.x < 16) {  // the slack past the last row (prefetch target of the generic entry list)
            smem[rows * RS + threadIdx.x] = 0.f;
            if (!mono) smem[plane_sz + rows * RS + threadIdx.x] = 0.f, smem[2 * plane_sz + rows * RS + threadIdx.x] = 0.f;
        }
    }
    // the grain LUT's cells next to the noise planes: 24 gathers per lane hit LDS instead of the vector L1
    const float4* gcells = a.grain_lut.cells;
    float4* cells_lds = reinterpret_cast<float4*>(smem + a.cells_off);
    if (a.cells_in_lds)
        for (int i = threadIdx.x; i < 3 * (a.grain_lut.m - 1); i += NT) cells_lds[i] = gcells[i];
    __syncthreads();

    const int tx = threadIdx.x % kTailBX, ty = threadIdx.x / kTailBX;
    float2v G[3][Q / 2][4];
    if (SEP && FR > 0) {
        // ---- pass along x, in place.  LDS column of image column tile_x0 + x is x + AX; segment s = 4 output columns
        // 4 s .. 4 s + 3 of one row of one plane; its result goes to columns 4 s .. of that row (16-byte aligned).
        constexpr int R = FR > 0 ? FR : 1, AX = fixed_stencil_ax(R), O = AX - R, NB = (O + 2 * R + 4 + 3) / 4;
        constexpr int SEGS = TW / 4, PER = (TH + 2 * R) * SEGS, MAXI = (PER + NT - 1) / NT;
        float4v held[3][MAXI];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            if (pl > 0 && mono) break;
            const float* wv = a.sep_v[pl];  // compile-time plane index: the taps stay in SGPRs
#pragma unroll
            for (int i = 0; i < MAXI; ++i) {
                const int rem = (int)threadIdx.x + NT * i;
                held[pl][i] = (float4v){0.f, 0.f, 0.f, 0.f};
                if (rem < PER) {
                    const int rr = rem / SEGS, sg = rem - rr * SEGS;
                    const float* src = smem + pl * plane_sz + rr * RS + 4 * sg;
                    float x[4 * NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const float4v v = *reinterpret_cast<const float4v*>(src + 4 * b);
                        x[4 * b] = v.x, x[4 * b + 1] = v.y, x[4 * b + 2] = v.z, x[4 * b + 3] = v.w;
                    }
                    float o[4];
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) {  // mirrored taps summed first (v is symmetric), outer taps to the centre
                        float acc = wv[0] * (x[O + pp] + x[O + pp + 2 * R]);
#pragma unroll
                        for (int j = 1; j < R; ++j) acc = fmaf(wv[j], x[O + pp + j] + x[O + pp + 2 * R - j], acc);
                        o[pp] = fmaf(wv[R], x[O + pp + R], acc);
                    }
                    held[pl][i] = (float4v){o[0], o[1], o[2], o[3]};
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            if (pl > 0 && mono) break;
#pragma unroll
            for (int i = 0; i < MAXI; ++i) {
                const int rem = (int)threadIdx.x + NT * i;
                if (rem < PER) {
                    const int rr = rem / SEGS, sg = rem - rr * SEGS;
                    *reinterpret_cast<float4v*>(smem + pl * plane_sz + rr * RS + 4 * sg) = held[pl][i];
                }
            }
        }
        __syncthreads();
        // ---- pass along y: output row y of the tile reads rows y .. y + 2 R of the filtered plane
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (c > 0 && mono && a.fixed_same) {
#pragma unroll
                for (int p = 0; p < 4; ++p) G[c][0][p] = G[0][0][p];
                continue;
            }
            const float* plane = smem + (mono ? 0 : c * plane_sz) + ty * Q * RS + 4 * tx;
            const float* wu = a.sep_u[c];
            static_assert(Q == 2, "the separable grain pass packs the lane's two output rows");
#pragma unroll
            for (int p = 0; p < 4; ++p) G[c][0][p] = (float2v){0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 2 * R + Q; ++i) {
                const float4v h = *reinterpret_cast<const float4v*>(plane + i * RS);
                const float2v w = {i <= 2 * R ? wu[i] : 0.f, i >= 1 ? wu[i - 1] : 0.f};  // (row 0's tap, row 1's tap)
                G[c][0][0] = __builtin_elementwise_fma(w, (float2v){h.x, h.x}, G[c][0][0]);
                G[c][0][1] = __builtin_elementwise_fma(w, (float2v){h.y, h.y}, G[c][0][1]);
                G[c][0][2] = __builtin_elementwise_fma(w, (float2v){h.z, h.z}, G[c][0][2]);
                G[c][0][3] = __builtin_elementwise_fma(w, (float2v){h.w, h.w}, G[c][0][3]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < ((SEP && FR > 0) ? 0 : 3); ++c) {
#pragma unroll
        for (int j = 0; j < Q / 2; ++j)
#pragma unroll
            for (int p = 0; p < 4; ++p) G[c][j][p] = (float2v){0.f, 0.f};
        const float* plane = smem + (mono ? 0 : c * plane_sz);
        if (R2F_TAIL_EXP & 2) continue;
        if (FR > 0) {

            if (c > 0 && mono && a.fixed_same) {  // one noise plane, one set of taps: the field of channel 0
#pragma unroll
                for (int p = 0; p < 4; ++p) G[c][0][p] = G[0][0][p];
            } else {
                constexpr int R = FR > 0 ? FR : 1;
                stencil_fixed<R, Q>(plane + ty * Q * RS + 4 * tx, RS,
                                    (const float2v R2F_CONSTANT*)a.fixed_w + c * (2 * R + Q) * (R + 1) * (Q / 2), G[c]);
            }
            continue;
        }
        if (a.gk[c].sym)
            stencil_accumulate_sym<Q, false>(plane + ty * Q * RS + 4 * tx, a.gk[c], 0, a.gk[c].n_rowsteps, 0, G[c]);
        else
            stencil_accumulate<Q>(plane + ty * Q * RS + 4 * tx, a.gk[c], 0, a.gk[c].n_rowsteps, 0, G[c]);
    }

    const int gx = tile_x0 + 4 * tx;
    if (gx >= a.W) return;
    const int nv = min(4, a.W - gx);
    const bool vec = a.vec != 0;
    const bool tetra_nonneg = a.lut3d_mode == 0 && a.lut3d.n >= 2 && a.lut3d.n <= 512;
    const float s3 = a.lut3d_scale * (float)(a.lut3d.n - 1);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int gy = tile_y0 + ty * Q + q;
        if (gy >= a.y1) break;
        float r[4], g[4], b[4];
        if (a.to_planes == 2) {  // the field itself: nothing of the image is read
#pragma unroll
            for (int p = 0; p < 4; ++p) r[p] = G[0][q / 2][p][q & 1], g[p] = G[1][q / 2][p][q & 1], b[p] = G[2][q / 2][p][q & 1];
            store_planes4(a.dst, gy, gx, a.W, nv, vec, r, g, b);
            continue;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) r[p] = pre_r[q][p], g[p] = pre_g[q][p], b[p] = pre_b[q][p];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // S6c grain.wgsl:78-89 + clip cpu_processor.py:397
            float ar, ag, ab;
            if (R2F_CURVE_BATCH3) {
                float v3[3] = {r[p], g[p], b[p]};
                if (a.cells_in_lds)
                    curve_eval_batch<3, 3>((const float4*)cells_lds, a.grain_lut, 0, v3);
                else
                    curve_eval_batch<3, 3>(gcells, a.grain_lut, 0, v3);
                ar = v3[0], ag = v3[1], ab = v3[2];
            } else if (a.cells_in_lds) {
                ar = curve_eval_at((const float4*)cells_lds, a.grain_lut, 0, r[p]);
                ag = curve_eval_at((const float4*)cells_lds, a.grain_lut, 1, g[p]);
                ab = curve_eval_at((const float4*)cells_lds, a.grain_lut, 2, b[p]);
            } else {
                ar = curve_eval(a.grain_lut, 0, r[p]);
                ag = curve_eval(a.grain_lut, 1, g[p]);
                ab = curve_eval(a.grain_lut, 2, b[p]);
            }
            r[p] = fmaxf(__fadd_rn(r[p], __fmul_rn(G[0][q / 2][p][q & 1], ar)), 0.f);
            g[p] = fmaxf(__fadd_rn(g[p], __fmul_rn(G[1][q / 2][p][q & 1], ag)), 0.f);
            b[p] = fmaxf(__fadd_rn(b[p], __fmul_rn(G[2][q / 2][p][q & 1], ab)), 0.f);
            if (!a.to_planes && !(R2F_TAIL_EXP & 4)) {
                if (!tetra_nonneg)
                    apply_lut3d(a.lut3d, a.lut3d_scale, a.lut3d_mode, r[p], g[p], b[p]);
                else if (R2F_TAIL_LUT_BATCH == 1)  // the clip above made every input >= 0
                    apply_lut3d_tetra_nonneg(a.lut3d, s3, r[p], g[p], b[p]);
                else if ((p + 1) % R2F_TAIL_LUT_BATCH == 0) {
                    constexpr int NB = R2F_TAIL_LUT_BATCH;
                    float rn[NB], gn[NB], bn[NB];
#pragma unroll
                    for (int k = 0; k < NB; ++k) rn[k] = r[p + 1 - NB + k], gn[k] = g[p + 1 - NB + k], bn[k] = b[p + 1 - NB + k];
                    apply_lut3d_tetra_nonneg_batch<NB>(a.lut3d, s3, rn, gn, bn);
#pragma unroll
                    for (int k = 0; k < NB; ++k) r[p + 1 - NB + k] = rn[k], g[p + 1 - NB + k] = gn[k], b[p + 1 - NB + k] = bn[k];
                }
            }
        }
        if (a.to_planes)
            store_planes4(a.dst, gy, gx, a.W, nv, vec, r, g, b);
        else
            emit_hwc(a.out_f32, a.out_u8, gy - a.out_gy0, gx, a.W, nv, vec, r, g, b);
    }
}

// ------------------------------------------------------------------------------ S7 highlight burn
// cv.resize INTER_AREA weight of source sample s for destination sample d (computeResizeAreaTab).
__device__ __forceinline__ void area_cell(int d, double scale, int ssize, int& s_first, int& s_last, int& s1, int& s2,
                                          double& w_first, double& w_full, double& w_last) {
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, (double)ssize - f1);
    s1 = (int)ceil(f1);
    s2 = min((int)floor(f2), ssize - 1);
    s1 = min(s1, s2);
    w_first = (s1 - f1 > 1e-3) ? (s1 - f1) / cell : 0.0;
    w_full = 1.0 / cell;
    w_last = (f2 - s2 > 1e-3) ? fmin(fmin(f2 - s2, 1.0), cell) / cell : 0.0;
    s_first = max(s1 - 1, 0);
    s_last = s2;
}

__device__ __forceinline__ float area_weight(int s, int s1, int s2, double w_first, double w_full, double w_last) {
    if (s == s1 - 1) return (float)w_first;
    if (s >= s1 && s < s2) return (float)w_full;
    if (s == s2) return (float)w_last;
    return 0.f;
}

// One workgroup per low-res cell: area-weighted sum of the green density over the cell's source pixels that
// lie in rows [y0, y1) (a row shard contributes its part; shards' arrays add up to the full-frame value).
__global__ __launch_bounds__(256) void burn_sums_kernel(const BurnSumsArgs a) {
    const int j = blockIdx.x, i = blockIdx.y;
    int ry0, ry1, rs1, rs2, cx0, cx1, cs1, cs2;
    double wyf, wy, wyl, wxf, wx, wxl;
    area_cell(i, (double)a.H_global / a.h_lo, a.H_global, ry0, ry1, rs1, rs2, wyf, wy, wyl);
    area_cell(j, (double)a.W / a.w_lo, a.W, cx0, cx1, cs1, cs2, wxf, wx, wxl);
    const int ra = max(ry0, a.y0), rb = min(ry1, a.y1 - 1);
    const int ncols = cx1 - cx0 + 1;
    const float* green = a.src.data + a.src.plane_stride;
    // float32 weights (cv.resize's), the products and the sum in float64 with ONE rounding of the cell's sum: the burn map goes
    // through the 3-D LUT, whose slope turns every ulp lost here into output error (a float32 sum of up to ~10^3 terms loses
    // ~1e-6 relative: 4 of 1 620 fuzz cases with hostile tables sat 3-10 % over their bound, profiles/r04_parity_budget.txt);
    // the map is tiny, the extra width costs nothing
    double acc = 0.0;
    if (rb >= ra) {
        const int n = (rb - ra + 1) * ncols;
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            const int rr = idx / ncols, cc = idx - rr * ncols;
            const int y = ra + rr, x = cx0 + cc;
            const float w = area_weight(y, rs1, rs2, wyf, wy, wyl) * area_weight(x, cs1, cs2, wxf, wx, wxl);
            acc += (double)w * (double)green[(long long)(y - a.src.gy0) * a.W + x];
        }
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.cell_sums[i * a.w_lo + j] = (float)red[0];
}

// scipy.ndimage "reflect" boundary: d c b a | a b c d | d c b a
__device__ __forceinline__ int reflect_sym(int i, int n) {
    while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i;
    return i;
}

// clip(x - d_ref, 0) then gaussian_filter(sigma=3, truncate=2): two 13-tap passes (axis 0, then axis 1), double
// accumulation and a float32 store after each pass, like scipy's correlate1d.  One workgroup; the map is tiny.
__global__ __launch_bounds__(256) void burn_map_kernel(const BurnMapArgs a) {
    const int n = a.h_lo * a.w_lo;
    float* t0 = a.scratch;
    float* t1 = a.scratch + n;
    for (int k = threadIdx.x; k < n; k += 256) t0[k] = fmaxf(a.cell_sums[k] - a.d_ref, 0.f);
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 256) {
        const int i = k / a.w_lo, j = k - i * a.w_lo;
        double acc = 0.0;
        for (int t = 0; t < 13; ++t) acc += a.w[t] * (double)t0[reflect_sym(i + t - 6, a.h_lo) * a.w_lo + j];
        t1[k] = (float)acc;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 256) {
        const int i = k / a.w_lo, j = k - i * a.w_lo;
        double acc = 0.0;
        for (int t = 0; t < 13; ++t) acc += a.w[t] * (double)t1[i * a.w_lo + reflect_sym(j + t - 6, a.w_lo)];
        a.map[k] = (float)acc;
    }
}

// ------------------------------------------------------------------------------ chroma NR (pre-path)
__device__ __forceinline__ void load_input1(const void* in_, int layout, int in_gy0, int in_rows, int W, int gy, int x,
                                            float& X, float& Y, float& Z) {
    const float* in = static_cast<const float*>(in_);
    const long long row = gy - in_gy0;
    if (layout == R2F_LAYOUT_CHW) {
        const long long plane = (long long)in_rows * W, o = row * W + x;
        X = in[o];
        Y = in[plane + o];
        Z = in[2 * plane + o];
    } else {
        const int nc = layout == R2F_LAYOUT_HWC4 ? 4 : 3;
        const float* p = in + (row * W + x) * nc;
        X = p[0];
        Y = p[1];
        Z = p[2];
    }
}

// effects.XYZ_to_xyY, effects.py:496-518
__device__ __forceinline__ void xyz_to_xy(float X, float Y, float Z, float& cx, float& cy) {
    const float denom = (X + Y) + Z;
    if (denom > 1e-8f) {
        cx = X / denom;
        cy = Y / denom;
    } else {
        cx = 0.f;
        cy = 0.f;
    }
}

// Pass 1: one workgroup = one row segment of 1024 pixels.  Chromaticities of the segment plus `radius` clamped
// neighbours on each side are staged in LDS (the one truly separable blur near this path), then every lane blurs its
// 4 pixels horizontally.  Output planes: x', y', Y.
constexpr int kChromaSeg = 1024;
__global__ __launch_bounds__(256) void chroma_h_kernel(const ChromaArgs a) {
    __shared__ float sx[kChromaSeg + 2 * 31], sy[kChromaSeg + 2 * 31];
    const int gy = a.y0 + blockIdx.y;
    const int seg0 = blockIdx.x * kChromaSeg;
    const int r = a.radius;
    const int n = kChromaSeg + 2 * r;
    float Yown[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < n; i += 256) {
        const int gx = clampi(seg0 - r + i, 0, a.W - 1);  // ix = min(max(x + i, 0), w - 1), effects.py:452
        float X, Y, Z;
        load_input1(a.in, a.in_layout, a.in_gy0, a.in_rows, a.W, gy, gx, X, Y, Z);
        xyz_to_xy(X, Y, Z, sx[i], sy[i]);
    }
    __syncthreads();
    const int x0 = seg0 + 4 * threadIdx.x;
    if (x0 >= a.W) return;
    const int nv = min(4, a.W - x0);
    float bx[4], by[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float accx = 0.f, accy = 0.f;
        const int c = 4 * threadIdx.x + p;  // window [c, c + 2r] in LDS
        for (int t = 0; t <= 2 * r; ++t) {
            accx = fmaf(sx[c + t], a.w[t], accx);
            accy = fmaf(sy[c + t], a.w[t], accy);
        }
        bx[p] = accx;
        by[p] = accy;
        if (p < nv) {
            float X, Y, Z;
            load_input1(a.in, a.in_layout, a.in_gy0, a.in_rows, a.W, gy, x0 + p, X, Y, Z);
            Yown[p] = Y;
        }
    }
    store_planes4(a.dst, gy, x0, a.W, nv, a.vec != 0, bx, by, Yown);
}

// Pass 2: vertical blur of x', y' (rows clamped to the frame) and effects.xyY_to_XYZ (effects.py:521-544).
__global__ __launch_bounds__(256) void chroma_v_kernel(const ChromaArgs a) {
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int gy = a.y0 + blockIdx.y * 4 + threadIdx.y;
    if (x >= a.W || gy >= a.y1) return;
    const int nv = min(4, a.W - x);
    const bool vec = a.vec != 0;
    const int r = a.radius;
    float ax[4] = {0.f, 0.f, 0.f, 0.f}, ay[4] = {0.f, 0.f, 0.f, 0.f};
    float Yc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t <= 2 * r; ++t) {
        const int sy = clampi(gy - r + t, 0, a.H_global - 1);  // iy = min(max(y + i, 0), h - 1), effects.py:476
        float px[4], py[4], pY[4];
        load_planes4(a.src, sy, x, a.W, nv, vec, px, py, pY);
        const float w = a.w[t];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ax[p] = fmaf(px[p], w, ax[p]);
            ay[p] = fmaf(py[p], w, ay[p]);
            if (t == r) Yc[p] = pY[p];
        }
    }
    float X[4], Z[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (ay[p] > 1e-8f) {
            const float inv = Yc[p] / ay[p];
            X[p] = ax[p] * inv;
            Z[p] = ((1.0f - ax[p]) - ay[p]) * inv;
        } else {
            X[p] = 0.f;
            Yc[p] = 0.f;
            Z[p] = 0.f;
        }
    }
    store_planes4(a.dst, gy, x, a.W, nv, vec, X, Yc, Z);
}

// ------------------------------------------------------------------------------ area down-scale (pre-path)
// One lane per destination pixel: weighted mean over its source footprint with the INTER_AREA weights of
// area_cell() above, all three channels at once.  Pre-path and run once per preview; no tuning beyond coalesced x.
__global__ __launch_bounds__(256) void resize_area_kernel(const ResizeArgs a) {
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= a.out_w || dy >= a.out_h) return;
    int ry0, ry1, rs1, rs2, cx0, cx1, cs1, cs2;
    double wyf, wy, wyl, wxf, wx, wxl;
    area_cell(dy, (double)a.H / a.out_h, a.H, ry0, ry1, rs1, rs2, wyf, wy, wyl);
    area_cell(dx, (double)a.W / a.out_w, a.W, cx0, cx1, cs1, cs2, wxf, wx, wxl);
    float accX = 0.f, accY = 0.f, accZ = 0.f;
    for (int y = ry0; y <= ry1; ++y) {
        const float wv = area_weight(y, rs1, rs2, wyf, wy, wyl);
        if (wv == 0.f) continue;
        float rX = 0.f, rY = 0.f, rZ = 0.f;
        for (int x = cx0; x <= cx1; ++x) {
            const float wh = area_weight(x, cs1, cs2, wxf, wx, wxl);
            float X, Y, Z;
            load_input1(a.in, a.in_layout, 0, a.H, a.W, y, x, X, Y, Z);
            rX = fmaf(wh, X, rX);
            rY = fmaf(wh, Y, rY);
            rZ = fmaf(wh, Z, rZ);
        }
        accX = fmaf(wv, rX, accX);
        accY = fmaf(wv, rY, accY);
        accZ = fmaf(wv, rZ, accZ);
    }
    float* p0 = a.dst.data + (long long)(dy - a.dst.gy0) * a.out_w + dx;
    p0[0] = accX;
    p0[a.dst.plane_stride] = accY;
    p0[2 * a.dst.plane_stride] = accZ;
}

// ------------------------------------------------------------------------------ free rotation (pre-path)
// effects.rotate (effects.py:46-75): cv.warpAffine(rgb, getRotationMatrix2D(centre, -degrees, 1), same size, INTER_LINEAR)
// followed by a centred crop; the kernel produces the cropped window only.  One lane per destination pixel, lanes along x.
// Source coordinates and the two-step lerp are float32, like OpenCV's linear warp kernels (>= 4.11); taps that fall
// outside the frame read the constant border 0.  HBM/L2-bound gather: neighbouring lanes read neighbouring texels for
// the small angles a horizon correction uses.
__global__ __launch_bounds__(256) void warp_affine_kernel(const WarpArgs a) {
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= a.out_w || dy >= a.out_h) return;
    const float xf = (float)(dx + a.ox), yf = (float)(dy + a.oy);
    const float sx = __fadd_rn(__fadd_rn(__fmul_rn(xf, a.m[0]), __fmul_rn(yf, a.m[1])), a.m[2]);
    const float sy = __fadd_rn(__fadd_rn(__fmul_rn(xf, a.m[3]), __fmul_rn(yf, a.m[4])), a.m[5]);
    const float fx0 = floorf(sx), fy0 = floorf(sy);
    const float ax = sx - fx0, ay = sy - fy0;
    float v[3] = {0.f, 0.f, 0.f};
    // int conversion only for coordinates that can touch the frame (also keeps huge values out of the cast)
    if (fx0 >= -1.f && fy0 >= -1.f && fx0 < (float)a.W && fy0 < (float)a.H) {
        const int x0 = (int)fx0, y0 = (int)fy0;
        float t[2][2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int xx = x0 + i, yy = y0 + j;
                if (xx >= 0 && xx < a.W && yy >= 0 && yy < a.H)
                    load_input1(a.in, a.in_layout, 0, a.H, a.W, yy, xx, t[j][i][0], t[j][i][1], t[j][i][2]);
                else
                    t[j][i][0] = t[j][i][1] = t[j][i][2] = 0.f;
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = __fadd_rn(t[0][0][c], __fmul_rn(ax, __fsub_rn(t[0][1][c], t[0][0][c])));
            const float bot = __fadd_rn(t[1][0][c], __fmul_rn(ax, __fsub_rn(t[1][1][c], t[1][0][c])));
            v[c] = __fadd_rn(top, __fmul_rn(ay, __fsub_rn(bot, top)));
        }
    }
    float* p0 = a.dst.data + (long long)(dy - a.dst.gy0) * a.out_w + dx;
    p0[0] = v[0];
    p0[a.dst.plane_stride] = v[1];
    p0[2 * a.dst.plane_stride] = v[2];
}

// ------------------------------------------------------------------------------ LANCZOS4 up-scale (post-path)
// utils.resolution_scaling -> cv.resize(uint8, INTER_LANCZOS4) (utils.py:237-242): 8 x 8 taps per output pixel with the
// 11-bit fixed-point weights OpenCV derives per destination column / row (built on the host, r2f_api.hip), exact int32
// accumulation, one rounding (+2^21 >> 22), replicated border.  One lane per output pixel, all three channels.
__global__ __launch_bounds__(256) void lanczos4_u8_kernel(const LanczosArgs a) {
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= a.out_w || dy >= a.out_h) return;
    const int sx = a.xofs[dx] - 3, sy = a.yofs[dy] - 3;
    int wx[8], wy[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) wx[k] = a.xcoef[dx * 8 + k], wy[k] = a.ycoef[dy * 8 + k];
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint8_t* row = a.src + (long long)clampi(sy + k, 0, a.H - 1) * a.W * 3;
        int h[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint8_t* p = row + clampi(sx + j, 0, a.W - 1) * 3;
            h[0] += (int)p[0] * wx[j];
            h[1] += (int)p[1] * wx[j];
            h[2] += (int)p[2] * wx[j];
        }
        acc[0] += h[0] * wy[k];
        acc[1] += h[1] * wy[k];
        acc[2] += h[2] * wy[k];
    }
    uint8_t* o = a.dst + ((long long)dy * a.out_w + dx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)clampi((acc[c] + (1 << 21)) >> 22, 0, 255);
}

// ------------------------------------------------------------------------------ noise (test)
__global__ __launch_bounds__(256) void noise_kernel(const NoiseArgs a) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int gy = a.y0 + blockIdx.y;
    if (x >= a.W) return;
    const long long rows = a.y1 - a.y0;
    const long long plane = rows * a.W;
    const long long o = (long long)(gy - a.y0) * a.W + x;
    const uint32_t seed = a.frame->seed;
    if (a.hash) {
        uint32_t vx = (uint32_t)x, vy = (uint32_t)gy, vz = seed;
        pcg3d(vx, vy, vz);
        a.hash[o] = vx;
        a.hash[plane + o] = vy;
        a.hash[2 * plane + o] = vz;
    }
    if (a.noise) {
        float nr, ng, nb;
        gaussian_noise((uint32_t)x, (uint32_t)gy, seed, a.mono != 0, nr, ng, nb);
        a.noise[o] = nr;
        a.noise[plane + o] = ng;
        a.noise[2 * plane + o] = nb;
    }
}

// ------------------------------------------------------------------------------ histogram
// utils.generate_histogram's counting loop (utils.py:160-165) / histogram.wgsl pass1_accumulate on the uint8 (H, W, 3)
// output: 3 x 256 counts.  HBM-bound (3 B/px); the byte stream is read 16 B per lane, channel = byte index mod 3.  Flat
// images put most pixels into a few bins, so every group of 8 lanes owns a private copy of the table in LDS (same-address
// LDS atomics serialise) and the copies are folded into global memory once per workgroup.
constexpr int kHistCopies = 8, kHistThreads = 256, kHistBytesPerLane = 16, kHistIters = 16;

__global__ __launch_bounds__(kHistThreads) void histogram_u8_kernel(const uint8_t* __restrict__ image, long long n_bytes,
                                                                    uint32_t* __restrict__ counts) {
    __shared__ uint32_t h[kHistCopies][768];
    for (int i = threadIdx.x; i < kHistCopies * 768; i += kHistThreads) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t* mine = h[threadIdx.x & (kHistCopies - 1)];
    const long long chunk = (long long)kHistThreads * kHistBytesPerLane;
    long long base = (long long)blockIdx.x * chunk * kHistIters;
    for (int it = 0; it < kHistIters; ++it, base += chunk) {
        const long long o = base + (long long)threadIdx.x * kHistBytesPerLane;
        if (o >= n_bytes) break;
        int ch = (int)(o % 3);
        if (o + kHistBytesPerLane <= n_bytes) {
            const uint4 v = *reinterpret_cast<const uint4*>(image + o);  // hipMalloc'ed images are 16-byte aligned
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    atomicAdd(&mine[ch * 256 + ((w[k] >> (8 * b)) & 255u)], 1u);
                    ch = ch == 2 ? 0 : ch + 1;
                }
        } else {
            for (long long i = o; i < n_bytes; ++i) {
                atomicAdd(&mine[ch * 256 + image[i]], 1u);
                ch = ch == 2 ? 0 : ch + 1;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += kHistThreads) {
        uint32_t s = 0;
#pragma unroll
        for (int c = 0; c < kHistCopies; ++c) s += h[c][i];
        if (s) atomicAdd(&counts[i], s);
    }
}

// ------------------------------------------------------------------------------ launchers
size_t stencil_lds_bytes(const StencilVariant& v, const DevStencil* st, int nchan) {
    size_t best = 0;
    for (int c = 0; c < nchan; ++c) {
        size_t b = ((size_t)st[c].RS * (size_t)st[c].max_lds_rows + 16) * sizeof(float);
        if (b > best) best = b;
    }
    return best;
}

// noise planes, then (when they fit next to them in half a CU's LDS) the grain LUT's cells
size_t tail_plane_floats(const DevStencil* gk, int mono) {
    const size_t plane = (size_t)gk[0].RS * (size_t)gk[0].max_lds_rows + 16;
    return (plane * (mono ? 1 : 3) + 3) / 4 * 4;
}
size_t tail_lds_bytes(const DevStencil* gk, int mono, int cells_in_lds, int grain_m) {
    return tail_plane_floats(gk, mono) * sizeof(float) + (cells_in_lds ? (size_t)3 * (grain_m - 1) * sizeof(float4) : 0);
}

hipError_t init_kernel_attributes() {
    hipError_t e;
#define R2F_SET_LDS(k)                                                                                \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)kMaxLds);                                                            \
    if (e != hipSuccess) return e;
    R2F_SET_LDS((stencil_kernel<32, 8, 4, 0>))
    R2F_SET_LDS((stencil_kernel<32, 8, 4, 1>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 0>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 1>))
    R2F_SET_LDS((stencil_kernel<16, 8, 4, 0>))
    R2F_SET_LDS((stencil_kernel<16, 8, 4, 1>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 0, 1>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 1, 1>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 0, 2>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 1, 2>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 0, 3>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 1, 3>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 0, 4>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 1, 4>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 0, 5>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 1, 5>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 0, 6>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 1, 6>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 0, 7>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 1, 7>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 0, 8>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 1, 8>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 0, 9>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 1, 9>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 0, 10>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 1, 10>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 0, 11>))
    R2F_SET_LDS((stencil_kernel<32, 16, 4, 1, 11>))
    R2F_SET_LDS((tail_kernel<0>))
    R2F_SET_LDS((tail_kernel<1>))
    R2F_SET_LDS((tail_kernel<2>))
    R2F_SET_LDS((tail_kernel<3>))
    R2F_SET_LDS((tail_kernel<4>))
    R2F_SET_LDS((tail_kernel<5>))
    R2F_SET_LDS((tail_kernel<6>))
    R2F_SET_LDS((tail_kernel<7>))
    R2F_SET_LDS((tail_kernel<8>))
    R2F_SET_LDS((tail_kernel<9>))
    R2F_SET_LDS((tail_kernel<1, true>))
    R2F_SET_LDS((tail_kernel<2, true>))
    R2F_SET_LDS((tail_kernel<3, true>))
    R2F_SET_LDS((tail_kernel<4, true>))
    R2F_SET_LDS((tail_kernel<5, true>))
    R2F_SET_LDS((tail_kernel<6, true>))
    R2F_SET_LDS((tail_kernel<7, true>))
    R2F_SET_LDS((tail_kernel<8, true>))
    R2F_SET_LDS((tail_kernel<9, true>))
    R2F_SET_LDS(front_kernel<true>)
#undef R2F_SET_LDS
    return hipSuccess;
}

hipError_t launch_front(const FrontArgs& a, hipStream_t s) {
    if (a.y1 <= a.y0 || a.W <= 0) return hipSuccess;
    if (a.fast && front_fast_eligible(a)) return launch_front_fast(a, s);
    const int quads = (a.W + 3) / 4;
    const int gx = (quads + 63) / 64;
    const int row_groups = (a.y1 - a.y0 + kFrontBY - 1) / kFrontBY;
    dim3 block(64, kFrontBY);
    const size_t cell_bytes = (size_t)3 * (a.curve.m > 1 ? a.curve.m - 1 : 0) * sizeof(float4);
    const bool ldsc = a.upto >= R2F_UPTO_DENSITY && cell_bytes > 0 && cell_bytes <= 64 * 1024;
    if (ldsc) {
        // persistent-ish grid: a few blocks per CU in total, so the table copy is amortised over many rows.  Three blocks
        // are resident per CU at 48 KB of cells each; 24 MP LUT-only frame: 0.51 / 0.35 / 0.26 / 0.25 ms at 1 / 2 / 3 / 6 per CU
        int gy = ((a.blocks_per_cu > 0 ? a.blocks_per_cu : 6) * 256 + gx - 1) / gx;
        if (gy > row_groups) gy = row_groups;
        if (gy < 1) gy = 1;
        launch_k(front_kernel<true>, dim3(gx, gy), block, cell_bytes, s, a);
    } else {
        launch_k(front_kernel<false>, dim3(gx, row_groups), block, 0, s, a);
    }
    return take_launch_status();
}

hipError_t launch_stencil(const StencilArgs& a, int variant, hipStream_t s) {
    if (a.y1 <= a.y0 || a.W <= 0) return hipSuccess;
    const StencilVariant& v = kStencilVariants[variant];
    const size_t lds = stencil_lds_bytes(v, a.st, 3);  // any subset of the channels may be in a.chan
    dim3 block(v.BX * v.BY), grid((a.W + v.TW() - 1) / v.TW(), (a.y1 - a.y0 + v.TH() - 1) / v.TH(), a.nchan);
    if (variant == 0 && a.fixed_r > 0) {  // small square mirror-symmetric stencils: the unrolled form
#define R2F_STENCIL_FIXED(R)                                                                        \
    case R:                                                                                        \
        if (a.epilogue == 1)                                                                       \
            launch_k((stencil_kernel<32, 16, 4, 1, R>), grid, block, lds, s, a);         \
        else                                                                                       \
            launch_k((stencil_kernel<32, 16, 4, 0, R>), grid, block, lds, s, a);         \
        return take_launch_status();
        switch (a.fixed_r) {
            R2F_STENCIL_FIXED(1)
            R2F_STENCIL_FIXED(2)
            R2F_STENCIL_FIXED(3)
            R2F_STENCIL_FIXED(4)
            R2F_STENCIL_FIXED(5)
            R2F_STENCIL_FIXED(6)
            R2F_STENCIL_FIXED(7)
            R2F_STENCIL_FIXED(8)
            R2F_STENCIL_FIXED(9)
            R2F_STENCIL_FIXED(10)
            R2F_STENCIL_FIXED(11)
            default:
                break;
        }
#undef R2F_STENCIL_FIXED
    }
    const int key = variant * 2 + (a.epilogue == 1 ? 1 : 0);
    switch (key) {
        case 0: launch_k((stencil_kernel<32, 16, 4, 0>), grid, block, lds, s, a); break;
        case 1: launch_k((stencil_kernel<32, 16, 4, 1>), grid, block, lds, s, a); break;
        case 2: launch_k((stencil_kernel<16, 8, 4, 0>), grid, block, lds, s, a); break;
        case 3: launch_k((stencil_kernel<16, 8, 4, 1>), grid, block, lds, s, a); break;
        case 4: launch_k((stencil_kernel<32, 8, 4, 0>), grid, block, lds, s, a); break;
        default: launch_k((stencil_kernel<32, 8, 4, 1>), grid, block, lds, s, a); break;
    }
    return take_launch_status();
}

hipError_t launch_tail(const TailArgs& a, hipStream_t s) {
    if (a.y1 <= a.y0 || a.W <= 0) return hipSuccess;
    if (!a.grain) {
        Lut3dArgs l;
        l.src = a.src;
        l.out_f32 = a.out_f32;
        l.out_u8 = a.out_u8;
        l.out_gy0 = a.out_gy0;
        l.y0 = a.y0;
        l.y1 = a.y1;
        l.W = a.W;
        l.lut3d = a.lut3d;
        l.lut3d_scale = a.lut3d_scale;
        l.lut3d_mode = a.lut3d_mode;
        l.vec = a.vec;
        l.burn = a.burn;
        l.gfield = a.gfield;
        l.has_gfield = a.has_gfield;
        l.grain_lut = a.grain_lut;
        const int quads = (a.W + 3) / 4;
        dim3 block(64, 4), grid((quads + 63) / 64, (a.y1 - a.y0 + 3) / 4);
        launch_k(lut3d_kernel, grid, block, 0, s, l);
        return take_launch_status();
    }
    const int TW = 4 * kTailBX, TH = kTailQ * kTailBY;
    dim3 block(kTailBX * kTailBY), grid((a.W + TW - 1) / TW, (a.y1 - a.y0 + TH - 1) / TH);
    TailArgs b = a;
    b.cells_off = (int)tail_plane_floats(a.gk, a.mono);
    b.cells_in_lds = tail_lds_bytes(a.gk, a.mono, 1, a.grain_lut.m) <= 80 * 1024 ? 1 : 0;  // keep two workgroups per CU
    const size_t lds = tail_lds_bytes(a.gk, a.mono, b.cells_in_lds, a.grain_lut.m);
    switch (a.fixed_r) {  // small square mirror-symmetric stencils take the unrolled form, separable ones two 1-D passes
#define R2F_TAIL_FIXED(R)                                                        \
    case R:                                                                     \
        if (a.sep)                                                              \
            launch_k((tail_kernel<R, true>), grid, block, lds, s, b); \
        else                                                                    \
            launch_k((tail_kernel<R>), grid, block, lds, s, b);       \
        break;
        R2F_TAIL_FIXED(1)
        R2F_TAIL_FIXED(2)
        R2F_TAIL_FIXED(3)
        R2F_TAIL_FIXED(4)
        R2F_TAIL_FIXED(5)
        R2F_TAIL_FIXED(6)
        R2F_TAIL_FIXED(7)
        R2F_TAIL_FIXED(8)
        R2F_TAIL_FIXED(9)
#undef R2F_TAIL_FIXED
        default:
            launch_k((tail_kernel<0>), grid, block, lds, s, b);
            break;
    }
    return take_launch_status();
}

hipError_t launch_burn_sums(const BurnSumsArgs& a, hipStream_t s) {
    if (a.h_lo <= 0 || a.w_lo <= 0) return hipSuccess;
    launch_k(burn_sums_kernel, dim3(a.w_lo, a.h_lo), dim3(256), 0, s, a);
    return take_launch_status();
}

hipError_t launch_burn_map(const BurnMapArgs& a, hipStream_t s) {
    launch_k(burn_map_kernel, dim3(1), dim3(256), 0, s, a);
    return take_launch_status();
}

hipError_t launch_chroma_h(const ChromaArgs& a, hipStream_t s) {
    if (a.y1 <= a.y0 || a.W <= 0) return hipSuccess;
    launch_k(chroma_h_kernel, dim3((a.W + kChromaSeg - 1) / kChromaSeg, a.y1 - a.y0), dim3(256), 0, s, a);
    return take_launch_status();
}

hipError_t launch_chroma_v(const ChromaArgs& a, hipStream_t s) {
    if (a.y1 <= a.y0 || a.W <= 0) return hipSuccess;
    const int quads = (a.W + 3) / 4;
    launch_k(chroma_v_kernel, dim3((quads + 63) / 64, (a.y1 - a.y0 + 3) / 4), dim3(64, 4), 0, s, a);
    return take_launch_status();
}

hipError_t launch_resize_area(const ResizeArgs& a, hipStream_t s) {
    if (a.out_h <= 0 || a.out_w <= 0) return hipSuccess;
    launch_k(resize_area_kernel, dim3((a.out_w + 63) / 64, (a.out_h + 3) / 4), dim3(64, 4), 0, s, a);
    return take_launch_status();
}

hipError_t launch_warp_affine(const WarpArgs& a, hipStream_t s) {
    if (a.out_h <= 0 || a.out_w <= 0) return hipSuccess;
    launch_k(warp_affine_kernel, dim3((a.out_w + 63) / 64, (a.out_h + 3) / 4), dim3(64, 4), 0, s, a);
    return take_launch_status();
}

hipError_t launch_single_tap(const TapArgs& a, hipStream_t s) {
    if (a.y1 <= a.y0 || a.W <= 0) return hipSuccess;
    launch_k(single_tap_kernel, dim3((a.W + 1023) / 1024, a.y1 - a.y0), dim3(256), 0, s, a);
    return take_launch_status();
}

hipError_t launch_lanczos4_u8(const LanczosArgs& a, hipStream_t s) {
    if (a.out_h <= 0 || a.out_w <= 0) return hipSuccess;
    launch_k(lanczos4_u8_kernel, dim3((a.out_w + 63) / 64, (a.out_h + 3) / 4), dim3(64, 4), 0, s, a);
    return take_launch_status();
}

hipError_t launch_noise(const NoiseArgs& a, hipStream_t s) {
    if (a.y1 <= a.y0 || a.W <= 0) return hipSuccess;
    dim3 block(256), grid((a.W + 255) / 256, a.y1 - a.y0);
    launch_k(noise_kernel, grid, block, 0, s, a);
    return take_launch_status();
}

// mode 0: only the seed (a stage entry's own write in the middle of a frame keeps the exposure range the front kernel recorded);
// 1: the whole block (seed + range reset: the start of a frame); 2: only the range reset (a caller that keeps the seed resident);
// 3: the range is made unusable (max = +inf) -- a front kernel that was asked to record the range of what it writes and cannot (its
// rows' TILES simply stay unwritten, which the per-window choice reads as "unknown": complex128).
// Modes 1 and 2 also reset the record's tile grid ({+inf, 0} = unknown), one thread per tile.
__global__ __launch_bounds__(256) void frame_params_kernel(FrameParams* dst, const FrameParams v, const int mode, int2* tiles, const int n_tiles) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        if (mode == 1)
            *dst = v;
        else if (mode == 0)
            dst->seed = v.seed;
        else if (mode == 2)
            dst->e_min = v.e_min, dst->e_max = v.e_max;
        else
            dst->e_max = 0x7f800000u;
    }
    if ((mode == 1 || mode == 2) && i < n_tiles) tiles[i] = make_int2((int)kFrameMinReset, (int)kFrameMaxReset);
}

hipError_t launch_frame_params(const RangeRecord& rec, const FrameParams& v, int mode, hipStream_t s) {
    const int n_tiles = (rec.tiles && (mode == 1 || mode == 2)) ? rec.tyn * rec.txn : 0;
    launch_k(frame_params_kernel, dim3(n_tiles > 0 ? (n_tiles + 255) / 256 : 1), dim3(n_tiles > 0 ? 256 : 1), 0, s, rec.blk, v, mode, rec.tiles,
             n_tiles);
    return take_launch_status();
}

// min / max |.| of rows [y0, y1) and [y2, y3) of the planes in `mask`, merged into the record's tiles like the front kernel's own
// (r2f_stage_exposure_range: the halo rows a row shard received from its neighbours above and below, one launch for both bands).
// A wave takes 256 columns of one row -- one tile column, like a wave of the front kernel --, a workgroup 16 consecutive rows whose
// waves combine through LDS before one of them merges into the (one or two) tiles they touch: grid (ceil(W / 256), ceil(rows / 16)),
// block (64, 16).  NaNs drop out of fminf / fmaxf, like there.
constexpr int kRangeRowsPerBlock = 16;
__global__ __launch_bounds__(64 * kRangeRowsPerBlock) void exposure_range_kernel(const DevPlanes src, const int y0, const int y1, const int y2,
                                                                                  const int y3, const int W, const int mask,
                                                                                  const RangeRecord rec) {
    __shared__ int wg_ty[kRangeRowsPerBlock];
    __shared__ float wg_lo[kRangeRowsPerBlock], wg_hi[kRangeRowsPerBlock];
    const int n0 = y1 - y0, r = blockIdx.y * kRangeRowsPerBlock + threadIdx.y;
    const bool live = r < n0 + (y3 - y2);  // (wave-uniform)
    const int gy = r < n0 ? y0 + r : y2 + (r - n0);
    float lo = __builtin_inff(), hi = 0.f;
    if (live) {
        const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
        for (int c = 0; c < 3; ++c) {
            if (!((mask >> c) & 1)) continue;
            const float* p = src.data + c * src.plane_stride + (long long)(gy - src.gy0) * W;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x0 + k < W) {
                    const float v = p[x0 + k];
                    lo = range_min(lo, v), hi = fmaxf(hi, fabsf(v));
                }
        }
    }
    lo = wave_extreme<false>(lo), hi = wave_extreme<true>(hi);
    if (threadIdx.x == 63) wg_ty[threadIdx.y] = live ? (gy >> kRangeTileRowsLog2) : -1, wg_lo[threadIdx.y] = lo, wg_hi[threadIdx.y] = hi;
    __syncthreads();
    if (threadIdx.y == 0 && threadIdx.x == 0) {
        for (int i = 0; i < kRangeRowsPerBlock; ++i) {
            const int ty = wg_ty[i];
            if (ty < 0) continue;
            float l2 = wg_lo[i], h2 = wg_hi[i];
            for (int j = i + 1; j < kRangeRowsPerBlock; ++j)
                if (wg_ty[j] == ty) l2 = fminf(l2, wg_lo[j]), h2 = fmaxf(h2, wg_hi[j]), wg_ty[j] = -1;
            merge_tile(rec, ty << kRangeTileRowsLog2, blockIdx.x, l2, h2);
        }
    }
}

hipError_t launch_exposure_range(const DevPlanes& src, int y0, int y1, int y2, int y3, int W, int mask, const RangeRecord& rec, hipStream_t s) {
    if (y1 < y0) y1 = y0;
    if (y3 < y2) y3 = y2;
    const int rows = (y1 - y0) + (y3 - y2);
    if (rows <= 0 || W <= 0 || !(mask & 7)) return hipSuccess;
    launch_k(exposure_range_kernel, dim3((W + kRangeTileCols - 1) / kRangeTileCols, (rows + kRangeRowsPerBlock - 1) / kRangeRowsPerBlock),
             dim3(64, kRangeRowsPerBlock), 0, s, src, y0, y1, y2, y3, W, mask, rec);
    return take_launch_status();
}

hipError_t launch_histogram_u8(const uint8_t* image, long long n_bytes, uint32_t* counts, hipStream_t s) {
    hipError_t e = hipMemsetAsync(counts, 0, 768 * sizeof(uint32_t), s);
    if (e != hipSuccess || n_bytes <= 0) return e;
    const long long per_block = (long long)kHistThreads * kHistBytesPerLane * kHistIters;
    launch_k(histogram_u8_kernel, dim3((unsigned)((n_bytes + per_block - 1) / per_block)), dim3(kHistThreads), 0, s,
                       image, n_bytes, counts);
    return take_launch_status();
}

}  // namespace r2f
