// r2f_front.hip -- the pointwise front of the path, S0 + S1 [+ S3 + S4 [+ S8 (+ S9)]], specialised for what a real frame looks
// like: W a multiple of 4 with aligned buffers, a 2-D input LUT of at most 64 x 64 texels, a density curve on a near-uniform
// axis small enough for LDS, tetrahedral output LUT.  Everything else keeps the generic front_kernel of r2f_kernels.hip (same
// arithmetic, more run-time switches).  upto = OUTPUT is BASELINE config 2 ("negative + print LUTs, effects off": one kernel,
// 12 B/px in, 12 B/px out); upto = EXPOSURE is the first kernel of the full pipeline.
//
// What bounds these passes is the rate at which a CU can gather LUT texels, not HBM and not the VALU: a divergent 16-byte
// global gather costs ~1.1 cycles per LANE per CU out of a 64 KB table and ~2 out of a 575 KB one, whatever the lanes' lines
// have in common (tools/ubench/gather_rate.hip, profiles/r02_gather_rate.txt: 0.9 lanes/clk/CU; 4 for a broadcast; 4+ for
// random ds_read_b128).  With 3 (2-D LUT) + 3 (curve) + 4 (3-D LUT) gathers per pixel the generic kernel's 0.28 ms on the
// 24 MP frame IS 7 global gathers per pixel at 1.1 lanes per clock.  So:
//   * the 2-D LUT (64 KB of float4 texels at n = 64) sits in LDS beside the curve's cells: only the 3-D LUT's four corners
//     (575 KB: L2) are left on the vector-memory path;
// and, against the instruction count (the generic kernel issues ~300 VALU instructions per pixel):
//   * LUT gathers address their table with a 32-bit byte offset from a wave-uniform base (SGPR pair + one VGPR) instead of
//     64-bit pointer arithmetic per gather;
//   * the three colour channels travel as (x, y) register pairs + z, so blends are v_pk_fma_f32 / v_pk_mul_f32;
//   * the 3-D LUT's cell indices take the non-negative path (min instead of the Python-style negative-index wrap) whenever no
//     lane of the wave holds a negative density -- always, after a density curve -- and the tetrahedron's corner offsets are
//     picked from three wave-uniform constants (hi - lo == 1 on that path) by the reference's own comparison rules
//     (utils.py:298-376), the sorted fractions by v_max3 / v_med3 / v_min3.
// Results are bit-identical to the generic kernel's (tests/test_gpu_parity.py::test_fused_pointwise_fast_path_matches_generic).
// Workgroups are persistent: 1 024 threads and up to 112 KB of LDS (one per CU) when the curve is needed, 512 threads and the
// 64 KB LUT alone (two per CU) for upto = EXPOSURE; each copies its tables once and walks the frame.
#include <algorithm>
#include <type_traits>

#include "r2f_launch.h"

#include "../../include/r2f.h"

#ifndef R2F_FRONT_LUT_BATCH
#define R2F_FRONT_LUT_BATCH 4  // pixels of a lane whose 3-D LUT gathers are in flight together (1, 2 or 4)
#endif

namespace r2f {

namespace {

struct V3 {
    float2v xy;
    float z;
};

__device__ __forceinline__ float2v splat(float v) { return (float2v){v, v}; }

// float4 texel at byte offset `off` of a table with a wave-uniform base
__device__ __forceinline__ V3 texel(const void* base, unsigned off) {
    const float4 t = *reinterpret_cast<const float4*>(static_cast<const char*>(base) + off);
    return V3{(float2v){t.x, t.y}, t.z};
}

__device__ __forceinline__ uint32_t u8_of(float v) {  // cpu_processor.py:407: (image * 255).astype(uint8) -- truncation
    return (uint32_t)__builtin_amdgcn_fmed3f(v * 255.0f, 0.0f, 255.0f);
}

// S0, apply_matrix of r2f_device.h on (x, y) pairs: ((m0 r + m1 g) + m2 b) per row.  The matrix travels as three column pairs
// + its last row, built once per kernel from the kernel arguments (indexing Mat3::m pair by pair made hipcc bounce the first
// four elements through a 24-byte stack slot in the EXPOSURE variants).
struct Mat3Pairs {
    float2v c0, c1, c2;  // (m0, m3), (m1, m4), (m2, m5)
    float r0, r1, r2;    // m6, m7, m8
};
__device__ __forceinline__ Mat3Pairs pairs_of(const Mat3& M) {
    // (readfirstlane: the elements become opaque scalars, so the pairs are assembled with register moves)
    float m[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(M.m[i])));
    return Mat3Pairs{(float2v){m[0], m[3]}, (float2v){m[1], m[4]}, (float2v){m[2], m[5]}, M.m[6], M.m[7], M.m[8]};
}
__device__ __forceinline__ V3 matrix3(const Mat3Pairs& M, float r, float g, float b) {
    V3 o;
    o.xy = __builtin_elementwise_fma(M.c2, splat(b), __builtin_elementwise_fma(M.c1, splat(g), M.c0 * splat(r)));
    o.z = fmaf(M.r2, b, fmaf(M.r1, g, M.r0 * r));
    return o;
}

// float4 texel at byte offset `off` of a table in LDS
__device__ __forceinline__ V3 texel_lds(const float4* base, unsigned off) {
    const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
    return V3{(float2v){t.x, t.y}, t.z};
}

// S1, apply_lut2d of r2f_device.h (lut_2d.wgsl:18-108); the LUT's texels are in LDS
__device__ __forceinline__ V3 lut2d(const float4* tex, const int n, const V3 p) {
#pragma clang fp contract(off)  // fract() of the rounded coordinate (see apply_lut2d); the blend is explicit fma
    const float S = (p.xy.x + p.xy.y) + p.z;
    const bool dark = S < 1e-12f;
    const float inv_sum = (float)(n - 1) / (dark ? 1.0f : S);
    const float2v rg = p.xy * splat(inv_sum);
    const float fr = floorf(rg.x), fg = floorf(rg.y);
    const int ri = clampi((int)fr, 0, n - 2), gi = clampi((int)fg, 0, n - 2);
    const float rf = rg.x - fr, gf = rg.y - fg;
    const float fsum = rf + gf;
    const bool lower = fsum <= 1.0f;
    const unsigned n16 = (unsigned)n << 4;
    const unsigned o0 = (unsigned)(ri * n + gi) << 4;
    const V3 rv = texel_lds(tex, o0 + n16);
    const V3 gv = texel_lds(tex, o0 + 16u);
    const V3 sv = texel_lds(tex, o0 + (lower ? 0u : n16 + 16u));
    const float wr = lower ? rf : 1.0f - gf;
    const float wg = lower ? gf : 1.0f - rf;
    const float ws = lower ? 1.0f - fsum : fsum - 1.0f;
    const float Se = dark ? 0.0f : S;  // S < 1e-12 -> 0 (the texels read with S = 1 are finite)
    V3 o;
    o.xy = __builtin_elementwise_fma(sv.xy, splat(ws), __builtin_elementwise_fma(gv.xy, splat(wg), rv.xy * splat(wr))) * splat(Se);
    o.z = fmaf(sv.z, ws, fmaf(gv.z, wg, rv.z * wr)) * Se;
    return o;
}

// S8, apply_lut3d_tetra of r2f_device.h for non-negative inputs: lo = min(int(t), n - 2), hi = lo + 1.
__device__ __forceinline__ void axis_nonneg(float x, float s, int n, int& lo, float& d) {
    const float t = x * s;
    const int i0 = (int)t;
    lo = min(i0, n - 2);
    d = i0 >= n - 1 ? 1.0f : fmaf(x, s, -(float)i0);  // fraction of the unrounded product, see lut3d_axis
}

__device__ __forceinline__ V3 lut3d_tetra_nonneg(const float4* tex, const int n, const float s, const float r, const float g,
                                                 const float b) {
    int rl, gl, bl;
    float dr, dg, db;
    axis_nonneg(r, s, n, rl, dr);
    axis_nonneg(g, s, n, gl, dg);
    axis_nonneg(b, s, n, bl, db);
    const unsigned er = (unsigned)(n * n) << 4, eg = (unsigned)n << 4, eb = 16u;  // wave-uniform corner strides in bytes
    const unsigned o0 = (unsigned)((rl * n + gl) * n + bl) << 4;
    // utils.py:298-376, ties included: the axis of the largest fraction first
    const bool c1 = dr >= dg, c2 = dg >= db, c3 = dr >= db, c4 = db >= dg, c5 = db >= dr;
    const unsigned e1 = c1 ? ((c2 || c3) ? er : eb) : (c4 ? eb : eg);
    const unsigned e12 = c1 ? (c2 ? er + eg : er + eb) : ((c4 || c5) ? eg + eb : eg + er);
    const float d1 = fmaxf(fmaxf(dr, dg), db), d3 = fminf(fminf(dr, dg), db), d2 = __builtin_amdgcn_fmed3f(dr, dg, db);
    const V3 c0 = texel(tex, o0);
    const V3 ca = texel(tex, o0 + e1);
    const V3 cb = texel(tex, o0 + e12);
    const V3 cz = texel(tex, o0 + (er + eg + eb));
    V3 o;
    o.xy = __builtin_elementwise_fma(splat(d3), cz.xy - cb.xy,
                                     __builtin_elementwise_fma(splat(d2), cb.xy - ca.xy, __builtin_elementwise_fma(splat(d1), ca.xy - c0.xy, c0.xy)));
    o.z = fmaf(d3, cz.z - cb.z, fmaf(d2, cb.z - ca.z, fmaf(d1, ca.z - c0.z, c0.z)));
    return o;
}

// FIN (with UPTO = EXPOSURE): the channels of a.finish_mask are finished here -- tap weight, log, density curve -- and written
// to a.finish_dst instead of the exposure planes (r2f_stage_front_split); the curve's cells then sit in LDS like for DENSITY.
template <int LAYOUT, int UPTO, int BY, bool FIN = false>
__global__ __launch_bounds__(64 * BY) void front_fast_kernel(const FrontArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    float4* lut_lds = smem4;                                // n x n texels of the 2-D LUT
    float4* cells_lds = smem4 + a.lut2d.n * a.lut2d.n;      // 3 x (m - 1) curve cells (UPTO >= DENSITY, or FIN)
    {
        const int tid = threadIdx.y * 64 + threadIdx.x;
        const int nt = a.lut2d.n * a.lut2d.n;
        for (int i = tid; i < nt; i += 64 * BY) lut_lds[i] = a.lut2d.tex[i];
        if (UPTO >= R2F_UPTO_DENSITY || FIN) {
            const int nc = 3 * (a.curve.m - 1);
            for (int i = tid; i < nc; i += 64 * BY) cells_lds[i] = a.curve.cells[i];
        }
        __syncthreads();
    }
#ifndef R2F_FRONT_UNITS
// 1: the fused LUT-only pass (UPTO = OUTPUT: BASELINE config 2) runs on a ONE-dimensional persistent grid: the frame's (row group,
// tile column) units in row-major order -- neighbouring workgroups read neighbouring pieces of the same image rows --, dealt out
// in equal contiguous runs to exactly as many workgroups as fit the chip (256 at one per CU).  0 (rounds 2-5), and what the other
// instances keep: a (columns, rows) grid whose row count is the chip's workgroups divided by the columns, ROUNDED DOWN -- 24 columns
// x 10 = 240 workgroups on 256 CUs for the 24 MP frame: 6 % of the chip idle.  Interleaved A/B of two builds (tools/ab_libs.py):
// 0.240 against 0.252 ms on the noise frame, 0.213 against 0.224 on the photograph-like one (-4.8 %).  Tried for the exposure
// instances too: the recording ones need a wave to stay inside one tile for a while, i.e. column- or band-major units -- +0.03 /
// +0.08 ms on the 100 MP frame (a workgroup that walks down a column reads 3-KB pieces a row pitch apart) --, the non-recording
// split kernel measured +0.02 ms in row-major order: both keep the 2-D grid.
#define R2F_FRONT_UNITS 1
#endif
    // (UPTO = EXPOSURE with a.track: the range of the exposure samples written for the halation's FFT passes goes into the record's tiles)
    const bool track = UPTO == R2F_UPTO_EXPOSURE && a.track.blk != nullptr;
    const int W = a.W;
    const float* in = static_cast<const float*>(a.in);
    const float s3 = a.lut3d_scale * (float)(a.lut3d.n - 1);
    const long long plane = (long long)a.in_rows * W;
    const Mat3Pairs mat = pairs_of(a.mat);
    // The row loop exists once per tracking mode (TRK: 0 none, 1 red + green -- a colour stock, one v_min3 + one v_max3 per pixel --,
    // 2 any mask): a wave-uniform test of a.track inside the pixel loop is a scalar branch per pixel, i.e. four basic-block borders in
    // what the scheduler otherwise treats as one block with the LUT gathers of four pixels in flight together (+21 us per 100 MP
    // frame with the test inside, rocprofv3).  Only the EXPOSURE instances can track; the others compile the loop once.
    // Tracking also fills the record's TILES (r2f_device.h RangeRecord: 64 rows x 256 columns of the global frame; a wave's 64 lanes x
    // 4 pixels are one tile column): a lane gathers its extremes while the wave stays inside one tile row and the wave merges them
    // into the tile when it leaves (one DPP reduction and at most two atomics per wave and tile) -- for which a tracking workgroup
    // row walks a CONTIGUOUS chunk of the row groups (a tile row = 64 / BY consecutive iterations) instead of every gridDim.y-th one.
    auto row_loop = [&](auto trk_tag) {
    constexpr int TRK = decltype(trk_tag)::value;
    const int groups = (a.y1 - a.y0 + BY - 1) / BY;
    constexpr bool UNITS = R2F_FRONT_UNITS && UPTO == R2F_UPTO_OUTPUT;
    const int units = groups * a.gx;
    const int g_begin = UNITS ? (int)((long long)blockIdx.x * units / gridDim.x)
                              : (TRK ? (int)((long long)blockIdx.y * groups / gridDim.y) : (int)blockIdx.y);
    const int g_end = UNITS ? (int)((long long)(blockIdx.x + 1) * units / gridDim.x)
                            : (TRK ? (int)((long long)(blockIdx.y + 1) * groups / gridDim.y) : groups);
    const int g_step = (UNITS || TRK) ? 1 : (int)gridDim.y;
    float p_lo = __builtin_inff(), p_hi = 0.f;  // this lane's extremes inside the tile (p_ty, p_tx)
    int p_ty = -1, p_tx = 0;
    auto flush = [&]() {
        if (p_ty < 0) return;
        const float w_lo = wave_extreme<false>(p_lo), w_hi = wave_extreme<true>(p_hi);
        if (threadIdx.x == 63) merge_tile(a.track, p_ty << kRangeTileRowsLog2, p_tx, w_lo, w_hi);
        p_lo = __builtin_inff(), p_hi = 0.f;
    };
    for (int unit = g_begin; unit < g_end; unit += g_step) {
        const int grp = UNITS ? unit / a.gx : unit;
        const int col = UNITS ? unit - grp * a.gx : (int)blockIdx.x;
        const int x = (col * 64 + threadIdx.x) * 4;
        const int gy = a.y0 + grp * BY + threadIdx.y;
        if (gy >= a.y1) continue;  // (the last row group of a column may be short)
        if (TRK) {  // (gy and col are wave-uniform: a wave is one row of one tile column)
            const int ty = gy >> kRangeTileRowsLog2;
            if (ty != p_ty || col != p_tx) {
                flush();
                p_ty = ty, p_tx = col;
            }
        }
        if (x >= W) continue;
        const long long irow = gy - a.in_gy0;
        float r[4], g[4], b[4];
        if (LAYOUT == R2F_LAYOUT_CHW) {
            const float4 v0 = *reinterpret_cast<const float4*>(in + irow * W + x);
            const float4 v1 = *reinterpret_cast<const float4*>(in + plane + irow * W + x);
            const float4 v2 = *reinterpret_cast<const float4*>(in + 2 * plane + irow * W + x);
            r[0] = v0.x, r[1] = v0.y, r[2] = v0.z, r[3] = v0.w;
            g[0] = v1.x, g[1] = v1.y, g[2] = v1.z, g[3] = v1.w;
            b[0] = v2.x, b[1] = v2.y, b[2] = v2.z, b[3] = v2.w;
        } else if (LAYOUT == R2F_LAYOUT_HWC3) {
            const float4* p4 = reinterpret_cast<const float4*>(in + (irow * W + x) * 3);
            const float4 v0 = p4[0], v1 = p4[1], v2 = p4[2];
            r[0] = v0.x, g[0] = v0.y, b[0] = v0.z;
            r[1] = v0.w, g[1] = v1.x, b[1] = v1.y;
            r[2] = v1.z, g[2] = v1.w, b[2] = v2.x;
            r[3] = v2.y, g[3] = v2.z, b[3] = v2.w;
        } else {
            const float4* p4 = reinterpret_cast<const float4*>(in + (irow * W + x) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = p4[q];
                r[q] = v.x, g[q] = v.y, b[q] = v.z;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            V3 p = a.use_matrix ? matrix3(mat, r[q], g[q], b[q]) : V3{(float2v){r[q], g[q]}, b[q]};
            p = lut2d(lut_lds, a.lut2d.n, p);
            if (UPTO == R2F_UPTO_EXPOSURE) {
                r[q] = p.xy.x, g[q] = p.xy.y, b[q] = p.z;
                // (fminf / fmaxf drop a NaN, and pass 1 of the FFT form takes a non-finite sample as 0: the outputs around it fall below
                // the samples that are left, so for the minimum a NaN counts as "below every floor"; an infinity stays in the maximum)
                if (TRK == 1) {
                    p_lo = __builtin_isunordered(r[q], g[q]) ? -__builtin_inff() : fminf(fminf(p_lo, r[q]), g[q]);
                    p_hi = fmaxf(fmaxf(p_hi, fabsf(r[q])), fabsf(g[q]));
                } else if (TRK == 2) {
                    if (a.track_mask & 1) p_lo = range_min(p_lo, r[q]), p_hi = fmaxf(p_hi, fabsf(r[q]));
                    if (a.track_mask & 2) p_lo = range_min(p_lo, g[q]), p_hi = fmaxf(p_hi, fabsf(g[q]));
                    if (a.track_mask & 4) p_lo = range_min(p_lo, b[q]), p_hi = fmaxf(p_hi, fabsf(b[q]));
                }
                if (FIN) {  // same arithmetic as single_tap_kernel with the halation epilogue: w * x, log10, curve
                    if (a.finish_mask & 1) r[q] = curve_eval_at((const float4*)cells_lds, a.curve, 0, log10_fast(a.finish_w[0] * r[q], a.log_eps));
                    if (a.finish_mask & 2) g[q] = curve_eval_at((const float4*)cells_lds, a.curve, 1, log10_fast(a.finish_w[1] * g[q], a.log_eps));
                    if (a.finish_mask & 4) b[q] = curve_eval_at((const float4*)cells_lds, a.curve, 2, log10_fast(a.finish_w[2] * b[q], a.log_eps));
                }
                continue;
            }
            float v3[3] = {log10_fast(p.xy.x, a.log_eps), log10_fast(p.xy.y, a.log_eps), log10_fast(p.z, a.log_eps)};
            curve_eval_batch<3, 3>((const float4*)cells_lds, a.curve, 0, v3);
            if (UPTO == R2F_UPTO_DENSITY) {
                r[q] = v3[0], g[q] = v3[1], b[q] = v3[2];
                continue;
            }
            if (R2F_FRONT_LUT_BATCH > 1) {  // S8 below, for R2F_FRONT_LUT_BATCH pixels at a time
                r[q] = v3[0], g[q] = v3[1], b[q] = v3[2];
                continue;
            }
            if (__builtin_amdgcn_ballot_w64(fminf(fminf(v3[0], v3[1]), v3[2]) < 0.0f) == 0) {
                p = lut3d_tetra_nonneg(a.lut3d.tex, a.lut3d.n, s3, v3[0], v3[1], v3[2]);
                r[q] = p.xy.x, g[q] = p.xy.y, b[q] = p.z;
            } else {  // a curve that dips below zero: the general cell arithmetic (negative indices wrap like Python's)
                apply_lut3d_tetra(a.lut3d, s3, v3[0], v3[1], v3[2]);
                r[q] = v3[0], g[q] = v3[1], b[q] = v3[2];
            }
        }
        if (UPTO == R2F_UPTO_OUTPUT && R2F_FRONT_LUT_BATCH > 1) {
            constexpr int NB = R2F_FRONT_LUT_BATCH > 1 ? R2F_FRONT_LUT_BATCH : 2;
#pragma unroll
            for (int q0 = 0; q0 < 4; q0 += NB) {
                float rn[NB], gn[NB], bn[NB], lo = 0.f;
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    rn[k] = r[q0 + k], gn[k] = g[q0 + k], bn[k] = b[q0 + k];
                    lo = fminf(lo, fminf(fminf(rn[k], gn[k]), bn[k]));
                }
                if (__builtin_amdgcn_ballot_w64(lo < 0.0f) == 0) {  // all gathers of the batch in flight together
                    apply_lut3d_tetra_nonneg_batch<NB>(a.lut3d, s3, rn, gn, bn);
                } else {  // a curve that dips below zero: the general cell arithmetic (negative indices wrap like Python's)
#pragma unroll
                    for (int k = 0; k < NB; ++k) apply_lut3d_tetra(a.lut3d, s3, rn[k], gn[k], bn[k]);
                }
#pragma unroll
                for (int k = 0; k < NB; ++k) r[q0 + k] = rn[k], g[q0 + k] = gn[k], b[q0 + k] = bn[k];
            }
        }
        if (UPTO != R2F_UPTO_OUTPUT) {
            float* d0 = a.dst.data + (long long)(gy - a.dst.gy0) * W + x;
            float* f0 = FIN ? a.finish_dst.data + (long long)(gy - a.finish_dst.gy0) * W + x : d0;
            const long long fs = FIN ? a.finish_dst.plane_stride : 0;
            *reinterpret_cast<float4*>((FIN && (a.finish_mask & 1)) ? f0 : d0) = make_float4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<float4*>((FIN && (a.finish_mask & 2)) ? f0 + fs : d0 + a.dst.plane_stride) = make_float4(g[0], g[1], g[2], g[3]);
            *reinterpret_cast<float4*>((FIN && (a.finish_mask & 4)) ? f0 + 2 * fs : d0 + 2 * a.dst.plane_stride) =
                make_float4(b[0], b[1], b[2], b[3]);
            continue;
        }
        const long long obase = ((long long)(gy - a.out_gy0) * W + x) * 3;
        if (a.out_f32) {
            float4* o4 = reinterpret_cast<float4*>(a.out_f32 + obase);
            o4[0] = make_float4(r[0], g[0], b[0], r[1]);
            o4[1] = make_float4(g[1], b[1], r[2], g[2]);
            o4[2] = make_float4(b[2], r[3], g[3], b[3]);
        }
        if (a.out_u8) {
            uint32_t* o32 = reinterpret_cast<uint32_t*>(a.out_u8 + obase);
            o32[0] = u8_of(r[0]) | (u8_of(g[0]) << 8) | (u8_of(b[0]) << 16) | (u8_of(r[1]) << 24);
            o32[1] = u8_of(g[1]) | (u8_of(b[1]) << 8) | (u8_of(r[2]) << 16) | (u8_of(g[2]) << 24);
            o32[2] = u8_of(b[2]) | (u8_of(r[3]) << 8) | (u8_of(g[3]) << 16) | (u8_of(b[3]) << 24);
        }
    }
    if (TRK) {
        // The LAST flush of every wave comes at the same moment (the end of the kernel), and in a short call -- the 59-row bands a row
        // shard makes first -- it is every wave's only one: 59 waves x 2 atomics on one tile's line cost such a call 12 us of its 11.
        // The workgroup's waves (BY consecutive rows of the same 256 columns: one tile, two at a tile-row border) combine theirs
        // through LDS first: wave 0 sends one merge per tile row present.
        __shared__ int wg_ty[BY], wg_tx[BY];
        __shared__ float wg_lo[BY], wg_hi[BY];
        const float w_lo = wave_extreme<false>(p_lo), w_hi = wave_extreme<true>(p_hi);
        if (threadIdx.x == 63) wg_ty[threadIdx.y] = p_ty, wg_tx[threadIdx.y] = p_tx, wg_lo[threadIdx.y] = w_lo, wg_hi[threadIdx.y] = w_hi;
        __syncthreads();
        if (threadIdx.y == 0 && threadIdx.x == 0) {
            for (int i = 0; i < BY; ++i) {
                const int ty = wg_ty[i], tx = wg_tx[i];
                if (ty < 0) continue;
                float lo = wg_lo[i], hi = wg_hi[i];
                for (int j = i + 1; j < BY; ++j)
                    if (wg_ty[j] == ty && wg_tx[j] == tx) lo = fminf(lo, wg_lo[j]), hi = fmaxf(hi, wg_hi[j]), wg_ty[j] = -1;
                merge_tile(a.track, ty << kRangeTileRowsLog2, tx, lo, hi);
            }
        }
    }
    };
    if (!track)
        row_loop(std::integral_constant<int, 0>{});
    else if (a.track_mask == 3)
        row_loop(std::integral_constant<int, 1>{});
    else
        row_loop(std::integral_constant<int, 2>{});
    // (no frame-level merge here: thousands of waves looking at ONE address at the end of the kernel serialise in its L2 channel --
    // 23 us on a 22-us front call of a 1/8 row shard -- and nothing decides on the frame's extremes any more; the measurement
    // harness reduces the tiles on the host, r2f_frame_exposure_range)
}

size_t fast_lds_bytes(const FrontArgs& a) {
    const size_t lut = (size_t)a.lut2d.n * a.lut2d.n * sizeof(float4);
    const size_t cells = (a.upto >= R2F_UPTO_DENSITY || a.finish_mask) ? (size_t)3 * (a.curve.m > 1 ? a.curve.m - 1 : 0) * sizeof(float4) : 0;
    return lut + cells;
}

}  // namespace

// Is the pointwise pass of `a` in the fast kernels' domain?
bool front_fast_eligible(const FrontArgs& a) {
    if (!a.vec || a.lut2d.n < 2 || (long long)a.W * 16 >= (1ll << 31) || fast_lds_bytes(a) > 144 * 1024) return false;
    if ((a.upto >= R2F_UPTO_DENSITY || a.finish_mask) && !(a.curve.near && a.curve.m >= 2)) return false;
    if (a.upto == R2F_UPTO_OUTPUT && !(a.lut3d_mode == 0 && a.lut3d.n >= 2 && a.lut3d.n <= 256)) return false;
    return true;
}

template <int UPTO, int BY, bool FIN = false>
static void launch_fast(const FrontArgs& a, hipStream_t s) {
    const int quads = (a.W + 3) / 4, gx = (quads + 63) / 64;
    const int row_groups = (a.y1 - a.y0 + BY - 1) / BY;
    const size_t lds = fast_lds_bytes(a);
    // persistent grid: as many workgroups as fit the chip at once (LDS-limited), each walks the frame in row strides
    // (rounded DOWN to whole rows of workgroups: one workgroup too many would wait for a free CU and run a second round alone)
    const int per_cu = lds > 80 * 1024 ? 1 : (BY == 16 ? 2 : (lds > 53 * 1024 ? 2 : 3));
    int gy = per_cu * 256 / gx;
    gy = gy > row_groups ? row_groups : (gy < 1 ? 1 : gy);
    dim3 grid(gx, gy);
    const dim3 block(64, BY);
    FrontArgs b = a;
    b.gx = gx;
    if (R2F_FRONT_UNITS && UPTO == R2F_UPTO_OUTPUT) {
        // the fused LUT-only pass: exactly as many workgroups as fit the chip at once, each with an equal contiguous run of the (row
        // group, tile column) units (front_fast_kernel, R2F_FRONT_UNITS)
        const long long units = (long long)gx * row_groups;
        grid = dim3((unsigned)std::min<long long>((long long)per_cu * 256, units));
    }
    switch (a.in_layout) {
        case R2F_LAYOUT_CHW: launch_k((front_fast_kernel<R2F_LAYOUT_CHW, UPTO, BY, FIN>), grid, block, lds, s, b); break;
        case R2F_LAYOUT_HWC3: launch_k((front_fast_kernel<R2F_LAYOUT_HWC3, UPTO, BY, FIN>), grid, block, lds, s, b); break;
        default: launch_k((front_fast_kernel<R2F_LAYOUT_HWC4, UPTO, BY, FIN>), grid, block, lds, s, b); break;
    }
}

hipError_t launch_front_fast(const FrontArgs& a, hipStream_t s) {
    // more than 80 KB of tables: one workgroup per CU, so give it 16 waves; else two (or three) of 8
    const bool big = fast_lds_bytes(a) > 80 * 1024;
    if (a.upto == R2F_UPTO_EXPOSURE && a.finish_mask)
        big ? launch_fast<R2F_UPTO_EXPOSURE, 16, true>(a, s) : launch_fast<R2F_UPTO_EXPOSURE, 8, true>(a, s);
    else if (a.upto == R2F_UPTO_EXPOSURE)
        big ? launch_fast<R2F_UPTO_EXPOSURE, 16>(a, s) : launch_fast<R2F_UPTO_EXPOSURE, 8>(a, s);
    else if (a.upto == R2F_UPTO_DENSITY)
        big ? launch_fast<R2F_UPTO_DENSITY, 16>(a, s) : launch_fast<R2F_UPTO_DENSITY, 8>(a, s);
    else
        big ? launch_fast<R2F_UPTO_OUTPUT, 16>(a, s) : launch_fast<R2F_UPTO_OUTPUT, 8>(a, s);
    return take_launch_status();
}

hipError_t front_fast_init_attributes() {
#define R2F_FAST_ATTR(L, U, B)                                                                                             \
    {                                                                                                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(front_fast_kernel<L, U, B>),                     \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);                       \
        if (e != hipSuccess) return e;                                                                                    \
        if (U == R2F_UPTO_EXPOSURE) {                                                                                     \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(front_fast_kernel<L, U, B, true>),                      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);                              \
            if (e != hipSuccess) return e;                                                                                \
        }                                                                                                                 \
    }
#define R2F_FAST_ATTR3(U, B) R2F_FAST_ATTR(R2F_LAYOUT_HWC3, U, B) R2F_FAST_ATTR(R2F_LAYOUT_HWC4, U, B) R2F_FAST_ATTR(R2F_LAYOUT_CHW, U, B)
    R2F_FAST_ATTR3(R2F_UPTO_EXPOSURE, 8)
    R2F_FAST_ATTR3(R2F_UPTO_EXPOSURE, 16)
    R2F_FAST_ATTR3(R2F_UPTO_DENSITY, 8)
    R2F_FAST_ATTR3(R2F_UPTO_DENSITY, 16)
    R2F_FAST_ATTR3(R2F_UPTO_OUTPUT, 8)
    R2F_FAST_ATTR3(R2F_UPTO_OUTPUT, 16)
#undef R2F_FAST_ATTR3
#undef R2F_FAST_ATTR
    return hipSuccess;
}

}  // namespace r2f
