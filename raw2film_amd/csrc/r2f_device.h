// r2f_device.h -- device-side tables and per-pixel stage functions (gfx950 only).
//
// Stage numbering and the reference lines each function follows are in DESIGN.md and
// include/r2f.h.  Everything here is fp32 VALU + integer work; there is no dense
// contraction on this path, so no MFMA.
#pragma once

#include <type_traits>

#include <hip/hip_runtime.h>
#include <stdint.h>

// Wave-uniform read-only tables are addressed through the constant address space so that
// hipcc emits scalar (s_load) instructions and the values sit in SGPRs.
#define R2F_CONSTANT __attribute__((address_space(4)))

namespace r2f {

// ---------------------------------------------------------------------------- tables
// 1-D curve (S4 density curve, S6c grain LUT): one float4 {xp[i], xp[i+1], fp[i], slope[i]} per cell
// and channel, so an evaluation is ONE 16-byte gather (plus a rare neighbour step, see curve_eval).
struct DevCurve {
    const float4* cells;  // [3][m-1]
    int m;
    float x0;        // xp[0]
    float x1;        // xp[m-1]
    float inv_step;  // (m-1)/(xp[m-1]-xp[0]): first guess of the cell, corrected against the cell's own bounds
    float f_first[3], f_last[3];  // fp[ch][0], fp[ch][m-1]: the clamped ends of np.interp
    int near;        // 1: the first guess is never more than one cell off (checked on the host for every breakpoint;
                     // true for any near-uniform axis) -> one corrective gather instead of a walk
};

struct DevLut2D {  // S1: n*n float4 texels (rgb + pad), texel (xi, yi) at xi*n + yi
    const float4* tex;
    int n;
};

struct DevLut3D {  // S8: n^3 float4 texels, (r, g, b) at (r*n + g)*n + b
    const float4* tex;
    int n;
};

// One channel of a stencil, cropped to the bounding box of its non-zero taps and flattened
// into the list of "entries" the inner loop consumes (see stencil_accumulate).  One entry =
// one input-row step m x one 4-tap chunk c; the entries of a row step are consecutive.  Row
// steps are grouped into PHASES: a phase keeps only the input rows its row steps touch in LDS,
// so the LDS footprint (and with it the number of resident workgroups) is a tuning knob that
// does not depend on the stencil height.  The list ends with two dummy entries (and the row-step
// table with two dummy records) so the loop can always prefetch ahead.
struct DevStencil {
    const float* wstream;  // per entry: 4*Q floats  w[t][q] = K[m-q][4c+t]  (0 outside the taps)
    const int* rowinfo;    // per non-empty row step 4 ints: {number of entries (>= 1),
                           //   LDS float offset of its first entry (m - m0(phase))*RS + 4*c_lo,
                           //   sym only: LDS offset of that entry's mirrored 8-float block (m - m0)*RS + 2r - 4*c_lo - 4,
                           //   live tap columns of the first entry (bits 0-3) and of the last entry (bits 4-7)};
                           // the following entries of the row step sit 4 floats further right (mirrored: further left)
    const int* phases;     // per phase 4 ints: {m0, lds_rows, first row step (index into rowinfo), first entry}
                           // + one terminator {., ., n_rowsteps, n_entries}
    int n_phases;
    int n_rowsteps;        // total non-empty row steps
    int n_entries;         // total entries (without the two dummies)
    int mask_first_or, mask_last_or;  // unions over the row steps of the live masks of their first / last entry
    int kh, kw;            // cropped taps
    int kw_pad;            // kw rounded up to a multiple of 4
    int ay, ax;            // anchor inside the cropped box
    int RS;                // LDS row stride (floats) the offsets were built for
    int max_lds_rows;      // largest lds_rows over the phases
    int sym;               // 1: taps are left-right mirror symmetric -> entries cover columns 0..r only (see below)
    int wmul;              // 1; 0 = profiling aid (every entry reads entry 0's weights -> scalar-cache hits)
};

struct DevPlanes {
    float* data;
    long long plane_stride;
    int gy0;
    int rows;
};

// What changes from one render of a session to the next without changing which kernels run: kept in a small device-resident
// block owned by the context and read through a pointer (wave-uniform scalar loads), so that a captured HIP graph of a
// frame's launches can be replayed with new values.  The reference keeps the same data in a uniform buffer it rewrites per
// render (gpu_processor.py:585-597 `buffer_params_grain`, read at noise.wgsl:1-6).
struct FrameParams {
    uint32_t seed;  // grain seed of this render
    // Range of the exposure samples the halation's FFT passes are about to read, as float bit patterns: min of the samples and max
    // of their magnitudes over the channels that take the FFT form, accumulated by the front kernel of a whole-frame render
    // (atomicMin / atomicMax on the bits: non-negative floats order like integers; a negative minimum only has to stay negative).
    // Reset with every write of the block (+inf, 0).  The passes choose their scratch element from it (FftConvArgs::dyn).
    uint32_t e_min, e_max;
    uint32_t reserved;
};
constexpr uint32_t kFrameMinReset = 0x7f800000u, kFrameMaxReset = 0u;

// ---------------------------------------------------------------------------- streaming accesses
// Non-temporal 16-byte accesses for frame-sized buffers that are written once and read back a stage later (1.2 GB per plane set
// at 100 MP, far beyond the 256 MB Infinity Cache).  On MI355X a float4 copy runs 6.57 TB/s that way against 6.23 with plain loads
// and stores (tools/ubench/copy_rate.hip).  Measured per use (A/B, 100 MP): the output stores of FFT pass 3 gain (halation 2.52 ->
// 2.48 ms, MTF 1.69 -> 1.66); the front kernel's loads and stores LOSE (0.50 -> 0.55 ms; the fused LUT-only pass 0.247 -> 0.266),
// the tail is indifferent -- so only pass 3 uses them.  R2F_NT=0 for A/B runs.
#ifndef R2F_NT
#define R2F_NT 1
#endif
typedef float r2f_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* p) {
#if R2F_NT
    const r2f_f4v v = __builtin_nontemporal_load(reinterpret_cast<const r2f_f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void st4_stream(float* p, const float4 v) {
#if R2F_NT
    __builtin_nontemporal_store((r2f_f4v){v.x, v.y, v.z, v.w}, reinterpret_cast<r2f_f4v*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}

// ---------------------------------------------------------------------------- helpers
// The thread id as a value the compiler has to take as new: what is derived from it is computed where it is used instead of being
// kept (and, at a kernel's VGPR cap, spilled: 20 bytes per lane of the generic stencil kernel in round 2) across the accumulation loops.
__device__ __forceinline__ int fresh_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}


// BORDER_REFLECT_101 (what cv.filter2D uses on the reference's CPU path): ... c b | a b c d | c b ...
__device__ __forceinline__ int reflect101(int i, int n) {
    if (i >= 0 && i < n) return i;
    if (n == 1) return 0;
    if (i < 0 && i > -n) return -i;
    if (i >= n && i < 2 * n - 1) return 2 * n - 2 - i;
    int period = 2 * n - 2;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// np.interp semantics: linear, clamped to fp[0] / fp[m-1] outside [xp[0], xp[m-1]].  The cell index is
// guessed from a uniform grid and then walked until the cell's own bounds contain x, so any
// non-decreasing xp gives exact np.interp results (uniform grids never take a step).
template <class CellPtr>
__device__ __forceinline__ float curve_eval_at(CellPtr cells_base, const DevCurve& cv, int ch, float x) {
    const int last = cv.m - 2;
    CellPtr cells = cells_base + ch * (cv.m - 1);
    int i = clampi((int)((x - cv.x0) * cv.inv_step), 0, last);
    float4 c = cells[i];
    if (cv.near) {
        // branch-free: the corrective gather is issued for every pixel (it re-reads the same cell when the guess was
        // right), so the evaluations of a lane's pixels and channels are independent and their gathers overlap --
        // a data-dependent walk serialises them into one memory round trip after the other
        const int adj = (x < c.x && i > 0) ? -1 : ((x >= c.y && i < last) ? 1 : 0);
        c = cells[i + adj];
    } else {
        while (x < c.x && i > 0) c = cells[--i];
        while (x >= c.y && i < last) c = cells[++i];
    }
    const float v = fmaf(c.w, x - c.x, c.z);
    return !(x > cv.x0) ? cv.f_first[ch] : (x >= cv.x1 ? cv.f_last[ch] : v);
}

__device__ __forceinline__ float curve_eval(const DevCurve& cv, int ch, float x) {
    return curve_eval_at(cv.cells, cv, ch, x);
}

// N evaluations at once, x[k] on channel ch0 + k % NCH (NCH = 1: all on ch0; NCH = 3: r, g, b interleaved).  All first
// gathers are issued together; on a `near` curve the corrective gathers are skipped unless some lane of the wave needs
// one (a pixel sitting exactly on a breakpoint whose guess rounded the other way), so the common case is one memory
// round trip for the whole batch.  Used by the stencil epilogue (4 pixels of one channel); in the pointwise and tail
// kernels a batch of 12 kept 48 more registers live and cost more than it saved (tail 1.85 -> 2.26 ms).
template <int N, int NCH, class CellPtr>
__device__ __forceinline__ void curve_eval_batch(CellPtr cells_base, const DevCurve& cv, int ch0, float (&x)[N]) {
    if (!cv.near) {
#pragma unroll
        for (int k = 0; k < N; ++k) x[k] = curve_eval_at(cells_base, cv, ch0 + k % NCH, x[k]);
        return;
    }
    const int last = cv.m - 2;
    int idx[N];
    float4 c[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        idx[k] = clampi((int)((x[k] - cv.x0) * cv.inv_step), 0, last) + (ch0 + k % NCH) * (cv.m - 1);
        c[k] = cells_base[idx[k]];
    }
    int adj[N];
    bool any = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int i = idx[k] - (ch0 + k % NCH) * (cv.m - 1);
        adj[k] = (x[k] < c[k].x && i > 0) ? -1 : ((x[k] >= c[k].y && i < last) ? 1 : 0);
        any |= adj[k] != 0;
    }
    if (__any(any)) {
#pragma unroll
        for (int k = 0; k < N; ++k) c[k] = cells_base[idx[k] + adj[k]];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float v = fmaf(c[k].w, x[k] - c[k].x, c[k].z);
        const int ch = ch0 + k % NCH;
        x[k] = !(x[k] > cv.x0) ? cv.f_first[ch] : (x[k] >= cv.x1 ? cv.f_last[ch] : v);
    }
}

#ifndef R2F_FFT_EPI2
// Round 6, the exact part of a leaner pass-3 epilogue (r2f_fft.hip): the transforms' 1 / (ny nx) lives in the
// kernel spectrum (one multiply per spectrum element at build time instead of one fp64 multiply per OUTPUT: 32 per lane and line),
// and the cell index is clamped with one v_med3_i32 below.  Bit-identical results.  0 = round 5's form (A/B).
#define R2F_FFT_EPI2 1
#endif
// curve_eval_batch for ONE channel of a `near` curve whose cells sit in LDS (the epilogue of FFT pass 3: 32 evaluations per lane
// next to a transform of ~950 instructions, so every instruction here counts).  Same arithmetic, same results; leaner control: the
// common case tests only "x outside its guessed cell" (2 compares), the neighbour logic runs when some lane of the wave needs it.
template <int N>
__device__ __forceinline__ void curve_eval_near_lds(const float4* cells, const DevCurve& cv, const float f_first, const float f_last,
                                                    float (&x)[N]) {
    const int last = cv.m - 2;
    int idx[N];
    float4 c[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
#if R2F_FFT_EPI2
        // clamp to [0, last] in one instruction (last >= 0; the compiler cannot know and builds clampi from a compare, a select and a min)
        const int raw = (int)((x[k] - cv.x0) * cv.inv_step);
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(idx[k]) : "v"(raw), "v"(last));
#else
        idx[k] = clampi((int)((x[k] - cv.x0) * cv.inv_step), 0, last);
#endif
        c[k] = cells[idx[k]];
    }
    bool off = false;
#pragma unroll
    for (int k = 0; k < N; ++k) off |= (x[k] < c[k].x) | (x[k] >= c[k].y);
    if (__any(off)) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int adj = (x[k] < c[k].x && idx[k] > 0) ? -1 : ((x[k] >= c[k].y && idx[k] < last) ? 1 : 0);
            c[k] = cells[idx[k] + adj];
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float v = fmaf(c[k].w, x[k] - c[k].x, c[k].z);
        x[k] = !(x[k] > cv.x0) ? f_first : (x[k] >= cv.x1 ? f_last : v);
    }
}

// ---------------------------------------------------------------------------- the exposure-range record, per tile
// The frame-level range fields above decide nothing any more (the measurement harness derives the frame's extremes from the tiles,
// r2f_frame_exposure_range; e_max = +inf remains the "a front kernel could not record" mark): the halation's FFT passes choose
// their scratch element PER WINDOW PAIR from the range of the samples that pair's two windows hold -- the error of the 12-byte
// element scales with the energy of the window a pixel shares, not of the frame (r2f_api.hip, dyn_rule) -- and a window's range
// comes from a grid of tiles over the GLOBAL frame: tile (gy / 64, x / 256) holds {min, max |.|} of the exposure samples recorded
// for its pixels (float bit patterns; reset = {+inf, 0}: a tile nobody wrote to says "unknown", and a window that touches one keeps
// complex128).  The front kernel writes it (a wave covers 256 columns of a row: one tile column), r2f_stage_exposure_range adds
// rows that came from elsewhere, fft_decide_kernel reads the <= 6 x 4 tiles a window touches.
constexpr int kRangeTileRows = 64, kRangeTileRowsLog2 = 6, kRangeTileCols = 256, kRangeTileColsLog2 = 8;
struct RangeRecord {
    FrameParams* blk;  // the context's frame block (its e_max doubles as the "could not record" mark); nullptr: nothing is recorded
    int2* tiles;       // tyn x txn tiles {min bits, max bits}; nullptr: none
    int tyn, txn;
};
__device__ __forceinline__ void merge_tile(const RangeRecord& rec, int gy, int tx, float lo, float hi) {
    if (!rec.tiles) return;
    const int ty = gy >> kRangeTileRowsLog2;
    if ((unsigned)ty >= (unsigned)rec.tyn || (unsigned)tx >= (unsigned)rec.txn) return;
    int2* t = rec.tiles + (long long)ty * rec.txn + tx;
    // Two fire-and-forget atomics (device scope, no return value).  No look first: a look is a LOAD the wave would have to wait for, and
    // waiting for it means waiting for every memory operation issued before it -- in the front kernel, which merges every fourth
    // iteration, the stores of the rows just finished: a drain of the memory pipeline per tile row, +50 us per 100 MP frame
    // (rocprofv3).  Contention is low by construction: a tile is merged into by the 8 / 16 waves of ONE workgroup of the front
    // kernel (spread over its iterations; their last merges combined through LDS) or by four workgroups of the range kernel.
    (void)__hip_atomic_fetch_min(&t->x, __float_as_int(lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_max(&t->y, __float_as_int(hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// min / max of a value over the 64 lanes of a wave, by DPP (no LDS): the result is valid in every lane of the LAST row (lanes 48-63);
// callers read lane 63.
// a tile's minimum with one more sample: a NaN counts as "below every floor" (fminf alone would drop it; see the front kernel)
__device__ __forceinline__ float range_min(float lo, float x) { return x != x ? -__builtin_inff() : fminf(lo, x); }

template <bool IS_MAX>
__device__ __forceinline__ float wave_extreme(float v) {
    auto op = [](float a, float b) { return IS_MAX ? fmaxf(a, b) : fminf(a, b); };
    auto dpp = [](float x, auto ctrl_tag, auto row_mask_tag) {
        constexpr int CTRL = decltype(ctrl_tag)::value, RM = decltype(row_mask_tag)::value;
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, RM, 0xF, false));
    };
    using std::integral_constant;
    v = op(v, dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xF>{}));   // quad_perm [1, 0, 3, 2]
    v = op(v, dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xF>{}));   // quad_perm [2, 3, 0, 1]
    v = op(v, dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xF>{}));  // row_half_mirror
    v = op(v, dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xF>{}));  // row_mirror: every lane of a row holds the row's
    v = op(v, dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{}));  // row_bcast15 into rows 1 and 3
    v = op(v, dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{}));  // row_bcast31 into rows 2 and 3
    return v;
}

// S0: out = M . in, ((m0*r + m1*g) + m2*b)
struct Mat3 {
    float m[9];
};
__device__ __forceinline__ void apply_matrix(const Mat3& M, float& r, float& g, float& b) {
    // explicit contraction (the same in every kernel that applies S0): m0 r rounded, then two fused multiply-adds
    float x = fmaf(M.m[2], b, fmaf(M.m[1], g, M.m[0] * r));
    float y = fmaf(M.m[5], b, fmaf(M.m[4], g, M.m[3] * r));
    float z = fmaf(M.m[8], b, fmaf(M.m[7], g, M.m[6] * r));
    r = x;
    g = y;
    b = z;
}

// S1: chromaticity-triangle 2-D LUT, lut_2d.wgsl:18-108.
__device__ __forceinline__ void apply_lut2d(const DevLut2D& L, float& X, float& Y, float& Z) {
// fract() of the ROUNDED coordinate, like the shader's (and the oracle's) float32 r: left to itself hipcc contracts r - floor(r)
// into fma(X, inv_sum, -floor(r)), i.e. the fraction of the unrounded product -- up to an ulp of r (8e-6 of a texel at n = 128)
// away, which a rough table turns into 3e-5 of the value (tests/test_gpu_hostile.py).  The blend below is explicit fmaf.
#pragma clang fp contract(off)
    const float S = (X + Y) + Z;
    if (S < 1e-12f) {
        X = Y = Z = 0.f;
        return;
    }
    const int n = L.n;
    const float inv_sum = (float)(n - 1) / S;
    const float r = X * inv_sum;
    const float g = Y * inv_sum;
    const float fr = floorf(r), fg = floorf(g);
    const int ri = clampi((int)fr, 0, n - 2);
    const int gi = clampi((int)fg, 0, n - 2);
    const float rf = r - fr, gf = g - fg;
    const float fsum = rf + gf;
    const bool lower = fsum <= 1.0f;
    const float4 rv = L.tex[(ri + 1) * n + gi];
    const float4 gv = L.tex[ri * n + gi + 1];
    const float4 sv = L.tex[(lower ? ri : ri + 1) * n + (lower ? gi : gi + 1)];
    const float wr = lower ? rf : 1.0f - gf;
    const float wg = lower ? gf : 1.0f - rf;
    const float ws = lower ? 1.0f - fsum : fsum - 1.0f;
    // explicit contraction, the same in every kernel that samples the LUT
    X = fmaf(sv.x, ws, fmaf(gv.x, wg, rv.x * wr)) * S;
    Y = fmaf(sv.y, ws, fmaf(gv.y, wg, rv.y * wr)) * S;
    Z = fmaf(sv.z, ws, fmaf(gv.z, wg, rv.z * wr)) * S;
}

// S3 + S4: log10(max(x, eps)) then the density curve.
__device__ __forceinline__ float log10_fast(float x, float eps) {
    return __log2f(fmaxf(x, eps)) * 0.30102999566398120f;
}

__device__ __forceinline__ float log_curve(const DevCurve& cv, int ch, float x, float eps) {
    // v_log_f32 (1 ulp) * log10(2): |error| <= ~1e-7 * |log2 x|, the same order as the ocml log10f result's own ulp
    return curve_eval(cv, ch, __log2f(fmaxf(x, eps)) * 0.30102999566398120f);
}

// S8 tetrahedral: utils.py:247-380 (tie rules `>=` kept), fp32.
__device__ __forceinline__ void lut3d_axis(float x, float s, int n, int& lo, int& hi, float& d) {
    const float t = x * s;
    int i0 = (int)t;  // truncation toward zero, like int(r) in the reference
    if (i0 >= n - 1) {
        i0 = n - 2;
        d = 1.0f;
    } else {
        // the fraction of the UNROUNDED product: numba runs utils.py:262-289 with a float32 pixel times a float64 scale, so r and
        // dr are float64 there (oracle/stages.py pins that against the golden vectors); one fused multiply-add is that value
        // rounded once.  (hipcc used to contract t - i0 into this by itself; now it is explicit.)
        d = fmaf(x, s, -(float)i0);
    }
    int i1 = i0 + 1;
    // negative indices wrap like Python/numba indexing; clamp what would be out of bounds there
    if (i0 < 0) i0 = i0 + n < 0 ? 0 : i0 + n;
    if (i1 < 0) i1 = i1 + n < 0 ? 0 : i1 + n;
    lo = i0;
    hi = i1;
}

__device__ __forceinline__ void apply_lut3d_tetra(const DevLut3D& L, float s, float& r, float& g, float& b) {
    const int n = L.n;
    int rl, rh, gl, gh, bl, bh;
    float dr, dg, db;
    lut3d_axis(r, s, n, rl, rh, dr);
    lut3d_axis(g, s, n, gl, gh, dg);
    lut3d_axis(b, s, n, bl, bh, db);
    const int base = (rl * n + gl) * n + bl;
    const int er = (rh - rl) * n * n, eg = (gh - gl) * n, eb = bh - bl;
    int e1, e2;
    float d1, d2, d3;
    if (dr >= dg) {
        if (dg >= db) { e1 = er; e2 = eg; d1 = dr; d2 = dg; d3 = db; }
        else if (dr >= db) { e1 = er; e2 = eb; d1 = dr; d2 = db; d3 = dg; }
        else { e1 = eb; e2 = er; d1 = db; d2 = dr; d3 = dg; }
    } else {
        if (db >= dg) { e1 = eb; e2 = eg; d1 = db; d2 = dg; d3 = dr; }
        else if (db >= dr) { e1 = eg; e2 = eb; d1 = dg; d2 = db; d3 = dr; }
        else { e1 = eg; e2 = er; d1 = dg; d2 = dr; d3 = db; }
    }
    const float4 c0 = L.tex[base];
    const float4 ca = L.tex[base + e1];
    const float4 cb = L.tex[base + e1 + e2];
    const float4 c1 = L.tex[base + er + eg + eb];
    // explicit fused multiply-adds: the same bits from every kernel that samples the LUT (the compiler's own contraction
    // choices differed between the tail kernel and the pointwise one)
    r = fmaf(d3, c1.x - cb.x, fmaf(d2, cb.x - ca.x, fmaf(d1, ca.x - c0.x, c0.x)));
    g = fmaf(d3, c1.y - cb.y, fmaf(d2, cb.y - ca.y, fmaf(d1, ca.y - c0.y, c0.y)));
    b = fmaf(d3, c1.z - cb.z, fmaf(d2, cb.z - ca.z, fmaf(d1, ca.z - c0.z, c0.z)));
}

// apply_lut3d_tetra for inputs known to be >= 0 (after the S6 clip, cpu_processor.py:397) and n <= 512: no index wrap, the
// tetrahedron chosen with selects instead of branches, 32-bit byte offsets from the wave-uniform table base.  The same
// fused multiply-adds on the same texels: bit-identical to the general form.
__device__ __forceinline__ void lut3d_axis_nonneg(float x, float s, int n, int& lo, float& d) {
    const float t = x * s;
    const int i0 = (int)t;
    lo = min(i0, n - 2);
    d = i0 >= n - 1 ? 1.0f : fmaf(x, s, -(float)i0);  // fraction of the unrounded product, see lut3d_axis
}

__device__ __forceinline__ float4 lut3d_texel(const float4* tex, unsigned off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tex) + off);
}

__device__ __forceinline__ void apply_lut3d_tetra_nonneg(const DevLut3D& L, float s, float& r, float& g, float& b) {
    const int n = L.n;
    int rl, gl, bl;
    float dr, dg, db;
    lut3d_axis_nonneg(r, s, n, rl, dr);
    lut3d_axis_nonneg(g, s, n, gl, dg);
    lut3d_axis_nonneg(b, s, n, bl, db);
    const unsigned er = (unsigned)(n * n) << 4, eg = (unsigned)n << 4, eb = 16u;
    const unsigned o0 = (unsigned)((rl * n + gl) * n + bl) << 4;
    // utils.py:298-376, ties included: the axis of the largest fraction first
    const bool c1 = dr >= dg, c2 = dg >= db, c3 = dr >= db, c4 = db >= dg, c5 = db >= dr;
    const unsigned e1 = c1 ? ((c2 || c3) ? er : eb) : (c4 ? eb : eg);
    const unsigned e12 = c1 ? (c2 ? er + eg : er + eb) : ((c4 || c5) ? eg + eb : eg + er);
    const float d1 = fmaxf(fmaxf(dr, dg), db), d3 = fminf(fminf(dr, dg), db), d2 = __builtin_amdgcn_fmed3f(dr, dg, db);
    const float4 c0 = lut3d_texel(L.tex, o0);
    const float4 ca = lut3d_texel(L.tex, o0 + e1);
    const float4 cb = lut3d_texel(L.tex, o0 + e12);
    const float4 cz = lut3d_texel(L.tex, o0 + (er + eg + eb));
    r = fmaf(d3, cz.x - cb.x, fmaf(d2, cb.x - ca.x, fmaf(d1, ca.x - c0.x, c0.x)));
    g = fmaf(d3, cz.y - cb.y, fmaf(d2, cb.y - ca.y, fmaf(d1, ca.y - c0.y, c0.y)));
    b = fmaf(d3, cz.z - cb.z, fmaf(d2, cb.z - ca.z, fmaf(d1, ca.z - c0.z, c0.z)));
}

// N pixels at once: all 4 N gathers are issued before the first is consumed.
template <int N>
__device__ __forceinline__ void apply_lut3d_tetra_nonneg_batch(const DevLut3D& L, float s, float (&r)[N], float (&g)[N], float (&b)[N]) {
    const int n = L.n;
    const unsigned er = (unsigned)(n * n) << 4, eg = (unsigned)n << 4, eb = 16u;
    float4 c0[N], ca[N], cb[N], cz[N];
    float d1[N], d2[N], d3[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        int rl, gl, bl;
        float dr, dg, db;
        lut3d_axis_nonneg(r[k], s, n, rl, dr);
        lut3d_axis_nonneg(g[k], s, n, gl, dg);
        lut3d_axis_nonneg(b[k], s, n, bl, db);
        const unsigned o0 = (unsigned)((rl * n + gl) * n + bl) << 4;
        const bool c1 = dr >= dg, c2 = dg >= db, c3 = dr >= db, c4 = db >= dg, c5 = db >= dr;
        const unsigned e1 = c1 ? ((c2 || c3) ? er : eb) : (c4 ? eb : eg);
        const unsigned e12 = c1 ? (c2 ? er + eg : er + eb) : ((c4 || c5) ? eg + eb : eg + er);
        d1[k] = fmaxf(fmaxf(dr, dg), db), d3[k] = fminf(fminf(dr, dg), db), d2[k] = __builtin_amdgcn_fmed3f(dr, dg, db);
        c0[k] = lut3d_texel(L.tex, o0);
        ca[k] = lut3d_texel(L.tex, o0 + e1);
        cb[k] = lut3d_texel(L.tex, o0 + e12);
        cz[k] = lut3d_texel(L.tex, o0 + (er + eg + eb));
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        r[k] = fmaf(d3[k], cz[k].x - cb[k].x, fmaf(d2[k], cb[k].x - ca[k].x, fmaf(d1[k], ca[k].x - c0[k].x, c0[k].x)));
        g[k] = fmaf(d3[k], cz[k].y - cb[k].y, fmaf(d2[k], cb[k].y - ca[k].y, fmaf(d1[k], ca[k].y - c0[k].y, c0[k].y)));
        b[k] = fmaf(d3[k], cz[k].z - cb[k].z, fmaf(d2[k], cb[k].z - ca[k].z, fmaf(d1[k], ca[k].z - c0[k].z, c0[k].z)));
    }
}

// S8 GPU-variant: trilinear between texel centres, lut_3d.wgsl:27-40 (fp32 LUT).
__device__ __forceinline__ void apply_lut3d_trilinear(const DevLut3D& L, float scale, float& r, float& g, float& b) {
    const int n = L.n;
    float t[3] = {r, g, b};
    int i0[3];
    float f[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float v = fminf(fmaxf(t[a] * scale, 0.f), 1.f) * (float)(n - 1);
        int i = (int)floorf(v);
        if (i > n - 2) i = n - 2;
        i0[a] = i;
        f[a] = v - (float)i;
    }
    float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int a = c >> 2, bq = (c >> 1) & 1, cq = c & 1;
        const float w = (a ? f[0] : 1.f - f[0]) * (bq ? f[1] : 1.f - f[1]) * (cq ? f[2] : 1.f - f[2]);
        const float4 v = L.tex[((i0[0] + a) * n + i0[1] + bq) * n + i0[2] + cq];
        o[0] += w * v.x;
        o[1] += w * v.y;
        o[2] += w * v.z;
    }
    r = o[0];
    g = o[1];
    b = o[2];
}

// S6a: PCG3D hash, noise.wgsl:14-20 (uint32 wrap-around arithmetic).
__device__ __forceinline__ void pcg3d(uint32_t& x, uint32_t& y, uint32_t& z) {
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    x += y * z;
    y += z * x;
    z += x * y;
    x ^= x >> 16;
    y ^= y >> 16;
    z ^= z >> 16;
    x += y * z;
    y += z * x;
    z += x * y;
}

// S6a: Box-Muller on the hashed uniforms, noise.wgsl:30-50 / noise_bw.wgsl:30-44.
__device__ __forceinline__ void gaussian_noise(uint32_t gx, uint32_t gy, uint32_t seed, bool mono, float& nr, float& ng,
                                               float& nb) {
    uint32_t vx = gx, vy = gy, vz = seed;
    pcg3d(vx, vy, vz);
    const float inv = 1.0f / 4294967296.0f;  // 1/f32(0xffffffff): f32(0xffffffff) rounds to 2^32
    const float ux = (float)vx * inv;
    const float uy = (float)vy * inv;
    const float u1 = fmaxf(ux, 1e-7f);
    // Hardware transcendentals: v_log_f32 / v_sqrt_f32 are 1 ulp, v_sin/v_cos take the angle in
    // revolutions, i.e. cos(2*pi*u) is v_cos_f32(u) with no range reduction and an exact 2*pi.  The
    // field differs from the fp32-libm evaluation of the WGSL by < 4e-6 absolute (|n| < 5.7), which is
    // 5e-7 in density after the grain LUT -- the PCG3D hash underneath stays bit-exact.
    const float LN2 = 0.69314718055994531f;
    const float r1 = __builtin_amdgcn_sqrtf(-2.0f * (__log2f(u1) * LN2));
    const float c1 = __builtin_amdgcn_cosf(uy);
    nr = r1 * c1;
    if (mono) {
        ng = nr;
        nb = nr;
        return;
    }
    ng = r1 * __builtin_amdgcn_sinf(uy);
    const float u3 = fmaxf((float)vz * inv, 1e-7f);
    const float s12 = u1 + uy;
    nb = __builtin_amdgcn_sqrtf(-2.0f * (__log2f(u3) * LN2)) * __builtin_amdgcn_cosf(s12 - floorf(s12));
}

// ---------------------------------------------------------------------------- stencil core
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float8v __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int N>
struct WVec;
template <>
struct WVec<8> {
    typedef float8v type;
};
template <>
struct WVec<16> {
    typedef float16v type;
};

// Register-tiled direct correlation out of an LDS tile.  Each lane owns P=4 consecutive
// pixels in x and Q consecutive rows; accumulators are packed in pairs of ROWS so that one
// v_pk_fma_f32 does acc[rows 2j,2j+1][p] += (w[t][2j], w[t][2j+1]) * pixel, with the weight
// pair in SGPRs (scalar-cache load, wave-uniform) and the pixel broadcast from one VGPR.
// Per entry the lane reads 8 consecutive floats from LDS (two ds_read_b128), the wave reads
// 4*Q weights with one s_load, and issues 8*Q v_pk_fma_f32.  Entry e+1's weights and pixels
// (and entry e+2's LDS offset) are requested before entry e's FMAs issue; the inner loop body
// is branch-free so that order survives to the ISA.
//
// Numerics: taps are consumed in row-major order per output pixel (independent of the tile
// or shard the pixel falls in -> bit-identical results for any tiling).  Each kernel row is
// summed into a row partial (its first entry starts the partial with a multiply), and the
// row partials are then added up: two-level summation, so the rounding error grows like
// sqrt(kw) + sqrt(kh) instead of sqrt(kh*kw) ulps.
template <int Q, bool FIRST>
__device__ __forceinline__ void entry_fma(const typename WVec<4 * Q>::type& w, const float4v& a, const float4v& b,
                                          float2v (&part)[Q / 2][4]) {
    const float win[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int j = 0; j < Q / 2; ++j) {
            const float2v wv = {w[t * Q + 2 * j], w[t * Q + 2 * j + 1]};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float2v px = {win[p + t], win[p + t]};
                if (FIRST && t == 0)
                    part[j][p] = wv * px;
                else
                    part[j][p] = __builtin_elementwise_fma(wv, px, part[j][p]);
            }
        }
    }
}

// Scalar-memory loads return out of order, so the only wait hipcc can emit for them is
// lgkmcnt(0) -- which also waits for anything issued since.  Touching the current entry's
// operands here makes that wait land BEFORE the next entry's loads are issued (they then fly
// during this entry's FMAs); sched_barrier keeps the scheduler from moving loads across it.
template <int Q>
__device__ __forceinline__ void wait_operands(const typename WVec<4 * Q>::type& w, const float4v& a, const float4v& b) {
    asm volatile("" ::"s"(w[0]), "s"(w[4 * Q - 1]), "v"(a.x), "v"(b.w));
    __builtin_amdgcn_sched_barrier(0);
}

// (Tried and dropped: touching the scalar-cache line of entry e+4 with a one-dword s_load.  Any
// use of a scalar load forces lgkmcnt(0), which also waits for the touch, so it cannot fly ahead.)
// One pipeline step: wait for the current entry's operands, request the next entry's
// (into the other register set), then issue the current entry's FMAs.  `noff` = LDS offset of the
// next entry: plain scalar arithmetic (+4 floats inside a row step, the next row step's first offset
// at its end) -- a per-entry offset TABLE made the compiler fetch two entries' offsets with one
// s_load_dwordx2 right before their use, i.e. one exposed scalar-load latency every other entry.
// The first step of row step r also requests the record of row step r+2 (`info_ptr` -> `info_nn`).
typedef int __attribute__((ext_vector_type(4))) int4v;

template <int Q, bool FIRST>
__device__ __forceinline__ void entry_step(const float* lds, const int noff,
                                           const typename WVec<4 * Q>::type R2F_CONSTANT* wstream, const int wmul, int& e,
                                           const typename WVec<4 * Q>::type& cw, const float4v& ca, const float4v& cb, typename WVec<4 * Q>::type& nw, float4v& na, float4v& nb,
                                           float2v (&part)[Q / 2][4], const int4v R2F_CONSTANT* info_ptr = nullptr,
                                           int4v* info_nn = nullptr) {
    wait_operands<Q>(cw, ca, cb);
    if (FIRST) *info_nn = *info_ptr;
    nw = wstream[(e + 1) * wmul];
    na = *reinterpret_cast<const float4v*>(lds + noff);
    nb = *reinterpret_cast<const float4v*>(lds + noff + 4);
    __builtin_amdgcn_sched_barrier(0);
    entry_fma<Q, FIRST>(cw, ca, cb, part);
    ++e;
}

// ---- mirror-symmetric stencils -------------------------------------------------------------
// When K[i][j] == K[i][2r - j] bit for bit (halation discs and |ifft2| MTF kernels are), the two
// mirrored taps of a kernel row share their weight:  w * x[+j] + w * x[-j]  ->  w * (x[+j] + x[-j]).
// An entry then covers 4 columns j = 4c..4c+3 of the LEFT half (the centre column j = r carries half
// its weight and is paired with itself, which is exact: (w/2) * (x + x) == w * x).  Per entry the lane
// reads two 8-float blocks, L at columns 4c.. and the mirrored R at columns 2r-4c-4.. (r is made even
// on the host so that R is 16-byte aligned), forms 16 sums s[p][t] = L[p+t] + R[4+p-t], and issues the
// same 8*Q packed FMAs as a plain entry -- for twice the taps.  1.5x fewer VALU instructions per tap.
// One tap column t of a symmetric entry: the 4 mirrored sums, then 2*Q packed FMAs.
// L index p+t and R index 4+p-t have the same parity, so the sums of two neighbouring pixels come out of one
// v_pk_add_f32 on two aligned register pairs whenever p+t is even: (p=0,1), (p=2,3) for even t; (p=1,2) for odd t
// (p=0 and p=3 are then single adds).  10 VALU instructions per entry instead of 16, same 16 sums, same rounding.
template <int Q, int T, bool MUL>
__device__ __forceinline__ void tap_fma_sym(const typename WVec<4 * Q>::type& w, const float (&lw)[8], const float (&rw)[8],
                                            float2v (&part)[Q / 2][4]) {
    float sum[4];
    if ((T & 1) == 0) {
        const float2v s01 = float2v{lw[T], lw[T + 1]} + float2v{rw[4 - T], rw[5 - T]};
        const float2v s23 = float2v{lw[T + 2], lw[T + 3]} + float2v{rw[6 - T], rw[7 - T]};
        sum[0] = s01.x, sum[1] = s01.y, sum[2] = s23.x, sum[3] = s23.y;
    } else {
        const float2v s12 = float2v{lw[T + 1], lw[T + 2]} + float2v{rw[5 - T], rw[6 - T]};
        sum[0] = lw[T] + rw[4 - T];
        sum[1] = s12.x, sum[2] = s12.y;
        sum[3] = lw[T + 3] + rw[7 - T];
    }
#pragma unroll
    for (int j = 0; j < Q / 2; ++j) {
        const float2v wv = {w[T * Q + 2 * j], w[T * Q + 2 * j + 1]};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float2v px = {sum[p], sum[p]};
            part[j][p] = MUL ? wv * px : __builtin_elementwise_fma(wv, px, part[j][p]);
        }
    }
}

// FIRST: the entry starts the row partial (multiply instead of fma).  MASKED: only the tap columns whose bit is set in
// `mask` are evaluated -- the first and last entry of a row step usually hold padding columns or the rim of a disc
// (all-zero weights); `mask` is wave-uniform (it comes from the row-step record), so each test is a scalar branch.
template <int Q, bool FIRST, bool MASKED>
__device__ __forceinline__ void entry_fma_sym(const typename WVec<4 * Q>::type& w, const float4v& la, const float4v& lb,
                                              const float4v& ra, const float4v& rb, float2v (&part)[Q / 2][4], const int mask) {
    const float lw[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w};
    const float rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
    if (!MASKED) {
        tap_fma_sym<Q, 0, FIRST>(w, lw, rw, part);
        tap_fma_sym<Q, 1, false>(w, lw, rw, part);
        tap_fma_sym<Q, 2, false>(w, lw, rw, part);
        tap_fma_sym<Q, 3, false>(w, lw, rw, part);
    } else {
        // (a single dispatch on the whole mask into straight-line prefix / suffix variants measured slower than these
        // four scalar branches: 14.7 vs 14.2 ms on the 100 MP halation)
        if (FIRST) {  // fma(w, x, +0) == w * x: starting from zero gives the same partial as starting with a multiply
#pragma unroll
            for (int j = 0; j < Q / 2; ++j)
#pragma unroll
                for (int p = 0; p < 4; ++p) part[j][p] = float2v{0.f, 0.f};
        }
        if (mask & 1) tap_fma_sym<Q, 0, false>(w, lw, rw, part);
        if (mask & 2) tap_fma_sym<Q, 1, false>(w, lw, rw, part);
        if (mask & 4) tap_fma_sym<Q, 2, false>(w, lw, rw, part);
        if (mask & 8) tap_fma_sym<Q, 3, false>(w, lw, rw, part);
    }
}

template <int Q>
struct SymOperands {
    typename WVec<4 * Q>::type w;
    float4v la, lb, ra, rb;
};

#ifndef R2F_EXP
#define R2F_EXP 0  // development switch (tools/ablate_stencil.py): bit 0 no LDS reads, 1 no weight loads, 2 no FMAs
#endif

template <int Q, bool FIRST, bool MASKED>
__device__ __forceinline__ void entry_step_sym(const float* lds, const int noff, const int noffr,
                                               const typename WVec<4 * Q>::type R2F_CONSTANT* wstream, const int wmul, int& e,
                                               const SymOperands<Q>& cur, SymOperands<Q>& nxt, float2v (&part)[Q / 2][4],
                                               const int mask = 15, const int4v R2F_CONSTANT* info_ptr = nullptr,
                                               int4v* info_nn = nullptr) {
    asm volatile("" ::"s"(cur.w[0]), "s"(cur.w[4 * Q - 1]), "v"(cur.la.x), "v"(cur.lb.w), "v"(cur.ra.x), "v"(cur.rb.w));
    __builtin_amdgcn_sched_barrier(0);
    if (FIRST) *info_nn = *info_ptr;  // see entry_step
    if (R2F_EXP & 2) {
        asm volatile("" : "=s"(nxt.w));  // "defined" without an instruction: garbage weights
    } else {
        nxt.w = wstream[(e + 1) * wmul];
    }
    if (R2F_EXP & 1) {
        asm volatile("" : "=v"(nxt.la), "=v"(nxt.lb), "=v"(nxt.ra), "=v"(nxt.rb));
    } else {
        nxt.la = *reinterpret_cast<const float4v*>(lds + noff);
        nxt.lb = *reinterpret_cast<const float4v*>(lds + noff + 4);
        nxt.ra = *reinterpret_cast<const float4v*>(lds + noffr);
        nxt.rb = *reinterpret_cast<const float4v*>(lds + noffr + 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (R2F_EXP & 4) {
        if (FIRST)
            for (int j = 0; j < Q / 2; ++j)
                for (int p = 0; p < 4; ++p) part[j][p] = float2v{cur.w[j], cur.la.x + cur.ra.y};
        else
            part[0][0] += float2v{cur.w[0] + cur.w[4 * Q - 1], cur.la.x + cur.lb.w + cur.ra.x + cur.rb.w};
    } else
        entry_fma_sym<Q, FIRST, MASKED>(cur.w, cur.la, cur.lb, cur.ra, cur.rb, part, mask);
    ++e;
}

// Row step = first entry (masked), middle entries (all four tap columns), last entry (masked).  The two operand sets
// alternate entry by entry; the current entry sits in set A at the top of every row step.  MASKS = false evaluates every
// tap column of every entry (the tail kernel's 9x9 grain stencil: two entries per row step, the branches cost more than
// the padding columns they skip -- 2.0 vs 1.8 ms; small square grain stencils now take grain_stencil_fixed anyway).
template <int Q, bool MASKS>
__device__ __forceinline__ void stencil_accumulate_sym(const float* lds, const DevStencil& st, int row_begin, int row_end,
                                                       int e0, float2v (&acc)[Q / 2][4]) {
    typedef typename WVec<4 * Q>::type wvec;
    const int4v R2F_CONSTANT* info = (const int4v R2F_CONSTANT*)st.rowinfo;
    const wvec R2F_CONSTANT* wstream = (const wvec R2F_CONSTANT*)st.wstream;
    const int wmul = st.wmul;
    float2v part[Q / 2][4];
    int e = e0;
    int4v ri = info[row_begin], ri_n = info[row_begin + 1], ri_nn;  // this row step, the next, the one after
    SymOperands<Q> A, B;
    A.w = wstream[e0 * wmul];
    A.la = *reinterpret_cast<const float4v*>(lds + ri.y);
    A.lb = *reinterpret_cast<const float4v*>(lds + ri.y + 4);
    A.ra = *reinterpret_cast<const float4v*>(lds + ri.z);
    A.rb = *reinterpret_cast<const float4v*>(lds + ri.z + 4);
    for (int r = row_begin; r < row_end; ++r) {
        const int cnt = ri.x;
        const int m_first = cnt > 1 ? (ri.w & 15) : (ri.w & (ri.w >> 4) & 15), m_last = (ri.w >> 4) & 15;
        int off = ri.y + 4, offr = ri.z - 4;  // LDS offsets of the entry after the current one, if it is in this row step
        {  // entry 0
            const bool only = cnt <= 1;
            const int no = only ? ri_n.y : off, nor = only ? ri_n.z : offr;
            off += 4, offr -= 4;
            entry_step_sym<Q, true, MASKS>(lds, no, nor, wstream, wmul, e, A, B, part, m_first, info + r + 2, &ri_nn);
        }
        int i = 1;
        for (; i + 2 < cnt; i += 2) {  // two middle entries: i (set B) and i + 1 (set A); entry i + 2 exists in this row step
            entry_step_sym<Q, false, false>(lds, off, offr, wstream, wmul, e, B, A, part);
            entry_step_sym<Q, false, false>(lds, off + 4, offr - 4, wstream, wmul, e, A, B, part);
            off += 8, offr -= 8;
        }
        if (i + 1 < cnt) {  // two entries left: a middle one (set B), then the last (set A)
            entry_step_sym<Q, false, false>(lds, off, offr, wstream, wmul, e, B, A, part);
            entry_step_sym<Q, false, MASKS>(lds, ri_n.y, ri_n.z, wstream, wmul, e, A, B, part, m_last);
            A = B;
        } else if (i < cnt) {  // one entry left: the last (set B)
            entry_step_sym<Q, false, MASKS>(lds, ri_n.y, ri_n.z, wstream, wmul, e, B, A, part, m_last);
        } else {  // cnt == 1
            A = B;
        }
#pragma unroll
        for (int j = 0; j < Q / 2; ++j)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[j][p] += part[j][p];
        ri = ri_n;
        ri_n = ri_nn;
    }
}

// ---- small square mirror-symmetric stencils, fully unrolled -------------------------------------------------------
// (2 R + 1)^2 taps, Q = 2 rows per lane (the tail kernel's grain stencil, R <= 6) or 4 (the direct stencil kernel below the
// FFT threshold, R <= 11).  The generic entry machinery above is built for stencils of dozens of row steps: for a 9 x 9 one
// it reads every 8-float block twice (two entries per row step), fetches its weights with two scalar loads per row step
// that the short loop cannot hide, and keeps a row partial.  Here every input row is read once (its 4 + 2 R floats, as
// aligned 16-byte blocks), the weight PAIRS (taps of output rows 2 j and 2 j + 1) come from one wave-uniform table that the
// unrolled code loads ahead, and the mirrored columns are summed before the packed FMAs.  Taps are consumed input row by
// input row (each row a partial, added to the total), outer columns to the centre: the same order for every pixel, so
// results do not depend on the tile or shard a pixel falls in.
// lds: the lane's first input row and pixel column (tile + ty * Q * RS + 4 * tx); the tile's column 0 is the image column
// tile_x0 - AX: AX = R below 9 x 9 (the host builds those boxes unpadded), else 6, 10 or 14 (its padding of r to 2 mod 4).  wp: [(2 R + Q)][R + 1][Q / 2] pairs; pair j of
// input row i = (K[i - 2 j][c], K[i - 2 j - 1][c]), zero outside the kernel.
constexpr int fixed_stencil_ax(int R) { return R < 4 ? R : (R <= 6 ? 6 : (R <= 10 ? 10 : 14)); }

template <int R, int Q>
__device__ __forceinline__ void stencil_fixed(const float* lds, const int RS, const float2v R2F_CONSTANT* wp,
                                              float2v (&acc)[Q / 2][4]) {
    constexpr int AX = fixed_stencil_ax(R), O = AX - R, NB = (O + 2 * R + 4 + 3) / 4;
    float2v part[Q / 2][4];
#pragma unroll
    for (int i = 0; i < 2 * R + Q; ++i) {
        // the row as aligned register pairs xp[k] = (x[2k], x[2k+1]): a mirrored pair of columns has indices of equal parity,
        // so two neighbouring pixels' sums are one v_pk_add_f32, and a packed FMA picks its pixel with op_sel (no copies)
        float2v xp[2 * NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4v v = *reinterpret_cast<const float4v*>(lds + i * RS + 4 * b);
            xp[2 * b] = float2v{v.x, v.y};
            xp[2 * b + 1] = float2v{v.z, v.w};
        }
#pragma unroll
        for (int c = 0; c <= R; ++c) {
            const int a0 = O + c, b0 = O + 2 * R - c;  // columns of pixel 0's left and mirrored tap (the centre: a0 == b0)
            float2v s[4];                              // s[p]: pixel p's sum in both halves
            if ((a0 & 1) == 0) {
                const float2v s01 = c < R ? xp[a0 >> 1] + xp[b0 >> 1] : xp[a0 >> 1];
                const float2v s23 = c < R ? xp[(a0 >> 1) + 1] + xp[(b0 >> 1) + 1] : xp[(a0 >> 1) + 1];
                s[0] = float2v{s01.x, s01.x}, s[1] = float2v{s01.y, s01.y};
                s[2] = float2v{s23.x, s23.x}, s[3] = float2v{s23.y, s23.y};
            } else {
                const float l0 = xp[a0 >> 1].y, r0 = xp[b0 >> 1].y;                    // columns a0, b0 (odd)
                const float l3 = xp[(a0 + 3) >> 1].x, r3 = xp[(b0 + 3) >> 1].x;        // columns a0 + 3, b0 + 3 (even)
                const float2v s12 = c < R ? xp[(a0 + 1) >> 1] + xp[(b0 + 1) >> 1] : xp[(a0 + 1) >> 1];
                const float s0 = c < R ? l0 + r0 : l0, s3 = c < R ? l3 + r3 : l3;
                s[0] = float2v{s0, s0}, s[1] = float2v{s12.x, s12.x};
                s[2] = float2v{s12.y, s12.y}, s[3] = float2v{s3, s3};
            }
#pragma unroll
            for (int j = 0; j < Q / 2; ++j) {
                if (i - 2 * j < 0 || i - 2 * j - 1 > 2 * R) continue;  // both taps of the pair lie outside the kernel
                const float2v wv = wp[(i * (R + 1) + c) * (Q / 2) + j];
#pragma unroll
                for (int p = 0; p < 4; ++p) part[j][p] = c == 0 ? wv * s[p] : __builtin_elementwise_fma(wv, s[p], part[j][p]);
            }
        }
        // one input row = one row partial, added to the total: the rounding error grows with rows + columns, not with their
        // product (a single FMA chain over the 529 taps of a 23 x 23 stencil was 3x further from the oracle)
#pragma unroll
        for (int j = 0; j < Q / 2; ++j) {
            if (i - 2 * j < 0 || i - 2 * j - 1 > 2 * R) continue;
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[j][p] += part[j][p];
        }
    }
}

// Accumulate the row steps [row_begin, row_end) of one phase, whose first entry is e0.
template <int Q>
__device__ __forceinline__ void stencil_accumulate(const float* lds, const DevStencil& st, int row_begin, int row_end,
                                                   int e0, float2v (&acc)[Q / 2][4]) {
    static_assert(Q == 2 || Q == 4, "Q must be 2 or 4");
    typedef typename WVec<4 * Q>::type wvec;
    const int4v R2F_CONSTANT* info = (const int4v R2F_CONSTANT*)st.rowinfo;
    const wvec R2F_CONSTANT* wstream = (const wvec R2F_CONSTANT*)st.wstream;
    const int wmul = st.wmul;
    float2v part[Q / 2][4];

    int e = e0;  // flat entry index
    int4v ri = info[row_begin], ri_n = info[row_begin + 1], ri_nn;
    // two operand sets: A holds the current entry at the top of every row step
    wvec wA = wstream[e0 * wmul], wB;
    float4v aA, bA, aB, bB;
    aA = *reinterpret_cast<const float4v*>(lds + ri.y);
    bA = *reinterpret_cast<const float4v*>(lds + ri.y + 4);
    for (int r = row_begin; r < row_end; ++r) {
        const int cnt = ri.x;
        int off = ri.y + 4;
        // first entry of the row step starts the row partial (multiply instead of fma)
        {
            const int no = cnt <= 1 ? ri_n.y : off;
            off += 4;
            entry_step<Q, true>(lds, no, wstream, wmul, e, wA, aA, bA, wB, aB, bB, part, info + r + 2, &ri_nn);
        }
        int i = 1;
        for (; i + 1 < cnt; i += 2) {
            {
                const int no = off;
                off += 4;
                entry_step<Q, false>(lds, no, wstream, wmul, e, wB, aB, bB, wA, aA, bA, part);
            }
            {
                const int no = i + 2 >= cnt ? ri_n.y : off;
                off += 4;
                entry_step<Q, false>(lds, no, wstream, wmul, e, wA, aA, bA, wB, aB, bB, part);
            }
        }
        if (i < cnt) {
            entry_step<Q, false>(lds, ri_n.y, wstream, wmul, e, wB, aB, bB, wA, aA, bA, part);
        } else {  // an odd number of steps was taken: the current entry sits in set B
            wA = wB;
            aA = aB;
            bA = bB;
        }
#pragma unroll
        for (int j = 0; j < Q / 2; ++j)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[j][p] += part[j][p];
        ri = ri_n;
        ri_n = ri_nn;
    }
}

}  // namespace r2f
