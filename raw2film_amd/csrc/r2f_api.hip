// r2f_api.hip -- C ABI (include/r2f.h) over the gfx950 kernels: context, table upload,
// stencil re-ordering, stage dispatch and the whole-frame render graph.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

#include "../../include/r2f.h"
#include "r2f_launch.h"
#include "r2f_plan.h"

using namespace r2f;

namespace {

struct DeviceBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// Host copy of one stencil as handed to r2f_set_kernel, plus its device form per Q.
struct StencilSet {
    bool present = false;
    int kh = 0, kw = 0, kc = 0;
    std::vector<float> host;  // (kh, kw, kc)
    int built_q = 0;          // 0 = device form stale
    int built_tw = 0, built_th = 0;
    size_t built_budget = 0;
    bool built_sym = false;
    bool common_box = false;
    DevStencil dev[3];
    plan::StencilGeom geom[3];  // the host-side geometry dev[] was filled from
    bool mixed_sign[3] = {false, false, false};  // the channel has taps of both signs (set by r2f_set_kernel)
    bool unit_gain[3] = {false, false, false};   // no negative tap and the taps sum to 1 (>= 0.99): an output is no smaller than the
                                                 // smallest sample under the stencil -- what the 12-byte element's guard presumes
    int single_tap_mask = 0;  // channels whose stencil is ONE tap at the anchor (set by r2f_set_kernel: a scan of every tap, which
                              // the per-frame front / range calls of a row shard must not repeat -- 15 us of host time per call)
    DeviceBuf wbuf[3], mbuf[3];
};

}  // namespace

struct r2f_ctx {
    int device = 0;
    std::string err;
    // Bumped whenever something a captured HIP graph may have frozen changes: a table or stencil upload, a context buffer that
    // was re-allocated (its old address is dangling), the matrix, an option.  r2f_generation() reports it; a caller that replays
    // captured launches (raw2film_amd/sharding.py) re-captures when it moves.
    uint64_t generation = 0;
    bool has_matrix = false;
    Mat3 mat;
    DeviceBuf lut2d_buf, lut3d_buf, curve_buf, grain_lut_buf;
    DevLut2D lut2d{nullptr, 0};
    DevLut3D lut3d{nullptr, 0};
    DevCurve curve{};
    DevCurve grain_lut{};
    StencilSet stencil[3];
    // tile-order tables of the stencil launches (xcd_remap = 2), keyed by the tile grid
    struct TileOrder {
        int gx = 0, gy = 0;
        DeviceBuf buf;
    } tile_order[4];
    int tile_order_next = 0;
    // FFT form of large stencils (r2f_fft.hip): twiddles, per stencil and channel the kernel spectrum, pass scratch
    // (one spectrum per WINDOW SHAPE: the calls of one frame may differ in it -- a row shard's interior halation and its boundary
    // bands cover different numbers of rows, and the window choice follows the rows of the call -- and must not evict each other)
    static constexpr int kFftShapes = 6;  // {256, 512} rows x {256, 512, 1024} columns
    static int fft_shape_index(int ny, int nx) { return (ny == 512 ? 3 : 0) + (nx == 256 ? 0 : (nx == 512 ? 1 : 2)); }
    DeviceBuf fft_tw, fft_kf[3][3][kFftShapes], fft_s1, fft_kimg;
    bool fft_kf_valid[3][3][kFftShapes] = {};
    int fft_kf_dims[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // window shape (ny * 4096 + nx) of the channel's last launch
    int fft_last_real[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // ... and whether that launch multiplied by a real spectrum
    int opt_fft_window = 0;      // window columns: 0 = the cheapest of 256 / 512 / 1024 per stencil and frame, or one of them forced
    int opt_fft_window_max = 512;  // widest window the automatic choice may take (1024 columns: 9 % fewer window elements for the
                                   // 87-tap disc, but the passes run 10-25 % slower per element)
    int opt_fft_window_rows = 0;  // window rows, likewise
    // optional per-launch timing of the FFT passes with events on the launch stream (bench.py's roofline): class 0 / 1 / 2 =
    // pass 1 / 2 / 3; algorithmic bytes are summed alongside
    int opt_timing = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timing_ev[6];  // [pass + 3 * (complex64 scratch ? 1 : 0)]
    double timing_bytes[6] = {0, 0, 0, 0, 0, 0};
    int opt_fft = 1;             // 1: stencil channels with a large enough kernel take the FFT form
    int opt_fft_min_taps = 400;   // ... "large enough": cropped box of at least this many taps (and at most 200 x 200);
                                 // measured crossover with the direct form: 17 x 17 ties, 23 x 23 is 1.5x faster by FFT
    // two internal streams take alternate batches (each with its own half of the scratch), so the tail of one launch
    // overlaps the head of the other stream's; fenced against the caller's stream with events
    hipStream_t fft_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fft_ev_in = nullptr, fft_ev_out[4] = {nullptr, nullptr, nullptr, nullptr};
    int opt_fft_streams = 2;
    int opt_fft_even = 1;        // batches sized so that every internal stream gets the same number of launch triples
    int opt_fft_batch = 192;     // window pairs per launch triple: 192 MB of scratch stay inside the 256 MB Infinity Cache
    // bit `which`: that stencil's FFT scratch images hold complex64 instead of complex128 elements (r2f_fft.hip, sld / sst).
    // Default: the MTF only -- it acts on density, whose values are bounded, so two fp32 roundings of the spectrum cost ~1e-7
    // absolute; the halation acts on linear exposure, where the same roundings are relative to the brightest pixel of the window.
    int opt_fft_s32 = 1 << R2F_KERNEL_MTF;
    int opt_fft_s96 = 0;  // bit `which`: 12-byte scratch elements (doubles rounded to 48 bits, 2^-37) whatever the frame holds -- A/B
    // 1: the halation's FFT passes choose between complex128 and the 12-byte element ON THE DEVICE, per frame and per WINDOW PAIR,
    // from the range of the exposure samples the pair's windows hold (the record's tiles, filled by the front kernel): the 12-byte
    // element costs a shadow at most 1.46e-11 x (max |x| / shadow) of itself (two roundings at 2^-37; kDynCoefficient below adds a
    // factor 1.5) -- which the density curve turns into 0.434 x slope x that; the bound keeps it under three fp32 ulps of a density
    // in [1, 2), what the MTF's complex64 scratch is allowed, and window pairs with a wider range keep complex128 (the stand-in Portra
    // curve: max / shadow <= 6.2e4; the headline's noise frame spans 1.4e5 as a whole, ~2e4 per window: 99 % of its pairs qualify).
    int opt_fft_s96_auto = 1;
    float curve_slope_max = 0.f;  // max |d density / d log10 exposure| over the density curve's cells (host copy, r2f_set_curve1d)
    bool frame_dyn_armed = false;  // the last whole-frame render's halation launches carried the rule (r2f_frame_exposure_range)
    bool capturing = false;        // r2f_render is capturing render_launches: the frame-block write stays outside the graph
    int opt_fft_epi_lds = 1;  // pass 3's epilogue gathers its curve cells from LDS (0: from global memory; A/B)
    // 1: a centrally symmetric tap box (k[i][j] == k[bh-1-i][bw-1-j] bit for bit, anchor at its centre -- every halation disc and
    // |ifft2| MTF kernel the reference builds, effects.py:200-217, :123-143) is laid out with its anchor on the window origin, so its
    // spectrum is real: pass 2 reads 8 instead of 16 bytes of it per element (FftConvArgs::kreal).  0: complex spectra for all (A/B).
    int opt_fft_real = 1;
    int opt_fft_mixed_sign = 1;  // channels with taps of both signs take the float64 FFT form on complex128 scratch whatever their size
    int opt_fft_cols_walk = 1;  // pass 2 of real-spectrum launches as a resident grid walking the launch's pairs (0: one workgroup per pair; A/B)
    int cu_count = 0;           // multiprocessors of the context's device (grid size of that form)
    bool fft_kf_real[3][3][kFftShapes] = {};  // what the cached spectrum of (stencil, channel, shape) holds
    DeviceBuf lanczos_buf;  // [xofs | yofs | xcoef | ycoef] of the last LANCZOS4 geometry
    DeviceBuf lanczos_f32_buf;  // the same for the float32 up-scale before the path
    int lanczos_key[4] = {0, 0, 0, 0};
    int opt_xcd_band = 0;  // tile columns per band of the xcd_remap = 2 order; 0 = auto
    int opt_variant = -1;  // -1 auto
    int opt_xcd_remap = 2;  // 0 = launch order, 1 = one contiguous row-major run of tiles per XCD, 2 = that run walked in column bands
    int opt_ablate = 0;
    int opt_sym = 1;      // use the mirror-symmetric entry form when a channel's taps allow it
    // the grain stencil as weight pairs for grain_stencil_fixed (small square symmetric kernels), built on first use
    DeviceBuf grain_fixed_w;
    bool grain_fixed_valid = false;
    int grain_fixed_r = 0, grain_fixed_same = 0;
    int opt_grain_fixed = 1;  // 0: always the generic entry list (A/B)
    // the grain stencil as two 1-D passes when every channel is u v^T to fp32 rounding (ensure_grain_fixed)
    bool grain_sep = false;
    float grain_sep_u[3][19] = {}, grain_sep_v[3][10] = {};
    int opt_grain_sep = 1;    // 0: never take the separable form (A/B)
    DeviceBuf stencil_fixed_w[3];  // the same for the direct stencil kernel (stencil_fixed<R, 4>), per stencil
    bool stencil_fixed_valid[3] = {false, false, false};
    int opt_stencil_fixed = 1;
    int opt_front_fast = 1;    // the fused LUT-only pass may take the specialised kernel (r2f_front.hip); 0 = always the generic one (A/B)
    int opt_front_blocks = 6;  // front kernel with the curve in LDS: workgroups per CU in its grid (3 are resident at 48 KB each)
    int opt_lds_kb = 80;  // LDS budget per stencil workgroup; 80 KB -> two workgroups per CU
    // Per-render values the kernels read through a pointer (FrameParams: the grain seed), so that a captured frame can be
    // replayed with a new seed; written in stream order by write_frame_params ahead of a frame's launches.
    DeviceBuf frame_buf;
    // The exposure-range record's tile grid (r2f_device.h RangeRecord: 64 x 256 tiles of the GLOBAL frame, {min, max |.|} each) and
    // the per-pair flags fft_decide_kernel derives from it for the halation's FFT passes of the call at hand (one per pair-in-channel)
    DeviceBuf range_tiles, dyn_flags;
    int tiles_tyn = 0, tiles_txn = 0;
    int dyn_flags_ppc = 0;  // pairs per channel of the last call that chose per pair (r2f_frame_scratch_choice)
    // r2f_render's graph cache: one entry per (buffers, shape, parameters without the seed).  An entry is rendered kernel by
    // kernel the first time the context sees its structure (tables, scratch and spectra get built then), captured on
    // `cap_stream` afterwards and replayed on the caller's stream from then on.  Everything is dropped when `generation` moves.
    struct RenderGraph {
        const void* in = nullptr;
        int in_layout = 0;
        float* out_f32 = nullptr;
        uint8_t* out_u8 = nullptr;
        int H = 0, W = 0;
        void* workspace = nullptr;
        r2f_params p{};  // seed zeroed
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipEvent_t done = nullptr;  // recorded behind every launch of `exec`: the executable graph must outlive its last replay
        uint64_t last_use = 0;
        bool never = false;  // a capture of this entry failed: kernel by kernel from now on
        bool dyn_armed = false;  // the captured halation launches choose their scratch element on the device
    };
    std::vector<RenderGraph> graphs;
    // Executable graphs that left the cache (evicted, or dropped because the generation moved) while a replay of them may still be
    // running: destroyed once their `done` event has completed -- polled at the next r2f_render, no device-wide synchronisation
    // inside a render (a caller cycling through more than 8 buffer sets would otherwise stall every stream of the device per eviction).
    struct RetiredGraph {
        hipGraph_t graph;
        hipGraphExec_t exec;
        hipEvent_t done;
    };
    std::vector<RetiredGraph> retired;
    uint64_t graphs_generation = 0;  // `generation` the entries (and `warm`) belong to
    uint64_t graph_clock = 0;
    RenderGraph warm;                // structure (shape, layout, parameters) of the last frame launched kernel by kernel
    bool warm_valid = false;
    // buffer sets seen on frames launched kernel by kernel (keys only, most recent last, at most 16): an entry is captured the
    // SECOND time its buffers come by -- a caller that hands in fresh buffers every frame never pays for a capture it cannot reuse
    std::vector<RenderGraph> seen;
    hipStream_t cap_stream = nullptr;
    int opt_render_graph = 1;
    uint64_t stat_replays = 0, stat_captures = 0, stat_eager = 0, stat_dropped = 0;
};

namespace {

int fail(r2f_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define R2F_HIP(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(ctx, R2F_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)


// Every entry point that touches HIP binds the context's device for the duration of the call and puts the caller's
// current device back on return: two contexts on two GPUs can be driven from one thread (and torch's notion of the current
// device is left alone).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t status = hipSuccess;
    explicit DeviceGuard(int device) {
        // (no hipGetLastError() here: a pending error of another user of the runtime in this process -- PyTorch, RCCL -- is theirs to
        // read; the library clears only what it produced itself, right after an abandoned capture in r2f_render)
        status = hipGetDevice(&prev);
        if (status == hipSuccess && prev != device) {
            status = hipSetDevice(device);
            switched = status == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define R2F_GUARD(ctx)                  \
    DeviceGuard guard_((ctx)->device); \
    if (guard_.status != hipSuccess) return fail(ctx, R2F_EHIP, "cannot bind device %d: %s", (ctx)->device, hipGetErrorString(guard_.status))

// Tables change only when a render parameter changes (the reference's caching rule), so the upload path
// is allowed to be slow: wait for every render still in flight on any stream before overwriting a table
// that those kernels may be reading, then copy synchronously.
int upload(r2f_ctx* ctx, DeviceBuf& buf, const void* host, size_t bytes) {
    R2F_HIP(ctx, hipDeviceSynchronize());
    if (buf.bytes < bytes) {
        buf.release();
        R2F_HIP(ctx, hipMalloc(&buf.p, bytes));
        buf.bytes = bytes;
    }
    R2F_HIP(ctx, hipMemcpy(buf.p, host, bytes, hipMemcpyHostToDevice));
    ++ctx->generation;
    return R2F_OK;
}

// (4, m) table -> per channel m-1 cells {xp[i], xp[i+1], fp[i], slope[i]} (plan::curve_cells), uploaded.
int upload_curve(r2f_ctx* ctx, DeviceBuf& buf, DevCurve& cv, const float* lut, int m) {
    if (!lut || m < 2) return fail(ctx, R2F_EINVAL, "curve: need a (4, m) table with m >= 2");
    plan::CurveCells cc;
    if (plan::curve_cells(lut, m, &cc)) return fail(ctx, R2F_EINVAL, "curve: xp must be non-decreasing");
    static_assert(sizeof(float4) == 4 * sizeof(float), "a cell is one float4");
    int rc = upload(ctx, buf, cc.cells.data(), cc.cells.size() * sizeof(float));
    if (rc) return rc;
    cv.cells = static_cast<const float4*>(buf.p);
    cv.m = cc.m;
    cv.x0 = cc.x0;
    cv.x1 = cc.x1;
    cv.inv_step = cc.inv_step;
    cv.near = cc.near;
    for (int c = 0; c < 3; ++c) cv.f_first[c] = cc.f_first[c], cv.f_last[c] = cc.f_last[c];
    return R2F_OK;
}

// Build (or reuse) the device form of stencil `which` for a tile TW x TH, Q rows per lane, and an LDS
// budget in bytes (0 = whole stencil height in one phase).
int ensure_stencil(r2f_ctx* ctx, int which, int Q, int TW, int TH, size_t lds_budget, bool common_box) {
    const bool allow_sym = ctx->opt_sym != 0;
    StencilSet& s = ctx->stencil[which];
    if (!s.present) return fail(ctx, R2F_EINVAL, "stencil %d not set (r2f_set_kernel)", which);
    if (s.built_q == Q && s.built_tw == TW && s.built_th == TH && s.built_budget == lds_budget && s.common_box == common_box &&
        s.built_sym == allow_sym)
        return R2F_OK;
    const plan::Taps taps{s.host.data(), s.kh, s.kw, s.kc};
    int box[3][4];
    for (int c = 0; c < 3; ++c) plan::tap_box(taps, c, box[c]);
    if (common_box) {
        for (int c = 1; c < 3; ++c) {
            box[0][0] = std::min(box[0][0], box[c][0]);
            box[0][1] = std::max(box[0][1], box[c][1]);
            box[0][2] = std::min(box[0][2], box[c][2]);
            box[0][3] = std::max(box[0][3], box[c][3]);
        }
        for (int c = 1; c < 3; ++c) memcpy(box[c], box[0], sizeof box[0]);
    }
    plan::StreamHost sh;
    for (int c = 0; c < 3; ++c) {
        // mirror symmetry about the anchor column, bit for bit?  (needs an odd box centred on the anchor)
        bool sym = allow_sym && plan::mirror_symmetric(taps, c, box[c]);
        if (sym && common_box)  // shared geometry (grain): pair taps only if every channel allows it
            sym = plan::mirror_symmetric(taps, 0, box[c]) && plan::mirror_symmetric(taps, 1, box[c]) && plan::mirror_symmetric(taps, 2, box[c]);
        plan::StencilGeom g;
        if (plan::plan_stencil_channel(taps, c, box[c], sym, Q, TW, TH, lds_budget, &g, &sh))
            return fail(ctx, R2F_ETOOLARGE, "stencil %d: %d-tap rows do not fit the LDS budget", which, g.kw);
        DevStencil& d = s.dev[c];
        d.kh = g.kh, d.kw = g.kw, d.kw_pad = g.kw_pad, d.RS = g.RS, d.ay = g.ay, d.ax = g.ax, d.sym = g.sym;
        d.wmul = 1;
        d.n_phases = g.n_phases, d.n_rowsteps = g.n_rowsteps, d.n_entries = g.n_entries;
        d.mask_first_or = g.mask_first_or, d.mask_last_or = g.mask_last_or, d.max_lds_rows = g.max_lds_rows;
        s.geom[c] = g;
        int rc = upload(ctx, s.wbuf[c], sh.w.data(), sh.w.size() * sizeof(float));
        if (rc) return rc;
        // row-step records (16-byte aligned: read with s_load_dwordx4) and phase records share one allocation
        std::vector<int> meta(sh.rowinfo);
        const size_t n_ri = meta.size();
        meta.insert(meta.end(), sh.phases.begin(), sh.phases.end());
        rc = upload(ctx, s.mbuf[c], meta.data(), meta.size() * sizeof(int));
        if (rc) return rc;
        d.wstream = static_cast<const float*>(s.wbuf[c].p);
        d.rowinfo = static_cast<const int*>(s.mbuf[c].p);
        d.phases = d.rowinfo + n_ri;
    }
    s.built_q = Q;
    s.built_tw = TW;
    s.built_th = TH;
    s.built_budget = lds_budget;
    s.built_sym = allow_sym;
    s.common_box = common_box;
    return R2F_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

bool planes_vec_ok(const r2f_planes* pl, int W) {
    return W % 4 == 0 && aligned16(pl->data) && pl->plane_stride % 4 == 0;
}

// Does any plane of `a` (rows x W floats, plane_stride apart) share bytes with any plane of `b`?
bool planes_overlap(const r2f_planes* a, const r2f_planes* b, int W) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const float* a0 = a->data + i * a->plane_stride;
            const float* b0 = b->data + j * b->plane_stride;
            const float* a1 = a0 + (int64_t)a->rows * W;
            const float* b1 = b0 + (int64_t)b->rows * W;
            if (reinterpret_cast<uintptr_t>(a0) < reinterpret_cast<uintptr_t>(b1) &&
                reinterpret_cast<uintptr_t>(b0) < reinterpret_cast<uintptr_t>(a1))
                return true;
        }
    return false;
}

DevPlanes to_dev(const r2f_planes* pl) {
    DevPlanes d;
    d.data = pl->data;
    d.plane_stride = pl->plane_stride;
    d.gy0 = pl->gy0;
    d.rows = pl->rows;
    return d;
}

int check_rows(r2f_ctx* ctx, const char* what, const r2f_planes* pl, int lo, int hi) {
    if (!pl || !pl->data) return fail(ctx, R2F_EINVAL, "%s: null planes", what);
    if (lo < pl->gy0 || hi > pl->gy0 + pl->rows)
        return fail(ctx, R2F_EINVAL, "%s: rows [%d, %d) not inside the buffer's [%d, %d)", what, lo, hi, pl->gy0,
                    pl->gy0 + pl->rows);
    return R2F_OK;
}

// Source rows a stencil with `above`/`below` taps needs for outputs [y0, y1), after reflect-101.
int check_stencil_source(r2f_ctx* ctx, const char* what, const r2f_planes* src, int y0, int y1, int above, int below,
                         int H) {
    int lo, hi;
    plan::stencil_source_rows(y0, y1, above, below, H, &lo, &hi);
    return check_rows(ctx, what, src, lo, hi);
}

// Tile order for a gx x gy grid of stencil workgroups (plan::tile_order: one contiguous run of tiles per XCD, walked in bands
// of tile columns), cached per grid.
int ensure_tile_order(r2f_ctx* ctx, int gx, int gy, const int** out) {
    for (auto& t : ctx->tile_order)
        if (t.gx == gx && t.gy == gy && t.buf.p) {
            *out = static_cast<const int*>(t.buf.p);
            return R2F_OK;
        }
    const std::vector<int> order = plan::tile_order(gx, gy, ctx->opt_xcd_band);
    r2f_ctx::TileOrder& slot = ctx->tile_order[ctx->tile_order_next];
    ctx->tile_order_next = (ctx->tile_order_next + 1) % 4;
    int rc = upload(ctx, slot.buf, order.data(), order.size() * sizeof(int));
    if (rc) return rc;
    slot.gx = gx;
    slot.gy = gy;
    *out = static_cast<const int*>(slot.buf.p);
    return R2F_OK;
}

int ensure_bytes(r2f_ctx* ctx, DeviceBuf& buf, size_t bytes) {
    if (buf.bytes >= bytes) return R2F_OK;
    R2F_HIP(ctx, hipDeviceSynchronize());
    buf.release();
    R2F_HIP(ctx, hipMalloc(&buf.p, bytes));
    buf.bytes = bytes;
    ++ctx->generation;
    return R2F_OK;
}

// p->seed -> the context's device-side frame block, in stream order (a one-lane kernel: its by-value argument is copied at
// launch time, so no host staging buffer has to outlive the call).
RangeRecord record_of(const r2f_ctx* ctx) {
    RangeRecord r;
    r.blk = static_cast<FrameParams*>(ctx->frame_buf.p);
    r.tiles = static_cast<int2*>(ctx->range_tiles.p);
    r.tyn = ctx->tiles_tyn, r.txn = ctx->tiles_txn;
    return r;
}

// The record's tile grid for an H x W frame (allocated on first use and when the frame grows; a fresh grid says "unknown").
int ensure_range_tiles(r2f_ctx* ctx, int H_global, int W) {
    const int tyn = (H_global + kRangeTileRows - 1) / kRangeTileRows, txn = (W + kRangeTileCols - 1) / kRangeTileCols;
    if (ctx->range_tiles.p && tyn <= ctx->tiles_tyn && txn == ctx->tiles_txn) return R2F_OK;
    R2F_HIP(ctx, hipDeviceSynchronize());  // (kernels of an earlier frame may still be reading the old grid)
    const int new_tyn = std::max(tyn, ctx->tiles_txn == txn ? ctx->tiles_tyn : 0);
    const size_t n = (size_t)new_tyn * txn;
    std::vector<int2> init(n, make_int2((int)kFrameMinReset, (int)kFrameMaxReset));
    if (ctx->range_tiles.bytes < n * sizeof(int2)) {
        ctx->range_tiles.release();
        R2F_HIP(ctx, hipMalloc(&ctx->range_tiles.p, n * sizeof(int2)));
        ctx->range_tiles.bytes = n * sizeof(int2);
    }
    R2F_HIP(ctx, hipMemcpy(ctx->range_tiles.p, init.data(), n * sizeof(int2), hipMemcpyHostToDevice));
    ctx->tiles_tyn = new_tyn, ctx->tiles_txn = txn;
    ++ctx->generation;  // (captured launches hold the grid's address and dimensions)
    return R2F_OK;
}

// mode 1: seed + reset of the exposure range (the start of a render); 0: a stage entry's own seed write in the middle of one;
// 2: the range reset alone (a render whose caller keeps the seed resident); 3: the range made unusable (frame_params_kernel).
// Modes 1 and 2 reset the tile grid too.
int write_frame_params(r2f_ctx* ctx, const r2f_params* p, hipStream_t s, int mode = 1) {
    FrameParams v{};
    v.seed = p->seed;
    v.e_min = kFrameMinReset, v.e_max = kFrameMaxReset;
    R2F_HIP(ctx, launch_frame_params(record_of(ctx), v, mode, s));
    return R2F_OK;
}

// Destroy the retired graphs whose last replay has completed (wait = true: all of them, after their events -- r2f_destroy).
void reap_retired_graphs(r2f_ctx* ctx, bool wait) {
    // Graphs whose completion cannot be asked for -- no event (its creation failed at capture time), or one recorded inside a
    // caller's stream capture -- are kept; when more than a handful of entries pile up the device is synchronised once and
    // everything goes (a missing event used to read as "complete", and the list could grow without bound).
    if (!wait && ctx->retired.size() > 16) {
        (void)hipDeviceSynchronize();
        wait = true;
    }
    size_t kept = 0;
    for (auto& r : ctx->retired) {
        hipError_t e = r.done ? (wait ? hipEventSynchronize(r.done) : hipEventQuery(r.done)) : (wait ? hipSuccess : hipErrorNotReady);
        if (e == hipErrorNotReady) {
            ctx->retired[kept++] = r;
            continue;
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();  // (an event recorded inside a caller's stream capture cannot be queried: keep the graph until
            if (!wait) {              //  the list is flushed behind a device synchronisation, above or in r2f_destroy)
                ctx->retired[kept++] = r;
                continue;
            }
        }
        if (r.exec) (void)hipGraphExecDestroy(r.exec);
        if (r.graph) (void)hipGraphDestroy(r.graph);
        if (r.done) (void)hipEventDestroy(r.done);
    }
    ctx->retired.resize(kept);
}

// Take entry `g` out of service: its executable graph may still be replaying, so it is parked behind its `done` event.
void retire_render_graph(r2f_ctx* ctx, r2f_ctx::RenderGraph& g) {
    if (g.exec) {
        ctx->retired.push_back({g.graph, g.exec, g.done});
        ++ctx->stat_dropped;
    } else {
        if (g.graph) (void)hipGraphDestroy(g.graph);
        if (g.done) (void)hipEventDestroy(g.done);
    }
    g.graph = nullptr, g.exec = nullptr, g.done = nullptr;
}

void drop_render_graphs(r2f_ctx* ctx) {
    // a replay of one of them may still be running (an option changed between two asynchronous frames; table uploads and buffer
    // growth have synchronised already): an executable graph must outlive its last launch -- parked, not waited for
    for (auto& g : ctx->graphs) retire_render_graph(ctx, g);
    ctx->graphs.clear();
    ctx->warm_valid = false;
    reap_retired_graphs(ctx, false);
}

plan::Taps taps_of(const StencilSet& s) { return plan::Taps{s.host.data(), s.kh, s.kw, s.kc}; }

// Bounding box of channel c's non-zero taps: {i_lo, i_hi, j_lo, j_hi}; an all-zero plane keeps its centre tap.
void tap_box(const StencilSet& s, int c, int box[4]) { plan::tap_box(taps_of(s), c, box); }

// R if the channels `chans` of a stencil all fill the same square (2 R + 1)^2 box of non-zero taps around the anchor,
// 1 <= R <= max_r, left-right mirror symmetric, and their device form has the geometry stencil_fixed expects; else 0.
int fixed_stencil_radius(const StencilSet& set, const int* chans, int nch, int max_r) {
    static_assert(plan::fixed_ax(4) == fixed_stencil_ax(4) && plan::fixed_ax(7) == fixed_stencil_ax(7) && plan::fixed_ax(11) == fixed_stencil_ax(11),
                  "plan::fixed_ax restates fixed_stencil_ax");
    return plan::fixed_stencil_radius(taps_of(set), set.geom, chans, nch, max_r);
}

std::vector<float> fixed_stencil_weights(const StencilSet& set, int R, int Q, bool* same) {
    return plan::fixed_stencil_weights(taps_of(set), R, Q, same);
}

constexpr int kFftMaxTaps = plan::kFftMaxTaps;
constexpr int kFixedMaxR = 11;  // largest unrolled direct form (23 x 23)
// Up to which radius the unrolled direct form is preferred over the FFT form (tools/fft_probe.py, 24 MP x 3 channels): against
// complex128 scratch it wins up to 23 x 23 (0.54 vs 0.55 ms; 25 x 25: 0.81 vs 0.57); against the complex64 scratch of the MTF
// passes (0.42-0.45 ms whatever the taps) only up to 19 x 19 (0.40; 21 x 21: 0.45 vs 0.45, 23 x 23: 0.54 vs 0.43).
int fixed_preferred_max_r(const r2f_ctx* ctx, int which) { return ((ctx->opt_fft_s32 >> which) & 1) ? 9 : kFixedMaxR; }

// Does channel c of stencil `which` take the overlap-save FFT form?
bool fft_eligible(const r2f_ctx* ctx, const StencilSet& s, int c) {
    if (!ctx->opt_fft) return false;
    int b[4];
    tap_box(s, c, b);
    const int bh = b[1] - b[0] + 1, bw = b[3] - b[2] + 1;
    // up to 400 taps a side: a 512-point window then still yields 113 x 112 outputs (boxes over 200 taps on an axis take the
    // 512-point window there, see fft_window)
    if (!(bh <= kFftMaxTaps && bw <= kFftMaxTaps)) return false;
    // A channel with taps of BOTH signs (the unsharp-masked MTF kernel, effects.py:179-183 with sharpening_strength > 0) cancels:
    // a float32 sum is accurate relative to sum |w x|, not to the result -- 4-8 ulp of a density, which a stepped grain LUT or a
    // steep output LUT then multiplies (found by the round-4 fuzz criterion against the float64 evaluation: one case of 1 100 at
    // 1.46 x its bound).  Such a channel takes the float64 FFT form whatever its size, on complex128 scratch (run_stencil_fft):
    // one rounding, like the oracle's.  Preview-scale unsharp masking is rare and its frames are small; the headline has none.
    // (`stencil_fft_mixed_sign = 0` hands such channels back to the size rules below -- the A/B switch, and the way to keep the
    // stencil_fft_min_taps / stencil_fft_scratch32 knobs in charge of them)
    if (s.mixed_sign[c] && ctx->opt_fft_mixed_sign) return true;
    if (bh * bw < ctx->opt_fft_min_taps) return false;
    // square mirror-symmetric stencils up to 23 x 23 are faster in the unrolled direct form (needs the device form: callers
    // run ensure_stencil first)
    const int which = (int)(&s - ctx->stencil);
    if (ctx->opt_stencil_fixed && ctx->opt_variant <= 0 && s.built_q && fixed_stencil_radius(s, &c, 1, fixed_preferred_max_r(ctx, which)))
        return false;
    return true;
}

// The rule of the halation's scratch element (FftConvArgs::dyn): the 12-byte element when max |x| <= bound x max(min x, floor).
//   |delta density| <= 0.434 x steepest curve cell x kDynCoefficient x (max / shadow) <= 3.6e-7  (three ulps of a density in [1, 2));
//   below the curve's first breakpoint np.interp clamps: no slope, so shadows under it do not count.
// kDynCoefficient = what the element costs a shadow's exposure, as a multiple of max / shadow: 1.5 x 2^-36.  The element keeps 36
// mantissa bits per component (half an ulp = 2^-37 of the value) and a scratch value is rounded twice (pass 1's store, pass 2's);
// the worst frame is a nearly flat bright field around a dark hole: all of a window's energy sits in one spectral line per row,
// whose two roundings reach every output of the window undiminished (the stencil's taps sum to 1) -- 2 x 2^-37 = 1.46e-11 of
// max / shadow.  A search over 1 050 random frames of seven families x five bright-region statistics found exactly that and nothing
// above it (1.43e-11 for holes in a jittered flat field; noise-like windows 1.0e-11, isolated speculars 4.5e-12:
// profiles/r06_scratch96_probe.txt; tests/test_gpu_fft.py repeats the search on a fixed budget).  Round 5 shipped 6.0e-12, the
// maximum of 16 hand-made probes without a flat field among them: too small by 2.4.  The factor
// 1.5 covers what no search of 7e5-pixel frames sees of a 1e8-pixel frame's tail.
constexpr double kDynCoefficient = 1.5 * 1.4551915228366852e-11;  // 1.5 x 2^-36 = 2.18e-11
void dyn_rule(const r2f_ctx* ctx, float* bound, float* floor) {
    const double slope = std::max((double)ctx->curve_slope_max, 1e-3);
    *bound = (float)std::min(3.6e-7 / (0.4343 * slope * kDynCoefficient), 1e7);
    *floor = (float)std::pow(10.0, (double)ctx->curve.x0);
}

plan::FftOptions fft_options(const r2f_ctx* ctx) {
    plan::FftOptions o;
    o.window = ctx->opt_fft_window;
    o.window_max = ctx->opt_fft_window_max;
    o.window_rows = ctx->opt_fft_window_rows;
    o.batch_mib = ctx->opt_fft_batch;
    o.streams = ctx->opt_fft_streams;
    o.even = ctx->opt_fft_even;
    return o;
}

// The channels `chans` of a stencil (all with the same tap box) as fp64 overlap-save FFT correlations (r2f_fft.hip);
// their window pairs share the launches.
// dyn: the caller vouches that the context's exposure-range record (its tiles) was kept for the samples `src` holds this frame: the
// passes may then choose their scratch element on the device, per window pair (FftConvArgs::dyn_flags, fft_decide_kernel).
int run_stencil_fft(r2f_ctx* ctx, int which, const int* chans, int nch, const r2f_planes* src, const r2f_planes* dst, int y0, int y1,
                    int W, int H, int epilogue, float log_eps, hipStream_t s, bool dyn) {
    StencilSet& set = ctx->stencil[which];
    int b[4];
    tap_box(set, chans[0], b);
    const int bh = b[1] - b[0] + 1, bw = b[3] - b[2] + 1;
    int ny = 256, nx = 256;
    // (the rows of THIS call, not of the global frame: a row shard's 1 058-row call is 6.15 window rows of 172 and should not be
    // planned as if the seventh were free -- VERDICT r3, next 4; a whole-frame call sees the frame as before)
    const plan::FftOptions fo = fft_options(ctx);
    // scratch element of THIS launch: complex64 only where the stencil's bit says so and no channel of the group has cancelling taps
    // (those keep complex128: no float32 rounding between the passes either) -- the window choice is priced with it
    bool s32_eff = (ctx->opt_fft_s32 >> which) & 1;
    for (int i = 0; i < nch; ++i)
        if (set.mixed_sign[chans[i]] && ctx->opt_fft_mixed_sign) s32_eff = false;
    if (!plan::fft_window(fo, bh, bw, W, y1 - y0, s32_eff, &ny, &nx))
        return fail(ctx, R2F_EINVAL, "stencil %d: no FFT window shape fits a %d x %d tap box under the current stencil_fft_window* options",
                    which, bh, bw);
    const size_t img = (size_t)ny * nx;
    const int shape = r2f_ctx::fft_shape_index(ny, nx);
    if (!ctx->fft_tw.p) {
        // W_256^k, k < 256, then W_512^k, k < 256, then W_1024^k, k < 64
        std::vector<double> tw(4 * kFftN + 2 * 64);
        const double pi = 3.14159265358979323846264338327950288;
        for (int k = 0; k < kFftN; ++k) {
            tw[2 * k] = std::cos(-2.0 * pi * k / kFftN), tw[2 * k + 1] = std::sin(-2.0 * pi * k / kFftN);
            tw[2 * (kFftN + k)] = std::cos(-pi * k / kFftN), tw[2 * (kFftN + k) + 1] = std::sin(-pi * k / kFftN);
        }
        for (int k = 0; k < 64; ++k)
            tw[2 * (2 * kFftN + k)] = std::cos(-2.0 * pi * k / 1024.0), tw[2 * (2 * kFftN + k) + 1] = std::sin(-2.0 * pi * k / 1024.0);
        int rc = upload(ctx, ctx->fft_tw, tw.data(), tw.size() * sizeof(double));
        if (rc) return rc;
    }
    FftConvArgs a;
    memset(&a, 0, sizeof a);
    a.tw = static_cast<const double2*>(ctx->fft_tw.p);
    a.tw512 = a.tw + kFftN;
    a.tw1024 = a.tw + 2 * kFftN;
    a.ny = ny, a.nx = nx;
    a.ay = set.kh / 2 - b[0];  // anchor (kh/2, kw/2): cv.filter2D's default
    a.ax = set.kw / 2 - b[2];
    a.vy = ny - bh + 1;
    a.vx = (nx - bw + 1) & ~3;  // a multiple of 4: window origins stay 16-byte aligned for the float4 stores of pass 3
    // Real spectra: every channel of the group centrally symmetric around an anchor at the centre of its (odd x odd) box
    bool kreal = ctx->opt_fft_real && (bh & 1) && (bw & 1) && a.ay == bh / 2 && a.ax == bw / 2;
    for (int i = 0; i < nch && kreal; ++i) {
        const int kc = set.kc == 1 ? 0 : chans[i];
        for (int y = 0; y < bh && kreal; ++y)
            for (int x = 0; x < bw; ++x) {
                const float u = set.host[((size_t)(b[0] + y) * set.kw + b[2] + x) * set.kc + kc];
                const float v = set.host[((size_t)(b[1] - y) * set.kw + b[3] - x) * set.kc + kc];
                if (memcmp(&u, &v, sizeof u)) {
                    kreal = false;
                    break;
                }
            }
    }
    a.kreal = kreal ? 1 : 0;
    a.oy = kreal ? a.ay : 0;
    a.ox = kreal ? a.ax : 0;
    if (!ctx->cu_count) {
        int n = 0;
        R2F_HIP(ctx, hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device));
        ctx->cu_count = n > 0 ? n : 256;
    }
    a.cols_walk = ctx->opt_fft_cols_walk;
    a.cols_slots = 2 * ctx->cu_count;
    int rc = ensure_bytes(ctx, ctx->fft_s1, img * sizeof(double2));
    if (rc) return rc;
    for (int i = 0; i < nch; ++i) {
        const int c = chans[i];
        ctx->fft_kf_dims[which][c] = ny * 4096 + nx;
        ctx->fft_last_real[which][c] = kreal ? 1 : 0;
        if (ctx->fft_kf_valid[which][c][shape] && ctx->fft_kf_real[which][c][shape] == kreal) continue;
        // the kernel's spectrum: the same two forward passes on its zero-padded image -- the box in the window's top left corner
        // (outputs from row / column 0 on), or, for a real spectrum, wrapped around the window with its anchor on the origin
        std::vector<float> kimg(img, 0.f);
        const int kc = set.kc == 1 ? 0 : c;
        for (int y = 0; y < bh; ++y)
            for (int x = 0; x < bw; ++x) {
                const int py = kreal ? (y - a.ay + ny) % ny : y, px = kreal ? (x - a.ax + nx) % nx : x;
                kimg[(size_t)py * nx + px] = set.host[((size_t)(b[0] + y) * set.kw + b[2] + x) * set.kc + kc];
            }
        rc = upload(ctx, ctx->fft_kimg, kimg.data(), img * sizeof(float));
        if (rc) return rc;
        rc = ensure_bytes(ctx, ctx->fft_kf[which][c][shape], img * sizeof(double2));
        if (rc) return rc;
        ctx->fft_kf_real[which][c][shape] = kreal;
        FftConvArgs k = a;
        k.src.data = static_cast<float*>(ctx->fft_kimg.p);
        k.raw = 1;
        k.s32 = 0;  // spectra are always built and kept in complex128
        k.nch = 1, k.chan[0] = 0, k.ppc = 1;
        k.ntiles = 1, k.gx = 1, k.npairs = 1, k.pair0 = 0;
        k.s1 = static_cast<double2*>(ctx->fft_s1.p);
        k.kf_out = static_cast<double2*>(ctx->fft_kf[which][c][shape].p);
        R2F_HIP(ctx, launch_fft_rows_fwd(k, s));
        R2F_HIP(ctx, launch_fft_cols(k, 1, s));
        ctx->fft_kf_valid[which][c][shape] = true;
    }
    a.src = to_dev(src);
    a.dst = to_dev(dst);
    a.nch = nch;
    for (int i = 0; i < nch; ++i) {
        a.chan[i] = chans[i];
        a.kfs[i] = static_cast<const double2*>(ctx->fft_kf[which][chans[i]][shape].p);
    }
    a.y0 = y0, a.y1 = y1, a.W = W, a.H_global = H;
    a.s32 = ((ctx->opt_fft_s96 >> which) & 1) ? 2 : (s32_eff ? 1 : 0);
    for (int i = 0; i < nch; ++i)
        if (set.mixed_sign[chans[i]] && ctx->opt_fft_mixed_sign) a.s32 = 0;  // (the 12-byte element neither)
    // the halation of a whole-frame render whose front kernel recorded the range of the exposure planes: element chosen on the device.
    // The guard compares a window's SAMPLES; it speaks for the outputs only under a stencil of unit gain (the reference's halation
    // kernels are normalised and non-negative, effects.py:200-217): taps that cancel or sum to 0.01 make outputs far below the samples
    bool unit_gain = true;
    for (int i = 0; i < nch; ++i) unit_gain = unit_gain && set.unit_gain[chans[i]];
    if (a.s32 == 0 && which == R2F_KERNEL_HALATION && epilogue == 1 && dyn && ctx->opt_fft_s96_auto && a.kreal && unit_gain &&
        ny == 256 && ctx->opt_fft_cols_walk && ctx->curve.cells) {
        a.s32 = 3;  // (the flags are worked out below, once the call's tiling is known)
        ctx->frame_dyn_armed = true;
    }
    const plan::FftBatches fb = plan::fft_batches(fo, ny, nx, bh, bw, W, y0, y1, nch, a.s32 == 1 ? 8 : (a.s32 == 2 ? 12 : 16));  // (3: sized for 16)
    a.gx = fb.gx;
    a.ntiles = fb.ntiles;
    a.ppc = fb.ppc;
    a.epilogue = epilogue;
    a.epi_lds = ctx->opt_fft_epi_lds;
    a.curve = ctx->curve;
    a.log_eps = log_eps;
    a.vec4 = planes_vec_ok(dst, W) ? 1 : 0;
    const size_t img_bytes = fb.img_bytes;
    const int pairs = fb.pairs, nstreams = fb.nstreams, batch = fb.batch;
    rc = ensure_bytes(ctx, ctx->fft_s1, fb.scratch_bytes);
    if (rc) return rc;
    if (a.s32 == 3) {
        // the element of every window pair of this call, from the exposure-range tiles its caller vouches for: one small launch on the
        // caller's stream ahead of the passes (they fan out to the internal streams behind it)
        rc = ensure_range_tiles(ctx, H, W);
        if (rc) return rc;
        rc = ensure_bytes(ctx, ctx->dyn_flags, (size_t)std::max(fb.ppc, 1) * sizeof(int));
        if (rc) return rc;
        float bound, floor_;
        dyn_rule(ctx, &bound, &floor_);
        a.dyn_flags = static_cast<const int*>(ctx->dyn_flags.p);
        ctx->dyn_flags_ppc = fb.ppc;
        R2F_HIP(ctx, launch_fft_decide(a, record_of(ctx), bound, floor_, static_cast<int*>(ctx->dyn_flags.p), s));
    }
    hipStream_t lanes[4] = {s, s, s, s};
    if (nstreams > 1) {
        for (int i = 0; i < nstreams; ++i) {
            if (!ctx->fft_stream[i]) R2F_HIP(ctx, hipStreamCreateWithFlags(&ctx->fft_stream[i], hipStreamNonBlocking));
            if (!ctx->fft_ev_out[i]) R2F_HIP(ctx, hipEventCreateWithFlags(&ctx->fft_ev_out[i], hipEventDisableTiming));
            lanes[i] = ctx->fft_stream[i];
        }
        if (!ctx->fft_ev_in) R2F_HIP(ctx, hipEventCreateWithFlags(&ctx->fft_ev_in, hipEventDisableTiming));
        R2F_HIP(ctx, hipEventRecord(ctx->fft_ev_in, s));  // everything queued on the caller's stream so far (src, spectra)
        for (int i = 0; i < nstreams; ++i) R2F_HIP(ctx, hipStreamWaitEvent(lanes[i], ctx->fft_ev_in, 0));
    }
    auto timed = [&](int cls, double bytes, hipStream_t st, auto&& launch) -> int {
        if (!(ctx->opt_timing & (1 << cls))) {
            R2F_HIP(ctx, launch());
            return R2F_OK;
        }
        hipEvent_t e0, e1;
        R2F_HIP(ctx, hipEventCreate(&e0));
        R2F_HIP(ctx, hipEventCreate(&e1));
        R2F_HIP(ctx, hipEventRecord(e0, st));
        R2F_HIP(ctx, launch());
        R2F_HIP(ctx, hipEventRecord(e1, st));
        ctx->timing_ev[cls + 3 * (a.s32 == 1)].push_back({e0, e1});
        ctx->timing_bytes[cls + 3 * (a.s32 == 1)] += bytes;
        return R2F_OK;
    };
    int turn = 0;
    for (int p0 = 0; p0 < pairs; p0 += batch, turn = (turn + 1) % nstreams) {
        const int li = turn;
        hipStream_t st = lanes[li];
        a.pair0 = p0;
        a.npairs = std::min(batch, pairs - p0);
        a.s1 = reinterpret_cast<double2*>(static_cast<char*>(ctx->fft_s1.p) + (size_t)li * batch * img_bytes);
        // algorithmic bytes of the passes: window floats in (2 per pair) + scratch image out; scratch in + valid rows out;
        // valid rows in + valid outputs out.  The kernel spectrum (1 MB) stays in L2.
        const double np = a.npairs, full = (double)img_bytes, part = full * a.vy / ny;
        rc = timed(0, np * (2.0 * img * sizeof(float) + full), st, [&] { return launch_fft_rows_fwd(a, st); });
        if (rc) return rc;
        rc = timed(1, np * (full + part), st, [&] { return launch_fft_cols(a, 0, st); });
        if (rc) return rc;
        rc = timed(2, np * (part + 2.0 * a.vy * a.vx * sizeof(float)), st, [&] { return launch_fft_rows_inv(a, st); });
        if (rc) return rc;
    }
    if (nstreams > 1)
        for (int i = 0; i < nstreams; ++i) {
            R2F_HIP(ctx, hipEventRecord(ctx->fft_ev_out[i], lanes[i]));
            R2F_HIP(ctx, hipStreamWaitEvent(s, ctx->fft_ev_out[i], 0));
        }
    return R2F_OK;
}

// Is channel c's stencil a single tap at the anchor?  (*w = its weight)
bool single_tap_channel(const StencilSet& set, int c, float* w) { return plan::single_tap_channel(taps_of(set), c, w); }

// skip_identity: the single-tap channels were finished by the front kernel (r2f_stage_front_split) -- leave them alone.
int run_stencil(r2f_ctx* ctx, int which, const r2f_planes* src, const r2f_planes* dst, int y0, int y1, int W, int H,
                int epilogue, float log_eps, hipStream_t s, bool skip_identity = false, bool dyn = false) {
    if (y1 <= y0) return R2F_OK;
    if (W <= 0 || H <= 0 || y0 < 0 || y1 > H) return fail(ctx, R2F_EINVAL, "stencil: bad geometry");
    StencilSet& set = ctx->stencil[which];
    if (!set.present) return fail(ctx, R2F_EINVAL, "stencil %d not set (r2f_set_kernel)", which);
    // choose the tile variant and LDS budget: the widest tile whose rows fit; wide stencils fall back to the
    // narrow tile and, if need be, to one workgroup per CU (the whole 160 KB)
    int variant = -1;
    const size_t budgets[2] = {(size_t)ctx->opt_lds_kb * 1024, kMaxLds};
    for (int b = 0; b < 2 && variant < 0; ++b) {
        for (int v = 0; v < kNumStencilVariants; ++v) {
            if (ctx->opt_variant >= 0 && v != ctx->opt_variant) continue;
            const StencilVariant& sv = kStencilVariants[v];
            int rc = ensure_stencil(ctx, which, sv.Q, sv.TW(), sv.TH(), budgets[b], false);
            if (rc == R2F_ETOOLARGE) continue;
            if (rc) return rc;
            if (stencil_lds_bytes(sv, set.dev, 3) <= kMaxLds) {
                variant = v;
                break;
            }
        }
    }
    if (variant < 0) return fail(ctx, R2F_ETOOLARGE, "stencil %d: %dx%d taps do not fit an LDS tile", which, set.kh, set.kw);
    int rc = check_rows(ctx, "stencil dst", dst, y0, y1);
    if (rc) return rc;
    for (int c = 0; c < 3; ++c) {
        const DevStencil& d = set.dev[c];
        rc = check_stencil_source(ctx, "stencil src", src, y0, y1, d.ay, d.kh - 1 - d.ay, H);
        if (rc) return rc;
    }
    if (epilogue == 1 && !ctx->curve.cells) return fail(ctx, R2F_EINVAL, "density curve not set (r2f_set_curve1d)");
    if (planes_overlap(src, dst, W))  // tiles (and FFT batches) read halo rows that others have already overwritten
        return fail(ctx, R2F_EINVAL, "stencil %d: source and destination planes overlap (the stencil stages are out of place)", which);
    StencilArgs a;
    for (int c = 0; c < 3; ++c) {
        a.st[c] = set.dev[c];
        if (ctx->opt_ablate == 3) a.st[c].wmul = 0;
    }
    a.src = to_dev(src);
    a.dst = to_dev(dst);
    a.y0 = y0;
    a.y1 = y1;
    a.W = W;
    a.H_global = H;
    a.epilogue = epilogue;
    a.curve = ctx->curve;
    a.log_eps = log_eps;
    // large kernels take the fp64 FFT form (channels with the same tap box share their launches); single taps are
    // pointwise; the direct kernel runs the rest
    a.nchan = 0;
    bool done[3] = {false, false, false};
    for (int c = 0; c < 3; ++c) {
        if (done[c]) continue;
        int tb[4];
        tap_box(set, c, tb);
        if (fft_eligible(ctx, set, c)) {
            int group[3], ng = 0;
            for (int d = c; d < 3; ++d) {
                int ob[4];
                tap_box(set, d, ob);
                if (!done[d] && fft_eligible(ctx, set, d) && !memcmp(ob, tb, sizeof tb)) group[ng++] = d, done[d] = true;
            }
            rc = run_stencil_fft(ctx, which, group, ng, src, dst, y0, y1, W, H, epilogue, log_eps, s, dyn);
            if (rc) return rc;
        } else if (tb[0] == tb[1] && tb[2] == tb[3] && tb[0] == set.kh / 2 && tb[2] == set.kw / 2 && ctx->opt_ablate == 0) {
            if (skip_identity) continue;
            TapArgs t;  // a single tap at the anchor: pointwise
            t.src = to_dev(src), t.dst = to_dev(dst);
            t.ch = c, t.y0 = y0, t.y1 = y1, t.W = W;
            t.w = set.host[((size_t)tb[0] * set.kw + tb[2]) * set.kc + (set.kc == 1 ? 0 : c)];
            t.epilogue = epilogue, t.curve = ctx->curve, t.log_eps = log_eps;
            t.vec = (planes_vec_ok(src, W) && planes_vec_ok(dst, W)) ? 1 : 0;
            R2F_HIP(ctx, launch_single_tap(t, s));
        } else {
            a.chan[a.nchan++] = c;
        }
    }
    if (a.nchan == 0) return R2F_OK;
    a.vec = planes_vec_ok(dst, W) ? 1 : 0;
    a.xcd_remap = ctx->opt_xcd_remap;
    a.order = nullptr;
    if (a.xcd_remap == 2) {
        const StencilVariant& sv = kStencilVariants[variant];
        rc = ensure_tile_order(ctx, (W + sv.TW() - 1) / sv.TW(), (y1 - y0 + sv.TH() - 1) / sv.TH(), &a.order);
        if (rc) return rc;
    }
    a.ablate = ctx->opt_ablate;
    // small square mirror-symmetric stencils (below the FFT threshold: up to 19 x 19) take the unrolled form
    a.fixed_r = 0;
    a.fixed_w = nullptr;
    if (variant == 0 && ctx->opt_stencil_fixed && ctx->opt_ablate == 0) {
        const int R = fixed_stencil_radius(set, a.chan, a.nchan, kFixedMaxR);
        if (R) {
            if (!ctx->stencil_fixed_valid[which]) {
                bool same;
                const std::vector<float> w = fixed_stencil_weights(set, R, kStencilVariants[0].Q, &same);
                rc = upload(ctx, ctx->stencil_fixed_w[which], w.data(), w.size() * sizeof(float));
                if (rc) return rc;
                ctx->stencil_fixed_valid[which] = true;
            }
            a.fixed_r = R;
            a.fixed_w = static_cast<const float*>(ctx->stencil_fixed_w[which].p);
        }
    }
    R2F_HIP(ctx, launch_stencil(a, variant, s));
    return R2F_OK;
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

const char* r2f_version(void) { return "r2f-hip 0.6 gfx950 abi6"; }

int r2f_create(int device, r2f_ctx** out) {
    if (!out) return R2F_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return R2F_EHIP;
    DeviceGuard guard(device);  // the caller's current device is restored on return
    if (guard.status != hipSuccess) return R2F_EHIP;
    if (init_kernel_attributes() != hipSuccess || fft_init_attributes() != hipSuccess ||
        front_fast_init_attributes() != hipSuccess)
        return R2F_EHIP;
    r2f_ctx* ctx = new r2f_ctx();
    ctx->device = device;
    if (hipMalloc(&ctx->frame_buf.p, sizeof(FrameParams)) != hipSuccess || hipMemset(ctx->frame_buf.p, 0, sizeof(FrameParams)) != hipSuccess) {
        delete ctx;
        return R2F_EHIP;
    }
    ctx->frame_buf.bytes = sizeof(FrameParams);
    *out = ctx;
    return R2F_OK;
}

void r2f_destroy(r2f_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    (void)hipDeviceSynchronize();  // nothing in flight may still read what is freed below
    drop_render_graphs(ctx);
    reap_retired_graphs(ctx, true);  // (the device has been synchronised: entries without a usable event go too)
    if (ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
    ctx->frame_buf.release();
    ctx->range_tiles.release();
    ctx->dyn_flags.release();
    ctx->lut2d_buf.release();
    ctx->lut3d_buf.release();
    ctx->curve_buf.release();
    ctx->grain_lut_buf.release();
    for (auto& s : ctx->stencil)
        for (int c = 0; c < 3; ++c) {
            s.wbuf[c].release();
            s.mbuf[c].release();
        }
    for (auto& t : ctx->tile_order) t.buf.release();
    ctx->lanczos_buf.release();
    ctx->lanczos_f32_buf.release();
    ctx->grain_fixed_w.release();
    for (auto& b : ctx->stencil_fixed_w) b.release();
    ctx->fft_tw.release();
    ctx->fft_s1.release();
    ctx->fft_kimg.release();
    for (auto& row : ctx->fft_kf)
        for (auto& per_shape : row)
            for (auto& b : per_shape) b.release();
    for (int i = 0; i < 4; ++i) {
        if (ctx->fft_stream[i]) (void)hipStreamDestroy(ctx->fft_stream[i]);
        if (ctx->fft_ev_out[i]) (void)hipEventDestroy(ctx->fft_ev_out[i]);
    }
    if (ctx->fft_ev_in) (void)hipEventDestroy(ctx->fft_ev_in);
    delete ctx;
}

const char* r2f_last_error(const r2f_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

uint64_t r2f_generation(const r2f_ctx* ctx) { return ctx ? ctx->generation : 0; }

int r2f_set_option(r2f_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return R2F_EINVAL;
    ++ctx->generation;  // by-value launch arguments and the choice of kernels depend on the options
    if (!strcmp(name, "stencil_variant")) {
        if (value < -1 || value >= kNumStencilVariants) return fail(ctx, R2F_EINVAL, "stencil_variant out of range");
        ctx->opt_variant = value;
        return R2F_OK;
    }
    if (!strcmp(name, "render_graph")) {
        ctx->opt_render_graph = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_lds_kb")) {
        if (value < 8 || value > 160) return fail(ctx, R2F_EINVAL, "stencil_lds_kb must be in [8, 160]");
        ctx->opt_lds_kb = value;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_sym")) {
        ctx->opt_sym = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_ablate")) {
        ctx->opt_ablate = value;
        return R2F_OK;
    }
    if (!strcmp(name, "kernel_timing")) {
        ctx->opt_timing = value & 7;  // bit per pass: 1 rows forward, 2 columns, 4 rows inverse
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft")) {
        ctx->opt_fft = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_window")) {
        if (value != 0 && value != 256 && value != 512 && value != 1024) return fail(ctx, R2F_EINVAL, "stencil_fft_window must be 0, 256, 512 or 1024");
        ctx->opt_fft_window = value;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_window_max")) {
        if (value != 256 && value != 512 && value != 1024) return fail(ctx, R2F_EINVAL, "stencil_fft_window_max must be 256, 512 or 1024");
        ctx->opt_fft_window_max = value;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_even_batches")) {
        ctx->opt_fft_even = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fixed")) {
        ctx->opt_stencil_fixed = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "grain_fixed")) {
        ctx->opt_grain_fixed = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "front_fast")) {
        ctx->opt_front_fast = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "grain_separable")) {
        ctx->opt_grain_sep = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "front_blocks_per_cu")) {
        if (value < 1 || value > 64) return fail(ctx, R2F_EINVAL, "front_blocks_per_cu must be in [1, 64]");
        ctx->opt_front_blocks = value;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_window_rows")) {
        if (value != 0 && value != 256 && value != 512) return fail(ctx, R2F_EINVAL, "stencil_fft_window_rows must be 0, 256 or 512");
        ctx->opt_fft_window_rows = value;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_min_taps")) {
        ctx->opt_fft_min_taps = value;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_streams")) {
        if (value < 1 || value > 4) return fail(ctx, R2F_EINVAL, "stencil_fft_streams must be in [1, 4]");
        ctx->opt_fft_streams = value;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_cols_walk")) {
        ctx->opt_fft_cols_walk = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_mixed_sign")) {
        ctx->opt_fft_mixed_sign = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_real_spectrum")) {
        ctx->opt_fft_real = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_epilogue_lds")) {
        ctx->opt_fft_epi_lds = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_scratch96_auto")) {
        ctx->opt_fft_s96_auto = value ? 1 : 0;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_scratch96")) {
        if (value < 0 || value > 7) return fail(ctx, R2F_EINVAL, "stencil_fft_scratch96 is a mask over the three stencils (0..7)");
        ctx->opt_fft_s96 = value;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_scratch32")) {
        if (value < 0 || value > 7) return fail(ctx, R2F_EINVAL, "stencil_fft_scratch32 is a mask over the three stencils (0..7)");
        ctx->opt_fft_s32 = value;
        return R2F_OK;
    }
    if (!strcmp(name, "stencil_fft_batch")) {
        if (value < 1) return fail(ctx, R2F_EINVAL, "stencil_fft_batch must be >= 1");
        ctx->opt_fft_batch = value;
        return R2F_OK;
    }
    if (!strcmp(name, "xcd_band")) {
        if (value < 0) return fail(ctx, R2F_EINVAL, "xcd_band must be >= 0");
        ctx->opt_xcd_band = value;
        for (auto& t : ctx->tile_order) t.gx = t.gy = 0;  // rebuild the tables
        return R2F_OK;
    }
    if (!strcmp(name, "xcd_remap")) {
        if (value < 0 || value > 2) return fail(ctx, R2F_EINVAL, "xcd_remap must be 0, 1 or 2");
        ctx->opt_xcd_remap = value;
        return R2F_OK;
    }
    return fail(ctx, R2F_EINVAL, "unknown option %s", name);
}

int r2f_set_matrix3x3(r2f_ctx* ctx, const float* m) {
    if (!ctx) return R2F_EINVAL;
    ctx->has_matrix = m != nullptr;
    if (m) memcpy(ctx->mat.m, m, sizeof ctx->mat.m);
    ++ctx->generation;
    return R2F_OK;
}

int r2f_set_lut2d(r2f_ctx* ctx, const float* lut, int n) {
    if (!ctx) return R2F_EINVAL;
    if (!lut || n < 2) return fail(ctx, R2F_EINVAL, "lut2d: need (n, n, 3) with n >= 2");
    R2F_GUARD(ctx);
    std::vector<float4> tex((size_t)n * n);
    for (size_t i = 0; i < tex.size(); ++i) tex[i] = make_float4(lut[3 * i], lut[3 * i + 1], lut[3 * i + 2], 0.f);
    int rc = upload(ctx, ctx->lut2d_buf, tex.data(), tex.size() * sizeof(float4));
    if (rc) return rc;
    ctx->lut2d.tex = static_cast<const float4*>(ctx->lut2d_buf.p);
    ctx->lut2d.n = n;
    return R2F_OK;
}

int r2f_set_lut3d(r2f_ctx* ctx, const float* lut, int n) {
    if (!ctx) return R2F_EINVAL;
    if (!lut || n < 2 || n > 256) return fail(ctx, R2F_EINVAL, "lut3d: need (n, n, n, 3) with 2 <= n <= 256");
    R2F_GUARD(ctx);
    std::vector<float4> tex((size_t)n * n * n);
    for (size_t i = 0; i < tex.size(); ++i) tex[i] = make_float4(lut[3 * i], lut[3 * i + 1], lut[3 * i + 2], 0.f);
    int rc = upload(ctx, ctx->lut3d_buf, tex.data(), tex.size() * sizeof(float4));
    if (rc) return rc;
    ctx->lut3d.tex = static_cast<const float4*>(ctx->lut3d_buf.p);
    ctx->lut3d.n = n;
    return R2F_OK;
}

int r2f_set_curve1d(r2f_ctx* ctx, const float* lut4xm, int m) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    int rc = upload_curve(ctx, ctx->curve_buf, ctx->curve, lut4xm, m);
    if (rc) return rc;
    double smax = 0.0;
    for (int c = 1; c <= 3; ++c)
        for (int i = 0; i + 1 < m; ++i) {
            const double dx = (double)lut4xm[i + 1] - (double)lut4xm[i];
            if (dx > 0.0) smax = std::max(smax, std::fabs(((double)lut4xm[(size_t)c * m + i + 1] - (double)lut4xm[(size_t)c * m + i]) / dx));
        }
    ctx->curve_slope_max = (float)smax;
    return R2F_OK;
}

int r2f_set_grain_lut(r2f_ctx* ctx, const float* lut4xm, int m) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    return upload_curve(ctx, ctx->grain_lut_buf, ctx->grain_lut, lut4xm, m);
}

int r2f_set_kernel(r2f_ctx* ctx, int which, const float* k, int kh, int kw, int kc) {
    if (!ctx) return R2F_EINVAL;
    if (which < 0 || which > 2) return fail(ctx, R2F_EINVAL, "set_kernel: which must be 0..2");
    if (!k || kh < 1 || kw < 1 || (kc != 1 && kc != 3)) return fail(ctx, R2F_EINVAL, "set_kernel: need (kh, kw, 1|3)");
    R2F_GUARD(ctx);
    StencilSet& s = ctx->stencil[which];
    s.present = true;
    s.kh = kh;
    s.kw = kw;
    s.kc = kc;
    s.host.assign(k, k + (size_t)kh * kw * kc);
    for (int c = 0; c < 3; ++c) {
        bool pos = false, neg = false;
        double sum = 0.0;
        for (size_t i = 0; i < (size_t)kh * kw; ++i) {
            const float v = k[i * kc + (kc == 1 ? 0 : c)];
            pos = pos || v > 0.f, neg = neg || v < 0.f;
            sum += (double)v;
        }
        s.mixed_sign[c] = pos && neg;
        s.unit_gain[c] = !neg && sum >= 0.99;  // (NaN taps: false)
    }
    s.single_tap_mask = 0;
    for (int c = 0; c < 3; ++c) {
        float w;
        if (plan::single_tap_channel(plan::Taps{s.host.data(), kh, kw, kc}, c, &w)) s.single_tap_mask |= 1 << c;
    }
    ++ctx->generation;
    s.built_q = 0;
    for (int c = 0; c < 3; ++c) {
        ctx->fft_kf_dims[which][c] = 0;
        ctx->fft_last_real[which][c] = 0;
        for (bool& v : ctx->fft_kf_valid[which][c]) v = false;
    }
    if (which == R2F_KERNEL_GRAIN) ctx->grain_fixed_valid = false;
    ctx->stencil_fixed_valid[which] = false;
    return R2F_OK;
}

// ------------------------------------------------------------------------------- stages
static int stage_front_impl(r2f_ctx* ctx, const r2f_params* p, const void* in, int in_layout, int in_gy0, int in_rows, int upto,
                            const r2f_planes* dst, float* out_f32, uint8_t* out_u8, int out_gy0, int y0, int y1, int W,
                            int H_global, void* stream, const r2f_planes* finish_dst, int* finished_mask, bool* tracked = nullptr);

int r2f_stage_front(r2f_ctx* ctx, const r2f_params* p, const void* in, int in_layout, int in_gy0, int in_rows, int upto,
                    const r2f_planes* dst, float* out_f32, uint8_t* out_u8, int out_gy0, int y0, int y1, int W,
                    int H_global, void* stream) {
    return stage_front_impl(ctx, p, in, in_layout, in_gy0, in_rows, upto, dst, out_f32, out_u8, out_gy0, y0, y1, W, H_global, stream,
                            nullptr, nullptr);
}

int r2f_stage_front_split(r2f_ctx* ctx, const r2f_params* p, const void* in, int in_layout, int in_gy0, int in_rows,
                          const r2f_planes* exposure, const r2f_planes* density, int y0, int y1, int W, int H_global,
                          int* finished_mask, void* stream) {
    if (!finished_mask) return R2F_EINVAL;
    *finished_mask = 0;
    return stage_front_impl(ctx, p, in, in_layout, in_gy0, in_rows, R2F_UPTO_EXPOSURE, exposure, nullptr, nullptr, 0, y0, y1, W, H_global,
                            stream, density, finished_mask);
}

// tracked (whole-frame renders): when given, the fast kernel records the range of the exposure planes it writes in the context's frame
// block and *tracked says whether that happened (only the split fast kernel does it).
static int stage_front_impl(r2f_ctx* ctx, const r2f_params* p, const void* in, int in_layout, int in_gy0, int in_rows, int upto,
                            const r2f_planes* dst, float* out_f32, uint8_t* out_u8, int out_gy0, int y0, int y1, int W,
                            int H_global, void* stream, const r2f_planes* finish_dst, int* finished_mask, bool* tracked) {
    if (tracked) *tracked = false;
    if (!ctx || !p) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (y1 <= y0) return R2F_OK;
    // R2F_F_TRACK_RANGE (a row shard's front calls): record like a whole-frame render's front kernel, or say that it did not happen
    const bool want_track = upto == R2F_UPTO_EXPOSURE && (tracked || (p->flags & R2F_F_TRACK_RANGE)) && ctx->opt_fft_s96_auto;
    auto cannot_track = [&]() -> int {
        if (!(p->flags & R2F_F_TRACK_RANGE) || upto != R2F_UPTO_EXPOSURE) return R2F_OK;
        return write_frame_params(ctx, p, static_cast<hipStream_t>(stream), 3);
    };
    if (!in || W <= 0 || y0 < in_gy0 || y1 > in_gy0 + in_rows || in_layout < 0 || in_layout > 2)
        return fail(ctx, R2F_EINVAL, "front: bad input geometry");
    if (!ctx->lut2d.tex) return fail(ctx, R2F_EINVAL, "input LUT not set (r2f_set_lut2d)");
    if ((p->flags & R2F_F_MATRIX) && !ctx->has_matrix) return fail(ctx, R2F_EINVAL, "matrix not set (r2f_set_matrix3x3)");
    FrontArgs a;
    memset(&a, 0, sizeof a);
    a.in = in;
    a.in_layout = in_layout;
    a.in_gy0 = in_gy0;
    a.in_rows = in_rows;
    a.upto = upto;
    a.y0 = y0;
    a.y1 = y1;
    a.W = W;
    a.H_global = H_global;
    a.use_matrix = (p->flags & R2F_F_MATRIX) ? 1 : 0;
    a.mat = ctx->mat;
    a.lut2d = ctx->lut2d;
    a.curve = ctx->curve;
    a.lut3d = ctx->lut3d;
    a.log_eps = p->log_eps;
    a.lut3d_scale = p->lut3d_scale;
    a.lut3d_mode = p->lut3d_mode;
    bool vec = W % 4 == 0 && aligned16(in);
    if (upto >= R2F_UPTO_DENSITY && !ctx->curve.cells) return fail(ctx, R2F_EINVAL, "density curve not set (r2f_set_curve1d)");
    if (upto == R2F_UPTO_OUTPUT) {
        if (!ctx->lut3d.tex) return fail(ctx, R2F_EINVAL, "output LUT not set (r2f_set_lut3d)");
        if (!out_f32 && !out_u8) return fail(ctx, R2F_EINVAL, "front: no output buffer");
        if (y0 < out_gy0) return fail(ctx, R2F_EINVAL, "front: y0 above the output buffer");
        a.out_f32 = out_f32;
        a.out_u8 = out_u8;
        a.out_gy0 = out_gy0;
        vec = vec && (!out_f32 || aligned16(out_f32)) && (!out_u8 || (reinterpret_cast<uintptr_t>(out_u8) & 3u) == 0);
    } else if (upto == R2F_UPTO_EXPOSURE || upto == R2F_UPTO_DENSITY) {
        int rc = check_rows(ctx, "front dst", dst, y0, y1);
        if (rc) return rc;
        a.dst = to_dev(dst);
        vec = vec && planes_vec_ok(dst, W);
    } else {
        return fail(ctx, R2F_EINVAL, "front: bad upto");
    }
    a.vec = vec ? 1 : 0;
    a.blocks_per_cu = ctx->opt_front_blocks;
    a.fast = ctx->opt_front_fast;
    if (finish_dst && upto == R2F_UPTO_EXPOSURE && a.fast && ctx->stencil[R2F_KERNEL_HALATION].present && ctx->curve.cells) {
        // channels the halation leaves to a single tap: finish them here when the fast kernel can take the job
        int rc = check_rows(ctx, "front density dst", finish_dst, y0, y1);
        if (rc) return rc;
        FrontArgs f = a;
        f.finish_dst = to_dev(finish_dst);
        for (int c = 0; c < 3; ++c)
            if (single_tap_channel(ctx->stencil[R2F_KERNEL_HALATION], c, &f.finish_w[c])) f.finish_mask |= 1 << c;
        f.vec = (vec && planes_vec_ok(finish_dst, W)) ? 1 : 0;
        if (f.finish_mask && f.finish_mask != 7 && front_fast_eligible(f)) {
            *finished_mask = f.finish_mask;
            if (want_track) {  // the exposure planes' range for the FFT passes
                rc = ensure_range_tiles(ctx, H_global, W);
                if (rc) return rc;
                f.track = record_of(ctx);
                f.track_mask = 7 & ~f.finish_mask;
                if (tracked) *tracked = true;
            }
            R2F_HIP(ctx, launch_front_fast(f, static_cast<hipStream_t>(stream)));
            return R2F_OK;
        }
    }
    if (want_track && !tracked && a.fast && front_fast_eligible(a)) {
        // a row shard writes every channel's exposure (its neighbours need them); the record covers the channels the halation's FFT
        // passes read, i.e. not the single-tap ones -- the same samples a whole-frame render records
        int rc = ensure_range_tiles(ctx, H_global, W);
        if (rc) return rc;
        a.track = record_of(ctx);
        a.track_mask = 7;
        if (ctx->stencil[R2F_KERNEL_HALATION].present) a.track_mask &= ~ctx->stencil[R2F_KERNEL_HALATION].single_tap_mask;
    } else {
        int rc = cannot_track();
        if (rc) return rc;
    }
    R2F_HIP(ctx, launch_front(a, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_stage_halation(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* exposure, const r2f_planes* density, int y0,
                       int y1, int W, int H_global, void* stream) {
    if (!ctx || !p) return R2F_EINVAL;
    R2F_GUARD(ctx);
    ctx->frame_dyn_armed = false;  // (set again by the FFT launches when they carry the rule: r2f_frame_exposure_range)
    return run_stencil(ctx, R2F_KERNEL_HALATION, exposure, density, y0, y1, W, H_global, 1, p->log_eps,
                       static_cast<hipStream_t>(stream), (p->flags & R2F_F_IDENTITY_DONE) != 0, (p->flags & R2F_F_RANGE_VALID) != 0);
}

int r2f_stage_exposure_range(r2f_ctx* ctx, const r2f_planes* exposure, int y0, int y1, int y2, int y3, int W, void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (y1 <= y0 && y3 <= y2) return R2F_OK;
    if (W <= 0) return fail(ctx, R2F_EINVAL, "exposure range: bad geometry");
    int rc = y1 > y0 ? check_rows(ctx, "exposure range", exposure, y0, y1) : R2F_OK;
    if (rc) return rc;
    rc = y3 > y2 ? check_rows(ctx, "exposure range", exposure, y2, y3) : R2F_OK;
    if (rc) return rc;
    int mask = 7;  // the channels the halation's FFT passes read: not the single-tap ones (as the front kernel records them)
    if (ctx->stencil[R2F_KERNEL_HALATION].present) mask &= ~ctx->stencil[R2F_KERNEL_HALATION].single_tap_mask;
    rc = ensure_range_tiles(ctx, std::max(std::max(y1, y3), exposure->gy0 + exposure->rows), W);
    if (rc) return rc;
    R2F_HIP(ctx, launch_exposure_range(to_dev(exposure), y0, y1, y2, y3, W, mask, record_of(ctx), static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_stage_mtf(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* din, const r2f_planes* dout, int y0, int y1,
                  int W, int H_global, void* stream) {
    if (!ctx || !p) return R2F_EINVAL;
    R2F_GUARD(ctx);
    return run_stencil(ctx, R2F_KERNEL_MTF, din, dout, y0, y1, W, H_global, 0, 0.f, static_cast<hipStream_t>(stream));
}

int r2f_stage_stencil(r2f_ctx* ctx, int which, const r2f_planes* src, const r2f_planes* dst, int y0, int y1, int W,
                      int H_global, void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (which < 0 || which > 2) return fail(ctx, R2F_EINVAL, "stencil: which must be 0..2");
    return run_stencil(ctx, which, src, dst, y0, y1, W, H_global, 0, 0.f, static_cast<hipStream_t>(stream));
}

int r2f_kernel_timing(r2f_ctx* ctx, int cls, double* total_ms, int* launches, double* bytes) {
    if (!ctx || cls < 0 || cls > 5 || !total_ms || !launches || !bytes) return R2F_EINVAL;
    R2F_GUARD(ctx);
    double sum = 0.0;
    for (auto& ev : ctx->timing_ev[cls]) {
        R2F_HIP(ctx, hipEventSynchronize(ev.second));
        float ms = 0.f;
        R2F_HIP(ctx, hipEventElapsedTime(&ms, ev.first, ev.second));
        sum += ms;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    *total_ms = sum;
    *launches = (int)ctx->timing_ev[cls].size();
    *bytes = ctx->timing_bytes[cls];
    ctx->timing_ev[cls].clear();
    ctx->timing_bytes[cls] = 0.0;
    return R2F_OK;
}

int r2f_stencil_stats(r2f_ctx* ctx, int which, int* out) {
    if (!ctx || !out) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (which < 0 || which > 2) return fail(ctx, R2F_EINVAL, "stencil: which must be 0..2");
    StencilSet& set = ctx->stencil[which];
    if (!set.present) return fail(ctx, R2F_EINVAL, "stencil %d not set (r2f_set_kernel)", which);
    if (!set.built_q) {  // not launched yet: build the device form the default launch would use
        const StencilVariant& sv = kStencilVariants[ctx->opt_variant >= 0 ? ctx->opt_variant : 0];
        int rc = ensure_stencil(ctx, which, sv.Q, sv.TW(), sv.TH(), (size_t)ctx->opt_lds_kb * 1024, which == 2);
        if (rc) return rc;
    }
    for (int c = 0; c < 3; ++c) {
        const DevStencil& d = set.dev[c];
        int* o = out + 8 * c;
        // bit 0: mirrored taps are paired in the entry list; bits 1..: R when the channel takes the unrolled stencil_fixed form
        // (the grain stencil: all three channels together, and the geometry of the tail tile, which this call may not have built)
        const int all[3] = {0, 1, 2};
        const int fr = which == R2F_KERNEL_GRAIN ? (ctx->opt_grain_fixed ? fixed_stencil_radius(set, all, 3, 9) : 0)
                                                 : (ctx->opt_stencil_fixed && ctx->opt_variant <= 0 && !fft_eligible(ctx, set, c)
                                                        ? fixed_stencil_radius(set, &c, 1, kFixedMaxR)
                                                        : 0);
        // bit 8: the grain stencil runs as two 1-D passes (known once a tail launch has looked at the taps)
        const int sep = which == R2F_KERNEL_GRAIN && fr && ctx->grain_fixed_valid && ctx->grain_sep && ctx->opt_grain_sep;
        o[0] = d.n_entries, o[1] = d.n_rowsteps, o[2] = d.n_phases, o[3] = d.sym | (fr << 1) | (sep << 8);
        o[4] = d.kh, o[5] = d.kw, o[6] = set.built_q;
        o[7] = fft_eligible(ctx, set, c) ? 1 | (ctx->fft_kf_dims[which][c] << 1) | (ctx->fft_last_real[which][c] << 30) : 0;
    }
    return R2F_OK;
}

static bool burn_geometry(const r2f_params* p, int H, int W, int* h_lo, int* w_lo) { return plan::burn_geometry(p->burn_cell, H, W, h_lo, w_lo); }

// density == nullptr && planes_out: the grain field alone (K_g * noise) -> planes_out.
// gfield: a grain field computed that way is applied pointwise instead of being generated here.
// The grain stencil in the form stencil_fixed<R, 2> wants, when it has one (fixed_stencil_radius); grain_fixed_r = 0
// otherwise (the generic entry list runs).
static int ensure_grain_fixed(r2f_ctx* ctx) {
    if (ctx->grain_fixed_valid) return R2F_OK;
    StencilSet& set = ctx->stencil[R2F_KERNEL_GRAIN];
    ctx->grain_fixed_valid = true;
    const int chans[3] = {0, 1, 2};
    ctx->grain_fixed_r = fixed_stencil_radius(set, chans, 3, 9);
    ctx->grain_sep = false;
    if (!ctx->grain_fixed_r) return R2F_OK;
    bool same = false;
    const std::vector<float> w = fixed_stencil_weights(set, ctx->grain_fixed_r, kTailQ, &same);
    ctx->grain_fixed_same = same ? 1 : 0;
    // separable (K = u v^T to 6e-7 of the largest tap: plan::separable_taps)?  Then two 1-D passes of 2 R + 1 taps replace (2 R + 1)^2
    ctx->grain_sep = plan::separable_taps(taps_of(set), ctx->grain_fixed_r, ctx->grain_sep_u, ctx->grain_sep_v);
    return upload(ctx, ctx->grain_fixed_w, w.data(), w.size() * sizeof(float));
}

static int run_tail(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* density, const r2f_planes* planes_out,
                    const float* burn_map, float* out_f32, uint8_t* out_u8, int out_gy0, int y0, int y1, int W, int H_global,
                    void* stream, const r2f_planes* gfield = nullptr) {
    if (y1 <= y0) return R2F_OK;
    const bool field_only = density == nullptr && planes_out != nullptr;
    if (field_only) {
        static const r2f_planes none = {nullptr, 0, 0, 0};
        density = &none;
    }
    const bool to_planes = planes_out != nullptr;
    if (W <= 0 || y0 < 0 || y1 > H_global || (!to_planes && y0 < out_gy0)) return fail(ctx, R2F_EINVAL, "tail: bad geometry");
    if (!to_planes && !out_f32 && !out_u8) return fail(ctx, R2F_EINVAL, "tail: no output buffer");
    if (!to_planes && !ctx->lut3d.tex) return fail(ctx, R2F_EINVAL, "output LUT not set (r2f_set_lut3d)");
    int rc = field_only ? R2F_OK : check_rows(ctx, "tail src", density, y0, y1);
    if (rc) return rc;
    TailArgs a;
    memset(&a, 0, sizeof a);
    a.src = to_dev(density);
    a.out_f32 = out_f32;
    a.out_u8 = out_u8;
    a.out_gy0 = out_gy0;
    a.y0 = y0;
    a.y1 = y1;
    a.W = W;
    a.H_global = H_global;
    a.grain = (p->flags & R2F_F_GRAIN) ? 1 : 0;
    if (gfield) {  // the field was made ahead of time: this call is the pointwise half
        if (!a.grain) return fail(ctx, R2F_EINVAL, "tail: a grain field was passed but the grain flag is off");
        rc = check_rows(ctx, "grain field", gfield, y0, y1);
        if (rc) return rc;
        if (!ctx->grain_lut.cells) return fail(ctx, R2F_EINVAL, "grain LUT not set (r2f_set_grain_lut)");
        a.grain = 0;
        a.gfield = to_dev(gfield);
        a.has_gfield = 1;
        a.grain_lut = ctx->grain_lut;
    }
    a.mono = (p->flags & R2F_F_GRAIN_MONO) ? 1 : 0;
    a.frame = static_cast<const FrameParams*>(ctx->frame_buf.p);
    a.lut3d = ctx->lut3d;
    a.lut3d_scale = p->lut3d_scale;
    a.lut3d_mode = p->lut3d_mode;
    bool vec = field_only ? true : planes_vec_ok(density, W);
    if (gfield) vec = vec && planes_vec_ok(gfield, W);
    if (to_planes) {
        if (!a.grain) return fail(ctx, R2F_EINVAL, "grain stage called with the grain flag off");
        rc = check_rows(ctx, "grain dst", planes_out, y0, y1);
        if (rc) return rc;
        // pointwise on the density, so exactly in place (same base, stride and first row) is fine; anything else that
        // overlaps would have one pixel's store land on another pixel's load
        if (!field_only && planes_overlap(density, planes_out, W) &&
            !(density->data == planes_out->data && density->plane_stride == planes_out->plane_stride && density->gy0 == planes_out->gy0))
            return fail(ctx, R2F_EINVAL, "grain: source and destination planes overlap without being the same buffer");
        a.to_planes = field_only ? 2 : 1;
        a.dst = to_dev(planes_out);
        vec = vec && planes_vec_ok(planes_out, W);
    } else {
        vec = vec && (!out_f32 || aligned16(out_f32)) && (!out_u8 || (reinterpret_cast<uintptr_t>(out_u8) & 3u) == 0);
        if (burn_map) {
            if (a.grain || a.has_gfield)
                return fail(ctx, R2F_EINVAL, "tail with a burn map: apply the grain first (r2f_stage_grain) and clear R2F_F_GRAIN");
            int h_lo, w_lo;
            if (!burn_geometry(p, H_global, W, &h_lo, &w_lo)) return fail(ctx, R2F_EINVAL, "burn: bad burn_cell");
            a.burn.map = burn_map;
            a.burn.h_lo = h_lo;
            a.burn.w_lo = w_lo;
            a.burn.h_up = h_lo * p->burn_cell;
            a.burn.w_up = w_lo * p->burn_cell;
            a.burn.zy = a.burn.h_up > 1 ? (double)(h_lo - 1) / (double)(a.burn.h_up - 1) : 0.0;
            a.burn.zx = a.burn.w_up > 1 ? (double)(w_lo - 1) / (double)(a.burn.w_up - 1) : 0.0;
            a.burn.strength = p->burn_strength;
        }
    }
    a.vec = vec ? 1 : 0;
    if (a.grain) {
        if (!ctx->grain_lut.cells) return fail(ctx, R2F_EINVAL, "grain LUT not set (r2f_set_grain_lut)");
        if (!ctx->stencil[R2F_KERNEL_GRAIN].present) {
            // gpu_processor.py:931-932: no grain kernel -> 1x1 ones
            const float one = 1.f;
            rc = r2f_set_kernel(ctx, R2F_KERNEL_GRAIN, &one, 1, 1, 1);
            if (rc) return rc;
        }
        rc = ensure_stencil(ctx, R2F_KERNEL_GRAIN, kTailQ, 4 * kTailBX, kTailQ * kTailBY, 0, true);
        if (rc) return rc;
        for (int c = 0; c < 3; ++c) a.gk[c] = ctx->stencil[R2F_KERNEL_GRAIN].dev[c];
        rc = ensure_grain_fixed(ctx);
        if (rc) return rc;
        a.fixed_r = ctx->opt_grain_fixed ? ctx->grain_fixed_r : 0;
        a.fixed_same = ctx->grain_fixed_same;
        a.fixed_w = static_cast<const float*>(ctx->grain_fixed_w.p);
        // (monochrome noise with per-channel taps: the one noise plane cannot be filtered in place three ways -> 2-D form)
        a.sep = (a.fixed_r && ctx->opt_grain_sep && ctx->grain_sep && (!a.mono || ctx->grain_fixed_same)) ? 1 : 0;
        memcpy(a.sep_u, ctx->grain_sep_u, sizeof a.sep_u);
        memcpy(a.sep_v, ctx->grain_sep_v, sizeof a.sep_v);
        if (tail_lds_bytes(a.gk, a.mono) > kMaxLds)
            return fail(ctx, R2F_ETOOLARGE, "grain stencil %dx%d does not fit the LDS noise tile", a.gk[0].kh, a.gk[0].kw);
        a.grain_lut = ctx->grain_lut;
    }
    if (a.grain && !(p->flags & R2F_F_FRAME_RESIDENT)) {  // the seed of THIS call, ahead of the kernel that reads it
        rc = write_frame_params(ctx, p, static_cast<hipStream_t>(stream), 0);
        if (rc) return rc;
    }
    R2F_HIP(ctx, launch_tail(a, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_stage_tail(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* density, const float* burn_map, float* out_f32,
                   uint8_t* out_u8, int out_gy0, int y0, int y1, int W, int H_global, void* stream) {
    if (!ctx || !p) return R2F_EINVAL;
    R2F_GUARD(ctx);
    return run_tail(ctx, p, density, nullptr, burn_map, out_f32, out_u8, out_gy0, y0, y1, W, H_global, stream);
}

int r2f_stage_grain_field(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* field, int y0, int y1, int W, int H_global,
                          void* stream) {
    if (!ctx || !p) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!field) return fail(ctx, R2F_EINVAL, "grain field: null destination");
    return run_tail(ctx, p, nullptr, field, nullptr, nullptr, nullptr, 0, y0, y1, W, H_global, stream);
}

int r2f_stage_tail_field(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* density, const r2f_planes* field, float* out_f32,
                         uint8_t* out_u8, int out_gy0, int y0, int y1, int W, int H_global, void* stream) {
    if (!ctx || !p) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!field) return fail(ctx, R2F_EINVAL, "tail: null grain field");
    return run_tail(ctx, p, density, nullptr, nullptr, out_f32, out_u8, out_gy0, y0, y1, W, H_global, stream, field);
}

int r2f_stage_grain(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* din, const r2f_planes* dout, int y0, int y1, int W,
                    int H_global, void* stream) {
    if (!ctx || !p) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!dout) return fail(ctx, R2F_EINVAL, "grain: null destination");
    return run_tail(ctx, p, din, dout, nullptr, nullptr, nullptr, 0, y0, y1, W, H_global, stream);
}

int r2f_stage_burn_sums(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* density, float* cell_sums, int y0, int y1, int W,
                        int H_global, void* stream) {
    if (!ctx || !p) return R2F_EINVAL;
    R2F_GUARD(ctx);
    int h_lo, w_lo;
    if (!cell_sums || !burn_geometry(p, H_global, W, &h_lo, &w_lo)) return fail(ctx, R2F_EINVAL, "burn sums: bad arguments");
    if (y0 < 0 || y1 > H_global || y1 < y0) return fail(ctx, R2F_EINVAL, "burn sums: bad rows");
    if (y1 > y0) {
        int rc = check_rows(ctx, "burn src", density, y0, y1);
        if (rc) return rc;
    }
    BurnSumsArgs a;
    a.src = to_dev(density);
    a.cell_sums = cell_sums;
    a.y0 = y0;
    a.y1 = y1;
    a.W = W;
    a.H_global = H_global;
    a.h_lo = h_lo;
    a.w_lo = w_lo;
    R2F_HIP(ctx, launch_burn_sums(a, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_stage_burn_map(r2f_ctx* ctx, const r2f_params* p, const float* cell_sums, float* burn_map, float* scratch, int W,
                       int H_global, void* stream) {
    if (!ctx || !p) return R2F_EINVAL;
    R2F_GUARD(ctx);
    int h_lo, w_lo;
    if (!cell_sums || !burn_map || !scratch || !burn_geometry(p, H_global, W, &h_lo, &w_lo))
        return fail(ctx, R2F_EINVAL, "burn map: bad arguments");
    BurnMapArgs a;
    a.cell_sums = cell_sums;
    a.map = burn_map;
    a.scratch = scratch;
    a.h_lo = h_lo;
    a.w_lo = w_lo;
    a.d_ref = p->burn_d_ref;
    plan::burn_weights(a.w);  // scipy's gaussian kernel for sigma = 3, truncate = 2
    R2F_HIP(ctx, launch_burn_map(a, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_resize_area(r2f_ctx* ctx, const void* in, int in_layout, int H, int W, const r2f_planes* dst, int out_h, int out_w,
                    void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!in || in_layout < 0 || in_layout > 2 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0 || out_h > H || out_w > W)
        return fail(ctx, R2F_EINVAL, "resize_area: the target must be a non-empty frame no larger than the source");
    int rc = check_rows(ctx, "resize dst", dst, 0, out_h);
    if (rc) return rc;
    ResizeArgs a;
    a.in = in;
    a.in_layout = in_layout;
    a.H = H;
    a.W = W;
    a.dst = to_dev(dst);
    a.out_h = out_h;
    a.out_w = out_w;
    R2F_HIP(ctx, launch_resize_area(a, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_warp_affine(r2f_ctx* ctx, const void* in, int in_layout, int H, int W, const double* m_dst_to_src, const r2f_planes* dst,
                    int out_h, int out_w, int oy, int ox, void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!in || !m_dst_to_src || in_layout < 0 || in_layout > 2 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0)
        return fail(ctx, R2F_EINVAL, "warp_affine: bad arguments");
    int rc = check_rows(ctx, "warp dst", dst, 0, out_h);
    if (rc) return rc;
    WarpArgs a;
    a.in = in;
    a.in_layout = in_layout;
    a.H = H;
    a.W = W;
    a.dst = to_dev(dst);
    a.out_h = out_h;
    a.out_w = out_w;
    a.oy = oy;
    a.ox = ox;
    for (int i = 0; i < 6; ++i) a.m[i] = (float)m_dst_to_src[i];
    R2F_HIP(ctx, launch_warp_affine(a, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_resize_lanczos4_f32(r2f_ctx* ctx, const void* in, int in_layout, int H, int W, const r2f_planes* dst, int out_h, int out_w,
                            void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!in || in_layout < 0 || in_layout > 2 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0)
        return fail(ctx, R2F_EINVAL, "resize_lanczos4_f32: bad arguments");
    int rc = check_rows(ctx, "lanczos dst", dst, 0, out_h);
    if (rc) return rc;
    const size_t n_ofs = (size_t)out_w + out_h, n_coef = 8 * n_ofs;
    const size_t coef_off = (n_ofs * sizeof(int) + 15) / 16 * 16;
    std::vector<unsigned char> host(coef_off + n_coef * sizeof(float));
    int* ofs = reinterpret_cast<int*>(host.data());
    float* coef = reinterpret_cast<float*>(host.data() + coef_off);
    r2f_lanczos4_table_f32(W, out_w, ofs, coef);
    r2f_lanczos4_table_f32(H, out_h, ofs + out_w, coef + 8 * (size_t)out_w);
    rc = upload(ctx, ctx->lanczos_f32_buf, host.data(), host.size());  // (waits for renders in flight, like every table upload)
    if (rc) return rc;
    const unsigned char* base = static_cast<const unsigned char*>(ctx->lanczos_f32_buf.p);
    const int* xofs = reinterpret_cast<const int*>(base);
    const float* xcoef = reinterpret_cast<const float*>(base + coef_off);
    R2F_HIP(ctx, launch_lanczos4_f32(in, in_layout, H, W, to_dev(dst), out_h, out_w, xofs, xcoef, xofs + out_w, xcoef + 8 * (size_t)out_w,
                                     static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_resize_lanczos4_u8(r2f_ctx* ctx, const uint8_t* src_hwc, int H, int W, uint8_t* dst_hwc, int out_h, int out_w, void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!src_hwc || !dst_hwc || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0)
        return fail(ctx, R2F_EINVAL, "resize_lanczos4: bad arguments");
    const size_t n_ofs = (size_t)out_w + out_h, n_coef = 8 * n_ofs;
    const size_t coef_off = (n_ofs * sizeof(int) + 15) / 16 * 16;
    const int key[4] = {H, W, out_h, out_w};
    if (memcmp(key, ctx->lanczos_key, sizeof key) != 0 || !ctx->lanczos_buf.p) {
        std::vector<unsigned char> host(coef_off + n_coef * sizeof(short));
        int* ofs = reinterpret_cast<int*>(host.data());
        short* coef = reinterpret_cast<short*>(host.data() + coef_off);
        r2f_lanczos4_table(W, out_w, ofs, coef);
        r2f_lanczos4_table(H, out_h, ofs + out_w, coef + 8 * (size_t)out_w);
        int rc = upload(ctx, ctx->lanczos_buf, host.data(), host.size());
        if (rc) return rc;
        memcpy(ctx->lanczos_key, key, sizeof key);
    }
    LanczosArgs a;
    a.src = src_hwc;
    a.dst = dst_hwc;
    a.H = H, a.W = W, a.out_h = out_h, a.out_w = out_w;
    const unsigned char* base = static_cast<const unsigned char*>(ctx->lanczos_buf.p);
    a.xofs = reinterpret_cast<const int*>(base);
    a.yofs = a.xofs + out_w;
    a.xcoef = reinterpret_cast<const short*>(base + coef_off);
    a.ycoef = a.xcoef + 8 * (size_t)out_w;
    R2F_HIP(ctx, launch_lanczos4_u8(a, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_resize_area_u8(r2f_ctx* ctx, const uint8_t* src_hwc, int H, int W, uint8_t* dst_hwc, int out_h, int out_w, void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!src_hwc || !dst_hwc || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0 || out_h > H || out_w > W)
        return fail(ctx, R2F_EINVAL, "resize_area_u8: the target must be a non-empty frame no larger than the source");
    R2F_HIP(ctx, launch_resize_area_u8(src_hwc, H, W, dst_hwc, out_h, out_w, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_decode_u16(r2f_ctx* ctx, const uint16_t* src_hwc, int H, int W, int channels, float divisor, float factor, float* dst_f32_hwc3,
                   void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!src_hwc || !dst_f32_hwc3 || H <= 0 || W <= 0 || (channels != 3 && channels != 4) || !(divisor > 0.f))
        return fail(ctx, R2F_EINVAL, "decode_u16: a non-empty 3- or 4-channel frame and a positive divisor are required");
    R2F_HIP(ctx, launch_decode_u16(src_hwc, (long long)H * W, channels, divisor, factor, dst_f32_hwc3, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_stream_copy(r2f_ctx* ctx, const void* src, void* dst, size_t bytes, void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!src || !dst || bytes % 16 != 0 || !aligned16(src) || !aligned16(dst))
        return fail(ctx, R2F_EINVAL, "stream_copy: 16-byte aligned buffers and a multiple of 16 bytes are required");
    R2F_HIP(ctx, launch_stream_copy(src, dst, (long long)bytes, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_blit_rgba8(r2f_ctx* ctx, const float* src_f32_hwc, int H, int W, uint8_t* dst_rgba, int dst_h, int dst_w, const r2f_blit* t,
                   void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!src_f32_hwc || !dst_rgba || !t || H <= 0 || W <= 0 || dst_h <= 0 || dst_w <= 0)
        return fail(ctx, R2F_EINVAL, "blit: bad arguments");
    if (reinterpret_cast<uintptr_t>(dst_rgba) & 3u) return fail(ctx, R2F_EINVAL, "blit: the destination must be 4-byte aligned");
    R2F_HIP(ctx, launch_blit_rgba8(src_f32_hwc, H, W, dst_rgba, dst_h, dst_w, *t, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_histogram_render(r2f_ctx* ctx, const uint32_t* counts, const uint8_t* mix_table_rgba, int height, uint8_t* image_rgba,
                         uint8_t* target_rgba, int target_h, int target_w, void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!counts || !mix_table_rgba || !image_rgba || height <= 0 || (target_rgba && (target_h <= 0 || target_w <= 0)))
        return fail(ctx, R2F_EINVAL, "histogram_render: bad arguments");
    if ((reinterpret_cast<uintptr_t>(image_rgba) & 3u) || (reinterpret_cast<uintptr_t>(target_rgba) & 3u))
        return fail(ctx, R2F_EINVAL, "histogram_render: images must be 4-byte aligned");
    R2F_HIP(ctx, launch_histogram_render(counts, mix_table_rgba, height, image_rgba, target_rgba, target_h, target_w,
                                         static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

static int chroma_weights(r2f_ctx* ctx, int size, ChromaArgs& a) {
    static_assert(kChromaMaxTaps == plan::kChromaMaxTaps, "one tap limit");
    if (!plan::chroma_weights(size, a.w)) return fail(ctx, R2F_EINVAL, "chroma_nr size must be in [1, %d]", (kChromaMaxTaps - 1) / 2);
    a.radius = size;
    return R2F_OK;
}

int r2f_stage_chroma_nr_h(r2f_ctx* ctx, const void* in, int in_layout, int in_gy0, int in_rows, const r2f_planes* dst, int size,
                          int y0, int y1, int W, void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (y1 <= y0) return R2F_OK;
    if (!in || W <= 0 || y0 < in_gy0 || y1 > in_gy0 + in_rows || in_layout < 0 || in_layout > 2)
        return fail(ctx, R2F_EINVAL, "chroma_nr: bad input geometry");
    ChromaArgs a;
    memset(&a, 0, sizeof a);
    int rc = chroma_weights(ctx, size, a);
    if (rc) return rc;
    rc = check_rows(ctx, "chroma_nr dst", dst, y0, y1);
    if (rc) return rc;
    a.in = in;
    a.in_layout = in_layout;
    a.in_gy0 = in_gy0;
    a.in_rows = in_rows;
    a.dst = to_dev(dst);
    a.y0 = y0;
    a.y1 = y1;
    a.W = W;
    a.H_global = in_gy0 + in_rows;
    a.vec = planes_vec_ok(dst, W) ? 1 : 0;
    R2F_HIP(ctx, launch_chroma_h(a, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_stage_chroma_nr_v(r2f_ctx* ctx, const r2f_planes* src, const r2f_planes* dst, int size, int y0, int y1, int W,
                          int H_global, void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (y1 <= y0) return R2F_OK;
    if (W <= 0 || y0 < 0 || y1 > H_global) return fail(ctx, R2F_EINVAL, "chroma_nr: bad geometry");
    ChromaArgs a;
    memset(&a, 0, sizeof a);
    int rc = chroma_weights(ctx, size, a);
    if (rc) return rc;
    rc = check_rows(ctx, "chroma_nr dst", dst, y0, y1);
    if (rc) return rc;
    rc = check_rows(ctx, "chroma_nr src", src, std::max(y0 - size, 0), std::min(y1 + size, H_global));
    if (rc) return rc;
    if (planes_overlap(src, dst, W)) return fail(ctx, R2F_EINVAL, "chroma_nr: source and destination planes overlap (out of place only)");
    a.src = to_dev(src);
    a.dst = to_dev(dst);
    a.y0 = y0;
    a.y1 = y1;
    a.W = W;
    a.H_global = H_global;
    a.vec = (planes_vec_ok(src, W) && planes_vec_ok(dst, W)) ? 1 : 0;
    R2F_HIP(ctx, launch_chroma_v(a, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_stage_noise(r2f_ctx* ctx, const r2f_params* p, uint32_t* hash_planes, float* noise_planes, int y0, int y1, int W,
                    void* stream) {
    if (!ctx || !p) return R2F_EINVAL;
    R2F_GUARD(ctx);
    NoiseArgs a;
    a.hash = hash_planes;
    a.noise = noise_planes;
    a.y0 = y0;
    a.y1 = y1;
    a.W = W;
    a.frame = static_cast<const FrameParams*>(ctx->frame_buf.p);
    a.mono = (p->flags & R2F_F_GRAIN_MONO) ? 1 : 0;
    if (!(p->flags & R2F_F_FRAME_RESIDENT)) {
        int rc = write_frame_params(ctx, p, static_cast<hipStream_t>(stream), 0);
        if (rc) return rc;
    }
    R2F_HIP(ctx, launch_noise(a, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

int r2f_histogram_u8(r2f_ctx* ctx, const uint8_t* image_hwc, int H, int W, uint32_t* counts, void* stream) {
    if (!ctx) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (!counts || H < 0 || W < 0 || (!image_hwc && H > 0 && W > 0)) return fail(ctx, R2F_EINVAL, "histogram: bad arguments");
    if (!aligned16(image_hwc)) return fail(ctx, R2F_EINVAL, "histogram: image must be 16-byte aligned");
    R2F_HIP(ctx, launch_histogram_u8(image_hwc, (long long)H * W * 3, counts, static_cast<hipStream_t>(stream)));
    return R2F_OK;
}

// ------------------------------------------------------------------------------- whole frame
static size_t plane_set_floats(int H, int W) { return plan::plane_set_floats(H, W); }

// The launches of one frame, in order, on stream `stream` (a capturing stream of the context's own or the caller's).
static int render_launches(r2f_ctx* ctx, const r2f_params* p, const void* in, int in_layout, float* out_f32, uint8_t* out_u8, int H,
                           int W, void* workspace, void* stream) {
    const bool hal = p->flags & R2F_F_HALATION, mtf = p->flags & R2F_F_MTF, grain = p->flags & R2F_F_GRAIN;
    const bool burn = p->flags & R2F_F_BURN;
    const size_t set_floats = plane_set_floats(H, W);
    float* base = static_cast<float*>(workspace);
    r2f_planes A{base, (int64_t)(set_floats / 3), 0, H};
    r2f_planes B{base + set_floats, (int64_t)(set_floats / 3), 0, H};
    int rc, finished = 0;
    bool tracked = false;
    ctx->frame_dyn_armed = false;
    if (hal) {  // the halation's identity channels (blue on a colour stock) are finished by the front kernel, straight into B
        // kernel by kernel: the frame block (seed, exposure range reset) ahead of the front kernel.  A caller that keeps the seed
        // resident still gets the range reset -- every frame's record starts empty, whatever the flag (the union of
        // earlier frames' ranges used to decide frame N's scratch element); inside a capture the reset stays outside with the seed
        // write (r2f_render issues one or the other ahead of every replay)
        if (!(p->flags & R2F_F_FRAME_RESIDENT) || !ctx->capturing) {
            rc = write_frame_params(ctx, p, static_cast<hipStream_t>(stream), (p->flags & R2F_F_FRAME_RESIDENT) ? 2 : 1);
            if (rc) return rc;
        }
        // ... and also records the range of the exposure planes it writes (every row of them)
        rc = stage_front_impl(ctx, p, in, in_layout, 0, H, R2F_UPTO_EXPOSURE, &A, nullptr, nullptr, 0, 0, H, W, H, stream, &B, &finished, &tracked);
    } else
        rc = r2f_stage_front(ctx, p, in, in_layout, 0, H, R2F_UPTO_DENSITY, &A, nullptr, nullptr, 0, 0, H, W, H, stream);
    if (rc) return rc;
    const r2f_planes* cur = &A;
    const r2f_planes* other = &B;
    if (hal) {
        // (r2f_stage_halation with the vouching bit: the record's tiles were filled for exactly these exposure planes)
        rc = run_stencil(ctx, R2F_KERNEL_HALATION, cur, other, 0, H, W, H, 1, p->log_eps, static_cast<hipStream_t>(stream),
                         finished != 0 || (p->flags & R2F_F_IDENTITY_DONE) != 0, tracked);
        if (rc) return rc;
        std::swap(cur, other);
    }
    if (mtf) {
        rc = r2f_stage_mtf(ctx, p, cur, other, 0, H, W, H, stream);
        if (rc) return rc;
        std::swap(cur, other);
    }
    if (!burn) return r2f_stage_tail(ctx, p, cur, nullptr, out_f32, out_u8, 0, 0, H, W, H, stream);
    // S7: the burn map depends on the whole grained frame -> grain to planes, reduce, blur, then finish
    int sets_used = (hal || mtf) ? 2 : 1;
    if (grain) {
        if (sets_used == 1) sets_used = 2;
        rc = r2f_stage_grain(ctx, p, cur, other, 0, H, W, H, stream);
        if (rc) return rc;
        std::swap(cur, other);
    }
    int h_lo, w_lo;
    if (!burn_geometry(p, H, W, &h_lo, &w_lo)) return fail(ctx, R2F_EINVAL, "render: bad burn_cell");
    float* sums = base + (size_t)sets_used * set_floats;
    float* map = sums + (size_t)h_lo * w_lo;
    float* scratch = map + (size_t)h_lo * w_lo;
    rc = r2f_stage_burn_sums(ctx, p, cur, sums, 0, H, W, H, stream);
    if (rc) return rc;
    rc = r2f_stage_burn_map(ctx, p, sums, map, scratch, W, H, stream);
    if (rc) return rc;
    r2f_params q = *p;
    q.flags &= ~(uint32_t)R2F_F_GRAIN;
    return r2f_stage_tail(ctx, &q, cur, map, out_f32, out_u8, 0, 0, H, W, H, stream);
}

int r2f_render(r2f_ctx* ctx, const r2f_params* p, const void* in, int in_layout, float* out_f32, uint8_t* out_u8, int H,
               int W, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx || !p) return R2F_EINVAL;
    R2F_GUARD(ctx);
    if (H <= 0 || W <= 0) return fail(ctx, R2F_EINVAL, "render: empty frame");
    const size_t need = r2f_workspace_bytes(p, H, W);
    if (need > workspace_bytes || (need && !workspace)) return fail(ctx, R2F_EINVAL, "render: workspace too small (%zu needed)", need);
    if (need && !aligned16(workspace)) return fail(ctx, R2F_EINVAL, "render: workspace must be 16-byte aligned");
    const bool hal = p->flags & R2F_F_HALATION, mtf = p->flags & R2F_F_MTF, grain = p->flags & R2F_F_GRAIN;
    const bool burn = p->flags & R2F_F_BURN;
    if (!(hal || mtf || grain || burn))  // config "LUTs only": one fused pointwise pass (one submit as it is)
        return r2f_stage_front(ctx, p, in, in_layout, 0, H, R2F_UPTO_OUTPUT, nullptr, out_f32, out_u8, 0, 0, H, W, H, stream);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // per-launch event timing (bench.py's breakdowns) creates and records events per launch: kernel by kernel only
    if (!ctx->opt_render_graph || ctx->opt_timing) {
        ++ctx->stat_eager;
        return render_launches(ctx, p, in, in_layout, out_f32, out_u8, H, W, workspace, stream);
    }
    if (!ctx->retired.empty()) reap_retired_graphs(ctx, false);
    if (ctx->graphs_generation != ctx->generation) {  // a table, stencil, option or internal buffer moved: frozen pointers are stale
        drop_render_graphs(ctx);
        ctx->graphs_generation = ctx->generation;
    }
    r2f_ctx::RenderGraph key;
    key.in = in, key.in_layout = in_layout, key.out_f32 = out_f32, key.out_u8 = out_u8, key.H = H, key.W = W, key.workspace = workspace;
    key.p = *p;
    key.p.seed = 0;
    key.p.flags &= ~(uint32_t)R2F_F_FRAME_RESIDENT;
    auto same_structure = [](const r2f_ctx::RenderGraph& a, const r2f_ctx::RenderGraph& b) {
        return a.in_layout == b.in_layout && a.H == b.H && a.W == b.W && !memcmp(&a.p, &b.p, sizeof a.p);
    };
    auto same_entry = [&](const r2f_ctx::RenderGraph& a, const r2f_ctx::RenderGraph& b) {
        return a.in == b.in && a.out_f32 == b.out_f32 && a.out_u8 == b.out_u8 && a.workspace == b.workspace && same_structure(a, b);
    };
    int slot = -1;
    for (size_t i = 0; i < ctx->graphs.size(); ++i)
        if (same_entry(ctx->graphs[i], key)) slot = (int)i;
    auto eager = [&]() -> int {
        ++ctx->stat_eager;
        const uint64_t g0 = ctx->generation;
        const int rc = render_launches(ctx, p, in, in_layout, out_f32, out_u8, H, W, workspace, stream);
        if (ctx->generation != g0) {  // this frame built something (a lazy table, a larger scratch): every captured pointer may dangle
            drop_render_graphs(ctx);
            ctx->graphs_generation = ctx->generation;
        }
        if (rc == R2F_OK) {
            ctx->warm = key, ctx->warm_valid = true;
            bool known = false;
            for (const auto& k : ctx->seen) known = known || same_entry(k, key);
            if (!known) {
                if (ctx->seen.size() >= 16) ctx->seen.erase(ctx->seen.begin());
                ctx->seen.push_back(key);
            }
        }
        return rc;
    };
    auto seen_before = [&]() {
        for (const auto& k : ctx->seen)
            if (same_entry(k, key)) return true;
        return false;
    };
    if (slot >= 0 && ctx->graphs[slot].exec) {
        r2f_ctx::RenderGraph& g = ctx->graphs[slot];
        g.last_use = ++ctx->graph_clock;
        ctx->frame_dyn_armed = g.dyn_armed;
        {  // seed + range reset; a caller that wrote the seed itself (the flag) gets the range reset alone
            int rc = write_frame_params(ctx, p, s, (p->flags & R2F_F_FRAME_RESIDENT) ? 2 : 1);
            if (rc) return rc;
        }
        R2F_HIP(ctx, hipGraphLaunch(g.exec, s));
        if (g.done) R2F_HIP(ctx, hipEventRecord(g.done, s));
        ++ctx->stat_replays;
        return R2F_OK;
    }
    // Not captured yet.  The first frame of a structure runs kernel by kernel (uploads and allocations synchronise and cannot be
    // captured); once the context has rendered this structure at this generation, an entry is captured the second time its buffers
    // come by (fresh buffers every frame -- results a caller keeps alive -- would pay ~0.9 ms per 24 MP frame for graphs never replayed).
    if ((slot >= 0 && ctx->graphs[slot].never) || !ctx->warm_valid || !same_structure(ctx->warm, key) || !seen_before()) return eager();
    if (!ctx->cap_stream) R2F_HIP(ctx, hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking));
    if (slot < 0) {
        if (ctx->graphs.size() >= 8) {  // callers that hand in fresh buffers every frame: bounded bookkeeping, least recently used out
            size_t lru = 0;
            for (size_t i = 1; i < ctx->graphs.size(); ++i)
                if (ctx->graphs[i].last_use < ctx->graphs[lru].last_use) lru = i;
            retire_render_graph(ctx, ctx->graphs[lru]);  // (a replay of it may still be running: destroyed behind its event, later)
            ctx->graphs.erase(ctx->graphs.begin() + (long)lru);
        }
        ctx->graphs.push_back(key);
        slot = (int)ctx->graphs.size() - 1;
    }
    const uint64_t gen0 = ctx->generation;
    r2f_params q = *p;
    q.flags |= R2F_F_FRAME_RESIDENT;  // the seed write stays outside the graph
    hipGraph_t graph = nullptr;
    // thread-local mode: other threads of the process (RCCL's watchdog, a producer thread) may keep calling into HIP meanwhile
    hipError_t e = hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeThreadLocal);
    int rc = R2F_OK;
    if (e == hipSuccess) {
        ctx->capturing = true;
        rc = render_launches(ctx, &q, in, in_layout, out_f32, out_u8, H, W, workspace, ctx->cap_stream);
        ctx->capturing = false;
        e = hipStreamEndCapture(ctx->cap_stream, &graph);
    }
    hipGraphExec_t exec = nullptr;
    if (e == hipSuccess && rc == R2F_OK && graph && ctx->generation == gen0) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess || rc != R2F_OK || !exec || ctx->generation != gen0) {
        // a failed capture must not cost the frame: clear the sticky error, remember not to try again, launch kernel by kernel
        (void)hipGetLastError();
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        ++ctx->stat_dropped;
        if (ctx->generation != gen0) {
            drop_render_graphs(ctx);
            ctx->graphs_generation = ctx->generation;
        } else {
            ctx->graphs[slot].never = true;
        }
        return eager();
    }
    r2f_ctx::RenderGraph& g = ctx->graphs[slot];
    g.graph = graph;
    g.exec = exec;
    g.last_use = ++ctx->graph_clock;
    g.dyn_armed = ctx->frame_dyn_armed;  // (what the captured halation launches carry; r2f_frame_exposure_range after a replay)
    if (hipEventCreateWithFlags(&g.done, hipEventDisableTiming) != hipSuccess) g.done = nullptr, (void)hipGetLastError();
    ++ctx->stat_captures;
    rc = write_frame_params(ctx, p, s, (p->flags & R2F_F_FRAME_RESIDENT) ? 2 : 1);
    if (rc) return rc;
    R2F_HIP(ctx, hipGraphLaunch(g.exec, s));
    if (g.done) R2F_HIP(ctx, hipEventRecord(g.done, s));
    ++ctx->stat_replays;
    return R2F_OK;
}

int r2f_render_stats(const r2f_ctx* ctx, uint64_t* out4) {
    if (!ctx || !out4) return R2F_EINVAL;
    out4[0] = ctx->stat_replays, out4[1] = ctx->stat_captures, out4[2] = ctx->stat_eager, out4[3] = ctx->stat_dropped;
    return R2F_OK;
}

int r2f_frame_exposure_range(r2f_ctx* ctx, float* out4, int* armed, int* packed) {
    if (!ctx || !out4 || !armed || !packed) return R2F_EINVAL;
    R2F_GUARD(ctx);
    R2F_HIP(ctx, hipDeviceSynchronize());
    FrameParams v{};
    R2F_HIP(ctx, hipMemcpy(&v, ctx->frame_buf.p, sizeof v, hipMemcpyDeviceToHost));
    // the frame's extremes = the extremes over the record's tiles (the kernels merge into tiles only: nothing decides on the frame's
    // range any more); a record marked unusable (a front kernel that could not record) reads as max = +inf
    int lo = (int)kFrameMinReset, hi = (int)kFrameMaxReset;
    if (ctx->range_tiles.p && ctx->tiles_tyn > 0) {
        std::vector<int2> tiles((size_t)ctx->tiles_tyn * ctx->tiles_txn);
        R2F_HIP(ctx, hipMemcpy(tiles.data(), ctx->range_tiles.p, tiles.size() * sizeof(int2), hipMemcpyDeviceToHost));
        for (const int2& t : tiles) lo = std::min(lo, t.x), hi = std::max(hi, t.y);
    }
    if (v.e_max == 0x7f800000u) hi = 0x7f800000;
    memcpy(&out4[0], &lo, 4);
    memcpy(&out4[1], &hi, 4);
    dyn_rule(ctx, &out4[2], &out4[3]);
    *armed = ctx->frame_dyn_armed ? 1 : 0;
    // the choice is made per window pair (r2f_frame_scratch_choice has the counts): *packed says whether EVERY pair of the last
    // halation call took the 12-byte element
    int pairs = 0, packed_pairs = 0;
    int rc = r2f_frame_scratch_choice(ctx, &pairs, &packed_pairs);
    if (rc) return rc;
    *packed = (*armed && pairs > 0 && packed_pairs == pairs) ? 1 : 0;
    return R2F_OK;
}

int r2f_frame_scratch_choice(r2f_ctx* ctx, int* pairs, int* packed_pairs) {
    if (!ctx || !pairs || !packed_pairs) return R2F_EINVAL;
    R2F_GUARD(ctx);
    *pairs = *packed_pairs = 0;
    if (!ctx->frame_dyn_armed || !ctx->dyn_flags.p || ctx->dyn_flags_ppc <= 0) return R2F_OK;
    R2F_HIP(ctx, hipDeviceSynchronize());
    std::vector<int> flags((size_t)ctx->dyn_flags_ppc);
    R2F_HIP(ctx, hipMemcpy(flags.data(), ctx->dyn_flags.p, flags.size() * sizeof(int), hipMemcpyDeviceToHost));
    *pairs = ctx->dyn_flags_ppc;
    for (int f : flags) *packed_pairs += f != 0;
    return R2F_OK;
}

int r2f_frame_scratch_flags(r2f_ctx* ctx, int32_t* out, int capacity, int* count) {
    if (!ctx || !count || capacity < 0 || (capacity > 0 && !out)) return R2F_EINVAL;
    R2F_GUARD(ctx);
    *count = 0;
    if (!ctx->frame_dyn_armed || !ctx->dyn_flags.p || ctx->dyn_flags_ppc <= 0) return R2F_OK;
    R2F_HIP(ctx, hipDeviceSynchronize());
    *count = ctx->dyn_flags_ppc;
    const int n = std::min(capacity, ctx->dyn_flags_ppc);
    if (n > 0) R2F_HIP(ctx, hipMemcpy(out, ctx->dyn_flags.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return R2F_OK;
}

int r2f_write_frame_params(r2f_ctx* ctx, const r2f_params* p, void* stream) {
    if (!ctx || !p) return R2F_EINVAL;
    R2F_GUARD(ctx);
    return write_frame_params(ctx, p, static_cast<hipStream_t>(stream));
}

}  // extern "C"
