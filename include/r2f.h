/*
 * r2f.h -- C ABI of libr2f_hip.so: the MI355X (gfx950) film-emulation render path.
 *
 * This is the drop-in boundary under raw2film's processor objects.  The reference has no
 * FFI for this path (it is Python + WGSL); each entry point below names the reference
 * interface it stands in for (paths relative to /root/reference/src/raw2film/).
 *
 * Conventions
 *   - every function returns 0 on success, a negative R2F_E* code on failure;
 *     r2f_last_error(ctx) gives the message (thread-compatible: one ctx, one thread at a time,
 *     like the reference's processor objects -- gui.py:2119-2129).
 *   - `const float* host_*` arguments are HOST pointers (LUTs / stencils, copied to the device);
 *     image buffers are DEVICE pointers owned by the caller (torch tensors on the Python side).
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default).
 *   - images are fp32.  Frame = H_global x W pixels; a call may address a row shard of it.
 *
 * No torch / C++ types cross this boundary.
 */
#ifndef R2F_H
#define R2F_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: the entry points below are its ONLY dynamic symbols (tests/test_lib_symbols.py
 * compares `nm -D --defined-only` with this header), so two builds of it can live in one process without interposing each other. */
#if defined(__GNUC__) || defined(__clang__)
#define R2F_API __attribute__((visibility("default")))
#else
#define R2F_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct r2f_ctx r2f_ctx;

enum {
    R2F_OK = 0,
    R2F_EINVAL = -1,   /* bad argument / missing LUT or stencil for an enabled stage */
    R2F_EHIP = -2,     /* a HIP runtime call failed */
    R2F_ETOOLARGE = -3 /* stencil does not fit the LDS tile configurations */
};

/* in_layout of r2f_render / r2f_stage_front */
enum {
    R2F_LAYOUT_HWC3 = 0, /* (H, W, 3) interleaved -- cpu_processor.py:136 `tex_input`            */
    R2F_LAYOUT_HWC4 = 1, /* (H, W, 4), alpha ignored -- gpu_processor.py:765 `image_array`        */
    R2F_LAYOUT_CHW = 2   /* 3 planes of (H, W) -- the device-native layout                         */
};

/* `which` of r2f_set_kernel */
enum { R2F_KERNEL_HALATION = 0, R2F_KERNEL_MTF = 1, R2F_KERNEL_GRAIN = 2 };

/* r2f_params.flags: stage gates, same conditions as cpu_processor.py:368,382,387 */
enum {
    R2F_F_MATRIX = 1u << 0,   /* S0: apply the 3x3 set by r2f_set_matrix3x3 (off = input already XYZ) */
    R2F_F_HALATION = 1u << 1, /* S2 */
    R2F_F_MTF = 1u << 2,      /* S5 */
    R2F_F_GRAIN = 1u << 3,    /* S6 (+ the clip of cpu_processor.py:397) */
    R2F_F_GRAIN_MONO = 1u << 4, /* grain == 1: noise_bw.wgsl */
    R2F_F_BURN = 1u << 5,       /* S7 highlight burn (effects.py:396-418), needs burn_* below */
    R2F_F_IDENTITY_DONE = 1u << 6, /* r2f_stage_halation only: the channels whose halation stencil is a single tap at the anchor
                                     (the blue layer of a colour stock, effects.py:248-263) were already finished into the
                                     density planes by r2f_stage_front_split for rows [y0, y1) -- skip them */
    R2F_F_FRAME_RESIDENT = 1u << 7, /* stage entry points: do NOT write p->seed into the context's device-side frame block first;
                                     the kernels read what r2f_write_frame_params last wrote there (in stream order).  For callers
                                     that replay captured launches: the write stays outside the capture, the launches inside */
    R2F_F_TRACK_RANGE = 1u << 8,   /* r2f_stage_front (upto = EXPOSURE) / r2f_stage_front_split: merge min and max |.| of the exposure
                                     samples this call writes for the halation's FFT channels into the context's exposure-range record
                                     -- a grid of 64 x 256-pixel tiles over the GLOBAL frame, reset by r2f_write_frame_params -- which is
                                     what r2f_render's front kernel does for a whole frame.  A call whose kernel cannot record (the
                                     generic pointwise kernel) leaves its tiles "unknown": the windows that touch them keep complex128,
                                     nothing goes wrong silently */
    R2F_F_RANGE_VALID = 1u << 9    /* r2f_stage_halation: the caller has kept the record for the rows the `exposure` buffer holds for the
                                     FFT channels (own rows by R2F_F_TRACK_RANGE front calls, rows received from neighbours by
                                     r2f_stage_exposure_range): the passes may then choose the 12-byte scratch element on the device,
                                     window pair by window pair, exactly like r2f_render's (below).  A tile nobody recorded reads as
                                     "unknown" and keeps complex128 for the windows that touch it; what the flag vouches for is that
                                     recorded tiles describe THIS frame's samples (the record was reset at the start of the frame) */
};

/* `upto` of r2f_stage_front */
enum { R2F_UPTO_EXPOSURE = 0, R2F_UPTO_DENSITY = 1, R2F_UPTO_OUTPUT = 2 };

typedef struct r2f_params {
    uint32_t flags;
    uint32_t seed;      /* grain seed; upstream: random per render (gpu_processor.py:586-592).  Never a launch argument: it is
                           written to a device-side block the grain kernels read (r2f_write_frame_params) */
    float log_eps;      /* lut_1d.wgsl:24 -> 1e-6 */
    float lut3d_scale;  /* cpu_processor.py:405 -> 0.25 */
    int32_t lut3d_mode; /* 0 tetrahedral (utils.py:247, the parity target), 1 trilinear (lut_3d.wgsl) */
    int32_t burn_cell;  /* S7: ceil(min(H, W) / burn_scale), the shrink factor of effects.down_up_blur (effects.py:365) */
    float burn_strength; /* S7: highlight_burn */
    float burn_d_ref;    /* S7: negative_film.d_ref[1] (or [0]), effects.py:406 */
} r2f_params;

/* Three fp32 planes holding global rows [gy0, gy0+rows) of a frame, W floats per row. */
typedef struct r2f_planes {
    float* data;          /* device pointer to plane 0, row gy0 */
    int64_t plane_stride; /* floats between consecutive channel planes (>= rows*W) */
    int32_t gy0;
    int32_t rows;
} r2f_planes;

/* --- lifetime: GpuProcessor.__init__ / device + pipeline creation, gpu_processor.py:73-257 --- */
R2F_API int r2f_create(int device, r2f_ctx** out);
R2F_API void r2f_destroy(r2f_ctx* ctx);
R2F_API const char* r2f_last_error(const r2f_ctx* ctx);
/* "gfx950" build id + ABI version, for the loader's sanity check */
R2F_API const char* r2f_version(void);

/* --- resource upload: the _ensure_* methods, gpu_processor.py:307-611 --- */
/* S0 matrix, row-major 3x3 (data.py:128-135 for linear Rec.709 input). */
R2F_API int r2f_set_matrix3x3(r2f_ctx* ctx, const float* host_m9);
/* S1 (n, n, 3) input LUT = negative_film.get_input_lut(...)   -- _ensure_lut_2d :349-376 */
R2F_API int r2f_set_lut2d(r2f_ctx* ctx, const float* host_lut, int n);
/* S4 (4, m) density curve: row 0 xp, rows 1..3 fp              -- _ensure_lut_1d :307-347 */
R2F_API int r2f_set_curve1d(r2f_ctx* ctx, const float* host_lut4xm, int m);
/* S8 (n, n, n, 3) output LUT = create_lut(..., linear_scaling=4) -- _ensure_lut_3d :378-409 */
R2F_API int r2f_set_lut3d(r2f_ctx* ctx, const float* host_lut, int n);
/* S6c (4, m) grain LUT indexed by density                      -- _ensure_grain_lut :565-611 */
R2F_API int r2f_set_grain_lut(r2f_ctx* ctx, const float* host_lut4xm, int m);
/* S2/S5/S6b stencil, (kh, kw, kc) row-major, kc in {1, 3}; correlation, anchor (kh/2, kw/2)
 *   -- _ensure_halation_kernel :498-545, _ensure_mtf_kernel :411-451, _ensure_grain_kernel :453-496 */
R2F_API int r2f_set_kernel(r2f_ctx* ctx, int which, const float* host_khwc, int kh, int kw, int kc);

/* --- whole-frame render: CpuProcessor.process hot loop cpu_processor.py:363-407,
 *     GpuProcessor._execute_gpu_pipeline gpu_processor.py:1756-1877 ---
 * in: device image (in_layout), H x W.  out_f32_hwc / out_u8_hwc: device (H, W, 3), either may be NULL.
 * workspace: device scratch of at least r2f_workspace_bytes(...) bytes (caller-owned): the plane sets between the stages.
 * Stencils of >= 400 taps run as fp64 overlap-save FFTs (like cv.filter2D's own DFT branch above 11 x 11 taps, which the
 * reference's CPU path takes for both of them); their pass scratch (1 MiB per window pair in flight, 192 by default) and
 * the kernels' spectra (1 MiB per stencil channel) belong to the context, allocated on first use.
 * Scratch element of the halation's passes: complex128, or -- chosen on the device, frame by frame and WINDOW PAIR by window pair -- a
 * 12-byte element (each component a double rounded to 48 bits) when the range of the exposure samples the pair's two windows hold
 * allows it (the range comes from a grid of 64 x 256-pixel tiles the front kernel fills as it writes the exposure planes; a window
 * takes the extremes of the tiles it touches): max |x| <= bound x max(min x, first breakpoint of the density curve), the bound derived from the curve's steepest
 * cell and the element's worst error (two roundings at 2^-37 of max / shadow, x 1.5: a searched constant, tests/test_gpu_fft.py) so
 * that the element costs a density at most three fp32 ulps -- the MTF's complex64 scratch is allowed the same -- (option
 * stencil_fft_scratch96_auto, default 1; with the stand-in Portra curve windows up to max / shadow = 6.2e4 take it, wider ones
 * keep complex128 -- the headline's noise frame spans 1.4e5 as a whole and ~2e4 per window: 99 % of its window pairs take it).  The stage entry points make the same choice when their caller keeps the
 * record (R2F_F_TRACK_RANGE, r2f_stage_exposure_range, R2F_F_RANGE_VALID) and use complex128 otherwise; a row shard's windows are
 * anchored at ITS first row, so a whole-frame render and a row-sharded one agree to that element's rounding, not bit for bit.
 * Non-finite samples: in the direct form a NaN / infinity in a stencil's input comes out as NaN in every output whose tap box
 * (plus up to three zero-weight padding rows / columns) covers it, like the per-tap loop of the reference's convolution.wgsl; the FFT form takes such a sample as 0 instead (a NaN
 * handed to the transforms would come back in every output of its 256 x 512 window) -- the outputs inside the tap box are then
 * those of the frame with that sample zeroed, everything further away is untouched in both forms.  The same stencil therefore
 * treats a bad sample differently depending on which form its size selects (r2f_stencil_stats); through r2f_render both end
 * finite, because S3's max(x, log_eps) replaces a NaN (tests/test_gpu_hostile.py pins both behaviours).
 *
 * One submit per frame, like the reference's single command encoder (gpu_processor.py:1760 create_command_encoder ...
 * :1877 queue.submit): the second time a frame arrives with the same buffers, shape and parameters (the seed excepted) its
 * launches are captured into a HIP graph on a stream of the context's own and, from then on, a frame costs one write of the
 * frame block (the seed: the reference re-creates its uniform buffer `buffer_params_grain` per render, gpu_processor.py:585-597) plus one graph
 * launch on `stream`.  Any table / stencil / option change (r2f_generation) drops the graphs.  Results are the eager
 * launches', bit for bit.  r2f_set_option(ctx, "render_graph", 0) turns the replay off (A/B). */
R2F_API size_t r2f_workspace_bytes(const r2f_params* p, int H, int W);
R2F_API int r2f_render(r2f_ctx* ctx, const r2f_params* p, const void* in, int in_layout, float* out_f32_hwc,
               uint8_t* out_u8_hwc, int H, int W, void* workspace, size_t workspace_bytes, void* stream);
/* Counters of r2f_render since r2f_create: out[0] frames replayed from a graph, out[1] graphs captured, out[2] frames launched
 * kernel by kernel, out[3] graphs dropped (context changes, evictions, failed captures). */
R2F_API int r2f_render_stats(const r2f_ctx* ctx, uint64_t* out4);

/* Introspection for the measurement harness (synchronises the device): what the front kernel of the last whole-frame render (or
 * the front / range calls of a row shard's last frame: R2F_F_TRACK_RANGE, r2f_stage_exposure_range) recorded about the exposure
 * planes the halation's FFT passes read, and what they made of it.  out4 = {min x, max |x| (the extremes over the record's tiles,
 * reduced on the host: the FRAME's range, which decides nothing by itself), bound, floor}; *armed = 1 when that render's halation
 * launches carried the rule (stencil_fft_scratch96_auto, 256-row windows, real spectrum: r2f_render above), *packed = 1 when EVERY
 * window pair then took the 12-byte scratch element (r2f_frame_scratch_choice has the counts).
 * Valid until the next write of the frame block (the next render).  Nothing upstream corresponds to it. */
R2F_API int r2f_frame_exposure_range(r2f_ctx* ctx, float* out4, int* armed, int* packed);
/* The choice itself, which is made PER WINDOW PAIR since round 6 (a pair's two windows' own range, from a grid of 64 x 256-pixel tiles
 * the front kernel fills beside the frame-level extremes): of the *pairs window pairs per channel of the last halation call that chose
 * on the device, *packed_pairs took the 12-byte element (r2f_frame_exposure_range's *packed = all of them).  0 / 0 when the last call
 * did not choose.  Synchronises the device; introspection for the measurement harness and the tests. */
R2F_API int r2f_frame_scratch_choice(r2f_ctx* ctx, int* pairs, int* packed_pairs);
/* The same choice pair by pair: flags[i] = 1 when window pair i (per channel; its windows are 2 i and 2 i + 1 in row-major order over the
 * call's grid of windows, r2f_plan_fft) took the 12-byte element.  At most `capacity` flags are written; *count = pairs per channel
 * (0 when the last halation call did not choose).  Synchronises the device.  What tools/scratch_choice_model.py checks against a host
 * model of the samples each pair's windows hold: a pair may take the element only if ITS samples allow it. */
R2F_API int r2f_frame_scratch_flags(r2f_ctx* ctx, int32_t* flags, int capacity, int* count);

/* The per-render uniform write: p->seed -> the context's device-side frame block, asynchronously on `stream`
 * (gpu_processor.py:585-597: the uniform buffer `buffer_params_grain` with a fresh random seed, made ahead of the dispatches;
 * noise.wgsl:1-6 reads it).  r2f_render and, unless R2F_F_FRAME_RESIDENT is set, every stage entry point that makes grain does this itself.
 * ONE STREAM IN FLIGHT PER CONTEXT: the frame block, like the context's FFT scratch, internal streams and capture stream, exists once
 * per context.  Frames (or stage calls) of one context must be ordered on one stream (or by events): issued on two streams at once,
 * the write for frame B may land before frame A's grain kernels have read the seed, and both would share the scratch.  The
 * reference's processors are entered by one thread at a time for the same reason (mutable caches; gui.py:2119-2129); callers that
 * want frames in flight side by side use one context each (BatchSharder does). */
R2F_API int r2f_write_frame_params(r2f_ctx* ctx, const r2f_params* p, void* stream);

/* --- stage entry points (row-shard aware): one per compute pass of
 *     gpu_processor.py:1763-1862; used by the multi-GPU row tiler and by the parity tests.
 * A call computes global rows [y0, y1) of an H_global x W frame.  Source rows outside
 * [0, H_global) are reflected (BORDER_REFLECT_101, as cv.filter2D does on the CPU path);
 * every reflected source row must lie inside the source buffer's [gy0, gy0+rows).
 * The stencil stages (r2f_stage_halation, r2f_stage_mtf, r2f_stage_stencil, r2f_stage_chroma_nr_v) are OUT OF PLACE: a
 * destination plane that shares bytes with a source plane is refused with R2F_EINVAL (tiles and FFT batches read halo rows
 * that others would already have overwritten).  r2f_stage_grain is pointwise on the density and may run exactly in place
 * (same base, stride and first row); any other overlap is refused.
 * Every entry point binds the context's device for the duration of the call and restores the caller's current device on
 * return, so several contexts (one per GPU) can be driven from one thread. --- */

/* S0+S1 (+S3+S4 (+S8)) pointwise.  `in` holds global rows [in_gy0, in_gy0+in_rows).
 * upto=EXPOSURE/DENSITY writes planes `dst`; upto=OUTPUT writes out_* (rows indexed from out_gy0). */
R2F_API int r2f_stage_front(r2f_ctx* ctx, const r2f_params* p, const void* in, int in_layout, int in_gy0, int in_rows,
                    int upto, const r2f_planes* dst, float* out_f32_hwc, uint8_t* out_u8_hwc, int out_gy0, int y0,
                    int y1, int W, int H_global, void* stream);
/* r2f_stage_front(upto = EXPOSURE) for a frame that goes on to r2f_stage_halation, split by what the halation does to each
 * channel: a channel with a real stencil gets its exposure written to `exposure`; a channel whose halation stencil is ONE tap
 * at the anchor (f_c = 0: the blue layer of a colour stock) needs no neighbours, so its tap weight, S3 and S4 are applied here
 * and the result goes straight to `density` -- its exposure plane is not written at all.  Saves that plane's round trip and
 * the pointwise pass over it.  *finished_mask receives the channels handled that way (bit c); when it is non-zero call
 * r2f_stage_halation with R2F_F_IDENTITY_DONE.  Whole-frame use only: a row shard's neighbours need the exposure rows of
 * every channel.  Falls back to plain r2f_stage_front (mask 0) when no channel qualifies or the fast kernel does not apply. */
R2F_API int r2f_stage_front_split(r2f_ctx* ctx, const r2f_params* p, const void* in, int in_layout, int in_gy0, int in_rows,
                          const r2f_planes* exposure, const r2f_planes* density, int y0, int y1, int W, int H_global,
                          int* finished_mask, void* stream);
/* A row shard's half of r2f_render's exposure-range record: min and max |.| of rows [y0, y1) and [y2, y3) of `exposure` (the
 * channels whose halation stencil takes the FFT form) merged into the context's record (its tiles), in stream order -- for the halo rows a
 * rank received from its neighbours above and below, in one launch (an empty range is skipped; its own rows are recorded by the
 * front kernel, R2F_F_TRACK_RANGE).  Nothing upstream corresponds to it. */
R2F_API int r2f_stage_exposure_range(r2f_ctx* ctx, const r2f_planes* exposure, int y0, int y1, int y2, int y3, int W, void* stream);
/* S2 halation stencil on exposure + S3 log + S4 curve -> density planes.  With R2F_F_RANGE_VALID the FFT passes choose their
 * scratch element on the device, per window pair, from the exposure-range record (see r2f_render); without it they keep complex128. */
R2F_API int r2f_stage_halation(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* exposure, const r2f_planes* density,
                       int y0, int y1, int W, int H_global, void* stream);
/* S5 MTF stencil on density -> density planes. */
R2F_API int r2f_stage_mtf(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* density_in, const r2f_planes* density_out,
                  int y0, int y1, int W, int H_global, void* stream);
/* [S6 grain + clip] + [S7 burn subtract + clip, when burn_map != NULL] + S8 3-D LUT (+ S9 uint8 truncation)
 * -> interleaved output.  With a burn map the grain (if any) must already have been applied by r2f_stage_grain. */
R2F_API int r2f_stage_tail(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* density, const float* burn_map,
                   float* out_f32_hwc, uint8_t* out_u8_hwc, int out_gy0, int y0, int y1, int W, int H_global,
                   void* stream);
/* The grain in two halves, so that the first can run on another stream while the stencils run: r2f_stage_grain_field writes
 * the grain field G = K_g * N (noise.wgsl + the convolution of grain.wgsl:63-75; a function of the seed and the pixel
 * coordinates only) for rows [y0, y1) to `field` planes; r2f_stage_tail_field is r2f_stage_tail with that field applied
 * pointwise (out = D + G * lut(D), clip, 3-D LUT) instead of being generated in the same kernel.  Same arithmetic, same
 * results as r2f_stage_tail.  Not combinable with a burn map (use r2f_stage_grain there). */
R2F_API int r2f_stage_grain_field(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* field, int y0, int y1, int W, int H_global,
                          void* stream);
R2F_API int r2f_stage_tail_field(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* density, const r2f_planes* field, float* out_f32,
                         uint8_t* out_u8, int out_gy0, int y0, int y1, int W, int H_global, void* stream);

/* S6 grain + clip alone, density planes -> density planes (the first half of the tail when S7 is on: the burn map
 * is a function of the WHOLE grained frame, so the frame has to exist before any pixel can be finished). */
R2F_API int r2f_stage_grain(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* density_in, const r2f_planes* density_out,
                    int y0, int y1, int W, int H_global, void* stream);
/* S7 part 1: area-weighted (cv.resize INTER_AREA) partial sums of the green density over rows [y0, y1) into
 * cell_sums[(H_global / burn_cell) * (W / burn_cell)] (device).  Row shards add their arrays (all-reduce SUM). */
R2F_API int r2f_stage_burn_sums(r2f_ctx* ctx, const r2f_params* p, const r2f_planes* density, float* cell_sums, int y0, int y1,
                        int W, int H_global, void* stream);
/* S7 part 2: clip(x - d_ref, 0) and gaussian_filter(sigma=3, truncate=2) on the low-res map (device -> device).
 * `scratch` holds 2 x the map size. */
R2F_API int r2f_stage_burn_map(r2f_ctx* ctx, const r2f_params* p, const float* cell_sums, float* burn_map, float* scratch, int W,
                       int H_global, void* stream);
/* Pre-path chroma noise reduction, effects.chroma_nr_filter (effects.py:547-561): XYZ -> xyY, a separable
 * (2*size+1)-tap Gaussian on the two chromaticity planes only (coordinates clamped to the frame), back to XYZ.
 * Pass 1 (rows independent): `in` -> planes {x blurred horizontally, y blurred horizontally, Y}.
 * Pass 2 (needs size rows above/below, clamped at the global edges): those planes -> XYZ planes, which
 * r2f_stage_front / r2f_render accept as R2F_LAYOUT_CHW input. */
R2F_API int r2f_stage_chroma_nr_h(r2f_ctx* ctx, const void* in, int in_layout, int in_gy0, int in_rows, const r2f_planes* dst,
                          int size, int y0, int y1, int W, void* stream);
R2F_API int r2f_stage_chroma_nr_v(r2f_ctx* ctx, const r2f_planes* src, const r2f_planes* dst, int size, int y0, int y1, int W,
                          int H_global, void* stream);

/* Pre-path down-scale to the preview / pipeline resolution: cv.resize(..., interpolation=cv.INTER_AREA) as
 * utils.resolution_scaling applies it when the target is smaller than the frame (utils.py:226-236, called at
 * cpu_processor.py:134).  `in` is a whole H x W frame (any in_layout); `dst` receives out_h x out_w planes. */
R2F_API int r2f_resize_area(r2f_ctx* ctx, const void* in, int in_layout, int H, int W, const r2f_planes* dst, int out_h, int out_w,
                    void* stream);

/* Pre-path free rotation: cv.warpAffine(rgb, rot_mat, same size, flags=cv.INTER_LINEAR) of effects.rotate (effects.py:46-52;
 * constant border 0), restricted to the window [oy, oy + out_h) x [ox, ox + out_w) that rotate()'s centred crop keeps
 * (effects.py:54-74).  m_dst_to_src: the INVERSE of cv.getRotationMatrix2D(...), 2 x 3 row-major doubles (host memory).
 * `in` is a whole H x W frame (any in_layout); `dst` receives out_h x out_w planes. */
R2F_API int r2f_warp_affine(r2f_ctx* ctx, const void* in, int in_layout, int H, int W, const double* m_dst_to_src, const r2f_planes* dst,
                    int out_h, int out_w, int oy, int ox, void* stream);

/* Post-path up-scale of the rendered uint8 frame: utils.resolution_scaling's cv.resize(image, dsize,
 * interpolation=cv.INTER_LANCZOS4) branch (utils.py:237-242), the way back from the `max_scale` pipeline resolution to the
 * requested one (cpu_processor.py:128-134, 411-412).  src/dst: uint8 (H, W, 3) / (out_h, out_w, 3) on the device.
 * r2f_lanczos4_table is the host-side table cv::resize derives per destination index (source index of tap 3, eight weights
 * in 11-bit fixed point); exported so the tests can pin it without a GPU. */
R2F_API int r2f_resize_lanczos4_u8(r2f_ctx* ctx, const uint8_t* src_hwc, int H, int W, uint8_t* dst_hwc, int out_h, int out_w, void* stream);
R2F_API int r2f_lanczos4_table(int ssize, int dsize, int* ofs, short* coef);

/* Pre-path up-scale to a preview LARGER than the frame: utils.resolution_scaling's cv.resize(float32 frame, dsize,
 * interpolation=cv.INTER_LANCZOS4) (utils.py:237-242, called at cpu_processor.py:134).  `in`: a whole H x W float32 frame (any
 * in_layout); `dst` receives out_h x out_w planes.  r2f_lanczos4_table_f32: the host-side per-destination tables (source index
 * of tap 3, eight float weights), exported for the tests. */
R2F_API int r2f_resize_lanczos4_f32(r2f_ctx* ctx, const void* in, int in_layout, int H, int W, const r2f_planes* dst, int out_h, int out_w,
                            void* stream);
R2F_API int r2f_lanczos4_table_f32(int ssize, int dsize, int* ofs, float* coef);

/* Caller-side RGB histogram of the rendered bitmap: the counting loop of utils.generate_histogram (utils.py:160-165; GPU twin
 * histogram.wgsl pass1_accumulate, dispatched at gpu_processor.py:1149).  image_hwc: uint8 (H, W, 3) on the device, 16-byte
 * aligned; counts: 3 x 256 uint32 on the device (R bins, G bins, B bins), overwritten.  The 768-value post-processing
 * (log1p, 3-bin smoothing, bar image) is host work: raw2film_amd/histogram.py. */
R2F_API int r2f_histogram_u8(r2f_ctx* ctx, const uint8_t* image_hwc, int H, int W, uint32_t* counts, void* stream);

/* The CPU processor's last step (cpu_processor.py:411-412 -> utils.resolution_scaling, utils.py:226-236): cv.resize(uint8 frame,
 * INTER_AREA) when the rendered (and canvas-framed) frame is larger than the requested resolution.  src / dst: uint8
 * (H, W, 3) / (out_h, out_w, 3) on the device, out_h <= H, out_w <= W. */
R2F_API int r2f_resize_area_u8(r2f_ctx* ctx, const uint8_t* src_hwc, int H, int W, uint8_t* dst_hwc, int out_h, int out_w, void* stream);

/* The hand-off from RAW decoding: the last two lines of raw_to_linear (raw_conversion.py:50-52) applied to LibRaw's 16-bit
 * output on the device, so that a decoded frame crosses PCIe as uint16 (6 bytes per pixel) instead of float32 (12 or 16):
 * dst = float(src) / divisor (65535, one correctly rounded fp32 division) * factor (the float32 of 2 ** calc_exposure(...),
 * color_processing.py:71-99 -- computed by the caller, raw2film_amd.decode.auto_exposure).  src: uint16 (H, W, channels) on
 * the device, channels 3 or 4 (a fourth is dropped); dst: float32 (H, W, 3), clamped at 65504 like every frame the GPU path
 * uploads (gpu_processor.py:275).  Bit-identical to the NumPy expressions. */
R2F_API int r2f_decode_u16(r2f_ctx* ctx, const uint16_t* src_hwc, int H, int W, int channels, float divisor, float factor, float* dst_f32_hwc3,
                   void* stream);

/* The GPU processor's preview blit, shaders/copy_to_int.wgsl as bound by gpu_processor.py:1416-1539: the display-referred float
 * frame (H, W, 3) sampled bilinearly (clamp to edge) into an RGBA8 destination (dst_h, dst_w, 4): inside the scaled image the
 * sample (alpha 255), elsewhere inside the canvas bounds the canvas colour, transparent outside.  The transform is the shader's
 * uniform block (raw2film_amd.geometry.blit_transform computes it like _bind_copy_to_dst). */
typedef struct r2f_blit {
    float scale_x, scale_y;    /* 1 / rendered size of the image inside the destination, in destination pixels */
    float offset_x, offset_y;  /* its top-left corner */
    float canvas_min_x, canvas_min_y, canvas_max_x, canvas_max_y;
    float canvas_color[3];     /* 0..1 */
} r2f_blit;
R2F_API int r2f_blit_rgba8(r2f_ctx* ctx, const float* src_f32_hwc, int H, int W, uint8_t* dst_rgba, int dst_h, int dst_w, const r2f_blit* t,
                   void* stream);

/* shaders/histogram.wgsl pass2_process + pass3_render and shaders/scale_texture.wgsl (gpu_processor.py:1245-1285, 1883-1889) on
 * the counts of r2f_histogram_u8: log1p of the normalised counts, 3-bin smoothing, bar heights, the (height, 256, 4) RGBA bar
 * image coloured by mix_table_rgba (HOST, 8 x 4 bytes, index is_r * 4 + is_g * 2 + is_b) and, when target_rgba is given, its
 * nearest-neighbour copy into a (target_h, target_w, 4) widget texture.  counts / image / target are device pointers. */
R2F_API int r2f_histogram_render(r2f_ctx* ctx, const uint32_t* counts, const uint8_t* mix_table_rgba, int height, uint8_t* image_rgba,
                         uint8_t* target_rgba, int target_h, int target_w, void* stream);

/* Test entry for S6a: raw PCG3D hash (3 uint32 planes) and Gaussian field (3 fp32 planes) for
 * global rows [y0, y1); either output may be NULL.  noise.wgsl:14-62 / noise_bw.wgsl. */
R2F_API int r2f_stage_noise(r2f_ctx* ctx, const r2f_params* p, uint32_t* hash_planes, float* noise_planes, int y0, int y1,
                    int W, void* stream);
/* Test entry: plain per-channel stencil (no epilogue) with the kernel set for `which`. */
R2F_API int r2f_stage_stencil(r2f_ctx* ctx, int which, const r2f_planes* src, const r2f_planes* dst, int y0, int y1, int W,
                      int H_global, void* stream);

/* Introspection for the measurement harness: what the device form of stencil `which` executes.  out: 3 channels x 8 ints
 * {entries, row steps, LDS phases, form word, cropped rows, cropped (padded) columns, rows per lane, FFT word}.
 * Form word: bit 0 = mirrored taps are paired in the entry list; the bits above = R when the channel takes the fully unrolled
 * (2 R + 1)^2 direct form instead of the entry list (bits 1-7); bit 8 = the grain stencil is u v^T to fp32 rounding and runs as two
 * 1-D passes (known after the first tail launch).  FFT word: bit 0 = the channel takes the FFT form (then the direct form's numbers before it are not what runs); the bits
 * above it = window rows * 4096 + window columns of its last launch (0 before the first); bit 30 = that launch multiplied by a REAL
 * kernel spectrum (taps centrally symmetric around an anchor at the centre of their box: option stencil_fft_real_spectrum).  One entry =
 * 32 packed FMAs per lane for 16 pixels (x2 taps when mirror-paired). */
R2F_API int r2f_stencil_stats(r2f_ctx* ctx, int which, int* out);

/* Per-launch device timing of the FFT stencil passes, for the roofline line of bench.py.  After r2f_set_option(ctx,
 * "kernel_timing", mask) every launch of a pass cls whose bit (1 << cls) is set (0 rows forward, 1 columns, 2 rows inverse) is bracketed by events on its own
 * stream; this call waits for them, returns their summed duration, the launch count and the summed algorithmic bytes
 * (scratch images and windows the pass has to move; the 1 MB kernel spectrum stays in L2), and resets the counters.
 * cls = pass (0..2) for launches on complex128 scratch (the halation), pass + 3 for launches on complex64 scratch (the MTF). */
R2F_API int r2f_kernel_timing(r2f_ctx* ctx, int cls, double* total_ms, int* launches, double* bytes);

/* Measurement aid for bench.py's `roofline.copy_ceiling` ("state the measured hipMemcpyDtoD / stream-triad
 * ceiling beside it"): a float4 streaming copy of `bytes` bytes from src to dst on the device -- 2 x bytes of HBM traffic and no
 * arithmetic.  Nothing upstream corresponds to it; bytes must be a multiple of 16 and both buffers 16-byte aligned. */
R2F_API int r2f_stream_copy(r2f_ctx* ctx, const void* src, void* dst, size_t bytes, void* stream);

/* Change counter of everything a captured HIP graph of this context's launches freezes: bumped by every table / stencil /
 * matrix upload, every option change and every re-allocation of a context-owned buffer (LUTs, stencil forms, FFT scratch and
 * spectra).  The reference re-binds its resources per dispatch (gpu_processor.py:1756-1877) and has nothing to invalidate; a
 * caller that replays captured launches must re-capture when this value moves. */
R2F_API uint64_t r2f_generation(const r2f_ctx* ctx);

/* Tuning knobs for A/B runs (every one of them bumps r2f_generation).  Round 5's, all defaulting to the faster form:
 *   stencil_fft_real_spectrum  1: centrally symmetric tap boxes (every halation disc / MTF kernel the reference builds) are laid
 *                                 out around the window origin, pass 2 multiplies by a REAL spectrum (8 B per element)
 *   stencil_fft_cols_walk      1: pass 2 of 256-row windows with a real spectrum runs as a resident grid walking the launch's
 *                                 pairs per column block (spectrum in registers); 0: one workgroup per (column block, pair)
 *   stencil_fft_scratch96_auto 1: the halation's passes take the 12-byte scratch element for the window pairs whose exposure range
 *                                 allows it (above); stencil_fft_scratch96 (mask per stencil) forces it for all
 *   stencil_fft_mixed_sign     1: channels with taps of both signs take the float64 FFT form whatever their size
 * (older ones: stencil_fft, stencil_fft_window[_rows|_max], stencil_fft_batch, stencil_fft_streams, stencil_fft_scratch32,
 *  stencil_fft_min_taps, stencil_fft_epilogue_lds, render_graph, front_fast, ... -- see r2f_set_option in r2f_api.hip) */
R2F_API int r2f_set_option(r2f_ctx* ctx, const char* name, int value);

/* --- plan-only entry points: the host-side planners of this library (raw2film_amd/csrc/r2f_plan.cpp), callable without a GPU and
 * without a context.  Nothing upstream corresponds to them (the reference leaves these decisions to OpenCV and wgpu:
 * cv.filter2D picks its own DFT sizes, effects.py:151-153); they exist so that what the library decides on the CPU before it
 * launches anything can be pinned and fuzzed -- also under AddressSanitizer / UBSan, tests/test_plan_sanitizers.py -- on a machine
 * without a device. */
typedef struct r2f_fft_plan {
    int32_t ny, nx;          /* window rows, columns */
    int32_t vy, vx;          /* valid outputs per window */
    int32_t gx, ntiles;      /* windows per row of windows, windows per channel */
    int32_t pairs_per_channel, pairs;
    int32_t streams, batch, launches; /* internal streams, window pairs per launch triple, launch triples */
    uint64_t scratch_bytes;  /* pass scratch the context allocates for it */
} r2f_fft_plan;
/* The FFT form's plan for a bh x bw tap box on a call that covers `rows` output rows of a W-column frame with nch channels:
 * window shape (the cheapest under the stencil_fft_window / _rows / _max options, 0 = not forced), tiling and batches
 * (batch_mib MiB of scratch in flight on `streams` internal streams).  scratch_elem_bytes: 16, 8 or 12.  R2F_EINVAL when no window
 * shape fits. */
R2F_API int r2f_plan_fft(int bh, int bw, int W, int rows, int nch, int scratch_elem_bytes, int window, int window_rows, int window_max,
                 int batch_mib, int streams, r2f_fft_plan* out);
/* The direct form's device entry list for channel `channel` of a (kh, kw, kc) stencil on a TW x TH tile with Q rows per lane and
 * an LDS budget (0 = one phase): out8 = {entries, row steps, LDS phases, mirrored taps paired, cropped rows, cropped (padded)
 * columns, LDS row stride, rows of the largest phase}.  The list is checked against the taps it was built from (every tap exactly
 * once, offsets inside the phase's LDS rows): R2F_EHIP would mean a planner bug.  R2F_ETOOLARGE: a row step does not fit. */
R2F_API int r2f_plan_stencil(const float* host_khwc, int kh, int kw, int kc, int channel, int Q, int TW, int TH, int lds_budget_bytes,
                     int allow_sym, int* out8);
/* Tile order of a gx x gy grid of stencil workgroups (a permutation of 0 .. gx gy - 1; band = 0: automatic band width). */
R2F_API int r2f_plan_tile_order(int gx, int gy, int band, int* order);

#ifdef __cplusplus
}
#endif
#endif /* R2F_H */
