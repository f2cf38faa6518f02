// r2f_post.hip -- what sits after the path on the caller's side, kept on the device:
//   resize_area_u8     cpu_processor.py:411-412 -> utils.resolution_scaling -> cv.resize(uint8 canvas, INTER_AREA): the CPU
//                      processor's final shrink of the rendered (and canvas-framed) uint8 frame to the requested resolution
//   blit_rgba8         shaders/copy_to_int.wgsl (bound by gpu_processor.py:1416-1539): the display-referred float frame
//                      letterboxed into the preview widget's RGBA8 texture -- bilinear sample, canvas colour inside the canvas
//                      bounds, transparent outside
//   histogram_render   shaders/histogram.wgsl pass2_process + pass3_render and shaders/scale_texture.wgsl: the 3 x 256 counts
//                      (r2f_histogram_u8 = pass 1) -> log1p, 3-bin smoothing, bar heights -> the 256 x height RGBA bar image ->
//                      nearest-neighbour copy into the histogram widget's texture
// None of this is on the timed hot path; the kernels are one lane per output pixel.
#include "r2f_launch.h"

#include "../../include/r2f.h"

namespace r2f {

namespace {

// ---------------------------------------------------------------------------------------------------- INTER_AREA, uint8
// cv::resize(CV_8UC3, INTER_AREA), shrinking (imgproc/src/resize.cpp).  Integer scale factors take resizeAreaFast_: the
// integer sum of the scale_y x scale_x block, 2 x 2 as (s + 2) >> 2 (the SIMD form), anything else as
// saturate_cast<uchar>(sum * (1.f / area)).  Other factors take resizeArea_ with the DecimateAlpha tables of
// computeResizeAreaTab (float weights from double arithmetic): per source row a float buffer buf[dx] = sum_k S[sx_k] * alpha_k
// (k ascending, starting from 0), rows combined as sum = beta_0 * buf_0, sum += beta_j * buf_j, saturate_cast<uchar>(sum).
// Multiplications and additions are separate roundings (the generic C++ path has no FMA contraction).
__device__ __forceinline__ void area_tab(int d, double scale, int ssize, int& s_first, int& n, float& w_first, float& w_full, float& w_last,
                                         int& has_first, int& n_full, int& has_last) {
#pragma clang fp contract(off)
    // (separate roundings, like the host code this restates: a contracted d * scale + scale can land on the other side of an
    // integer.  HIP's __fmul_rn / __dadd_rn are plain operators the compiler is free to fuse; the pragma is what forbids it)
    // (plain operators: HIP's __dmul_rn / __fadd_rn wrappers are compiled with contraction allowed and fuse after inlining)
    const double f1 = (double)d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, (double)ssize - f1);
    int s1 = (int)ceil(f1), s2 = (int)floor(f2);
    s2 = min(s2, ssize - 1);
    s1 = min(s1, s2);
    has_first = ((double)s1 - f1 > 1e-3) ? 1 : 0;
    w_first = (float)(((double)s1 - f1) / cell);
    n_full = s2 - s1;
    w_full = (float)(1.0 / cell);
    has_last = (f2 - (double)s2 > 1e-3) ? 1 : 0;
    w_last = (float)(fmin(fmin(f2 - (double)s2, 1.0), cell) / cell);
    s_first = has_first ? s1 - 1 : s1;
    n = has_first + n_full + has_last;
}

__device__ __forceinline__ float area_w(int k, int has_first, int n_full, float w_first, float w_full, float w_last) {
    if (has_first && k == 0) return w_first;
    if (k - has_first < n_full) return w_full;
    return w_last;
}

__device__ __forceinline__ uint8_t sat_u8(float v) {  // saturate_cast<uchar>(float): cvRound (nearest even), then clamp
    const int r = __float2int_rn(v);
    return (uint8_t)min(max(r, 0), 255);
}

struct AreaU8Args {
    const uint8_t* src;
    uint8_t* dst;
    int H, W, out_h, out_w;
};

__global__ __launch_bounds__(256) void resize_area_u8_kernel(const AreaU8Args a) {
#pragma clang fp contract(off)
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= a.out_w || dy >= a.out_h) return;
    const double sx = (double)a.W / a.out_w, sy = (double)a.H / a.out_h;
    const int isx = (int)sx, isy = (int)sy;
    uint8_t* o = a.dst + ((long long)dy * a.out_w + dx) * 3;
    if ((double)isx == sx && (double)isy == sy) {  // resizeAreaFast_
        int sum[3] = {0, 0, 0};
        for (int y = 0; y < isy; ++y) {
            const uint8_t* row = a.src + ((long long)(dy * isy + y) * a.W + (long long)dx * isx) * 3;
            for (int x = 0; x < isx; ++x)
                for (int c = 0; c < 3; ++c) sum[c] += row[3 * x + c];
        }
        if (isx == 2 && isy == 2) {
            for (int c = 0; c < 3; ++c) o[c] = (uint8_t)((sum[c] + 2) >> 2);
        } else {
            const float scale = 1.f / (float)(isx * isy);
            for (int c = 0; c < 3; ++c) o[c] = sat_u8((float)sum[c] * scale);
        }
        return;
    }
    int x0, nx, hfx, nfx, hlx, y0, ny, hfy, nfy, hly;
    float wfx, wx, wlx, wfy, wy, wly;
    area_tab(dx, sx, a.W, x0, nx, wfx, wx, wlx, hfx, nfx, hlx);
    area_tab(dy, sy, a.H, y0, ny, wfy, wy, wly, hfy, nfy, hly);
    float sum[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < ny; ++j) {
        const uint8_t* row = a.src + ((long long)(y0 + j) * a.W + x0) * 3;
        float buf[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < nx; ++k) {
            const float alpha = area_w(k, hfx, nfx, wfx, wx, wlx);
            for (int c = 0; c < 3; ++c) {
                const float prod = (float)row[3 * k + c] * alpha;
                buf[c] = buf[c] + prod;
            }
        }
        const float beta = area_w(j, hfy, nfy, wfy, wy, wly);
        for (int c = 0; c < 3; ++c) {
            const float term = beta * buf[c];
            sum[c] = j == 0 ? term : sum[c] + term;
        }
    }
    for (int c = 0; c < 3; ++c) o[c] = sat_u8(sum[c]);
}

// ---------------------------------------------------------------------------------------------------- LANCZOS4, float32
// cv.resize(float32 frame, INTER_LANCZOS4): utils.resolution_scaling's other branch (utils.py:237-242), taken BEFORE the path
// when the preview is larger than the frame (cpu_processor.py:134).  cv::resize's generic path for CV_32F: float weights from
// interpolateLanczos4 (host tables, r2f_api.hip), HResizeLanczos4 = the 8 products of a row summed left to right, then
// VResizeLanczos4 = the 8 rows times beta summed top to bottom; indices outside the frame repeat the edge sample.
struct LanczosF32Args {
    const void* in;
    int in_layout, H, W;
    DevPlanes dst;
    int out_h, out_w;
    const int* xofs;    // out_w: source column of tap 3
    const float* xcoef;  // out_w x 8
    const int* yofs;
    const float* ycoef;
};

__device__ __forceinline__ void load_px(const void* in_, int layout, int H, int W, int y, int x, float (&v)[3]) {
    const float* in = static_cast<const float*>(in_);
    if (layout == R2F_LAYOUT_CHW) {
        const long long plane = (long long)H * W, o = (long long)y * W + x;
        v[0] = in[o], v[1] = in[plane + o], v[2] = in[2 * plane + o];
    } else {
        const float* p = in + ((long long)y * W + x) * (layout == R2F_LAYOUT_HWC4 ? 4 : 3);
        v[0] = p[0], v[1] = p[1], v[2] = p[2];
    }
}

__global__ __launch_bounds__(256) void lanczos4_f32_kernel(const LanczosF32Args a) {
#pragma clang fp contract(off)
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= a.out_w || dy >= a.out_h) return;
    const int sx = a.xofs[dx] - 3, sy = a.yofs[dy] - 3;
    float wx[8], wy[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) wx[k] = a.xcoef[dx * 8 + k], wy[k] = a.ycoef[dy * 8 + k];
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int yy = clampi(sy + k, 0, a.H - 1);
        float h[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v[3];
            load_px(a.in, a.in_layout, a.H, a.W, yy, clampi(sx + j, 0, a.W - 1), v);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float prod = v[c] * wx[j];
                h[c] = j == 0 ? prod : h[c] + prod;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float prod = h[c] * wy[k];
            acc[c] = k == 0 ? prod : acc[c] + prod;
        }
    }
    float* p0 = a.dst.data + (long long)(dy - a.dst.gy0) * a.out_w + dx;
    p0[0] = acc[0];
    p0[a.dst.plane_stride] = acc[1];
    p0[2 * a.dst.plane_stride] = acc[2];
}

// ---------------------------------------------------------------------------------------------------- preview blit
struct BlitArgs {
    const float* src;  // (H, W, 3) display-referred float
    uint8_t* dst;      // (dst_h, dst_w, 4)
    int H, W, dst_h, dst_w;
    r2f_blit t;
};

__device__ __forceinline__ uint8_t unorm8(float v) { return (uint8_t)__float2int_rn(fminf(fmaxf(v, 0.f), 1.f) * 255.f); }

// textureSampleLevel(linear, clamp-to-edge) at normalised uv: texel centres at (i + 0.5) / size
__device__ __forceinline__ void sample_bilinear(const float* src, int H, int W, float u, float v, float (&rgb)[3]) {
    const float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float tx = fx - x0f, ty = fy - y0f;
    const int x0 = clampi((int)x0f, 0, W - 1), x1 = clampi((int)x0f + 1, 0, W - 1);
    const int y0 = clampi((int)y0f, 0, H - 1), y1 = clampi((int)y0f + 1, 0, H - 1);
    const float* p00 = src + ((long long)y0 * W + x0) * 3;
    const float* p01 = src + ((long long)y0 * W + x1) * 3;
    const float* p10 = src + ((long long)y1 * W + x0) * 3;
    const float* p11 = src + ((long long)y1 * W + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = p00[c] + tx * (p01[c] - p00[c]), bot = p10[c] + tx * (p11[c] - p10[c]);
        rgb[c] = top + ty * (bot - top);
    }
}

__global__ __launch_bounds__(256) void blit_rgba8_kernel(const BlitArgs a) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= a.dst_w || y >= a.dst_h) return;
    const float cx = (float)x + 0.5f, cy = (float)y + 0.5f;
    const float u = (cx - a.t.offset_x) * a.t.scale_x, v = (cy - a.t.offset_y) * a.t.scale_y;
    uchar4 out = make_uchar4(0, 0, 0, 0);  // outside everything: transparent
    if (u >= 0.f && u <= 1.f && v >= 0.f && v <= 1.f) {
        float rgb[3];
        sample_bilinear(a.src, a.H, a.W, u, v, rgb);
        out = make_uchar4(unorm8(rgb[0]), unorm8(rgb[1]), unorm8(rgb[2]), 255);
    } else if (cx >= a.t.canvas_min_x && cx <= a.t.canvas_max_x && cy >= a.t.canvas_min_y && cy <= a.t.canvas_max_y) {
        out = make_uchar4(unorm8(a.t.canvas_color[0]), unorm8(a.t.canvas_color[1]), unorm8(a.t.canvas_color[2]), 255);
    }
    reinterpret_cast<uchar4*>(a.dst)[(long long)y * a.dst_w + x] = out;
}

// ---------------------------------------------------------------------------------------------------- histogram image
struct HistArgs {
    const uint32_t* counts;  // [3][256]
    uint8_t* image;          // (height, 256, 4)
    uint8_t* target;         // (th, tw, 4) or null
    int height, th, tw;
    uint32_t mix[8];         // the 2 x 2 x 2 colour table, RGBA packed little-endian, index is_r * 4 + is_g * 2 + is_b
};

// One workgroup of 256 lanes: histogram.wgsl pass2_process (float32 throughout, like the shader), then pass3_render for the
// whole 256 x height image, then scale_texture.wgsl into the widget texture.
__global__ __launch_bounds__(256) void histogram_render_kernel(const HistArgs a) {
    __shared__ float sh[3][256];
    __shared__ float red[256];
    __shared__ uint32_t heights[3][256];
    const int i = threadIdx.x;
    float v[3];
    for (int c = 0; c < 3; ++c) v[c] = (float)a.counts[c * 256 + i];
    auto block_max = [&](float mine) {
        red[i] = mine;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (i < s) red[i] = fmaxf(red[i], red[i + s]);
            __syncthreads();
        }
        const float m = red[0];
        __syncthreads();
        return m;
    };
    float m = block_max(fmaxf(v[0], fmaxf(v[1], v[2])));
    if (!(m > 0.f)) m = 1.f;
    for (int c = 0; c < 3; ++c) sh[c][i] = logf(1.0f + v[c] / m);
    __syncthreads();
    const int l = i == 0 ? i : i - 1, r = i == 255 ? i : i + 1;
    float s3[3];
    for (int c = 0; c < 3; ++c) s3[c] = (sh[c][l] + sh[c][i] + sh[c][r]) / 3.0f;
    float fm = block_max(fmaxf(s3[0], fmaxf(s3[1], s3[2])));
    if (fm == 0.f) fm = 1.f;
    for (int c = 0; c < 3; ++c) heights[c][i] = (uint32_t)((s3[c] * (float)a.height) / fm);
    __syncthreads();
    uint32_t* img = reinterpret_cast<uint32_t*>(a.image);
    const uint32_t hr = heights[0][i], hg = heights[1][i], hb = heights[2][i];
    for (int y = 0; y < a.height; ++y) {  // lane i = bin column i
        const uint32_t is_r = (uint32_t)y >= (uint32_t)a.height - hr, is_g = (uint32_t)y >= (uint32_t)a.height - hg,
                       is_b = (uint32_t)y >= (uint32_t)a.height - hb;
        img[y * 256 + i] = a.mix[is_r * 4 + is_g * 2 + is_b];
    }
    if (!a.target) return;
    __threadfence_block();
    __syncthreads();
    uint32_t* tgt = reinterpret_cast<uint32_t*>(a.target);
    for (long long p = i; p < (long long)a.th * a.tw; p += 256) {  // scale_texture.wgsl: nearest, uv = id / target size
        const int ty = (int)(p / a.tw), tx = (int)(p - (long long)ty * a.tw);
        const int sx = (int)(((float)tx / (float)a.tw) * 256.0f), sy = (int)(((float)ty / (float)a.th) * (float)a.height);
        tgt[p] = img[min(sy, a.height - 1) * 256 + min(sx, 255)];
    }
}

}  // namespace

namespace {

// The hand-off from RAW decoding, raw_to_linear's last two lines (raw_conversion.py:50-52) on LibRaw's 16-bit output:
// x = float(u) / 65535 (one correctly rounded fp32 division, like NumPy's), then x *= the float32 exposure factor, then the
// upload clamp of the GPU path (gpu_processor.py:275: min(x, 65504); the input cannot be negative).
// One lane = 4 pixels of an (n, ch) uint16 frame -> 12 floats of the (n, 3) float frame.
struct DecodeU16Args {
    const uint16_t* src;
    float* dst;
    long long n;  // pixels
    int ch;       // 3 or 4 (a fourth channel is dropped)
    float divisor, factor;
    int vec;      // 1: both buffers 16-byte aligned and ch == 3 -> 8-byte loads, 16-byte stores
};

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void decode_u16_kernel(const DecodeU16Args a) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;  // quad of pixels
    const long long p0 = q * 4;
    if (p0 >= a.n) return;
    if (a.vec && p0 + 4 <= a.n) {
        const uint2* s2 = reinterpret_cast<const uint2*>(a.src + p0 * 3);  // 24 bytes = 12 samples
        const uint2 w0 = s2[0], w1 = s2[1], w2 = s2[2];
        const unsigned w[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
        float f[12];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            f[2 * i] = fminf((float)(w[i] & 0xffffu) / a.divisor * a.factor, 65504.0f);
            f[2 * i + 1] = fminf((float)(w[i] >> 16) / a.divisor * a.factor, 65504.0f);
        }
        float4* d4 = reinterpret_cast<float4*>(a.dst + p0 * 3);
        d4[0] = make_float4(f[0], f[1], f[2], f[3]);
        d4[1] = make_float4(f[4], f[5], f[6], f[7]);
        d4[2] = make_float4(f[8], f[9], f[10], f[11]);
        return;
    }
    for (long long p = p0; p < min(p0 + 4, a.n); ++p)
        for (int c = 0; c < 3; ++c) a.dst[p * 3 + c] = fminf((float)a.src[p * a.ch + c] / a.divisor * a.factor, 65504.0f);
}

// Measurement aid (bench.py's `copy_ceiling`): a float4 streaming copy, 2 x `bytes` of HBM traffic -- what this chip moves when a
// kernel does nothing but load and store coalesced 16-byte lanes.  One float4 per lane, non-temporal both ways, one workgroup per
// 4 KB: the fastest of the shapes in tools/ubench/copy_rate.hip on MI355X (6.57 TB/s; plain loads / stores 6.23, four float4 per
// lane 5.84, a persistent grid-stride loop 4.6-4.8, hipMemcpyDtoD 5.10).
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(const f4v* __restrict__ src, f4v* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

}  // namespace

hipError_t launch_stream_copy(const void* src, void* dst, long long bytes, hipStream_t s) {
    const long long n = bytes / 16;
    launch_k(stream_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, static_cast<const f4v*>(src),
                       static_cast<f4v*>(dst), n);
    return take_launch_status();
}

hipError_t launch_decode_u16(const uint16_t* src, long long n, int ch, float divisor, float factor, float* dst, hipStream_t s) {
    DecodeU16Args a{src, dst, n, ch, divisor, factor,
                    (ch == 3 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) ? 1 : 0};
    const long long quads = (n + 3) / 4;
    launch_k(decode_u16_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a);
    return take_launch_status();
}

hipError_t launch_resize_area_u8(const uint8_t* src, int H, int W, uint8_t* dst, int out_h, int out_w, hipStream_t s) {
    AreaU8Args a{src, dst, H, W, out_h, out_w};
    launch_k(resize_area_u8_kernel, dim3((out_w + 63) / 64, (out_h + 3) / 4), dim3(64, 4), 0, s, a);
    return take_launch_status();
}

hipError_t launch_lanczos4_f32(const void* in, int in_layout, int H, int W, const DevPlanes& dst, int out_h, int out_w, const int* xofs,
                               const float* xcoef, const int* yofs, const float* ycoef, hipStream_t s) {
    LanczosF32Args a{in, in_layout, H, W, dst, out_h, out_w, xofs, xcoef, yofs, ycoef};
    launch_k(lanczos4_f32_kernel, dim3((out_w + 63) / 64, (out_h + 3) / 4), dim3(64, 4), 0, s, a);
    return take_launch_status();
}

hipError_t launch_blit_rgba8(const float* src, int H, int W, uint8_t* dst, int dst_h, int dst_w, const r2f_blit& t, hipStream_t s) {
    BlitArgs a{src, dst, H, W, dst_h, dst_w, t};
    launch_k(blit_rgba8_kernel, dim3((dst_w + 63) / 64, (dst_h + 3) / 4), dim3(64, 4), 0, s, a);
    return take_launch_status();
}

hipError_t launch_histogram_render(const uint32_t* counts, const uint8_t* mix_rgba, int height, uint8_t* image, uint8_t* target, int th,
                                   int tw, hipStream_t s) {
    HistArgs a;
    a.counts = counts, a.image = image, a.target = target, a.height = height, a.th = th, a.tw = tw;
    for (int k = 0; k < 8; ++k)
        a.mix[k] = (uint32_t)mix_rgba[4 * k] | ((uint32_t)mix_rgba[4 * k + 1] << 8) | ((uint32_t)mix_rgba[4 * k + 2] << 16) |
                   ((uint32_t)mix_rgba[4 * k + 3] << 24);
    launch_k(histogram_render_kernel, dim3(1), dim3(256), 0, s, a);
    return take_launch_status();
}

}  // namespace r2f
