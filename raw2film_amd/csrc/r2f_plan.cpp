// r2f_plan.cpp -- host-side planners of libr2f_hip.so (see r2f_plan.h).  No HIP in this file: it is compiled by hipcc into the
// library and by g++ -fsanitize=address,undefined into the fuzz harness (tests/plan_fuzz.cpp).
#include "r2f_plan.h"

#include "../../include/r2f.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace r2f {
namespace plan {

// ------------------------------------------------------------------------------------------------ stencil taps
void tap_box(const Taps& t, int c, int box[4]) {
    int i_lo = t.kh, i_hi = -1, j_lo = t.kw, j_hi = -1;
    for (int i = 0; i < t.kh; ++i)
        for (int j = 0; j < t.kw; ++j)
            if (t.at(i, j, c) != 0.f) {
                i_lo = std::min(i_lo, i), i_hi = std::max(i_hi, i);
                j_lo = std::min(j_lo, j), j_hi = std::max(j_hi, j);
            }
    if (i_hi < 0) i_lo = i_hi = t.kh / 2, j_lo = j_hi = t.kw / 2;  // all-zero stencil: keep one (zero) tap
    box[0] = i_lo, box[1] = i_hi, box[2] = j_lo, box[3] = j_hi;
}

bool single_tap_channel(const Taps& t, int c, float* w) {
    int tb[4];
    tap_box(t, c, tb);
    if (!(tb[0] == tb[1] && tb[2] == tb[3] && tb[0] == t.kh / 2 && tb[2] == t.kw / 2)) return false;
    if (w) *w = t.at(tb[0], tb[2], c);
    return true;
}

bool mirror_symmetric(const Taps& t, int c, const int box[4]) {
    const int i_lo = box[0], j_lo = box[2], bh = box[1] - box[0] + 1, bw = box[3] - box[2] + 1;
    if (!(bw % 2 == 1 && bw >= 9 && t.kw / 2 - j_lo == (bw - 1) / 2)) return false;
    for (int i = 0; i < bh; ++i)
        for (int j = 0; j < bw / 2; ++j) {
            const float a = t.at(i + i_lo, j_lo + j, c), b = t.at(i + i_lo, j_lo + bw - 1 - j, c);
            if (memcmp(&a, &b, sizeof a) != 0) return false;
        }
    return true;
}

// Flatten one channel of a stencil into the entry list of stencil_accumulate<Q> (layout in r2f_device.h).  Taps are cropped to
// the bounding box; per input-row step m only the 4-tap chunks between the first and last chunk holding a non-zero tap of any
// of the Q kernel rows m - q are emitted (the halation disc skips its empty corners this way).  Row steps are grouped into
// phases of at most `mp` steps; LDS offsets are relative to the phase.
// `tap(i, j)` = weight of the (virtual) cropped stencil, 0 outside; kh x kw virtual taps.
// sym: kw = 2 r + 1 with r even, mirror symmetric; entries cover columns 0 .. r, centre column at half weight.
template <class Tap>
static void build_stream(Tap tap, int kh, int kw, bool sym, int Q, int RS, int TH, int mp, StreamHost& out) {
    const int r = (kw - 1) / 2;
    const int ncols = sym ? r + 1 : kw;  // columns that own entries
    const int nch = (ncols + 3) / 4;
    const int M = kh + Q - 1;
    out = StreamHost();
    auto wt = [&](int i, int j) -> float {
        if (!sym) return tap(i, j);
        if (j > r) return 0.f;
        return j == r ? 0.5f * tap(i, j) : tap(i, j);
    };
    if (mp < 1) mp = 1;
    for (int m0 = 0; m0 < M; m0 += mp) {
        const int m1 = std::min(M, m0 + mp);
        const int lds_rows = TH - Q + (m1 - m0);
        out.phases.push_back(m0);
        out.phases.push_back(lds_rows);
        out.phases.push_back((int)out.rowinfo.size() / 4);
        out.phases.push_back(out.n_entries);
        out.max_lds_rows = std::max(out.max_lds_rows, lds_rows);
        ++out.n_phases;
        for (int m = m0; m < m1; ++m) {
            int c_lo = nch, c_hi = -1;
            for (int c = 0; c < nch; ++c) {
                bool nz = false;
                for (int q = 0; q < Q && !nz; ++q)
                    for (int t = 0; t < 4; ++t)
                        if (wt(m - q, 4 * c + t) != 0.f) {
                            nz = true;
                            break;
                        }
                if (nz) {
                    if (c < c_lo) c_lo = c;
                    c_hi = c;
                }
            }
            if (c_hi < 0) continue;  // no work on this row step
            // live tap columns of the first and of the last entry (bit t: some row of column 4c+t is non-zero)
            auto live = [&](int c) {
                int bits = 0;
                for (int t = 0; t < 4; ++t)
                    for (int q = 0; q < Q; ++q)
                        if (wt(m - q, 4 * c + t) != 0.f) bits |= 1 << t;
                return bits;
            };
            out.rowinfo.push_back(c_hi - c_lo + 1);
            out.rowinfo.push_back((m - m0) * RS + 4 * c_lo);
            out.rowinfo.push_back(sym ? (m - m0) * RS + 2 * r - 4 * c_lo - 4 : 0);
            out.rowinfo.push_back(live(c_lo) | live(c_hi) << 4);
            out.mask_first_or |= live(c_lo);
            out.mask_last_or |= live(c_hi);
            for (int c = c_lo; c <= c_hi; ++c) {
                ++out.n_entries;
                for (int t = 0; t < 4; ++t)
                    for (int q = 0; q < Q; ++q) out.w.push_back(wt(m - q, 4 * c + t));
            }
        }
    }
    out.n_rowsteps = (int)out.rowinfo.size() / 4;
    // terminator phase record: {., ., n_rowsteps, n_entries}
    out.phases.push_back(0);
    out.phases.push_back(0);
    out.phases.push_back(out.n_rowsteps);
    out.phases.push_back(out.n_entries);
    for (int d = 0; d < 2; ++d) {  // two dummy entries and row-step records: targets of the last prefetches
        for (int i = 0; i < 4; ++i) out.rowinfo.push_back(0);
        for (int i = 0; i < 4 * Q; ++i) out.w.push_back(0.f);
    }
}

int plan_stencil_channel(const Taps& t, int c, const int box[4], bool sym, int Q, int TW, int TH, size_t lds_budget, StencilGeom* g,
                         StreamHost* sh) {
    const int i_lo = box[0], j_lo0 = box[2];
    const int bh = box[1] - box[0] + 1, bw = box[3] - box[2] + 1;
    // virtual stencil: the cropped box widened by zero columns on both sides until r = 2 (mod 4).  Even r keeps the mirrored
    // block 16-byte aligned; r + 1 = 3 (mod 4) puts the one padded column of the left half next to the centre and the slack at
    // the OUTER edge, where whole chunks are empty on most rows and get skipped.
    const int pad = sym ? ((2 - ((bw - 1) / 2) % 4) + 4) % 4 : 0;
    const int vkw = bw + 2 * pad, vkh = bh;
    auto tap = [&](int i, int j) -> float {
        j -= pad;
        if (i < 0 || i >= bh || j < 0 || j >= bw) return 0.f;
        return t.at(i + i_lo, j + j_lo0, c);
    };
    g->kh = vkh;
    g->kw = vkw;
    g->kw_pad = (vkw + 3) / 4 * 4;
    g->RS = TW + g->kw_pad;
    g->ay = t.kh / 2 - i_lo;  // anchor (kh/2, kw/2): convolution.wgsl:31, cv.filter2D default
    g->ax = t.kw / 2 - j_lo0 + pad;
    g->sym = sym ? 1 : 0;
    const int M = g->kh + Q - 1;
    int mp = M;
    if (lds_budget) {
        const long long fit = ((long long)(lds_budget / sizeof(float)) - 16) / g->RS;  // rows of RS floats (+ 16 floats of slack)
        mp = (int)std::min<long long>(M, fit - (TH - Q));
        if (mp < 1) return -3;
        // equalise the phases instead of leaving a short last one
        const int nph = (M + mp - 1) / mp;
        mp = (M + nph - 1) / nph;
    }
    build_stream(tap, vkh, vkw, sym, Q, g->RS, TH, mp, *sh);
    g->n_phases = sh->n_phases;
    g->n_rowsteps = sh->n_rowsteps;
    g->n_entries = sh->n_entries;
    g->mask_first_or = sh->mask_first_or;
    g->mask_last_or = sh->mask_last_or;
    g->max_lds_rows = sh->max_lds_rows;
    return 0;
}

int fixed_stencil_radius(const Taps& t, const StencilGeom* geom, const int* chans, int nch, int max_r) {
    if (nch <= 0) return 0;
    int b[4];
    tap_box(t, chans[0], b);
    const int n = b[1] - b[0] + 1, R = n / 2;
    if (!((n & 1) && R >= 1 && R <= max_r && b[3] - b[2] + 1 == n && b[0] + R == t.kh / 2 && b[2] + R == t.kw / 2)) return 0;
    for (int i = 0; i < nch; ++i) {
        int o[4];
        tap_box(t, chans[i], o);
        const StencilGeom& d = geom[chans[i]];
        // the device form pairs mirrored taps (and pads the box) from 9 x 9 up; below that the box is built as it is
        if (memcmp(o, b, sizeof b) || d.sym != (R >= 4 ? 1 : 0) || d.ay != R || d.ax != fixed_ax(R) || d.kh != n || d.n_phases != 1)
            return 0;
        for (int y = 0; y < n; ++y)  // left-right mirror symmetric, bit for bit
            for (int x = 0; x < R; ++x) {
                const float l = t.at(b[0] + y, b[2] + x, chans[i]), r = t.at(b[0] + y, b[2] + 2 * R - x, chans[i]);
                if (memcmp(&l, &r, sizeof l)) return 0;
            }
    }
    return R;
}

std::vector<float> fixed_stencil_weights(const Taps& t, int R, int Q, bool* same) {
    const int n = 2 * R + 1, per = (2 * R + Q) * (R + 1) * (Q / 2);
    std::vector<float> w((size_t)3 * per * 2, 0.f);
    int ref = -1;
    *same = true;
    for (int c = 0; c < 3; ++c) {
        int b[4];
        tap_box(t, c, b);
        if (b[1] - b[0] + 1 != n || b[3] - b[2] + 1 != n) {
            *same = false;
            continue;
        }
        auto tap = [&](int i, int j) { return i >= 0 && i <= 2 * R ? t.at(b[0] + i, b[2] + j, c) : 0.f; };
        for (int i = 0; i < 2 * R + Q; ++i)
            for (int j = 0; j <= R; ++j)
                for (int h = 0; h < Q / 2; ++h) {
                    float* pair = &w[((size_t)c * per + ((size_t)i * (R + 1) + j) * (Q / 2) + h) * 2];
                    pair[0] = tap(i - 2 * h, j);      // output row 2 h of the lane
                    pair[1] = tap(i - 2 * h - 1, j);  // output row 2 h + 1
                }
        if (ref < 0)
            ref = c;
        else
            *same = *same && !memcmp(&w[(size_t)c * per * 2], &w[(size_t)ref * per * 2], (size_t)per * 2 * sizeof(float));
    }
    return w;
}

// Separable?  K[i][j] = u[i] v[j] with u = the centre column and v = the centre row / K[R][R]: accepted when the rank-1 form
// reproduces every tap to 6e-7 of the largest one (u_i v_j rebuilt from fp32 taps carries ~5 roundings of 6e-8), so the field
// differs from the full stencil's by < 1e-6 -- far inside the 4e-6 the hardware transcendentals of the noise are worth -- and
// 2 (2 R + 1) taps per pixel replace (2 R + 1)^2.
bool separable_taps(const Taps& t, int R, float u[3][19], float v[3][10]) {
    if (R < 1 || R > 9) return false;
    const int n = 2 * R + 1;
    for (int c = 0; c < 3; ++c) {
        int b[4];
        tap_box(t, c, b);
        if (b[1] - b[0] + 1 != n || b[3] - b[2] + 1 != n) return false;
        auto K = [&](int i, int j) { return (double)t.at(b[0] + i, b[2] + j, c); };
        const double centre = K(R, R);
        double kmax = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) kmax = std::max(kmax, std::fabs(K(i, j)));
        if (!(centre > 0.0) || centre < 0.25 * kmax) return false;  // a rank-1 form anchored on a small centre tap is ill-conditioned
        for (int i = 0; i < n; ++i) u[c][i] = (float)K(i, R);
        for (int j = 0; j <= R; ++j) v[c][j] = (float)(K(R, j) / centre);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const double vv = (double)v[c][j <= R ? j : 2 * R - j];
                if (std::fabs((double)u[c][i] * vv - K(i, j)) > 6e-7 * kmax) return false;
            }
    }
    return true;
}

void stencil_source_rows(int y0, int y1, int above, int below, int H, int* lo_out, int* hi_out) {
    int lo = y0 - above, hi = y1 - 1 + below;  // inclusive
    int need_lo = std::max(lo, 0), need_hi = std::min(hi, H - 1);
    if (H > 1) {
        if (lo < 0) need_hi = std::max(need_hi, std::min(-lo, H - 1));
        if (hi > H - 1) need_lo = std::min(need_lo, std::max(2 * (H - 1) - hi, 0));
    }
    *lo_out = need_lo;
    *hi_out = need_hi + 1;
}

// ------------------------------------------------------------------------------------------------ FFT form
// Window shape for a bh x bw tap box: of {256, 512} rows x {256, 512, 1024} columns the one whose three passes move the fewest
// scratch bytes over a W-column frame of H rows (an 87-tap disc keeps 44 % of a 256 x 256 window and 61 % of a 256 x 1024 one).
// H is the number of rows the caller wants covered: the whole frame for a whole-frame render, the rows of the call for a row
// shard (a 1 058-row call is 6.15 window rows of 172: the cost of the seventh counts).  `window` / `window_rows` force an axis
// (ignored for a box over 200 taps on that axis, which needs the 512-point window).
bool fft_window(const FftOptions& o, int bh, int bw, int W, int H, bool s32, int* ny, int* nx) {
    double best = -1.0;
    // a box wider than 200 columns needs 512 columns at least, whatever window_max says (with the accepted value 256
    // every candidate used to be rejected and the caller's 256 x 256 default went on to a division by zero)
    const int window_max = bw > 200 ? std::max(o.window_max, 512) : o.window_max;
    for (int y = 256; y <= 512; y *= 2) {
        if (bh > 200 ? y != 512 : (o.window_rows && y != o.window_rows)) continue;
        for (int x = 256; x <= 1024; x *= 2) {
            if (x > window_max && x != o.window) continue;
            if (bw > 200 ? (x < 512 || (o.window >= 512 && x != o.window)) : (o.window && x != o.window)) continue;
            const int vy = y - bh + 1, vx = (x - bw + 1) & ~3;
            if (vy < 1 || vx < 4) continue;  // the window has to keep outputs (and vx, vy divide below)
            const double n = (double)y * x, part = n * vy / y;
            // per window: pass 1 (floats in, image out), pass 2 (image in, valid rows out), pass 3 (valid rows in, floats out);
            // two windows share one complex image.  The 512-row pass 2 moves its bytes ~1.3 x slower (r2f_fft.hip).
            const double p2 = y == 512 ? 1.3 : 1.0;
            const double e = s32 ? 4.0 : 8.0;  // scratch bytes per window element (half a complex64 / complex128)
            const double bytes = 4.0 * n + e * n + p2 * (e * n + e * part) + e * part + 4.0 * vy * vx;
            const double cost = (double)((W + vx - 1) / vx) * ((H + vy - 1) / vy) * bytes;
            if (best < 0.0 || cost < best) best = cost, *ny = y, *nx = x;
        }
    }
    return best >= 0.0;
}

FftBatches fft_batches(const FftOptions& o, int ny, int nx, int bh, int bw, int W, int y0, int y1, int nch, int elem_bytes) {
    FftBatches b;
    b.ny = ny, b.nx = nx;
    b.vy = ny - bh + 1;
    b.vx = (nx - bw + 1) & ~3;  // a multiple of 4: window origins stay 16-byte aligned for the float4 stores of pass 3
    if (b.vy < 1 || b.vx < 4 || nch < 1 || y1 <= y0 || W < 1) return b;
    b.gx = (W + b.vx - 1) / b.vx;
    b.ntiles = b.gx * ((y1 - y0 + b.vy - 1) / b.vy);
    b.ppc = (b.ntiles + 1) / 2;
    b.pairs = b.ppc * nch;
    b.img_bytes = (size_t)ny * nx * (size_t)elem_bytes;
    // batches alternate between the internal streams when there is enough work for that to matter; batch_mib counts MiB of
    // scratch in flight (a 256 x 256 complex128 pair is 1 MiB)
    const int fft_batch = std::max(1, (int)(((size_t)std::max(o.batch_mib, 1) << 20) / b.img_bytes));
    b.nstreams = b.pairs > fft_batch ? std::max(1, std::min(o.streams, 4)) : 1;
    b.batch = std::min(b.pairs, std::max(1, fft_batch / b.nstreams));
    if (o.even) {
        // as many launch triples per stream: 240 pairs in batches of 48 would be 3 + 2 launches on the two streams, and the
        // stream with two idles for a fifth of the stage; 6 batches of 40 keep both busy
        int nb = (b.pairs + b.batch - 1) / b.batch;
        nb = (nb + b.nstreams - 1) / b.nstreams * b.nstreams;
        b.batch = (b.pairs + nb - 1) / nb;
    }
    b.launches = (b.pairs + b.batch - 1) / b.batch;
    b.scratch_bytes = (size_t)b.batch * b.nstreams * b.img_bytes;
    return b;
}

// ------------------------------------------------------------------------------------------------ tables
// Linear workgroup ids are dealt round-robin to the 8 XCDs (each with its own 4 MB L2), so XCD x runs ids x, x + 8, ...  Give
// each XCD one contiguous row-major run of tiles (balanced to one tile), and let it walk that run in bands of `b` tile columns,
// top to bottom: the ~64 tiles an XCD has in flight then form a compact block whose halo rows AND columns are re-read from
// that L2 instead of from HBM.
std::vector<int> tile_order(int gx, int gy, int band) {
    const int nwg = gx * gy, q = nwg / 8, r = nwg % 8;
    std::vector<int> order((size_t)std::max(nwg, 0));
    std::vector<int> tiles;
    for (int x = 0; x < 8; ++x) {
        const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, len = q + (x < r ? 1 : 0);
        if (len == 0) continue;
        const int rows = (start + len - 1) / gx - start / gx + 1;
        int b = std::max(1, std::min(gx, (64 + rows / 2) / rows));  // 32 CUs x 2 workgroups in flight per XCD
        if (band > 0) b = std::min(gx, band);
        tiles.resize((size_t)len);
        for (int i = 0; i < len; ++i) tiles[(size_t)i] = start + i;
        std::stable_sort(tiles.begin(), tiles.end(), [&](int a, int c) {
            const int ba = (a % gx) / b, bc = (c % gx) / b;
            return ba != bc ? ba < bc : a < c;  // band, then row-major inside the band
        });
        for (int i = 0; i < len; ++i) order[(size_t)i * 8 + x] = tiles[(size_t)i];
    }
    return order;
}

// cv::interpolateLanczos4 (imgproc/src/resize.cpp), float / double mixed exactly as there.
void lanczos4_coeffs(float x, float* coeffs) {
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    const double kPi = 3.1415926535897932384626433832795;
    float sum = 0;
    const double y0 = -(x + 3) * kPi * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
    for (int i = 0; i < 8; i++) {
        const float y0_ = (x + 3 - i);
        if (std::fabs(y0_) >= 1e-6f) {
            const double y = -y0_ * kPi * 0.25;
            coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        } else {
            coeffs[i] = 1e30f;  // x ~ 0 or ~ 1: this tap takes everything after the normalisation
        }
        sum += coeffs[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) coeffs[i] *= sum;
}

// cv::resize's per-destination tables for INTER_LANCZOS4 on CV_8U: source index of tap 3 and eight weights in 11-bit fixed
// point (saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE), round half to even).
int lanczos4_table_u8(int ssize, int dsize, int* ofs, short* coef) {
    if (ssize <= 0 || dsize <= 0 || !ofs || !coef) return -1;
    const double scale = 1. / ((double)dsize / ssize);
    for (int d = 0; d < dsize; ++d) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        const int sx = (int)std::floor(fx);
        fx -= sx;
        ofs[d] = sx;
        float cbuf[8];
        lanczos4_coeffs(fx, cbuf);
        for (int k = 0; k < 8; ++k) {
            // (clamped in float first: the tap that "takes everything" is 1e30 / sum before the normalisation rounds it to 1)
            const float scaled = std::min(std::max(cbuf[k] * 2048.f, -32768.f), 32767.f);
            coef[(size_t)d * 8 + k] = (short)std::lrintf(scaled);
        }
    }
    return 0;
}

// cv::resize's tables for INTER_LANCZOS4 on CV_32F: the same source index and interpolateLanczos4 weights, kept as floats.
int lanczos4_table_f32(int ssize, int dsize, int* ofs, float* coef) {
    if (ssize <= 0 || dsize <= 0 || !ofs || !coef) return -1;
    const double scale = 1. / ((double)dsize / ssize);
    for (int d = 0; d < dsize; ++d) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        const int sx = (int)std::floor(fx);
        fx -= sx;
        ofs[d] = sx;
        lanczos4_coeffs(fx, coef + 8 * (size_t)d);
    }
    return 0;
}

bool chroma_weights(int size, float* w) {
    // gaussian_kernel_1d(2*size+1, 0.3*((taps-1)/2 - 1) + 0.8), effects.py:421-435,554-556: exp in double, float32 taps
    // normalised by their float32 sum
    if (size < 1 || 2 * size + 1 > kChromaMaxTaps) return false;
    const int taps = 2 * size + 1;
    const double sigma = 0.3 * ((taps - 1) * 0.5 - 1) + 0.8, s2 = 2.0 * sigma * sigma;
    float sum = 0.f;
    for (int i = 0; i < taps; ++i) {
        const double x = i - size;
        w[i] = (float)std::exp(-(x * x) / s2);
    }
    // numpy's float32 .sum() is pairwise; for <= 63 elements it reduces to 8 interleaved partial sums -- restate that order
    {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int i = 0;
        if (taps >= 8) {
            for (int j = 0; j < 8; ++j) acc[j] = w[j];
            for (i = 8; i + 8 <= taps; i += 8)
                for (int j = 0; j < 8; ++j) acc[j] += w[i + j];
            sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        }
        for (; i < taps; ++i) sum += w[i];
    }
    for (int i = 0; i < taps; ++i) w[i] /= sum;
    return true;
}

void burn_weights(double* w) {
    // scipy.ndimage._filters._gaussian_kernel1d(sigma=3, order=0, radius=int(truncate*sigma + 0.5) = 6)
    double sum = 0.0;
    for (int t = 0; t < 13; ++t) {
        const double x = t - 6;
        w[t] = std::exp(-0.5 / 9.0 * x * x);
        sum += w[t];
    }
    for (int t = 0; t < 13; ++t) w[t] /= sum;
}

// (4, m) table -> per channel m-1 cells {xp[i], xp[i+1], fp[i], slope[i]}; slopes in double like np.interp.
int curve_cells(const float* lut, int m, CurveCells* cv) {
    if (!lut || m < 2) return -1;
    for (int i = 0; i + 1 < m; ++i)
        if (!(lut[i + 1] >= lut[i])) return -1;  // xp must be non-decreasing (NaNs fail here too)
    cv->cells.assign((size_t)3 * (m - 1) * 4, 0.f);
    for (int c = 0; c < 3; ++c) {
        const float* fp = lut + (size_t)(1 + c) * m;
        for (int i = 0; i + 1 < m; ++i) {
            const double dx = (double)lut[i + 1] - (double)lut[i];
            const float slope = dx != 0.0 ? (float)(((double)fp[i + 1] - (double)fp[i]) / dx) : 0.f;
            float* cell = &cv->cells[((size_t)c * (m - 1) + i) * 4];
            cell[0] = lut[i], cell[1] = lut[i + 1], cell[2] = fp[i], cell[3] = slope;
        }
        cv->f_first[c] = fp[0];
        cv->f_last[c] = fp[m - 1];
    }
    cv->m = m;
    cv->x0 = lut[0];
    cv->x1 = lut[m - 1];
    const float range = lut[m - 1] - lut[0];
    cv->inv_step = range > 0.f ? (float)(m - 1) / range : 0.f;
    // `near`: is the device's first guess (same float32 arithmetic) within one cell of the true cell for every x?  Both
    // are monotone step functions of x, so it is enough to look at each breakpoint and at the float just below it.
    cv->near = 1;
    const int last = m - 2;
    auto guess = [&](float x) {
        // (the product is clamped in float before the conversion: a huge or non-finite value must not reach the int cast)
        const float gf = (x - cv->x0) * cv->inv_step;
        if (!(gf > 0.f)) return 0;
        return gf >= (float)last ? last : (int)gf;
    };
    for (int k = 1; k + 1 < m && cv->near; ++k) {
        if (lut[k] == lut[k - 1] || lut[k] == lut[k + 1]) cv->near = 0;  // repeated abscissae: keep the exact walk
        const int at = guess(lut[k]), below = guess(std::nextafterf(lut[k], -INFINITY));
        const int t_at = k > last ? last : k, t_below = k - 1;
        if (std::abs(at - t_at) > 1 || std::abs(below - t_below) > 1) cv->near = 0;
    }
    return 0;
}

size_t plane_set_floats(int H, int W) { return ((size_t)H * W + 3) / 4 * 4 * 3; }

bool burn_geometry(int burn_cell, int H, int W, int* h_lo, int* w_lo) {
    if (burn_cell < 1) return false;
    *h_lo = H / burn_cell;
    *w_lo = W / burn_cell;
    return *h_lo >= 1 && *w_lo >= 1;
}

size_t workspace_floats(unsigned flags, int burn_cell, int H, int W) {
    if (H <= 0 || W <= 0) return 0;
    const bool hal = flags & 2u, mtf = flags & 4u, grain = flags & 8u, burn = flags & 32u;  // R2F_F_* of include/r2f.h
    int sets = 0;
    if (hal || mtf || grain || burn) sets = 1;
    if (hal || mtf || (grain && burn)) sets = 2;
    size_t burn_floats = 0;
    int h_lo, w_lo;
    if (burn && burn_geometry(burn_cell, H, W, &h_lo, &w_lo)) burn_floats = ((size_t)4 * h_lo * w_lo + 3) / 4 * 4;  // sums + map + 2 x scratch
    return (size_t)sets * plane_set_floats(H, W) + burn_floats;
}

}  // namespace plan
}  // namespace r2f

// ------------------------------------------------------------------------------------------------ plan-only C ABI (include/r2f.h)
namespace {

// Does the entry list reproduce the taps it was built from?  Every non-zero tap of the cropped box exactly once per output row of
// a lane (the centre column of a mirrored list at half weight), every LDS offset inside its phase's rows.
#ifdef R2F_PLAN_DEBUG
static bool fail_at(int n) { fprintf(stderr, "stream_consistent: check %d failed\n", n); return false; }
#else
static inline bool fail_at(int) { return false; }
#endif

bool stream_consistent(const r2f::plan::Taps& t, int c, const int box[4], const r2f::plan::StencilGeom& g, const r2f::plan::StreamHost& sh, int Q,
                       int TH) {
    using namespace r2f::plan;
    const int bh = box[1] - box[0] + 1, bw = box[3] - box[2] + 1;
    const int pad = (g.kw - bw) / 2, r = (g.kw - 1) / 2;
    if (g.kh != bh || g.kw < bw || (g.kw - bw) % 2) return fail_at(1);
    const int M = g.kh + Q - 1;
    // (a tap K[i][j] is used once per output row q of a lane: in row step m = i + q)
    std::vector<float> seen((size_t)g.kh * g.kw * Q, 0.f);
    std::vector<char> hit((size_t)g.kh * g.kw * Q, 0);
    if ((int)sh.phases.size() != 4 * (sh.n_phases + 1) || (int)sh.rowinfo.size() != 4 * (sh.n_rowsteps + 2)) return fail_at(2);
    if ((int)sh.w.size() != 4 * Q * (sh.n_entries + 2)) return fail_at(3);
    int entry = 0, rowstep = 0;
    for (int ph = 0; ph < sh.n_phases; ++ph) {
        const int m0 = sh.phases[4 * ph], lds_rows = sh.phases[4 * ph + 1];
        if (sh.phases[4 * ph + 2] != rowstep || sh.phases[4 * ph + 3] != entry) return fail_at(4);
        const int rs_end = sh.phases[4 * (ph + 1) + 2];
        if (lds_rows > sh.max_lds_rows || lds_rows < TH - Q + 1) return fail_at(5);
        // the row steps of this phase, in increasing m: recover m from the LDS offset
        for (; rowstep < rs_end; ++rowstep) {
            const int n = sh.rowinfo[4 * rowstep], off = sh.rowinfo[4 * rowstep + 1], offr = sh.rowinfo[4 * rowstep + 2];
            if (n < 1) return fail_at(6);
            const int mrel = off / g.RS, c_lo = (off - mrel * g.RS) / 4;
            const int m = m0 + mrel;
            if (m < m0 || m >= M || (off - mrel * g.RS) % 4) return fail_at(7);
            // a lane reads 8 floats from `off + 4 e` (and from the mirrored block) for tile column 0 .. TW - 4: inside the phase's rows
            if (mrel + (TH - Q) >= lds_rows) return fail_at(8);
            // ... and inside the row: an entry's 4-tap chunk (plus the lane's 4 pixels) ends at TW + 4 (c_lo + e) + 7 <= RS - 1
            if (4 * (c_lo + n) > (g.sym ? (r + 1 + 3) / 4 * 4 : g.kw_pad)) return fail_at(9);
            // the mirrored 8-float block of entry e starts at 2 r - 4 (c_lo + e) - 4 >= 0
            if (g.sym && (offr != mrel * g.RS + 2 * r - 4 * c_lo - 4 || 2 * r - 4 * (c_lo + n - 1) - 4 < 0)) return fail_at(10);
            for (int e = 0; e < n; ++e, ++entry)
                for (int tt = 0; tt < 4; ++tt)
                    for (int q = 0; q < Q; ++q) {
                        const float w = sh.w[((size_t)entry * 4 + tt) * Q + q];
                        const int i = m - q, j = 4 * (c_lo + e) + tt;
                        if (w == 0.f) continue;
                        if (i < 0 || i >= g.kh || j < 0 || j >= g.kw) return fail_at(11);
                        if (hit[((size_t)i * g.kw + j) * Q + q]) return fail_at(12);
                        hit[((size_t)i * g.kw + j) * Q + q] = 1;
                        seen[((size_t)i * g.kw + j) * Q + q] = w;
                    }
        }
    }
    if (entry != sh.n_entries || rowstep != sh.n_rowsteps) return fail_at(13);
    for (int i = 0; i < g.kh; ++i)
        for (int j = 0; j < g.kw; ++j) {
            const int bj = j - pad;
            float want = (bj >= 0 && bj < bw) ? t.at(box[0] + i, box[2] + bj, c) : 0.f;
            if (g.sym) want = j > r ? 0.f : (j == r ? 0.5f * want : want);
            for (int q = 0; q < Q; ++q) {
                const float got = seen[((size_t)i * g.kw + j) * Q + q];
                if (memcmp(&want, &got, sizeof want) && !(want == 0.f && got == 0.f)) return fail_at(14);
            }
        }
    return true;
}

}  // namespace

extern "C" {

int r2f_plan_fft(int bh, int bw, int W, int rows, int nch, int scratch_elem_bytes, int window, int window_rows, int window_max,
                 int batch_mib, int streams, r2f_fft_plan* out) {
    using namespace r2f::plan;
    if (!out || bh < 1 || bw < 1 || bh > kFftMaxTaps || bw > kFftMaxTaps || W < 1 || rows < 1 || nch < 1 || nch > 3) return R2F_EINVAL;
    if (scratch_elem_bytes != 16 && scratch_elem_bytes != 8 && scratch_elem_bytes != 12) return R2F_EINVAL;
    if ((window != 0 && window != 256 && window != 512 && window != 1024) || (window_rows != 0 && window_rows != 256 && window_rows != 512) ||
        (window_max != 256 && window_max != 512 && window_max != 1024) || batch_mib < 1 || streams < 1 || streams > 4)
        return R2F_EINVAL;
    FftOptions o;
    o.window = window, o.window_rows = window_rows, o.window_max = window_max, o.batch_mib = batch_mib, o.streams = streams;
    int ny = 0, nx = 0;
    if (!fft_window(o, bh, bw, W, rows, scratch_elem_bytes == 8, &ny, &nx)) return R2F_EINVAL;
    const FftBatches b = fft_batches(o, ny, nx, bh, bw, W, 0, rows, nch, scratch_elem_bytes);
    out->ny = b.ny, out->nx = b.nx, out->vy = b.vy, out->vx = b.vx, out->gx = b.gx, out->ntiles = b.ntiles;
    out->pairs_per_channel = b.ppc, out->pairs = b.pairs, out->streams = b.nstreams, out->batch = b.batch, out->launches = b.launches;
    out->scratch_bytes = b.scratch_bytes;
    return R2F_OK;
}

int r2f_plan_stencil(const float* k, int kh, int kw, int kc, int channel, int Q, int TW, int TH, int lds_budget_bytes, int allow_sym,
                     int* out8) {
    using namespace r2f::plan;
    if (!k || !out8 || kh < 1 || kw < 1 || (kc != 1 && kc != 3) || channel < 0 || channel > 2) return R2F_EINVAL;
    if ((Q != 2 && Q != 4) || TW < 4 || TW % 4 || TH < Q || TH % Q || lds_budget_bytes < 0) return R2F_EINVAL;
    const Taps t{k, kh, kw, kc};
    int box[4];
    tap_box(t, channel, box);
    const bool sym = allow_sym && mirror_symmetric(t, channel, box);
    StencilGeom g;
    StreamHost sh;
    const int rc = plan_stencil_channel(t, channel, box, sym, Q, TW, TH, (size_t)lds_budget_bytes, &g, &sh);
    if (rc) return R2F_ETOOLARGE;
    if (!stream_consistent(t, channel, box, g, sh, Q, TH)) return R2F_EHIP;
    out8[0] = g.n_entries, out8[1] = g.n_rowsteps, out8[2] = g.n_phases, out8[3] = g.sym, out8[4] = g.kh, out8[5] = g.kw, out8[6] = g.RS,
    out8[7] = g.max_lds_rows;
    return R2F_OK;
}

// cv::resize's per-destination tables for INTER_LANCZOS4: CV_8U weights in 11-bit fixed point, CV_32F weights as floats
int r2f_lanczos4_table(int ssize, int dsize, int* ofs, short* coef) { return r2f::plan::lanczos4_table_u8(ssize, dsize, ofs, coef) ? R2F_EINVAL : R2F_OK; }

int r2f_lanczos4_table_f32(int ssize, int dsize, int* ofs, float* coef) {
    return r2f::plan::lanczos4_table_f32(ssize, dsize, ofs, coef) ? R2F_EINVAL : R2F_OK;
}

size_t r2f_workspace_bytes(const r2f_params* p, int H, int W) {
    if (!p || H <= 0 || W <= 0) return 0;
    return r2f::plan::workspace_floats(p->flags, p->burn_cell, H, W) * sizeof(float);
}

int r2f_plan_tile_order(int gx, int gy, int band, int* order) {
    if (!order || gx < 1 || gy < 1 || band < 0 || (long long)gx * gy > (1 << 24)) return R2F_EINVAL;
    const std::vector<int> o = r2f::plan::tile_order(gx, gy, band);
    memcpy(order, o.data(), o.size() * sizeof(int));
    return R2F_OK;
}

}  // extern "C"
