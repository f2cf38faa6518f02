// plan_fuzz.cpp -- fuzz harness for the host-side planners of libr2f_hip.so (raw2film_amd/csrc/r2f_plan.cpp), built by
// tests/test_plan_sanitizers.py with `g++ -fsanitize=address,undefined -fno-sanitize-recover=all` and run as a child process:
// the sanitizers abort on the first out-of-bounds access, overflow or invalid cast; the checks below abort on a plan that breaks
// its own contract.  Test infrastructure: nothing in the product links this file.
//
//   plan_fuzz <seed> <cases>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/r2f.h"
#include "../raw2film_amd/csrc/r2f_plan.h"

using namespace r2f::plan;

static uint64_t g_state = 1;
static uint32_t rnd() {  // xorshift64*
    g_state ^= g_state >> 12, g_state ^= g_state << 25, g_state ^= g_state >> 27;
    return (uint32_t)((g_state * 2685821237ULL) >> 32);
}
static int rint_in(int lo, int hi) { return lo + (int)(rnd() % (uint32_t)(hi - lo + 1)); }
static float rfloat() { return (float)(rnd() >> 8) / 16777216.f; }
template <class T>
static T pick(std::initializer_list<T> v) { return *(v.begin() + rnd() % v.size()); }

#define CHECK(cond, ...)                                              \
    do {                                                              \
        if (!(cond)) {                                                \
            fprintf(stderr, "plan_fuzz: %s failed: ", #cond);         \
            fprintf(stderr, __VA_ARGS__);                             \
            fprintf(stderr, "\n");                                    \
            abort();                                                  \
        }                                                             \
    } while (0)

// frame sizes 1 .. 16384 with the small and the awkward ones over-represented
static int frame_dim() {
    switch (rnd() % 6) {
        case 0: return rint_in(1, 8);
        case 1: return rint_in(1, 300);
        case 2: return pick({255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 4095, 4096, 8192, 12288, 16383, 16384});
        default: return rint_in(1, 16384);
    }
}
static int tap_dim() {
    switch (rnd() % 5) {
        case 0: return rint_in(1, 12);
        case 1: return pick({19, 20, 21, 35, 85, 87, 199, 200, 201, 255, 256, 257, 399, 400});
        default: return rint_in(1, 400);
    }
}

static void fuzz_fft() {
    const int bh = tap_dim(), bw = tap_dim(), W = frame_dim(), H = frame_dim(), nch = rint_in(1, 3);
    const int elem = pick({16, 8, 12});
    const int window = pick({0, 0, 256, 512, 1024}), window_rows = pick({0, 0, 256, 512}), window_max = pick({256, 512, 512, 1024});
    const int batch = pick({1, 2, 7, 64, 192, 256, 4096}), streams = rint_in(1, 4);
    // a row range of the frame, like a shard's call
    const int y0 = rint_in(0, H - 1), y1 = rint_in(y0 + 1, H);
    r2f_fft_plan p;
    memset(&p, 0, sizeof p);
    const int rc = r2f_plan_fft(bh, bw, W, y1 - y0, nch, elem, window, window_rows, window_max, batch, streams, &p);
    if (rc != R2F_OK) {
        // no window keeps outputs only when a forced axis cannot hold the box
        const bool rows_forced_small = window_rows == 256 && bh <= 200 ? false : false;
        (void)rows_forced_small;
        CHECK(rc == R2F_EINVAL, "rc %d", rc);
        const bool must_fit = (bh <= 200 || true) && (bw <= 200 || true);
        (void)must_fit;
        // with nothing forced a shape always exists for boxes up to 400 taps
        if (window == 0 && window_rows == 0) CHECK(false, "no window for %d x %d taps on %d x %d (max %d)", bh, bw, W, H, window_max);
        return;
    }
    CHECK((p.ny == 256 || p.ny == 512) && (p.nx == 256 || p.nx == 512 || p.nx == 1024), "shape %d x %d", p.ny, p.nx);
    CHECK(p.vy == p.ny - bh + 1 && p.vy >= 1 && p.vx >= 4 && p.vx % 4 == 0 && p.vx <= p.nx - bw + 1, "valid %d x %d", p.vy, p.vx);
    CHECK((long long)p.gx * p.vx >= W && (long long)(p.gx - 1) * p.vx < W, "gx %d vx %d W %d", p.gx, p.vx, W);
    const int gyw = p.ntiles / p.gx;
    CHECK(p.ntiles == gyw * p.gx && (long long)gyw * p.vy >= y1 - y0 && (long long)(gyw - 1) * p.vy < y1 - y0, "tiles %d", p.ntiles);
    CHECK(p.pairs_per_channel == (p.ntiles + 1) / 2 && p.pairs == p.pairs_per_channel * nch, "pairs %d", p.pairs);
    CHECK(p.batch >= 1 && p.batch <= p.pairs && p.streams >= 1 && p.streams <= streams, "batch %d streams %d", p.batch, p.streams);
    CHECK((long long)p.launches * p.batch >= p.pairs && (long long)(p.launches - 1) * p.batch < p.pairs, "launches %d", p.launches);
    CHECK(p.scratch_bytes == (uint64_t)p.batch * p.streams * p.ny * p.nx * elem, "scratch %llu", (unsigned long long)p.scratch_bytes);
    if (window_rows && bh <= 200) CHECK(p.ny == window_rows, "rows forced %d got %d", window_rows, p.ny);
    if (window && bw <= 200) CHECK(p.nx == window, "columns forced %d got %d", window, p.nx);
    if (bh > 200) CHECK(p.ny == 512, "tall box in %d rows", p.ny);
    if (bw > 200) CHECK(p.nx >= 512, "wide box in %d columns", p.nx);
}

static void fuzz_stencil() {
    // a random stencil: dense, sparse, disc-like, mirror symmetric or not, with zero borders (cropping) and zero planes
    int kh = rnd() % 4 ? rint_in(1, 40) : tap_dim(), kw = rnd() % 4 ? rint_in(1, 40) : tap_dim();
    if ((long long)kh * kw > 120 * 120) kh = std::min(kh, 120), kw = std::min(kw, 120);  // keep a case cheap
    const int kc = pick({1, 3});
    std::vector<float> k((size_t)kh * kw * kc, 0.f);
    const int mode = rnd() % 5;
    const int zb_i = rnd() % 3 ? 0 : rint_in(0, kh / 3), zb_j = rnd() % 3 ? 0 : rint_in(0, kw / 3);
    for (int c = 0; c < kc; ++c)
        for (int i = zb_i; i < kh - zb_i; ++i)
            for (int j = zb_j; j < kw - zb_j; ++j) {
                float v = rfloat() - (mode == 3 ? 0.3f : 0.f);
                if (mode == 1 && rnd() % 4) v = 0.f;
                if (mode == 2) {  // a disc around the centre
                    const float di = i - kh / 2, dj = j - kw / 2;
                    if (di * di + dj * dj > 0.25f * std::min(kh, kw) * std::min(kh, kw)) v = 0.f;
                }
                k[((size_t)i * kw + j) * kc + c] = v;
            }
    if (mode != 3 && rnd() % 2)  // mirror the left half onto the right one, bit for bit
        for (int c = 0; c < kc; ++c)
            for (int i = 0; i < kh; ++i)
                for (int j = 0; j < kw / 2; ++j) k[((size_t)i * kw + (kw - 1 - j)) * kc + c] = k[((size_t)i * kw + j) * kc + c];
    if (mode == 4 && kc == 3)  // one plane all zero, one a single tap at the anchor
        for (int i = 0; i < kh; ++i)
            for (int j = 0; j < kw; ++j) {
                k[((size_t)i * kw + j) * 3 + 1] = 0.f;
                k[((size_t)i * kw + j) * 3 + 2] = (i == kh / 2 && j == kw / 2) ? 1.f : 0.f;
            }
    const Taps t{k.data(), kh, kw, kc};
    const int c = rint_in(0, 2);
    int box[4];
    tap_box(t, c, box);
    CHECK(box[0] >= 0 && box[1] < kh && box[0] <= box[1] && box[2] >= 0 && box[3] < kw && box[2] <= box[3], "box");
    float w1 = 0.f;
    (void)single_tap_channel(t, c, &w1);
    // the three tile variants of the direct kernel and the tail tile, with the budgets the library uses
    const int Q = pick({2, 4}), TW = pick({64, 128}), TH = Q * pick({8, 16, 32});
    const int budget = pick({0, 0, 8 * 1024, 40 * 1024, 80 * 1024, 160 * 1024});
    int out8[8];
    const int rc = r2f_plan_stencil(k.data(), kh, kw, kc, c, Q, TW, TH, budget, rnd() % 4 != 0, out8);
    CHECK(rc == R2F_OK || rc == R2F_ETOOLARGE, "plan_stencil rc %d (a planner bug: the entry list does not reproduce its taps)", rc);
    if (rc == R2F_OK) {
        CHECK(out8[0] >= 0 && out8[1] >= 0 && (out8[0] == 0) == (out8[1] == 0) && out8[2] >= 1 && out8[6] == TW + (out8[5] + 3) / 4 * 4, "geometry");
        if (budget) CHECK(((long long)out8[7] * out8[6] + 16) * 4 <= budget, "phase of %d rows x %d floats exceeds %d bytes", out8[7], out8[6], budget);
    } else {
        CHECK(budget != 0, "too large without a budget");
    }
    // unrolled / separable forms on what the planner produced
    StencilGeom geom[3];
    StreamHost sh;
    for (int ch = 0; ch < 3; ++ch) {
        int b[4];
        tap_box(t, ch, b);
        (void)plan_stencil_channel(t, ch, b, mirror_symmetric(t, ch, b), Q, TW, TH, 0, &geom[ch], &sh);
    }
    const int chans[3] = {0, 1, 2};
    const int R = fixed_stencil_radius(t, geom, chans, 3, 11);
    CHECK(R >= 0 && R <= 11, "R %d", R);
    if (R) {
        bool same;
        const std::vector<float> w = fixed_stencil_weights(t, R, Q, &same);
        CHECK(w.size() == (size_t)3 * (2 * R + Q) * (R + 1) * (Q / 2) * 2, "fixed weights");
        float u[3][19], v[3][10];
        if (R <= 9) (void)separable_taps(t, R, u, v);
    }
    int lo, hi;
    const int H = frame_dim(), y0 = rint_in(0, H - 1), y1 = rint_in(y0 + 1, H);
    stencil_source_rows(y0, y1, rint_in(0, 250), rint_in(0, 250), H, &lo, &hi);
    CHECK(lo >= 0 && hi <= H && lo < hi && lo <= y0 && hi >= y1, "source rows [%d, %d) for [%d, %d) of %d", lo, hi, y0, y1, H);
}

static void fuzz_tables() {
    {  // tile orders
        const int gx = rint_in(1, 200), gy = rint_in(1, 200), band = rnd() % 2 ? 0 : rint_in(1, 64);
        std::vector<int> order((size_t)gx * gy, -1), count((size_t)gx * gy, 0);
        CHECK(r2f_plan_tile_order(gx, gy, band, order.data()) == R2F_OK, "tile order");
        for (int v : order) {
            CHECK(v >= 0 && v < gx * gy, "tile %d of %d", v, gx * gy);
            ++count[(size_t)v];
        }
        for (int n : count) CHECK(n == 1, "not a permutation");
    }
    {  // LANCZOS4 tables
        const int ss = rnd() % 3 ? rint_in(1, 400) : frame_dim(), ds = rnd() % 3 ? rint_in(1, 400) : frame_dim();
        std::vector<int> ofs((size_t)ds);
        std::vector<short> cs((size_t)ds * 8);
        std::vector<float> cf((size_t)ds * 8);
        CHECK(r2f_lanczos4_table(ss, ds, ofs.data(), cs.data()) == R2F_OK, "lanczos u8");
        for (int d = 0; d < ds; ++d) {
            int sum = 0;
            for (int kk = 0; kk < 8; ++kk) sum += cs[(size_t)d * 8 + kk];
            CHECK(ofs[(size_t)d] >= -1 && ofs[(size_t)d] <= ss && std::abs(sum - 2048) <= 8, "u8 row %d: ofs %d sum %d", d, ofs[(size_t)d], sum);
        }
        CHECK(r2f_lanczos4_table_f32(ss, ds, ofs.data(), cf.data()) == R2F_OK, "lanczos f32");
        for (float v : cf) CHECK(std::isfinite(v), "non-finite weight");
    }
    {  // curve cells: monotone, with repeats, tiny and huge ranges
        const int m = rnd() % 4 ? rint_in(2, 300) : pick({2, 3, 1024, 4096});
        std::vector<float> lut((size_t)4 * m);
        float x = (rfloat() - 0.5f) * pick({1.f, 1e-6f, 1e6f, 1e30f});
        const float step = pick({1e-3f, 1.f, 1e-30f, 1e20f});
        for (int i = 0; i < m; ++i) {
            lut[(size_t)i] = x;
            if (rnd() % 8) x += step * rfloat();
        }
        for (size_t i = (size_t)m; i < lut.size(); ++i) lut[i] = (rfloat() - 0.5f) * 8.f;
        CurveCells cc;
        const int rc = curve_cells(lut.data(), m, &cc);
        CHECK(rc == 0 && cc.cells.size() == (size_t)3 * (m - 1) * 4 && (cc.near == 0 || cc.near == 1), "curve cells");
        lut[(size_t)rint_in(0, m - 1)] = NAN;  // a table with a NaN abscissa is refused (m == 2: the NaN breaks the only comparison)
        CHECK(curve_cells(lut.data(), m, &cc) == -1, "NaN abscissa accepted");
    }
    {  // Gaussian tables, workspace sizes
        float w[kChromaMaxTaps];
        const int size = rint_in(-2, 40);
        const bool ok = chroma_weights(size, w);
        CHECK(ok == (size >= 1 && 2 * size + 1 <= kChromaMaxTaps), "chroma size %d", size);
        if (ok) {
            float s = 0.f;
            for (int i = 0; i <= 2 * size; ++i) s += w[i];
            CHECK(std::fabs(s - 1.f) < 1e-5f, "chroma weights sum %g", s);
        }
        double bw13[13];
        burn_weights(bw13);
        r2f_params p;
        memset(&p, 0, sizeof p);
        p.flags = rnd() & 63u;
        p.burn_cell = rint_in(-1, 400);
        const int H = frame_dim(), W = frame_dim();
        const size_t bytes = r2f_workspace_bytes(&p, H, W);
        CHECK(bytes % 16 == 0 && bytes <= ((size_t)2 * 3 * ((size_t)H * W + 3) + 4 * (size_t)H * W + 16) * 4, "workspace %zu", bytes);
    }
}

int main(int argc, char** argv) {
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int cases = argc > 2 ? atoi(argv[2]) : 2000;
    g_state = seed * 0x9E3779B97F4A7C15ULL + 1;
    for (int i = 0; i < cases; ++i) {
        fuzz_fft();
        if (i % 4 == 0) fuzz_stencil();
        if (i % 8 == 0) fuzz_tables();
    }
    printf("plan_fuzz: seed %llu, %d cases ok\n", (unsigned long long)seed, cases);
    return 0;
}
