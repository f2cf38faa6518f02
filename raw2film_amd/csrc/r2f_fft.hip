// Overlap-save FFT form of the large per-channel stencils (S2 halation, S5 MTF), fp64.
//
// The reference's own CPU path runs cv.filter2D's DFT branch for kernels larger than 11 x 11; in fp32 that carries an error
// relative to the tile's energy (a shadow pixel next to a specular highlight misses the 1e-5 bar), in fp64 it is exact to
// ~1e-13 and costs ~400 flops per pixel instead of the 2 x 5 721 of the direct sum.  MI355X's fp64 vector rate (78 TFLOP/s)
// makes that the cheaper exact algorithm for the 87 x 87 disc.
//
// Unit of work: a PAIR of ny x nx input windows of one channel (ny, nx = 256 or 512, FftConvArgs) packed as real +
// imaginary part of one complex image (the kernel is real, so the correlation of the complex image is the pair of
// correlations -- no Hermitian bookkeeping at all).  Window (ty, tx) produces the outputs [ty, ty + ny - kh] x
// [tx, tx + nx - kw] from the input rows ty - ay .. + ny - 1.  One scratch image S(r, k) (complex128, 1 MB per 256 x 256,
// column-blocked layout, see sidx) per pair:
//
//   pass 1  rows:     load (reflect-101 on the global frame), forward FFT along x                      -> S[r][k]
//   pass 2  columns:  forward FFT along r, multiply by conj(K^)[r'][k], inverse FFT along r', in place  -> S[r][k]
//   pass 3  rows:     inverse FFT along k, scale 1 / (ny nx), crop to the valid outputs, [log + density curve], store
//
// Round 5, all three behind options that default on (records: profiles/r05_fft_levers_ab.txt, r05_scratch96_probe.txt):
//   * a centrally symmetric tap box (both production stencils) is laid out around the window origin, so its spectrum is REAL:
//     8 bytes per element in pass 2, two multiplies, and the valid outputs of a window start at (oy, ox) = the anchor (kreal);
//   * pass 2 of 256-row windows with a real spectrum runs as a resident grid whose workgroups WALK the launch's pairs for their
//     16 columns, the spectrum in registers from pair to pair (fft_cols_walk_kernel);
//   * the scratch element is a template parameter of every pass: complex128, complex64 (the MTF: density is bounded), or a
//     12-byte element of doubles rounded to 48 bits -- for the halation chosen ON THE DEVICE per window pair from the exposure
//     range recorded for the pair's samples (ST = 3 kernels read the pair's flag, wave-uniformly; round 6: per pair, not per frame).
//
// A 256-point line is transformed by 16 lanes holding 16 elements each (element n = lane + 16 m): a 16-point DFT in
// registers, the twiddles W_256^(lane p), a 16 x 16 transpose through LDS, a second 16-point DFT -- natural order in and
// out, one LDS round trip per transform.  (The first version ran four radix-4 stages through LDS and was bound by
// ds_write_b128 issue: 0.6 ms per 1 024 pairs in pass 2.)  A wave carries 4 lines; lines of a wave are 4 neighbouring rows
// (passes 1, 3: 256-byte segments per access) or 4 neighbouring columns (pass 2: 1 KB contiguous per access).
//
// A 512-point line is transformed by 32 lanes holding 16 elements each (fft512: 512 = 32 x 16, the same two register DFTs
// around an LDS transpose, plus one radix-2 step across neighbouring lanes by DPP), so every pass keeps the register budget
// of the 256-point version and a wave carries 2 lines instead of 4.  Its output order differs from its input order, which
// the row passes absorb in their addressing; the column pass of 512-row windows comes back through fft512_rev, the
// transposed flow graph.  The host picks the window shape that moves the fewest scratch bytes for the stencil and the
// frame (an 87-tap disc keeps 66 % of a 256-wide window's columns but 83 % of a 512-wide one's); cfg 4: 256 x 512.
#include <algorithm>

#include "r2f_launch.h"
#include "r2f_fft_math.h"

#include "../../include/r2f.h"

namespace r2f {

constexpr int kN = kFftN;             // 256
constexpr int kFftThreads = 256;      // 4 waves x 4 lines
constexpr int kTPitch = 17;           // transpose tile row pitch (doubles): 16 + 1
constexpr int kTLine = 16 * kTPitch;  // 272 doubles per line: = 16 (mod 32), so two lines fill the 32 8-byte slots exactly

// 256-point transform of the line whose element (l + 16 m) sits in v[m] of lane l (l = lane & 15); on return v[q] holds
// output element (l + 16 q).  w1 = exp(-2 pi i l / 256).  tbuf: this WAVE's transpose buffer, 4 lines x 272 doubles.
#ifndef R2F_FFT_EXPT
#define R2F_FFT_EXPT 0  // development switch for fft256: bit 0 no LDS transpose (wrong results, timing only)
#endif
template <bool INV>
__device__ __forceinline__ void fft256(cplx (&v)[16], const cplx w1, double* tbuf, int lane) {
    dft16<INV>(v);
    twiddle_powers<INV>(v, w1);
    if (R2F_FFT_EXPT & 1) {
        dft16<INV>(v);
        return;
    }
    // 16 x 16 transpose inside each 16-lane group, real parts then imaginary parts through the same buffer
    const int l = lane & 15;
    double* t = tbuf + (lane >> 4) * kTLine;
#pragma unroll
    for (int p = 0; p < 16; ++p) t[p * kTPitch + l] = v[p].x;
    __builtin_amdgcn_wave_barrier();
    double re[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) re[j] = t[l * kTPitch + j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 16; ++p) t[p * kTPitch + l] = v[p].y;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = make_double2(re[j], t[l * kTPitch + j]);
    __builtin_amdgcn_wave_barrier();
    dft16<INV>(v);
}

// A lane's double moved to / from its neighbour lane ^ 1 (DPP quad_perm [1, 0, 3, 2], no LDS).
__device__ __forceinline__ double swap_lane1(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

constexpr int kTPitch512 = 34;                // transpose tile row pitch of a 512-point line: 32 + 2 (conflict-free both ways)
constexpr int kTLine512 = 16 * kTPitch512;    // 544 doubles per line; a wave's 2 lines fill the same 1 088 doubles as 4 x 272

// W_32^k = (kC32[k], -kS32[k]), k < 16: the radix-2 step of the 512-point transforms
constexpr double kC32[16] = {1.0, 0.9807852804032304, 0.9238795325112867, 0.8314696123025452, 0.7071067811865476, 0.5555702330196023, 0.38268343236508984, 0.19509032201612833, 0.0, -0.19509032201612833, -0.38268343236508984, -0.5555702330196023, -0.7071067811865476, -0.8314696123025452, -0.9238795325112867, -0.9807852804032304};
constexpr double kS32[16] = {0.0, 0.19509032201612833, 0.38268343236508984, 0.5555702330196023, 0.7071067811865476, 0.8314696123025452, 0.9238795325112867, 0.9807852804032304, 1.0, 0.9807852804032304, 0.9238795325112867, 0.8314696123025452, 0.7071067811865476, 0.5555702330196023, 0.38268343236508984, 0.19509032201612833};

// 512-point transform of the line whose element (l + 32 j) sits in v[j] of lane l (l = lane & 31): 512 = 32 x 16.  A 16-point
// DFT over j, the twiddles W_512^(l p), a transpose after which the lane pair (2 p, 2 p + 1) holds the even / odd l of
// output residue p, a second 16-point DFT, and one radix-2 step across the pair.  On return v[i] of lane l holds output
// element 256 (l & 1) + 16 i + (l >> 1).  w1 = exp(-2 pi i l / 512).  tbuf: this WAVE's transpose buffer.
template <bool INV>
__device__ __forceinline__ void fft512(cplx (&v)[16], const cplx w1, double* tbuf, int lane) {
    dft16<INV>(v);
    twiddle_powers<INV>(v, w1);
    const int l = lane & 31;
    double* t = tbuf + (lane >> 5) * kTLine512;
    const double* rd = t + (l >> 1) * kTPitch512 + (l & 1);
#pragma unroll
    for (int p = 0; p < 16; ++p) t[p * kTPitch512 + l] = v[p].x;
    __builtin_amdgcn_wave_barrier();
    double re[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) re[i] = rd[2 * i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 16; ++p) t[p * kTPitch512 + l] = v[p].y;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = make_double2(re[i], rd[2 * i]);
    __builtin_amdgcn_wave_barrier();
    dft16<INV>(v);  // even lanes: E[i] over the even l, odd lanes: O[i] over the odd l
    // X[i] = E[i] + W_32^i O[i] (even lane), X[16 + i] = E[i] - W_32^i O[i] (odd lane)
    const bool odd = l & 1;
    if (odd) {
#pragma unroll
        for (int i = 1; i < 16; ++i) v[i] = ctw<INV>(v[i], make_double2(kC32[i], -kS32[i]));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const cplx o = make_double2(swap_lane1(v[i].x), swap_lane1(v[i].y));
        v[i] = odd ? csub(o, v[i]) : cadd(v[i], o);
    }
}

// The same transform with the index maps exchanged (the transposed flow graph): on entry v[i] of lane l holds element
// 256 (l & 1) + 16 i + (l >> 1) -- what fft512 leaves -- and on return v[j] holds output element l + 32 j.  Pass 2 runs
// fft512 forward, multiplies in place, and comes back through this one: no re-ordering in between.
template <bool INV>
__device__ __forceinline__ void fft512_rev(cplx (&v)[16], const cplx w1, double* tbuf, int lane) {
    const int l = lane & 31;
    const bool odd = l & 1;
    // radix-2 across the lane pair, decimation in frequency: even lane S[i] = X[i] + X[16 + i] (-> even outputs),
    // odd lane D[i] = (X[i] - X[16 + i]) W_32^i (-> odd outputs)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const cplx o = make_double2(swap_lane1(v[i].x), swap_lane1(v[i].y));
        v[i] = odd ? csub(o, v[i]) : cadd(v[i], o);
    }
    if (odd) {
#pragma unroll
        for (int i = 1; i < 16; ++i) v[i] = ctw<INV>(v[i], make_double2(kC32[i], -kS32[i]));
    }
    dft16<INV>(v);  // lane (p, h), register m: B[2 m + h][p]
    double* t = tbuf + (lane >> 5) * kTLine512;
    double* wr = t + (l >> 1) * kTPitch512 + (l & 1);
#pragma unroll
    for (int m = 0; m < 16; ++m) wr[2 * m] = v[m].x;
    __builtin_amdgcn_wave_barrier();
    double re[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) re[p] = t[p * kTPitch512 + l];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 16; ++m) wr[2 * m] = v[m].y;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 16; ++p) v[p] = make_double2(re[p], t[p * kTPitch512 + l]);
    __builtin_amdgcn_wave_barrier();
    twiddle_powers<INV>(v, w1);
    dft16<INV>(v);
}

// A lane's double moved to / from lane ^ 2 of its quad (DPP quad_perm [2, 3, 0, 1]).
__device__ __forceinline__ double swap_lane2(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x4E, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x4E, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

constexpr int kTPitch1024 = 68;  // transpose tile row pitch of a 1024-point line: 64 + 4 (conflict-free both ways); 16 rows = 1 088 doubles

// 1024-point transform by a whole wave: element (l + 64 j) sits in v[j] of lane l.  1024 = 64 x 16: a 16-point DFT over j,
// the twiddles W_1024^(l p), a transpose after which the lane quad (4 p .. 4 p + 3) holds the l = 4 i + h of residue p,
// a second 16-point DFT over i, the twiddles W_64^(h c), and a radix-4 step across the quad as two DPP exchanges.
// On return v[c] of lane l holds output element (l >> 2) + 16 c + 256 g, g = the bit reversal of l & 3.
// w1 = exp(-2 pi i l / 1024).  tbuf: this WAVE's transpose buffer (all of it).
template <bool INV>
__device__ __forceinline__ void fft1024(cplx (&v)[16], const cplx w1, double* t, int lane) {
    dft16<INV>(v);
    twiddle_powers<INV>(v, w1);
    const int h = lane & 3;
    const double* rd = t + (lane >> 2) * kTPitch1024 + h;
#pragma unroll
    for (int p = 0; p < 16; ++p) t[p * kTPitch1024 + lane] = v[p].x;
    __builtin_amdgcn_wave_barrier();
    double re[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) re[i] = rd[4 * i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 16; ++p) t[p * kTPitch1024 + lane] = v[p].y;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = make_double2(re[i], rd[4 * i]);
    __builtin_amdgcn_wave_barrier();
    dft16<INV>(v);  // lane (p, h): E_h[c] over the l = 4 i + h
    // W_64^h, h < 4
    const cplx wh = make_double2(h == 0 ? 1.0 : (h == 1 ? 0.9951847266721969 : (h == 2 ? 0.9807852804032304 : 0.9569403357322088)),
                                 h == 0 ? 0.0 : (h == 1 ? -0.0980171403295606 : (h == 2 ? -0.19509032201612825 : -0.2902846772544623)));
    twiddle_powers<INV>(v, wh);
    // 4-point DFT across the quad, decimation in frequency: b0 = a0 + a2, b1 = a1 + a3, b2 = a0 - a2, b3 = (a1 - a3) W_4;
    // X0 = b0 + b1 (lane 0), X2 = b0 - b1 (lane 1), X1 = b2 + b3 (lane 2), X3 = b2 - b3 (lane 3)
    const bool upper = h & 2, odd = h & 1, rot = h == 3;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        cplx o = make_double2(swap_lane2(v[c].x), swap_lane2(v[c].y));
        cplx b = upper ? csub(o, v[c]) : cadd(v[c], o);
        if (rot) b = INV ? make_double2(-b.y, b.x) : make_double2(b.y, -b.x);  // W_4 = -i (forward), +i (inverse)
        o = make_double2(swap_lane1(b.x), swap_lane1(b.y));
        v[c] = odd ? csub(o, b) : cadd(b, o);
    }
}

// window origin (first input row / column) and validity of window `t` of the launch
__device__ __forceinline__ bool window_of(const FftConvArgs& a, int t, int& wy, int& wx) {
    if (t >= a.ntiles) return false;
    const int ty = a.y0 + (t / a.gx) * a.vy, tx = (t % a.gx) * a.vx;
    wy = ty - a.ay;
    wx = tx - a.ax;
    return true;
}

__device__ __forceinline__ double* wave_tbuf(double* smem) { return smem + (threadIdx.x >> 6) * 4 * kTLine; }

// Scratch / spectrum layout: 16 x 16 blocks [r / 16][k / 16] of 4 KB; inside a block four neighbouring columns are
// interleaved row by row: [(k / 4) % 4][r % 16][k % 4].  The column pass (a wave = 4 neighbouring columns x 16 rows per
// access) then moves 1 KB contiguous per instruction, and the row passes (a wave = 4 neighbouring rows x 16 columns) four
// 256-byte pieces of one block.
// nbx = blocks per block row = nx / 16.
// (unsigned: with the uniform image base in SGPRs an access then needs one offset VGPR, not a 64-bit address pair)
__device__ __forceinline__ unsigned sidx(int r, int k, int nbx) {
    return (unsigned)((((r >> 4) * nbx + (k >> 4)) << 8) + ((((k >> 2) & 3) * 16 + (r & 15)) << 2) + (k & 3));
}

// element `idx` of a scratch image: the byte offset stays a 32-bit value, so the access is SGPR base + one offset VGPR
__device__ __forceinline__ cplx& at(cplx* base, unsigned idx) {
    return *reinterpret_cast<cplx*>(reinterpret_cast<char*>(base) + (idx << 4));
}
__device__ __forceinline__ const cplx& at(const cplx* base, unsigned idx) {
    return *reinterpret_cast<const cplx*>(reinterpret_cast<const char*>(base) + (idx << 4));
}

// Scratch images come in two element types.  S32 = false: complex128, what the halation needs (linear exposure: a shadow
// pixel shares its window with speculars 10^4 times brighter, and a rounding to fp32 is relative to the WINDOW's energy).
// S32 = true: complex64 -- only the two roundings between the passes are fp32, every butterfly stays fp64 -- for stencils on
// DENSITY (the MTF): values in [0, 4], so those roundings cost ~1e-7 absolute (under one fp32 ulp of a density >= 1; measured
// in tests/test_gpu_fft.py) and the three passes move half the bytes.
// ST = 2: a 12-byte element, each component a double rounded to its upper 48 bits (sign, exponent, 36 mantissa bits: 2^-37
// relative to the component, full double range): {upper word of re, upper word of im, bits 16..31 of both lower words}.  A
// quarter fewer bytes than complex128 for 2^-37 of the WINDOW's magnitude per rounding (round 2's first form of this element, two
// fp32 heads + two bf16 residuals, was 2^-33 and cost 14 instructions per element to pack where this one costs 5).
struct __attribute__((packed, aligned(4))) C96 {
    unsigned hr, hi;
    unsigned lo;
};
template <int ST>
__device__ __forceinline__ cplx sld(const void* base, unsigned idx) {
    if (ST == 1) {
        const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base) + (idx << 3));
        return make_double2((double)v.x, (double)v.y);
    }
    if (ST == 2) {
        const C96 e = *reinterpret_cast<const C96*>(reinterpret_cast<const char*>(base) + idx * 12u);
        return make_double2(__hiloint2double((int)e.hr, (int)(e.lo << 16)), __hiloint2double((int)e.hi, (int)(e.lo & 0xffff0000u)));
    }
    return *reinterpret_cast<const cplx*>(reinterpret_cast<const char*>(base) + (idx << 4));
}
template <int ST>
__device__ __forceinline__ void sst(void* base, unsigned idx, const cplx v) {
    if (ST == 1)
        *reinterpret_cast<float2*>(reinterpret_cast<char*>(base) + (idx << 3)) = make_float2((float)v.x, (float)v.y);
    else if (ST == 2) {
        // round to nearest at bit 16 of the lower word: the IEEE bit pattern is monotonic, a carry out of the mantissa lands in the
        // exponent as it should (finite values only: pass 1 zeroes the others)
        const unsigned long long br = (unsigned long long)__double_as_longlong(v.x) + 0x8000ull;
        const unsigned long long bi = (unsigned long long)__double_as_longlong(v.y) + 0x8000ull;
        C96 e;
        e.hr = (unsigned)(br >> 32), e.hi = (unsigned)(bi >> 32);
        e.lo = ((unsigned)br >> 16) | ((unsigned)bi & 0xffff0000u);
        *reinterpret_cast<C96*>(reinterpret_cast<char*>(base) + idx * 12u) = e;
    } else
        *reinterpret_cast<cplx*>(reinterpret_cast<char*>(base) + (idx << 4)) = v;
}
// s32 == 3 (FftConvArgs::dyn_flags): the element is chosen PER WINDOW PAIR -- the 12-byte one when the range of the samples the pair's
// two windows hold allows it: max |x| within `bound` times the smallest sample that matters (samples below `floor` end on the clamped
// part of the density curve).  fft_decide_kernel (below) works that out once per call from the exposure-range tiles and leaves one flag
// per pair-in-channel; the passes read theirs with a scalar load (the pair is wave-uniform).  Scratch images of such a launch are
// 16 bytes per element apart whatever they hold, so that pairs of either kind can sit next to each other.
__device__ __forceinline__ bool pair_packed(const FftConvArgs& a, int pc) { return a.dyn_flags[pc] != 0; }

// scratch image of pair `pair` (n elements each); ST = 3: 16-byte pitch (see above)
template <int ST>
__device__ __forceinline__ char* simg(double2* s1, long long pair, long long n) {
    return reinterpret_cast<char*>(s1) + pair * n * (ST == 1 ? 8 : (ST == 2 ? 12 : 16));
}

// One thread per pair-in-channel: {min, max |.|} over the tiles the pair's two windows touch -> flags[pc].  A window reads the
// global rows [wy, wy + ny) and columns [wx, wx + nx), reflected (101) into the frame and, for a row shard, clamped to the rows
// its source buffer holds (pass 1).  What counts is every sample whose rounding can reach a KEPT output:
// * along a row the transforms mix all nx columns, so all columns of the window count, also the reflected ones that feed only
//   discarded outputs.  At the left edge the reflection lands inside the part of the window that is in the frame; at the right
//   edge it need not -- the last window column may keep 44 frame columns and bring in 468 reflected ones from up to 424 columns
//   to the LEFT of the window -- so the column range is widened by the reflection's reach (reflected_reach: a superset when a
//   narrow frame is reflected more than once);
// * across the rows both roundings are row-local (pass 1 rounds a row's own spectrum; pass 2 rounds an output row's, which the
//   stencil makes from the rows within its reach): rows that feed only discarded outputs cannot reach a kept one, and the rows
//   within the reach of a kept output are reflected into the frame rows of the window (a window keeps an output only if it holds the
//   reach above it).  tests/test_gpu_fft.py holds both edge cases.
// A tile nobody recorded (reset values) means "unknown": complex128.
#ifndef R2F_DECIDE_OWN_COLUMNS_ONLY
#define R2F_DECIDE_OWN_COLUMNS_ONLY 0
#endif
__device__ __forceinline__ void reflected_reach(int w0, int n, int size, int& lo, int& hi) {
    const int last = size - 1, w1 = w0 + n - 1;
    lo = max(w0, 0), hi = min(w1, last);
    if (w0 < 0) hi = max(hi, min(-w0, last));
    if (w1 > last) lo = min(lo, max(2 * last - w1, 0));
}
__global__ __launch_bounds__(256) void fft_decide_kernel(const FftConvArgs a, const RangeRecord rec, const float bound, const float floor_, int* flags) {
    const int pc = blockIdx.x * 256 + threadIdx.x;
    if (pc >= a.ppc) return;
    int lo = (int)kFrameMinReset, hi = 0;
    bool known = rec.tiles != nullptr;
    for (int half = 0; half < 2 && known; ++half) {
        int wy, wx;
        if (!window_of(a, 2 * pc + half, wy, wx)) continue;
        const int b0 = a.src.gy0, b1 = a.src.gy0 + a.src.rows - 1;
        const int r0 = clampi(max(wy, 0), b0, b1), r1 = clampi(min(wy + a.ny - 1, a.H_global - 1), b0, b1);
        int c0, c1;
        reflected_reach(wx, a.nx, a.W, c0, c1);
#if R2F_DECIDE_OWN_COLUMNS_ONLY  // development switch: the decide kernel as it was before the reflected reach counted (the hole
        c0 = max(wx, 0), c1 = min(wx + a.nx - 1, a.W - 1);  // tools/scratch_choice_model.py finds at once; never in a shipped build)
#endif
        for (int ty = r0 >> kRangeTileRowsLog2; ty <= (r1 >> kRangeTileRowsLog2); ++ty)
            for (int tx = c0 >> kRangeTileColsLog2; tx <= (c1 >> kRangeTileColsLog2); ++tx) {
                if ((unsigned)ty >= (unsigned)rec.tyn || (unsigned)tx >= (unsigned)rec.txn) {
                    known = false;
                    continue;
                }
                const int2 t = rec.tiles[(long long)ty * rec.txn + tx];
                if (t.x == (int)kFrameMinReset && t.y == (int)kFrameMaxReset) known = false;
                lo = min(lo, t.x), hi = max(hi, t.y);
            }
    }
    const float flo = __int_as_float(lo), fhi = __int_as_float(hi);
    flags[pc] = (known && fhi <= bound * fmaxf(flo, floor_)) ? 1 : 0;  // (a NaN anywhere makes the comparison false)
}

// ---------------------------------------------------------------------------------------------------- pass 1
// grid (ny / rows per workgroup, pairs).  256-point rows: 16 lanes per row, a workgroup transforms 16 rows, a wave 4 of them;
// 512-point rows: 32 lanes per row, 8 rows per workgroup, 2 per wave; 1024-point rows: a wave per row, 4 rows per workgroup.
#ifndef R2F_FFT_EXP
#define R2F_FFT_EXP 0  // development switch for pass 1: bit 0 no input loads, bit 1 no stores, bit 2 no transform,
                       // bit 3 timing model of "shared rows are transformed once and stored into both windows' images" (wrong results)
#endif
// XL: row length 256 << XL
template <int XL>
struct RowGeom {
    static constexpr int NX = 256 << XL, NBX = NX / 16, LPL = 16 << XL, ROWS = kFftThreads / LPL;
    // column (or frequency) held in register q of lane l after a transform
    static __device__ __forceinline__ int out_col(int l, int q) {
        return XL == 2 ? (l >> 2) + 16 * q + 256 * (((l & 1) << 1) | ((l >> 1) & 1)) : (XL == 1 ? 256 * (l & 1) + 16 * q + (l >> 1) : l + 16 * q);
    }
    static __device__ __forceinline__ double* line_buf(double* wave_buf, int lane) {
        return XL == 2 ? wave_buf : wave_buf + (XL == 1 ? (lane >> 5) * kTLine512 : (lane >> 4) * kTLine);
    }
    template <bool INV>
    static __device__ __forceinline__ void fft(cplx (&v)[16], const FftConvArgs& a, int l, double* wave_buf, int lane) {
        if (XL == 2)
            fft1024<INV>(v, a.tw1024[l], wave_buf, lane);
        else if (XL == 1)
            fft512<INV>(v, a.tw512[l], wave_buf, lane);
        else
            fft256<INV>(v, a.tw[l], wave_buf, lane);
    }
};

template <int XL, int ST>
__device__ __forceinline__ void fft_rows_fwd_body(const FftConvArgs& a, double* fsm) {
    typedef RowGeom<XL> G;
    constexpr int NX = G::NX, LPL = G::LPL;
    const int lane = threadIdx.x & 63, l = lane & (LPL - 1);
    // (the row index is made from an opaque copy of the thread id here and again behind the transform: as one value it lived in two
    // VGPRs -- with its sign extension -- from the first address to the last store, and in the kernel that holds both element
    // forms it was computed ahead of the branch and spilled: 2-6 VGPRs of scratch at four waves per SIMD)
    unsigned tid0 = threadIdx.x;
    asm volatile("" : "+v"(tid0));
    const int pair = blockIdx.y, r = blockIdx.x * G::ROWS + tid0 / LPL;
    int wyA = 0, wxA = 0, wyB = 0, wxB = 0;
    const int gp = a.pair0 + pair, ci = gp / a.ppc, pc = gp - ci * a.ppc;  // channel-major pair numbering
    const bool hasA = window_of(a, 2 * pc, wyA, wxA), hasB = window_of(a, 2 * pc + 1, wyB, wxB);
    const float* src = a.src.data + (long long)a.chan[ci] * a.src.plane_stride;
    if (R2F_FFT_EXP & 8) {  // the window row above (same channel, same launch) transforms this workgroup's rows: nothing to do here
        const int above = pair - (a.gx + 1) / 2;
        if ((int)(blockIdx.x + 1) * G::ROWS <= a.ny - a.vy && above >= 0 && (a.pair0 + above) / a.ppc == ci && !a.raw) return;
    }
    cplx v[16];  // v[m]: element l + LPL m of the row
    if (R2F_FFT_EXP & 1) {
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = make_double2(1.0 + m + l, 0.5 * r);
    } else if (a.raw) {  // the zero-padded kernel image itself: a plain ny x nx plane, no reflection
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = make_double2((double)src[(long long)r * NX + l + LPL * m], 0.0);
    } else {
        const float* rowA = src;
        const float* rowB = src;
        if (hasA) {
            const int sy = reflect101(wyA + r, a.H_global) - a.src.gy0;
            rowA = src + (long long)clampi(sy, 0, a.src.rows - 1) * a.W;  // rows outside the buffer only feed discarded outputs
        }
        if (hasB) {
            const int sy = reflect101(wyB + r, a.H_global) - a.src.gy0;
            rowB = src + (long long)clampi(sy, 0, a.src.rows - 1) * a.W;
        }
        // windows that do not touch the left / right frame edge (all but two per row of windows) need no reflection
        const bool inA = wxA >= 0 && wxA + NX <= a.W, inB = wxB >= 0 && wxB + NX <= a.W;
        // all loads of a window are issued before the first one is used (a per-element `has ? load : 0` had compiled into
        // thirty-two branches, each waiting for its own load)
        float fa[16], fb[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) fa[m] = fb[m] = 0.f;
        if (hasA) {
            if (inA) {
                const float* pa = rowA + wxA + l;
#pragma unroll
                for (int m = 0; m < 16; ++m) fa[m] = pa[LPL * m];
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m) fa[m] = rowA[reflect101(wxA + l + LPL * m, a.W)];
            }
        }
        if (hasB) {
            if (inB) {
                const float* pb = rowB + wxB + l;
#pragma unroll
                for (int m = 0; m < 16; ++m) fb[m] = pb[LPL * m];
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m) fb[m] = rowB[reflect101(wxB + l + LPL * m, a.W)];
            }
        }
        // A NaN or an infinity in ONE sample would come back from the transforms in every output of its window (the direct form
        // keeps it inside the stencil's reach, like the reference's): such a sample enters the correlation as 0.
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = make_double2(__builtin_isfinite(fa[m]) ? (double)fa[m] : 0.0, __builtin_isfinite(fb[m]) ? (double)fb[m] : 0.0);
    }
    // ST = 3 (element chosen per window pair on the device): the choice is needed at the stores only, so the pair's flag is read HERE --
    // behind the issue of the window loads, its round trip under the loads and the transform -- instead of at the top of the kernel,
    // where every workgroup's first global load would wait for it
    bool packed = false;
    if (ST == 3) packed = pair_packed(a, pc);
    if (!(R2F_FFT_EXP & 4)) {
        G::template fft<false>(v, a, l, wave_tbuf(fsm), lane);
    }
    // register q sits 16 q columns on: the next 16 x 16 block of the layout, 256 elements further
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // the row index again (see the top)
    const int rs = blockIdx.x * G::ROWS + tid / LPL;
    const unsigned sb = sidx(rs, G::out_col(l, 0), G::NBX);
    if (ST == 3) {
        char* s1 = simg<3>(a.s1, pair, (long long)a.ny * NX);  // (16-byte pitch between the pairs' images whatever they hold)
        if (packed) {
#pragma unroll
            for (int q = 0; q < 16; ++q) sst<2>(s1, sb + q * 256, v[q]);
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) sst<0>(s1, sb + q * 256, v[q]);
        }
        return;
    }
    constexpr int STS = ST == 3 ? 0 : ST;
    char* s1 = simg<STS>(a.s1, pair, (long long)a.ny * NX);
    if (R2F_FFT_EXP & 2) {
        if (v[3].x == 1.2345e300) sst<STS>(s1, 0, v[5]);  // keep the transform alive without storing
        return;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) sst<STS>(s1, sb + q * 256, v[q]);
    if (R2F_FFT_EXP & 8) {  // ... and this workgroup's rows below the valid outputs are the first rows of the window row below
        const int below = pair + (a.gx + 1) / 2;
        if (r >= a.vy && below < a.npairs && (a.pair0 + below) / a.ppc == ci && !a.raw) {
            char* s2 = simg<STS>(a.s1, below, (long long)a.ny * NX);
#pragma unroll
            for (int q = 0; q < 16; ++q) sst<STS>(s2, sidx(r - a.vy, G::out_col(l, q), G::NBX), v[q]);
        }
    }
}

#ifndef R2F_FFT_WPE1
#define R2F_FFT_WPE1 4
#endif
template <int XL, int ST>
__global__ __launch_bounds__(kFftThreads) __attribute__((amdgpu_waves_per_eu(R2F_FFT_WPE1, 8))) void fft_rows_fwd_kernel(const FftConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) double fsm[];
    fft_rows_fwd_body<XL, ST>(a, fsm);  // (ST = 3: the body reads the choice where it needs it, at its stores)
}

// ---------------------------------------------------------------------------------------------------- pass 2
// grid (nx / 16, pairs): a workgroup transforms 16 neighbouring columns in place, a wave 4 of them.  NBX = nx / 16.
// mode 0: forward along r, multiply by the kernel spectrum, inverse, store back
// mode 1: forward only; the conjugate IS the kernel spectrum (input = the padded kernel image)
#ifndef R2F_FFT_EXP2
#define R2F_FFT_EXP2 0  // development switch for pass 2: bit 0 no scratch loads, bit 1 no stores, bit 2 no spectrum loads
#endif
// element `idx` of a REAL kernel spectrum (one double per element of the scratch layout)
__device__ __forceinline__ double& rat(cplx* base, unsigned idx) {
    return *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + (idx << 3));
}
__device__ __forceinline__ double rat(const cplx* base, unsigned idx) {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + (idx << 3));
}

// KR: the kernel spectrum is real (FftConvArgs::kreal): 8 bytes per element and two multiplies instead of a complex product.
template <int NBX, int ST, bool KR>
__device__ __forceinline__ void fft_cols_body(const FftConvArgs& a, const int mode, double* fsm) {
    const int lane = threadIdx.x & 63, l = lane & 15;
    const int pair = blockIdx.y, k = blockIdx.x * 16 + (threadIdx.x >> 4);
    char* s1 = simg<ST>(a.s1, pair, (long long)kN * (NBX * 16));
    const cplx* kf = a.kfs[(a.pair0 + pair) / a.ppc];
#ifndef R2F_FFT_KR_EARLY
#define R2F_FFT_KR_EARLY 0  // (measured neutral either way: profiles/r05_fft_levers_ab.txt) bit 0: complex128 / 12-byte scratch, bit 1: complex64 scratch (there the 32 extra VGPRs spill)
#endif
    constexpr bool kEarly = (R2F_FFT_KR_EARLY >> (ST == 1 ? 1 : 0)) & 1;
    // The real spectrum's 16 loads go out AHEAD of the scratch loads (32 VGPRs held through the forward transform): L2 hits that
    // return first -- loads complete in order, so the transform can still start on the first scratch elements while the later ones
    // travel -- instead of a second exposed round trip between the two transforms.
    double kr[16];
    if (KR && kEarly && mode != 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) kr[q] = (R2F_FFT_EXP2 & 4) ? 0.5 + 0.25 * q : rat(kf, sidx(l + 16 * q, k, NBX));
    }
    cplx v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = (R2F_FFT_EXP2 & 1) ? make_double2(1.0 + m + l, 0.5 * k) : sld<ST>(s1, sidx(l + 16 * m, k, NBX));
    const cplx w1 = a.tw[l];
    double* tbuf = wave_tbuf(fsm);
    fft256<false>(v, w1, tbuf, lane);
    if (mode == 1) {
        // the spectrum carries the transforms' 1 / (ny nx) (a power of two: exact), so pass 3 rounds what it reads (R2F_FFT_EPI2)
        const double sc = R2F_FFT_EPI2 ? 1.0 / ((double)kN * (NBX * 16)) : 1.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (KR)
                rat(a.kf_out, sidx(l + 16 * q, k, NBX)) = v[q].x * sc;  // (the imaginary part of a centred symmetric kernel's spectrum is rounding noise)
            else
                at(a.kf_out, sidx(l + 16 * q, k, NBX)) = make_double2(v[q].x * sc, -v[q].y * sc);
        }
        return;
    }
    if (KR) {
        if (!kEarly) {
#pragma unroll
            for (int q = 0; q < 16; ++q) kr[q] = (R2F_FFT_EXP2 & 4) ? 0.5 + 0.25 * q : rat(kf, sidx(l + 16 * q, k, NBX));
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = make_double2(v[q].x * kr[q], v[q].y * kr[q]);
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = cmul(v[q], (R2F_FFT_EXP2 & 4) ? make_double2(0.5, 0.25 * q) : at(kf, sidx(l + 16 * q, k, NBX)));
    }
    fft256<true>(v, w1, tbuf, lane);
    if (R2F_FFT_EXP2 & 2) {
        if (v[3].x == 1.2345e300) sst<ST>(s1, 0, v[5]);
        return;
    }
    // pass 3 only reads the rows that hold valid outputs: [oy, oy + vy)
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if ((unsigned)(l + 16 * q - a.oy) < (unsigned)a.vy) sst<ST>(s1, sidx(l + 16 * q, k, NBX), v[q]);
}

// 512-row windows: 32 lanes per column, 8 columns per workgroup, 2 per wave; forward by fft512, back by fft512_rev.
// Spectrum rows are in fft512's output order (the kernel spectrum is built by the same pass, mode 1).  A wave touches two
// of the four interleaved columns of the scratch layout, i.e. 32-byte pieces: measured ~25 % slower per byte than the
// 256-row pass (the host's window choice prices that in; tall kernels have no alternative).  (Pairs as the fast grid index,
// to share spectrum blocks between the workgroups in flight, was slower for every shape: 7.04 -> 7.46 ms at 256 x 256.)
template <int NBX, int ST, bool KR>
__device__ __forceinline__ void fft_cols_y512_body(const FftConvArgs& a, const int mode, double* fsm) {
    const int lane = threadIdx.x & 63, l = lane & 31;
    const int pair = blockIdx.y, k = blockIdx.x * 8 + (threadIdx.x >> 5);
    char* s1 = simg<ST>(a.s1, pair, (long long)512 * (NBX * 16));
    cplx v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = sld<ST>(s1, sidx(l + 32 * m, k, NBX));
    const cplx w1 = a.tw512[l];
    double* tbuf = wave_tbuf(fsm);
    fft512<false>(v, w1, tbuf, lane);
    const cplx* kf = a.kfs[(a.pair0 + pair) / a.ppc];
    const int f0 = 256 * (l & 1) + (l >> 1);  // spectrum row of register q: f0 + 16 q
    if (mode == 1) {
        const double sc = R2F_FFT_EPI2 ? 1.0 / (512.0 * (NBX * 16)) : 1.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (KR)
                rat(a.kf_out, sidx(f0 + 16 * q, k, NBX)) = v[q].x * sc;
            else
                at(a.kf_out, sidx(f0 + 16 * q, k, NBX)) = make_double2(v[q].x * sc, -v[q].y * sc);
        }
        return;
    }
    if (KR) {
        double kr[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) kr[q] = rat(kf, sidx(f0 + 16 * q, k, NBX));
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = make_double2(v[q].x * kr[q], v[q].y * kr[q]);
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = cmul(v[q], at(kf, sidx(f0 + 16 * q, k, NBX)));
    }
    fft512_rev<true>(v, w1, tbuf, lane);
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if ((unsigned)(l + 32 * q - a.oy) < (unsigned)a.vy) sst<ST>(s1, sidx(l + 32 * q, k, NBX), v[q]);  // pass 3 only reads the valid rows
}

#ifndef R2F_FFT_WPE2
#define R2F_FFT_WPE2 2
#endif
template <int NBX, bool Y512, int ST, bool KR>
__global__ __launch_bounds__(kFftThreads) __attribute__((amdgpu_waves_per_eu(R2F_FFT_WPE2, 8))) void fft_cols_kernel(const FftConvArgs a, const int mode) {
    extern __shared__ __attribute__((aligned(16))) double fsm[];
    if (Y512)
        fft_cols_y512_body<NBX, ST, KR>(a, mode, fsm);
    else
        fft_cols_body<NBX, ST, KR>(a, mode, fsm);
}


// ------------------------------------------------------------------------------------- pass 2 as a walk over the launch's pairs
// 256-row windows with a real spectrum (both production stencils): a grid of resident workgroups (2 per CU), each walking the
// launch's pairs for ITS 16 columns -- grid (nx / 16, G): pair = blockIdx.y, + G, ... -- with the kernel spectrum's 16 reals per
// lane kept in registers from pair to pair (same columns, same channel) and no workgroup start-up per pair.  Frame -0.11 ms at
// 100 MP (4.93 -> 4.82, interleaved A/B, profiles/r05_fft_levers_ab.txt); alone on the GPU the complex64 pass gains (0.64 ->
// 0.61 ms) and the complex128 pass loses a little (0.82 -> 0.85), the frame gains either way.  A variant whose next pair travelled
// into a wave-private LDS tile by LDS-DMA (global_load_lds_dwordx4, counted vmcnt waits; complex64 only, 8 KB per wave) while the
// current one is transformed measured the same as this one (0.59-0.60 against 0.61 alone, 4.84 against 4.82 in the frame): what
// the pass does not hide is not the latency of its loads (round 3 found the same with a register prefetch); removed again.
// Round 6: fft256 re-cut around ONE LDS round trip with the round trip under the butterflies (a complex tile, the first DFT-16's
// outputs written group by group while the next group's butterflies issue, the reads back in the order the second DFT-16 takes
// them; the ISA came out as placed, results bit-identical) measured 0.623 against 0.628 ms alone on complex64, 0.894 against 0.885 on
// complex128, and the frame +0.02 ... +0.04 ms (interleaved, two builds in one process: profiles/r06_fft_levers_ab.txt): the LDS
// round trips are not what this pass fails to hide either; removed again (commit "Column walk: fft256 around one LDS round trip").
// Loop-invariant values the compiler would otherwise keep in VGPRs from pair to pair (the fifteen twiddle powers: 60 VGPRs; sixteen
// store offsets) are made opaque per iteration -- recomputing them is what the one-shot kernel does too: 207-209 VGPRs, no spill.
template <int NBX, int ST>
__device__ __forceinline__ void fft_cols_walk_body(const FftConvArgs& a, double* fsm) {
    const int lane = threadIdx.x & 63, l = lane & 15;
    const int k = blockIdx.x * 16 + (threadIdx.x >> 4);
    double* tbuf = wave_tbuf(fsm);
    const long long img = (long long)kN * (NBX * 16);
    const int G = gridDim.y;
    const cplx w1 = a.tw[l];
    double kr[16];
    int ci_have = -1;
    for (int pair = blockIdx.y; pair < a.npairs; pair += G) {
        const int gp = a.pair0 + pair, ci = gp / a.ppc;
        const bool packed = ST == 3 ? pair_packed(a, gp - ci * a.ppc) : false;  // (wave-uniform: a scalar load and a scalar branch)
        char* s1 = simg<ST>(a.s1, pair, img);
        unsigned sbase = sidx(l, k, NBX);  // element (row l + 16 q, column k) = sbase + q * NBX * 256
        cplx w = w1;
        asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(sbase));
        if (ci != ci_have) {
            const cplx* kf = a.kfs[ci];
#pragma unroll
            for (int q = 0; q < 16; ++q) kr[q] = rat(kf, sbase + q * (NBX * 256));
            ci_have = ci;
        }
        cplx v[16];
        if (ST == 3) {  // the transforms in between are shared: only the loads and the stores come in two forms
            if (packed) {
#pragma unroll
                for (int m = 0; m < 16; ++m) v[m] = sld<2>(s1, sbase + m * (NBX * 256));
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m) v[m] = sld<0>(s1, sbase + m * (NBX * 256));
            }
        } else {
            constexpr int STL = ST == 3 ? 0 : ST;
#pragma unroll
            for (int m = 0; m < 16; ++m) v[m] = sld<STL>(s1, sbase + m * (NBX * 256));
        }
        fft256<false>(v, w, tbuf, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = make_double2(v[q].x * kr[q], v[q].y * kr[q]);
        asm volatile("" : "+v"(w.x), "+v"(w.y));
        fft256<true>(v, w, tbuf, lane);
        if (ST == 3) {
            if (packed) {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if ((unsigned)(l + 16 * q - a.oy) < (unsigned)a.vy) sst<2>(s1, sbase + q * (NBX * 256), v[q]);
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if ((unsigned)(l + 16 * q - a.oy) < (unsigned)a.vy) sst<0>(s1, sbase + q * (NBX * 256), v[q]);
            }
        } else {
            constexpr int STS = ST == 3 ? 0 : ST;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if ((unsigned)(l + 16 * q - a.oy) < (unsigned)a.vy) sst<STS>(s1, sbase + q * (NBX * 256), v[q]);
        }
    }
}

template <int NBX, int ST>
__global__ __launch_bounds__(kFftThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void fft_cols_walk_kernel(const FftConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) double fsm[];
    fft_cols_walk_body<NBX, ST>(a, fsm);  // (ST = 3: the body reads every pair's own flag)
}

// ---------------------------------------------------------------------------------------------------- pass 3
// grid (ceil(vy / rows per workgroup), pairs); lanes and rows as in pass 1
#ifndef R2F_FFT_EXP3
#define R2F_FFT_EXP3 0  // development switch for pass 3: bit 0 no loads, bit 1 no stores, bit 2 no transform
#endif
#ifndef R2F_FFT_CURVE_BATCH
// outputs of a lane whose curve cells are gathered together.  16 (all of them) keeps 64 VGPRs of cells live and holds the kernel
// at 2 waves per SIMD; 4 fits 3 waves per SIMD without spills and is the faster one (halation 2.46 -> 2.43 ms at 100 MP)
#define R2F_FFT_CURVE_BATCH 4
#endif
template <int XL, int EPI, int ST>
__device__ __forceinline__ void fft_rows_inv_body(const FftConvArgs& a, double* fsm) {
    typedef RowGeom<XL> G;
    constexpr int NX = G::NX, LPL = G::LPL;
    const int lane = threadIdx.x & 63, l = lane & (LPL - 1);
    // The grid starts at the workgroup-aligned row at or above oy: a wave's 4 / 2 rows then share their 16-row scratch blocks
    // and 128-byte lines as they do for oy = 0 (an odd first row had cost the MTF's pass 0.10 of 0.44 ms).  ro: output row.
    const int pair = blockIdx.y, r = (a.oy & ~(G::ROWS - 1)) + blockIdx.x * G::ROWS + threadIdx.x / LPL, ro = r - a.oy;
    const bool live = (unsigned)ro < (unsigned)a.vy;  // dead lines still take part in the wave's transposes (their data is never stored)
    // ST = 3: the pair's flag is read here, ahead of the copy of the curve cells, so that its round trip is over when the loads need it
    bool packed = false;
    if (ST == 3) packed = pair_packed(a, (a.pair0 + pair) % a.ppc);
    // The epilogue's curve cells: 32 divergent 16-byte gathers per lane.  From global memory they go through the texture path at
    // 0.9 lanes per clock and CU (profiles/r02_gather_rate.txt) -- more of its cycles than all of the pass's coalesced scratch
    // loads and stores; from LDS at 4.3.  The workgroup's channel has (m - 1) cells of 16 bytes: copied behind the transpose
    // buffers when they fit (a.epi_lds_off, in doubles; 0 = gather from global memory).
    float4* cells_lds = reinterpret_cast<float4*>(fsm + a.epi_lds_off);
    if (EPI == 2) {
        const int gp0 = a.pair0 + pair, ch0 = a.chan[gp0 / a.ppc];
        const float4* src_cells = a.curve.cells + (long long)ch0 * (a.curve.m - 1);
        for (int i = threadIdx.x; i < a.curve.m - 1; i += kFftThreads) cells_lds[i] = src_cells[i];
        __syncthreads();
    }
    cplx v[16];
    if (R2F_FFT_EXP3 & 1) {
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = make_double2(1.0 + m + l, 0.5 * r);
    } else if (ST == 3) {
        const char* s1 = simg<3>(a.s1, pair, (long long)a.ny * NX);
        if (packed) {
#pragma unroll
            for (int m = 0; m < 16; ++m) v[m] = sld<2>(s1, sidx(live ? r : a.oy, l + LPL * m, G::NBX));
        } else {
#pragma unroll
            for (int m = 0; m < 16; ++m) v[m] = sld<0>(s1, sidx(live ? r : a.oy, l + LPL * m, G::NBX));
        }
    } else {
        constexpr int STL = ST == 3 ? 0 : ST;
        const char* s1 = simg<STL>(a.s1, pair, (long long)a.ny * NX);
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = sld<STL>(s1, sidx(live ? r : a.oy, l + LPL * m, G::NBX));
    }
    if (!(R2F_FFT_EXP3 & 4)) {
        G::template fft<true>(v, a, l, wave_tbuf(fsm), lane);
    }
    if (!live) return;
    if (R2F_FFT_EXP3 & 2) {
        if (v[3].x == 1.2345e300) a.dst.data[0] = (float)v[5].y;
        return;
    }
    const int gp = a.pair0 + pair, ci = gp / a.ppc, pc = gp - ci * a.ppc, ch = a.chan[ci];
    float* dplane = a.dst.data + (long long)ch * a.dst.plane_stride;
    const double scale = R2F_FFT_EPI2 ? 1.0 : 1.0 / ((double)NX * a.ny);  // a power of two; carried by the kernel spectrum (pass 2, mode 1)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int wy, wx;
        if (!window_of(a, 2 * pc + half, wy, wx)) continue;
        const int gy = wy + a.ay + ro;
        if (gy >= a.y1) continue;
        float* dbase = dplane + (long long)(gy - a.dst.gy0) * a.W + wx + a.ax;
        // the 16 outputs of this lane first, then the curve on all of them at once (independent gathers), then the stores
        float o[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) o[q] = R2F_FFT_EPI2 ? (float)(half ? v[q].y : v[q].x) : (float)((half ? v[q].y : v[q].x) * scale);
        if (EPI) {
            constexpr int CB = R2F_FFT_CURVE_BATCH;
#pragma unroll
            for (int b = 0; b < 16 / CB; ++b) {
                float ob[CB];
#ifndef R2F_FFT_EPI_ABLATE
#define R2F_FFT_EPI_ABLATE 0  // development switch: 1 no curve (log only), 2 no log (curve only), 4 one-instruction stand-in for both
#endif
#pragma unroll
                for (int q = 0; q < CB; ++q) ob[q] = (R2F_FFT_EPI_ABLATE & 6) ? o[CB * b + q] * 0.5f : log10_fast(o[CB * b + q], a.log_eps);
                if (R2F_FFT_EPI_ABLATE & 5) {
                } else if (EPI == 2)
                    curve_eval_near_lds<CB>((const float4*)cells_lds, a.curve, a.curve.f_first[ch], a.curve.f_last[ch], ob);
                else
                    curve_eval_batch<CB, 1>(a.curve.cells, a.curve, ch, ob);
#pragma unroll
                for (int q = 0; q < CB; ++q) o[CB * b + q] = ob[q];
            }
        }
        // output columns [0, c_end) of the window -- its scratch columns [ox, ox + c_end) -- are valid outputs inside the frame
        const int c_end = min(a.vx, a.W - (wx + a.ax));
        if (a.vec4) {
            // through this line's (idle) transpose buffer every lane gets four neighbouring columns and stores them as one
            // float4 (window origins and the frame width are multiples of 4 here); scratch column c lands at tf[c - ox + pad],
            // pad = ox rounded up to a multiple of 4, so that output column 0 sits on a 16-byte boundary (a line buffer holds
            // at least 2 nx + 32 floats and ox < nx / 2: room for the pad in front)
            float* tf = reinterpret_cast<float*>(G::line_buf(wave_tbuf(fsm), lane)) + ((a.ox + 3) & ~3);
#pragma unroll
            for (int q = 0; q < 16; ++q) tf[G::out_col(l, q) - a.ox] = o[q];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 4 * (l + LPL * j);
                const float4 val = *reinterpret_cast<const float4*>(tf + c);
                if (c < c_end) st4_stream(dbase + c, val);  // (a plane set: read back a stage later, 1.2 GB on)
            }
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if ((unsigned)(G::out_col(l, q) - a.ox) < (unsigned)c_end) dbase[G::out_col(l, q) - a.ox] = o[q];
        }
    }
}

#ifndef R2F_FFT_WPE3
#define R2F_FFT_WPE3 3
#endif
template <int XL, int EPI, int ST>
__global__ __launch_bounds__(kFftThreads) __attribute__((amdgpu_waves_per_eu(EPI ? R2F_FFT_WPE3 : (XL ? 3 : 4), 8))) void fft_rows_inv_kernel(const FftConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) double fsm[];
    fft_rows_inv_body<XL, EPI, ST>(a, fsm);  // (ST = 3: the body reads the choice ahead of its loads)
}

// ---------------------------------------------------------------------------------------------------- launchers
static size_t fft_lds_bytes() { return (size_t)(kFftThreads / 64) * 4 * kTLine * sizeof(double); }


template <int XL>
static void launch_rows_fwd(const FftConvArgs& a, hipStream_t s) {
    const dim3 block(kFftThreads), grid(a.ny / RowGeom<XL>::ROWS, a.npairs);
    if (a.s32 == 1)
        launch_k((fft_rows_fwd_kernel<XL, 1>), grid, block, fft_lds_bytes(), s, a);
    else if (a.s32 == 3)
        launch_k((fft_rows_fwd_kernel<XL, 3>), grid, block, fft_lds_bytes(), s, a);
    else if (a.s32 == 2)
        launch_k((fft_rows_fwd_kernel<XL, 2>), grid, block, fft_lds_bytes(), s, a);
    else
        launch_k((fft_rows_fwd_kernel<XL, 0>), grid, block, fft_lds_bytes(), s, a);
}

hipError_t launch_fft_rows_fwd(const FftConvArgs& a, hipStream_t s) {
    if (a.nx == 1024)
        launch_rows_fwd<2>(a, s);
    else if (a.nx == 512)
        launch_rows_fwd<1>(a, s);
    else
        launch_rows_fwd<0>(a, s);
    return take_launch_status();
}

template <int NBX, bool Y512, bool KR>
static void launch_cols_kr(const FftConvArgs& a, int mode, hipStream_t s) {
    const dim3 block(kFftThreads), grid(a.nx / (Y512 ? 8 : 16), a.npairs);
    if (a.s32 == 1)
        launch_k((fft_cols_kernel<NBX, Y512, 1, KR>), grid, block, fft_lds_bytes(), s, a, mode);
    else if (a.s32 == 2)
        launch_k((fft_cols_kernel<NBX, Y512, 2, KR>), grid, block, fft_lds_bytes(), s, a, mode);
    else
        launch_k((fft_cols_kernel<NBX, Y512, 0, KR>), grid, block, fft_lds_bytes(), s, a, mode);
}

template <int NBX, bool Y512>
static void launch_cols(const FftConvArgs& a, int mode, hipStream_t s) {
    if (a.kreal)
        launch_cols_kr<NBX, Y512, true>(a, mode, s);
    else
        launch_cols_kr<NBX, Y512, false>(a, mode, s);
}

template <bool Y512>
static void launch_cols_nx(const FftConvArgs& a, int mode, hipStream_t s) {
    if (a.nx == 1024)
        launch_cols<64, Y512>(a, mode, s);
    else if (a.nx == 512)
        launch_cols<32, Y512>(a, mode, s);
    else
        launch_cols<16, Y512>(a, mode, s);
}

// The walk applies to 256-row windows with a real spectrum in convolution mode.
static bool cols_walk_applies(const FftConvArgs& a, int mode) { return a.cols_walk && mode == 0 && a.ny == 256 && a.kreal; }

template <int NBX>
static void launch_cols_walk_st(const FftConvArgs& a, dim3 grid, hipStream_t s) {
    const dim3 block(kFftThreads);
    if (a.s32 == 1)
        launch_k((fft_cols_walk_kernel<NBX, 1>), grid, block, fft_lds_bytes(), s, a);
    else if (a.s32 == 3)
        launch_k((fft_cols_walk_kernel<NBX, 3>), grid, block, fft_lds_bytes(), s, a);
    else if (a.s32 == 2)
        launch_k((fft_cols_walk_kernel<NBX, 2>), grid, block, fft_lds_bytes(), s, a);
    else
        launch_k((fft_cols_walk_kernel<NBX, 0>), grid, block, fft_lds_bytes(), s, a);
}

static void launch_cols_walk(const FftConvArgs& a, hipStream_t s) {
    // as many workgroups as fit the GPU at two per CU (cols_slots), every one walking the same number of pairs when that divides
    const int cb = a.nx / 16, per = std::max(1, a.cols_slots / cb), iters = (a.npairs + per - 1) / per, G = (a.npairs + iters - 1) / iters;
    const dim3 grid(cb, G);
    if (a.nx == 1024)
        launch_cols_walk_st<64>(a, grid, s);
    else if (a.nx == 512)
        launch_cols_walk_st<32>(a, grid, s);
    else
        launch_cols_walk_st<16>(a, grid, s);
}

hipError_t fft_init_attributes() { return hipSuccess; }

hipError_t launch_fft_decide(const FftConvArgs& a, const RangeRecord& rec, float bound, float floor_, int* flags, hipStream_t s) {
    if (a.ppc <= 0) return hipSuccess;
    launch_k(fft_decide_kernel, dim3((a.ppc + 255) / 256), dim3(256), 0, s, a, rec, bound, floor_, flags);
    return take_launch_status();
}

hipError_t launch_fft_cols(const FftConvArgs& a, int mode, hipStream_t s) {
    if (cols_walk_applies(a, mode)) {
        launch_cols_walk(a, s);
        return take_launch_status();
    }
    if (a.ny == 512)
        launch_cols_nx<true>(a, mode, s);
    else
        launch_cols_nx<false>(a, mode, s);
    return take_launch_status();
}

template <int XL, int EPI>
static void launch_rows_inv(const FftConvArgs& a0, hipStream_t s) {
    const int rows = RowGeom<XL>::ROWS;
    const dim3 grid(((a0.oy & (rows - 1)) + a0.vy + rows - 1) / rows, a0.npairs);  // rows without valid outputs are never stored
    FftConvArgs a = a0;
    size_t lds = fft_lds_bytes();
    a.epi_lds_off = 0;
    // the epilogue's curve cells behind the transpose buffers, while three workgroups still fit a CU (the kernel's registers allow
    // three waves per SIMD): up to 1 152 cells (18 KB) -- a 1 024-point curve; longer curves gather from global memory
    if (EPI == 2) {
        a.epi_lds_off = (int)(lds / sizeof(double));
        lds += (size_t)(a.curve.m - 1) * sizeof(float4);
    }
    if (a.s32 == 1)
        launch_k((fft_rows_inv_kernel<XL, EPI, 1>), grid, dim3(kFftThreads), lds, s, a);
    else if (a.s32 == 3)
        launch_k((fft_rows_inv_kernel<XL, EPI, 3>), grid, dim3(kFftThreads), lds, s, a);
    else if (a.s32 == 2)
        launch_k((fft_rows_inv_kernel<XL, EPI, 2>), grid, dim3(kFftThreads), lds, s, a);
    else
        launch_k((fft_rows_inv_kernel<XL, EPI, 0>), grid, dim3(kFftThreads), lds, s, a);
}

template <int XL>
static void launch_rows_inv_epi(const FftConvArgs& a, hipStream_t s) {
    // the epilogue's curve cells behind the transpose buffers while three workgroups still fit a CU (the kernel's registers allow
    // three waves per SIMD): up to 1 152 cells (18 KB) -- a 1 024-point curve -- of a curve whose first guess is never more than
    // a cell off (DevCurve::near: any near-uniform axis); anything else gathers from global memory through the generic walk
    if (a.epilogue == 1 && a.epi_lds && a.curve.near && (size_t)(a.curve.m - 1) * sizeof(float4) + fft_lds_bytes() <= 53 * 1024)
        launch_rows_inv<XL, 2>(a, s);
    else if (a.epilogue == 1)
        launch_rows_inv<XL, 1>(a, s);
    else
        launch_rows_inv<XL, 0>(a, s);
}

hipError_t launch_fft_rows_inv(const FftConvArgs& a, hipStream_t s) {
    if (a.nx == 1024)
        launch_rows_inv_epi<2>(a, s);
    else if (a.nx == 512)
        launch_rows_inv_epi<1>(a, s);
    else
        launch_rows_inv_epi<0>(a, s);
    return take_launch_status();
}

}  // namespace r2f
