// r2f_fft_math.h -- fp64 complex helpers and the register DFTs of the overlap-save FFT kernels (r2f_fft.hip: three passes over a
// global scratch image).
#pragma once

#include <hip/hip_runtime.h>

namespace r2f {

typedef double2 cplx;

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) {  // a * conj(b)
    return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
template <bool INV>
__device__ __forceinline__ cplx ctw(cplx a, cplx w) {  // a * w (forward) or a * conj(w) (inverse)
    return INV ? cmulc(a, w) : cmul(a, w);
}

// 4-point DFT, natural order in and out (forward kernel exp(-2 pi i / 4) = -i).
template <bool INV>
__device__ __forceinline__ void dft4(cplx& a, cplx& b, cplx& c, cplx& d) {
    const cplx s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = csub(b, d);
    const cplx r3 = INV ? make_double2(-s3.y, s3.x) : make_double2(s3.y, -s3.x);  // (-+ i) * s3
    a = cadd(s0, s2);
    c = csub(s0, s2);
    b = cadd(s1, r3);
    d = csub(s1, r3);
}

// 16-point DFT in registers, natural order in and out: 4 x 4 Cooley-Tukey with the constants W_16^(c r).
template <bool INV>
__device__ __forceinline__ void dft16(cplx (&v)[16]) {
    constexpr double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4<INV>(v[c], v[c + 4], v[c + 8], v[c + 12]);  // u[c][r] at v[c + 4 r]
    // u[c][r] *= W_16^(c r): W^1 = (c1, -s1), W^2 = (h, -h), W^3 = (s1, -c1), W^4 = -i, W^6 = (-h, -h), W^9 = (-c1, s1)
    v[1 + 4] = ctw<INV>(v[1 + 4], make_double2(c1, -s1));
    v[1 + 8] = ctw<INV>(v[1 + 8], make_double2(h, -h));
    v[1 + 12] = ctw<INV>(v[1 + 12], make_double2(s1, -c1));
    v[2 + 4] = ctw<INV>(v[2 + 4], make_double2(h, -h));
    v[2 + 8] = ctw<INV>(v[2 + 8], make_double2(0.0, -1.0));
    v[2 + 12] = ctw<INV>(v[2 + 12], make_double2(-h, -h));
    v[3 + 4] = ctw<INV>(v[3 + 4], make_double2(s1, -c1));
    v[3 + 8] = ctw<INV>(v[3 + 8], make_double2(-h, -h));
    v[3 + 12] = ctw<INV>(v[3 + 12], make_double2(-c1, s1));
#pragma unroll
    for (int r = 0; r < 4; ++r) dft4<INV>(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]);  // out[r + 4 s] at v[4 r + s]
    // 4 x 4 index transpose (register renaming only)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = r + 1; s < 4; ++s) {
            const cplx t = v[4 * r + s];
            v[4 * r + s] = v[4 * s + r];
            v[4 * s + r] = t;
        }
}

// v[p] *= w1^p (conjugated powers for the inverse); the powers come from a multiplication tree of depth <= 5 (a few ulp)
template <bool INV>
__device__ __forceinline__ void twiddle_powers(cplx (&v)[16], const cplx w1) {
    const cplx w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2), w5 = cmul(w4, w1), w6 = cmul(w3, w3),
               w7 = cmul(w4, w3), w8 = cmul(w4, w4);
    v[1] = ctw<INV>(v[1], w1);
    v[2] = ctw<INV>(v[2], w2);
    v[3] = ctw<INV>(v[3], w3);
    v[4] = ctw<INV>(v[4], w4);
    v[5] = ctw<INV>(v[5], w5);
    v[6] = ctw<INV>(v[6], w6);
    v[7] = ctw<INV>(v[7], w7);
    v[8] = ctw<INV>(v[8], w8);
    v[9] = ctw<INV>(v[9], cmul(w8, w1));
    v[10] = ctw<INV>(v[10], cmul(w5, w5));
    v[11] = ctw<INV>(v[11], cmul(w8, w3));
    v[12] = ctw<INV>(v[12], cmul(w6, w6));
    v[13] = ctw<INV>(v[13], cmul(w8, w5));
    v[14] = ctw<INV>(v[14], cmul(w7, w7));
    v[15] = ctw<INV>(v[15], cmul(w8, w7));
}

}  // namespace r2f
