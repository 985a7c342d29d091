// Wave-level FFT core shared by fft.hip, samples.hip and stems.hip (gfx950): one 64-lane wave transforms one frame.
// A real n_fft-point transform is a complex M = n_fft/2 point Stockham autosort FFT: the lane holds
// M/64 points, the first radix-(M/64) pass runs in registers, two radix-8 passes exchange through a
// padded per-wave LDS buffer; an even/odd split recovers the n_fft/2+1 real-input bins.
#pragma once

#include "common.h"

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// Complex product in two packed instructions.  The compiler's own lowering of (ax bx - ay by, ax by + ay bx) builds the
// swapped operand with two v_mov per product (a fifth of the vector instructions of a 512-point transform); VOP3P's op_sel
// picks the halves directly:  t = (ax bx, ax by);  r = (-ay by + t.lo, ay bx + t.hi).
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    v2f_t va = {a.x, a.y}, vb = {b.x, b.y}, t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(va), "v"(vb));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(va), "v"(vb), "v"(t));
    return make_float2(r.x, r.y);
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)

// In-register forward DFTs (exp(-2 pi i nk/R)), natural-order output, decimation in frequency.
// The butterflies work on native two-element vectors (clang ext_vector_type): complex add / subtract are one packed
// instruction each, and the multiplications by -i and by (+-1 -+ i)/sqrt2 are element swaps + sign flips that the back end
// folds into the op_sel / neg modifiers of the packed add that consumes them — written on {x, y} structs the SLP
// vectoriser pairs unrelated scalars instead and spends a v_mov per element putting the pairs together.
typedef v2f_t cplx;
__device__ __forceinline__ cplx to_c(float2 a) { return cplx{a.x, a.y}; }
__device__ __forceinline__ float2 to_f2(cplx a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ cplx c_mi(cplx a) { return cplx{a.y, -a.x}; }                 // a * (-i)
template <int R> struct dft;

template <> struct dft<2> {
    __device__ __forceinline__ static void run(float2 *v)
    {
        const cplx a = to_c(v[0]), b = to_c(v[1]);
        v[0] = to_f2(a + b);
        v[1] = to_f2(a - b);
    }
};

// a + (-i) b and a - (-i) b as ONE packed add each: the swap of b's halves is op_sel, the sign a neg modifier
__device__ __forceinline__ cplx cadd_mi(cplx a, cplx b)
{
    cplx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));   // (ax + by, ay - bx)
    return r;
}
__device__ __forceinline__ cplx csub_mi(cplx a, cplx b)
{
    cplx r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));   // (ax - by, ay + bx)
    return r;
}

// irFFT input stage for bin k (conj trick: a real 2M-point inverse as a complex M-point forward transform):
//   A = X[k] + conj X[M-k],  D = X[k] - conj X[M-k],  C = wc D  (wc = conj of the half-bin twiddle),  point = conj(A + i C).
// The factor 1/2 of Z = (A + i C)/2 is NOT applied: the transform is linear, callers fold 0.5 into their output scale
// (exact, a power of two).  Five packed instructions.
__device__ __forceinline__ float2 irfft_pre(float2 xk, float2 xm, float2 wc)
{
    const cplx k = to_c(xk), m = to_c(xm);
    cplx A, D, r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(A) : "v"(k), "v"(m));                 // (kx + mx, ky - my)
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(D) : "v"(k), "v"(m));                 // (kx - mx, ky + my)
    const cplx C = to_c(cmul(wc, to_f2(D)));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(r) : "v"(A), "v"(C));   // (Ax - Cy, -Ay - Cx)
    return to_f2(r);
}

// rFFT output stage for bin k: X[k] = (Z[k] + conj Z[M-k])/2 - i/2 e^{-i pi k/M} (Z[k] - conj Z[M-k]) with wc the CONJUGATE of
// the half-bin twiddle (the table the inverse input stage uses):  A = zk + conj zm,  B = zk - conj zm,  C = conj(wc) B,
// X = ((A.x + C.y)/2, (A.y - C.x)/2).  Six packed instructions; written on {x, y} structs the compiler spends five more
// moves per bin putting the halves together.  Same operations, same roundings.
__device__ __forceinline__ float2 rfft_post(float2 zk, float2 zm, float2 wc)
{
    const cplx k = to_c(zk), m = to_c(zm), w = to_c(wc), half = {0.5f, 0.5f};
    cplx A, B, t, C, S, X;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(A) : "v"(k), "v"(m));                 // (kx + mx, ky - my)
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(B) : "v"(k), "v"(m));                 // (kx - mx, ky + my)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(w), "v"(B));              // cmul(conj(w), B), see cmul_conj
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(C) : "v"(w), "v"(B), "v"(t));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(S) : "v"(A), "v"(C));   // (Ax + Cy, Ay - Cx)
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(X) : "v"(S), "v"(half));
    return to_f2(X);
}

// ROT2: element 2 still has to be multiplied by -i (the W8^2 twiddle of the odd half of a radix-8 butterfly)
template <bool ROT2>
__device__ __forceinline__ void dft4_c(cplx *v)
{
    const cplx a0 = ROT2 ? cadd_mi(v[0], v[2]) : v[0] + v[2], a1 = ROT2 ? csub_mi(v[0], v[2]) : v[0] - v[2];
    const cplx a2 = v[1] + v[3], d = v[1] - v[3];
    v[0] = a0 + a2;
    v[2] = a0 - a2;
    v[1] = cadd_mi(a1, d);
    v[3] = csub_mi(a1, d);
}

template <> struct dft<4> {
    __device__ __forceinline__ static void run(float2 *v)
    {
        cplx c[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) c[n] = to_c(v[n]);
        dft4_c<false>(c);
#pragma unroll
        for (int n = 0; n < 4; ++n) v[n] = to_f2(c[n]);
    }
};

__device__ __forceinline__ void dft8_c(cplx *v)
{
    const float h = 0.70710678118654752440f;
    cplx e[4], o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        e[n] = v[n] + v[n + 4];
        o[n] = v[n] - v[n + 4];
    }
    o[1] = cadd_mi(o[1], o[1]) * h;                           // * W8^1 = (1-i)/sqrt2 : (h (x + y), h (y - x))
    o[3] = csub_mi(o[3], o[3]) * -h;                          // * W8^3 = (-1-i)/sqrt2 : (h (y - x), -h (x + y))
    dft4_c<false>(e);
    dft4_c<true>(o);                                          // * W8^2 = -i on o[2], folded into its first butterfly
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        v[2 * m] = e[m];
        v[2 * m + 1] = o[m];
    }
}

template <> struct dft<8> {
    __device__ __forceinline__ static void run(float2 *v)
    {
        cplx c[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) c[n] = to_c(v[n]);
        dft8_c(c);
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = to_f2(c[n]);
    }
};

template <> struct dft<16> {
    __device__ __forceinline__ static void run(float2 *v)
    {
        // W16^n, n = 0..7
        const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
        const float2 w[8] = {{1.f, 0.f}, {c1, -s1}, {h, -h}, {s1, -c1}, {0.f, -1.f}, {-s1, -c1}, {-h, -h}, {-c1, -s1}};
        cplx e[8], o[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            e[n] = to_c(v[n]) + to_c(v[n + 8]);
            o[n] = to_c(cmul(to_f2(to_c(v[n]) - to_c(v[n + 8])), w[n]));
        }
        dft8_c(e);
        dft8_c(o);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            v[2 * m] = to_f2(e[m]);
            v[2 * m + 1] = to_f2(o[m]);
        }
    }
};

// 32 points per lane: the 2048-point transform of the Bluestein path for even n_fft between 1024 and 2048 (one more
// decimation-in-frequency split in front of two 16-point transforms)
template <> struct dft<32> {
    __device__ __forceinline__ static void run(float2 *v)
    {
        const float2 w[16] = {{1.f, 0.f},
                              {0.98078528040323043f, -0.19509032201612825f},
                              {0.92387953251128674f, -0.38268343236508978f},
                              {0.83146961230254524f, -0.55557023301960218f},
                              {0.70710678118654757f, -0.70710678118654746f},
                              {0.55557023301960229f, -0.83146961230254524f},
                              {0.38268343236508984f, -0.92387953251128674f},
                              {0.19509032201612833f, -0.98078528040323043f},
                              {0.f, -1.f},
                              {-0.19509032201612819f, -0.98078528040323043f},
                              {-0.38268343236508973f, -0.92387953251128674f},
                              {-0.55557023301960196f, -0.83146961230254546f},
                              {-0.70710678118654746f, -0.70710678118654757f},
                              {-0.83146961230254535f, -0.55557023301960218f},
                              {-0.92387953251128674f, -0.38268343236508989f},
                              {-0.98078528040323043f, -0.19509032201612861f}};   // W32^n, n = 0..15
        float2 e[16], o[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            e[n] = cadd(v[n], v[n + 16]);
            const float2 d = make_float2(v[n].x - v[n + 16].x, v[n].y - v[n + 16].y);
            o[n] = n == 0 ? d : cmul(d, w[n]);
        }
        dft<16>::run(e);
        dft<16>::run(o);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            v[2 * m] = e[m];
            v[2 * m + 1] = o[m];
        }
    }
};

// Sizes with a factor 3 (n_fft 768 / 1536: M = 384 / 768 = 64 x 6 / 64 x 12): the first pass is a 6- or 12-point DFT, written
// as the plain sum over a table of the twelfth roots of unity (R^2 complex products: these sizes are entry-point
// conveniences — gf.synthesize / stft / istft take any n_fft, GOOFER.py:972 — not the hot geometry).
template <int R>
__device__ __forceinline__ void dft_direct(float2 *v)
{
    static_assert(12 % R == 0, "roots taken from the W12 table");
    const float2 w12[12] = {{1.f, 0.f}, {0.86602540378443865f, -0.5f}, {0.5f, -0.86602540378443865f}, {0.f, -1.f},
                            {-0.5f, -0.86602540378443865f}, {-0.86602540378443865f, -0.5f}, {-1.f, 0.f},
                            {-0.86602540378443865f, 0.5f}, {-0.5f, 0.86602540378443865f}, {0.f, 1.f},
                            {0.5f, 0.86602540378443865f}, {0.86602540378443865f, 0.5f}};
    float2 y[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        float2 acc = v[0];
#pragma unroll
        for (int n = 1; n < R; ++n) {
            const int e = ((n * t) % R) * (12 / R);
            const float2 p = (e == 0) ? v[n] : cmul(v[n], w12[e]);
            acc = cadd(acc, p);
        }
        y[t] = acc;
    }
#pragma unroll
    for (int t = 0; t < R; ++t) v[t] = y[t];
}
template <> struct dft<6> {
    __device__ __forceinline__ static void run(float2 *v) { dft_direct<6>(v); }
};
template <> struct dft<12> {
    __device__ __forceinline__ static void run(float2 *v) { dft_direct<12>(v); }
};

// index into the M-entry twiddle table: products t k STEP of a radix-8 pass stay below 7 M / 8, so no wrap is ever taken
// (the mask only documents the power-of-two sizes)
template <int M> __device__ __forceinline__ constexpr int tw_wrap(int i) { return (M & (M - 1)) == 0 ? (i & (M - 1)) : i; }

__host__ __device__ __forceinline__ constexpr int lds_pad(int i) { return i + (i >> 5); }

template <int M> struct fft_cfg {
    static constexpr int R = M / 64;           // points per lane == first-pass radix
    static constexpr int BUF = M + (M >> 5);   // padded float2 slots per wave
};

// Radix-8 Stockham pass over the per-wave LDS buffer.  NS = product of the radices already applied.
// Reads x[b + t*M/8], writes y[(b/NS)*NS*8 + b%NS + t*NS]; all reads precede all writes.
template <int M, int NS>
__device__ __forceinline__ void radix8_pass(float2 *buf, const float2 *tw, int lane)
{
    constexpr int NB = M / 8;                       // butterflies in this pass
    constexpr int PER = (NB + WAVE - 1) / WAVE;     // per lane
    float2 v[PER][8];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        int b = lane + WAVE * u;
        if (NB % WAVE == 0 || b < NB) {
#pragma unroll
            for (int t = 0; t < 8; ++t) v[u][t] = buf[lds_pad(b + t * NB)];
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        int b = lane + WAVE * u;
        if (NB % WAVE == 0 || b < NB) {
            int k = b % NS;
            // twiddle exp(-2 pi i t k / (8 NS)) = tw[t * k * (M / (8 NS))]
            constexpr int STEP = M / (8 * NS);
#pragma unroll
            for (int t = 1; t < 8; ++t) v[u][t] = cmul(v[u][t], tw[tw_wrap<M>(t * k * STEP)]);
            dft<8>::run(v[u]);
            int j0 = (b / NS) * NS * 8 + k;
#pragma unroll
            for (int t = 0; t < 8; ++t) buf[lds_pad(j0 + t * NS)] = v[u][t];
        }
    }
    wave_lds_sync();
}

// The 8 x 8 transpose between a lane's register index and bits 3..5 of its lane id — x[t] of lane (h, lo) <-> x[h] of lane
// (t, lo) — in registers: v_permlane32_swap (lane bit 5), v_permlane16_swap (bit 4), DPP row_ror:8 under bank masks (bit 3);
// each step swaps "a in the lanes with the bit set" with "b in the lanes with it clear".  32 vector instructions for eight
// float2 per lane, no LDS (checked on the device: scripts/micro/lane_transpose.hip).
__device__ __forceinline__ void swap_lane32(float &a, float &b)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap_lane16(float &a, float &b)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap_lane8(float &a, float &b)
{
    const int ai = __float_as_int(a), bi = __float_as_int(b);
    const int na = __builtin_amdgcn_update_dpp(ai, bi, 0x128, 0xf, 0xc, false);   // row_ror:8 into lanes 8..15 of each row
    const int nb = __builtin_amdgcn_update_dpp(bi, ai, 0x128, 0xf, 0x3, false);   // ... into lanes 0..7
    a = __int_as_float(na);
    b = __int_as_float(nb);
}
__device__ __forceinline__ void transpose_reg_lanehi(float2 (&x)[8])
{
#pragma unroll
    for (int t = 0; t < 4; ++t) { swap_lane32(x[t].x, x[t + 4].x); swap_lane32(x[t].y, x[t + 4].y); }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = (q & 1) | ((q & 2) << 1);                // 0, 1, 4, 5
        swap_lane16(x[t].x, x[t + 2].x);
        swap_lane16(x[t].y, x[t + 2].y);
    }
#pragma unroll
    for (int t = 0; t < 8; t += 2) { swap_lane8(x[t].x, x[t + 1].x); swap_lane8(x[t].y, x[t + 1].y); }
}

// Pass 0's output to pass 1's butterflies without LDS, for 8 or 16 points per lane (M = 512 / 1024), then pass 1 itself
// (radix8_pass<M, R>: same twiddles, same butterfly, same stores).  Pass 0 leaves y[R l + t] in register t of lane l; butterfly b
// of pass 1 wants y[b + t' M / 8], t' < 8, i.e. register b % R of the lanes (t', b / R) — eight lanes that differ in their top
// digit only — so after transpose_reg_lanehi on registers 8 g .. 8 g + 7, lane (s, lo) holds the inputs of butterfly
// b = R lo + 8 g + s.  Every value goes through the same operations as in radix8_pass (bit-identical; the next exchange, through
// LDS, puts the points back in order); 32 vector instructions per eight points instead of 16 LDS instructions, a barrier
// and an LDS round trip.
template <int M>
__device__ __forceinline__ void pass1_regx(const float2 *v, float2 *buf, const float2 *tw, int lane)
{
    constexpr int R = fft_cfg<M>::R, STEP = M / (8 * R);
    static_assert(R == 8 || R == 16, "eight-register groups");
    const int s = lane >> 3, lo = lane & 7;
#pragma unroll
    for (int g = 0; g < R / 8; ++g) {
        float2 x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = v[8 * g + t];
        transpose_reg_lanehi(x);
        const int k = 8 * g + s;                               // b % R
#pragma unroll
        for (int t = 1; t < 8; ++t) x[t] = cmul(x[t], tw[tw_wrap<M>(t * k * STEP)]);
        dft<8>::run(x);
        const int j0 = lo * R * 8 + k;
#pragma unroll
        for (int t = 0; t < 8; ++t) buf[lds_pad(j0 + t * R)] = x[t];
    }
    wave_lds_sync();
}
// (16 points per lane, n_fft 2048: measured no faster — k_irfft_ola1<1024> 8.93 -> 9.01 ms, eight more spilled registers in a kernel
// that already holds 256 — so only the eight-point sizes take it)
template <int M> constexpr bool fft_regx = fft_cfg<M>::R == 8;

// Complex forward FFT of size M for one wave.  `v` holds x[lane + 64 t], t < R on entry; the result
// is left in natural order in `buf` (padded).
template <int M>
__device__ __forceinline__ void wave_fft(float2 *v, float2 *buf, const float2 *tw, int lane)
{
    constexpr int R = fft_cfg<M>::R;
    dft<R>::run(v);                       // pass 0: NS = 1, no twiddle, y[lane*R + t]
    if constexpr (fft_regx<M>) {
        pass1_regx<M>(v, buf, tw, lane);
    } else {
#pragma unroll
        for (int t = 0; t < R; ++t) buf[lds_pad(lane * R + t)] = v[t];
        wave_lds_sync();
        radix8_pass<M, R>(buf, tw, lane);
    }
    radix8_pass<M, R * 8>(buf, tw, lane);
    static_assert(R * 64 == M, "M must be 64 * first radix");
}

// Same transform with the last pass left in registers (M >= 512).  In that pass NS = M/8, so butterfly b = lane + 64 u
// produces y[b + t M/8], t < 8 — exactly the points m = lane + 64 r (r = u + t M/512) this lane consumes next when the
// consumer works on m = lane + 64 r.  Saves the write, the exchange and the read back of the whole frame.
template <int M>
__device__ __forceinline__ void wave_fft_keep(float2 *v, float2 *buf, const float2 *tw, int lane, float2 *out)
{
    static_assert(M >= 512, "the last pass needs at least 64 butterflies");
    constexpr int R = fft_cfg<M>::R, NB = M / 8, PER = NB / WAVE;
    dft<R>::run(v);
    if constexpr (fft_regx<M>) {
        pass1_regx<M>(v, buf, tw, lane);
    } else {
#pragma unroll
        for (int t = 0; t < R; ++t) buf[lds_pad(lane * R + t)] = v[t];
        wave_lds_sync();
        radix8_pass<M, R>(buf, tw, lane);
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int b = lane + WAVE * u;
        float2 x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = buf[lds_pad(b + t * NB)];
#pragma unroll
        for (int t = 1; t < 8; ++t) x[t] = cmul(x[t], tw[tw_wrap<M>(t * b)]);    // NS = M/8: exp(-2 pi i t b / M)
        dft<8>::run(x);
#pragma unroll
        for (int t = 0; t < 8; ++t) out[u + PER * t] = x[t];
    }
    wave_lds_sync();
}

// The same with the two passes' twiddles in tables of their own (M = 512: one butterfly per lane and pass).  Indexing the
// M-entry table, pass 1 reads tw[t * (lane % 8) * 8] — eight addresses 64 t bytes apart, which fall on 4, 2 or 1 of the
// 64 banks (t = 4: an 8-way conflict) — and the last pass tw[t * lane], a stride of 8 t bytes over 64 lanes.  tw1[t][k]
// (7 x 8 entries) and tw2[t][lane] (7 x 64) hold the same values contiguously: every read is conflict-free.
template <int M> struct fft_tw_tabs {
    static constexpr int N1 = 7 * 8, N2 = 7 * WAVE;           // float2 entries
    static constexpr int TOTAL = N1 + N2;
};
template <int M>
__device__ __forceinline__ void fill_tw_tabs(float2 *tw1, float2 *tw2, const float2 *tw)
{
    static_assert(M == 512, "one butterfly per lane and pass");
    constexpr int R = fft_cfg<M>::R;
    for (int i = threadIdx.x; i < fft_tw_tabs<M>::N1; i += blockDim.x) {
        const int t = i / 8 + 1, k = i % 8;
        tw1[i] = tw[(t * k * (M / (8 * R))) & (M - 1)];
    }
    for (int i = threadIdx.x; i < fft_tw_tabs<M>::N2; i += blockDim.x) {
        const int t = i / WAVE + 1, b = i % WAVE;
        tw2[i] = tw[(t * b) & (M - 1)];
    }
}
// The first exchange runs in registers (pass1_regx above, with the pass's twiddles from their own table): the lane runs butterfly
// b = 8 (lane % 8) + lane / 8 of pass 1, whose stores 64 (lane % 8) + lane / 8 + 8 t fall on 32 different banks per 16 lanes.
template <int M>
__device__ __forceinline__ void wave_fft_keep_tab(float2 *v, float2 *buf, const float2 *tw1, const float2 *tw2, int lane, float2 *out)
{
    static_assert(M == 512, "one butterfly per lane and pass");
    constexpr int R = fft_cfg<M>::R, NB = M / 8;
    dft<R>::run(v);
    {
        float2 x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = v[t];
        transpose_reg_lanehi(x);
        const int k = lane >> 3;                               // b % 8 of this lane's butterfly
#pragma unroll
        for (int t = 1; t < 8; ++t) x[t] = cmul(x[t], tw1[(t - 1) * 8 + k]);
        dft<8>::run(x);
        const int j0 = (lane & 7) * R * 8 + k;
#pragma unroll
        for (int t = 0; t < 8; ++t) buf[lds_pad(j0 + t * R)] = x[t];
        wave_lds_sync();
    }
    float2 x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = buf[lds_pad(lane + t * NB)];
#pragma unroll
    for (int t = 1; t < 8; ++t) x[t] = cmul(x[t], tw2[(t - 1) * WAVE + lane]);
    dft<8>::run(x);
#pragma unroll
    for (int t = 0; t < 8; ++t) out[t] = x[t];
    wave_lds_sync();
}

// The same with both passes' twiddles held in registers (M = 512: one butterfly per lane and pass, 2 x 7 twiddles that
// depend on the lane only): a caller that transforms many frames loads them once and saves 14 LDS reads per transform.
template <int M>
__device__ __forceinline__ void fft_lane_twiddles(const float2 *tw, int lane, float2 *tw1, float2 *tw2)
{
    static_assert(M == 512, "one butterfly per lane and pass");
    constexpr int R = fft_cfg<M>::R;
    const int k1 = lane >> 3;                                 // pass 1: NS = R, twiddle exp(-2 pi i t k / (8 R)), k = b % 8 of the lane's butterfly b = 8 (lane % 8) + lane / 8 (pass1_regx)
#pragma unroll
    for (int t = 1; t < 8; ++t) {
        tw1[t - 1] = tw[(t * k1 * (M / (8 * R))) & (M - 1)];
        tw2[t - 1] = tw[(t * lane) & (M - 1)];                // last pass: NS = M/8, exp(-2 pi i t b / M)
    }
}

template <int M>
__device__ __forceinline__ void wave_fft_keep_tw(float2 *v, float2 *buf, const float2 *tw1, const float2 *tw2, int lane, float2 *out)
{
    static_assert(M == 512, "one butterfly per lane and pass");
    constexpr int R = fft_cfg<M>::R, NB = M / 8;
    dft<R>::run(v);
    {   // pass 1 (NS = R) on butterfly b = 8 (lane % 8) + lane / 8, its inputs through the register transpose (pass1_regx)
        float2 x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = v[t];
        transpose_reg_lanehi(x);
#pragma unroll
        for (int t = 1; t < 8; ++t) x[t] = cmul(x[t], tw1[t - 1]);
        dft<8>::run(x);
        const int j0 = (lane & 7) * R * 8 + (lane >> 3);
#pragma unroll
        for (int t = 0; t < 8; ++t) buf[lds_pad(j0 + t * R)] = x[t];
        wave_lds_sync();
    }
    float2 x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = buf[lds_pad(lane + t * NB)];
#pragma unroll
    for (int t = 1; t < 8; ++t) x[t] = cmul(x[t], tw2[t - 1]);
    dft<8>::run(x);
#pragma unroll
    for (int t = 0; t < 8; ++t) out[t] = x[t];
    wave_lds_sync();
}

template <int M>
__device__ __forceinline__ void load_tables(float2 *tw, float2 *twh, float *win, const float2 *g_tw,
                                            const float2 *g_twh, const float *g_win)
{
    for (int i = threadIdx.x; i < M; i += blockDim.x) tw[i] = g_tw[i];
    for (int i = threadIdx.x; i < M / 2 + 1; i += blockDim.x) twh[i] = g_twh[i];
    for (int i = threadIdx.x; i < 2 * M; i += blockDim.x) win[i] = g_win[i];
    __syncthreads();
}

constexpr int FRAMES_PER_BLOCK = 32;   // contiguous frames per workgroup: overlapping reads stay in L1/L2
constexpr int WAVES_PER_BLOCK = 4;

template <int M> constexpr size_t fft_lds_bytes()
{
    return sizeof(float2) * (M + M / 2 + 1 + WAVES_PER_BLOCK * fft_cfg<M>::BUF) + sizeof(float) * 2 * M + 16;
}


// irFFT input stage: X[k], k = 0..M (any addressable array: global row or LDS) -> the M/64 values this
// lane feeds into wave_fft (conj trick).  Im(DC) and Im(Nyquist) are ignored like pocketfft's c2r.
template <int M, typename Spec>
__device__ __forceinline__ void irfft_load(float2 *v, const Spec &X, const float2 *twh, int lane)
{
    constexpr int R = fft_cfg<M>::R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int k = lane + WAVE * r;
        float2 xk = X(k);
        float2 xm = X(M - k);
        if (k == 0) { xk.y = 0.f; xm.y = 0.f; }
        float2 wc = (k <= M / 2) ? cconj(twh[k]) : make_float2(-twh[M - k].x, -twh[M - k].y);
        v[r] = irfft_pre(xk, xm, wc);                                     // conj(A + iC); the 1/2 rides on irfft_store's scale
    }
}

// irFFT output stage: natural-order result in `buf` -> windowed time frame (`frames*window`, GOOFER.py:383)
template <int M>
__device__ __forceinline__ void irfft_store(const float2 *buf, const float *win, float *frame, int lane)
{
    constexpr int R = fft_cfg<M>::R;
    const float inv_m = 0.5f / (float)M;                    // 1/M of the transform and the 1/2 of the input stage
    float2 *out = reinterpret_cast<float2 *>(frame);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int m = lane + WAVE * r;
        float2 z = buf[lds_pad(m)];
        float a = z.x * inv_m, b = -z.y * inv_m;
        out[m] = make_float2(a * win[2 * m], b * win[2 * m + 1]);
    }
}

// rFFT input stage: M/64 (even, odd) windowed sample pairs of the frame starting at un-padded sample
// `start` of a note of n samples (numpy 'reflect' padding; n == 1 is 'edge')   GOOFER.py:358-369
template <int M>
__device__ __forceinline__ void rfft_load(float2 *v, const float *xs, int64_t start, int64_t n, const float *win, int lane)
{
    constexpr int R = fft_cfg<M>::R;
    if (start >= 0 && start + 2 * M <= n) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int m = lane + WAVE * r;
            v[r] = make_float2(xs[start + 2 * m] * win[2 * m], xs[start + 2 * m + 1] * win[2 * m + 1]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int m = lane + WAVE * r;
            float a = n > 0 ? xs[reflect_index(start + 2 * m, n)] : 0.f;
            float b = n > 0 ? xs[reflect_index(start + 2 * m + 1, n)] : 0.f;
            v[r] = make_float2(a * win[2 * m], b * win[2 * m + 1]);
        }
    }
}

// rFFT output stage: even/odd split of the complex result in `buf` -> X[k], k = 0..M, through a sink
//   X[k] = (Z[k] + conj Z[M-k])/2 - i/2 e^{-i pi k/M} (Z[k] - conj Z[M-k])
template <int M, typename Sink>
__device__ __forceinline__ void rfft_split(const float2 *buf, const float2 *twh, int lane, Sink &&put)
{
    constexpr int R = fft_cfg<M>::R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int k = lane + WAVE * r;
        float2 zk = buf[lds_pad(k)];
        float2 zm = buf[lds_pad(k == 0 ? 0 : M - k)];
        float2 w = (k <= M / 2) ? twh[k] : make_float2(-twh[M - k].x, twh[M - k].y);
        float2 A = make_float2(zk.x + zm.x, zk.y - zm.y);
        float2 B = make_float2(zk.x - zm.x, zk.y + zm.y);
        float2 C = cmul(w, B);
        put(k, make_float2(0.5f * (A.x + C.y), 0.5f * (A.y - C.x)));
    }
    if (lane == 0) {
        float2 z0 = buf[0];
        put(M, make_float2(z0.x - z0.y, 0.f));
    }
}
