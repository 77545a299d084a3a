// Pieces shared by the fp32 (igemm.hip) and bf16-activation (bgemm.hip) implicit-GEMM kernels: vector typedefs, the GEGLU gate's
// erf-GELU and the epilogues that take 32x32 MFMA accumulators (WEIGHT fragment as the A operand) to memory.
#pragma once
#include "kernels.h"

namespace e2v {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


// erf-GELU of the GEGLU gate (diffusers GEGLU.gelu = F.gelu, exact form).  The library erff costs 38 vector instructions with
// three branches, and on gfx950 they are paid at the fp32-MFMA's own rate (the GEGLU tiles spent a fifth of their time in the
// epilogue); Abramowitz-Stegun 7.1.26 -- erf(z) = 1 - (a1 t + .. + a5 t^5) exp(-z^2), t = 1 / (1 + p z), |error| <= 1.5e-7,
// i.e. at the level of fp32 rounding of values in [-1, 1] -- is 14, branch-free.  Measured |gelu error| <= 4.6e-7.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);
    const float erf_abs = fmaf(-p, e, 1.0f);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Epilogue shared by the fp32, bf16 and split-bf16 tiles (all use 32x32 MFMA results with the WEIGHT fragment as the A
// operand): in each 32x32 result a lane holds pixel m = lane & 31 and output channels n = 8 g + 4 (lane >> 5) + e in register
// r = 4 g + e -- four consecutive channels per register quad, i.e. 16-byte bias / time-embedding / residual loads and 16-byte
// stores (a short-K layer -- every Winograd-domain GEMM, every attention projection -- spends a tenth of its time here).
template <int BM, int TM, int TN, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& p, f32x16 (&acc)[TM][TN], float* __restrict__ out, const int bm,
                                               const int n0, const int wm, const int wn, const int lane, float* stage) {
    const int mrow = lane & 31;
    const int nq = (lane >> 5) * 4;
    if (p.geglu) {
        if constexpr (TN == 2) {
            const int nb = n0 + wn * WN;                           // value columns nb .. nb+31, gate columns nb+32 .. nb+63
            if (nb + 64 <= p.N) {
                // v * gelu(g) in the accumulator layout, staged per wave as WM x 32 and written row-contiguously (8 lanes = one
                // 128-byte row segment) like the plain epilogue below
                constexpr int SLD = 36;
                float* st = stage + (size_t)(threadIdx.x >> 6) * 32 * SLD;
                const int col = (lane & 7) * 4;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nb + 8 * g + nq) : f32x4{0.f, 0.f, 0.f, 0.f};
                        const f32x4 bg = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nb + 32 + 8 * g + nq) : f32x4{0.f, 0.f, 0.f, 0.f};
                        f32x4 y;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            y[e] = (acc[mi][0][4 * g + e] * p.alpha + bv[e]) * gelu_erf(acc[mi][1][4 * g + e] * p.alpha + bg[e]);
                        *reinterpret_cast<f32x4*>(st + mrow * SLD + 8 * g + nq) = y;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = (lane >> 3) + 8 * i;
                        const int m = bm * BM + wm * WM + mi * 32 + row;
                        if (m < p.M)
                            *reinterpret_cast<f32x4*>(out + (size_t)m * p.ldc + nb / 2 + col) = *reinterpret_cast<const f32x4*>(st + row * SLD + col);
                    }
                }
            }
        }
        return;
    }
    const bool vec = ((p.N | p.ldc | p.ldr | p.rb_ld) & 3) == 0;   // strides of absent operands are 0
    if (vec) {
        // Row-contiguous epilogue: the wave transposes its accumulators, 32 rows at a time, through LDS (the tile buffers are dead: every
        // real fragment read precedes the last barrier of the k-loop) so that a 16-byte access instruction covers whole rows --
        // 64 / (WN / 4) rows x WN floats -- instead of 32 rows x 32 bytes.  Residual reads and stores then move full 128-byte
        // lines (+5 % on the K = 320 layers, whose output write is a tenth of their time).
        constexpr int SLD = WN + 4;                            // padded staging row: conflict-free b128 writes and reads
        constexpr int LPR = WN / 4;                            // lanes per staged row
        constexpr int RPI = 64 / LPR;                          // rows per access instruction
        float* st = stage + (size_t)(threadIdx.x >> 6) * 32 * SLD;   // one 32-row staging block per wave, reused per mi
        const int col = (lane % LPR) * 4;
        const int n = n0 + wn * WN + col;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = acc[mi][ni][4 * g + e];
                    *reinterpret_cast<f32x4*>(st + mrow * SLD + ni * 32 + 8 * g + nq) = y;
                }
#pragma unroll
            for (int i = 0; i < 32 / RPI; ++i) {
                const int row = lane / LPR + RPI * i;
                const int m = bm * BM + wm * WM + mi * 32 + row;
                if (m >= p.M || n >= p.N) continue;            // N % 4 == 0: the lane's four columns are in or out together
                f32x4 y = *reinterpret_cast<const f32x4*>(st + row * SLD + col);
                if (p.alpha != 1.0f) y *= p.alpha;             // only the VAE attention scales
                y += bias4;
                if (p.rowbias) y += *reinterpret_cast<const f32x4*>(p.rowbias + (size_t)(m / p.rows_per_sample) * p.rb_ld + n);
                if (p.resid) y += *reinterpret_cast<const f32x4*>(p.resid + (size_t)m * p.ldr + n);
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
                }
                *reinterpret_cast<f32x4*>(out + (size_t)m * p.ldc + n) = y;
            }
        }
        return;
    }
    // scalar fallback (N, or a stride, not a multiple of 4: the VAE's 3-channel conv_out)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = bm * BM + wm * WM + mi * 32 + mrow;
        if (m >= p.M) continue;
        const float* rbp = p.rowbias ? p.rowbias + (size_t)(m / p.rows_per_sample) * p.rb_ld : nullptr;
        const float* rsp = p.resid ? p.resid + (size_t)m * p.ldr : nullptr;
        float* orow = out + (size_t)m * p.ldc;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * WN + ni * 32 + 8 * g + nq;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= p.N) break;
                    float v = acc[mi][ni][4 * g + e] * p.alpha + (p.bias ? p.bias[n + e] : 0.f);
                    if (rbp) v += rbp[n + e];
                    if (rsp) v += rsp[n + e];
                    if (p.relu) v = fmaxf(v, 0.f);
                    orow[n + e] = v;
                }
            }
    }
}



// bf16 OUTPUT (bf16-activation mode): same staging through LDS, but a lane finishes 8 consecutive columns -- two 16-byte
// reads of the fp32 staging row, bias / time-embedding row in fp32, the residual as 8 bf16 (16 bytes), one rounding to bf16 at
// the very end, a 16-byte store.  `out` / `p.resid` are bf16 (p.resid_bf16) or the residual is fp32 (latents never are).
template <int BM, int TM, int TN, int WM, int WN>
__device__ __forceinline__ void bgemm_epilogue_bf16(const IgemmArgs& p, f32x16 (&acc)[TM][TN], __bf16* __restrict__ out, const int bm,
                                                    const int n0, const int wm, const int wn, const int lane, float* stage) {
    const int mrow = lane & 31;
    const int nq = (lane >> 5) * 4;
    if (p.geglu) {
        if constexpr (TN == 2) {
            const int nb = n0 + wn * WN;                           // value columns nb .. nb+31, gate columns nb+32 .. nb+63
            if (nb + 64 <= p.N) {
                constexpr int SLD = 36;
                float* st = stage + (size_t)(threadIdx.x >> 6) * 32 * SLD;
                const int col = (lane & 3) * 8;                    // 4 lanes x 8 bf16 = one 64-byte output row segment
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nb + 8 * g + nq) : f32x4{0.f, 0.f, 0.f, 0.f};
                        const f32x4 bg = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nb + 32 + 8 * g + nq) : f32x4{0.f, 0.f, 0.f, 0.f};
                        f32x4 y;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            y[e] = (acc[mi][0][4 * g + e] * p.alpha + bv[e]) * gelu_erf(acc[mi][1][4 * g + e] * p.alpha + bg[e]);
                        *reinterpret_cast<f32x4*>(st + mrow * SLD + 8 * g + nq) = y;
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = (lane >> 2) + 16 * i;
                        const int m = bm * BM + wm * WM + mi * 32 + row;
                        if (m < p.M) {
                            const f32x4 y0 = *reinterpret_cast<const f32x4*>(st + row * SLD + col);
                            const f32x4 y1 = *reinterpret_cast<const f32x4*>(st + row * SLD + col + 4);
                            bf16x8 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { o[e] = (__bf16)y0[e]; o[4 + e] = (__bf16)y1[e]; }
                            *reinterpret_cast<bf16x8*>(out + (size_t)m * p.ldc + nb / 2 + col) = o;
                        }
                    }
                }
            }
        }
        return;
    }
    const bool vec = ((p.N | p.ldc | p.ldr) & 7) == 0 && (p.rb_ld & 3) == 0;
    if (vec) {
        constexpr int SLD = WN + 4;
        constexpr int LPR = WN / 8;                            // lanes per staged row (8 columns each)
        constexpr int RPI = 64 / LPR;                          // rows per access instruction
        float* st = stage + (size_t)(threadIdx.x >> 6) * 32 * SLD;
        const int col = (lane % LPR) * 8;
        const int n = n0 + wn * WN + col;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) {
            b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
            b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = acc[mi][ni][4 * g + e];
                    *reinterpret_cast<f32x4*>(st + mrow * SLD + ni * 32 + 8 * g + nq) = y;
                }
#pragma unroll
            for (int i = 0; i < 32 / RPI; ++i) {
                const int row = lane / LPR + RPI * i;
                const int m = bm * BM + wm * WM + mi * 32 + row;
                if (m >= p.M || n >= p.N || (p.ablate & 1)) continue;   // N % 8 == 0: the lane's eight columns are in or out together
                f32x4 y0 = *reinterpret_cast<const f32x4*>(st + row * SLD + col);
                f32x4 y1 = *reinterpret_cast<const f32x4*>(st + row * SLD + col + 4);
                if (p.alpha != 1.0f) { y0 *= p.alpha; y1 *= p.alpha; }
                y0 += b0; y1 += b1;
                if (p.rowbias) {
                    const float* rb = p.rowbias + (size_t)(m / p.rows_per_sample) * p.rb_ld + n;
                    y0 += *reinterpret_cast<const f32x4*>(rb);
                    y1 += *reinterpret_cast<const f32x4*>(rb + 4);
                }
                if (p.resid) {
                    if (p.resid_bf16) {
                        const bf16x8 r = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(p.resid) + (size_t)m * p.ldr + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { y0[e] += (float)r[e]; y1[e] += (float)r[4 + e]; }
                    } else {
                        y0 += *reinterpret_cast<const f32x4*>(p.resid + (size_t)m * p.ldr + n);
                        y1 += *reinterpret_cast<const f32x4*>(p.resid + (size_t)m * p.ldr + n + 4);
                    }
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { y0[e] = fmaxf(y0[e], 0.f); y1[e] = fmaxf(y1[e], 0.f); }
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (__bf16)y0[e]; o[4 + e] = (__bf16)y1[e]; }
                *reinterpret_cast<bf16x8*>(out + (size_t)m * p.ldc + n) = o;
            }
        }
        return;
    }
    // scalar fallback (N or a stride not a multiple of 8)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = bm * BM + wm * WM + mi * 32 + mrow;
        if (m >= p.M) continue;
        const float* rbp = p.rowbias ? p.rowbias + (size_t)(m / p.rows_per_sample) * p.rb_ld : nullptr;
        __bf16* orow = out + (size_t)m * p.ldc;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * WN + ni * 32 + 8 * g + nq;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= p.N) break;
                    float v = acc[mi][ni][4 * g + e] * p.alpha + (p.bias ? p.bias[n + e] : 0.f);
                    if (rbp) v += rbp[n + e];
                    if (p.resid) v += p.resid_bf16 ? (float)reinterpret_cast<const __bf16*>(p.resid)[(size_t)m * p.ldr + n + e]
                                                   : p.resid[(size_t)m * p.ldr + n + e];
                    if (p.relu) v = fmaxf(v, 0.f);
                    orow[n + e] = (__bf16)v;
                }
            }
    }
}

}  // namespace e2v
