// Pieces shared by the fp32 (igemm.hip) and bf16-activation (bgemm.hip) implicit-GEMM kernels: vector typedefs, the GEGLU gate's
// erf-GELU and the epilogues that take 32x32 MFMA accumulators (WEIGHT fragment as the A operand) to memory.
#pragma once
#include "h16.h"
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

// The same gate in the bf16-activation mode, where the product value * gelu(gate) is rounded to bf16 (2^-9 relative) on its way out:
// gelu(x) = x Phi(x) with the normal CDF as a logistic of an odd cubic, Phi(x) ~ 1 / (1 + exp(-(a x + b x^3))), a = 1.60031415,
// b = 0.06940179 (minimax fit of x Phi(x) over the real line; monotone, so no clamp): |gelu error| <= 2.8e-4 = 1/14 of the rounding
// of a bf16 value near 1 (bound restated in tests/test_oracle_anchors.py).  Five plain vector instructions + v_exp_f32 + v_rcp_f32 instead
// of thirteen + two: the GEGLU 320 -> 2560 tiles of a B = 32 pass spent as long in this epilogue (64 gates per lane) as in their five
// K steps.  fp32 mode keeps gelu_erf above.
__device__ __forceinline__ float gelu_bf16_grade(float x) {
    const float u = x * x;
    const float t = x * fmaf(-0.06940179f * 1.44269504088896340736f, u, -1.60031415f * 1.44269504088896340736f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
// (fast: wave-uniform, IgemmArgs::a_bf16 -- the arithmetic mode, never the kernel or the batch, picks the form: the logistic form in
// bf16 mode only; fp16 rounds 8x finer than bf16, so the fp16 mode keeps the erf form like fp32)
__device__ __forceinline__ float gelu_gate(float x, const bool fast) { return fast ? gelu_bf16_grade(x) : gelu_erf(x); }
// (E2V_F16_FAST_GATE: a measurement build -- `make EXTRA=-DE2V_F16_FAST_GATE` -- that gives the fp16 mode the logistic gate too, to
// price the erf form: profiles/r05_fp16_gate_ab.log)
#ifdef E2V_F16_FAST_GATE
constexpr bool kF16FastGate = true;
#else
constexpr bool kF16FastGate = false;
#endif
template <typename H> __device__ __forceinline__ float gelu_gate16(float x) {       // the same choice at compile time, by the kernel's 16-bit type
    if constexpr (__is_same(H, __bf16) || kF16FastGate) return gelu_bf16_grade(x);
    else return gelu_erf(x);
}

typedef hx4<__bf16> bf16x4;        // (the split-bf16 fp32 path of igemm.hip names its operands)
typedef hx8<__bf16> bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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
                            y[e] = (acc[mi][0][4 * g + e] * p.alpha + bv[e]) * gelu_gate(acc[mi][1][4 * g + e] * p.alpha + bg[e], p.a_bf16 == H16_BF16 || (kF16FastGate && p.a_bf16 == H16_FP16));
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
// SWZ (the persistent kernel): the staging area is the stage buffer the tile has just consumed and no byte more -- 32 rows x WN
// floats per wave, UNPADDED, the 16-byte chunk of a row XOR-swizzled with the row (256-byte rows: row & 15; 128-byte rows:
// (row >> 1) & 7) so that the column-major writes and the row-major reads both cover all 64 banks once per 16 lanes.
template <int BM, int TM, int TN, int WM, int WN, bool SWZ = false, typename H = __bf16>      // H: the 16-bit type (deduced from `out`)
__device__ __forceinline__ void bgemm_epilogue_bf16(const IgemmArgs& p, f32x16 (&acc)[TM][TN], H* __restrict__ out, const int bm,
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
                            y[e] = (acc[mi][0][4 * g + e] * p.alpha + bv[e]) * gelu_gate(acc[mi][1][4 * g + e] * p.alpha + bg[e], p.a_bf16 == H16_BF16 || (kF16FastGate && p.a_bf16 == H16_FP16));
                        *reinterpret_cast<f32x4*>(st + mrow * SLD + 8 * g + nq) = y;
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = (lane >> 2) + 16 * i;
                        const int m = bm * BM + wm * WM + mi * 32 + row;
                        if (m < p.M) {
                            const f32x4 y0 = *reinterpret_cast<const f32x4*>(st + row * SLD + col);
                            const f32x4 y1 = *reinterpret_cast<const f32x4*>(st + row * SLD + col + 4);
                            hx8<H> o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { o[e] = (H)y0[e]; o[4 + e] = (H)y1[e]; }
                            *reinterpret_cast<hx8<H>*>(out + (size_t)m * p.ldc + nb / 2 + col) = o;
                        }
                    }
                }
            }
        }
        return;
    }
    const bool vec = ((p.N | p.ldc | p.ldr) & 7) == 0 && (p.rb_ld & 3) == 0;
    if (vec) {
        constexpr int SLD = SWZ ? WN : WN + 4;
        constexpr int LPR = WN / 8;                            // lanes per staged row (8 columns each)
        constexpr int RPI = 64 / LPR;                          // rows per access instruction
        float* st = stage + (size_t)(threadIdx.x >> 6) * 32 * SLD;
        const int col = (lane % LPR) * 8;
        auto swz = [](const int row) { return !SWZ ? 0 : (WN == 64 ? (row & 15) : ((row >> 1) & 7)); };
        const int wsw = swz(mrow);
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
                    *reinterpret_cast<f32x4*>(st + mrow * SLD + (((ni * 8 + 2 * g + (lane >> 5)) ^ wsw) << 2)) = y;
                }
#pragma unroll
            for (int i = 0; i < 32 / RPI; ++i) {
                const int row = lane / LPR + RPI * i;
                const int m = bm * BM + wm * WM + mi * 32 + row;
                if (m >= p.M || n >= p.N || (p.ablate & 1)) continue;   // N % 8 == 0: the lane's eight columns are in or out together
                const int rsw = swz(row);
                f32x4 y0 = *reinterpret_cast<const f32x4*>(st + row * SLD + ((((col >> 2)) ^ rsw) << 2));
                f32x4 y1 = *reinterpret_cast<const f32x4*>(st + row * SLD + ((((col >> 2) + 1) ^ rsw) << 2));
                if (p.alpha != 1.0f) { y0 *= p.alpha; y1 *= p.alpha; }
                y0 += b0; y1 += b1;
                if (p.rowbias) {
                    const float* rb = p.rowbias + (size_t)(m / p.rows_per_sample) * p.rb_ld + n;
                    y0 += *reinterpret_cast<const f32x4*>(rb);
                    y1 += *reinterpret_cast<const f32x4*>(rb + 4);
                }
                if (p.resid) {
                    if (p.resid_bf16) {
                        const hx8<H> r = *reinterpret_cast<const hx8<H>*>(reinterpret_cast<const H*>(p.resid) + (size_t)m * p.ldr + n);
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
                hx8<H> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (H)y0[e]; o[4 + e] = (H)y1[e]; }
                *reinterpret_cast<hx8<H>*>(out + (size_t)m * p.ldc + n) = o;
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
        H* orow = out + (size_t)m * p.ldc;
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
                    if (p.resid) v += p.resid_bf16 ? (float)reinterpret_cast<const H*>(p.resid)[(size_t)m * p.ldr + n + e]
                                                   : p.resid[(size_t)m * p.ldr + n + e];
                    if (p.relu) v = fmaxf(v, 0.f);
                    orow[n + e] = (H)v;
                }
            }
    }
}

// The same epilogue cut in two for the persistent kernel (bgemm.hip).  `prefetch` issues the only global loads the epilogue
// has left -- the bf16 residual -- BEFORE the caller starts the next tile's first stage, so that they are older than that DMA
// in the wave's in-order vmcnt queue and waiting for them never waits for it.  Bias and time-embedding rows do not come from
// global memory here: the caller has put them into LDS for the tile's 128 columns -- `brow[0][128]` = bias, `brow[1 + j][128]` =
// rowbias of sample s_lo + j, j = 0, 1 (a tile's rows belong to at most two samples; sample of row m = m / rows_per_sample).  `finish` stages the
// accumulators through `st` (this wave's 32 x WN floats, unpadded and chunk-swizzled as above), adds, rounds once and stores.
// Residuals are bf16 (the launcher sends an fp32 residual to the non-persistent kernels).
template <typename H, int BM, int TM, int TN, int WM, int WN, bool ROWBIAS = true>      // H: bf16 / fp16; ROWBIAS false: launches that never carry one (linears)
struct BgEpilogue {
    static constexpr int LPR = WN / 8, RPI = 64 / LPR, NI = 32 / RPI;
    hx8<H> res[TM][NI];
    bool vec;

    __device__ __forceinline__ void prefetch(const IgemmArgs& p, const int bm, const int n0, const int wm, const int wn, const int lane) {
        vec = ((p.N | p.ldc | p.ldr) & 7) == 0;
        if (p.geglu || !vec || !p.resid) return;
        // branch-free (a load under a branch cannot be counted by the compiler's vmcnt bookkeeping, which then drains the queue
        // -- the next tile's DMA included -- at the first use): buffer loads against a descriptor based at the tile's first
        // residual row, rows / columns outside the matrix as out-of-window offsets
        const int n = n0 + wn * WN + (lane % LPR) * 8;
        const unsigned long long base = reinterpret_cast<unsigned long long>(reinterpret_cast<const H*>(p.resid) + (size_t)bm * BM * p.ldr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), (short)0,
                                                                            0x7FFFFFF0, 0x00020000);
        // (the empty asm keeps the per-row offsets from being hoisted out of the caller's tile loop as eight loop-invariant
        // registers, which the allocator then spills: a scratch reload in the epilogue is a vmcnt wait behind the DMA)
        unsigned off0 = (unsigned)(((wm * WM + lane / LPR) * p.ldr + n) * 2);
        asm volatile("" : "+v"(off0));
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = wm * WM + mi * 32 + lane / LPR + RPI * i;
                const bool ok = bm * BM + row < p.M && n < p.N;
                const unsigned off = ok ? off0 + (unsigned)((mi * 32 + RPI * i) * p.ldr * 2) : 0x80000000u;
                const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                res[mi][i] = __builtin_bit_cast(hx8<H>, r);
            }
    }

    // WAIT false: no counted vmcnt at the end (the caller runs several finish() calls behind one DMA and waits once, for the sum
    // of their stores: STORES each, when vec / geglu hold)
    static constexpr int STORES = TM * NI, STORES_GEGLU = TM * 2;
    template <bool WAIT = true>
    __device__ __forceinline__ void finish(const IgemmArgs& p, f32x16 (&acc)[TM][TN], H* __restrict__ out, const int bm, const int n0,
                                           const int wm, const int wn, const int lane, float* st, const float* brow, const int s_lo) {
        const int mrow = lane & 31;
        const int nq = (lane >> 5) * 4;
        // stores: buffer stores against a descriptor based at the tile's first output row, masked elements as out-of-window
        // offsets (dropped by the buffer unit) -- issued unconditionally, so that the number of stores behind the next tile's
        // DMA is a compile-time constant and the wait in front of the barrier can be COUNTED (vmcnt(stores)): the DMA has
        // landed, the stores are left in flight
        const unsigned long long obase = reinterpret_cast<unsigned long long>(out + (size_t)bm * BM * p.ldc);
        const unsigned olo = __builtin_amdgcn_readfirstlane((unsigned)obase);
        const unsigned ohi = __builtin_amdgcn_readfirstlane((unsigned)(obase >> 32));
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)ohi << 32) | olo), (short)0,
                                                                             0x7FFFFFF0, 0x00020000);
        const bool drop = (p.ablate & 1) != 0;
        if (p.geglu) {
            if constexpr (TN == 2) {
                const int nb = n0 + wn * WN;
                if (nb + 64 > p.N) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    return;
                }
                const int col = (lane & 3) * 8;                    // 4 lanes x 8 bf16 = one 64-byte output row segment
                const int wsw = (mrow >> 1) & 7;                   // 32-float staging rows
                f32x4 gv[4], gg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    gv[g] = *reinterpret_cast<const f32x4*>(brow + wn * WN + 8 * g + nq);
                    gg[g] = *reinterpret_cast<const f32x4*>(brow + wn * WN + 32 + 8 * g + nq);
                }
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 y;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            y[e] = (acc[mi][0][4 * g + e] * p.alpha + gv[g][e]) * gelu_gate(acc[mi][1][4 * g + e] * p.alpha + gg[g][e], p.a_bf16 == H16_BF16 || (kF16FastGate && p.a_bf16 == H16_FP16));
                        *reinterpret_cast<f32x4*>(st + mrow * 32 + (((2 * g + (lane >> 5)) ^ wsw) << 2)) = y;
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = (lane >> 2) + 16 * i;
                        const int m = bm * BM + wm * WM + mi * 32 + row;
                        const int rsw = (row >> 1) & 7;
                        const f32x4 y0 = *reinterpret_cast<const f32x4*>(st + row * 32 + (((col >> 2) ^ rsw) << 2));
                        const f32x4 y1 = *reinterpret_cast<const f32x4*>(st + row * 32 + ((((col >> 2) + 1) ^ rsw) << 2));
                        hx8<H> o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o[e] = (H)y0[e]; o[4 + e] = (H)y1[e]; }
                        const unsigned off = (m < p.M && !drop) ? (unsigned)(((wm * WM + mi * 32 + row) * p.ldc + nb / 2 + col) * 2) : 0x80000000u;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ors, off, 0, 0);
                    }
                }
                if constexpr (WAIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TM * 2) : "memory");
                return;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        if (!vec) {                                            // N or a stride not a multiple of 8: element-wise, no staging
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int m = bm * BM + wm * WM + mi * 32 + mrow;
                if (m >= p.M) continue;
                const float* bb = brow + wn * WN;
                const float* br = brow + 128 + ((ROWBIAS && p.rowbias) ? (m / p.rows_per_sample - s_lo) * 128 : 0) + wn * WN;
                H* orow = out + (size_t)m * p.ldc;
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = ni * 32 + 8 * g + nq;
                        const int n = n0 + wn * WN + c;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (n + e >= p.N) break;
                            float v = acc[mi][ni][4 * g + e] * p.alpha + bb[c + e];
                            if (ROWBIAS && p.rowbias) v += br[c + e];
                            if (p.resid) v += (float)reinterpret_cast<const H*>(p.resid)[(size_t)m * p.ldr + n + e];
                            if (p.relu) v = fmaxf(v, 0.f);
                            orow[n + e] = (H)v;
                        }
                    }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        auto swz = [](const int row) { return WN == 64 ? (row & 15) : ((row >> 1) & 7); };
        const int wsw = swz(mrow);
        const int col = (lane % LPR) * 8;
        const int n = n0 + wn * WN + col;
        unsigned ooff0 = (unsigned)(((wm * WM + lane / LPR) * p.ldc + n) * 2);
        asm volatile("" : "+v"(ooff0));                    // as in prefetch(): recomputed per tile, not eight spilled invariants
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = acc[mi][ni][4 * g + e];
                    *reinterpret_cast<f32x4*>(st + mrow * WN + (((ni * 8 + 2 * g + (lane >> 5)) ^ wsw) << 2)) = y;
                }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = lane / LPR + RPI * i;
                const int m = bm * BM + wm * WM + mi * 32 + row;
                const int rsw = swz(row);
                f32x4 y0 = *reinterpret_cast<const f32x4*>(st + row * WN + (((col >> 2) ^ rsw) << 2));
                f32x4 y1 = *reinterpret_cast<const f32x4*>(st + row * WN + ((((col >> 2) + 1) ^ rsw) << 2));
                const float* bb = brow + wn * WN + col;
                if (p.alpha != 1.0f) { y0 *= p.alpha; y1 *= p.alpha; }
                y0 += *reinterpret_cast<const f32x4*>(bb);
                y1 += *reinterpret_cast<const f32x4*>(bb + 4);
                if (ROWBIAS && p.rowbias) {                        // separately, in the order of the tile kernels: bit-identical results
                    const float* br = bb + 128 + (m < p.M ? (m / p.rows_per_sample - s_lo) * 128 : 0);
                    y0 += *reinterpret_cast<const f32x4*>(br);
                    y1 += *reinterpret_cast<const f32x4*>(br + 4);
                }
                if (p.resid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { y0[e] += (float)res[mi][i][e]; y1[e] += (float)res[mi][i][4 + e]; }
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { y0[e] = fmaxf(y0[e], 0.f); y1[e] = fmaxf(y1[e], 0.f); }
                }
                hx8<H> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (H)y0[e]; o[4 + e] = (H)y1[e]; }
                const unsigned off = (m < p.M && n < p.N && !drop) ? ooff0 + (unsigned)((mi * 32 + RPI * i) * p.ldc * 2) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ors, off, 0, 0);
            }
        }
        if constexpr (WAIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TM * NI) : "memory");
    }
};

}  // namespace e2v
