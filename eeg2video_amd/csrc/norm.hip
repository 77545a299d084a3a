// GroupNorm(+SiLU) and LayerNorm over channel-last fp32 activations (HBM-bound kernels).
//
// GroupNorm (nn.GroupNorm call sites resnet.py:177,188; unet.py:406; attention.py:99): a group is
// (C/groups) channels x P rows of one slab -- up to 1.66 MB (C=960 @6x36x64), far beyond LDS, and the
// group width (10/20/30/40/60/80 channels) does not align to 16-byte lanes.  So statistics are taken
// PER CHANNEL first: (1) every block sums x and x^2 of one row chunk for all channels (coalesced
// float4 rows, fp32 partials over <=64 values per thread); (2) one wave per (slab, group) folds the
// partials of its channels in fp64 and emits per-(slab, channel) scale/shift; (3) a streaming pass
// applies y = act(x*scale + shift).  Two sources = the channel concat of the up blocks
// (unet_blocks.py:487), whose groups may straddle the seam (1920/32 = 60 does not divide 1280).
#include "h16.h"
#include "kernels.h"
#include "prof.h"
#include "act_io.h"
#ifdef E2V_AB
#include <hip/hip_cooperative_groups.h>      // gn_coop_kernel (measured, not adopted)
#endif
#include <string>
#include <type_traits>

namespace e2v {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int GN_ROWS_PER_CHUNK = 64;      // granularity of the partial-sum slab (callers size it with groupnorm_chunks)

int groupnorm_chunks(int P) { return (P + GN_ROWS_PER_CHUNK - 1) / GN_ROWS_PER_CHUNK; }

// largest divisor of cq that is <= 64
static int quad_tile(int cq) {
    int best = 1;
    for (int d = 1; d <= 64 && d <= cq; ++d)
        if (cq % d == 0) best = d;
    return best;
}

// grid (chunks, slabs); part[((slab*chunks + chunk)*Ctot + coff + c)*2 + {0,1}]
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int ld, int C, int P, int chunks,
                                                         float* __restrict__ part, int Ctot, int coff, int QT, int chunk_rows) {
    __shared__ f32x4 red[2][256];
    const int chunk = blockIdx.x, slab = blockIdx.y;
    const int R = 256 / QT;
    const int q = threadIdx.x % QT, r = threadIdx.x / QT;
    const int p0 = chunk * chunk_rows;
    const int p1 = min(P, p0 + chunk_rows);
    const T* base = x + (size_t)slab * P * ld;
    float* dst = part + ((size_t)(slab * chunks + chunk) * Ctot + coff) * 2;
    const int CQ = C / 4;
    for (int q0 = 0; q0 < CQ; q0 += QT) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
        if (r < R) {
            const T* col = base + (q0 + q) * 4;
            int pr = p0 + r;
            for (; pr + 3 * R < p1; pr += 4 * R) {          // four rows in flight per thread
                const f32x4 v0 = ld4(col + (size_t)pr * ld);
                const f32x4 v1 = ld4(col + (size_t)(pr + R) * ld);
                const f32x4 v2 = ld4(col + (size_t)(pr + 2 * R) * ld);
                const f32x4 v3 = ld4(col + (size_t)(pr + 3 * R) * ld);
                s += (v0 + v1) + (v2 + v3);
                ss += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
            }
            for (; pr < p1; pr += R) {
                const f32x4 v = ld4(col + (size_t)pr * ld);
                s += v;
                ss += v * v;
            }
        }
        red[0][threadIdx.x] = s;
        red[1][threadIdx.x] = ss;
        __syncthreads();
        if (threadIdx.x < QT) {
            f32x4 ts = {0.f, 0.f, 0.f, 0.f}, tss = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < R; ++k) {
                ts += red[0][k * QT + threadIdx.x];
                tss += red[1][k * QT + threadIdx.x];
            }
            float* d = dst + (size_t)(q0 + threadIdx.x) * 8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d[2 * e] = ts[e];
                d[2 * e + 1] = tss[e];
            }
        }
        __syncthreads();
    }
}

// the fold of one (slab, group) by one wave: fp64 over the chunks' per-channel partials, (scale, shift) per channel out
__device__ __forceinline__ void gn_finalize_unit(const float* __restrict__ part, int chunks, int Ctot, int groups, int P, float eps,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ scsh,
                                                 const int g, const int slab, const int lane) {
    const int cpg = Ctot / groups;
    double s = 0.0, ss = 0.0;
    const int total = chunks * cpg;
    // A lane adds its entries i = lane, lane + 64, ... IN THAT ORDER (the bits of every earlier build); eight loads are in flight at a
    // time -- the one-at-a-time loop was a chain of L2 latencies: 13 us per launch at B = 1 (216 chunks x 10 channels: 34 rounds), 2.6 %
    // of a step (profiles/r05_timeline_b1_bf16.json).
    typedef float f32x2g __attribute__((ext_vector_type(2)));
    for (int i0 = lane; i0 < total; i0 += 64 * 8) {
        f32x2g e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + 64 * u, total - 1);
            const int ch = i / cpg, c = g * cpg + (i - ch * cpg);
            e[u] = *reinterpret_cast<const f32x2g*>(part + ((size_t)(slab * chunks + ch) * Ctot + c) * 2);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + 64 * u < total) { s += (double)e[u][0]; ss += (double)e[u][1]; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        ss += __shfl_xor(ss, off);
    }
    const double cnt = (double)cpg * (double)P;
    const double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
        const double ga = (double)gamma[c];
        float* o = scsh + ((size_t)slab * Ctot + c) * 2;
        o[0] = (float)(rstd * ga);
        o[1] = (float)((double)beta[c] - mean * rstd * ga);
    }
}

// grid (groups, slabs), one wave each
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part, int chunks, int Ctot, int groups,
                                                         int P, float eps, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ scsh) {
    gn_finalize_unit(part, chunks, Ctot, groups, P, eps, gamma, beta, scsh, blockIdx.x, blockIdx.y, threadIdx.x);
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x0, const T* __restrict__ x1, int c0,
                                                       int c1, int ld0, int ld1, const float* __restrict__ scsh,
                                                       T* __restrict__ out, int ldo, int P, size_t rows, int act) {
    const int Ctot = c0 + c1;
    const int CQ = Ctot / 4;
    const size_t total = rows * CQ;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    auto one = [&](size_t i, f32x4& v, f32x4& a, f32x4& b, size_t& row, int& c) {
        row = i / CQ;
        c = (int)(i - row * CQ) * 4;
        const int slab = (int)(row / P);
        v = (c < c0) ? ld4(x0 + row * ld0 + c) : ld4(x1 + row * ld1 + (c - c0));
        const float* sc = scsh + ((size_t)slab * Ctot + c) * 2;
        a = *reinterpret_cast<const f32x4*>(sc);
        b = *reinterpret_cast<const f32x4*>(sc + 4);
    };
    auto fin = [&](const f32x4& v, const f32x4& a, const f32x4& b, size_t row, int c) {
        f32x4 y;
        y[0] = v[0] * a[0] + a[1];
        y[1] = v[1] * a[2] + a[3];
        y[2] = v[2] * b[0] + b[1];
        y[3] = v[3] * b[2] + b[3];
        if (act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = silu_f(y[e]);
        }
        st4(out + row * ldo + c, y);
    };
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + stride < total; i += 2 * stride) {           // two elements in flight per thread
        f32x4 v0, a0, b0, v1, a1, b1;
        size_t r0, r1;
        int cc0, cc1;
        one(i, v0, a0, b0, r0, cc0);
        one(i + stride, v1, a1, b1, r1, cc1);
        fin(v0, a0, b0, r0, cc0);
        fin(v1, a1, b1, r1, cc1);
    }
    if (i < total) {
        f32x4 v, a, b;
        size_t r;
        int cc;
        one(i, v, a, b, r, cc);
        fin(v, a, b, r, cc);
    }
}

// ---- bf16 activations: 16-byte (8-channel) accesses ---------------------------------------------------------------------
// The templated kernels above move 4 channels per lane: 16 bytes of fp32 but only 8 of bf16, and at half the bytes per
// instruction the bf16 passes ran at 2.2-2.9 TB/s effective.  These variants give every lane 8 channels (one 16-byte load /
// store), the arithmetic is unchanged (fp32 sums of <= 64 values per thread, fp64 fold, fp32 affine).
struct F8 { f32x4 lo, hi; };
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
template <typename H>      // H: bf16 / fp16 (h16.h)
__device__ __forceinline__ F8 ld8(const H* p) {
    const hx8<H> v = *reinterpret_cast<const hx8<H>*>(p);
    F8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r.lo[e] = (float)v[e]; r.hi[e] = (float)v[4 + e]; }
    return r;
}
template <typename H>
__device__ __forceinline__ void st8(H* p, const f32x4& lo, const f32x4& hi) {
    hx8<H> v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (H)lo[e]; v[4 + e] = (H)hi[e]; }
    *reinterpret_cast<hx8<H>*>(p) = v;
}

// grid (chunks, slabs); same partial layout as gn_partial_kernel; OT = octets handled side by side (divides C / 8, <= 64)
template <typename H>
__global__ __launch_bounds__(256) void gn_partial8_kernel(const H* __restrict__ x, int ld, int C, int P, int chunks,
                                                          float* __restrict__ part, int Ctot, int coff, int OT, int chunk_rows) {
    __shared__ f32x4 red[4][256];
    const int chunk = blockIdx.x, slab = blockIdx.y;
    const int R = 256 / OT;
    const int q = threadIdx.x % OT, r = threadIdx.x / OT;
    const int p0 = chunk * chunk_rows;
    const int p1 = min(P, p0 + chunk_rows);
    const H* base = x + (size_t)slab * P * ld;
    float* dst = part + ((size_t)(slab * chunks + chunk) * Ctot + coff) * 2;
    const int CO = C / 8;
    for (int q0 = 0; q0 < CO; q0 += OT) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, ss0 = s0, ss1 = s0;
        if (r < R) {
            const H* col = base + (q0 + q) * 8;
            int pr = p0 + r;
            for (; pr + 3 * R < p1; pr += 4 * R) {          // four rows in flight per thread
                const F8 a = ld8(col + (size_t)pr * ld), b = ld8(col + (size_t)(pr + R) * ld);
                const F8 c = ld8(col + (size_t)(pr + 2 * R) * ld), d = ld8(col + (size_t)(pr + 3 * R) * ld);
                s0 += (a.lo + b.lo) + (c.lo + d.lo);
                s1 += (a.hi + b.hi) + (c.hi + d.hi);
                ss0 += (a.lo * a.lo + b.lo * b.lo) + (c.lo * c.lo + d.lo * d.lo);
                ss1 += (a.hi * a.hi + b.hi * b.hi) + (c.hi * c.hi + d.hi * d.hi);
            }
            for (; pr < p1; pr += R) {
                const F8 a = ld8(col + (size_t)pr * ld);
                s0 += a.lo; s1 += a.hi;
                ss0 += a.lo * a.lo; ss1 += a.hi * a.hi;
            }
        }
        red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = ss0; red[3][threadIdx.x] = ss1;
        __syncthreads();
        if (threadIdx.x < 2 * OT) {                       // thread (half, octet): channels 4 half .. + 3 of the octet
            const int o = threadIdx.x % OT, half = threadIdx.x / OT;
            f32x4 ts = {0.f, 0.f, 0.f, 0.f}, tss = ts;
            for (int k = 0; k < R; ++k) {
                ts += red[half][k * OT + o];
                tss += red[2 + half][k * OT + o];
            }
            float* d = dst + ((size_t)(q0 + o) * 8 + 4 * half) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d[2 * e] = ts[e];
                d[2 * e + 1] = tss[e];
            }
        }
        __syncthreads();
    }
}

#ifdef E2V_AB          // the flat-index apply pass (E2V_GN_ROWS = 0): the other arm of the A/B that adopted the row-tiled one
template <typename H>
__global__ __launch_bounds__(256) void gn_apply8_kernel(const H* __restrict__ x0, const H* __restrict__ x1, int c0, int c1,
                                                        int ld0, int ld1, const float* __restrict__ scsh, H* __restrict__ out,
                                                        int ldo, int P, size_t rows, int act) {
    const int Ctot = c0 + c1;
    const int CO = Ctot / 8;
    const size_t total = rows * CO;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t row = i / CO;
        const int c = (int)(i - row * CO) * 8;
        const int slab = (int)(row / P);
        const F8 v = (c < c0) ? ld8(x0 + row * ld0 + c) : ld8(x1 + row * ld1 + (c - c0));
        const float* sc = scsh + ((size_t)slab * Ctot + c) * 2;
        const f32x4 a = *reinterpret_cast<const f32x4*>(sc), b = *reinterpret_cast<const f32x4*>(sc + 4);
        const f32x4 d = *reinterpret_cast<const f32x4*>(sc + 8), e = *reinterpret_cast<const f32x4*>(sc + 12);
        f32x4 lo, hi;
        lo[0] = v.lo[0] * a[0] + a[1]; lo[1] = v.lo[1] * a[2] + a[3]; lo[2] = v.lo[2] * b[0] + b[1]; lo[3] = v.lo[3] * b[2] + b[3];
        hi[0] = v.hi[0] * d[0] + d[1]; hi[1] = v.hi[1] * d[2] + d[3]; hi[2] = v.hi[2] * e[0] + e[1]; hi[3] = v.hi[3] * e[2] + e[3];
        if (act) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { lo[k] = silu_f(lo[k]); hi[k] = silu_f(hi[k]); }
        }
        st8(out + row * ldo + c, lo, hi);
    }
}
#endif

// Row-tiled form of the pass above: grid (chunks, slabs) like the statistics pass, thread -> (row r of R at a time, octet q of OT side by
// side), the scale / shift pairs of the thread's eight channels held in registers over the rows of the chunk, four rows in flight
// per thread.  No index division (the flat form above spent more vector instructions on a 64-bit i / CO than on the affine),
// and SiLU as x * rcp(1 + exp2(-x log2 e)): two transcendentals and three plain instructions per value.
__device__ __forceinline__ float silu_fast(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}
template <typename H, bool ACT>
__global__ __launch_bounds__(256) void gn_apply8_rows_kernel(const H* __restrict__ x0, const H* __restrict__ x1, int c0, int c1,
                                                             int ld0, int ld1, const float* __restrict__ scsh, H* __restrict__ out,
                                                             int ldo, int P, int OT, int chunk_rows) {
    const int chunk = blockIdx.x, slab = blockIdx.y;
    const int R = 256 / OT;
    const int q = threadIdx.x % OT, r = threadIdx.x / OT;
    if (r >= R) return;
    const int p0 = chunk * chunk_rows;
    const int p1 = min(P, p0 + chunk_rows);
    const int Ctot = c0 + c1;
    const size_t row0 = (size_t)slab * P;
    const float* scb = scsh + (size_t)slab * Ctot * 2;
    for (int q0 = 0; q0 < Ctot / 8; q0 += OT) {             // (OT divides c0 / 8 and c1 / 8: a column tile has one source)
        const int c = (q0 + q) * 8;
        const H* col = c < c0 ? x0 + row0 * ld0 + c : x1 + row0 * ld1 + (c - c0);
        const int ld = c < c0 ? ld0 : ld1;
        H* dst = out + row0 * ldo + c;
        const float* sc = scb + (size_t)c * 2;
        const f32x4 a = *reinterpret_cast<const f32x4*>(sc), b = *reinterpret_cast<const f32x4*>(sc + 4);
        const f32x4 d = *reinterpret_cast<const f32x4*>(sc + 8), e = *reinterpret_cast<const f32x4*>(sc + 12);
        auto fin = [&](const F8& v, const int pr) {
            f32x4 lo, hi;
            lo[0] = v.lo[0] * a[0] + a[1]; lo[1] = v.lo[1] * a[2] + a[3]; lo[2] = v.lo[2] * b[0] + b[1]; lo[3] = v.lo[3] * b[2] + b[3];
            hi[0] = v.hi[0] * d[0] + d[1]; hi[1] = v.hi[1] * d[2] + d[3]; hi[2] = v.hi[2] * e[0] + e[1]; hi[3] = v.hi[3] * e[2] + e[3];
            if constexpr (ACT) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { lo[k] = silu_fast(lo[k]); hi[k] = silu_fast(hi[k]); }
            }
            st8(dst + (size_t)pr * ldo, lo, hi);
        };
        int pr = p0 + r;
        for (; pr + 3 * R < p1; pr += 4 * R) {
            const F8 v0 = ld8(col + (size_t)pr * ld), v1 = ld8(col + (size_t)(pr + R) * ld);
            const F8 v2 = ld8(col + (size_t)(pr + 2 * R) * ld), v3 = ld8(col + (size_t)(pr + 3 * R) * ld);
            fin(v0, pr); fin(v1, pr + R); fin(v2, pr + 2 * R); fin(v3, pr + 3 * R);
        }
        for (; pr < p1; pr += R) fin(ld8(col + (size_t)pr * ld), pr);
    }
}

static int oct_tile(int co) {
    int best = 1;
    for (int d = 1; d <= 64 && d <= co; ++d)
        if (co % d == 0) best = d;
    return best;
}

// Rows per workgroup of the statistics / apply passes, a multiple of GN_ROWS_PER_CHUNK chosen from the SAMPLE's size only (never
// from the batch: the grouping of the fp32 partial sums must not depend on how many clips run together).  Level 0 (13 824 rows per
// sample) streams 256-row chunks; the deeper levels (3456 / 864 / 240 rows) get 64-row chunks -- at 256 they were 14 / 4 / 1
// workgroups per sample and the chip ran a quarter full (tools/norm_micro.py: 1.7 TB/s at level 2).
// (small_family: the small-batch dispatch family -- two samples at level 0 are 108 chunks of 256 rows for 256 CUs: 64-row chunks there too;
// a different grouping of the partial sums, so the FAMILY picks it, never the batch within a family)
static int gn_chunk_rows(const int P, const bool small_family = false) {
    static const int* const big = E2V_AB_KNOB("E2V_GN_CHUNK_ROWS", 256);            // (tuned in round 3, tools/norm_micro.py: `make ab` only)
    static const int* const small = E2V_AB_KNOB("E2V_GN_CHUNK_ROWS_SMALL", 64);
    const int want = (P >= 8192 && !small_family) ? *big : *small;
    const int v = want / GN_ROWS_PER_CHUNK * GN_ROWS_PER_CHUNK;
    return v < GN_ROWS_PER_CHUNK ? GN_ROWS_PER_CHUNK : v;
}

// ---- GroupNorm statistics that came with the tensor (IgemmArgs::rbsum) ------------------------------------------------------------
// Canonical row-block sums of a stored bf16 tensor: out[b][c][2] = (sum, sum of squares) of rows 64 b .. 64 b + 63, column c, added in
// the order the staged epilogue of bgemm_t256_kernel adds them (bgemm256.hip rb_fold): RPP "lanes" j = 0 .. RPP - 1 each take rows
// j, j + RPP, ... < 32 of the block's first 32-row chunk into one fp32 sum (plain add; squares by fma), likewise of its second chunk,
// add the two; the lane sums are then added in lane order.  A tensor that the epilogue did not write (small batch: another kernel served the layer) gets
// its sums from here -- the same bits, so that the batch size never shows in a result.  grid = row blocks; a thread = (8-column
// piece, lane j).
__global__ __launch_bounds__(256) void rowblock_sums_kernel(const __bf16* __restrict__ x, int ld, int C, int rpp, float* __restrict__ out) {
    extern __shared__ float rbs_lds[];                    // [pairs][16]
    const size_t row0 = (size_t)blockIdx.x * 64;
    const int pieces = C / 8, pairs = pieces * rpp;
    for (int i = threadIdx.x; i < pairs; i += 256) {
        const int q = i / rpp, j = i - q * rpp;
        f32x4 acc[2][4];
        for (int c = 0; c < 2; ++c) {
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, q0 = s0, q1 = s0;
            for (int r = j; r < 32; r += rpp) {
                const F8 v = ld8(x + (row0 + 32 * c + r) * ld + q * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0[e] += v.lo[e]; s1[e] += v.hi[e];
                    q0[e] = __builtin_fmaf(v.lo[e], v.lo[e], q0[e]); q1[e] = __builtin_fmaf(v.hi[e], v.hi[e], q1[e]);
                }
            }
            acc[c][0] = s0; acc[c][1] = s1; acc[c][2] = q0; acc[c][3] = q1;
        }
        const f32x4 s0 = acc[0][0] + acc[1][0], s1 = acc[0][1] + acc[1][1], q0 = acc[0][2] + acc[1][2], q1 = acc[0][3] + acc[1][3];
        float* l = rbs_lds + (size_t)i * 16;
        *reinterpret_cast<f32x4*>(l) = s0; *reinterpret_cast<f32x4*>(l + 4) = s1;
        *reinterpret_cast<f32x4*>(l + 8) = q0; *reinterpret_cast<f32x4*>(l + 12) = q1;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < pieces; q += 256) {
        const float* l = rbs_lds + (size_t)q * rpp * 16;
        f32x4 t0 = *reinterpret_cast<const f32x4*>(l), t1 = *reinterpret_cast<const f32x4*>(l + 4);
        f32x4 u0 = *reinterpret_cast<const f32x4*>(l + 8), u1 = *reinterpret_cast<const f32x4*>(l + 12);
        for (int j = 1; j < rpp; ++j) {
            const float* o = l + j * 16;
            t0 += *reinterpret_cast<const f32x4*>(o); t1 += *reinterpret_cast<const f32x4*>(o + 4);
            u0 += *reinterpret_cast<const f32x4*>(o + 8); u1 += *reinterpret_cast<const f32x4*>(o + 12);
        }
        float* d = out + ((size_t)blockIdx.x * C + q * 8) * 2;
        *reinterpret_cast<f32x4*>(d) = f32x4{t0[0], u0[0], t0[1], u0[1]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{t0[2], u0[2], t0[3], u0[3]};
        *reinterpret_cast<f32x4*>(d + 8) = f32x4{t1[0], u1[0], t1[1], u1[1]};
        *reinterpret_cast<f32x4*>(d + 12) = f32x4{t1[2], u1[2], t1[3], u1[3]};
    }
}
void rowblock_sums(const void* x, int ld, int C, long long rows, int rpp, float* out, hipStream_t s) {
    if (rows <= 0) return;
    const size_t smem = (size_t)(C / 8) * rpp * 16 * sizeof(float);
    if (smem > 48 * 1024) E2V_KATTR(rowblock_sums_kernel, smem);
    ProfScope ps("rowblock_sums", 3.0 * rows * C, 2.0 * rows * C, s);
    E2V_KLAUNCH(rowblock_sums_kernel, dim3((unsigned)(rows / 64)), dim3(256), smem, s, static_cast<const __bf16*>(x), ld, C, rpp, out);
}

// The fold of gn_finalize_kernel over sources whose partial sums live in different places: source i contributes channels
// [off_i, off_i + c_i) of the concatenation from part_i[(slab * chunks_i + chunk) * ld_i + (channel - off_i) + coff_i][2] -- the statistics
// workspace of this call (ld = c0 + c1, coff = off) or row-block sums that came with the tensor (ld = c_i, coff = 0, chunks = P / 64).
__global__ __launch_bounds__(64) void gn_finalize_mixed_kernel(const float* __restrict__ part0, int chunks0, int ld0, int coff0, int c0,
                                                               const float* __restrict__ part1, int chunks1, int ld1, int coff1, int Ctot,
                                                               int groups, int P, float eps, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ scsh) {
    const int g = blockIdx.x, slab = blockIdx.y, lane = threadIdx.x;
    const int cpg = Ctot / groups;
    double s = 0.0, ss = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {           // a group's channels may straddle the seam of the two sources
        const bool first = c < c0;
        const float* part = first ? part0 : part1;
        const int chunks = first ? chunks0 : chunks1, ld = first ? ld0 : ld1;
        const int col = first ? c + coff0 : c - c0 + coff1;
        for (int ch = lane; ch < chunks; ch += 64) {
            const float* e = part + ((size_t)(slab * chunks + ch) * ld + col) * 2;
            s += (double)e[0];
            ss += (double)e[1];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        ss += __shfl_xor(ss, off);
    }
    const double cnt = (double)cpg * (double)P;
    const double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
        const double ga = (double)gamma[c];
        float* o = scsh + ((size_t)slab * Ctot + c) * 2;
        o[0] = (float)(rstd * ga);
        o[1] = (float)((double)beta[c] - mean * rstd * ga);
    }
}

// ---- one kernel for a small (sample, group): the small-batch family ---------------------------------------------------------------
// Two UNet samples (the reference's clip-by-clip loop) turn the three-launch GroupNorm above -- statistics per row chunk, fold, apply --
// into three dependent launches of a few dozen workgroups each: 38-46 us per call at the deep levels for tensors of 1-5 MB, 13 % of a
// B = 1 pass.  Here ONE workgroup owns a (sample, group): its P x cpg slice (both sources of a concat; <= 144 KB as 16-bit) is read
// once into LDS with the sums taken on the way, the fold is a workgroup reduction in fp64, and the affine (+ SiLU) is applied from LDS.
// Pieces are channel PAIRS (4 bytes): cpg is even for every width of the model (10, 20, 30, 40, 60, 80) while 8-channel pieces would
// straddle group boundaries.  Another summation order than the chunked path, so the choice is the dispatch family's (GroupNormArgs::
// fused_small), never the batch's within a family.
template <typename H, bool ACT>
__global__ __launch_bounds__(1024) void gn_fused_small_kernel(const H* __restrict__ x0, const H* __restrict__ x1, int c0, int c1, int ld0,
                                                             int ld1, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             H* __restrict__ out, int ldo, int P, int groups, float eps) {
    extern __shared__ __attribute__((aligned(16))) char gnf_lds[];
    const int g = blockIdx.x, slab = blockIdx.y, tid = threadIdx.x;
    const int Ctot = c0 + c1, cpg = Ctot / groups, ppr = cpg / 2;      // channel pairs per row of the slice
    unsigned* const tile = reinterpret_cast<unsigned*>(gnf_lds);       // [P][ppr] packed pairs
    float* const gb = reinterpret_cast<float*>(tile + (size_t)P * ppr);   // [2][cpg]: gamma, beta -> scale, shift
    double* const red = reinterpret_cast<double*>(gb + 2 * cpg + ((P * ppr) & 1));      // [2][16] wave sums, 8-byte aligned (2 cpg is even: pad when P x ppr is odd)
    const int cbase = g * cpg;
    const size_t row0 = (size_t)slab * P;
    const int items = P * ppr;
    float s = 0.f, ss = 0.f;
    for (int i = tid; i < items; i += 1024) {
        const int r = i / ppr, c = cbase + 2 * (i - r * ppr);
        const H* src = c < c0 ? x0 + (row0 + r) * ld0 + c : x1 + (row0 + r) * ld1 + (c - c0);
        const unsigned u = *reinterpret_cast<const unsigned*>(src);
        tile[i] = u;
        const float a = h16_unpack_lo<H>(u), b = h16_unpack_hi<H>(u);
        s += a + b;
        ss = fmaf(a, a, fmaf(b, b, ss));
    }
    for (int c = tid; c < cpg; c += 1024) { gb[c] = gamma[cbase + c]; gb[cpg + c] = beta[cbase + c]; }
    double ds = (double)s, dss = (double)ss;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ds += __shfl_xor(ds, off); dss += __shfl_xor(dss, off); }
    if ((tid & 63) == 0) { red[tid >> 6] = ds; red[16 + (tid >> 6)] = dss; }
    __syncthreads();
    double ts = 0.0, tss = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { ts += red[w]; tss += red[16 + w]; }      // every thread: the same order
    const double cnt = (double)cpg * (double)P;
    const double mean = ts / cnt;
    double var = tss / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const float fr = (float)rstd, fm = (float)(mean * rstd);
    for (int i = tid; i < items; i += 1024) {
        const int r = i / ppr, cl = 2 * (i - r * ppr);
        const unsigned u = tile[i];
        // scale / shift as the chunked path forms them: (rstd * gamma) x + (beta - mean * rstd * gamma), both rounded to fp32 once
        float a = fmaf(h16_unpack_lo<H>(u), fr * gb[cl], gb[cpg + cl] - fm * gb[cl]);
        float b = fmaf(h16_unpack_hi<H>(u), fr * gb[cl + 1], gb[cpg + cl + 1] - fm * gb[cl + 1]);
        if (ACT) { a = silu_f(a); b = silu_f(b); }
        hx2<H> o;
        o[0] = (H)a; o[1] = (H)b;
        *reinterpret_cast<hx2<H>*>(out + (row0 + r) * ldo + cbase + cl) = o;
    }
}

// Does the fused kernel serve this call?  16-bit rows, even group width and strides, the (sample, group) slice within 144 KB of LDS.
static size_t gn_fused_small_bytes(const GroupNormArgs& a) {
    const int Ctot = a.c0 + a.c1, cpg = Ctot / a.groups;
    return (size_t)a.P * cpg * 2 + (size_t)(2 * cpg + 1) * 4 + 2 * 16 * 8 + 8;
}
static bool gn_fused_small_applies(const GroupNormArgs& a) {
    static const int* const on = knob("E2V_GN_FUSED_SMALL", 1);
    if (!*on || !a.fused_small || !a.bf16 || !a.out || a.rb0 || a.rb1) return false;
    const int Ctot = a.c0 + a.c1, cpg = Ctot / a.groups;
    if ((cpg & 1) || (a.c0 & 1) || ((a.ld0 | a.ld1 | a.ldo) & 1)) return false;
    // a slice row is cpg x 2 contiguous bytes of a Ctot x 2-byte tensor row: below 64 bytes the reads waste most of every sector they
    // touch and the chunked path (whole rows, coalesced) wins unless the slice is tiny (measured at B = 1, profiles/r05_shape_ab_b1_*:
    // cpg 10 / 20 at 46 / 138 KB: 1.9x / 1.4x slower; cpg 40 / 80 at 19 .. 138 KB: 0.23 .. 0.58x)
    if (cpg < 32 && (size_t)a.P * cpg * 2 > (size_t)24 * 1024) return false;
    return gn_fused_small_bytes(a) <= (size_t)144 * 1024;
}

#ifdef E2V_AB
// ---- ONE launch for a tensor that fits the chip's LDS: the small-batch family at levels 0 / 1 ------------------------------------------
// Two samples at level 0 are 17.7 MB (35 MB with a concatenated skip): 69-138 KB per CU.  The three-launch path reads that tensor twice
// and spends most of its 37 us in the ramp and drain of three dependent launches of ~100 workgroups.  Here the tensor is read ONCE into
// the LDS of <= 256 co-resident workgroups (cooperative launch: the runtime refuses a grid that is not resident as a whole), each owning
// a run of rows of one sample: statistics per channel from LDS (partials in the layout of gn_partial8_kernel), grid barrier, the fold of
// every (sample, group) by one wave (gn_finalize_unit: the same fp64 fold), grid barrier, the affine (+ SiLU) applied from LDS.  Another
// grouping of the partial sums than the chunked path (rows per workgroup instead of 64-row chunks), so the small-batch family picks it.
// BUILT, PARITY-TESTED, MEASURED AND NOT ADOPTED (`make ab`, E2V_GN_COOP = 1; profiles/r05_shape_ab_b1_gn_cooperative.log): the
// cooperative launch itself costs more than the two launches it saves -- S2 P13824 C320 32 -> 92 us, S12 P2304 C320 20 -> 86 us, a B = 1
// step 27.1 -> 29.2 ms (1.44 -> 1.30 clips/s).
template <typename H, bool ACT>
__global__ __launch_bounds__(1024) void gn_coop_kernel(const H* __restrict__ x0, const H* __restrict__ x1, int c0, int c1, int ld0, int ld1,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, H* __restrict__ out, int ldo,
                                                      int P, int groups, float eps, int wps, int rows_per_wg, int RS, float* __restrict__ part,
                                                      float* __restrict__ scsh) {
    extern __shared__ __attribute__((aligned(16))) char gnc_lds[];
    namespace cg = cooperative_groups;
    cg::grid_group grid = cg::this_grid();
    const int tid = threadIdx.x;
    const int Ctot = c0 + c1, CO = Ctot / 8;
    const int slab = blockIdx.x / wps, chunk = blockIdx.x - slab * wps;
    const int r0 = chunk * rows_per_wg, r1 = min(P, r0 + rows_per_wg), nrows = max(r1 - r0, 0);
    u32x4g* const tile = reinterpret_cast<u32x4g*>(gnc_lds);                                   // [nrows][CO] 16-byte pieces
    float* const scratch = reinterpret_cast<float*>(gnc_lds + (size_t)rows_per_wg * CO * 16);   // [RS][CO][16] floats, then the sample's (scale, shift)
    const size_t row0 = (size_t)slab * P + r0;
    const int pieces = nrows * CO;
    for (int i = tid; i < pieces; i += 1024) {                // coalesced: consecutive lanes = consecutive 16-byte pieces of a row
        const int r = i / CO, q = i - r * CO, c = q * 8;
        const H* src = c < c0 ? x0 + (row0 + r) * ld0 + c : x1 + (row0 + r) * ld1 + (c - c0);
        tile[i] = *reinterpret_cast<const u32x4g*>(src);
    }
    __syncthreads();
    // statistics from LDS: thread (row subset rs, octet q) sums rows rs, rs + RS, ... of its 8 channels
    if (tid < RS * CO) {
        const int rs = tid / CO, q = tid - rs * CO;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, ss0 = s0, ss1 = s0;
        for (int r = rs; r < nrows; r += RS) {
            const hx8<H> v = __builtin_bit_cast(hx8<H>, tile[r * CO + q]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = (float)v[e], b = (float)v[4 + e];
                s0[e] += a; s1[e] += b;
                ss0[e] = fmaf(a, a, ss0[e]); ss1[e] = fmaf(b, b, ss1[e]);
            }
        }
        f32x4* d = reinterpret_cast<f32x4*>(scratch + ((size_t)rs * CO + q) * 16);
        d[0] = s0; d[1] = s1; d[2] = ss0; d[3] = ss1;
    }
    __syncthreads();
    if (tid < CO) {                                           // row subsets in order -> this workgroup's partial of the octet's 8 channels
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, ss0 = s0, ss1 = s0;
        for (int rs = 0; rs < RS; ++rs) {
            const f32x4* d = reinterpret_cast<const f32x4*>(scratch + ((size_t)rs * CO + tid) * 16);
            s0 += d[0]; s1 += d[1]; ss0 += d[2]; ss1 += d[3];
        }
        float* dst = part + ((size_t)(slab * wps + chunk) * Ctot + tid * 8) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dst[2 * e] = s0[e]; dst[2 * e + 1] = ss0[e];
            dst[8 + 2 * e] = s1[e]; dst[8 + 2 * e + 1] = ss1[e];
        }
    }
    grid.sync();
    {   // (sample, group) units dealt to the workgroups' first waves
        const int units = (int)(gridDim.x / wps) * groups;
        if (tid < 64)
            for (int u = blockIdx.x; u < units; u += gridDim.x)
                gn_finalize_unit(part, wps, Ctot, groups, P, eps, gamma, beta, scsh, u % groups, u / groups, tid);
    }
    grid.sync();
    for (int i = tid; i < 2 * Ctot; i += 1024) scratch[i] = scsh[(size_t)slab * Ctot * 2 + i];       // the sample's (scale, shift) pairs
    __syncthreads();
    for (int i = tid; i < pieces; i += 1024) {
        const int r = i / CO, q = i - r * CO, c = q * 8;
        const hx8<H> v = __builtin_bit_cast(hx8<H>, tile[i]);
        const f32x4* sc = reinterpret_cast<const f32x4*>(scratch + (size_t)c * 2);
        const f32x4 a = sc[0], b = sc[1], d = sc[2], e = sc[3];
        f32x4 lo, hi;
        lo[0] = (float)v[0] * a[0] + a[1]; lo[1] = (float)v[1] * a[2] + a[3]; lo[2] = (float)v[2] * b[0] + b[1]; lo[3] = (float)v[3] * b[2] + b[3];
        hi[0] = (float)v[4] * d[0] + d[1]; hi[1] = (float)v[5] * d[2] + d[3]; hi[2] = (float)v[6] * e[0] + e[1]; hi[3] = (float)v[7] * e[2] + e[3];
        if (ACT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { lo[k] = silu_f(lo[k]); hi[k] = silu_f(hi[k]); }
        }
        st8(out + (row0 + r) * ldo + c, lo, hi);
    }
}

// The plan of a cooperative launch: workgroups per sample, rows per workgroup, row subsets of the statistics phase, LDS bytes; ok = false
// when the call is not eligible (the three-launch path serves it)
struct GnCoopPlan { bool ok; int wps, rows, RS; size_t smem; };
static GnCoopPlan gn_coop_plan(const GroupNormArgs& a) {
    GnCoopPlan p{false, 0, 0, 0, 0};
    static const int* const on = knob("E2V_GN_COOP", 0);
    const int Ctot = a.c0 + a.c1;
    if (!*on || !a.small_chunks || !a.bf16 || !a.out || a.rb0 || a.rb1 || a.samples < 1 || a.samples > 256) return p;
    if ((a.c0 & 7) || (a.c1 & 7) || ((a.ld0 | a.ld1 | a.ldo) & 7) || Ctot > 2560 || Ctot % a.groups) return p;
    const int CO = Ctot / 8;
    p.wps = 256 / a.samples;
    if (p.wps > groupnorm_chunks(a.P)) p.wps = groupnorm_chunks(a.P);      // (the partial-sum workspace holds one entry per 64-row chunk)
    if (p.wps < 1) return p;
    p.rows = (a.P + p.wps - 1) / p.wps;
    p.wps = (a.P + p.rows - 1) / p.rows;                      // no empty workgroups
    p.RS = (int)((size_t)20 * 1024 / ((size_t)CO * 64));
    p.RS = p.RS < 1 ? 1 : (p.RS > 8 ? 8 : p.RS);
    if (p.RS * CO > 1024) return p;
    const size_t scratch = (size_t)p.RS * CO * 64 > (size_t)Ctot * 8 ? (size_t)p.RS * CO * 64 : (size_t)Ctot * 8;
    p.smem = (size_t)p.rows * CO * 16 + scratch;
    p.ok = p.smem <= (size_t)158 * 1024;
    return p;
}
#endif

static std::string gn_shape_tag(const GroupNormArgs& a) {
    return " S" + std::to_string(a.samples) + " P" + std::to_string(a.P) + " C" + std::to_string(a.c0) + (a.c1 ? "+" + std::to_string(a.c1) : "") + " g" +
           std::to_string(a.groups);
}

static void groupnorm_stats_launch(const GroupNormArgs& a, hipStream_t s) {
    const int Ctot = a.c0 + a.c1;
    const int crows = gn_chunk_rows(a.P, a.small_chunks != 0);
    const int chunks = (a.P + crows - 1) / crows;
    auto part = [&](const float* x, int ld, int C, int coff) {
        const int qt = quad_tile(C / 4);
        if (a.bf16) {
            h16_dispatch(a.bf16, [&](auto h16_tag) {
                using H = decltype(h16_tag);
                if (C % 8 == 0 && ld % 8 == 0)
                    E2V_KLAUNCH(gn_partial8_kernel<H>, dim3(chunks, a.samples), dim3(256), 0, s, reinterpret_cast<const H*>(x), ld, C, a.P,
                                chunks, a.ws_part, Ctot, coff, oct_tile(C / 8), crows);
                else
                    E2V_KLAUNCH(gn_partial_kernel<H>, dim3(chunks, a.samples), dim3(256), 0, s, reinterpret_cast<const H*>(x), ld, C,
                                a.P, chunks, a.ws_part, Ctot, coff, qt, crows);
            });
        }
        else
            E2V_KLAUNCH(gn_partial_kernel<float>, dim3(chunks, a.samples), dim3(256), 0, s, x, ld, C, a.P, chunks, a.ws_part, Ctot,
                               coff, qt, crows);
    };
    // sources that came with their row-block sums skip the statistics pass
    const bool use_rb = a.bf16 && a.P % 64 == 0 && (a.rb0 || (a.c1 > 0 && a.rb1));
    const bool rb0 = use_rb && a.rb0, rb1 = use_rb && a.c1 > 0 && a.rb1;
    if (dry_run()) {
        const std::string pk = std::string(a.bf16 && a.c0 % 8 == 0 && a.c1 % 8 == 0 && ((a.ld0 | a.ld1) & 7) == 0 ? "gn_partial8_kernel" : "gn_partial_kernel") + " rows" + std::to_string(crows);
        std::string t = " ->";
        if (!rb0) t += " " + pk + (a.c1 > 0 ? "[0]" : "");
        if (rb0) t += " sums-from-producer[0]";
        if (a.c1 > 0) t += rb1 ? " sums-from-producer[1]" : " " + pk + "[1]";
        dry_tag(t + (use_rb ? " + gn_finalize_mixed_kernel" : " + gn_finalize_kernel"));
    }
    // timing experiment (make ab): E2V_GN_SKIP_PARTIAL = 1 drops the statistics pass over the tensor -- RESULTS ARE WRONG -- to measure
    // the ceiling of what statistics taken in the producers' epilogues could save (profiles/r04_gn_stats_ceiling.log)
    static const int* const skip_partial = E2V_AB_KNOB("E2V_GN_SKIP_PARTIAL", 0);
    if (!*skip_partial) {
        if (!rb0) part(a.x0, a.ld0, a.c0, 0);
        if (a.c1 > 0 && !rb1) part(a.x1, a.ld1, a.c1, a.c0);
    }
    if (use_rb) {
        const int rbc = a.P / 64;
        E2V_KLAUNCH(gn_finalize_mixed_kernel, dim3(a.groups, a.samples), dim3(64), 0, s,
                    rb0 ? a.rb0 : a.ws_part, rb0 ? rbc : chunks, rb0 ? a.c0 : Ctot, 0, a.c0,
                    rb1 ? a.rb1 : a.ws_part, rb1 ? rbc : chunks, rb1 ? a.c1 : Ctot, rb1 ? 0 : a.c0, Ctot,
                    a.groups, a.P, a.eps, a.gamma, a.beta, a.ws_scale);
        return;
    }
    E2V_KLAUNCH(gn_finalize_kernel, dim3(a.groups, a.samples), dim3(64), 0, s, a.ws_part, chunks, Ctot, a.groups,
                       a.P, a.eps, a.gamma, a.beta, a.ws_scale);
}

void groupnorm_stats(const GroupNormArgs& a, hipStream_t s) {
    const double elems = (double)a.samples * a.P * (a.c0 + a.c1);
    std::string pname = "groupnorm_stats";
    if (prof_detail()) pname += gn_shape_tag(a);
    ProfScope ps(pname.c_str(), 3.0 * elems, (a.bf16 ? 2.0 : 4.0) * elems, s);              // algorithmic: one read
    groupnorm_stats_launch(a, s);
}

static int gcd_int(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// bf16 rows, 16-byte accesses: statistics and apply of one run of samples
static void groupnorm_bf16_launch(const GroupNormArgs& a, hipStream_t s) {
    static const int* const rowsp = E2V_AB_KNOB("E2V_GN_ROWS", 1);        // 0: the flat-index apply pass
    groupnorm_stats_launch(a, s);
    const int Ctot = a.c0 + a.c1;
    const size_t rows = (size_t)a.samples * a.P;
    if (*rowsp) {
        const int ot = oct_tile(a.c1 > 0 ? gcd_int(a.c0 / 8, a.c1 / 8) : a.c0 / 8);
        const int crows = gn_chunk_rows(a.P, a.small_chunks != 0);
        const dim3 grid((a.P + crows - 1) / crows, a.samples);
        dry_tag(" + gn_apply8_rows_kernel rows" + std::to_string(crows));
        h16_dispatch(a.bf16, [&](auto h16_tag) {
            using H = decltype(h16_tag);
            auto go = [&](auto kern) {
                E2V_KLAUNCH(kern, grid, dim3(256), 0, s, reinterpret_cast<const H*>(a.x0), reinterpret_cast<const H*>(a.x1), a.c0,
                            a.c1, a.ld0, a.ld1, a.ws_scale, reinterpret_cast<H*>(a.out), a.ldo, a.P, ot, crows);
            };
            if (a.silu) go(gn_apply8_rows_kernel<H, true>); else go(gn_apply8_rows_kernel<H, false>);
        });
        return;
    }
#ifdef E2V_AB
    const size_t tot8 = rows * (Ctot / 8);
    const int blk8 = (int)((tot8 + 255) / 256 < 16384 ? (tot8 + 255) / 256 : 16384);
    dry_tag(" + gn_apply8_kernel");
    h16_dispatch(a.bf16, [&](auto h16_tag) {
        using H = decltype(h16_tag);
        E2V_KLAUNCH(gn_apply8_kernel<H>, dim3(blk8), dim3(256), 0, s, reinterpret_cast<const H*>(a.x0),
                    reinterpret_cast<const H*>(a.x1), a.c0, a.c1, a.ld0, a.ld1, a.ws_scale, reinterpret_cast<H*>(a.out), a.ldo,
                    a.P, rows, a.silu);
    });
#else
    (void)rows; (void)Ctot;
#endif
}

void groupnorm(const GroupNormArgs& a, hipStream_t s) {
    const int Ctot = a.c0 + a.c1;
    const double elems = (double)a.samples * a.P * Ctot;
    std::string pname = a.silu ? "groupnorm_silu" : "groupnorm";
    if (prof_detail()) pname += gn_shape_tag(a);
    ProfScope ps(pname.c_str(), 8.0 * elems, 2.0 * (a.bf16 ? 2.0 : 4.0) * elems, s);   // algorithmic: read + write
    if (gn_fused_small_applies(a)) {
        const size_t smem = gn_fused_small_bytes(a);
        dry_tag(" -> gn_fused_small_kernel");
        h16_dispatch(a.bf16, [&](auto h16_tag) {
            using H = decltype(h16_tag);
            auto go = [&](auto kern) {
                E2V_KATTR(kern, 144 * 1024);
                E2V_KLAUNCH(kern, dim3(a.groups, a.samples), dim3(1024), smem, s, reinterpret_cast<const H*>(a.x0), reinterpret_cast<const H*>(a.x1),
                            a.c0, a.c1, a.ld0, a.ld1, a.gamma, a.beta, reinterpret_cast<H*>(a.out), a.ldo, a.P, a.groups, a.eps);
            };
            if (a.silu) go(gn_fused_small_kernel<H, true>); else go(gn_fused_small_kernel<H, false>);
        });
        return;
    }
#ifdef E2V_AB              // (the cooperative one-launch form: measured slower than the launches it saves, see gn_coop_kernel)
    if (const GnCoopPlan cp = gn_coop_plan(a); cp.ok) {
        bool launched = dry_run();
        if (!launched) {
            h16_dispatch(a.bf16, [&](auto h16_tag) {
                using H = decltype(h16_tag);
                auto go = [&](auto kern) {
                    E2V_KATTR(kern, 158 * 1024);
                    const H* x0 = reinterpret_cast<const H*>(a.x0); const H* x1 = reinterpret_cast<const H*>(a.x1);
                    H* out = reinterpret_cast<H*>(a.out);
                    int c0 = a.c0, c1 = a.c1, ld0 = a.ld0, ld1 = a.ld1, ldo = a.ldo, P = a.P, groups = a.groups, wps = cp.wps, rows = cp.rows, RS = cp.RS;
                    float eps = a.eps;
                    const float* gamma = a.gamma; const float* beta = a.beta;
                    float* part = a.ws_part; float* scsh = a.ws_scale;
                    void* args[] = {&x0, &x1, &c0, &c1, &ld0, &ld1, &gamma, &beta, &out, &ldo, &P, &groups, &eps, &wps, &rows, &RS, &part, &scsh};
                    // (the runtime refuses a cooperative grid that would not be resident as a whole: then the three launches below serve the call)
                    launched = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3(a.samples * cp.wps), dim3(1024), args, (unsigned)cp.smem, s) == hipSuccess;
                    if (!launched) (void)hipGetLastError();
                };
                if (a.silu) go(gn_coop_kernel<H, true>); else go(gn_coop_kernel<H, false>);
            });
        }
        if (launched) { dry_tag(" -> gn_coop_kernel wps" + std::to_string(cp.wps)); return; }
    }
#endif
    const size_t rows = (size_t)a.samples * a.P;
    const size_t total = rows * (Ctot / 4);
    const int blocks = (int)((total + 511) / 512 < 16384 ? (total + 511) / 512 : 16384);
    if (a.bf16 && a.c0 % 8 == 0 && a.c1 % 8 == 0 && ((a.ld0 | a.ld1 | a.ldo) & 7) == 0) {
        // The tensor is read twice (statistics, then the affine), and a B = 32 pass normalises 566 MB .. 1.7 GB tensors: the second
        // read finds nothing in the 256 MiB Infinity Cache.  E2V_GN_GROUP_MB > 0 sends runs of samples of that size through
        // statistics -> apply one after the other, so that a run's apply pass re-reads what its statistics pass has just brought
        // on-die.  MEASURED SLOWER at every level (tools/norm_micro.py, 64 samples: 0.32 ms as one run, 0.56 ms in runs of 64 MB,
        // 0.98 ms in runs of 32 MB at 13824 x 320): a run of 7 samples is 378 workgroups per launch and three dependent launches, the
        // chip is never full and the launch gaps cost more than the cache returns.  Default 0 = one run; the switch stays for the
        // measurement.
        static const int* const group_mb = E2V_AB_KNOB("E2V_GN_GROUP_MB", 0);
        const double sample_bytes = 2.0 * a.P * Ctot;
        int per = a.samples;
        if (*group_mb > 0 && sample_bytes * a.samples > 1.5e6 * *group_mb) {
            per = (int)(1.0e6 * *group_mb / sample_bytes);
            if (per < 1) per = 1;
        }
        const int chunks = groupnorm_chunks(a.P);
        for (int g0 = 0; g0 < a.samples; g0 += per) {
            GroupNormArgs b = a;
            b.samples = g0 + per <= a.samples ? per : a.samples - g0;
            const size_t r0 = (size_t)g0 * a.P;
            b.x0 = reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.x0) + r0 * a.ld0);       // (2-byte elements)
            if (a.c1 > 0) b.x1 = reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.x1) + r0 * a.ld1);
            b.out = reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(a.out) + r0 * a.ldo);
            b.ws_part = a.ws_part + (size_t)g0 * chunks * Ctot * 2;
            b.ws_scale = a.ws_scale + (size_t)g0 * Ctot * 2;
            groupnorm_bf16_launch(b, s);
        }
        return;
    }
    groupnorm_stats_launch(a, s);
    dry_tag(" + gn_apply_kernel");
    if (a.bf16)
        h16_dispatch(a.bf16, [&](auto h16_tag) {
            using H = decltype(h16_tag);
            E2V_KLAUNCH(gn_apply_kernel<H>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const H*>(a.x0),
                        reinterpret_cast<const H*>(a.x1), a.c0, a.c1, a.ld0, a.ld1, a.ws_scale, reinterpret_cast<H*>(a.out), a.ldo,
                        a.P, rows, a.silu);
        });
    else
        E2V_KLAUNCH(gn_apply_kernel<float>, dim3(blocks), dim3(256), 0, s, a.x0, a.x1, a.c0, a.c1, a.ld0, a.ld1, a.ws_scale, a.out,
                           a.ldo, a.P, rows, a.silu);
}

// ---- LayerNorm: one wave per row, row held in registers (C <= 64*4*NV) ------------------------------
template <int NV, typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out, int ldo,
                                                        int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int CQ = C / 4;
    const T* xr = x + (size_t)row * ldx;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int q = lane + 64 * k;
        if (q < CQ) {
            v[k] = ld4(xr + q * 4);
            s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        } else {
            v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int q = lane + 64 * k;
        if (q < CQ) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[k][e] - mean;
                ss += d * d;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float rstd = rsqrtf(ss / (float)C + eps);
    T* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int q = lane + 64 * k;
        if (q < CQ) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + q * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + q * 4);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[k][e] - mean) * rstd * g[e] + b[e];
            st4(orow + q * 4, y);
        }
    }
}

// bf16 rows: 8 channels (16 bytes) per lane and R rows per wave, all loads of the R rows issued before the first reduction
// (a wave that owns one 640-byte row spends its life on launch, two shuffle chains and a store: 2.9 TB/s effective)
template <typename H, int NV, int R>
__global__ __launch_bounds__(256) void layernorm_bf16_kernel(const H* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, H* __restrict__ out, int ldo,
                                                             int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int CO = C / 8;
    F8 v[R][NV];
    float s[R], ss[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = min(row0 + r, rows - 1);
        const H* xr = x + (size_t)row * ldx;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int q = lane + 64 * k;
            if (q < CO) v[r][k] = ld8(xr + q * 8);
            else { v[r][k].lo = f32x4{0.f, 0.f, 0.f, 0.f}; v[r][k].hi = v[r][k].lo; }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
            t += ((v[r][k].lo[0] + v[r][k].lo[1]) + (v[r][k].lo[2] + v[r][k].lo[3])) + ((v[r][k].hi[0] + v[r][k].hi[1]) + (v[r][k].hi[2] + v[r][k].hi[3]));
        s[r] = t;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] += __shfl_xor(s[r], off);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float mean = s[r] / (float)C;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (lane + 64 * k < CO) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = v[r][k].lo[e] - mean, d1 = v[r][k].hi[e] - mean;
                    t += d0 * d0 + d1 * d1;
                }
            }
        }
        ss[r] = t;
        s[r] = mean;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) ss[r] += __shfl_xor(ss[r], off);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int q = lane + 64 * k;
        if (q < CO) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + q * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + q * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + q * 8), b1 = *reinterpret_cast<const f32x4*>(beta + q * 8 + 4);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (row0 + r < rows) {
                    const float rstd = rsqrtf(ss[r] / (float)C + eps);
                    f32x4 lo, hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] = (v[r][k].lo[e] - s[r]) * rstd * g0[e] + b0[e];
                        hi[e] = (v[r][k].hi[e] - s[r]) * rstd * g1[e] + b1[e];
                    }
                    st8(out + (size_t)(row0 + r) * ldo + q * 8, lo, hi);
                }
            }
        }
    }
}

// Sub-wave rows: C = 320 / 640 / 1280 are 40 / 80 / 160 octets -- LPR = 8 / 16 / 32 lanes share a row (five 16-byte octets each, at a lane
// stride of LPR octets: every load instruction of the wave covers 128 contiguous bytes of 64 / LPR rows), all 64 lanes carry data
// (the wave-per-row form above: 40 of 64 at C = 320), the two reductions are 3-5 DPP adds inside the row's lanes instead of six
// wave-wide shuffles, and a wave keeps G row groups in flight.
template <int LPR>
__device__ __forceinline__ float rowlanes_allreduce(float x) {
    auto dpp = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});                       // quad_perm [1,0,3,2]
    x += dpp(x, std::integral_constant<int, 0x4E>{});                       // quad_perm [2,3,0,1]
    x += dpp(x, std::integral_constant<int, 0x141>{});                      // row_half_mirror: the other quad of the 8
    if constexpr (LPR >= 16) x += dpp(x, std::integral_constant<int, 0x140>{});  // row_mirror: the other 8 of the 16
    if constexpr (LPR >= 32) x += __shfl_xor(x, 16);
    return x;
}
template <typename H, int LPR, int G>
__global__ __launch_bounds__(256) void layernorm_bf16_rows_kernel(const H* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, H* __restrict__ out, int ldo,
                                                                  int rows, float eps, int stats_only) {
    constexpr int NV = 5, RW = 64 / LPR, C = LPR * NV * 8;
    const int lane = threadIdx.x & 63;
    const int c = lane % LPR, rl = lane / LPR;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (RW * G) + rl;
    if (row0 - rl >= rows) return;
    F8 v[G][NV];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int row = min(row0 + g * RW, rows - 1);
        const H* xr = x + (size_t)row * ldx + c * 8;
#pragma unroll
        for (int k = 0; k < NV; ++k) v[g][k] = ld8(xr + k * LPR * 8);
    }
    float mean[G], rstd[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
            t += ((v[g][k].lo[0] + v[g][k].lo[1]) + (v[g][k].lo[2] + v[g][k].lo[3])) + ((v[g][k].hi[0] + v[g][k].hi[1]) + (v[g][k].hi[2] + v[g][k].hi[3]));
        mean[g] = rowlanes_allreduce<LPR>(t) * (1.0f / (float)C);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = v[g][k].lo[e] - mean[g], d1 = v[g][k].hi[e] - mean[g];
                t += d0 * d0 + d1 * d1;
            }
        rstd[g] = rsqrtf(rowlanes_allreduce<LPR>(t) * (1.0f / (float)C) + eps);
    }
    if (stats_only) {      // TIMING EXPERIMENT (`make ab`, E2V_LN_STATS_ONLY = 1; results are wrong): what a statistics-only pass would cost --
#pragma unroll             // the ceiling of folding LayerNorm into its consumer GEMM: 8 bytes per row instead of the row
        for (int g = 0; g < G; ++g)
            if (c == 0 && row0 + g * RW < rows) {
                float* o = reinterpret_cast<float*>(out + (size_t)(row0 + g * RW) * ldo);
                o[0] = mean[g]; o[1] = rstd[g];
            }
        return;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int ch = (c + k * LPR) * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + ch), g1 = *reinterpret_cast<const f32x4*>(gamma + ch + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + ch), b1 = *reinterpret_cast<const f32x4*>(beta + ch + 4);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (row0 + g * RW < rows) {
                f32x4 lo, hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = (v[g][k].lo[e] - mean[g]) * rstd[g] * g0[e] + b0[e];
                    hi[e] = (v[g][k].hi[e] - mean[g]) * rstd[g] * g1[e] + b1[e];
                }
                st8(out + (size_t)(row0 + g * RW) * ldo + ch, lo, hi);
            }
        }
    }
}

// fp32 rows, same scheme: C = 320 / 640 / 1280 are 80 / 160 / 320 float4 quads = five per lane of 16 / 32 / 64 lanes
template <int LPR, int G>
__global__ __launch_bounds__(256) void layernorm_f32_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ out, int ldo,
                                                                 int rows, float eps) {
    constexpr int NV = 5, RW = 64 / LPR, C = LPR * NV * 4;
    const int lane = threadIdx.x & 63;
    const int c = lane % LPR, rl = lane / LPR;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (RW * G) + rl;
    if (row0 - rl >= rows) return;
    f32x4 v[G][NV];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int row = min(row0 + g * RW, rows - 1);
        const float* xr = x + (size_t)row * ldx + c * 4;
#pragma unroll
        for (int k = 0; k < NV; ++k) v[g][k] = *reinterpret_cast<const f32x4*>(xr + k * LPR * 4);
    }
    auto allreduce = [](float t) {
        if constexpr (LPR == 64) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            return t;
        } else {
            return rowlanes_allreduce<LPR>(t);
        }
    };
    float mean[G], rstd[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) t += (v[g][k][0] + v[g][k][1]) + (v[g][k][2] + v[g][k][3]);
        mean[g] = allreduce(t) * (1.0f / (float)C);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[g][k][e] - mean[g];
                t += d * d;
            }
        rstd[g] = rsqrtf(allreduce(t) * (1.0f / (float)C) + eps);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int ch = (c + k * LPR) * 4;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + ch), b0 = *reinterpret_cast<const f32x4*>(beta + ch);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (row0 + g * RW < rows) {
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (v[g][k][e] - mean[g]) * rstd[g] * g0[e] + b0[e];
                *reinterpret_cast<f32x4*>(out + (size_t)(row0 + g * RW) * ldo + ch) = y;
            }
        }
    }
}

template <typename T>
static void layernorm_launch(const T* x, int ldx, const float* gamma, const float* beta, T* out, int ldo, int rows, int C, float eps,
                             hipStream_t s) {
    const int blocks = (rows + 3) / 4;
    if (C <= 256)
        E2V_KLAUNCH((layernorm_kernel<1, T>), dim3(blocks), dim3(256), 0, s, x, ldx, gamma, beta, out, ldo, rows, C, eps);
    else if (C <= 768)
        E2V_KLAUNCH((layernorm_kernel<3, T>), dim3(blocks), dim3(256), 0, s, x, ldx, gamma, beta, out, ldo, rows, C, eps);
    else
        E2V_KLAUNCH((layernorm_kernel<5, T>), dim3(blocks), dim3(256), 0, s, x, ldx, gamma, beta, out, ldo, rows, C, eps);
}

void layernorm(const float* x, int ldx, const float* gamma, const float* beta, float* out, int ldo, int rows, int C,
               float eps, hipStream_t s, int bf16) {
    std::string pname = "layernorm";
    if (prof_detail()) pname += " rows" + std::to_string(rows) + " C" + std::to_string(C);
    ProfScope ps(pname.c_str(), 8.0 * rows * C, 2.0 * (bf16 ? 2.0 : 4.0) * rows * C, s);
    static const int* const rowsp = knob("E2V_LN_ROWS", 1);         // 0: one wave per row group of four rows
    if (dry_run()) {
        const bool shared = *rowsp && (C == 320 || C == 640 || C == 1280) && ldx % (bf16 ? 8 : 4) == 0 && ldo % (bf16 ? 8 : 4) == 0;
        dry_tag(std::string(" -> ") + (shared ? (bf16 ? "layernorm_bf16_rows_kernel" : "layernorm_f32_rows_kernel")
                                       : (bf16 && C % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && C <= 1536 ? "layernorm_bf16_kernel" : "layernorm_kernel")));
    }
    if (bf16) {
        h16_dispatch(bf16, [&](auto h16_tag) {
            using H = decltype(h16_tag);
            const H* xi = reinterpret_cast<const H*>(x);
            H* xo = reinterpret_cast<H*>(out);
            if (*rowsp && (C == 320 || C == 640 || C == 1280) && ldx % 8 == 0 && ldo % 8 == 0) {
                auto go = [&](auto kern, const int rows_per_wave) {
                    const int blocks = (rows + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
                    static const int* const stats_only = E2V_AB_KNOB("E2V_LN_STATS_ONLY", 0);      // (timing experiment, see the kernel)
                    E2V_KLAUNCH(kern, dim3(blocks), dim3(256), 0, s, xi, ldx, gamma, beta, xo, ldo, rows, eps, *stats_only);
                };
                if (C == 320) go(layernorm_bf16_rows_kernel<H, 8, 2>, 16);
                else if (C == 640) go(layernorm_bf16_rows_kernel<H, 16, 2>, 8);
                else go(layernorm_bf16_rows_kernel<H, 32, 2>, 4);
            } else if (C % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && C <= 1536) {
                constexpr int R = 4;
                const int blocks = (rows + 4 * R - 1) / (4 * R);
                if (C <= 512) E2V_KLAUNCH((layernorm_bf16_kernel<H, 1, R>), dim3(blocks), dim3(256), 0, s, xi, ldx, gamma, beta, xo, ldo, rows, C, eps);
                else if (C <= 1024) E2V_KLAUNCH((layernorm_bf16_kernel<H, 2, R>), dim3(blocks), dim3(256), 0, s, xi, ldx, gamma, beta, xo, ldo, rows, C, eps);
                else E2V_KLAUNCH((layernorm_bf16_kernel<H, 3, R>), dim3(blocks), dim3(256), 0, s, xi, ldx, gamma, beta, xo, ldo, rows, C, eps);
            } else {
                layernorm_launch(xi, ldx, gamma, beta, xo, ldo, rows, C, eps, s);
            }
        });
    }
    else if (*rowsp && (C == 320 || C == 640 || C == 1280) && ldx % 4 == 0 && ldo % 4 == 0) {
        auto go = [&](auto kern, const int rows_per_wave) {
            const int blocks = (rows + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
            E2V_KLAUNCH(kern, dim3(blocks), dim3(256), 0, s, x, ldx, gamma, beta, out, ldo, rows, eps);
        };
        if (C == 320) go(layernorm_f32_rows_kernel<16, 2>, 8);
        else if (C == 640) go(layernorm_f32_rows_kernel<32, 2>, 4);
        else go(layernorm_f32_rows_kernel<64, 2>, 2);
    } else
        layernorm_launch(x, ldx, gamma, beta, out, ldo, rows, C, eps, s);
}

}  // namespace e2v
