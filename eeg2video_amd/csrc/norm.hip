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
#include "kernels.h"
#include "prof.h"
#include "act_io.h"

namespace e2v {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int GN_ROWS_PER_CHUNK = 256;

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
                                                         float* __restrict__ part, int Ctot, int coff, int QT) {
    __shared__ f32x4 red[2][256];
    const int chunk = blockIdx.x, slab = blockIdx.y;
    const int R = 256 / QT;
    const int q = threadIdx.x % QT, r = threadIdx.x / QT;
    const int p0 = chunk * GN_ROWS_PER_CHUNK;
    const int p1 = min(P, p0 + GN_ROWS_PER_CHUNK);
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

// grid (groups, slabs), one wave each
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part, int chunks, int Ctot, int groups,
                                                         int P, float eps, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ scsh) {
    const int g = blockIdx.x, slab = blockIdx.y, lane = threadIdx.x;
    const int cpg = Ctot / groups;
    double s = 0.0, ss = 0.0;
    const int total = chunks * cpg;
    for (int i = lane; i < total; i += 64) {
        const int ch = i / cpg, c = g * cpg + (i - ch * cpg);
        const float* e = part + ((size_t)(slab * chunks + ch) * Ctot + c) * 2;
        s += (double)e[0];
        ss += (double)e[1];
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

static void groupnorm_stats_launch(const GroupNormArgs& a, hipStream_t s) {
    const int Ctot = a.c0 + a.c1;
    const int chunks = groupnorm_chunks(a.P);
    auto part = [&](const float* x, int ld, int C, int coff) {
        const int qt = quad_tile(C / 4);
        if (a.bf16)
            hipLaunchKernelGGL(gn_partial_kernel<__bf16>, dim3(chunks, a.samples), dim3(256), 0, s, reinterpret_cast<const __bf16*>(x), ld, C,
                               a.P, chunks, a.ws_part, Ctot, coff, qt);
        else
            hipLaunchKernelGGL(gn_partial_kernel<float>, dim3(chunks, a.samples), dim3(256), 0, s, x, ld, C, a.P, chunks, a.ws_part, Ctot,
                               coff, qt);
    };
    part(a.x0, a.ld0, a.c0, 0);
    if (a.c1 > 0) part(a.x1, a.ld1, a.c1, a.c0);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(a.groups, a.samples), dim3(64), 0, s, a.ws_part, chunks, Ctot, a.groups,
                       a.P, a.eps, a.gamma, a.beta, a.ws_scale);
}

void groupnorm_stats(const GroupNormArgs& a, hipStream_t s) {
    const double elems = (double)a.samples * a.P * (a.c0 + a.c1);
    ProfScope ps("groupnorm_stats", 3.0 * elems, (a.bf16 ? 2.0 : 4.0) * elems, s);              // algorithmic: one read
    groupnorm_stats_launch(a, s);
}

void groupnorm(const GroupNormArgs& a, hipStream_t s) {
    const int Ctot = a.c0 + a.c1;
    const double elems = (double)a.samples * a.P * Ctot;
    ProfScope ps(a.silu ? "groupnorm_silu" : "groupnorm", 8.0 * elems, 2.0 * (a.bf16 ? 2.0 : 4.0) * elems, s);   // algorithmic: read + write
    groupnorm_stats_launch(a, s);
    const size_t rows = (size_t)a.samples * a.P;
    const size_t total = rows * (Ctot / 4);
    const int blocks = (int)((total + 511) / 512 < 16384 ? (total + 511) / 512 : 16384);
    if (a.bf16)
        hipLaunchKernelGGL(gn_apply_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const __bf16*>(a.x0),
                           reinterpret_cast<const __bf16*>(a.x1), a.c0, a.c1, a.ld0, a.ld1, a.ws_scale, reinterpret_cast<__bf16*>(a.out), a.ldo,
                           a.P, rows, a.silu);
    else
        hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(blocks), dim3(256), 0, s, a.x0, a.x1, a.c0, a.c1, a.ld0, a.ld1, a.ws_scale, a.out,
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

template <typename T>
static void layernorm_launch(const T* x, int ldx, const float* gamma, const float* beta, T* out, int ldo, int rows, int C, float eps,
                             hipStream_t s) {
    const int blocks = (rows + 3) / 4;
    if (C <= 256)
        hipLaunchKernelGGL((layernorm_kernel<1, T>), dim3(blocks), dim3(256), 0, s, x, ldx, gamma, beta, out, ldo, rows, C, eps);
    else if (C <= 768)
        hipLaunchKernelGGL((layernorm_kernel<3, T>), dim3(blocks), dim3(256), 0, s, x, ldx, gamma, beta, out, ldo, rows, C, eps);
    else
        hipLaunchKernelGGL((layernorm_kernel<5, T>), dim3(blocks), dim3(256), 0, s, x, ldx, gamma, beta, out, ldo, rows, C, eps);
}

void layernorm(const float* x, int ldx, const float* gamma, const float* beta, float* out, int ldo, int rows, int C,
               float eps, hipStream_t s, int bf16) {
    ProfScope ps("layernorm", 8.0 * rows * C, 2.0 * (bf16 ? 2.0 : 4.0) * rows * C, s);
    if (bf16)
        layernorm_launch(reinterpret_cast<const __bf16*>(x), ldx, gamma, beta, reinterpret_cast<__bf16*>(out), ldo, rows, C, eps, s);
    else
        layernorm_launch(x, ldx, gamma, beta, out, ldo, rows, C, eps, s);
}

}  // namespace e2v
