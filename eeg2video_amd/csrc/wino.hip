// Winograd F(2x2, 3x3) form of the stride-1 3x3 convolutions of ResnetBlock3D / Upsample3D (resnet.py:58-66,
// 180, 197) and of the VAE's resnets: 16 multiplies per 2x2 output tile instead of 36, i.e. 2.25x fewer
// matrix-core flops than the direct implicit GEMM, paid for with two HBM-bound transform passes.
//
//   V[k][t][c] = (B^T d B)[k]        d = 4x4 input patch of tile t, channel c            (wino_in_kernel)
//   M[k][t][o] = sum_c V[k][t][c] U[k][o][c]      16 independent GEMMs = igemm(), batch 16
//   y[2x2]     = A^T M A (+ bias, time-embedding row, residual)                          (wino_out_kernel)
//   U[k][o][c] = (G g G^T)[k]        once, at e2v_finalize_weights                       (wino_weight_kernel)
//
// The input transform also absorbs what precedes the conv in the graph: the channel concat of the up blocks
// (two sources), the nearest 2x resize of Upsample3D (torch's fp32-scale index formula) and, for the resnets,
// the GroupNorm affine + SiLU (per-(slab, channel) scale / shift from groupnorm_stats) -- the normalised
// activation is then never written to HBM.  Zero padding is applied after the activation, as F.conv2d does.
//
// fp32 throughout; F(2x2, 3x3) has transform constants 0, +-1, +-1/2 only, its rounding error stays within a small
// multiple of the direct sum's (tests/test_hip_ops.py::test_conv3x3_winograd pins 2e-5 relative to the output scale).
#include "kernels.h"
#include "prof.h"

#include <string>

namespace e2v {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// x * sigmoid(x) on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each).  The input transforms evaluate it
// up to 2.25x per element (once per tile that touches the pixel), so the IEEE expf + division of norm.hip's
// gn_apply_kernel would make these HBM-bound kernels ALU-bound; the two forms differ by ~1e-7 relative.
__device__ __forceinline__ float wino_silu(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}

// one thread = one tile x four channels; consecutive threads = consecutive channel quads (16-byte lanes, coalesced)
__global__ __launch_bounds__(256) void wino_in_kernel(const WinoArgs p, int img_lo, int nimg, float* __restrict__ V) {
    const int Ctot = p.c0 + p.c1;
    const int CQ = Ctot / 4;
    const int th = (p.Ho + 1) / 2, tw = (p.Wo + 1) / 2;
    const size_t T = (size_t)nimg * th * tw;
    const size_t total = T * CQ;
    const size_t plane = T * Ctot;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / CQ;
        const int c = (int)(i - t * CQ) * 4;
        const int img = (int)(t / (th * tw));
        const int rem = (int)(t - (size_t)img * th * tw);
        const int ty = rem / tw, tx = rem - ty * tw;
        const bool second = c >= p.c0;
        const float* __restrict__ src = second ? p.x1 + (c - p.c0) : p.x0 + c;
        const int ld = second ? p.ld1 : p.ld0;
        const size_t img_row = (size_t)(img_lo + img) * p.Hs * p.Ws;
        f32x4 ga = {1.f, 0.f, 1.f, 0.f}, gb = {1.f, 0.f, 1.f, 0.f};
        if (p.gn_scsh) {                                   // (scale, shift) pairs of the 4 channels
            const size_t slab = img_row / (size_t)p.gn_P;
            const float* sc = p.gn_scsh + (slab * Ctot + c) * 2;
            ga = *reinterpret_cast<const f32x4*>(sc);
            gb = *reinterpret_cast<const f32x4*>(sc + 4);
        }
        f32x4 d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = 2 * ty - 1 + r;
            const bool yok = (unsigned)iy < (unsigned)p.Ho;
            int sy = yok ? iy : 0;
            if (p.upsample) sy = min((int)floorf((float)sy * p.ups_h), p.Hs - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ix = 2 * tx - 1 + q;
                const bool ok = yok && (unsigned)ix < (unsigned)p.Wo;
                int sx = ok ? ix : 0;
                if (p.upsample) sx = min((int)floorf((float)sx * p.ups_w), p.Ws - 1);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    v = *reinterpret_cast<const f32x4*>(src + (img_row + (size_t)sy * p.Ws + sx) * ld);
                    if (p.gn_scsh) {
                        v[0] = v[0] * ga[0] + ga[1];
                        v[1] = v[1] * ga[2] + ga[3];
                        v[2] = v[2] * gb[0] + gb[1];
                        v[3] = v[3] * gb[2] + gb[3];
                        if (p.gn_silu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = wino_silu(v[e]);
                        }
                    }
                }
                d[r][q] = v;
            }
        }
        // B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        f32x4 u[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u[0][q] = d[0][q] - d[2][q];
            u[1][q] = d[1][q] + d[2][q];
            u[2][q] = d[2][q] - d[1][q];
            u[3][q] = d[1][q] - d[3][q];
        }
        float* __restrict__ o = V + t * Ctot + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            *reinterpret_cast<f32x4*>(o + (size_t)(4 * r + 0) * plane) = u[r][0] - u[r][2];
            *reinterpret_cast<f32x4*>(o + (size_t)(4 * r + 1) * plane) = u[r][1] + u[r][2];
            *reinterpret_cast<f32x4*>(o + (size_t)(4 * r + 2) * plane) = u[r][2] - u[r][1];
            *reinterpret_cast<f32x4*>(o + (size_t)(4 * r + 3) * plane) = u[r][1] - u[r][3];
        }
    }
}

// one thread = one tile x four output channels: y = A^T M A,  A^T = [1 1 1 0; 0 1 -1 -1]
__global__ __launch_bounds__(256) void wino_out_kernel(const WinoArgs p, int img_lo, int nimg, const float* __restrict__ Mb) {
    const int NQ = p.N / 4;
    const int th = (p.Ho + 1) / 2, tw = (p.Wo + 1) / 2;
    const size_t T = (size_t)nimg * th * tw;
    const size_t total = T * NQ;
    const size_t plane = T * p.N;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / NQ;
        const int n = (int)(i - t * NQ) * 4;
        const int img = (int)(t / (th * tw));
        const int rem = (int)(t - (size_t)img * th * tw);
        const int ty = rem / tw, tx = rem - ty * tw;
        const float* __restrict__ m = Mb + t * p.N + n;
        f32x4 w[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(m + (size_t)(0 + q) * plane);
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(m + (size_t)(4 + q) * plane);
            const f32x4 m2 = *reinterpret_cast<const f32x4*>(m + (size_t)(8 + q) * plane);
            const f32x4 m3 = *reinterpret_cast<const f32x4*>(m + (size_t)(12 + q) * plane);
            w[0][q] = m0 + m1 + m2;
            w[1][q] = m1 - m2 - m3;
        }
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = 2 * ty + a;
            if (oy >= p.Ho) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ox = 2 * tx + b;
                if (ox >= p.Wo) continue;
                f32x4 y = b == 0 ? w[a][0] + w[a][1] + w[a][2] : w[a][1] - w[a][2] - w[a][3];
                const size_t row = ((size_t)(img_lo + img) * p.Ho + oy) * p.Wo + ox;
                y += bias;
                if (p.rowbias) y += *reinterpret_cast<const f32x4*>(p.rowbias + (row / p.rows_per_sample) * p.rb_ld + n);
                if (p.resid) y += *reinterpret_cast<const f32x4*>(p.resid + row * p.ldr + n);
                *reinterpret_cast<f32x4*>(p.out + row * p.ldc + n) = y;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// F(4x4, 3x3): 36 multiplies per 4x4 output tile instead of 144 (4x fewer than direct, 1.78x fewer than F(2x2)).
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// Transform constants up to 8 make its fp32 rounding error ~17x the direct sum's (5e-6 of the output scale on a
// 640-channel conv): opt-in (E2V_CONV_WINOGRAD4 / E2V_WINO_F4), see DESIGN 3.6.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bt6(const f32x4 (&d)[6], f32x4 (&o)[6]) {
    o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    o[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
    o[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    o[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    o[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
__device__ __forceinline__ void at6(const f32x4 (&m)[6], f32x4 (&o)[4]) {
    const f32x4 a = m[1] + m[2], b = m[1] - m[2], c = m[3] + m[4], e = m[3] - m[4];
    o[0] = m[0] + a + c;
    o[1] = b + 2.f * e;
    o[2] = a + 4.f * c;
    o[3] = b + 8.f * e + m[5];
}

__global__ __launch_bounds__(256) void wino4_in_kernel(const WinoArgs p, int img_lo, int nimg, float* __restrict__ V) {
    const int Ctot = p.c0 + p.c1;
    const int CQ = Ctot / 4;
    const int th = (p.Ho + 3) / 4, tw = (p.Wo + 3) / 4;
    const size_t T = (size_t)nimg * th * tw;
    const size_t total = T * CQ;
    const size_t plane = T * Ctot;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / CQ;
        const int c = (int)(i - t * CQ) * 4;
        const int img = (int)(t / (th * tw));
        const int rem = (int)(t - (size_t)img * th * tw);
        const int ty = rem / tw, tx = rem - ty * tw;
        const bool second = c >= p.c0;
        const float* __restrict__ src = second ? p.x1 + (c - p.c0) : p.x0 + c;
        const int ld = second ? p.ld1 : p.ld0;
        const size_t img_row = (size_t)(img_lo + img) * p.Hs * p.Ws;
        f32x4 ga = {1.f, 0.f, 1.f, 0.f}, gb = {1.f, 0.f, 1.f, 0.f};
        if (p.gn_scsh) {
            const size_t slab = img_row / (size_t)p.gn_P;
            const float* sc = p.gn_scsh + (slab * Ctot + c) * 2;
            ga = *reinterpret_cast<const f32x4*>(sc);
            gb = *reinterpret_cast<const f32x4*>(sc + 4);
        }
        // column pass first (B^T d), one patch column at a time so that only the 6x6 intermediate stays live
        f32x4 u[6][6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int ix = 4 * tx - 1 + q;
            const bool xok = (unsigned)ix < (unsigned)p.Wo;
            int sx = xok ? ix : 0;
            if (p.upsample) sx = min((int)floorf((float)sx * p.ups_w), p.Ws - 1);
            f32x4 d[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int iy = 4 * ty - 1 + r;
                const bool ok = xok && (unsigned)iy < (unsigned)p.Ho;
                int sy = ok ? iy : 0;
                if (p.upsample) sy = min((int)floorf((float)sy * p.ups_h), p.Hs - 1);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    v = *reinterpret_cast<const f32x4*>(src + (img_row + (size_t)sy * p.Ws + sx) * ld);
                    if (p.gn_scsh) {
                        v[0] = v[0] * ga[0] + ga[1];
                        v[1] = v[1] * ga[2] + ga[3];
                        v[2] = v[2] * gb[0] + gb[1];
                        v[3] = v[3] * gb[2] + gb[3];
                        if (p.gn_silu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = wino_silu(v[e]);
                        }
                    }
                }
                d[r] = v;
            }
            f32x4 o[6];
            bt6(d, o);
#pragma unroll
            for (int r = 0; r < 6; ++r) u[r][q] = o[r];
        }
        float* __restrict__ o = V + t * Ctot + c;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            f32x4 v[6];
            bt6(u[r], v);
#pragma unroll
            for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(o + (size_t)(6 * r + q) * plane) = v[q];
        }
    }
}

__global__ __launch_bounds__(256) void wino4_out_kernel(const WinoArgs p, int img_lo, int nimg, const float* __restrict__ Mb) {
    const int NQ = p.N / 4;
    const int th = (p.Ho + 3) / 4, tw = (p.Wo + 3) / 4;
    const size_t T = (size_t)nimg * th * tw;
    const size_t total = T * NQ;
    const size_t plane = T * p.N;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / NQ;
        const int n = (int)(i - t * NQ) * 4;
        const int img = (int)(t / (th * tw));
        const int rem = (int)(t - (size_t)img * th * tw);
        const int ty = rem / tw, tx = rem - ty * tw;
        const float* __restrict__ m = Mb + t * p.N + n;
        f32x4 w[4][6];                                         // A^T M, one column of M at a time
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            f32x4 col[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) col[r] = *reinterpret_cast<const f32x4*>(m + (size_t)(6 * r + q) * plane);
            f32x4 o[4];
            at6(col, o);
#pragma unroll
            for (int a = 0; a < 4; ++a) w[a][q] = o[a];
        }
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oy = 4 * ty + a;
            if (oy >= p.Ho) continue;
            f32x4 y[4];
            at6(w[a], y);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ox = 4 * tx + b;
                if (ox >= p.Wo) continue;
                const size_t row = ((size_t)(img_lo + img) * p.Ho + oy) * p.Wo + ox;
                f32x4 v = y[b] + bias;
                if (p.rowbias) v += *reinterpret_cast<const f32x4*>(p.rowbias + (row / p.rows_per_sample) * p.rb_ld + n);
                if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + row * p.ldr + n);
                *reinterpret_cast<f32x4*>(p.out + row * p.ldc + n) = v;
            }
        }
    }
}

// [O][I][3][3] -> U[36][O][I] = G g G^T
__global__ void wino4_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int cout, int cin) {
    const size_t total = (size_t)cout * cin;
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float* g = w + i * 9;
        float t[6][3];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int x = 0; x < 3; ++x) t[r][x] = G[r][0] * g[x] + G[r][1] * g[3 + x] + G[r][2] * g[6 + x];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int q = 0; q < 6; ++q)
                U[(size_t)(6 * r + q) * total + i] = t[r][0] * G[q][0] + t[r][1] * G[q][1] + t[r][2] * G[q][2];
    }
}

// [O][I][3][3] -> U[16][O][I] = G g G^T,  G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
__global__ void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int cout, int cin) {
    const size_t total = (size_t)cout * cin;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float* g = w + i * 9;
        float t[4][3];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            t[0][x] = g[x];
            t[1][x] = 0.5f * (g[x] + g[3 + x] + g[6 + x]);
            t[2][x] = 0.5f * (g[x] - g[3 + x] + g[6 + x]);
            t[3][x] = g[6 + x];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            U[(size_t)(4 * r + 0) * total + i] = t[r][0];
            U[(size_t)(4 * r + 1) * total + i] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
            U[(size_t)(4 * r + 2) * total + i] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
            U[(size_t)(4 * r + 3) * total + i] = t[r][2];
        }
    }
}

void wino_pack_weights(const float* w_oihw, float* U, int cout, int cin, int m, hipStream_t s) {
    const size_t total = (size_t)cout * cin;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (m == 4) E2V_KLAUNCH(wino4_weight_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, U, cout, cin);
    else E2V_KLAUNCH(wino_weight_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, U, cout, cin);
}

static inline size_t wino_tiles(const WinoArgs& a, int nimg) {
    return (size_t)nimg * ((a.Ho + a.m - 1) / a.m) * ((a.Wo + a.m - 1) / a.m);
}

size_t wino_workspace_floats(const WinoArgs& a, int nimg) {
    return (size_t)(a.m + 2) * (a.m + 2) * wino_tiles(a, nimg) * (size_t)(a.c0 + a.c1 + a.N);
}

int wino_chunk_images(const WinoArgs& a, size_t max_floats) {
    const size_t per_img = wino_workspace_floats(a, 1);
    size_t n = max_floats / (per_img ? per_img : 1);
    if (n < 1) n = 1;
    return (int)(n < (size_t)a.nimg ? n : (size_t)a.nimg);
}

void wino_conv3x3(const WinoArgs& a, float* ws, int chunk_images, hipStream_t s) {
    const int Ctot = a.c0 + a.c1;
    const int P = (a.m + 2) * (a.m + 2);                          // 16 or 36 GEMMs
    for (int lo = 0; lo < a.nimg; lo += chunk_images) {
        const int n = a.nimg - lo < chunk_images ? a.nimg - lo : chunk_images;
        const size_t T = wino_tiles(a, n);
        float* V = ws;
        float* Mb = ws + (size_t)P * T * Ctot;
        {
            const size_t total = T * (Ctot / 4);
            const double px = (double)a.m * a.m;                  // output pixels per tile
            std::string nm = a.gn_scsh ? "wino_in_gn_silu" : "wino_in";
            if (prof_detail())
                nm += " T" + std::to_string(T) + " C" + std::to_string(Ctot) + " m" + std::to_string(a.m) + (a.c1 ? " cat" : "") + (a.upsample ? " up" : "");
            ProfScope ps(nm.c_str(), 2.0 * P * T * Ctot, 4.0 * ((px + P) * T * Ctot), s);
            const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
            if (a.m == 4) E2V_KLAUNCH(wino4_in_kernel, dim3(blocks), dim3(256), 0, s, a, lo, n, V);
            else E2V_KLAUNCH(wino_in_kernel, dim3(blocks), dim3(256), 0, s, a, lo, n, V);
        }
        IgemmArgs g;
        g.a0 = V; g.c0 = Ctot; g.lda0 = Ctot; g.w = a.U; g.ldw = Ctot; g.ldw16 = Ctot;
        g.out = Mb; g.ldc = a.N; g.M = (int)T; g.N = a.N; g.taps = 1;
        g.batch = P; g.sa0 = (long long)T * Ctot; g.sw = (long long)a.N * Ctot; g.sout = (long long)T * a.N;
        if (a.U3) { g.x3 = 1; g.w3 = a.U3; g.w3_plane = (long long)P * a.N * Ctot; }
        igemm(g, s);
        {
            const size_t total = T * (a.N / 4);
            const double px = (double)a.m * a.m;
            std::string nm = "wino_out";
            if (prof_detail())
                nm += " T" + std::to_string(T) + " N" + std::to_string(a.N) + " m" + std::to_string(a.m) + (a.resid ? " res" : "") + (a.rowbias ? " temb" : "");
            ProfScope ps(nm.c_str(), 1.5 * P * T * a.N, 4.0 * (P * T * a.N + px * T * a.N * (a.resid ? 2 : 1)), s);
            const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
            if (a.m == 4) E2V_KLAUNCH(wino4_out_kernel, dim3(blocks), dim3(256), 0, s, a, lo, n, Mb);
            else E2V_KLAUNCH(wino_out_kernel, dim3(blocks), dim3(256), 0, s, a, lo, n, Mb);
        }
    }
}

}  // namespace e2v
