// fp32 implicit-GEMM convolution / linear on the gfx950 matrix cores.
//
// One kernel serves InflatedConv3d 3x3 (resnet.py:10-18, stride 1/2, with the nearest resize of
// Upsample3D folded into the gather), every 1x1 conv / nn.Linear of the path, and the batched
// QK^T / PV products of the VAE attention block.  Arithmetic is v_mfma_f32_32x32x2_f32: exact
// fp32 products, fp32 accumulate (bitwise an fmaf chain) -- the parity configuration of BASELINE.json.
//
// Tiling: 256 threads = 4 waves as 2x2; block tile BM x BN x 32, LDS double-buffered, global loads
// of step k+1 issued before the MFMAs of step k (register staging, one barrier per step).
// LDS rows are padded to 36 floats: a wave's ds_read_b128 of 16 distinct rows then covers all 64 banks
// exactly once.  Each lane feeds 4 consecutive k (one b128) to 4 MFMAs: the two lane halves own
// k = 8g + {0..3} and 8g + {4..7}; A and B use the same assignment so the sum is over all 32 k.
#include "kernels.h"
#include "prof.h"

namespace e2v {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int BK = 32;
static constexpr int LDS_LD = 36;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int BM, int BN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AR = BM / 32, BR = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDS_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: the 8 XCDs are dealt blocks round-robin; give each XCD a contiguous run of
    // tiles (n fastest) so that the column blocks sharing one gathered A tile hit the same L2.
    const int nbn = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int tile;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bm = tile / nbn, bn = tile % nbn;

    const float* __restrict__ a0 = p.a0 + (size_t)blockIdx.z * p.sa0;
    const float* __restrict__ a1 = p.a1;
    const float* __restrict__ w = p.w + (size_t)blockIdx.z * p.sw;
    float* __restrict__ out = p.out + (size_t)blockIdx.z * p.sout;

    const int Ctot = p.c0 + p.c1;
    const int cchunks = (Ctot + BK - 1) / BK;
    const int nk = p.taps * cchunks;

    const int c4 = tid & 7;
    const int r0 = tid >> 3;

    // per-thread A row descriptors
    int a_img[AR], a_y[AR], a_x[AR];
    bool a_ok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = bm * BM + r0 + 32 * i;
        a_ok[i] = m < p.M;
        if (p.taps == 1) {
            a_img[i] = 0; a_y[i] = 0; a_x[i] = a_ok[i] ? m : 0;
        } else {
            const int hw = p.Ho * p.Wo;
            const int mm = a_ok[i] ? m : 0;
            const int img = mm / hw;
            const int rem = mm - img * hw;
            const int oy = rem / p.Wo;
            a_img[i] = img;
            a_y[i] = oy * p.stride - p.pad;
            a_x[i] = (rem - oy * p.Wo) * p.stride - p.pad;
        }
    }
    size_t b_row[BR];
    bool b_ok[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int n = bn * BN + r0 + 32 * j;
        b_ok[j] = n < p.N;
        b_row[j] = (size_t)(b_ok[j] ? n : 0) * p.ldw;
    }

    f32x4 ra[AR], rb[BR];
    auto load_tile = [&](int ks) {
        const int tap = ks / cchunks;
        const int c = (ks - tap * cchunks) * BK + c4 * 4;
        const bool c_ok = c < Ctot;
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            bool ok = a_ok[i] && c_ok;
            size_t pix;
            if (p.taps == 1) {
                pix = (size_t)a_x[i];
            } else {
                const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                ok = ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                int sy = ok ? iy : 0, sx = ok ? ix : 0;
                if (p.upsample) {      // torch nearest: src = min(floor(dst * (in/out)), in - 1), fp32 scale
                    sy = min((int)floorf((float)sy * p.ups_h), p.Hs - 1);
                    sx = min((int)floorf((float)sx * p.ups_w), p.Ws - 1);
                }
                pix = ((size_t)a_img[i] * p.Hs + sy) * p.Ws + sx;
            }
            if (ok) {
                const float* src = (c < p.c0) ? (a0 + pix * p.lda0 + c) : (a1 + pix * p.lda1 + (c - p.c0));
                v = *reinterpret_cast<const f32x4*>(src);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b_ok[j] && c_ok) v = *reinterpret_cast<const f32x4*>(w + b_row[j] + (size_t)tap * Ctot + c);
            rb[j] = v;
        }
    };
    auto store_tile = [&](int buf) {
        float* A = As + buf * BM * LDS_LD;
        float* B = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(A + (r0 + 32 * i) * LDS_LD + c4 * 4) = ra[i];
#pragma unroll
        for (int j = 0; j < BR; ++j) *reinterpret_cast<f32x4*>(B + (r0 + 32 * j) * LDS_LD + c4 * 4) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frag_off = (lane & 31) * LDS_LD + (lane >> 5) * 4;
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_tile(ks + 1);
        const float* A = As + buf * BM * LDS_LD + wm * WM * LDS_LD + frag_off;
        const float* B = Bs + buf * BN * LDS_LD + wn * WN * LDS_LD + frag_off;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(A + mi * 32 * LDS_LD + g * 8);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bf[ni] = *reinterpret_cast<const f32x4*>(B + ni * 32 * LDS_LD + g * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][s], bf[ni][s], acc[mi][ni], 0, 0, 0);
        }
        if (ks + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA -- column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int col = lane & 31;
    const int rquad = (lane >> 5) * 4;
    if (p.geglu) {
        if constexpr (TN == 2) {
            const int nv = bn * BN + wn * WN + col;
            if (nv + 32 < p.N) {
                const float bv = p.bias ? p.bias[nv] : 0.f;
                const float bg = p.bias ? p.bias[nv + 32] : 0.f;
                const int no = (bn * BN + wn * WN) / 2 + col;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = bm * BM + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + rquad;
                        if (m < p.M) {
                            const float v = acc[mi][0][r] * p.alpha + bv;
                            const float g = acc[mi][1][r] * p.alpha + bg;
                            out[(size_t)m * p.ldc + no] = v * gelu_erf(g);
                        }
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int n = bn * BN + wn * WN + ni * 32 + col;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm * BM + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + rquad;
                if (m < p.M) {
                    float v = acc[mi][ni][r] * p.alpha + bv;
                    if (p.rowbias) v += p.rowbias[(size_t)(m / p.rows_per_sample) * p.rb_ld + n];
                    if (p.resid) v += p.resid[(size_t)m * p.ldr + n];
                    out[(size_t)m * p.ldc + n] = v;
                }
            }
    }
}

template <int BM, int BN>
static void launch_igemm(const IgemmArgs& a, hipStream_t s) {
    static bool configured = false;
    constexpr size_t smem = (size_t)(2 * BM + 2 * BN) * LDS_LD * sizeof(float);
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        configured = true;
    }
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    dim3 grid(nbm * nbn, 1, a.batch);
    const double K = (double)a.taps * (a.c0 + a.c1);
    const double rows_in = a.taps == 1 ? (double)a.M : (double)a.M * a.Hs * a.Ws / ((double)a.Ho * a.Wo);
    ProfScope ps(BN == 128 ? "igemm_f32_128x128" : "igemm_f32_128x64", 2.0 * a.M * a.N * K * a.batch,
                 4.0 * a.batch * (rows_in * (a.c0 + a.c1) + (double)a.N * K + (double)a.M * (a.geglu ? a.N / 2 : a.N)), s);
    hipLaunchKernelGGL((igemm_kernel<BM, BN>), grid, dim3(256), smem, s, a);
}

void igemm(const IgemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return;
    if (a.geglu || a.N % 128 == 0 || a.N > 1024)
        launch_igemm<128, 128>(a, s);
    else
        launch_igemm<128, 64>(a, s);
}

// ---- one-off weight re-layout ---------------------------------------------------------------------
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ o, int cout, int cin, int cin_pad) {
    const size_t total = (size_t)cout * 9 * cin_pad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % cin_pad;
        const int tap = (i / cin_pad) % 9;
        const int oc = i / ((size_t)cin_pad * 9);
        o[i] = (c < cin) ? w[((size_t)oc * cin + c) * 9 + tap] : 0.f;
    }
}
void pack_conv3x3(const float* w, float* o, int cout, int cin, int cin_pad, hipStream_t s) {
    const size_t total = (size_t)cout * 9 * cin_pad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(blocks), dim3(256), 0, s, w, o, cout, cin, cin_pad);
}

__global__ void copy_rows_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int rows,
                                 int cols) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = i / cols, c = i % cols;
        dst[(size_t)r * ldd + c] = src[(size_t)r * lds + c];
    }
}
void copy_rows(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s) {
    const size_t total = (size_t)rows * cols;
    if (!total) return;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(blocks), dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
}

}  // namespace e2v
