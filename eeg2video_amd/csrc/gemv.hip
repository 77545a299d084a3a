// Weight-streaming GEMV for the Semantic Predictor MLP at small batch (CLIP.forward,
// EEG2Video/models/train_semantic_predictor.py:11-32; 0.89 G parameters, batch 1 .. a few in the reference's loop).
//
// out[b][n] = act(sum_k x[b][k] W[n][k] + bias[n]),  B <= 16 rows, W [N][K] fp32 or bf16.
// Every weight byte is used once, so the op is bound by streaming W from HBM (3.6 GB per call in fp32, 1.8 GB in bf16): the MFMA
// tile kernel would spend a 128-row tile on B rows.  Here a wave owns RW weight rows at a time and walks them with 16-byte
// loads, 1 KB per wave-instruction (12 waves per CU keep the stream busy); x is staged once per block in LDS as fp32 (rounded to
// bf16 first in the bf16 mode, so that the operands are the ones the MFMA path would multiply) and read with conflict-free
// 16-byte LDS reads; fp32 accumulation, one butterfly reduction per (row, batch entry) at the end.
#include "h16.h"
#include "kernels.h"
#include "prof.h"

namespace e2v {

typedef float gf32x4 __attribute__((ext_vector_type(4)));

template <typename H> struct WVec {                               // H: bf16 / fp16 (h16.h)
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const H* p, float (&v)[8]) {
        const hx8<H> a = *reinterpret_cast<const hx8<H>*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
    }
};
template <> struct WVec<float> {
    static constexpr int N = 4;                                  // elements per 16-byte load
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const gf32x4 a = *reinterpret_cast<const gf32x4*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a[e];
    }
};

// grid: ceil(N / (4 * RW)) blocks of 4 waves; dynamic LDS: B * Kp floats (Kp = K rounded up to the wave stride)
template <typename WT, int MAXB, int RW>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ x, int ldx, const WT* __restrict__ w, int ldw,
                                                        const float* __restrict__ bias, float* __restrict__ out, int ldo, int B,
                                                        int N, int K, int relu, int round_x) {
    extern __shared__ __attribute__((aligned(16))) float xs[];    // [B][Kp]
    constexpr int E = WVec<WT>::N;
    const int Kp = (K + 64 * E - 1) / (64 * E) * (64 * E);
    for (int i = threadIdx.x; i < B * Kp; i += 256) {
        const int b = i / Kp, k = i - b * Kp;
        float v = k < K ? x[(size_t)b * ldx + k] : 0.f;
        if constexpr (!__is_same(WT, float)) { if (round_x) v = (float)(WT)v; }
        xs[i] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * RW;
    float acc[RW][MAXB];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int b = 0; b < MAXB; ++b) acc[r][b] = 0.f;
    for (int k = lane * E; k < Kp; k += 64 * E) {
        float wv[RW][8];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int n = min(n0 + r, N - 1);
            if (k < K) WVec<WT>::load(w + (size_t)n * ldw + k, wv[r]);       // ldw >= K rounded up to E: rows are padded with zeros
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) wv[r][e] = 0.f;
            }
        }
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            if (b < B) {
                float xv[8];
#pragma unroll
                for (int q = 0; q < E / 4; ++q) {
                    const gf32x4 t = *reinterpret_cast<const gf32x4*>(xs + (size_t)b * Kp + k + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[4 * q + e] = t[e];
                }
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[r][b] = __builtin_fmaf(wv[r][e], xv[e], acc[r][b]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            if (b < B) {
                float v = acc[r][b];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                if (lane == 0 && n0 + r < N) {
                    v += bias ? bias[n0 + r] : 0.f;
                    if (relu) v = fmaxf(v, 0.f);
                    out[(size_t)b * ldo + n0 + r] = v;
                }
            }
        }
}

bool gemv_rows_supported(int B, int K, int w_bf16) { return B >= 1 && B <= 16 && (size_t)B * (K + 512) * 4 <= 160 * 1024 - 1024 && K % (w_bf16 ? 8 : 4) == 0; }

void gemv_rows(const float* x, int ldx, const void* w, int ldw, int w_bf16, const float* bias, float* out, int ldo, int B, int N, int K,
               int relu, hipStream_t s) {
    constexpr int RW = 2;
    const int E = w_bf16 ? 8 : 4;
    const int Kp = (K + 64 * E - 1) / (64 * E) * (64 * E);
    const size_t smem = (size_t)B * Kp * sizeof(float);
    const dim3 grid((N + 4 * RW - 1) / (4 * RW));
    ProfScope ps("gemv_weight_stream", 2.0 * B * N * K, (w_bf16 ? 2.0 : 4.0) * (double)N * K + 4.0 * B * (K + N), s);
    if (w_bf16) {
        h16_dispatch(w_bf16, [&](auto h16_tag) {
            using H = decltype(h16_tag);
            auto k4 = gemv_rows_kernel<H, 4, RW>;
            auto k16 = gemv_rows_kernel<H, 16, RW>;
            E2V_KATTR(k4, 160 * 1024);
            E2V_KATTR(k16, 160 * 1024);
            if (B <= 4) E2V_KLAUNCH(k4, grid, dim3(256), smem, s, x, ldx, static_cast<const H*>(w), ldw, bias, out, ldo, B, N, K, relu, 1);
            else E2V_KLAUNCH(k16, grid, dim3(256), smem, s, x, ldx, static_cast<const H*>(w), ldw, bias, out, ldo, B, N, K, relu, 1);
        });
    } else {
        auto k4 = gemv_rows_kernel<float, 4, RW>;
        auto k16 = gemv_rows_kernel<float, 16, RW>;
        E2V_KATTR(k4, 160 * 1024);
        E2V_KATTR(k16, 160 * 1024);
        if (B <= 4) E2V_KLAUNCH(k4, grid, dim3(256), smem, s, x, ldx, static_cast<const float*>(w), ldw, bias, out, ldo, B, N, K, relu, 0);
        else E2V_KLAUNCH(k16, grid, dim3(256), smem, s, x, ldx, static_cast<const float*>(w), ldw, bias, out, ldo, B, N, K, relu, 0);
    }
}

}  // namespace e2v
