// bf16 against fp16 on the matrix pipe: back-to-back v_mfma_f32_32x32x16 from registers (no memory), 2 waves per SIMD, operands with
// realistic (random-normal-like) bit patterns -- the two opcodes have the same issue rate, so a difference in sustained TFLOP/s is the
// clock the chip holds under each datapath's power.  Alternates the two types five times (thermal drift shows as a trend, not a gap).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_rate16.hip -o tools/micro/mfma_rate16 && tools/micro/mfma_rate16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <typename H> using hx8 = H __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mma(hx8<__bf16> a, hx8<__bf16> b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mma(hx8<_Float16> a, hx8<_Float16> b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <typename H>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    hx8<H> a, b;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int e = 0; e < 8; ++e) {                           // pseudo-random values in (-2, 2): every mantissa bit toggles
        s = s * 1664525u + 1013904223u; a[e] = (H)(((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 22)));
        s = s * 1664525u + 1013904223u; b[e] = (H)(((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 22)));
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mma(a, b, acc[t]);
    }
    float r = 0.f;
    for (int t = 0; t < 4; ++t) for (int q = 0; q < 16; ++q) r += acc[t][q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
    float* out; (void)hipMalloc(&out, 512 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 60000;                                // ~0.3 s per launch: long enough for the power management to settle
    const double flops = 512.0 * 4 * iters * 24 * 32.0 * 32 * 16 * 2;
    for (int rep = 0; rep < 5; ++rep)
        for (int mode = 0; mode < 2; ++mode) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<__bf16>, dim3(512), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(probe<_Float16>, dim3(512), dim3(256), 0, 0, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s 32x32x16: %8.3f ms  %7.1f TFLOP/s\n", mode == 0 ? "bf16" : "fp16", ms, flops / ms / 1e9);
        }
    return 0;
}
