// Peak-rate probe: back-to-back MFMAs from registers (no memory), 2 waves per SIMD like the igemm kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(threadIdx.x * 3 + e); }
    float fa = (float)threadIdx.x, fb = (float)(threadIdx.x * 7);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 512 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(512), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(probe<1>, dim3(512), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 512.0 * 4 * iters * 24 * (mode == 0 ? 32.0 * 32 * 16 * 2 : 32.0 * 32 * 2 * 2);
            printf("%s: %.3f ms  %.1f TFLOP/s\n", mode == 0 ? "bf16 32x32x16" : "f32 32x32x2", ms, flops / ms / 1e9);
        }
    }
    return 0;
}
