// Probe: the inner loop of the f32x3 tile without global memory -- 12 ds_read_b128 fragments (3 planes x (2 + 2) tiles)
// per 24 bf16 MFMAs, 2 workgroups of 4 waves per CU.  MODE 0: read, wait, multiply.  MODE 1: fragments double-buffered.
// MODE 2: MODE 0 per 32-k stage + barrier + split-and-store of a tile (VALU + 18 LDS writes) + barrier, like igemm_tile_x3.
// MODE 3: as 2 but with the store placed in the middle of the MFMAs (no second barrier).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    constexpr int PLD = 80, BM = 128;
    for (int i = threadIdx.x; i < 61440 / 4; i += 256) reinterpret_cast<float*>(sm)[i] = 1.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* Afr = sm + ((wave >> 1) * 64 + (lane & 31)) * PLD + (lane >> 5) * 16;
    const char* Bfr = sm + 3 * BM * PLD + ((wave & 1) * 64 + (lane & 31)) * PLD + (lane >> 5) * 16;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    bf16x8 af[2][3][2], bf[2][3][2];
    auto rd = [&](int set, int g) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[set][pl][t] = *reinterpret_cast<const bf16x8*>(Afr + pl * BM * PLD + t * 32 * PLD + g * 32);
                bf[set][pl][t] = *reinterpret_cast<const bf16x8*>(Bfr + pl * BM * PLD + t * 32 * PLD + g * 32);
            }
    };
    auto mm = [&](int set) {
        constexpr int TW[6] = {1, 2, 0, 1, 0, 0}, TA[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[set][TW[t]][b], af[set][TA[t]][a], acc[a][b], 0, 0, 0);
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 ra[4], rb[6];
    for (int i = 0; i < 4; ++i) ra[i] = f32x4{1.5f + lane, 2.5f, 3.5f, 4.5f};
    for (int i = 0; i < 6; ++i) rb[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h1[4], h2[4], h3[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = ra[i][e];
                const unsigned b1 = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
                const float r1 = x - __builtin_bit_cast(float, b1);
                const unsigned b2 = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
                const float r2 = r1 - __builtin_bit_cast(float, b2);
                h1[e] = b1; h2[e] = b2; h3[e] = __builtin_bit_cast(unsigned, r2);
            }
            char* dst = sm + (r0 + 32 * i) * PLD + c4 * 8;
            *reinterpret_cast<u32x2*>(dst) = u32x2{(h1[0] >> 16) | (h1[1] & 0xFFFF0000u), (h1[2] >> 16) | (h1[3] & 0xFFFF0000u)};
            *reinterpret_cast<u32x2*>(dst + BM * PLD) = u32x2{(h2[0] >> 16) | (h2[1] & 0xFFFF0000u), (h2[2] >> 16) | (h2[3] & 0xFFFF0000u)};
            *reinterpret_cast<u32x2*>(dst + 2 * BM * PLD) = u32x2{(h3[0] >> 16) | (h3[1] & 0xFFFF0000u), (h3[2] >> 16) | (h3[3] & 0xFFFF0000u)};
            ra[i][0] += 1.0f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = threadIdx.x + 256 * j;
            char* dst = sm + 3 * BM * PLD + (idx >> 2) * PLD + (idx & 3) * 16;
            *reinterpret_cast<f32x4*>(dst) = rb[3 * j];
            *reinterpret_cast<f32x4*>(dst + BM * PLD) = rb[3 * j + 1];
            *reinterpret_cast<f32x4*>(dst + 2 * BM * PLD) = rb[3 * j + 2];
        }
    };
    if (MODE == 2) {
        for (int i = 0; i < iters; i += 2) { rd(0, 0); mm(0); rd(0, 1); mm(0); __syncthreads(); store_tile(); __syncthreads(); }
    } else if (MODE == 3) {
        for (int i = 0; i < iters; i += 2) { rd(0, 0); mm(0); store_tile(); rd(0, 1); mm(0); __syncthreads(); }
    } else if (MODE == 0) {
        for (int i = 0; i < iters; ++i) { rd(0, i & 1); mm(0); }
    } else {
        rd(0, 0);
        for (int i = 0; i < iters; i += 2) { rd(1, 1); mm(0); rd(0, 0); mm(1); }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; (void)hipMalloc(&out, 512 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 61440);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 61440);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 61440);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 61440);
    for (int mode = 0; mode < 4; ++mode) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(512), dim3(256), 61440, 0, out, iters);
            else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(512), dim3(256), 61440, 0, out, iters);
            else if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(512), dim3(256), 61440, 0, out, iters);
            else hipLaunchKernelGGL(probe<3>, dim3(512), dim3(256), 61440, 0, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double flops = 512.0 * 4 * iters * 24 * 32.0 * 32 * 16 * 2;
            printf("mode %d: %.3f ms  %.1f TFLOP/s bf16 = %.1f fp32-equivalent\n", mode, ms, flops / ms / 1e9, flops / ms / 1e9 / 6);
        }
    }
    return 0;
}
