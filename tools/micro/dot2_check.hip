// What v_dot2c_f32_bf16 (__builtin_amdgcn_fdot2_f32_bf16 on gfx950) computes, on exact small integers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
__global__ void k(float* o) {
    b2 a = {(__bf16)3.0f, (__bf16)5.0f}, b = {(__bf16)7.0f, (__bf16)11.0f};
    o[0] = __builtin_amdgcn_fdot2_f32_bf16(a, b, 100.0f, false);      // 100 + 21 + 55 = 176 expected
    unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    o[1] = __builtin_bit_cast(float, ua << 16) * __builtin_bit_cast(float, ub << 16);                    // element 0 product: 21
    o[2] = __builtin_bit_cast(float, ua & 0xFFFF0000u) * __builtin_bit_cast(float, ub & 0xFFFF0000u);    // element 1 product: 55
    b2 c = {(__bf16)1.5f, (__bf16)-2.0f}, d = {(__bf16)0.25f, (__bf16)4.0f};
    o[3] = __builtin_amdgcn_fdot2_f32_bf16(c, d, 0.0f, false);        // 0.375 - 8 = -7.625
}
int main() {
    float* d; hipMalloc(&d, 16); hipLaunchKernelGGL(k, 1, 1, 0, 0, d);
    float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("dot2: %g (176)  e0: %g (21)  e1: %g (55)  dot2b: %g (-7.625)\n", h[0], h[1], h[2], h[3]);
    return 0;
}
