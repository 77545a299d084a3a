// ds_read_b64_tr_b16 lane map as flash_attn_b16io_kernel uses it: a 16-lane group reads a block of 4 rows x 16 columns of a
// row-major bf16 image and receives it column-major (lane i: column i, rows 0..3 in elements 0..3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, float* out, int ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* sm = reinterpret_cast<__bf16*>(smem);
    for (int i = threadIdx.x; i < 32 * ld; i += 64) sm[i] = (__bf16)in[i];
    __syncthreads();
    const int lane = threadIdx.x, h = lane >> 5, ti = lane & 15;
    const int off = (4 * h + (ti >> 2)) * ld + 16 * ((lane >> 4) & 1) + 4 * (ti & 3);
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sm + off));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (float)v[e];
}
int main() {
    const int ld = 72;                       // 64 columns + 8 pad (rows 8-byte aligned)
    std::vector<float> h(32 * ld);
    for (int r = 0; r < 32; ++r) for (int c = 0; c < ld; ++c) h[r * ld + c] = (float)(r * 64 + c % 64);   // exact in bf16 up to 256... keep small
    for (auto& x : h) x = (float)((int)x % 251);
    float *d, *o; (void)hipMalloc(&d, h.size() * 4); (void)hipMalloc(&o, 256 * 4);
    (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k<<<1, 64, 32 * ld * 2>>>(d, o, ld);
    std::vector<float> r(256); (void)hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
        const int hh = lane >> 5, col = lane & 31, row = 4 * hh + e;         // lane (dv = lane & 31, half hh) wants keys 4 hh + e
        const float want = h[row * ld + col];
        if (r[lane * 4 + e] != want) { if (bad < 8) printf("lane %d e %d got %f want %f\n", lane, e, r[lane * 4 + e], want); ++bad; }
    }
    printf("tr_read_map: %d mismatches\n", bad);
    return bad != 0;
}
