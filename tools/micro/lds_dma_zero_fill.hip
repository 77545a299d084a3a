#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// each wave: 64 lanes x 16 B -> 1 KB of LDS at a wave-uniform base; lanes with OOB offset must produce zeros
__global__ void k(const float* src, float* out, unsigned nbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)0, nbytes, 0x00020000);
    unsigned off = (wave * 64 + lane) * 16;
    if (lane % 5 == 3) off = 0x80000000u;          // out of range: expect zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, off, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 4096 / 4; i += 256) out[i] = reinterpret_cast<float*>(smem)[i];
}
int main() {
    std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = i + 1;
    float *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    k<<<1, 256, 4096>>>(d, o, 4096);
    std::vector<float> r(1024); hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) { int lane = (i / 4) % 64; float want = (lane % 5 == 3) ? 0.f : (float)(i + 1); if (r[i] != want) { if (bad < 8) printf("i %d got %f want %f\n", i, r[i], want); ++bad; } }
    printf("dma test: %d mismatches (OOB lanes: r[12]=%f)\n", bad, r[12]);
    return bad != 0;
}
