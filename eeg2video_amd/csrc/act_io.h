// Activation element access shared by the HBM-bound kernels: they compute in fp32 on groups of four channels whatever the
// storage type of the activations (fp32, or bf16 / fp16 in the 16-bit activation modes: BASELINE configs[2] / the reference's .half() run).
#pragma once
#include <hip/hip_runtime.h>

namespace e2v {

typedef float aio_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 aio_bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 aio_f16x4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ aio_f32x4 ld4(const T* p);
template <> __device__ __forceinline__ aio_f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const aio_f32x4*>(p); }
template <> __device__ __forceinline__ aio_f32x4 ld4<__bf16>(const __bf16* p) {
    return __builtin_convertvector(*reinterpret_cast<const aio_bf16x4*>(p), aio_f32x4);
}
template <> __device__ __forceinline__ aio_f32x4 ld4<_Float16>(const _Float16* p) {
    return __builtin_convertvector(*reinterpret_cast<const aio_f16x4*>(p), aio_f32x4);
}
template <typename T> __device__ __forceinline__ void st4(T* p, aio_f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, aio_f32x4 v) { *reinterpret_cast<aio_f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<__bf16>(__bf16* p, aio_f32x4 v) {
    *reinterpret_cast<aio_bf16x4*>(p) = __builtin_convertvector(v, aio_bf16x4);
}
template <> __device__ __forceinline__ void st4<_Float16>(_Float16* p, aio_f32x4 v) {
    *reinterpret_cast<aio_f16x4*>(p) = __builtin_convertvector(v, aio_f16x4);
}
template <typename T> __device__ __forceinline__ float ld1(const T* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st1(T* p, float v) { *p = (T)v; }

}  // namespace e2v
