// The two 16-bit storage / MFMA-operand types of the library.
//
// The reduced-precision mode stores every activation as a 16-bit float and multiplies on the 16-bit matrix pipe with fp32 accumulation.
// WHICH 16-bit float is a property of the run (e2v_set_compute_dtype): bf16 (E2V_BF16 -- BASELINE configs[2]) or IEEE half (E2V_FP16 --
// the reference's own inference dtype, inference_eeg2video.py:69-70 `torch_dtype=torch.float16`, pipeline_tuneeeg2video.py:150).  The
// kernels are the same code: LDS-DMA, fragment reads and the epilogues move 16-bit payloads without looking at them; what differs is the
// MFMA opcode (v_mfma_f32_32x32x16_{bf16,f16} / 16x16x32: same rate, same register layout), the fp32 <-> 16-bit conversions (the
// compiler's casts: v_cvt_pk_bf16_f32 / v_cvt_f16_f32, both round to nearest even) and a few bit patterns (the constant 1.0 written
// into LDS rows).  So every 16-bit kernel is a template over its element type H and the launchers pick the instance from the mode flag
// the argument structs carry (IgemmArgs::a_bf16, AttnArgs::io_bf16, GroupNormArgs::bf16, ...: 0 = fp32 rows, 1 = bf16, 2 = fp16).
#pragma once
#include <hip/hip_runtime.h>

namespace e2v {

typedef _Float16 f16;
enum { H16_NONE = 0, H16_BF16 = 1, H16_FP16 = 2 };          // values of the 16-bit mode flags

template <typename H> using hx2 = H __attribute__((ext_vector_type(2)));
template <typename H> using hx4 = H __attribute__((ext_vector_type(4)));
template <typename H> using hx8 = H __attribute__((ext_vector_type(8)));

typedef float h16_f32x4 __attribute__((ext_vector_type(4)));
typedef float h16_f32x16 __attribute__((ext_vector_type(16)));

template <typename H> struct H16Traits;
template <> struct H16Traits<__bf16> {
    static constexpr int mode = H16_BF16;
    static constexpr unsigned short one_bits = 0x3F80;       // 1.0
    static constexpr const char* name = "bf16";
    // attn_q64.hip: a key past its segment's end is masked by the matrix pipe -- its K row's marker column times the query's mask slot
    // must drown any real score whatever the row's reference maximum: 2^60 x -2^60 (bf16 has fp32's exponent range)
    static constexpr float mask_marker = 1152921504606846976.0f;     // 2^60
    static constexpr unsigned short mask_marker_bits = 0x5D80;
    // the running maximum rides in two Q slots against marker columns holding `max_marker`
    static constexpr float max_marker = 1.0f;
    static constexpr unsigned short max_marker_bits = 0x3F80;
};
template <> struct H16Traits<_Float16> {
    static constexpr int mode = H16_FP16;
    static constexpr unsigned short one_bits = 0x3C00;
    static constexpr const char* name = "fp16";
    // fp16 tops out at 65504: 2^15 x -2^15 = -2^30, far below any score whose fp16 run in the reference stays finite
    static constexpr float mask_marker = 32768.0f;
    static constexpr unsigned short mask_marker_bits = 0x7800;
    // scores are kept in log2 units (x 1.4427): a reference-side score of 6e4 is 8.7e4 here, past fp16's range for the -m that rides in
    // Q -- the marker columns hold 2.0 and the slots -m / 2 (exact: a power of two)
    static constexpr float max_marker = 2.0f;
    static constexpr unsigned short max_marker_bits = 0x4000;
};

// the bit patterns above are what the compiler's own conversions produce (checked where it can be: at compile time)
static_assert(__builtin_bit_cast(unsigned short, (_Float16)1.0f) == H16Traits<_Float16>::one_bits &&
              __builtin_bit_cast(unsigned short, (_Float16)2.0f) == H16Traits<_Float16>::max_marker_bits &&
              __builtin_bit_cast(unsigned short, (_Float16)32768.0f) == H16Traits<_Float16>::mask_marker_bits, "fp16 constants");
static_assert((__builtin_bit_cast(unsigned, 1.0f) >> 16) == H16Traits<__bf16>::one_bits &&
              (__builtin_bit_cast(unsigned, 1.0f) >> 16) == H16Traits<__bf16>::max_marker_bits &&
              (__builtin_bit_cast(unsigned, 1152921504606846976.0f) >> 16) == H16Traits<__bf16>::mask_marker_bits, "bf16 constants (the high half of the fp32 pattern)");

// one MFMA step of the two tile shapes the kernels use, by operand type
__device__ __forceinline__ h16_f32x16 mfma_32x32x16(const hx8<__bf16> a, const hx8<__bf16> b, const h16_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ h16_f32x16 mfma_32x32x16(const hx8<_Float16> a, const hx8<_Float16> b, const h16_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ h16_f32x4 mfma_16x16x32(const hx8<__bf16> a, const hx8<__bf16> b, const h16_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ h16_f32x4 mfma_16x16x32(const hx8<_Float16> a, const hx8<_Float16> b, const h16_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// transposing LDS read of a 4 x 4 block of 16-bit elements (ds_read_b64_tr_b16)
typedef __attribute__((address_space(3))) hx4<__bf16>* lds_bf16x4_ptr;
typedef __fp16 lds_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));     // the builtin's own vector type (storage-only half)
typedef __attribute__((address_space(3))) lds_fp16x4* lds_f16x4_ptr;
template <typename H> __device__ __forceinline__ hx4<H> lds_read_tr16(const void* p);
template <> __device__ __forceinline__ hx4<__bf16> lds_read_tr16<__bf16>(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
}
template <> __device__ __forceinline__ hx4<_Float16> lds_read_tr16<_Float16>(const void* p) {
    return __builtin_bit_cast(hx4<_Float16>, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4_ptr)(p)));
}

// the two elements of a packed pair as fp32 (temporal attention works on raw dwords): bf16 IS the high half of the float
template <typename H> __device__ __forceinline__ float h16_unpack_lo(unsigned u);
template <typename H> __device__ __forceinline__ float h16_unpack_hi(unsigned u);
template <> __device__ __forceinline__ float h16_unpack_lo<__bf16>(unsigned u) { return __builtin_bit_cast(float, u << 16); }
template <> __device__ __forceinline__ float h16_unpack_hi<__bf16>(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
template <> __device__ __forceinline__ float h16_unpack_lo<_Float16>(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)u); }
template <> __device__ __forceinline__ float h16_unpack_hi<_Float16>(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }

// host side: run `f(H{})` with the element type of a 16-bit mode flag (1: bf16, 2: fp16)
template <class F>
inline void h16_dispatch(const int mode, F&& f) {
    if (mode == H16_FP16) f(_Float16{});
    else f(__bf16{});
}

}  // namespace e2v
