// Element-wise and layout kernels of the path (all HBM-bound, grid-stride, 16-byte lanes where the
// layout allows): boundary layout conversion NCFHW <-> channel-last, the fused classifier-free
// guidance + DDIM update, the timestep sinusoid, SiLU, and a tiled transpose for the VAE attention.
#include "h16.h"
#include "kernels.h"
#include "prof.h"
#include "act_io.h"

namespace e2v {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int grid_for(size_t total, int cap = 8192) {
    size_t b = (total + 255) / 256;
    return (int)(b < (size_t)cap ? (b ? b : 1) : cap);
}

// in [n][C][FHW] -> out [n][FHW][Cpad] (extra channels zero), times `scale`
template <typename T>
__global__ void ncfhw_to_cl_kernel(const float* __restrict__ in, T* __restrict__ out, int n, int C, int Cpad, int FHW,
                                   float scale) {
    const size_t total = (size_t)n * FHW * Cpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % Cpad;
        const size_t r = i / Cpad;
        const int p = r % FHW;
        const int b = r / FHW;
        out[i] = (T)(c < C ? in[((size_t)b * C + c) * FHW + p] * scale : 0.f);
    }
}
void ncfhw_to_cl(const float* in, float* out, int n, int C, int Cpad, int FHW, float scale, hipStream_t s, int out_bf16) {
    const size_t total = (size_t)n * FHW * Cpad;
    if (out_bf16)
        h16_dispatch(out_bf16, [&](auto h16_tag) {
            using H = decltype(h16_tag);
            E2V_KLAUNCH(ncfhw_to_cl_kernel<H>, dim3(grid_for(total)), dim3(256), 0, s, in, reinterpret_cast<H*>(out), n, C, Cpad, FHW, scale);
        });
    else
        E2V_KLAUNCH(ncfhw_to_cl_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, in, out, n, C, Cpad, FHW, scale);
}

// in [n][FHW][ld] (first C channels) -> out [n][C][FHW]; y = x*mul + add, optional clamp to [0,1]
__global__ void cl_to_ncfhw_kernel(const float* __restrict__ in, int ld, float* __restrict__ out, int n, int C, int FHW,
                                   float mul, float add, int clamp, float lo, float hi) {
    const size_t total = (size_t)n * C * FHW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = i % FHW;
        const size_t r = i / FHW;
        const int c = r % C;
        const int b = r / C;
        float v = in[((size_t)b * FHW + p) * ld + c] * mul + add;
        if (clamp) v = fminf(fmaxf(v, lo), hi);
        out[i] = v;
    }
}
void cl_to_ncfhw(const float* in, int ld, float* out, int n, int C, int FHW, float mul, float add, int clamp, float lo,
                 float hi, hipStream_t s) {
    const size_t total = (size_t)n * C * FHW;
    E2V_KLAUNCH(cl_to_ncfhw_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, ld, out, n, C, FHW, mul, add, clamp, lo,
                       hi);
}

// frames decoded as (b f) images, channel-last [n*F][HW][ld] -> video [n][C][F][HW]
// ('(b f) c h w -> b c f h w' of pipeline_tuneeeg2video.py:180, then (v/2+0.5).clamp(0,1) of :181)
__global__ void frames_to_ncfhw_kernel(const float* __restrict__ in, int ld, float* __restrict__ out, int n, int F, int C,
                                       int HW, float mul, float add, int clamp01) {
    const size_t total = (size_t)n * C * F * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = i % HW;
        size_t r = i / HW;
        const int f = r % F; r /= F;
        const int c = r % C;
        const int b = r / C;
        float v = in[(((size_t)b * F + f) * HW + p) * ld + c] * mul + add;
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        out[i] = v;
    }
}
void nchw_frames_to_ncfhw(const float* in, int ld, float* out, int n, int F, int C, int HW, float mul, float add, int clamp01,
                          hipStream_t s) {
    const size_t total = (size_t)n * C * F * HW;
    E2V_KLAUNCH(frames_to_ncfhw_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, ld, out, n, F, C, HW, mul, add,
                       clamp01);
}

// diffusers get_timestep_embedding: w_i = exp(-ln(10000) i / (half - shift)),
// emb = [sin(t w), cos(t w)], swapped to [cos, sin] when flip_sin_to_cos.  t has nt entries (1 = broadcast).
template <typename TT>
__global__ void timestep_sinusoid_kernel(const TT* __restrict__ t, int nt, float* __restrict__ out, int n, int dim,
                                         int flip, float shift) {
    const int half = dim / 2;
    const int total = n * half;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / half, k = i - b * half;
        const float tv = (float)t[nt == 1 ? 0 : b];
        const float expo = (-9.210340371976184f * (float)k) / ((float)half - shift);
        const float arg = tv * expf(expo);
        const float sv = sinf(arg), cv = cosf(arg);
        float* o = out + (size_t)b * dim;
        if (flip) { o[k] = cv; o[half + k] = sv; } else { o[k] = sv; o[half + k] = cv; }
    }
}
void timestep_sinusoid(const long long* t, int nt, float* out, int n, int dim, int flip, float shift, hipStream_t s, int t_is_f32) {
    if (t_is_f32)       // fractional timesteps of the sigma-space schedulers (Euler, LMS): the fp32 value the reference's .float() makes
        E2V_KLAUNCH(timestep_sinusoid_kernel<float>, dim3(grid_for((size_t)n * dim / 2)), dim3(256), 0, s,
                           reinterpret_cast<const float*>(t), nt, out, n, dim, flip, shift);
    else
        E2V_KLAUNCH(timestep_sinusoid_kernel<long long>, dim3(grid_for((size_t)n * dim / 2)), dim3(256), 0, s, t, nt, out, n, dim,
                           flip, shift);
}

__global__ void silu_kernel(const float* __restrict__ in, float* __restrict__ out, size_t count) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const float v = in[i];
        out[i] = v / (1.0f + expf(-v));
    }
}
void silu(const float* in, float* out, long long count, hipStream_t s) {
    if (count <= 0) return;
    E2V_KLAUNCH(silu_kernel, dim3(grid_for((size_t)count)), dim3(256), 0, s, in, out, (size_t)count);
}

// out[b][c][r] = in[b][r][c], 32x32 LDS tiles
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, int ld_in, T* __restrict__ out,
                                                        int ld_out, int rows, int cols, long long sb_in, long long sb_out) {
    __shared__ T tile[32][33];
    const T* src = in + (size_t)blockIdx.z * sb_in;
    T* dst = out + (size_t)blockIdx.z * sb_out;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < rows && c < cols) ? src[(size_t)r * ld_in + c] : (T)0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < cols && r < rows) dst[(size_t)c * ld_out + r] = tile[tx][k];
    }
}
void transpose2d(const float* in, int ld_in, float* out, int ld_out, int rows, int cols, int batch, long long sb_in,
                 long long sb_out, hipStream_t s, int bf16) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch);
    if (bf16)            // (a 2-byte payload: one instance serves bf16 and fp16)
        E2V_KLAUNCH(transpose_kernel<unsigned short>, grid, dim3(256), 0, s, reinterpret_cast<const unsigned short*>(in), ld_in,
                           reinterpret_cast<unsigned short*>(out), ld_out, rows, cols, sb_in, sb_out);
    else
        E2V_KLAUNCH(transpose_kernel<float>, grid, dim3(256), 0, s, in, ld_in, out, ld_out, rows, cols, sb_in, sb_out);
}

// pipeline_tuneeeg2video.py:320-325 fused: guidance, then DDIM (eta = 0):
//   x0 = (x - sqrt(1-a_t) eps) / sqrt(a_t);  x' = sqrt(a_p) x0 + sqrt(1-a_p) eps
// eps_c == nullptr: no guidance (guidance_scale <= 1), eps = eps_u.
__global__ void ddim_cfg_step_kernel(const float* __restrict__ eu, const float* __restrict__ ec, const float* __restrict__ x,
                                     float* __restrict__ xo, size_t count, float g, float sa, float sb, float sap, float sbp) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        float e = eu[i];
        if (ec) e = e + g * (ec[i] - e);
        const float x0 = (x[i] - sb * e) / sa;
        xo[i] = sap * x0 + sbp * e;
    }
}
void ddim_cfg_step(const float* eps_u, const float* eps_c, const float* x, float* x_out, long long count, float guidance,
                   float sqrt_a_t, float sqrt_1m_a_t, float sqrt_a_p, float sqrt_1m_a_p, hipStream_t s) {
    if (count <= 0) return;
    ProfScope ps("ddim_cfg_step", 8.0 * count, 4.0 * count * (eps_c ? 4.0 : 3.0), s);
    E2V_KLAUNCH(ddim_cfg_step_kernel, dim3(grid_for((size_t)count)), dim3(256), 0, s, eps_u, eps_c, x, x_out,
                       (size_t)count, guidance, sqrt_a_t, sqrt_1m_a_t, sqrt_a_p, sqrt_1m_a_p);
}

// out = sum_i c_i x_i over up to five tensors (the linear multistep combinations of PNDM/PLMS and its sample update);
// cfg (ec != nullptr): out = eu + g (ec - eu), the guidance line of the pipeline (pipeline_tuneeeg2video.py:320-322)
struct LinComb { const float* x[5]; float c[5]; int n; };
__global__ void lincomb_kernel(const LinComb a, float* __restrict__ out, size_t count) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        float v = a.c[0] * a.x[0][i];
#pragma unroll
        for (int k = 1; k < 5; ++k)
            if (k < a.n) v += a.c[k] * a.x[k][i];
        out[i] = v;
    }
}
void lincomb(int n, const float* const* xs, const float* coefs, float* out, long long count, hipStream_t s) {
    if (count <= 0) return;
    LinComb a{};
    a.n = n;
    for (int k = 0; k < n; ++k) { a.x[k] = xs[k]; a.c[k] = coefs[k]; }
    ProfScope ps("lincomb", 2.0 * n * count, 4.0 * count * (n + 1), s);
    E2V_KLAUNCH(lincomb_kernel, dim3(grid_for((size_t)count)), dim3(256), 0, s, a, out, (size_t)count);
}
__global__ void cfg_combine_kernel(const float* __restrict__ eu, const float* __restrict__ ec, float g, float* __restrict__ out,
                                   size_t count) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const float e = eu[i];
        out[i] = e + g * (ec[i] - e);
    }
}
void cfg_combine(const float* eu, const float* ec, float g, float* out, long long count, hipStream_t s) {
    if (count <= 0) return;
    ProfScope ps("cfg_combine", 3.0 * count, 12.0 * count, s);
    E2V_KLAUNCH(cfg_combine_kernel, dim3(grid_for((size_t)count)), dim3(256), 0, s, eu, ec, g, out, (size_t)count);
}

// ---- SURVEY 8(f) rows ---------------------------------------------------------------------------------------
__global__ void dana_kernel(const float* __restrict__ x0, const float* __restrict__ ed, const float* __restrict__ es,
                            const float* __restrict__ coef, float s1b, float sb, float* __restrict__ out, int B, int F, int C,
                            int HW) {
    const size_t total = (size_t)B * C * F * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = i % HW;
        size_t r = i / HW;
        const int f = r % F; r /= F;
        const int c = r % C;
        const int b = r / C;
        const size_t src = (((size_t)b * F + f) * C + c) * HW + p;          // [B,F,C,HW]
        const size_t same = ((size_t)b * C + c) * HW + p;                   // [B,1,C,HW]
        const float noise = ed[src] * s1b + es[same] * sb;                  // DANA_module.py:60-61
        out[i] = coef[2 * b] * x0[src] + coef[2 * b + 1] * noise;           // :71-72
    }
}
void dana_noise(const float* x0, const float* eps_div, const float* eps_same, const float* coef, float sqrt_1m_beta,
                float sqrt_beta, float* out, int B, int F, int C, int HW, hipStream_t s) {
    const size_t total = (size_t)B * C * F * HW;
    if (!total) return;
    ProfScope ps("dana_noise", 6.0 * total, 4.0 * total * 3.2, s);
    E2V_KLAUNCH(dana_kernel, dim3(grid_for(total)), dim3(256), 0, s, x0, eps_div, eps_same, coef, sqrt_1m_beta, sqrt_beta,
                       out, B, F, C, HW);
}

__global__ void frames_to_u8_kernel(const float* __restrict__ in, unsigned char* __restrict__ out, size_t count) {
    const size_t n4 = count / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + 4 * i);
        uchar4 o;
        o.x = (unsigned char)(v[0] * 255.f); o.y = (unsigned char)(v[1] * 255.f);
        o.z = (unsigned char)(v[2] * 255.f); o.w = (unsigned char)(v[3] * 255.f);
        *reinterpret_cast<uchar4*>(out + 4 * i) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = n4 * 4; i < count; ++i) out[i] = (unsigned char)(in[i] * 255.f);
}
void frames_to_u8(const float* in, unsigned char* out, long long count, hipStream_t s) {
    if (count <= 0) return;
    ProfScope ps("frames_to_u8", 1.0 * count, 5.0 * count, s);
    E2V_KLAUNCH(frames_to_u8_kernel, dim3(grid_for((size_t)count / 4 + 1)), dim3(256), 0, s, in, out, (size_t)count);
}

template <typename T>
__global__ void pad_cols_kernel(const float* __restrict__ in, int cols, T* __restrict__ out, int cp, size_t rows) {
    const size_t total = rows * cp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % cp;
        const size_t r = i / cp;
        out[i] = (T)(c < cols ? in[r * cols + c] : 0.f);
    }
}
void pad_cols(const float* in, int cols, float* out, int cols_pad, long long rows, hipStream_t s, int out_bf16) {
    if (rows <= 0) return;
    if (out_bf16)
        h16_dispatch(out_bf16, [&](auto h16_tag) {
            using H = decltype(h16_tag);
            E2V_KLAUNCH(pad_cols_kernel<H>, dim3(grid_for((size_t)rows * cols_pad)), dim3(256), 0, s, in, cols, reinterpret_cast<H*>(out), cols_pad, (size_t)rows);
        });
    else
        E2V_KLAUNCH(pad_cols_kernel<float>, dim3(grid_for((size_t)rows * cols_pad)), dim3(256), 0, s, in, cols, out, cols_pad, (size_t)rows);
}

// strided row copy with a storage-type change: out[r][c] = in[r][c] for c < cols, 0 for cols <= c < cols_out
template <typename TI, typename TO>
__global__ void cvt_rows_kernel(const TI* __restrict__ in, int ld_in, TO* __restrict__ out, int ld_out, size_t rows, int cols, int cols_out) {
    const size_t total = rows * cols_out;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % cols_out;
        const size_t r = i / cols_out;
        out[r * ld_out + c] = c < cols ? (TO)(float)in[r * ld_in + c] : (TO)0.f;
    }
}
void cvt_rows(const void* in, int ld_in, int in_bf16, void* out, int ld_out, int out_bf16, long long rows, int cols, int cols_out,
              hipStream_t s) {
    if (rows <= 0 || cols_out <= 0) return;
    const dim3 g(grid_for((size_t)rows * cols_out)), b(256);
    if (in_bf16 || out_bf16) {           // (both 16-bit: the same type -- a run has one 16-bit mode)
        h16_dispatch(in_bf16 ? in_bf16 : out_bf16, [&](auto h16_tag) {
            using H = decltype(h16_tag);
            if (in_bf16 && out_bf16)
                E2V_KLAUNCH((cvt_rows_kernel<H, H>), g, b, 0, s, static_cast<const H*>(in), ld_in, static_cast<H*>(out), ld_out, (size_t)rows, cols, cols_out);
            else if (in_bf16)
                E2V_KLAUNCH((cvt_rows_kernel<H, float>), g, b, 0, s, static_cast<const H*>(in), ld_in, static_cast<float*>(out), ld_out, (size_t)rows, cols, cols_out);
            else
                E2V_KLAUNCH((cvt_rows_kernel<float, H>), g, b, 0, s, static_cast<const float*>(in), ld_in, static_cast<H*>(out), ld_out, (size_t)rows, cols, cols_out);
        });
    }
    else
        E2V_KLAUNCH((cvt_rows_kernel<float, float>), g, b, 0, s, static_cast<const float*>(in), ld_in, static_cast<float*>(out), ld_out, (size_t)rows, cols, cols_out);
}

}  // namespace e2v
