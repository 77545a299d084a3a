// fp32 implicit-GEMM convolution / linear on the gfx950 matrix cores.
//
// One kernel serves InflatedConv3d 3x3 (resnet.py:10-18, stride 1/2, with the nearest resize of
// Upsample3D folded into the gather), every 1x1 conv / nn.Linear of the path, and the batched
// QK^T / PV products of the VAE attention block.  Arithmetic is v_mfma_f32_32x32x2_f32: exact
// fp32 products, fp32 accumulate (bitwise an fmaf chain) -- the parity configuration of BASELINE.json.
//
// Tiling: 256 threads = 4 waves as 2x2; block tile BM x BN x 32, LDS double-buffered, global loads
// of step k+1 issued before the MFMAs of step k (register staging, one barrier per step).
// LDS rows are padded to 36 floats: a wave's ds_read_b128 of 16 distinct rows then covers all 64 banks
// exactly once.  Each lane feeds 4 consecutive k (one b128) to 4 MFMAs: the two lane halves own
// k = 8g + {0..3} and 8g + {4..7}; A and B use the same assignment so the sum is over all 32 k.
#include "kernels.h"
#include "prof.h"

namespace e2v {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int BK = 32;
static constexpr int LDS_LD = 36;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int BM, int BN, int WGM, int WGN, int ABL = 0, int NBUF = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void igemm_kernel(const IgemmArgs p) {
    constexpr int NT = 64 * WGM * WGN;              // threads; waves are laid out WGM x WGN over the tile
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPP = NT / 8;                     // tile rows covered by one pass of the loader (8 lanes per row)
    constexpr int AR = BM / RPP, BR = BN / RPP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + NBUF * BM * LDS_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // XCD-aware tile order: the 8 XCDs are dealt blocks round-robin; give each XCD a contiguous run of
    // tiles (n fastest) so that the column blocks sharing one gathered A tile hit the same L2.
    const int nbn = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int tile;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bm = tile / nbn, bn = tile % nbn;

    const float* __restrict__ a0 = p.a0 + (size_t)blockIdx.z * p.sa0;
    const float* __restrict__ a1 = p.a1;
    const float* __restrict__ w = p.w + (size_t)blockIdx.z * p.sw;
    float* __restrict__ out = p.out + (size_t)blockIdx.z * p.sout;

    const int Ctot = p.c0 + p.c1;
    // K is walked as segments (tap, source): all channels of source 0 at tap 0, then source 1, then tap 1 ...
    // A segment is ceil(c/32) k-steps; its last step is masked to the segment's channel count.
    const int steps0 = (p.c0 + BK - 1) / BK, steps1 = (p.c1 + BK - 1) / BK;
    const int nk = p.taps * (steps0 + steps1);

    const int c4 = tid & 7;
    const int r0 = tid >> 3;

    // Per-thread A row descriptors.  Linear / 1x1 (taps == 1) is the degenerate conv with a 1 x M map
    // (the launcher sets Ho = Hi = Hs = 1, Wo = Wi = Ws = M, pad = 0), so one gather serves both.
    int a_img[AR], a_y[AR], a_x[AR];
    bool a_ok[AR];
    {
        const int hw = p.Ho * p.Wo;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int m = bm * BM + r0 + RPP * i;
            a_ok[i] = m < p.M;
            const int mm = a_ok[i] ? m : 0;
            const int img = mm / hw;
            const int rem = mm - img * hw;
            const int oy = rem / p.Wo;
            a_img[i] = img;
            a_y[i] = oy * p.stride - p.pad;
            a_x[i] = (rem - oy * p.Wo) * p.stride - p.pad;
        }
    }
    size_t b_row[BR];
    bool b_ok[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int n = bn * BN + r0 + RPP * j;
        b_ok[j] = n < p.N;
        b_row[j] = (size_t)(b_ok[j] ? n : 0) * p.ldw + c4 * 4;
    }

    // Gather state of the NEXT tile to load.  Inside a segment a k-step only bumps eight pointers by 32 floats;
    // the per-row pixel / padding arithmetic runs once per segment (every >= c/32 steps), behind a uniform branch
    // that sits AFTER the MFMAs.  Masked elements (zero padding, rows >= M, channels >= c) read 16 zero bytes
    // through a pointer select made BEFORE the load, so the loaded registers are not touched until the LDS store
    // and the loads + pointer bumps share one basic block with the MFMAs of the current tile.
    const float* __restrict__ zeros = p.zeros;
    const float* pa[AR];
    const float* pb[BR];
    bool va[AR];
    int seg_tap = 0, seg_src = 0, cb = 0, cseg = p.c0;
    bool done = false;
    auto enter_segment = [&]() {
        const int ky = (seg_tap * 11) >> 5, kx = seg_tap - 3 * ky;          // tap < 9
        cseg = seg_src ? p.c1 : p.c0;
        const float* base = seg_src ? a1 : a0;
        const int ld = seg_src ? p.lda1 : p.lda0;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int iy = a_y[i] + ky, ix = a_x[i] + kx;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            int sy = ok ? iy : 0, sx = ok ? ix : 0;
            if (p.upsample) {      // torch nearest (resnet.py:58-61): src = min(floor(dst * (in/out)), in - 1), fp32 scale
                sy = min((int)floorf((float)sy * p.ups_h), p.Hs - 1);
                sx = min((int)floorf((float)sx * p.ups_w), p.Ws - 1);
            }
            const size_t pix = ((size_t)a_img[i] * p.Hs + sy) * p.Ws + sx;
            pa[i] = base + pix * ld + c4 * 4;
            va[i] = ok;
        }
        const size_t koff = (size_t)seg_tap * Ctot + (seg_src ? p.c0 : 0);
#pragma unroll
        for (int j = 0; j < BR; ++j) pb[j] = w + b_row[j] + koff;
    };
    f32x4 ra[AR], rb[BR];
    auto issue_loads = [&]() {
        const bool cok = !done && (cb + c4 * 4 < cseg);
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const float* src = (va[i] && cok) ? pa[i] : zeros;
            ra[i] = *reinterpret_cast<const f32x4*>(src);
            pa[i] += BK;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const float* src = (b_ok[j] && cok) ? pb[j] : zeros;
            rb[j] = *reinterpret_cast<const f32x4*>(src);
            pb[j] += BK;
        }
        cb += BK;
    };
    auto advance_segment = [&]() {           // uniform: every lane sees the same cb / cseg
        if (cb >= cseg) {
            cb = 0;
            if (seg_src == 0 && p.c1 > 0) {
                seg_src = 1;
            } else {
                seg_src = 0;
                ++seg_tap;
            }
            done = seg_tap >= p.taps;
            if (!done) enter_segment();
        }
    };
    auto store_tile = [&](int buf) {
        float* A = As + buf * BM * LDS_LD;
        float* B = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(A + (r0 + RPP * i) * LDS_LD + c4 * 4) = ra[i];
#pragma unroll
        for (int j = 0; j < BR; ++j) *reinterpret_cast<f32x4*>(B + (r0 + RPP * j) * LDS_LD + c4 * 4) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    enter_segment();
    issue_loads();
    advance_segment();
    store_tile(0);
    __syncthreads();

    // Fragment double-buffering across the barrier: the last k-group of tile ks is multiplied AFTER the barrier
    // that publishes tile ks+1, while the first fragment reads of tile ks+1 are in flight -- the LDS round trip
    // behind the barrier is then covered by 16 register-only MFMAs instead of stalling the matrix pipe.
    const int frag_off = (lane & 31) * LDS_LD + (lane >> 5) * 4;
    const float* Afr = As + wm * WM * LDS_LD + frag_off;
    const float* Bfr = Bs + wn * WN * LDS_LD + frag_off;
    f32x4 af[2][TM], bf[2][TN];
    auto read_frags = [&](int set, int buf, int g) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
            af[set][mi] = *reinterpret_cast<const f32x4*>(Afr + buf * BM * LDS_LD + mi * 32 * LDS_LD + g * 8);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
            bf[set][ni] = *reinterpret_cast<const f32x4*>(Bfr + buf * BN * LDS_LD + ni * 32 * LDS_LD + g * 8);
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi][s], bf[set][ni][s], acc[mi][ni], 0, 0, 0);
    };
    static_assert(BK == 32, "the k-step below is written for four k-groups of 8");
    read_frags(0, 0, 0);
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = NBUF == 2 ? (ks & 1) : 0;
        const int nbuf = NBUF == 2 ? (buf ^ 1) : 0;
        if (ABL == 0) issue_loads();            // tile ks+1 (all-zero dummy loads behind the last tile)
        // keep the eight loads at the head of the k-step (their latency then hides behind the MFMAs below);
        // left alone the scheduler sinks them to the end of the block, right in front of the waiting LDS stores
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, buf, 1);
        mma(0);
        read_frags(0, buf, 2);
        mma(1);
        read_frags(1, buf, 3);
        mma(0);
        if (ABL == 0) advance_segment();
        if (ABL != 2) {
            if (NBUF == 1) __syncthreads();      // single LDS buffer: everyone has finished reading it
            store_tile(nbuf);
            __syncthreads();
        }
        read_frags(0, nbuf, 0);                 // first fragments of tile ks+1 ...
        __builtin_amdgcn_sched_barrier(0);
        mma(1);                                 // ... land while the last k-group of tile ks is multiplied
    }

    // epilogue: C/D layout of the 32x32 MFMA -- column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int col = lane & 31;
    const int rquad = (lane >> 5) * 4;
    if (p.geglu) {
        if constexpr (TN == 2) {
            const int nv = bn * BN + wn * WN + col;
            if (nv + 32 < p.N) {
                const float bv = p.bias ? p.bias[nv] : 0.f;
                const float bg = p.bias ? p.bias[nv + 32] : 0.f;
                const int no = (bn * BN + wn * WN) / 2 + col;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = bm * BM + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + rquad;
                        if (m < p.M) {
                            const float v = acc[mi][0][r] * p.alpha + bv;
                            const float g = acc[mi][1][r] * p.alpha + bg;
                            out[(size_t)m * p.ldc + no] = v * gelu_erf(g);
                        }
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int n = bn * BN + wn * WN + ni * 32 + col;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm * BM + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + rquad;
                if (m < p.M) {
                    float v = acc[mi][ni][r] * p.alpha + bv;
                    if (p.rowbias) v += p.rowbias[(size_t)(m / p.rows_per_sample) * p.rb_ld + n];
                    if (p.resid) v += p.resid[(size_t)m * p.ldr + n];
                    out[(size_t)m * p.ldc + n] = v;
                }
            }
    }
}

template <int BM, int BN, int WGM = 2, int WGN = 2, int ABL = 0, int NBUF = 2>
static void launch_igemm(const IgemmArgs& a, hipStream_t s) {
    static bool configured = false;
    constexpr size_t smem = (size_t)NBUF * (BM + BN) * LDS_LD * sizeof(float);
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, WGM, WGN, ABL, NBUF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        configured = true;
    }
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    dim3 grid(nbm * nbn, 1, a.batch);
    const double K = (double)a.taps * (a.c0 + a.c1);
    const double rows_in = a.taps == 1 ? (double)a.M : (double)a.M * a.Hs * a.Ws / ((double)a.Ho * a.Wo);
    ProfScope ps(BN == 128 ? "igemm_f32_128x128" : "igemm_f32_128x64", 2.0 * a.M * a.N * K * a.batch,
                 4.0 * a.batch * (rows_in * (a.c0 + a.c1) + (double)a.N * K + (double)a.M * (a.geglu ? a.N / 2 : a.N)), s);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WGM, WGN, ABL, NBUF>), grid, dim3(64 * WGM * WGN), smem, s, a);
}

// 256 zero bytes per device: the source of every masked 16-byte load
static const float* zero_page() {
    static float* z[64] = {nullptr};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!z[dev]) {
        (void)hipMalloc((void**)&z[dev], 256);
        (void)hipMemset(z[dev], 0, 256);
    }
    return z[dev];
}

void igemm(const IgemmArgs& a_in, hipStream_t s) {
    if (a_in.M <= 0 || a_in.N <= 0) return;
    IgemmArgs a = a_in;
    a.zeros = zero_page();
    if (a.taps == 1) {      // linear / 1x1: a 1 x M map without padding -- the same gather as the 3x3 case
        a.Ho = a.Hi = a.Hs = 1;
        a.Wo = a.Wi = a.Ws = a.M;
        a.stride = 1; a.pad = 0; a.upsample = 0;
    }
    static const int abl = [] { const char* e = std::getenv("E2V_IGEMM_ABLATE"); return e ? std::atoi(e) : 0; }();
    static const int cfg = [] { const char* e = std::getenv("E2V_IGEMM_CFG"); return e ? std::atoi(e) : 0; }();
    // 128x64 only where a 128-wide tile would waste > 12 % of its columns (N = 320: 384 vs 320)
    const long n128 = (a.N + 127) / 128 * 128;
    const bool narrow = !a.geglu && (double)(n128 - a.N) > 0.12 * (double)n128;
    if (abl == 1) {      // timing experiment only (wrong results): no global loads inside the k-loop
        if (!narrow) launch_igemm<128, 128, 2, 2, 1>(a, s); else launch_igemm<128, 64, 2, 2, 1>(a, s);
        return;
    }
    if (abl == 2) {      // timing experiment only: no loads, no LDS stores, no barriers (MFMA + fragment reads)
        if (!narrow) launch_igemm<128, 128, 2, 2, 2>(a, s); else launch_igemm<128, 64, 2, 2, 2>(a, s);
        return;
    }
    if (narrow) { launch_igemm<128, 64, 2, 2>(a, s); return; }
    switch (cfg) {       // experiments: alternative wave layouts of the wide tile
        case 2: launch_igemm<128, 128, 1, 2>(a, s); break;
        case 3: launch_igemm<256, 128, 2, 2>(a, s); break;
        case 4: launch_igemm<256, 128, 4, 2>(a, s); break;
        case 5: launch_igemm<128, 128, 2, 2, 0, 1>(a, s); break;
        default: launch_igemm<128, 128, 2, 2>(a, s); break;
    }
}

// ---- one-off weight re-layout ---------------------------------------------------------------------
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ o, int cout, int cin, int cin_pad) {
    const size_t total = (size_t)cout * 9 * cin_pad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % cin_pad;
        const int tap = (i / cin_pad) % 9;
        const int oc = i / ((size_t)cin_pad * 9);
        o[i] = (c < cin) ? w[((size_t)oc * cin + c) * 9 + tap] : 0.f;
    }
}
void pack_conv3x3(const float* w, float* o, int cout, int cin, int cin_pad, hipStream_t s) {
    const size_t total = (size_t)cout * 9 * cin_pad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(blocks), dim3(256), 0, s, w, o, cout, cin, cin_pad);
}

__global__ void copy_rows_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int rows,
                                 int cols) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = i / cols, c = i % cols;
        dst[(size_t)r * ldd + c] = src[(size_t)r * lds + c];
    }
}
void copy_rows(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s) {
    const size_t total = (size_t)rows * cols;
    if (!total) return;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(blocks), dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
}

}  // namespace e2v
