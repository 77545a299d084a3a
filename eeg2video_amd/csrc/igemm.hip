// fp32 implicit-GEMM convolution / linear on the gfx950 matrix cores.
//
// One kernel serves InflatedConv3d 3x3 (resnet.py:10-18, stride 1/2, with the nearest resize of
// Upsample3D folded into the gather), every 1x1 conv / nn.Linear of the path, the 16 / 36 batched
// GEMMs of the Winograd convs (wino.hip) and the batched QK^T / PV products of the VAE attention block.
// Arithmetic is v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (bitwise an fmaf chain) --
// the parity configuration of BASELINE.json.
//
// Tiling: 256 threads = 4 waves as 2x2; block tile BM x BN x 32, LDS double-buffered; global loads
// (buffer loads driven by an LDS gather table, hardware zero-fill for masked elements) run TWO k-steps
// ahead through two register sets; one LDS-only-relevant barrier per step; 16-byte epilogue accesses
// because the weight fragment is the MFMA's A operand.  DESIGN.md 3.1 has the measurements behind each choice.
// LDS rows are padded to 36 floats: a wave's ds_read_b128 of 16 distinct rows then covers all 64 banks
// exactly once.  Each lane feeds 4 consecutive k (one b128) to 4 MFMAs: the two lane halves own
// k = 8g + {0..3} and 8g + {4..7}; A and B use the same assignment so the sum is over all 32 k.
#include "kernels.h"
#include "prof.h"
#include "igemm_epi.h"
#include "runtime.h"

namespace e2v {

static constexpr int BK = 32;
static constexpr int LDS_LD = 36;

// fp32 operands, v_mfma_f32_32x32x2_f32, 32 k per stage (the parity configuration).  (The bf16-activation mode has its own
// kernels: bgemm.hip.)
template <int BM, int BN, int WGM, int WGN, int ABL, int NBUF>
__device__ __forceinline__ void igemm_tile(const IgemmArgs& p, const int rbg, const int n0, float* smem) {
    constexpr bool BF = false;
    // rbg counts row blocks over all batch entries (batch-major): entry z, row block bm within it
    const int z = p.batch > 1 ? rbg / p.nbm_per : 0;
    const int bm = rbg - z * p.nbm_per;
    constexpr int BKE = BF ? 64 : 32;               // k elements per stage
    constexpr int KH = BF ? 2 : 1;                  // 32-float pieces of an A row per stage
    constexpr int ESZ = BF ? 2 : 4;                 // bytes per weight element
    constexpr int NT = 64 * WGM * WGN;              // threads; waves are laid out WGM x WGN over the tile
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPP = NT / 8;                     // tile rows covered by one pass of the loader (8 lanes per row)
    constexpr int AR = BM / RPP, BR = BN / RPP;
    float* As = smem;
    float* Bs = smem + NBUF * BM * LDS_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    const float* __restrict__ a0 = p.a0 + (size_t)z * p.sa0;
    const float* __restrict__ a1 = p.a1;
    const char* __restrict__ w = BF ? reinterpret_cast<const char*>(p.w16) + (size_t)z * p.sw * 2
                                    : reinterpret_cast<const char*>(p.w + (size_t)z * p.sw);
    float* __restrict__ out = p.out + (size_t)z * p.sout;

    // K order: source (the two halves of a channel concat) > 32/64-channel chunk > tap.  The 9 taps of one channel chunk
    // run back to back, so the (shifted) re-reads of the same input rows hit L1 / L2 instead of going back out to the
    // fabric one full channel sweep later; weights are packed [N][tap][C], any order reads each byte once.
    const int steps0 = (p.c0 + BKE - 1) / BKE, steps1 = (p.c1 + BKE - 1) / BKE;
    const int nk = p.taps * (steps0 + steps1);

    const int c4 = tid & 7;
    const int r0 = tid >> 3;

    // Gather table in LDS: source pixel of (tap, tile row), ~0u for zero padding and rows >= M.  It folds the conv
    // geometry -- padding, stride, the fp32-scale nearest resize of Upsample3D (resnet.py:58-61) -- out of the k-loop;
    // a linear / 1x1 layer is the degenerate 1 x M map (pixel = row).  Per k-step a thread then needs, per row, one
    // LDS word, one compare, one 64-bit multiply-add and a pointer select: masked elements read 16 zero bytes through
    // that select, made BEFORE the load, so the loaded registers are untouched until the LDS store and the whole k-step
    // is one basic block (on gfx950 VALU work is not hidden behind the fp32 MFMA: fewer instructions = more TFLOP/s).
    unsigned* tab = reinterpret_cast<unsigned*>(smem + NBUF * (BM + BN) * LDS_LD);
    // Block-relative addressing: every load of the block is  base (SGPR descriptor) + 32-bit byte offset, with the
    // base moved to the first image (conv) / first row (linear) the block touches, so that offsets stay far below the
    // 2 GB window of the descriptor whatever the tensor size.  Masked elements use an offset outside the window: the
    // buffer unit returns zeros for them, no pointer select and no 64-bit arithmetic per load.
    const int hw_out = p.Ho * p.Wo, hw_in = p.Hs * p.Ws;
    const int img0 = p.taps == 1 ? 0 : (bm * BM) / hw_out;
    const size_t row_base = p.taps == 1 ? (size_t)bm * BM : (size_t)img0 * hw_in;
    {
        for (int e = tid; e < p.taps * BM; e += NT) {
            const int tap = e / BM, row = e - tap * BM;
            const int m = bm * BM + row;
            unsigned pix = ~0u;
            if (m < p.M) {
                if (p.taps == 1) {
                    pix = (unsigned)row;
                } else {
                    const int img = m / hw_out;
                    const int rem = m - img * hw_out;
                    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;          // tap < 9
                    const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
                    if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) {
                        int sy = iy, sx = ix;
                        if (p.upsample) {      // torch nearest: src = min(floor(dst * (in/out)), in - 1), fp32 scale
                            sy = min((int)floorf((float)iy * p.ups_h), p.Hs - 1);
                            sx = min((int)floorf((float)ix * p.ups_w), p.Ws - 1);
                        }
                        pix = (unsigned)(((img - img0) * p.Hs + sy) * p.Ws + sx);
                    }
                }
            }
            tab[e] = pix;
        }
    }
    constexpr unsigned OOB = 0x80000000u;                   // beyond the descriptor window (with or without soffset): reads zeros
    const float* const a0b = a0 + row_base * p.lda0;
    const float* const a1b = p.c1 > 0 ? a1 + row_base * p.lda1 : a0b;
    // descriptor of the current source, rebuilt from a pointer forced into SGPRs (a select between two ready-made
    // descriptors makes hipcc wrap every load in a waterfall loop: it cannot prove the selected one wave-uniform)
    auto rsrc_of = [](const void* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), (short)0, 0x7FFFFFF0,
                                                 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(w + (size_t)n0 * p.ldw * ESZ), (short)0, 0x7FFFFFF0, 0x00020000);
    unsigned b_off[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int r = r0 + RPP * j;
        b_off[j] = (n0 + r < p.N) ? (unsigned)(r * p.ldw * ESZ + c4 * 16) : OOB;
    }
    __syncthreads();

    // state of the NEXT tile to load (all wave-uniform except pixn)
    int k_src = 0, k_cb = 0, k_tap = 0, cseg = p.c0, ldb = p.lda0 * 4;
    bool done = false;
    unsigned pixn[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) pixn[i] = tab[r0 + RPP * i];
    // two register sets: the loads of tile ks+2 are issued while tile ks+1 still sits in the other set, so a load has a
    // k-step and a half (~6000 cycles at 2 waves per SIMD) to come back before its LDS store needs it
    f32x4 rga[2][AR * KH], rgb[2][BR];
    auto issue_loads = [&](f32x4 (&ra)[AR * KH], f32x4 (&rb)[BR]) {
        const unsigned colb = (unsigned)(k_cb + c4 * 4) * 4u;
        const __amdgpu_buffer_rsrc_t rsa = rsrc_of(k_src ? a1b : a0b);
        // bitwise (not short-circuit) conditions: the offset arithmetic stays branch-free -- one 24-bit multiply-add,
        // one compare and one select per row (block-relative pixel indices and row strides are < 2^24)
        const bool cok0 = (!done) & (k_cb + c4 * 4 < cseg);
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const unsigned rowb = __umul24(pixn[i], (unsigned)ldb) + colb;
            const bool rok = (pixn[i] != ~0u) & cok0;
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                const bool ok = kh == 0 ? rok : (rok & (k_cb + kh * 32 + c4 * 4 < cseg));
                ra[i * KH + kh] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, ok ? rowb + kh * 128 : OOB, 0, 0));
            }
        }
        const bool cokb = (!done) & (k_cb + c4 * (16 / ESZ) < cseg);
        const int cbase = (k_src ? p.c0 : 0) + k_cb;                               // channel of this k-step in the concat
        const int koffb = (p.taps == 1 ? cbase : (cbase / BKE * 9 + k_tap) * BKE) * ESZ;   // wave-uniform: rides in soffset
#pragma unroll
        for (int j = 0; j < BR; ++j)
            rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, cokb ? b_off[j] : OOB, koffb, 0));
        // advance (tap fastest, then chunk, then source) -- scalar selects, no branch
        const int t2 = k_tap + 1;
        const bool wrap_t = t2 == p.taps;
        k_tap = wrap_t ? 0 : t2;
        const int cb2 = wrap_t ? k_cb + BKE : k_cb;
        const bool wrap = cb2 >= cseg;
        k_cb = wrap ? 0 : cb2;
        const bool more = k_src == 0 && p.c1 > 0;
        done = done || (wrap && !more);
        k_src = (wrap && more) ? 1 : k_src;
        cseg = k_src ? p.c1 : p.c0;
        ldb = (k_src ? p.lda1 : p.lda0) * 4;
        // pixel indices of the tile after this one: consumed one k-step from now, so the LDS latency is off the path
#pragma unroll
        for (int i = 0; i < AR; ++i) pixn[i] = tab[k_tap * BM + r0 + RPP * i];
    };
    auto store_tile = [&](int buf, const f32x4 (&ra)[AR * KH], const f32x4 (&rb)[BR]) {
        float* A = As + buf * BM * LDS_LD;
        float* B = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            if constexpr (BF) {
#pragma unroll
                for (int kh = 0; kh < KH; ++kh)            // 4 fp32 -> 4 bf16 (v_cvt_pk_bf16_f32), 8 bytes at k = 32 kh + 4 c4
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(A + (r0 + RPP * i) * LDS_LD) + kh * 64 + c4 * 8) =
                        __builtin_convertvector(ra[i * KH + kh], bf16x4);
            } else {
                *reinterpret_cast<f32x4*>(A + (r0 + RPP * i) * LDS_LD + c4 * 4) = ra[i];
            }
        }
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

    issue_loads(rga[0], rgb[0]);                // tile 0
    issue_loads(rga[1], rgb[1]);                // tile 1 (all-zero dummy loads if nk == 1)
    store_tile(0, rga[0], rgb[0]);
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
        if constexpr (BF) {        // one 16-deep bf16 MFMA per tile: the lane's 16 bytes are k = 16 g + 8 h .. + 7
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[set][ni]),
                                                                          __builtin_bit_cast(bf16x8, af[set][mi]),
                                                                          acc[mi][ni], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[set][ni][s], af[set][mi][s], acc[mi][ni], 0, 0, 0);
        }
    };
    static_assert(BK == 32 && NBUF == 2, "the k-step below is written for two LDS buffers and four fragment groups of 32 bytes per row");
    read_frags(0, 0, 0);
    // one k-step; P = ks & 1 is a template constant (the loop is unrolled by two), so LDS buffer and register set are static
    auto kstep = [&](auto Pc) {
        constexpr int P = decltype(Pc)::value;
        if (ABL == 0) issue_loads(rga[P], rgb[P]);      // tile ks+2 (all-zero dummy loads behind the last tile)
        // keep the loads at the head of the k-step; left alone the scheduler sinks them to the end of the block
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, P, 1);
        mma(0);
        read_frags(0, P, 2);
        mma(1);
        read_frags(1, P, 3);
        mma(0);
        if (ABL != 2) {
            store_tile(P ^ 1, rga[P ^ 1], rgb[P ^ 1]);  // tile ks+1, loaded one k-step ago
            __syncthreads();
        }
        read_frags(0, P ^ 1, 0);                // first fragments of tile ks+1 ...
        __builtin_amdgcn_sched_barrier(0);
        mma(1);                                 // ... land while the last k-group of tile ks is multiplied
    };
    int ks = 0;
    for (; ks + 1 < nk; ks += 2) {
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
    }
    if (ks < nk) kstep(std::integral_constant<int, 0>{});

    igemm_epilogue<BM, TM, TN, WM, WN>(p, acc, out, bm, n0, wm, wn, lane, reinterpret_cast<float*>(smem));
}

// =====================================================================================================
// "f32x3": fp32 GEMM on the bf16 matrix pipe.  Every fp32 operand is split EXACTLY into three bf16 pieces by truncation
// (x = x1 + x2 + x3: 8 + 8 + 8 = 24 mantissa bits), and a product is the sum of the six piece products of weight at
// least 2^-16: x1 y1 + x1 y2 + x2 y1 + x1 y3 + x3 y1 + x2 y2 (the three dropped ones are below 2^-24 of the product, the
// rounding of an fp32 multiply).  Piece products are exact in fp32 and accumulate in the MFMA's fp32 accumulator, so the
// result carries fp32-level error (measured 2.5e-7 of the output scale at K = 1280, the fp32 FMA chain 6e-7) while the
// arithmetic runs on v_mfma_f32_32x32x16_bf16: 6 x 32 = 192 cycles per 32x32x16 block instead of 512 on the fp32 MFMA, on
// a pipe that -- unlike the fp32 MFMA -- does not share its ALUs with the VALU.
// Weights are split once (e2v_finalize_weights: three bf16 planes); activations are split on their way into LDS.
// Opt-in (E2V_F32X3): it changes "computes in f32" into "computes f32-equivalent products on the bf16 pipe".
// taps == 1 only (linears, Winograd-domain GEMMs = 95 % of the igemm time in fp32 mode).
// First version: one 32-k LDS stage (61 KB, 2 workgroups per CU), two barriers per stage, 150-170 TFLOP/s fp32-equivalent
// (36-40 % of the bf16 pipe / 6).  A variant with 16-k stages, double-buffered LDS, one barrier per stage and the split
// interleaved with the MFMAs (sched_group_barrier) measured no better (136-160): the six piece products read one 16-byte
// LDS fragment per MFMA, 4x the fp32 tile's LDS traffic per matrix-pipe cycle.  A third form -- one 16-k LDS stage (37 KB),
// streamed activation-plane fragments, 132 registers, three workgroups per CU, loads one stage ahead -- gained 3 % on K = 320
// and lost 10-17 % on K >= 1280 (prefetch too shallow).  tools/micro/lds_mfma.hip: the fragment-read + MFMA core alone
// sustains 350-410 TFLOP/s fp32-equivalent, so the loss is in the load / split / store phases around it.
// =====================================================================================================
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int BM, int BN, int WGM, int WGN, int ABLX = 0>
__device__ __forceinline__ void igemm_tile_x3(const IgemmArgs& p, const int rbg, const int n0, char* smem) {
    const int z = p.batch > 1 ? rbg / p.nbm_per : 0;
    const int bm = rbg - z * p.nbm_per;
    constexpr int BKE = 32;
    constexpr int PLD = 80;                         // bytes per LDS row of one plane: 32 bf16 + 16 pad (conflict-free b128 reads)
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPP = NT / 8;
    constexpr int AR = BM / RPP;
    constexpr int BRX = BN * 4 / NT;                // 16-byte weight loads per thread and plane (4 lanes per 64-byte row)
    char* Ap = smem;                                // [3][BM][PLD]
    char* Bp = smem + 3 * BM * PLD;                 // [3][BN][PLD]
    unsigned* tab = reinterpret_cast<unsigned*>(smem + 3 * (BM + BN) * PLD);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const float* __restrict__ a0 = p.a0 + (size_t)z * p.sa0;
    const float* __restrict__ a1 = p.a1;
    const char* __restrict__ w = reinterpret_cast<const char*>(p.w3) + ((size_t)z * p.sw + (size_t)n0 * p.ldw) * 2;
    float* __restrict__ out = p.out + (size_t)z * p.sout;

    const int steps0 = (p.c0 + BKE - 1) / BKE, steps1 = (p.c1 + BKE - 1) / BKE;
    const int nk = steps0 + steps1;
    const int c4 = tid & 7;
    const int r0 = tid >> 3;

    // taps == 1: the gather table is the identity over the block's rows (kept so that masked rows read through the same
    // out-of-window offset as in the fp32 tile)
    const size_t row_base = (size_t)bm * BM;
    for (int e = tid; e < BM; e += NT) tab[e] = (bm * BM + e < p.M) ? (unsigned)e : ~0u;
    constexpr unsigned OOB = 0x80000000u;
    const float* const a0b = a0 + row_base * p.lda0;
    const float* const a1b = p.c1 > 0 ? a1 + row_base * p.lda1 : a0b;
    auto rsrc_of = [](const void* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), (short)0, 0x7FFFFFF0,
                                                 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rw0 = rsrc_of(w), rw1 = rsrc_of(w + (size_t)p.w3_plane * 2), rw2 = rsrc_of(w + (size_t)p.w3_plane * 4);
    unsigned b_off[BRX];
    int b_q[BRX];
#pragma unroll
    for (int j = 0; j < BRX; ++j) {
        const int idx = tid + NT * j;
        const int row = idx >> 2, q = idx & 3;
        b_q[j] = q;
        b_off[j] = (n0 + row < p.N) ? (unsigned)(row * p.ldw * 2 + q * 16) : OOB;
    }
    __syncthreads();

    int k_src = 0, k_cb = 0, cseg = p.c0, ldb = p.lda0 * 4;
    bool done = false;
    unsigned pixn[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) pixn[i] = tab[r0 + RPP * i];
    f32x4 rga[2][AR], rgb[2][3 * BRX];
    auto issue_loads = [&](f32x4 (&ra)[AR], f32x4 (&rb)[3 * BRX]) {
        const unsigned colb = (unsigned)(k_cb + c4 * 4) * 4u;
        const __amdgpu_buffer_rsrc_t rsa = rsrc_of(k_src ? a1b : a0b);
        const bool cok0 = (!done) & (k_cb + c4 * 4 < cseg);
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const unsigned rowb = __umul24(pixn[i], (unsigned)ldb) + colb;
            const bool ok = (pixn[i] != ~0u) & cok0;
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, ok ? rowb : OOB, 0, 0));
        }
        const int koffb = ((k_src ? p.c0 : 0) + k_cb) * 2;                          // wave-uniform: rides in soffset
#pragma unroll
        for (int j = 0; j < BRX; ++j) {
            const bool ok = (!done) & (k_cb + b_q[j] * 8 < cseg);
            const unsigned off = ok ? b_off[j] : OOB;
            rb[3 * j + 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw0, off, koffb, 0));
            rb[3 * j + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw1, off, koffb, 0));
            rb[3 * j + 2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw2, off, koffb, 0));
        }
        const int cb2 = k_cb + BKE;
        const bool wrap = cb2 >= cseg;
        k_cb = wrap ? 0 : cb2;
        const bool more = k_src == 0 && p.c1 > 0;
        done = done || (wrap && !more);
        k_src = (wrap && more) ? 1 : k_src;
        cseg = k_src ? p.c1 : p.c0;
        ldb = (k_src ? p.lda1 : p.lda0) * 4;
    };
    // activations: exact three-way truncation split of four fp32 -> three packed bf16x4 (8 bytes per plane)
    auto store_tile = [&](const f32x4 (&ra)[AR], const f32x4 (&rb)[3 * BRX]) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
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
            char* dst = Ap + (r0 + RPP * i) * PLD + c4 * 8;
            const u32x2 v1 = {(h1[0] >> 16) | (h1[1] & 0xFFFF0000u), (h1[2] >> 16) | (h1[3] & 0xFFFF0000u)};
            const u32x2 v2 = {(h2[0] >> 16) | (h2[1] & 0xFFFF0000u), (h2[2] >> 16) | (h2[3] & 0xFFFF0000u)};
            const u32x2 v3 = {(h3[0] >> 16) | (h3[1] & 0xFFFF0000u), (h3[2] >> 16) | (h3[3] & 0xFFFF0000u)};
            *reinterpret_cast<u32x2*>(dst) = v1;
            *reinterpret_cast<u32x2*>(dst + BM * PLD) = v2;
            *reinterpret_cast<u32x2*>(dst + 2 * BM * PLD) = v3;
        }
#pragma unroll
        for (int j = 0; j < BRX; ++j) {
            const int idx = tid + NT * j;
            char* dst = Bp + (idx >> 2) * PLD + (idx & 3) * 16;
            *reinterpret_cast<f32x4*>(dst) = rb[3 * j + 0];
            *reinterpret_cast<f32x4*>(dst + BN * PLD) = rb[3 * j + 1];
            *reinterpret_cast<f32x4*>(dst + 2 * BN * PLD) = rb[3 * j + 2];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    issue_loads(rga[0], rgb[0]);
    issue_loads(rga[1], rgb[1]);
    store_tile(rga[0], rgb[0]);
    __syncthreads();

    const char* Afr = Ap + (wm * WM + (lane & 31)) * PLD + (lane >> 5) * 16;
    const char* Bfr = Bp + (wn * WN + (lane & 31)) * PLD + (lane >> 5) * 16;
    auto kstep = [&](auto Pc) {
        constexpr int P = decltype(Pc)::value;
        if (ABLX < 1) issue_loads(rga[P], rgb[P]);      // tile ks+2
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {                   // two 16-deep chunks of the 32-k stage
            bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
                    af[pl][mi] = *reinterpret_cast<const bf16x8*>(Afr + pl * BM * PLD + mi * 32 * PLD + g * 32);
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    bf[pl][ni] = *reinterpret_cast<const bf16x8*>(Bfr + pl * BN * PLD + ni * 32 * PLD + g * 32);
            }
            // smallest terms first; (weight piece, activation piece)
            constexpr int TW[6] = {1, 2, 0, 1, 0, 0}, TA[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[TW[t]][ni], af[TA[t]][mi], acc[mi][ni], 0, 0, 0);
        }
        if (ABLX < 2) {
            __syncthreads();                            // everyone has read tile ks
            store_tile(rga[P ^ 1], rgb[P ^ 1]);         // tile ks+1
            __syncthreads();
        }
    };
    int ks = 0;
    for (; ks + 1 < nk; ks += 2) {
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
    }
    if (ks < nk) kstep(std::integral_constant<int, 0>{});
    igemm_epilogue<BM, TM, TN, WM, WN>(p, acc, out, bm, n0, wm, wn, lane, reinterpret_cast<float*>(smem));
}

template <int ABLX>
__global__ __launch_bounds__(256) void igemm_x3_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_x3[];
    const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int nrb = rb_hi - rb_lo;
    const int tail = min(nrb, p.tail_rb);
    const int per1 = p.w1 + p.s1;
    const int n1 = (nrb - tail) * per1;
    if (loc < n1) {
        const int r = loc / per1, j = loc - r * per1;
        if (j < p.w1)
            igemm_tile_x3<128, 128, 2, 2, ABLX>(p, rb_lo + r, j * 128, smem_x3);
        else
            igemm_tile_x3<128, 64, 2, 2, ABLX>(p, rb_lo + r, p.w1 * 128 + (j - p.w1) * 64, smem_x3);
    } else {
        const int t = loc - n1;
        if (t >= tail * p.s2) return;
        const int r = t / p.s2;
        igemm_tile_x3<128, 64, 2, 2, ABLX>(p, rb_lo + (nrb - tail) + r, (t - r * p.s2) * 64, smem_x3);
    }
}

static void launch_igemm_x3(const IgemmArgs& a, int ntiles, hipStream_t s) {
    static const int ablx = [] { const char* e = std::getenv("E2V_X3_ABLATE"); return e ? std::atoi(e) : 0; }();   // timing experiments only
    constexpr size_t smem = (size_t)3 * (128 + 128) * 80 + 128 * sizeof(unsigned);   // tile planes + gather table (epilogue staging: 34 KB)
    E2V_KATTR(&igemm_x3_kernel<0>, smem);
    E2V_KATTR(&igemm_x3_kernel<1>, smem);
    E2V_KATTR(&igemm_x3_kernel<2>, smem);
    const double K = (double)(a.c0 + a.c1);
    std::string pname = "igemm_f32x3";
    if (prof_detail())
        pname += " M" + std::to_string(a.M) + " N" + std::to_string(a.N) + " K" + std::to_string((long)K) +
                 (a.geglu ? " geglu" : "") + (a.batch > 1 ? " b" + std::to_string(a.batch) : "");
    ProfScope ps(pname.c_str(), 2.0 * a.M * a.N * K * a.batch,
                 4.0 * a.batch * ((double)a.M * K + 1.5 * a.N * K + (double)a.M * (a.geglu ? a.N / 2 : a.N)), s);
    if (ablx == 1) E2V_KLAUNCH(igemm_x3_kernel<1>, dim3(ntiles, 1, 1), dim3(256), smem, s, a);
    else if (ablx == 2) E2V_KLAUNCH(igemm_x3_kernel<2>, dim3(ntiles, 1, 1), dim3(256), smem, s, a);
    else E2V_KLAUNCH(igemm_x3_kernel<0>, dim3(ntiles, 1, 1), dim3(256), smem, s, a);
}

// -----------------------------------------------------------------------------------------------------
// The fp32 tile for taps == 1 (linears, Winograd-domain GEMMs: 95 % of the igemm time) with 16-k stages.  LDS drops to 40 KB
// per workgroup, so THREE workgroups fit a CU (3 waves per SIMD instead of 2) -- more phases to hide the prologue / epilogue
// of short-K tiles and each other's barriers behind -- at the price of a barrier and the loop bookkeeping every 32 MFMAs
// instead of every 64.  Measured against the 32-k tile (same box): linears +5-8 %, Winograd convs +3-5 %, whole pass +3.5 %.
// Same k order, so results are bit-identical.  E2V_IGEMM_K16=0 falls back to the 32-k tile (which direct convs still use).
// -----------------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void igemm_tile_k16(const IgemmArgs& p, const int rbg, const int n0, float* smem) {
    const int z = p.batch > 1 ? rbg / p.nbm_per : 0;
    const int bm = rbg - z * p.nbm_per;
    constexpr int BKE = 16, LD = 20;                // 16 floats + 4 pad per LDS row: conflict-free b128 reads
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPP = NT / 4;                     // rows per loader pass (4 lanes per 64-byte row piece)
    constexpr int AR = BM / RPP;
    constexpr int BR = (BN + RPP - 1) / RPP;        // BN = 64: one pass, upper half of the threads idle
    float* As = smem;                               // [2][BM][LD]
    float* Bs = smem + 2 * BM * LD;                 // [2][BN][LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const float* __restrict__ a0 = p.a0 + (size_t)z * p.sa0;
    const char* __restrict__ w = reinterpret_cast<const char*>(p.w + (size_t)z * p.sw + (size_t)n0 * p.ldw);
    float* __restrict__ out = p.out + (size_t)z * p.sout;
    const int nk = (p.c0 + BKE - 1) / BKE + (p.c1 + BKE - 1) / BKE;
    const int q = tid & 3, r0 = tid >> 2;
    constexpr unsigned OOB = 0x80000000u;
    const size_t row_base = (size_t)bm * BM;
    const float* const a0b = a0 + row_base * p.lda0;
    const float* const a1b = p.c1 > 0 ? p.a1 + row_base * p.lda1 : a0b;
    auto rsrc_of = [](const void* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), (short)0, 0x7FFFFFF0,
                                                 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rw = rsrc_of(w);
    unsigned a_row[AR], b_off[BR];
#pragma unroll
    for (int i = 0; i < AR; ++i) a_row[i] = (bm * BM + r0 + RPP * i < p.M) ? (unsigned)(r0 + RPP * i) : ~0u;
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int r = r0 + RPP * j;
        b_off[j] = (r < BN && n0 + r < p.N) ? (unsigned)(r * p.ldw * 4 + q * 16) : OOB;
    }
    int k_src = 0, k_cb = 0, cseg = p.c0, ldb = p.lda0 * 4;
    bool done = false;
    f32x4 rga[2][AR], rgb[2][BR];
    auto issue_loads = [&](f32x4 (&ra)[AR], f32x4 (&rb)[BR]) {
        const unsigned colb = (unsigned)(k_cb + q * 4) * 4u;
        const __amdgpu_buffer_rsrc_t rsa = rsrc_of(k_src ? a1b : a0b);
        const bool cok = (!done) & (k_cb + q * 4 < cseg);
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const unsigned rowb = __umul24(a_row[i], (unsigned)ldb) + colb;
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, ((a_row[i] != ~0u) & cok) ? rowb : OOB, 0, 0));
        }
        const int koffb = ((k_src ? p.c0 : 0) + k_cb) * 4;
#pragma unroll
        for (int j = 0; j < BR; ++j)
            rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, cok ? b_off[j] : OOB, koffb, 0));
        const int cb2 = k_cb + BKE;
        const bool wrap = cb2 >= cseg;
        k_cb = wrap ? 0 : cb2;
        const bool more = k_src == 0 && p.c1 > 0;
        done = done || (wrap && !more);
        k_src = (wrap && more) ? 1 : k_src;
        cseg = k_src ? p.c1 : p.c0;
        ldb = (k_src ? p.lda1 : p.lda0) * 4;
    };
    auto store_tile = [&](int buf, const f32x4 (&ra)[AR], const f32x4 (&rb)[BR]) {
#pragma unroll
        for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(As + buf * BM * LD + (r0 + RPP * i) * LD + q * 4) = ra[i];
#pragma unroll
        for (int j = 0; j < BR; ++j)
            if (r0 + RPP * j < BN) *reinterpret_cast<f32x4*>(Bs + buf * BN * LD + (r0 + RPP * j) * LD + q * 4) = rb[j];
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    issue_loads(rga[0], rgb[0]);
    issue_loads(rga[1], rgb[1]);
    store_tile(0, rga[0], rgb[0]);
    __syncthreads();
    const int frag_off = (lane & 31) * LD + (lane >> 5) * 4;
    const float* Afr = As + wm * WM * LD + frag_off;
    const float* Bfr = Bs + wn * WN * LD + frag_off;
    f32x4 af[2][TM], bf[2][TN];
    auto read_frags = [&](int set, int buf, int g) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) af[set][mi] = *reinterpret_cast<const f32x4*>(Afr + buf * BM * LD + mi * 32 * LD + g * 8);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) bf[set][ni] = *reinterpret_cast<const f32x4*>(Bfr + buf * BN * LD + ni * 32 * LD + g * 8);
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[set][ni][e], af[set][mi][e], acc[mi][ni], 0, 0, 0);
    };
    read_frags(0, 0, 0);
    auto kstep = [&](auto Pc) {
        constexpr int P = decltype(Pc)::value;
        issue_loads(rga[P], rgb[P]);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, P, 1);
        mma(0);
        store_tile(P ^ 1, rga[P ^ 1], rgb[P ^ 1]);
        __syncthreads();
        read_frags(0, P ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(1);
    };
    int ks = 0;
    for (; ks + 1 < nk; ks += 2) {
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
    }
    if (ks < nk) kstep(std::integral_constant<int, 0>{});
    igemm_epilogue<BM, TM, TN, WM, WN>(p, acc, out, bm, n0, wm, wn, lane, smem);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void igemm_k16_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem_k16[];
    const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int nrb = rb_hi - rb_lo;
    const int tail = min(nrb, p.tail_rb);
    const int per1 = p.w1 + p.s1;
    const int n1 = (nrb - tail) * per1;
    if (loc < n1) {
        const int r = loc / per1, j = loc - r * per1;
        if (j < p.w1) igemm_tile_k16<128, 128, 2, 2>(p, rb_lo + r, j * 128, smem_k16);
        else igemm_tile_k16<128, 64, 2, 2>(p, rb_lo + r, p.w1 * 128 + (j - p.w1) * 64, smem_k16);
    } else {
        const int t = loc - n1;
        if (t >= tail * p.s2) return;
        const int r = t / p.s2;
        igemm_tile_k16<128, 64, 2, 2>(p, rb_lo + (nrb - tail) + r, (t - r * p.s2) * 64, smem_k16);
    }
}

// One launch runs a MIX of tile shapes (IgemmArgs::rb1/w1/s1/s2): row blocks [0, rb1) are cut into w1 tiles of
// 128x128 followed by s1 tiles of 128x64 (N = 320 -> 2 + 1, no wasted columns); row blocks [rb1, nbm) are cut into s2
// tiles of 128x64 only.  The launcher sizes rb1 so that the 128x128 part fills whole rounds of the 512 resident
// tiles and the ragged last round runs as half-size tiles (L1: 5 rounds -> 4.5, L2: 3 -> 2.5).
// XCD-aware order: the 8 XCDs are dealt blocks round-robin; each XCD gets a contiguous run of tiles (columns
// fastest), so the column tiles that share one gathered A tile hit the same L2.
template <int ABL>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD x = blockIdx % 8 owns the contiguous row blocks [x*nbm/8, (x+1)*nbm/8); the last `tail` of them are cut
    // into 128x64 tiles only, the others into w1 tiles of 128x128 + s1 of 128x64.  Every XCD therefore ends on
    // half-size tiles (an even tail), and column tiles sharing one gathered A tile stay on one L2.
    const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int nrb = rb_hi - rb_lo;
    const int tail = min(nrb, p.tail_rb);
    const int per1 = p.w1 + p.s1;
    const int n1 = (nrb - tail) * per1;
    if (loc < n1) {
        const int r = loc / per1, j = loc - r * per1;
        if (j < p.w1)
            igemm_tile<128, 128, 2, 2, ABL, 2>(p, rb_lo + r, j * 128, smem);
        else
            igemm_tile<128, 64, 2, 2, ABL, 2>(p, rb_lo + r, p.w1 * 128 + (j - p.w1) * 64, smem);
    } else {
        const int t = loc - n1;
        if (t >= tail * p.s2) return;                     // padding block of the 8 x max-chunk grid
        const int r = t / p.s2;
        igemm_tile<128, 64, 2, 2, ABL, 2>(p, rb_lo + (nrb - tail) + r, (t - r * p.s2) * 64, smem);
    }
}

template <int ABL>
static void launch_igemm(const IgemmArgs& a, int ntiles, const char* cls, hipStream_t s) {
    constexpr size_t smem = (size_t)2 * (128 + 128) * LDS_LD * sizeof(float) + 9 * 128 * sizeof(unsigned);   // tiles + gather table
    E2V_KATTR(&igemm_kernel<ABL>, smem);
    dim3 grid(ntiles, 1, 1);
    const double K = (double)a.taps * (a.c0 + a.c1);
    const double rows_in = a.taps == 1 ? (double)a.M : (double)a.M * a.Hs * a.Ws / ((double)a.Ho * a.Wo);
    std::string pname = cls;
    if (prof_detail())
        pname += " M" + std::to_string(a.M) + " N" + std::to_string(a.N) + " K" + std::to_string((long)K) + " t" +
                 std::to_string(a.taps) + (a.stride > 1 ? " s2" : "") + (a.upsample ? " up" : "") + (a.c1 ? " cat" : "") +
                 (a.geglu ? " geglu" : "") + (a.batch > 1 ? " b" + std::to_string(a.batch) : "") + " rb1=" +
                 std::to_string(a.rb1) + " w" + std::to_string(a.w1) + " s" + std::to_string(a.s1);
    ProfScope ps(pname.c_str(), 2.0 * a.M * a.N * K * a.batch,
                 4.0 * a.batch * (rows_in * (a.c0 + a.c1) + (double)a.N * K + (double)a.M * (a.geglu ? a.N / 2 : a.N)), s);
    dry_tag(" -> igemm_kernel 128x128x32");
    E2V_KLAUNCH((igemm_kernel<ABL>), grid, dim3(256), smem, s, a);
}

bool igemm_writes_rbsum(const IgemmArgs& a_in) {
    if (!a_in.rbsum || !a_in.a_bf16 || a_in.M <= 0 || a_in.N <= 0) return false;
    IgemmArgs a = a_in;
    if (a.taps == 1) {
        a.Ho = a.Hi = a.Hs = 1;
        a.Wo = a.Wi = a.Ws = a.M;
        a.stride = 1; a.pad = 0; a.upsample = 0;
    }
    a.ldw = a.ldw16;
    return bgemm_t256_writes_rbsum(a);
}

void igemm(const IgemmArgs& a_in, hipStream_t s) {
    if (a_in.M <= 0 || a_in.N <= 0) return;
    IgemmArgs a = a_in;
    if (a.taps == 1) {      // linear / 1x1: a 1 x M map without padding -- the same gather as the 3x3 case
        a.Ho = a.Hi = a.Hs = 1;
        a.Wo = a.Wi = a.Ws = a.M;
        a.stride = 1; a.pad = 0; a.upsample = 0;
    }
    // ---- tile schedule ----------------------------------------------------------------------------------
    static const int abl = [] { const char* e = std::getenv("E2V_IGEMM_ABLATE"); return e ? std::atoi(e) : 0; }();
    static const int sched = [] { const char* e = std::getenv("E2V_IGEMM_SCHED"); return e ? std::atoi(e) : 1; }();   // 0: no half-size tails
    // bf16-activation launches with enough row blocks run 256-row tiles (bgemm256_kernel: 8 waves, one workgroup per CU): the
    // fill rate of a CU's LDS (~35 B/clk through L1) bounds a 128 x 128 x 64 bf16 tile at ~55 % of the matrix pipe; a 256 x 128
    // tile moves 0.75x the bytes per flop.  The schedule below is the same with 256-row blocks and 256 resident tiles.
    // split-K (the small-batch family, bgemm.hip): asked for by the caller (a.sk runs, a.sk_ws) and taken when the launch is eligible
    const bool want_sk = a.a_bf16 && a.sk >= 2 && a.sk_ws && splitk_plan(a, a.sk) >= 2;
    const int BMrows = (a.a_bf16 && !want_sk && bgemm_use_256(a)) ? 256 : 128;
    a.bm256 = BMrows == 256 ? 1 : 0;
    a.nbm_per = (a.M + BMrows - 1) / BMrows;
    const int nbm = a.nbm_per * a.batch;                     // batch entries are just more row blocks (dealt to the XCDs together)
    static const int k16 = [] { const char* e = std::getenv("E2V_IGEMM_K16"); return e ? std::atoi(e) : 1; }();   // 0: the 32-k tile for everything
    // The tail heuristics below size rounds of 512 tiles (2 workgroups per CU).  The 16-k tile runs 3 per CU, but sizing its
    // rounds at 768 measured worse (level-2 linears 129 -> 116 TFLOP/s, UNet step 272.9 -> 278.9 ms): the third workgroup is
    // better spent overlapping than being planned for.
    const int slots = BMrows == 256 ? 256 : 512;
    a.s2 = (a.N + 63) / 64;
    a.w1 = a.N / 128;                                        // full 128-wide column tiles
    const int rem = a.N - a.w1 * 128;
    a.s1 = rem == 0 ? 0 : (rem <= 64 ? 1 : 0);
    if (rem > 64) a.w1 += 1;                                 // 65..127 leftover columns: one more (masked) wide tile
    a.rb1 = nbm;
    constexpr bool use_bf16 = false;
    // split-bf16 fp32 (see igemm_tile_x3): linears / Winograd GEMMs whose K is a multiple of 8 and whose weights were split
    const bool use_x3 = !use_bf16 && a.x3 && a.w3 && a.taps == 1 && !a.relu && a.c0 % 8 == 0 && a.c1 % 8 == 0 && a.ldw % 8 == 0 &&
                        abl == 0;
    if (a.a_bf16) {       // bf16 activations in HBM: the LDS-DMA kernels of bgemm.hip (16-byte pieces: 8 bf16)
        if (!a.w16 || a.c0 % 8 || a.c1 % 8 || a.lda0 % 8 || a.lda1 % 8 || a.ldw16 % 8 || (a.c1 > 0 && a.taps != 1 && a.c0 % 64) ||
            (reinterpret_cast<uintptr_t>(a.a0) & 15) || (reinterpret_cast<uintptr_t>(a.a1) & 15))
            throw Error(E2V_ESHAPE, "bf16 GEMM: channel counts / row strides must be multiples of 8 (concat seam of a 3x3 conv: 64) and rows 16-byte aligned");
        a.ldw = a.ldw16;
        if (!want_sk && bgemm_t256_launch(a, s)) return;     // 256 x 256 / 256 x 320 deep-pipelined tiles (bgemm256.hip), by layer shape
    }
    const char* cls = "igemm_f32";
    if (a.geglu) {                                           // the GEGLU epilogue pairs the two 32-column halves of a wave
        a.w1 = (a.N + 127) / 128; a.s1 = 0;
    } else if (a.w1 == 0) {                                  // N <= 64
        a.rb1 = 0;
    } else if (want_sk) {                                    // full-size tiles: the runs multiply the workgroup count
    } else if (sched == 1) {
        // per XCD: 64 resident tiles per round.  Keep whole rounds of full-size tiles; if what is left of the XCD's
        // chunk is at most half a round, run it as half-size tiles (it then takes about half a round's time).
        const double per_rb = a.w1 + 0.5 * a.s1;             // cost of one row block in 128x128 units
        const int nrb = (nbm + 7) / 8;                       // row blocks of the largest chunk
        const double units = per_rb * nrb;
        const int xslots = slots / 8;
        if (units * 8 < slots) {
            a.rb1 = 0;                                       // less than one round: half-size tiles spread over more CUs
        } else {
            const int full = (int)std::floor(units / xslots);
            const int wide_rb = (int)std::floor(full * xslots / per_rb);
            const int tail = nrb - wide_rb;
            if (tail > 0 && tail * per_rb <= 0.5 * xslots) a.rb1 = nbm - 8 * tail < 0 ? 0 : nbm - 8 * tail;
        }
    }
    // Launches of only a few rounds (the level-2 linears of a B = 8 pass: 1080 tiles of 128 x 128 on 768 slots = 1.4 rounds) leave the
    // chip part idle in their last round.  E2V_IGEMM_HALF_BELOW = r x 100 > 0: a launch of fewer than r rounds of full-size tiles runs
    // ENTIRELY as 128 x 64 tiles (twice the workgroups at half the work: the round granularity halves; same k order per output, so
    // bit-identical).  Same-process A/B: profiles/r04_shape_ab_fp32_tail.log.
    static const int* const half_below = E2V_AB_KNOB("E2V_IGEMM_HALF_BELOW", 0);      // (measured +-0: `make ab` only)
    if (!a.a_bf16 && *half_below > 0 && !a.geglu && a.w1 > 0 && a.rb1 > 0) {
        const double per_rb = a.w1 + 0.5 * a.s1;
        const double rounds = per_rb * ((nbm + 7) / 8) / (slots / 8);
        if (rounds * 100.0 < (double)*half_below) a.rb1 = 0;
    }
    if (a.a_bf16 && bgemm_all_n64(a)) a.rb1 = 0;
    a.nbm = nbm;
    a.tail_rb = (nbm - a.rb1 + 7) / 8;                       // per XCD chunk (rb1 == 0: every row block is "tail")
    if (a.rb1 == 0) a.tail_rb = nbm;
    int ntiles = 0;                                          // grid = 8 x the largest chunk's tile count
    for (int x = 0; x < 8; ++x) {
        const int nrb = (int)(((long)(x + 1) * nbm) >> 3) - (int)(((long)x * nbm) >> 3);
        const int tail = nrb < a.tail_rb ? nrb : a.tail_rb;
        const int t = (nrb - tail) * (a.w1 + a.s1) + tail * a.s2;
        ntiles = t > ntiles ? t : ntiles;
    }
    ntiles *= 8;
    if (want_sk) { bgemm_splitk_launch(a, s); return; }
    if (a.a_bf16) { bgemm_launch(a, ntiles, s); return; }
    if (use_x3) { launch_igemm_x3(a, ntiles, s); return; }
    if (k16 && !use_bf16 && a.taps == 1 && abl == 0) {
        constexpr size_t smem16 = (size_t)2 * (128 + 128) * 20 * sizeof(float);   // 40 KB (epilogue staging: 4 x 32 x 68 floats = 34 KB)
        E2V_KATTR(&igemm_k16_kernel, smem16);
        const double K = (double)(a.c0 + a.c1);
        std::string pname = cls;
        if (prof_detail())
            pname += " M" + std::to_string(a.M) + " N" + std::to_string(a.N) + " K" + std::to_string((long)K) + " t1" + (a.c1 ? " cat" : "") +
                     (a.geglu ? " geglu" : "") + (a.batch > 1 ? " b" + std::to_string(a.batch) : "") + " k16";
        ProfScope ps(pname.c_str(), 2.0 * a.M * a.N * K * a.batch,
                     4.0 * a.batch * ((double)a.M * K + (double)a.N * K + (double)a.M * (a.geglu ? a.N / 2 : a.N)), s);
        dry_tag(" rb1=" + std::to_string(a.rb1) + " w" + std::to_string(a.w1) + " s" + std::to_string(a.s1) + " -> igemm_k16_kernel 128x128x16");
        E2V_KLAUNCH(igemm_k16_kernel, dim3(ntiles, 1, 1), dim3(256), smem16, s, a);
        return;
    }
    if (abl == 1) launch_igemm<1>(a, ntiles, cls, s);
    else if (abl == 2) launch_igemm<2>(a, ntiles, cls, s);
    else launch_igemm<0>(a, ntiles, cls, s);
}

// ---- one-off weight re-layout ---------------------------------------------------------------------
// [O][I][3][3] -> [O][chunk][tap][bke]: the k order the kernel walks (channel chunk > tap), so that every k-step reads the
// next 128 contiguous bytes of each weight row; channels beyond I are zero (a ragged last chunk costs no masking).
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ o, int cout, int cin, int bke) {
    const int nq = (cin + bke - 1) / bke;
    const size_t per = (size_t)nq * 9 * bke;
    const size_t total = (size_t)cout * per;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int oc = i / per;
        const size_t r = i - (size_t)oc * per;
        const int e = r % bke;
        const int tap = (r / bke) % 9;
        const int q = r / ((size_t)bke * 9);
        const int c = q * bke + e;
        o[i] = (c < cin) ? w[((size_t)oc * cin + c) * 9 + tap] : 0.f;
    }
}
int conv3x3_packed_ld(int cin, int bke) { return (cin + bke - 1) / bke * 9 * bke; }
void pack_conv3x3(const float* w, float* o, int cout, int cin, int bke, hipStream_t s) {
    const size_t total = (size_t)cout * conv3x3_packed_ld(cin, bke);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    E2V_KLAUNCH(pack_conv3x3_kernel, dim3(blocks), dim3(256), 0, s, w, o, cout, cin, bke);
}

// fp32 -> three bf16 planes by truncation: x = p0 + p1 + p2 exactly (planes `plane` elements apart)
__global__ void split_bf16x3_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, size_t n, size_t plane) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = in[i];
        const unsigned b1 = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
        const float r1 = x - __builtin_bit_cast(float, b1);
        const unsigned b2 = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
        const float r2 = r1 - __builtin_bit_cast(float, b2);
        out[i] = (unsigned short)(b1 >> 16);
        out[plane + i] = (unsigned short)(b2 >> 16);
        out[2 * plane + i] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
    }
}
void split_bf16x3(const float* in, void* out, size_t n, size_t plane, hipStream_t s) {
    if (!n) return;
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    E2V_KLAUNCH(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, s, in, static_cast<unsigned short*>(out), n, plane);
}

template <typename H>
__global__ void to_h16_kernel(const float* __restrict__ in, H* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (H)in[i];
}
void to_h16(const float* in, void* out, size_t n, int mode, hipStream_t s) {
    if (!n) return;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    h16_dispatch(mode, [&](auto h16_tag) {
        using H = decltype(h16_tag);
        E2V_KLAUNCH(to_h16_kernel<H>, dim3(blocks), dim3(256), 0, s, in, static_cast<H*>(out), n);
    });
}
void to_bf16(const float* in, void* out, size_t n, hipStream_t s) { to_h16(in, out, n, H16_BF16, s); }

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
    E2V_KLAUNCH(copy_rows_kernel, dim3(blocks), dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
}

}  // namespace e2v
