// bf16-activation implicit-GEMM convolution / linear on the gfx950 matrix cores (BASELINE configs[2]).
//
// Same contract as igemm.hip -- out[m][n] = epilogue(sum_{tap,c} A[src(m,tap)][c] W[n][k(tap,c)]), two channel-last sources,
// the conv geometry folded into an LDS gather table -- but the activations are bf16 IN HBM and the product is
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  What changes with the operand width:
//   * no conversion anywhere in the loop: a k-stage is 64 channels = 128 bytes per tile row, moved HBM/L2 -> LDS by LDS-DMA
//     (buffer_load_dwordx4 ... lds: 64 lanes x 16 bytes = eight 128-byte tile rows per wave-instruction, no VGPR round trip, no
//     ds_write); the per-lane SOURCE address carries the gather (pixel row from the table, zero padding / ragged rows and
//     channels as out-of-window offsets that the buffer unit turns into zeros);
//   * LDS rows are exactly 128 bytes (a DMA piece is 1 KB contiguous, so rows cannot be padded); bank conflicts of the
//     ds_read_b128 fragment reads are removed by an XOR swizzle of the 16-byte chunk index with (row >> 1) & 7, applied on
//     the source side of the DMA and on the read side (cdna_hip_programming.md rule 21);
//   * two LDS stages, the next stage's DMA in flight while this one is multiplied; one barrier per 64-deep stage.
// Tiles 128x128 and 128x64 (4 waves as 2x2), dealt to the XCDs by the same schedule as the fp32 kernel (igemm.hip).
#include "igemm_epi.h"
#include "prof.h"

#include <cstdlib>
#include <string>

namespace e2v {

// LIN: taps == 1 (linear / 1x1 conv).  Rows are their own pixels, so there is no gather table, a lane's source offsets never
// change, and a k-step is: eight LDS-DMA instructions whose k position rides in the scalar offset, one scalar add -- the
// general (3x3) path spends ~130 scalar + vector instructions per k-step on the gather, which an in-order wave pays in issue
// slots next to its 16 MFMAs.
template <int BM, int BN, int WGM, int WGN, int STAGE = 128 * 128 * 2, bool LIN = false>      // STAGE: stage stride, sized for the largest tile of the launch
__device__ __forceinline__ void bgemm_tile(const IgemmArgs& p, const int rbg, const int n0, char* smem) {
    constexpr int BKE = 64;                         // bf16 elements per stage
    constexpr int ROWB = 128;                       // bytes per LDS tile row
    constexpr int NW = WGM * WGN, NT = 64 * NW;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int APW = BM / 8 / NW, BPW = BN / 8 / NW;   // 1-KB DMA pieces (8 rows) per wave and stage
    static_assert(APW >= 1 && BPW >= 1 && A_BYTES + B_BYTES <= STAGE, "tile does not fit the stage");
    const int z = p.batch > 1 ? rbg / p.nbm_per : 0;
    const int bm = rbg - z * p.nbm_per;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    const __bf16* __restrict__ a0 = reinterpret_cast<const __bf16*>(p.a0) + (size_t)z * p.sa0;
    const __bf16* __restrict__ a1 = reinterpret_cast<const __bf16*>(p.a1);
    const char* __restrict__ w = reinterpret_cast<const char*>(p.w16) + ((size_t)z * p.sw + (size_t)n0 * p.ldw) * 2;

    const int steps0 = (p.c0 + BKE - 1) / BKE, steps1 = (p.c1 + BKE - 1) / BKE;
    const int nk = p.taps * (steps0 + steps1);

    // gather table (see igemm.hip): source pixel of (tap, tile row), block-relative; ~0u = zero padding / row >= M
    unsigned* tab = reinterpret_cast<unsigned*>(smem + 2 * STAGE);
    const int hw_out = p.Ho * p.Wo, hw_in = p.Hs * p.Ws;
    const int img0 = p.taps == 1 ? 0 : (bm * BM) / hw_out;
    const size_t row_base = p.taps == 1 ? (size_t)bm * BM : (size_t)img0 * hw_in;
    for (int e = tid; !LIN && e < p.taps * BM; e += NT) {
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
    constexpr unsigned OOB = 0x80000000u;                   // beyond the descriptor window: the buffer unit returns zeros
    const __bf16* const a0b = a0 + row_base * p.lda0;
    const __bf16* const a1b = p.c1 > 0 ? a1 + row_base * p.lda1 : a0b;
    const int a_records = (p.ablate & 2) ? 0 : 0x7FFFFFF0;      // timing experiment: a zero-record descriptor drops every A load
    auto rsrc_of = [](const void* ptr, const int records = 0x7FFFFFF0) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), (short)0, records,
                                                 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rw = rsrc_of(w);

    // DMA geometry of this lane: piece q covers tile rows 8q .. 8q+7, lane -> (row 8q + (lane >> 3), LDS chunk lane & 7); the
    // chunk FETCHED for LDS position p of row r is p ^ ((r >> 1) & 7)
    const int r8 = lane >> 3, pp = lane & 7;
    int a_row[APW];
    unsigned a_kc[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        a_row[i] = 8 * (wave * APW + i) + r8;
        a_kc[i] = (unsigned)(pp ^ ((a_row[i] >> 1) & 7));
    }
    unsigned b_off[BPW], b_kc[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
        const int r = 8 * (wave * BPW + j) + r8;
        b_kc[j] = (unsigned)(pp ^ ((r >> 1) & 7));
        b_off[j] = (n0 + r < p.N) ? (unsigned)(r * p.ldw * 2) + b_kc[j] * 16u : OOB;
    }
    unsigned a_voff0[APW], a_voff1[APW];            // LIN: the lane's fixed source offsets into the two sources
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const bool in = bm * BM + a_row[i] < p.M;
        a_voff0[i] = in ? (unsigned)(a_row[i] * p.lda0 * 2) + a_kc[i] * 16u : OOB;
        a_voff1[i] = in ? (unsigned)(a_row[i] * p.lda1 * 2) + a_kc[i] * 16u : OOB;
    }
    if constexpr (!LIN) __syncthreads();

    int k_src = 0, k_cb = 0, k_tap = 0, cseg = p.c0, ldb = p.lda0 * 2;
    bool done = false;
    unsigned pixn[APW];
    if constexpr (!LIN) {
#pragma unroll
        for (int i = 0; i < APW; ++i) pixn[i] = tab[a_row[i]];
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue = [&](const int buf) {
        char* Ab = smem + buf * STAGE;
        char* Bb = Ab + A_BYTES;
        if constexpr (LIN) {
            if (!done) {                                        // wave-uniform
                const unsigned so = (unsigned)k_cb * 2u;
                const unsigned sob = (unsigned)((k_src ? p.c0 : 0) + k_cb) * 2u;
                const bool whole = k_cb + BKE <= cseg;          // the stage lies inside the segment: no channel masks
                const __amdgpu_buffer_rsrc_t rsa = rsrc_of(k_src ? a1b : a0b, a_records);
#pragma unroll
                for (int i = 0; i < APW; ++i) {
                    const unsigned vo = k_src ? a_voff1[i] : a_voff0[i];
                    const unsigned off = (whole || k_cb + (int)a_kc[i] * 8 < cseg) ? vo : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)(Ab + (wave * APW + i) * 1024), 16, off, so, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < BPW; ++j) {
                    const unsigned off = (whole || k_cb + (int)b_kc[j] * 8 < cseg) ? b_off[j] : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(Bb + (wave * BPW + j) * 1024), 16, off, sob, 0, 0);
                }
                k_cb += BKE;
                if (k_cb >= cseg) {
                    if (k_src == 0 && p.c1 > 0) { k_src = 1; k_cb = 0; cseg = p.c1; }
                    else done = true;
                }
            }
            return;
        }
        if (!done) {                                            // wave-uniform
            const __amdgpu_buffer_rsrc_t rsa = rsrc_of(k_src ? a1b : a0b);
            const unsigned colb = (unsigned)k_cb * 2u;
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const bool ok = (pixn[i] != ~0u) & (k_cb + (int)a_kc[i] * 8 < cseg);
                const unsigned off = ok ? __umul24(pixn[i], (unsigned)ldb) + colb + a_kc[i] * 16u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)(Ab + (wave * APW + i) * 1024), 16, off, 0, 0, 0);
            }
            const int cbase = (k_src ? p.c0 : 0) + k_cb;                               // channel of this k-step in the concat
            const int koffb = (p.taps == 1 ? cbase : (cbase / BKE * 9 + k_tap) * BKE) * 2;   // wave-uniform: rides in soffset
#pragma unroll
            for (int j = 0; j < BPW; ++j) {
                const bool ok = k_cb + (int)b_kc[j] * 8 < cseg;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(Bb + (wave * BPW + j) * 1024), 16, ok ? b_off[j] : OOB, koffb, 0, 0);
            }
        }
        // advance (tap fastest, then chunk, then source)
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
        ldb = (k_src ? p.lda1 : p.lda0) * 2;
#pragma unroll
        for (int i = 0; i < APW; ++i) pixn[i] = tab[k_tap * BM + a_row[i]];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment addressing: lane -> tile row (lane & 31) of each 32-row MFMA block, k chunk 2 g + (lane >> 5) of the stage
    const int fr = lane & 31, fh = lane >> 5;
    const int fs = (fr >> 1) & 7;                   // the blocks start at multiples of 32 rows: they do not enter (row >> 1) & 7
    unsigned foff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) foff[g] = (unsigned)(((2 * g + fh) ^ fs) * 16);
    const char* Afr = smem + (wm * WM + fr) * ROWB;
    const char* Bfr = smem + A_BYTES + (wn * WN + fr) * ROWB;
    bf16x8 af[2][TM], bfr[2][TN];
    auto read_frags = [&](int set, int buf, int g) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
            af[set][mi] = *reinterpret_cast<const bf16x8*>(Afr + buf * STAGE + mi * 32 * ROWB + foff[g]);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
            bfr[set][ni] = *reinterpret_cast<const bf16x8*>(Bfr + buf * STAGE + ni * 32 * ROWB + foff[g]);
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[set][ni], af[set][mi], acc[mi][ni], 0, 0, 0);
    };

    issue(0);
    __syncthreads();                            // LDS-DMA is a pending LDS write: the fence waits for it (vmcnt(0))
    read_frags(0, 0, 0);
    auto kstep = [&](auto Pc) {
        constexpr int P = decltype(Pc)::value;
        issue(P ^ 1);                           // stage ks+1 (nothing behind the last one)
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, P, 1);
        mma(0);
        read_frags(0, P, 2);
        mma(1);
        read_frags(1, P, 3);
        mma(0);
        __syncthreads();                        // stage ks+1 has landed; every wave is done reading stage ks
        read_frags(0, P ^ 1, 0);                // first fragments of stage ks+1 ...
        __builtin_amdgcn_sched_barrier(0);
        mma(1);                                 // ... land while the last k-group of stage ks is multiplied
    };
    int ks = 0;
    for (; ks + 1 < nk; ks += 2) {
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
    }
    if (ks < nk) kstep(std::integral_constant<int, 0>{});

    if (p.out_f32) {
        float* __restrict__ out = p.out + (size_t)z * p.sout;
        igemm_epilogue<BM, TM, TN, WM, WN>(p, acc, out, bm, n0, wm, wn, lane, reinterpret_cast<float*>(smem));
    } else {
        __bf16* __restrict__ out = reinterpret_cast<__bf16*>(p.out) + (size_t)z * p.sout;
        bgemm_epilogue_bf16<BM, TM, TN, WM, WN>(p, acc, out, bm, n0, wm, wn, lane, reinterpret_cast<float*>(smem));
    }
}

// ---- taps == 1, four-stage ring (E2V_BGEMM_RING) --------------------------------------------------------------------------
// The two-stage tile above hands its loads one k-step of cover; a linear layer streams its activations straight from HBM
// (K = 320: five stages in all), and PMC counters put the waves of such a launch 56 % of their life in s_waitcnt / s_barrier.
// Here a stage is 32 channels (64-byte tile rows: a DMA piece is 16 rows, chunk swizzle (row >> 2) & 3), four stages ring
// through 64 KB, and the loads of stage j + 3 are issued as soon as stage j - 1 has been consumed: three stages are always in
// flight per workgroup, the consumer waits with a COUNTED vmcnt (the two youngest stages stay outstanding) and a raw
// s_barrier -- no drain in the loop.
template <int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void bgemm_ring_tile(const IgemmArgs& p, const int rbg, const int n0, char* smem) {
    constexpr int BKE = 32, ROWB = 64, NS = 4, PF = 3;
    constexpr int NW = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_BYTES = BM * ROWB, STAGE = 128 * ROWB * 2;
    constexpr int APW = BM / 16 / NW, BPW = BN / 16 / NW;     // 1-KB DMA pieces (16 rows) per wave and stage
    constexpr int PIECES = APW + BPW;
    static_assert(APW >= 1 && BPW >= 1, "tile too small for the DMA split");
    const int z = p.batch > 1 ? rbg / p.nbm_per : 0;
    const int bm = rbg - z * p.nbm_per;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const __bf16* __restrict__ a0 = reinterpret_cast<const __bf16*>(p.a0) + (size_t)z * p.sa0;
    const __bf16* __restrict__ a1 = reinterpret_cast<const __bf16*>(p.a1);
    const char* __restrict__ w = reinterpret_cast<const char*>(p.w16) + ((size_t)z * p.sw + (size_t)n0 * p.ldw) * 2;
    const int nk = (p.c0 + BKE - 1) / BKE + (p.c1 + BKE - 1) / BKE;
    constexpr unsigned OOB = 0x80000000u;
    const size_t row_base = (size_t)bm * BM;
    const __bf16* const a0b = a0 + row_base * p.lda0;
    const __bf16* const a1b = p.c1 > 0 ? a1 + row_base * p.lda1 : a0b;
    auto rsrc_of = [](const void* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), (short)0, 0x7FFFFFF0,
                                                 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rw = rsrc_of(w);
    const int r16 = lane >> 2, pp = lane & 3;
    unsigned a_voff0[APW], a_voff1[APW], a_kc[APW], b_off[BPW], b_kc[BPW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int r = 16 * (wave * APW + i) + r16;
        a_kc[i] = (unsigned)(pp ^ ((r >> 2) & 3));
        const bool in = bm * BM + r < p.M;
        a_voff0[i] = in ? (unsigned)(r * p.lda0 * 2) + a_kc[i] * 16u : OOB;
        a_voff1[i] = in ? (unsigned)(r * p.lda1 * 2) + a_kc[i] * 16u : OOB;
    }
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
        const int r = 16 * (wave * BPW + j) + r16;
        b_kc[j] = (unsigned)(pp ^ ((r >> 2) & 3));
        b_off[j] = (n0 + r < p.N) ? (unsigned)(r * p.ldw * 2) + b_kc[j] * 16u : OOB;
    }
    int k_src = 0, k_cb = 0, cseg = p.c0;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue = [&](const int buf) {                       // one stage; never called past the last one
        char* Ab = smem + buf * STAGE;
        char* Bb = Ab + A_BYTES;
        const unsigned so = (unsigned)k_cb * 2u;
        const unsigned sob = (unsigned)((k_src ? p.c0 : 0) + k_cb) * 2u;
        const bool whole = k_cb + BKE <= cseg;
        // the descriptor is rebuilt from a pointer forced into SGPRs: a select between two ready-made descriptors is not provably
        // wave-uniform to hipcc, which then parks them in scratch and wraps every load in a waterfall loop (with a vmcnt(0))
        const __amdgpu_buffer_rsrc_t rsa = rsrc_of(k_src ? a1b : a0b);
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const unsigned vo = k_src ? a_voff1[i] : a_voff0[i];
            const unsigned off = (whole || k_cb + (int)a_kc[i] * 8 < cseg) ? vo : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)(Ab + (wave * APW + i) * 1024), 16, off, so, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const unsigned off = (whole || k_cb + (int)b_kc[j] * 8 < cseg) ? b_off[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(Bb + (wave * BPW + j) * 1024), 16, off, sob, 0, 0);
        }
        k_cb += BKE;
        if (k_cb >= cseg && k_src == 0 && p.c1 > 0) { k_src = 1; k_cb = 0; cseg = p.c1; }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    const int fs = (fr >> 2) & 3;
    const unsigned foff0 = (unsigned)(((0 + fh) ^ fs) * 16), foff1 = (unsigned)(((2 + fh) ^ fs) * 16);
    const char* Afr = smem + (wm * WM + fr) * ROWB;
    const char* Bfr = smem + A_BYTES + (wn * WN + fr) * ROWB;

    int issued = 0;
    for (; issued < PF && issued < nk; ++issued) issue(issued);
    for (int j = 0; j < nk; ++j) {
        const int left = nk - j - 1;                            // stages behind this one: min(left, PF - 1) of them are in flight
        if (left >= PF - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * PIECES) : "memory");
        else if (left == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // stage j landed for every wave; stage j - 1 is consumed
        __builtin_amdgcn_sched_barrier(0);
        if (issued < nk) { issue(issued & (NS - 1)); ++issued; }
        const int bo = (j & (NS - 1)) * STAGE;
        bf16x8 af0[TM], bf0[TN], af1[TM], bf1[TN];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) af0[mi] = *reinterpret_cast<const bf16x8*>(Afr + bo + mi * 32 * ROWB + foff0);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) bf0[ni] = *reinterpret_cast<const bf16x8*>(Bfr + bo + ni * 32 * ROWB + foff0);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) af1[mi] = *reinterpret_cast<const bf16x8*>(Afr + bo + mi * 32 * ROWB + foff1);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) bf1[ni] = *reinterpret_cast<const bf16x8*>(Bfr + bo + ni * 32 * ROWB + foff1);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf0[ni], af0[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf1[ni], af1[mi], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();                                            // the ring becomes the epilogue's staging area
    if (p.out_f32) {
        float* __restrict__ out = p.out + (size_t)z * p.sout;
        igemm_epilogue<BM, TM, TN, WM, WN>(p, acc, out, bm, n0, wm, wn, lane, reinterpret_cast<float*>(smem));
    } else {
        __bf16* __restrict__ out = reinterpret_cast<__bf16*>(p.out) + (size_t)z * p.sout;
        bgemm_epilogue_bf16<BM, TM, TN, WM, WN>(p, acc, out, bm, n0, wm, wn, lane, reinterpret_cast<float*>(smem));
    }
}

__global__ __launch_bounds__(256) void bgemm_ring_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_ring[];
    const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int nrb = rb_hi - rb_lo;
    const int tail = min(nrb, p.tail_rb);
    const int per1 = p.w1 + p.s1;
    const int n1 = (nrb - tail) * per1;
    if (loc < n1) {
        const int r = loc / per1, j = loc - r * per1;
        if (j < p.w1) bgemm_ring_tile<128, 128, 2, 2>(p, rb_lo + r, j * 128, smem_ring);
        else bgemm_ring_tile<128, 64, 2, 2>(p, rb_lo + r, p.w1 * 128 + (j - p.w1) * 64, smem_ring);
    } else {
        const int t = loc - n1;
        if (t >= tail * p.s2) return;
        const int r = t / p.s2;
        bgemm_ring_tile<128, 64, 2, 2>(p, rb_lo + (nrb - tail) + r, (t - r * p.s2) * 64, smem_ring);
    }
}

template <bool LIN>
__global__ __launch_bounds__(256) void bgemm_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_bg[];
    const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int nrb = rb_hi - rb_lo;
    const int tail = min(nrb, p.tail_rb);
    const int per1 = p.w1 + p.s1;
    const int n1 = (nrb - tail) * per1;
    if (loc < n1) {
        const int r = loc / per1, j = loc - r * per1;
        if (j < p.w1) bgemm_tile<128, 128, 2, 2, 128 * 128 * 2, LIN>(p, rb_lo + r, j * 128, smem_bg);
        else bgemm_tile<128, 64, 2, 2, 128 * 128 * 2, LIN>(p, rb_lo + r, p.w1 * 128 + (j - p.w1) * 64, smem_bg);
    } else {
        const int t = loc - n1;
        if (t >= tail * p.s2) return;
        const int r = t / p.s2;
        bgemm_tile<128, 64, 2, 2, 128 * 128 * 2, LIN>(p, rb_lo + (nrb - tail) + r, (t - r * p.s2) * 64, smem_bg);
    }
}

// 256-row tiles: 8 waves (4 x 2, the same 64 x 64 wave tile), 256 x 128 and 256 x 64, 48 KB stages, one workgroup per CU.
template <bool LIN>
__global__ __launch_bounds__(512) void bgemm256_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_bg256[];
    constexpr int ST = (256 + 128) * 128;
    const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int nrb = rb_hi - rb_lo;
    const int tail = min(nrb, p.tail_rb);
    const int per1 = p.w1 + p.s1;
    const int n1 = (nrb - tail) * per1;
    if (loc < n1) {
        const int r = loc / per1, j = loc - r * per1;
        if (j < p.w1) bgemm_tile<256, 128, 4, 2, ST, LIN>(p, rb_lo + r, j * 128, smem_bg256);
        else bgemm_tile<256, 64, 4, 2, ST, LIN>(p, rb_lo + r, p.w1 * 128 + (j - p.w1) * 64, smem_bg256);
    } else {
        const int t = loc - n1;
        if (t >= tail * p.s2) return;
        const int r = t / p.s2;
        bgemm_tile<256, 64, 4, 2, ST, LIN>(p, rb_lo + (nrb - tail) + r, (t - r * p.s2) * 64, smem_bg256);
    }
}

bool bgemm_use_256(const IgemmArgs& a) {
    static const int mode = [] { const char* e = std::getenv("E2V_BGEMM_256"); return e ? std::atoi(e) : 1; }();   // 0: never, 2: always
    if (mode == 0) return false;
    if (mode == 2) return true;
    // 3x3 convs only: their long k-loop (K >= 2880) is bound by the LDS fill rate, which the bigger tile relieves (+5-10 %);
    // the linears' short k-loops lean on a second resident workgroup to cover prologue and epilogue (256-row tiles: -6..-20 %)
    if (a.taps == 1) return false;
    // at least four rounds of 256 resident tiles, else the finer 128-row grid wastes less on its last round
    const double tiles = (double)((a.M + 255) / 256) * a.batch * ((a.N + 127) / 128);
    return tiles >= 4.0 * 256;
}

// Launches whose tiles are all 128 x 64 (N <= 64, grids below one round, and -- E2V_BGEMM_N64_MAXK -- short-K layers): 24 KB
// stages, 53 KB of LDS per workgroup, so THREE workgroups share a CU.  A short-K tile (K = 320: five stages) spends most of its
// life waiting for its first HBM bytes; a third resident workgroup is one more tile's worth of loads in flight per CU.
template <bool LIN>
__global__ __launch_bounds__(256) void bgemm_n64_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_bg64[];
    const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int nrb = rb_hi - rb_lo;
    if (loc >= nrb * p.s2) return;
    const int r = loc / p.s2;
    bgemm_tile<128, 64, 2, 2, (128 + 64) * 128, LIN>(p, rb_lo + r, (loc - r * p.s2) * 64, smem_bg64);
}

bool bgemm_all_n64(const IgemmArgs& a) {
    static const int maxk = [] { const char* e = std::getenv("E2V_BGEMM_N64_MAXK"); return e ? std::atoi(e) : 0; }();
    static const int conv = [] { const char* e = std::getenv("E2V_BGEMM_N64_CONV"); return e ? std::atoi(e) : 0; }();
    if (a.taps != 1) return conv != 0;
    return (a.c0 + a.c1) <= maxk && !a.geglu;
}

void bgemm_launch(const IgemmArgs& a, int ntiles, hipStream_t s) {
    constexpr size_t smem = (size_t)2 * 128 * 128 * 2 + 9 * 128 * sizeof(unsigned);     // two stages of (128 + 128) rows + gather table
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bgemm_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bgemm_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        configured = true;
    }
    static const int lean = [] { const char* e = std::getenv("E2V_BGEMM_LIN"); return e ? std::atoi(e) : 1; }();   // 0: linears through the gather path
    const bool lin = a.taps == 1 && lean;
    static const int ring = [] { const char* e = std::getenv("E2V_BGEMM_RING"); return e ? std::atoi(e) : 0; }();
    if (ring && a.taps == 1 && !a.bm256) {
        constexpr size_t smem_r = (size_t)4 * 128 * 64 * 2;                               // four stages of (128 + 128) rows x 64 bytes
        static bool cfgr = false;
        if (!cfgr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bgemm_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_r);
            cfgr = true;
        }
        const double Kr = (double)(a.c0 + a.c1);
        std::string rn = "igemm_bf16";
        if (profiler().on && profiler().detail)
            rn += " M" + std::to_string(a.M) + " N" + std::to_string(a.N) + " K" + std::to_string((long)Kr) + " t1" + (a.c1 ? " cat" : "") +
                  (a.geglu ? " geglu" : "") + (a.batch > 1 ? " b" + std::to_string(a.batch) : "");
        ProfScope psr(rn.c_str(), 2.0 * a.M * a.N * Kr * a.batch,
                      a.batch * (2.0 * a.M * Kr + 2.0 * a.N * Kr + (a.out_f32 ? 4.0 : 2.0) * a.M * (a.geglu ? a.N / 2 : a.N)), s);
        hipLaunchKernelGGL(bgemm_ring_kernel, dim3(ntiles, 1, 1), dim3(256), smem_r, s, a);
        return;
    }
    const double K = (double)a.taps * (a.c0 + a.c1);
    const double rows_in = a.taps == 1 ? (double)a.M : (double)a.M * a.Hs * a.Ws / ((double)a.Ho * a.Wo);
    std::string pname = "igemm_bf16";
    if (profiler().on && profiler().detail)
        pname += " M" + std::to_string(a.M) + " N" + std::to_string(a.N) + " K" + std::to_string((long)K) + " t" + std::to_string(a.taps) +
                 (a.stride > 1 ? " s2" : "") + (a.upsample ? " up" : "") + (a.c1 ? " cat" : "") + (a.geglu ? " geglu" : "") +
                 (a.batch > 1 ? " b" + std::to_string(a.batch) : "");
    static const int ablate = [] { const char* e = std::getenv("E2V_BGEMM_ABLATE"); return e ? std::atoi(e) : 0; }();
    const_cast<IgemmArgs&>(a).ablate = ablate;
    const double out_b = a.out_f32 ? 4.0 : 2.0;
    ProfScope ps(pname.c_str(), 2.0 * a.M * a.N * K * a.batch,
                 a.batch * (2.0 * rows_in * (a.c0 + a.c1) + 2.0 * a.N * K + out_b * a.M * (a.geglu ? a.N / 2 : a.N)), s);
    if (a.bm256) {
        constexpr size_t smem256 = (size_t)2 * (256 + 128) * 128 + 9 * 256 * sizeof(unsigned);
        static bool cfg256 = false;
        if (!cfg256) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bgemm256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem256);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bgemm256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem256);
            cfg256 = true;
        }
        if (lin) hipLaunchKernelGGL(bgemm256_kernel<true>, dim3(ntiles, 1, 1), dim3(512), smem256, s, a);
        else hipLaunchKernelGGL(bgemm256_kernel<false>, dim3(ntiles, 1, 1), dim3(512), smem256, s, a);
        return;
    }
    if (a.rb1 == 0 && !a.geglu) {          // every row block is cut into 128 x 64 tiles only
        constexpr size_t smem64 = (size_t)2 * (128 + 64) * 128 + 9 * 128 * sizeof(unsigned);
        static bool cfg64 = false;
        if (!cfg64) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bgemm_n64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem64);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bgemm_n64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem64);
            cfg64 = true;
        }
        int nt = 0;
        for (int x = 0; x < 8; ++x) {
            const int nrb = (int)(((long)(x + 1) * a.nbm) >> 3) - (int)(((long)x * a.nbm) >> 3);
            nt = nrb * a.s2 > nt ? nrb * a.s2 : nt;
        }
        if (lin) hipLaunchKernelGGL(bgemm_n64_kernel<true>, dim3(nt * 8, 1, 1), dim3(256), smem64, s, a);
        else hipLaunchKernelGGL(bgemm_n64_kernel<false>, dim3(nt * 8, 1, 1), dim3(256), smem64, s, a);
        return;
    }
    if (lin) hipLaunchKernelGGL(bgemm_kernel<true>, dim3(ntiles, 1, 1), dim3(256), smem, s, a);
    else hipLaunchKernelGGL(bgemm_kernel<false>, dim3(ntiles, 1, 1), dim3(256), smem, s, a);
}

}  // namespace e2v
