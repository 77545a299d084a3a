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
#include "runtime.h"

#include <cstdio>
#include <cstdlib>
#include <string>

#include <map>
#include <mutex>

namespace e2v {

static std::mutex& knob_mutex() { static std::mutex m; return m; }
static std::map<std::string, int>& knob_table() { static std::map<std::string, int> t; return t; }
int* knob(const char* name, int dflt) {
    std::lock_guard<std::mutex> lk(knob_mutex());
    auto& t = knob_table();
    auto it = t.find(name);
    if (it == t.end()) {
        const char* e = std::getenv(name);
        it = t.emplace(name, e ? std::atoi(e) : dflt).first;
    }
    return &it->second;                          // std::map nodes do not move
}
bool set_knob(const char* name, int value) {
    std::lock_guard<std::mutex> lk(knob_mutex());
    auto& t = knob_table();
    auto it = t.find(name);
    if (it == t.end()) {
        static const char* const known[] = {"E2V_BGEMM_PERS", "E2V_BGEMM_256LIN", "E2V_BGEMM_256", "E2V_BGEMM_T256", "E2V_BGEMM_T256P", "E2V_BGEMM_T256P_BIAS_LDS",
                                            "E2V_ATTN_KT64", "E2V_ATTN_Q64", "E2V_ATTN_CROSS_RESIDENT", "E2V_TATTN_WAVE", "E2V_LN_ROWS", "E2V_BGEMM_UP2X",
                                            "E2V_SPLITK", "E2V_SPLITK_FORCE", "E2V_GN_FUSED_SMALL", "E2V_BGEMM_S3_SMALL", "E2V_SMALL_FAMILY_CLIPS"
#ifdef E2V_AB                                // variants measured and not adopted / the other arm of an A/B / thresholds that measured +-0: `make AB=1` builds only
                                            , "E2V_BGEMM_S3", "E2V_BGEMM_LIN", "E2V_BGEMM_T256_TAIL", "E2V_ATTN_FOLD", "E2V_ATTN_Q64P", "E2V_ATTN_Q64_NW", "E2V_GN_ROWS", "E2V_GN_GROUP_MB", "E2V_GN_SKIP_PARTIAL", "E2V_GN_RB", "E2V_GN_RB_EPILOGUE",
                                            "E2V_BGEMM_256_MINROUNDS", "E2V_BGEMM_T256_MINK", "E2V_BGEMM_T256_MINTILES", "E2V_BGEMM_T256P_MAXK", "E2V_BGEMM_T256P_MINTILES", "E2V_GN_CHUNK_ROWS",
                                            "E2V_GN_CHUNK_ROWS_SMALL", "E2V_IGEMM_HALF_BELOW", "E2V_SPLITK_MIN_DEPTH", "E2V_SPLITK_MAX_TILES", "E2V_GN_COOP", "E2V_LN_STATS_ONLY"
#endif
#ifdef E2V_ABLATE
                                            , "E2V_BGEMM_ABLATE"
#endif
        };
        bool ok = false;
        for (const char* k : known) ok = ok || std::string(k) == name;
        if (!ok) return false;
        it = t.emplace(name, value).first;
    }
    it->second = value;
    return true;
}

// LIN: taps == 1 (linear / 1x1 conv).  Rows are their own pixels, so there is no gather table, a lane's source offsets never
// change, and a k-step is: eight LDS-DMA instructions whose k position rides in the scalar offset, one scalar add -- the
// general (3x3) path spends ~130 scalar + vector instructions per k-step on the gather, which an in-order wave pays in issue
// slots next to its 16 MFMAs.
template <class F, int... T>
__device__ __forceinline__ void static_for_taps(F&& f, std::integer_sequence<int, T...>) {
    (f(std::integral_constant<int, T>{}), ...);
}

// NST: LDS stages.  2: the next stage's DMA is in flight while this one is multiplied.  3 (one 512-thread workgroup per CU: nothing
// else on the CU covers a wait): TWO stages in flight, the consumer waits with a counted vmcnt that leaves the younger one outstanding.
// (CALLER: a tag that only makes the specialisation distinct per calling kernel -- two kernels calling the SAME specialisation of this
// inlined template fail the host pass of hipcc 7.2 with "no matching function: substitution failure" at the second call site)
template <typename H, int BM, int BN, int WGM, int WGN, int STAGE = 128 * 128 * 2, bool LIN = false, int NST = 2, int CALLER = 0>      // H: bf16 / fp16 (h16.h); STAGE: stage stride, sized for the largest tile of the launch
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

    const H* __restrict__ a0 = reinterpret_cast<const H*>(p.a0) + (size_t)z * p.sa0;
    const H* __restrict__ a1 = reinterpret_cast<const H*>(p.a1);
    const char* __restrict__ w = reinterpret_cast<const char*>(p.w16) + ((size_t)z * p.sw + (size_t)n0 * p.ldw) * 2;

    const int steps0 = (p.c0 + BKE - 1) / BKE, steps1 = (p.c1 + BKE - 1) / BKE;
    const int nk = p.taps * (steps0 + steps1);

    // gather table (see igemm.hip): source pixel of (tap, tile row), block-relative; ~0u = zero padding / row >= M
    unsigned* tab = reinterpret_cast<unsigned*>(smem + NST * STAGE);
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
    const H* const a0b = a0 + row_base * p.lda0;
    const H* const a1b = p.c1 > 0 ? a1 + row_base * p.lda1 : a0b;
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
    hx8<H> af[2][TM], bfr[2][TN];
    auto read_frags = [&](int set, int buf, int g) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
            af[set][mi] = *reinterpret_cast<const hx8<H>*>(Afr + buf * STAGE + mi * 32 * ROWB + foff[g]);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
            bfr[set][ni] = *reinterpret_cast<const hx8<H>*>(Bfr + buf * STAGE + ni * 32 * ROWB + foff[g]);
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = mfma_32x32x16(bfr[set][ni], af[set][mi], acc[mi][ni]);
    };

    if constexpr (NST == 3 && LIN) {
        // linears: nothing but the DMA and the fragment reads touches LDS in the loop; buffers rotate at run time
        constexpr int PIECES = APW + BPW;                   // DMA instructions of one stage and wave
        issue(0);
        if (nk > 1) {
            issue(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                       // stage 0 has landed for every wave
        __builtin_amdgcn_sched_barrier(0);
        read_frags(0, 0, 0);
        int cb = 0;                                         // buffer of the stage being multiplied
        for (int ks = 0; ks < nk; ++ks) {
            const int nb1 = cb == 2 ? 0 : cb + 1, nb2 = nb1 == 2 ? 0 : nb1 + 1;
            if (ks + 2 < nk) issue(nb2);                    // into the buffer consumed at step ks - 1 (every wave is past that barrier)
            __builtin_amdgcn_sched_barrier(0);
            read_frags(1, cb, 1);
            mma(0);
            read_frags(0, cb, 2);
            mma(1);
            read_frags(1, cb, 3);
            mma(0);
            if (ks + 1 < nk) {
                // stage ks+1 has to have landed; stage ks+2 (if there is one) stays in flight
                if (ks + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIECES) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                read_frags(0, nb1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            mma(1);
            cb = nb1;
        }
        __syncthreads();                                    // the ring becomes the epilogue's staging area
    } else if constexpr (NST == 3) {
        // 3x3 convs only (taps == 9, general path): the k-loop is unrolled over the nine taps of a 64-channel chunk, so that the tap,
        // the ring buffer of every step (9 = 3 x 3) and hence every LDS address are compile-time constants, and the source pixels of
        // all nine taps sit in registers -- the loop reads no gather table (an LDS read behind an LDS-DMA makes the compiler wait
        // for that DMA, which would put the two-stages-in-flight ring back to one)
        static_assert(!LIN, "three-stage ring: 3x3 convs");
        constexpr int PIECES = APW + BPW;                   // DMA instructions of one stage and wave
        // per (tap, piece) the lane's complete byte offset into the current source -- pixel x row stride + its 16-byte chunk, or out of
        // window -- so that a load is issued with NO vector arithmetic: the chunk's channel offset rides in the scalar offset.  (The
        // in-order wave pays every vector instruction in issue slots next to its MFMAs: with all waits, barriers and A loads removed
        // the loop still ran at 0.61 matrix-pipe busy.)  Recomputed from the LDS table when the source changes (concat).
        unsigned aoff[9][APW];
        auto load_aoff = [&](const unsigned ld2) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < APW; ++i) {
                    const unsigned pix = tab[t * BM + a_row[i]];
                    aoff[t][i] = pix != ~0u ? __umul24(pix, ld2) + a_kc[i] * 16u : OOB;
                }
        };
        load_aoff((unsigned)ldb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        auto issue_t = [&](auto Tc, auto Bc) {
            constexpr int T = decltype(Tc)::value, BUF = decltype(Bc)::value;
            char* Ab = smem + BUF * STAGE;
            char* Bb = Ab + A_BYTES;
            const __amdgpu_buffer_rsrc_t rsa = rsrc_of(k_src ? a1b : a0b);
            const unsigned colb = (unsigned)k_cb * 2u;
            const int cbase = (k_src ? p.c0 : 0) + k_cb;
            const int koffb = ((cbase / BKE) * 9 + T) * BKE * 2;
            if (k_cb + BKE <= cseg) {                       // the chunk lies inside the segment (always, but for a 4- or 8-channel input)
#pragma unroll
                for (int i = 0; i < APW; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)(Ab + (wave * APW + i) * 1024), 16, aoff[T][i], colb, 0, 0);
#pragma unroll
                for (int j = 0; j < BPW; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(Bb + (wave * BPW + j) * 1024), 16, b_off[j], koffb, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < APW; ++i) {
                    const bool ok = k_cb + (int)a_kc[i] * 8 < cseg;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)(Ab + (wave * APW + i) * 1024), 16, ok ? aoff[T][i] : OOB, colb, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < BPW; ++j) {
                    const bool ok = k_cb + (int)b_kc[j] * 8 < cseg;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(Bb + (wave * BPW + j) * 1024), 16, ok ? b_off[j] : OOB, koffb, 0, 0);
                }
            }
            if constexpr (T == 8) {                         // next chunk (then next source)
                k_cb += BKE;
                if (k_cb >= cseg && k_src == 0 && p.c1 > 0) {
                    k_src = 1; k_cb = 0; cseg = p.c1; ldb = p.lda1 * 2;
                    load_aoff((unsigned)ldb);               // (the table is not written after the tile's first barrier)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        };
        const int ngroups = steps0 + steps1;                // nk = 9 * ngroups
        issue_t(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        issue_t(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        __builtin_amdgcn_s_barrier();                       // stage 0 has landed for every wave
        __builtin_amdgcn_sched_barrier(0);
        read_frags(0, 0, 0);
        for (int gi = 0; gi < ngroups; ++gi) {
            const bool lastg = gi + 1 == ngroups;
            auto step = [&](auto Tc) {
                constexpr int T = decltype(Tc)::value;
                constexpr int CB = T % 3, NB1 = (T + 1) % 3, NB2 = (T + 2) % 3;
                const bool has2 = T + 2 < 9 || !lastg, has1 = T + 1 < 9 || !lastg;
                // stage ks+2 into the buffer consumed at step ks-1 (every wave is past that step's barrier)
                if (has2) issue_t(std::integral_constant<int, (T + 2) % 9>{}, std::integral_constant<int, NB2>{});
                __builtin_amdgcn_sched_barrier(0);
                read_frags(1, CB, 1);
                mma(0);
                read_frags(0, CB, 2);
                mma(1);
                read_frags(1, CB, 3);
                mma(0);
                if (has1) {
                    // stage ks+1 has to have landed; stage ks+2 (if there is one) stays in flight
                    if (has2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIECES) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    read_frags(0, NB1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                mma(1);
            };
            static_for_taps(step, std::make_integer_sequence<int, 9>{});
        }
        __syncthreads();                                    // the ring becomes the epilogue's staging area
    } else {
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
    }

    if (p.out_f32) {
        float* __restrict__ out = p.out + (size_t)z * p.sout;
        igemm_epilogue<BM, TM, TN, WM, WN>(p, acc, out, bm, n0, wm, wn, lane, reinterpret_cast<float*>(smem));
    } else {
        H* __restrict__ out = reinterpret_cast<H*>(p.out) + (size_t)z * p.sout;
        bgemm_epilogue_bf16<BM, TM, TN, WM, WN>(p, acc, out, bm, n0, wm, wn, lane, reinterpret_cast<float*>(smem));
    }
}

// ---- persistent workgroups ---------------------------------------------------------------------------------------------------
// A tile of the kernels below pays, at full price and alone on its half of the CU: the workgroup launch, the table build, the
// latency of its first stage (HBM for a linear layer), and at the end the drain of its stores before the next workgroup may
// start.  Timing with the epilogue's stores dropped (E2V_BGEMM_ABLATE=1) puts that at 17-22 % of a 3x3 conv and 40-50 % of a
// K = 320 linear.  Here a workgroup stays resident and walks the tile list of its XCD (the same order as the launches above,
// slot s takes tiles s, s + slots, ...).  The LDS ring never stops: the last k-stage of tile i issues stage 0 of tile i + 1
// (whose gather table was built while tile i was being multiplied), the epilogue of tile i stages its accumulators through the
// stage buffer it has just consumed while that DMA is in flight, and its stores are never waited for.
struct BgTile {
    int rbg, n0, wide, valid;
};

__device__ __forceinline__ BgTile bgemm_decode(const IgemmArgs& p, const int x, const int loc) {
    BgTile t{0, 0, 0, 0};
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int nrb = rb_hi - rb_lo;
    const int tail = min(nrb, p.tail_rb);
    const int per1 = p.w1 + p.s1;
    const int n1 = (nrb - tail) * per1;
    if (loc < n1) {
        const int r = loc / per1, j = loc - r * per1;
        t.rbg = rb_lo + r;
        t.wide = j < p.w1;
        t.n0 = t.wide ? j * 128 : p.w1 * 128 + (j - p.w1) * 64;
        t.valid = 1;
    } else {
        const int q = loc - n1;
        if (q < tail * p.s2) {
            const int r = q / p.s2;
            t.rbg = rb_lo + (nrb - tail) + r;
            t.n0 = (q - r * p.s2) * 64;
            t.valid = 1;
        }
    }
    return t;
}

template <typename H, int BM, int WGM, bool LIN>
__device__ __forceinline__ void bgemm_pers_body(const IgemmArgs& p, char* smem, const int x, const int slot, const int nslots) {
    constexpr int BKE = 64, ROWB = 128, WGN = 2;
    constexpr bool PREFETCH = BM == 128;                      // next tile's first stage in flight under the epilogue
    constexpr int NW = WGM * WGN, NT = 64 * NW;
    constexpr int WM = BM / WGM, TM = WM / 32;
    constexpr int STAGE = (BM + 128) * ROWB, A_BYTES = BM * ROWB;
    constexpr int APW = BM / 8 / NW, BPWM = 128 / 8 / NW;     // 1-KB DMA pieces per wave and stage (B: of a 128-wide tile)
    constexpr int TABN = 9 * BM;
    static_assert(WM == 64 && APW >= 1 && BPWM >= 2, "wave tile is 64 rows");
    constexpr unsigned OOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    unsigned* const tabs = reinterpret_cast<unsigned*>(smem + 2 * STAGE);
    float* const brow = reinterpret_cast<float*>(tabs + 2 * TABN);      // bias, time-embedding rows of two samples: [3][128]
    const int steps0 = (p.c0 + BKE - 1) / BKE, steps1 = (p.c1 + BKE - 1) / BKE;
    const int nk = p.taps * (steps0 + steps1);
    const int hw_out = p.Ho * p.Wo, hw_in = p.Hs * p.Ws;
    const int a_records = (p.ablate & 2) ? 0 : 0x7FFFFFF0;
    auto rsrc_of = [](const void* ptr, const int records = 0x7FFFFFF0) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), (short)0, records,
                                                 0x00020000);
    };

    // ---- issue side: the tile whose stages are being fetched ----------------------------------------------------------------
    // DMA geometry of a lane: piece q covers tile rows 8q .. 8q+7, lane -> (row 8q + (lane >> 3), LDS chunk lane & 7); the chunk
    // FETCHED for LDS position c of row r is c ^ ((r >> 1) & 7) = kc0 ^ 4 (q & 1): one register and a wave-uniform bit
    const int r8 = lane >> 3, pp = lane & 7;
    const unsigned kc0 = (unsigned)(pp ^ (r8 >> 1));
    auto a_row = [&](const int i) { return 8 * (wave * APW + i) + r8; };
    auto a_kc = [&](const int i) { return kc0 ^ (unsigned)((((wave * APW + i) & 1)) << 2); };
    const H* i_a0b = nullptr;
    const H* i_a1b = nullptr;
    const char* i_w = nullptr;
    const unsigned* i_tab = tabs;
    int i_bpw = BPWM;
    unsigned a_voff0[APW], b_off[BPWM], pixn[APW];
    auto b_kc = [&](const int j) { return kc0 ^ (unsigned)((((wave * i_bpw + j) & 1)) << 2); };
    // LDS piece of this wave's j-th B instruction; the instructions a 64-wide tile does not need (all lanes out of window: they
    // write zeros) are pointed at the unused upper half of the B area
    auto b_piece = [&](const int j) { return j < i_bpw ? wave * i_bpw + j : 128 / 16 + wave * (BPWM - i_bpw) + (j - i_bpw); };
    int k_src = 0, k_cb = 0, k_tap = 0, cseg = p.c0, ldb = p.lda0 * 2;
    bool i_live = false;                                     // a tile is being fetched and has stages left

    // gather table of tile t into table buffer `which` (3x3: all threads take part; the caller orders it against the first use
    // with a barrier)
    auto build_table = [&](const BgTile t, const int which) {
        if constexpr (!LIN) {
            const int z = p.batch > 1 ? t.rbg / p.nbm_per : 0;
            const int bm = t.rbg - z * p.nbm_per;
            const int img0 = p.taps == 1 ? 0 : (bm * BM) / hw_out;
            unsigned* tab = tabs + which * TABN;
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
    };
    // point the issue side at tile t, whose table sits in buffer `which`
    auto aim = [&](const BgTile t, const int which) {
        const int z = p.batch > 1 ? t.rbg / p.nbm_per : 0;
        const int bm = t.rbg - z * p.nbm_per;
        const H* a0 = reinterpret_cast<const H*>(p.a0) + (size_t)z * p.sa0;
        const H* a1 = reinterpret_cast<const H*>(p.a1);
        i_w = reinterpret_cast<const char*>(p.w16) + ((size_t)z * p.sw + (size_t)t.n0 * p.ldw) * 2;
        const int img0 = p.taps == 1 ? 0 : (bm * BM) / hw_out;
        const size_t row_base = p.taps == 1 ? (size_t)bm * BM : (size_t)img0 * hw_in;
        i_a0b = a0 + row_base * p.lda0;
        i_a1b = p.c1 > 0 ? a1 + row_base * p.lda1 : i_a0b;
        i_bpw = t.wide ? BPWM : BPWM / 2;
        i_tab = tabs + which * TABN;
#pragma unroll
        for (int j = 0; j < BPWM; ++j) {
            const int r = 8 * (wave * i_bpw + j) + r8;
            b_off[j] = (j < i_bpw && t.n0 + r < p.N) ? (unsigned)(r * p.ldw * 2) + b_kc(j) * 16u : OOB;
        }
#pragma unroll
        for (int i = 0; i < APW; ++i) a_voff0[i] = bm * BM + a_row(i) < p.M ? (unsigned)(a_row(i) * p.lda0 * 2) + a_kc(i) * 16u : OOB;
        k_src = 0, k_cb = 0, k_tap = 0, cseg = p.c0, ldb = p.lda0 * 2;
        i_live = true;
    };
    auto first_pixels = [&]() {                             // after the table of the aimed tile is visible
        if constexpr (!LIN) {
#pragma unroll
            for (int i = 0; i < APW; ++i) pixn[i] = i_tab[a_row(i)];
        }
    };
    // one stage of the aimed tile into the buffer at byte offset bo; clears i_live behind the tile's last stage
    auto issue = [&](const int bo) {
        char* Ab = smem + bo;
        char* Bb = Ab + A_BYTES;
        if constexpr (LIN) {
            const unsigned so = (unsigned)k_cb * 2u;
            const unsigned sob = (unsigned)((k_src ? p.c0 : 0) + k_cb) * 2u;
            const bool whole = k_cb + BKE <= cseg;          // the stage lies inside the segment: no channel masks
            const __amdgpu_buffer_rsrc_t rsa = rsrc_of(k_src ? i_a1b : i_a0b, a_records);
            const __amdgpu_buffer_rsrc_t rw = rsrc_of(i_w);
            // every call issues the same APW + BPWM instructions (no tile behind this one: all of them out of window; a 64-wide
            // tile: the unused B pieces likewise): the compiler's vmcnt bookkeeping then counts the epilogue's waits exactly
            // instead of draining the queue
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                // second source (concat): same rows, its own row stride
                const unsigned vo = !k_src ? a_voff0[i] : a_voff0[i] == OOB ? OOB : (unsigned)(a_row(i) * p.lda1 * 2) + a_kc(i) * 16u;
                const unsigned off = (i_live && (whole || k_cb + (int)a_kc(i) * 8 < cseg)) ? vo : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)(Ab + (wave * APW + i) * 1024), 16, off, so, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < BPWM; ++j) {
                const unsigned off = (i_live && (whole || k_cb + (int)b_kc(j) * 8 < cseg)) ? b_off[j] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(Bb + b_piece(j) * 1024), 16, off, sob, 0, 0);
            }
            if (!i_live) return;
            k_cb += BKE;
            if (k_cb >= cseg) {
                if (k_src == 0 && p.c1 > 0) { k_src = 1; k_cb = 0; cseg = p.c1; }
                else i_live = false;
            }
            return;
        } else {
            const __amdgpu_buffer_rsrc_t rsa = rsrc_of(k_src ? i_a1b : i_a0b, a_records);
            const __amdgpu_buffer_rsrc_t rw = rsrc_of(i_w);
            const unsigned colb = (unsigned)k_cb * 2u;
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const bool ok = i_live & (pixn[i] != ~0u) & (k_cb + (int)a_kc(i) * 8 < cseg);
                const unsigned off = ok ? __umul24(pixn[i], (unsigned)ldb) + colb + a_kc(i) * 16u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)(Ab + (wave * APW + i) * 1024), 16, off, 0, 0, 0);
            }
            const int cbase = (k_src ? p.c0 : 0) + k_cb;                               // channel of this k-step in the concat
            const int koffb = (p.taps == 1 ? cbase : (cbase / BKE * 9 + k_tap) * BKE) * 2;   // wave-uniform: rides in soffset
#pragma unroll
            for (int j = 0; j < BPWM; ++j) {
                const bool ok = i_live & (k_cb + (int)b_kc(j) * 8 < cseg);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(Bb + b_piece(j) * 1024), 16, ok ? b_off[j] : OOB, koffb, 0, 0);
            }
            if (!i_live) return;
            // advance (tap fastest, then chunk, then source)
            const int t2 = k_tap + 1;
            const bool wrap_t = t2 == p.taps;
            k_tap = wrap_t ? 0 : t2;
            const int cb2 = wrap_t ? k_cb + BKE : k_cb;
            const bool wrap = cb2 >= cseg;
            k_cb = wrap ? 0 : cb2;
            const bool more = k_src == 0 && p.c1 > 0;
            if (wrap && !more) i_live = false;
            k_src = (wrap && more) ? 1 : k_src;
            cseg = k_src ? p.c1 : p.c0;
            ldb = (k_src ? p.lda1 : p.lda0) * 2;
            if (i_live) {
#pragma unroll
                for (int i = 0; i < APW; ++i) pixn[i] = i_tab[k_tap * BM + a_row(i)];
            }
        }
    };

    // ---- compute side ---------------------------------------------------------------------------------------------------------
    const int fr = lane & 31, fh = lane >> 5;
    const int fs = (fr >> 1) & 7;
    unsigned foff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) foff[g] = (unsigned)(((2 * g + fh) ^ fs) * 16);
    const char* const Afr = smem + (wm * WM + fr) * ROWB;

    int loc = slot;
    BgTile cur = bgemm_decode(p, x, loc);
    if (!cur.valid) return;
    build_table(cur, 0);
    aim(cur, 0);
    if constexpr (!LIN) __syncthreads();
    first_pixels();
    int bo = 0;                                             // byte offset of the buffer holding the stage being multiplied
    issue(0);
    int tabsel = 0;
    __syncthreads();                                        // stage 0 of the first tile has landed

    for (;;) {
        loc += nslots;
        const BgTile nxt = bgemm_decode(p, x, loc);
        if constexpr (!LIN) {
            // the next tile's table, now: the k-loop's barriers (or the one here, for a single-stage tile) put it in front of
            // its first use at this tile's last stage
            if (nxt.valid) build_table(nxt, tabsel ^ 1);
            if (nk < 2) __syncthreads();
        }
        // bias (+ the time-embedding row of the two samples the tile's rows can belong to) for the tile's columns: loaded now,
        // parked in LDS at the tile's last stage (by then the k-loop's barriers have waited for it anyway)
        const int c_z = p.batch > 1 ? cur.rbg / p.nbm_per : 0;
        const int c_bm = cur.rbg - c_z * p.nbm_per;
        const int s_lo = p.rowbias ? (c_bm * BM) / p.rows_per_sample : 0;
        // (unconditional buffer loads, a null pointer as a zero-record descriptor: a load under a branch would leave the compiler's
        // vmcnt bookkeeping with a "maybe pending" register at the loop head and a queue drain there)
        float brv, brr;
        {
            const int n = cur.n0 + (tid & 127), sr = s_lo + ((tid >> 7) & 1);
            const bool rok = n < p.N && (long)sr * p.rows_per_sample < (long)p.M;
            brv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_of(p.bias, p.bias ? 0x7FFFFFF0 : 0),
                                                                                  n < p.N ? (unsigned)n * 4u : OOB, 0, 0));
            brr = LIN ? 0.f                                 // (time-embedding rows belong to the resnets' 3x3 convs)
                      : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_of(p.rowbias, p.rowbias ? 0x7FFFFFF0 : 0),
                                                                                        rok ? (unsigned)(sr * p.rb_ld + n) * 4u : OOB, 0, 0));
        }
        auto run = [&](auto TNc) {
            constexpr int TN = decltype(TNc)::value;
            constexpr int WN = 32 * TN;
            const char* const Bfr = smem + A_BYTES + (wn * WN + fr) * ROWB;
            f32x16 acc[TM][TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            hx8<H> af[2][TM], bfr[2][TN];
            auto read_frags = [&](const int set, const int off, const int g) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) af[set][mi] = *reinterpret_cast<const hx8<H>*>(Afr + off + mi * 32 * ROWB + foff[g]);
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) bfr[set][ni] = *reinterpret_cast<const hx8<H>*>(Bfr + off + ni * 32 * ROWB + foff[g]);
            };
            auto mma = [&](const int set) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = mfma_32x32x16(bfr[set][ni], af[set][mi], acc[mi][ni]);
            };
            const int z = c_z, bm = c_bm;
            read_frags(0, bo, 0);
            for (int ks = 0; ks + 1 < nk; ++ks) {
                const int bn = STAGE - bo;                  // the other buffer
                issue(bn);                                  // stage ks + 1 of this tile
                __builtin_amdgcn_sched_barrier(0);
                read_frags(1, bo, 1);
                mma(0);
                read_frags(0, bo, 2);
                mma(1);
                read_frags(1, bo, 3);
                mma(0);
                __syncthreads();                            // stage ks+1 has landed; every wave is done reading stage ks
                read_frags(0, bn, 0);                       // first fragments of stage ks+1 ...
                __builtin_amdgcn_sched_barrier(0);
                mma(1);                                     // ... land while the last k-group of stage ks is multiplied
                bo = bn;
            }
            // the tile's last stage
            const int bn = STAGE - bo;
            brow[tid & 127] = brv;                          // read behind the barrier below
            brow[128 + (tid & 255)] = brr;
            read_frags(1, bo, 1);
            mma(0);
            read_frags(0, bo, 2);
            mma(1);
            read_frags(1, bo, 3);
            mma(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // every wave is done reading the stage: it becomes the staging area
            __builtin_amdgcn_sched_barrier(0);
            mma(1);
            BgEpilogue<H, BM, TM, TN, WM, WN, !LIN> epi;
            epi.prefetch(p, bm, cur.n0, wm, wn, lane);      // bias / residual loads: older than the DMA below in the vmcnt queue
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PREFETCH) {
                if (nxt.valid) {                            // the ring runs on into the next tile while this one is written out
                    aim(nxt, tabsel ^ 1);
                    first_pixels();
                    tabsel ^= 1;
                }
                issue(bn);
                __builtin_amdgcn_sched_barrier(0);
                H* __restrict__ out = reinterpret_cast<H*>(p.out) + (size_t)z * p.sout;
                epi.finish(p, acc, out, bm, cur.n0, wm, wn, lane, reinterpret_cast<float*>(smem + bo) + wave * 32 * WN, brow, s_lo);
                // finish() ends on a counted vmcnt: this wave's part of the next tile's stage 0 has landed, its stores have not
                // been waited for
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();               // staging reads done; stage 0 of the next tile is in LDS
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // 256-row tiles: eight waves stage 64 KB, more than one 48 KB stage buffer -- the staging area is both buffers and
                // the next tile's first stage is fetched behind the epilogue (one exposed round trip per ~100 us tile); the
                // workgroup still keeps its place, its next table is built and its stores are not waited for
                H* __restrict__ out = reinterpret_cast<H*>(p.out) + (size_t)z * p.sout;
                epi.finish(p, acc, out, bm, cur.n0, wm, wn, lane, reinterpret_cast<float*>(smem) + wave * 32 * WN, brow, s_lo);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();               // staging reads done
                __builtin_amdgcn_sched_barrier(0);
                if (nxt.valid) {
                    aim(nxt, tabsel ^ 1);
                    first_pixels();
                    tabsel ^= 1;
                    issue(bn);
                    __syncthreads();
                }
            }
            bo = bn;
        };
        if (cur.wide) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 1>{});
        if (!nxt.valid) break;
        cur = nxt;
    }
}

// (Two weight-stationary designs for the K <= 640 projections were built, validated and measured against the kernels above on a
// B = 32 pass, and removed: a weight PANEL resident in LDS -- 80 KB, 256-row blocks of A streamed past it through a two-stage ring,
// one 512-thread workgroup per CU, half the L2 -> LDS traffic -- ran 8-76 % SLOWER (one stage in flight per CU: 1.4 us per stage);
// weights in REGISTERS -- 160 VGPRs per wave for K = 320, every wave streaming its own 32 rows through a private four-slot ring
// with counted vmcnt and no barrier at all, 96 KB in flight per CU -- ran -7 % (GEGLU) .. +5 %: with the latency gone the same
// 8-10 TB/s of L2 -> LDS traffic remain, because a 64-column wave panel re-reads A as often as 128 x 128 tiles re-read A and B.
// DESIGN.md 3.5.)

template <typename H, bool LIN>
__global__ __launch_bounds__(256, 2) void bgemm_pers_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_bp[];
    bgemm_pers_body<H, 128, 2, LIN>(p, smem_bp, blockIdx.x & 7, blockIdx.x >> 3, gridDim.x >> 3);
}

// (bgemm_pers_body<H, 256, 4, LIN> -- persistent 256-row tiles, 512 threads -- was built and measured bit-identical and +-0 % against
// bgemm256_kernel on every 3x3 conv of a B = 32 pass: a 256-row conv tile lives ~100 us, its launch and drain are noise.  Not
// instantiated.)

template <typename H, bool LIN>
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
        if (j < p.w1) bgemm_tile<H, 128, 128, 2, 2, 128 * 128 * 2, LIN>(p, rb_lo + r, j * 128, smem_bg);
        else bgemm_tile<H, 128, 64, 2, 2, 128 * 128 * 2, LIN>(p, rb_lo + r, p.w1 * 128 + (j - p.w1) * 64, smem_bg);
    } else {
        const int t = loc - n1;
        if (t >= tail * p.s2) return;
        const int r = t / p.s2;
        bgemm_tile<H, 128, 64, 2, 2, 128 * 128 * 2, LIN>(p, rb_lo + (nrb - tail) + r, (t - r * p.s2) * 64, smem_bg);
    }
}

// ---- split-K: the small-batch dispatch family ------------------------------------------------------------------------------------
// The reference generates clip by clip (inference_eeg2video.py:90-100): one clip is two UNet samples, and the deep levels then hand
// the tile kernels 480 or 1728 rows -- 40 to 140 tiles of 128 x 128 for 256 CUs, each walking K = 11 520 .. 23 040 alone (measured at
// B = 1: 95 TFLOP/s on the 5x8 level, 148 us per conv whose 29 MB of weights an idle chip streams in 10).  Here blockIdx.y cuts K into
// runs of whole 64-channel chunks -- each inside ONE source of a concat, so a run is the same tile kernel on shifted pointers: source
// rows + first channel, weight rows + first k of the run, no epilogue, fp32 partial tile to IgemmArgs::sk_ws[run] -- and
// splitk_reduce_kernel adds the runs in order and applies the epilogue (bias, time-embedding row, residual, ReLU, one rounding).
// Deterministic; equal to the unsplit kernels up to fp32 summation order (the family is held to the oracle bounds, not to bit-identity).
template <typename H, bool LIN, int NST>      // NST = 3: the three-stage ring (one workgroup per CU: launches of at most ~one round)
__global__ __launch_bounds__(256) void bgemm_splitk_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_sk[];
    IgemmArgs q = p;
    const int run = blockIdx.y;
    const bool src1 = run >= p.sk_s0;
    const int per = src1 ? p.sk_q1 : p.sk_q0;
    const int qlo = (src1 ? run - p.sk_s0 : run) * per;                  // first 64-channel chunk of the run, within its source
    const int cseg = src1 ? p.c1 : p.c0;
    const int clo = qlo * 64;
    const int chi = min(cseg, clo + per * 64);
    q.a0 = reinterpret_cast<const float*>(reinterpret_cast<const H*>(src1 ? p.a1 : p.a0) + clo);
    q.lda0 = src1 ? p.lda1 : p.lda0;
    q.c0 = chi - clo; q.c1 = 0; q.a1 = nullptr;
    // first k of the run in a weight row: taps = 1: channel (c0 +) clo; 3x3: chunk-major [chunk][tap][64], chunks of source 1 behind source 0's
    const long kfirst = p.taps == 1 ? (long)(src1 ? p.c0 : 0) + clo : (long)((src1 ? (p.c0 + 63) / 64 : 0) + qlo) * p.taps * 64;
    q.w16 = reinterpret_cast<const char*>(p.w16) + kfirst * 2;
    q.out = p.sk_ws + (size_t)run * p.M * p.N; q.ldc = p.N; q.out_f32 = 1;
    q.bias = nullptr; q.rowbias = nullptr; q.resid = nullptr; q.alpha = 1.f; q.relu = 0;
    const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int nrb = rb_hi - rb_lo;
    const int per_rb = p.w1 + p.s1;
    if (loc >= nrb * per_rb) return;
    const int r = loc / per_rb, j = loc - r * per_rb;
    if (j < p.w1) bgemm_tile<H, 128, 128, 2, 2, 128 * 128 * 2, LIN, NST, 1>(q, rb_lo + r, j * 128, smem_sk);
    else bgemm_tile<H, 128, 64, 2, 2, 128 * 128 * 2, LIN, NST, 1>(q, rb_lo + r, p.w1 * 128 + (j - p.w1) * 64, smem_sk);
}

// out[m][n .. n+3] = epilogue(sum over runs, run 0 first); one thread = four consecutive columns of a row
template <typename H>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IgemmArgs p, const int runs) {
    const int nq = p.N >> 2;
    const long total = (long)p.M * nq;
    const size_t plane = (size_t)p.M * p.N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / nq;
        const int n = (int)(i - m * nq) * 4;
        const float* src = p.sk_ws + (size_t)m * p.N + n;
        f32x4 y = *reinterpret_cast<const f32x4*>(src);
        for (int r = 1; r < runs; ++r) y += *reinterpret_cast<const f32x4*>(src + r * plane);
        if (p.alpha != 1.0f) y *= p.alpha;
        if (p.bias) y += *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.rowbias) y += *reinterpret_cast<const f32x4*>(p.rowbias + (size_t)(m / p.rows_per_sample) * p.rb_ld + n);
        if (p.resid) {
            if (p.resid_bf16) {
                const hx4<H> rr = *reinterpret_cast<const hx4<H>*>(reinterpret_cast<const H*>(p.resid) + (size_t)m * p.ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] += (float)rr[e];
            } else {
                y += *reinterpret_cast<const f32x4*>(p.resid + (size_t)m * p.ldr + n);
            }
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
        }
        if (p.out_f32) {
            *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.ldc + n) = y;
        } else {
            hx4<H> o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (H)y[e];
            *reinterpret_cast<hx4<H>*>(reinterpret_cast<H*>(p.out) + (size_t)m * p.ldc + n) = o;
        }
    }
}

// 128-row tiles on the THREE-stage ring: launches of at most one round of workgroups (the small-batch operating point: two UNet
// samples give a level-2 linear 140 tiles).  With a CU to itself a two-stage tile is bound by one L2 / HBM round trip per 64-deep stage
// (measured at B = 1: 25 us for the 20 stages of a 1728 x 1280 x 1280 linear, 225 TFLOP/s); three 32 KB stages keep two in flight
// behind a counted vmcnt.  Same k order, same epilogue: bit-identical to the other tile kernels (the choice may follow the launch size).
template <typename H, bool LIN>
__global__ __launch_bounds__(256) void bgemm_s3_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_bgs3[];
    const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int nrb = rb_hi - rb_lo;
    const int tail = min(nrb, p.tail_rb);
    const int per1 = p.w1 + p.s1;
    const int n1 = (nrb - tail) * per1;
    if (loc < n1) {
        const int r = loc / per1, j = loc - r * per1;
        if (j < p.w1) bgemm_tile<H, 128, 128, 2, 2, 128 * 128 * 2, LIN, 3, 2>(p, rb_lo + r, j * 128, smem_bgs3);
        else bgemm_tile<H, 128, 64, 2, 2, 128 * 128 * 2, LIN, 3, 2>(p, rb_lo + r, p.w1 * 128 + (j - p.w1) * 64, smem_bgs3);
    } else {
        const int t = loc - n1;
        if (t >= tail * p.s2) return;
        const int r = t / p.s2;
        bgemm_tile<H, 128, 64, 2, 2, 128 * 128 * 2, LIN, 3, 2>(p, rb_lo + (nrb - tail) + r, (t - r * p.s2) * 64, smem_bgs3);
    }
}

// 256-row tiles: 8 waves (4 x 2, the same 64 x 64 wave tile), 256 x 128 and 256 x 64, 48 KB stages, one workgroup per CU.
#ifdef E2V_AB          // the two-stage form of the 256-row tile (E2V_BGEMM_S3 = 0): the other arm of the A/B that adopted the three-stage ring
template <typename H, bool LIN>
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
        if (j < p.w1) bgemm_tile<H, 256, 128, 4, 2, ST, LIN>(p, rb_lo + r, j * 128, smem_bg256);
        else bgemm_tile<H, 256, 64, 4, 2, ST, LIN>(p, rb_lo + r, p.w1 * 128 + (j - p.w1) * 64, smem_bg256);
    } else {
        const int t = loc - n1;
        if (t >= tail * p.s2) return;
        const int r = t / p.s2;
        bgemm_tile<H, 256, 64, 4, 2, ST, LIN>(p, rb_lo + (nrb - tail) + r, (t - r * p.s2) * 64, smem_bg256);
    }
}
#endif

// (A variant with one A stage per KERNEL ROW -- the input pixels under output pixels m0-1 .. m0+256 fetched once per ky and read at
// LDS rows r + kx by the three taps of that row, image-row ends zeroed in registers: activations fetched 3x instead of 9x, L2
// requests -42 % -- was built, came out bit-identical, and ran at the SAME speed (1.576 ms against 1.577 ms on 640 -> 640 @ 18x32,
// B = 32): with two stages in flight these kernels are not bound by what they fetch.  Removed)

// the same with a three-stage ring (144 KB of stages + the gather table: all of a CU's LDS)
template <typename H, bool LIN>
__global__ __launch_bounds__(512) void bgemm256s3_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_bg256s3[];
    constexpr int ST = (256 + 128) * 128;
    const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int nrb = rb_hi - rb_lo;
    const int tail = min(nrb, p.tail_rb);
    const int per1 = p.w1 + p.s1;
    const int n1 = (nrb - tail) * per1;
    if (loc < n1) {
        const int r = loc / per1, j = loc - r * per1;
        if (j < p.w1) bgemm_tile<H, 256, 128, 4, 2, ST, LIN, 3>(p, rb_lo + r, j * 128, smem_bg256s3);
        else bgemm_tile<H, 256, 64, 4, 2, ST, LIN, 3>(p, rb_lo + r, p.w1 * 128 + (j - p.w1) * 64, smem_bg256s3);
    } else {
        const int t = loc - n1;
        if (t >= tail * p.s2) return;
        const int r = t / p.s2;
        bgemm_tile<H, 256, 64, 4, 2, ST, LIN, 3>(p, rb_lo + (nrb - tail) + r, (t - r * p.s2) * 64, smem_bg256s3);
    }
}

bool bgemm_use_256(const IgemmArgs& a) {
    static const int* const modep = knob("E2V_BGEMM_256", 1);      // 0: never, 2: always
    const int mode = *modep;
    if (mode == 0) return false;
    if (mode == 2) return true;
    // 3x3 convs: one 512-thread workgroup per CU on a three-stage ring (two stages in flight) -- with two stages a workgroup that is
    // alone on its CU pays an L2 round trip per 64-deep stage (matrix pipe 0.45 busy); 1150-1260 TFLOP/s on the big convs.
    // Linears (same-box A/B over every shape of a B = 32 pass): the residual-free projections at K >= 640 (QKV, GEGLU, concat
    // shortcuts) and K > 2560 gain 4-17 % on 256-row tiles; the ones with a residual to read are better off on the persistent
    // kernel (256-row: +5..+18 %), and K = 320 is five stages deep -- nothing for a ring to do.  E2V_BGEMM_256LIN: that smallest K
    // (0: no linear takes 256-row tiles)
    if (a.taps == 1) {
        static const int* const lin256 = knob("E2V_BGEMM_256LIN", 640);
        const int Kc = a.c0 + a.c1;
        if (*lin256 <= 0 || Kc < *lin256 || (a.resid && Kc <= 2560)) return false;
    }
    // enough rounds of 256 resident tiles, else the finer 128-row grid wastes less on its last round
    const double tiles = (double)((a.M + 255) / 256) * a.batch * ((a.N + 127) / 128);
    // (two rounds: with the three-stage ring the 256-row tile wins from there on -- level-3 convs -11 %; it was four with two stages)
    static const int* const minr = E2V_AB_KNOB("E2V_BGEMM_256_MINROUNDS", 2);
    return tiles >= (double)*minr * 256;
}

// Launches whose tiles are all 128 x 64 (N <= 64, grids below one round, and -- E2V_BGEMM_N64_MAXK -- short-K layers): 24 KB
// stages, 53 KB of LDS per workgroup, so THREE workgroups share a CU.  A short-K tile (K = 320: five stages) spends most of its
// life waiting for its first HBM bytes; a third resident workgroup is one more tile's worth of loads in flight per CU.
template <typename H, bool LIN>
__global__ __launch_bounds__(256) void bgemm_n64_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_bg64[];
    const int x = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int nrb = rb_hi - rb_lo;
    if (loc >= nrb * p.s2) return;
    const int r = loc / p.s2;
    bgemm_tile<H, 128, 64, 2, 2, (128 + 64) * 128, LIN>(p, rb_lo + r, (loc - r * p.s2) * 64, smem_bg64);
}

bool bgemm_all_n64(const IgemmArgs& a) {
    static const int maxk = [] { const char* e = std::getenv("E2V_BGEMM_N64_MAXK"); return e ? std::atoi(e) : 0; }();
    static const int conv = [] { const char* e = std::getenv("E2V_BGEMM_N64_CONV"); return e ? std::atoi(e) : 0; }();
    if (a.taps != 1) return conv != 0;
    return (a.c0 + a.c1) <= maxk && !a.geglu;
}

// launch KERNEL<H, LIN> for the launch's 16-bit type (IgemmArgs::a_bf16: bf16 / fp16) and gather path, opted into `bytes` of LDS
#define E2V_BG_LAUNCH(KERNEL, grid, block, bytes)                                                                              \
    h16_dispatch(a.a_bf16, [&](auto h16_tag) {                                                                                 \
        using H = decltype(h16_tag);                                                                                           \
        if (lin) { E2V_KATTR((&KERNEL<H, true>), bytes); E2V_KLAUNCH((KERNEL<H, true>), grid, block, bytes, s, a); }           \
        else { E2V_KATTR((&KERNEL<H, false>), bytes); E2V_KLAUNCH((KERNEL<H, false>), grid, block, bytes, s, a); }             \
    })

static int splitk_layout(const IgemmArgs& a, int want, int& s0, int& q0, int& q1) {
    if (want < 2 || !a.a_bf16 || a.batch != 1 || a.geglu || a.rbsum || a.osy || (a.taps != 1 && a.taps != 9)) return 0;
    if (((a.N | a.ldc) & 3) || (a.resid && (a.ldr & 3)) || (a.rowbias && (a.rb_ld & 3)) || a.M <= 0) return 0;
    if (a.c1 > 0 && a.c0 % 64) return 0;                          // (a run starts on a whole chunk of the weight row)
    const int Q0 = (a.c0 + 63) / 64, Q1 = (a.c1 + 63) / 64, Q = Q0 + Q1;
    if (want > Q) want = Q;
    if (want < 2) return 0;
    int s1 = 0;
    if (Q1 > 0) {
        s0 = (int)((double)want * Q0 / Q + 0.5);
        s0 = s0 < 1 ? 1 : (s0 > want - 1 ? want - 1 : s0);
        s1 = want - s0;
    } else {
        s0 = want;
    }
    q0 = (Q0 + s0 - 1) / s0; s0 = (Q0 + q0 - 1) / q0;
    q1 = 0;
    if (Q1 > 0) { q1 = (Q1 + s1 - 1) / s1; s1 = (Q1 + q1 - 1) / q1; }
    return s0 + s1 >= 2 ? s0 + s1 : 0;
}
int splitk_plan(const IgemmArgs& a, int want) {
    int s0 = 0, q0 = 0, q1 = 0;
    return splitk_layout(a, want, s0, q0, q1);
}

// igemm() has filled the 128-row tile schedule (full-size tiles: rb1 = nbm)
void bgemm_splitk_launch(const IgemmArgs& a_in, hipStream_t s) {
    IgemmArgs a = a_in;
    const int runs = splitk_layout(a, a.sk, a.sk_s0, a.sk_q0, a.sk_q1);
    if (runs < 2 || !a.sk_ws) throw Error(E2V_EINVAL, "split-K launch without a plan (splitk_plan) or workspace");
    constexpr size_t smem = (size_t)2 * 128 * 128 * 2 + 9 * 128 * sizeof(unsigned);
    const bool lin = a.taps == 1;
    const double K = (double)a.taps * (a.c0 + a.c1);
    const double rows_in = a.taps == 1 ? (double)a.M : (double)a.M * a.Hs * a.Ws / ((double)a.Ho * a.Wo);
    std::string pname = a.a_bf16 == H16_FP16 ? "igemm_fp16" : "igemm_bf16";
    if (prof_detail())
        pname += " M" + std::to_string(a.M) + " N" + std::to_string(a.N) + " K" + std::to_string((long)K) + " t" + std::to_string(a.taps) +
                 (a.stride > 1 ? " s2" : "") + (a.upsample ? " up" : "") + (a.c1 ? " cat" : "");
    ProfScope ps(pname.c_str(), 2.0 * a.M * a.N * K, 2.0 * rows_in * (a.c0 + a.c1) + 2.0 * a.N * K + (a.out_f32 ? 4.0 : 2.0) * a.M * a.N, s);
    int per_xcd = 0;
    for (int x = 0; x < 8; ++x) {
        const int nrb = (int)(((long)(x + 1) * a.nbm) >> 3) - (int)(((long)x * a.nbm) >> 3);
        per_xcd = nrb > per_xcd ? nrb : per_xcd;
    }
    const dim3 grid(8 * per_xcd * (a.w1 + a.s1), runs, 1);
    // at most ~one round of workgroups: the three-stage ring (one workgroup per CU, two stages in flight); more: two co-resident
    // two-stage workgroups per CU cover each other's waits
    const bool s3 = (long)grid.x * runs <= 320 && (a.taps == 9 || lin);
    constexpr size_t smem3 = (size_t)3 * 128 * 128 * 2 + 9 * 128 * sizeof(unsigned);
    dry_tag(" -> bgemm_splitk_kernel 128x128 x" + std::to_string(runs) + (s3 ? " s3" : "") + " + splitk_reduce_kernel");
    h16_dispatch(a.a_bf16, [&](auto h16_tag) {
        using H = decltype(h16_tag);
        auto go = [&](auto kern, const size_t bytes) { E2V_KATTR(kern, bytes); E2V_KLAUNCH(kern, grid, dim3(256), bytes, s, a); };
        if (s3) { if (lin) go(bgemm_splitk_kernel<H, true, 3>, smem3); else go(bgemm_splitk_kernel<H, false, 3>, smem3); }
        else    { if (lin) go(bgemm_splitk_kernel<H, true, 2>, smem); else go(bgemm_splitk_kernel<H, false, 2>, smem); }
    });
    const long quads = (long)a.M * (a.N / 4);
    const int blocks = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
    h16_dispatch(a.a_bf16, [&](auto h16_tag) {
        using H = decltype(h16_tag);
        E2V_KLAUNCH(splitk_reduce_kernel<H>, dim3(blocks), dim3(256), 0, s, a, runs);
    });
}

void bgemm_launch(const IgemmArgs& a_in, int ntiles, hipStream_t s) {
    IgemmArgs a = a_in;
    constexpr size_t smem = (size_t)2 * 128 * 128 * 2 + 9 * 128 * sizeof(unsigned);     // two stages of (128 + 128) rows + gather table
    static const int* const lean = E2V_AB_KNOB("E2V_BGEMM_LIN", 1);        // 0: linears through the gather path
    const bool lin = a.taps == 1 && *lean;
    const double K = (double)a.taps * (a.c0 + a.c1);
    const double rows_in = a.taps == 1 ? (double)a.M : (double)a.M * a.Hs * a.Ws / ((double)a.Ho * a.Wo);
    std::string pname = a.a_bf16 == H16_FP16 ? "igemm_fp16" : "igemm_bf16";
    if (prof_detail())
        pname += " M" + std::to_string(a.M) + " N" + std::to_string(a.N) + " K" + std::to_string((long)K) + " t" + std::to_string(a.taps) +
                 (a.stride > 1 ? " s2" : "") + (a.upsample ? " up" : "") + (a.c1 ? " cat" : "") + (a.geglu ? " geglu" : "") +
                 (a.batch > 1 ? " b" + std::to_string(a.batch) : "");
#ifdef E2V_ABLATE
    // timing experiments (make EXTRA=-DE2V_ABLATE): drop the output stores / zero the A loads -- the results are WRONG, so the
    // switch does not exist in the shipped build
    static const int* const ablate = knob("E2V_BGEMM_ABLATE", 0);
    a.ablate = *ablate;
    if (a.ablate) { static bool said = false; if (!said) { said = true; fprintf(stderr, "libeeg2video_hip: E2V_BGEMM_ABLATE=%d -- GEMM RESULTS ARE WRONG (timing build)\n", a.ablate); } }
#else
    a.ablate = 0;
#endif
    const double out_b = a.out_f32 ? 4.0 : 2.0;
    ProfScope ps(pname.c_str(), 2.0 * a.M * a.N * K * a.batch,
                 a.batch * (2.0 * rows_in * (a.c0 + a.c1) + 2.0 * a.N * K + out_b * a.M * (a.geglu ? a.N / 2 : a.N)), s);
    // E2V_BGEMM_PERS: 0 never, 1 where it measured faster (same-box A/B over every GEMM shape of a B = 32 pass, tools/shape_profile.py:
    // linears with a residual to read, K <= 2560: -3..-20 %; K = 320 and the K = 640 GEGLU without one: -4..-6 %; the wide
    // residual-free projections at K >= 640 and the 3x3 convs that are too small for 256-row tiles: +2..+6 %), 2 wherever it applies
    // launches of at most one round of 128-row tiles: the three-stage ring (bgemm_s3_kernel), one workgroup per CU
    static const int* const s3on = knob("E2V_BGEMM_S3_SMALL", 1);
    if (*s3on && !a.bm256 && (a.taps == 9 || lin)) {
        long real = 0;                                                                // tiles the grid really holds
        for (int x = 0; x < 8; ++x) {
            const int nrb = (int)(((long)(x + 1) * a.nbm) >> 3) - (int)(((long)x * a.nbm) >> 3);
            const int tail = nrb < a.tail_rb ? nrb : a.tail_rb;
            real += (long)(nrb - tail) * (a.w1 + a.s1) + (long)tail * a.s2;
        }
        const int nk = a.taps * ((a.c0 + 63) / 64 + (a.c1 + 63) / 64);
        // (same-process A/B at B = 1, profiles/r05_shape_ab_b1_s3_small.log: 40 tiles -- the 5x8 level -- 24 -> 18 us; 140 tiles +-0 .. +7 %:
        // from there on two co-resident two-stage workgroups per CU cover each other as well)
        if (real <= 96 && nk >= 4) {
            constexpr size_t smem3 = (size_t)3 * 128 * 128 * 2 + 9 * 128 * sizeof(unsigned);
            dry_tag(" -> bgemm_s3_kernel 128x128 s3");
            E2V_BG_LAUNCH(bgemm_s3_kernel, dim3(ntiles, 1, 1), dim3(256), smem3);
            return;
        }
    }
    static const int* const persp = knob("E2V_BGEMM_PERS", 1);
    const int Kc = a.c0 + a.c1;
    const int pers = *persp == 2 ? 1 : *persp == 0 ? 0 : (a.taps == 1 && (a.resid ? Kc <= 2560 : (Kc <= 320 || (a.geglu && Kc <= 640))));
    // (an fp32 residual, or time-embedding rows of more than two samples under one tile -- toy sizes -- take the kernels below)
    if (pers && !a.out_f32 && !a.bm256 && !(a.resid && !a.resid_bf16) && !(a.rowbias && (lin || a.rows_per_sample < 128))) {
        constexpr size_t smem_p = (size_t)2 * 128 * 128 * 2 + 2 * 9 * 128 * sizeof(unsigned) + 3 * 128 * sizeof(float);   // stages, tables, bias rows
        const int grid = ntiles < 512 ? ntiles : 512;                                             // two workgroups per CU
        dry_tag(" -> bgemm_pers_kernel 128x128");
        E2V_BG_LAUNCH(bgemm_pers_kernel, dim3(grid, 1, 1), dim3(256), smem_p);
        return;
    }
    static const int* const s3p = E2V_AB_KNOB("E2V_BGEMM_S3", 1);          // 256-row tiles on a three-stage ring
    if (a.bm256 && *s3p && (a.taps == 9 || lin)) {
        constexpr size_t smem256s3 = (size_t)3 * (256 + 128) * 128 + 9 * 256 * sizeof(unsigned);
        dry_tag(" -> bgemm256s3_kernel 256x128");
        E2V_BG_LAUNCH(bgemm256s3_kernel, dim3(ntiles, 1, 1), dim3(512), smem256s3);
        return;
    }
#ifdef E2V_AB
    if (a.bm256) {
        constexpr size_t smem256 = (size_t)2 * (256 + 128) * 128 + 9 * 256 * sizeof(unsigned);
        dry_tag(" -> bgemm256_kernel 256x128");
        E2V_BG_LAUNCH(bgemm256_kernel, dim3(ntiles, 1, 1), dim3(512), smem256);
        return;
    }
#else
    if (a.bm256) throw Error(E2V_EINVAL, "bf16 GEMM: 256-row tiles serve taps = 1 and taps = 9 only");
#endif
    if (a.rb1 == 0 && !a.geglu) {          // every row block is cut into 128 x 64 tiles only
        constexpr size_t smem64 = (size_t)2 * (128 + 64) * 128 + 9 * 128 * sizeof(unsigned);
        int nt = 0;
        for (int x = 0; x < 8; ++x) {
            const int nrb = (int)(((long)(x + 1) * a.nbm) >> 3) - (int)(((long)x * a.nbm) >> 3);
            nt = nrb * a.s2 > nt ? nrb * a.s2 : nt;
        }
        dry_tag(" -> bgemm_n64_kernel 128x64");
        E2V_BG_LAUNCH(bgemm_n64_kernel, dim3(nt * 8, 1, 1), dim3(256), smem64);
        return;
    }
    dry_tag(" -> bgemm_kernel 128x128" + std::string(a.rb1 < a.nbm ? "+128x64" : ""));
    E2V_BG_LAUNCH(bgemm_kernel, dim3(ntiles, 1, 1), dim3(256), smem);
}

}  // namespace e2v
