// Sparse-causal self-attention on bf16 rows, 64 queries per wave (round 4).
//
// Serves SparseCausalAttention (attention.py:272-328) in bf16 mode at the head sizes whose softmax denominator rides in O^T
// (D % 32 != 0: 40 and 80, the UNet's levels 0 and 1).  Same arithmetic as flash_attn_b16io_kernel<D, true, 64> (attn.hip): S^T = K Q^T
// with the query on the lane, Q pre-multiplied by scale * log2(e), the running maximum subtracted by the matrix pipe (the score chain
// starts from an accumulator block that holds -m), a row's maximum moved only when a score tops it by more than 2^8, P^T registers as
// the B operand of O^T = V^T P^T, the V^T fragment through the transposing LDS read, the denominator as row D of O^T.
//
// What is different is who shares what.  flash_attn_b16io_kernel gives a wave 32 queries: every K fragment (ds_read_b128) and every V^T
// fragment (2 x ds_read_b64_tr_b16) feeds ONE MFMA, and the loop's fixed costs -- staging a 64-key stage through registers, the
// stage's barrier, loop and address arithmetic -- are paid per 32 queries.  That kernel is bound by what its waves ISSUE (vector ALU
// busy 0.69, matrix pipe 0.48, 9.1 vector instructions per MFMA at d = 40), so here a wave owns TWO 32-query blocks: a fragment read
// feeds two MFMAs, staging / barrier / loop instructions per query halve, and the two blocks' chains (scores -> softmax -> PV) are
// independent, so one block's MFMAs run under the other's softmax inside ONE wave.  Per 32 keys and 32 queries the vector work is
// now 8 integer max3 (the deferred-maximum test needs no float maximum: a score above the 2^8 threshold is positive, and positive
// floats order like their bit patterns, so v_max3_i32 on the raw MFMA output finds it without the canonicalising v_max the
// compiler puts in front of every fmaxf of an MFMA result) + 1 compare + 16 v_exp_f32 + 8 v_cvt_pk_bf16_f32; the cross-half
// exchange of the maximum happens only inside the (rare) branch that moves a maximum.
#include "h16.h"
#include "kernels.h"
#include "prof.h"
#include "runtime.h"

#include <cstdlib>
#include <string>
#include <type_traits>

namespace e2v {

typedef float qf32x16 __attribute__((ext_vector_type(16)));
typedef float qf32x4 __attribute__((ext_vector_type(4)));

// Workgroup -> (query block, head, sample-frame), XCD-aware exactly as attn_block of attn.hip (XCD b & 7 takes whole samples and walks
// them frame by frame, head by head: the K / V rows of a frame are fetched into ONE L2), for query blocks of QB rows.
struct Q64Block { int qb, head, sf; bool valid; };
__host__ __device__ __forceinline__ int q64_xcd_unit(const int n, const int F) { return n % 8 == 0 ? 0 : (n * F) % 8 == 0 ? 1 : 2; }      // as attn_xcd_unit (attn.hip)
__device__ __forceinline__ Q64Block q64_block(const AttnArgs& p, const int QB) {
    const int nqb = (p.Nq + QB - 1) / QB;
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int S = p.n * p.F;
    const int unit = q64_xcd_unit(p.n, p.F);
    Q64Block r;
    if (unit == 0) {
        const int per = p.F * p.heads * nqb;
        const int ul = idx / per, w = idx - ul * per;
        const int smp = ul * 8 + xcd;
        const int f = w / (p.heads * nqb), w2 = w - f * (p.heads * nqb);
        r.sf = smp * p.F + f; r.head = w2 / nqb; r.qb = w2 - r.head * nqb; r.valid = smp < p.n;
    } else if (unit == 1) {
        const int per = p.heads * nqb;
        const int ul = idx / per, w = idx - ul * per;
        r.sf = ul * 8 + xcd; r.head = w / nqb; r.qb = w - r.head * nqb; r.valid = r.sf < S;
    } else {                                         // (sample-frame, head) pairs: the units always divide over the XCDs
        const int ul = idx / nqb;
        const int pair = ul * 8 + xcd;
        r.qb = idx - ul * nqb; r.sf = pair / p.heads; r.head = pair - r.sf * p.heads; r.valid = pair < S * p.heads;
    }
    return r;
}
static inline unsigned q64_grid(const AttnArgs& a, const int QB) {
    const unsigned nqb = (a.Nq + QB - 1) / QB;
    const int unit = q64_xcd_unit(a.n, a.F);
    if (unit == 0) return 8u * (a.n / 8) * (unsigned)(a.F * a.heads) * nqb;
    if (unit == 1) return 8u * ((a.n * a.F) / 8) * (unsigned)a.heads * nqb;
    return 8u * ((a.n * a.F * a.heads + 7) / 8) * nqb;
}

__device__ __forceinline__ int imax3(const int a, const int b, const int c) { return max(max(a, b), c); }

template <int D>
struct Q64Layout {
    static constexpr int DP = (D + 15) / 16 * 16;     // head dim padded to the 16-deep MFMA step
    static constexpr int KS = DP / 16;
    static constexpr int T = (D + 31) / 32;           // 32-row tiles of O^T
    static constexpr int KT = 64;                     // keys per LDS stage
    static constexpr int KROW = DP * 2 + 16;          // bytes per K row: conflict-free ds_read_b128 of 16 rows
    static constexpr int VROW = ((T * 16) % 32 == 16) ? T * 64 : T * 64 + 64;       // as in attn.hip: the four rows of a tr read tile the banks
    static constexpr int KBYTES = KT * KROW, VBYTES = KT * VROW;
    static constexpr int STAGE = (KBYTES + VBYTES + 15) / 16 * 16;
};

#ifdef E2V_AB          // the phase-by-phase form (E2V_ATTN_Q64P = 0; d = 80 with E2V_ATTN_Q64 = 2): the other arm of the A/B
template <typename H, int D, int NW>
__global__ __launch_bounds__(64 * NW) void flash_attn_b16q64_kernel(const AttnArgs p) {
    static_assert(D % 32 != 0 && D % 8 == 0, "the denominator rides in a spare row of the last O^T tile");
    static_assert(NW >= 2 && NW <= 4, "");
    typedef Q64Layout<D> L;
    constexpr int NT = 64 * NW, QB = 64 * NW;
    constexpr int KS = L::KS, T = L::T, KT = L::KT, KROW = L::KROW, VROW = L::VROW, KBYTES = L::KBYTES, STAGE = L::STAGE;
    constexpr int LROW = D % 32, LREG = 4 * (LROW / 8) + (LROW & 3), LHALF = (LROW >> 2) & 1;      // where row D of O^T lives
    constexpr int C8 = D / 8;                   // 16-byte pieces per row
    constexpr int NP = KT * C8;
    constexpr int LPT = (NP + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem_q[];      // [2][K rows | V rows]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const Q64Block blk = q64_block(p, QB);
    if (!blk.valid) return;
    const int sf = blk.sf;
    const int smp = sf / p.F, f = sf - smp * p.F;
    const int head = blk.head;
    const int q0 = blk.qb * QB + wave * 64;
    const bool active = q0 < p.Nq;
    const H* __restrict__ Q = reinterpret_cast<const H*>(p.q);
    const H* __restrict__ K = reinterpret_cast<const H*>(p.k);
    const H* __restrict__ V = reinterpret_cast<const H*>(p.v);

    int nseg = 1;
    size_t kvbase[2];
    kvbase[0] = (size_t)(smp * p.F) * p.Nk;
    kvbase[1] = (size_t)(smp * p.F + (f > 0 ? f - 1 : 0)) * p.Nk;
    nseg = f >= 2 ? 2 : 1;                      // frames 0 and 1 see [K0; K0]: softmax over a duplicated key set = softmax over the set
    const int tps = (p.Nk + KT - 1) / KT;
    const int ntiles = nseg * tps;

    for (int i = tid * 16; i < 2 * STAGE; i += NT * 16)                  // pad columns are never rewritten: keep them finite
        *reinterpret_cast<qf32x4*>(smem_q + i) = qf32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (tid < 2 * KT)                                                     // value column D of every key row, both stages: 1.0
        *reinterpret_cast<H*>(smem_q + (tid / KT) * STAGE + KBYTES + (tid % KT) * VROW + D * 2) = (H)1.0f;

    hx8<H> qf[2][KS];
    {
        const float qs = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int qrow = min(q0 + 32 * b + j, p.Nq - 1);
            const H* qp = Q + ((size_t)sf * p.Nq + qrow) * p.ldq + head * D;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int k0 = 16 * s + 8 * h;
                hx8<H> a;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = (H)0.f;
                if (k0 < D) a = *reinterpret_cast<const hx8<H>*>(qp + k0);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = (H)((float)a[e] * qs);
                qf[b][s] = a;
            }
        }
    }
    __syncthreads();

    constexpr unsigned OOB = 0x80000000u;
    auto rsrc_of = [](const void* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), (short)0, 0x7FFFFFF0,
                                                 0x00020000);
    };
    int ld_row[LPT], ld_c8[LPT];
    unsigned ld_off[LPT];
#pragma unroll
    for (int e = 0; e < LPT; ++e) {
        const int idx = tid + NT * e;
        const int row = idx / C8, c8 = idx - row * C8;
        ld_row[e] = row; ld_c8[e] = c8;
        ld_off[e] = idx < NP ? (unsigned)(row * p.ldkv + c8 * 8) * 2u : OOB;
    }
    qf32x4 kreg[LPT], vreg[LPT];
    auto load_tile = [&](const int seg, const int key0) {
        const size_t first = (kvbase[seg] + key0) * p.ldkv + head * D;
        const __amdgpu_buffer_rsrc_t rk = rsrc_of(K + first), rv = rsrc_of(V + first);
        const int left = p.Nk - key0;                           // keys this tile really has (uniform)
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            unsigned off = ld_off[e];
            if (left < KT) off = ld_row[e] < left ? off : OOB;  // ragged last tile of a segment only
            kreg[e] = __builtin_bit_cast(qf32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0));
            vreg[e] = __builtin_bit_cast(qf32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0));
        }
    };
    auto store_tile = [&](int buf) {
        char* Kl = smem_q + buf * STAGE;
        char* Vl = Kl + KBYTES;
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            if (tid + NT * e < NP) {
                *reinterpret_cast<qf32x4*>(Kl + ld_row[e] * KROW + ld_c8[e] * 16) = kreg[e];
                *reinterpret_cast<qf32x4*>(Vl + ld_row[e] * VROW + ld_c8[e] * 16) = vreg[e];
            }
        }
    };

    qf32x16 acc[2][T];
    qf32x16 negm[2];                              // -(reference maximum) of the lane's query, in every register of the block
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[b][r] = 0.f;
    }

    load_tile(0, 0);
    store_tile(0);
    __syncthreads();

    // transposing read of the V^T fragment (as in attn.hip): 16-lane group g = lane >> 4 covers value columns 16 (g & 1) .. + 15 of the
    // tile and keys 4 (g >> 1) .. + 3 of each 8-key half of the k-step; lane 4 q + p of the group supplies row q, columns 4 p ..
    const int ti = lane & 15;
    const int tr_off = (4 * h + (ti >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (ti & 3)) * 2;
    constexpr int THRESH_BITS = 0x41000000;        // 8.0f: positive floats compare like their bit patterns

    int key0 = 0;
    for (int tt = 0; tt < ntiles; ++tt) {
        const int buf = tt & 1;
        if (tt == tps) key0 = 0;                                  // second key segment
        if (tt + 1 < ntiles) {
            if (tt + 1 == tps) load_tile(1, 0); else load_tile(tt + 1 > tps ? 1 : 0, key0 + KT);
        }
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const int keyb = key0 + 32 * sub;                       // first key of this 32-key block
            if (active && keyb < p.Nk) {
                const char* Kl = smem_q + buf * STAGE + sub * 32 * KROW;
                const char* Vl = smem_q + buf * STAGE + KBYTES + sub * 32 * VROW;
                const bool first = tt == 0 && sub == 0;
                const char* kp = Kl + j * KROW + h * 16;
                hx8<H> kf[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) kf[s] = *reinterpret_cast<const hx8<H>*>(kp + s * 32);
                qf32x16 st[2];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        st[b] = mfma_32x32x16(kf[s], qf[b][s], s == 0 ? negm[b] : st[b]);
                if (keyb + 32 > p.Nk) {
                    asm volatile("" ::: "memory");              // (ragged last tile of a segment: a branch, not 32 selects on every tile)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = keyb + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (key >= p.Nk) { st[0][r] = -INFINITY; st[1][r] = -INFINITY; }
                    }
                }
                hx8<H> vf[T][2];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const char* vb = Vl + tr_off + t * 64;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const hx4<H> lo = lds_read_tr16<H>((vb + (16 * s) * VROW));       // keys 16s + 4h + 0..3
                        const hx4<H> hi = lds_read_tr16<H>((vb + (16 * s + 8) * VROW));   // keys 16s + 8 + 4h + 0..3
                        vf[t][s] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    // st = scaled score - reference maximum.  The maximum moves only when some score of the wave tops its row's by more than
                    // 2^8 (deferred maximum, attn.hip) -- such a score is positive, and positive floats order like their bit patterns:
                    int mi = imax3(__float_as_int(st[b][0]), __float_as_int(st[b][1]), __float_as_int(st[b][2]));
#pragma unroll
                    for (int r = 3; r < 15; r += 2) mi = imax3(mi, __float_as_int(st[b][r]), __float_as_int(st[b][r + 1]));
                    mi = max(mi, __float_as_int(st[b][15]));
                    if (first || __any(mi > THRESH_BITS)) {
                        asm volatile("" ::: "memory");          // (rare after the first tile)
                        float mt = st[b][0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) mt = __builtin_fmaxf(mt, st[b][r]);
                        {
                            // the other half-wave holds the other 16 keys of the same query (v_permlane32_swap: lanes 32-63 of a <-> lanes
                            // 0-31 of b; inline asm, the builtin drops its second result in ROCm 7.2; the wait states are inside the string)
                            float a = mt, c = mt;
                            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c));
                            mt = __builtin_fmaxf(a, c);
                        }
                        constexpr float DEFER = 8.0f;
                        // first tile: the row takes the tile's true maximum whatever its sign (nothing accumulated yet, alpha unused)
                        const float d = first ? mt : (mt > DEFER ? mt : 0.f);
                        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-d);
#pragma unroll
                        for (int r = 0; r < 16; ++r) { st[b][r] -= d; negm[b][r] -= d; }
#pragma unroll
                        for (int t = 0; t < T; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[b][t][r] *= alpha;
                    }
                    hx8<H> pf[2];                       // P^T fragments: registers 8s..8s+7 are k-step s as they stand
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int e = 0; e < 8; ++e) pf[s][e] = (H)__builtin_amdgcn_exp2f(st[b][8 * s + e]);
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int s = 0; s < 2; ++s) acc[b][t] = mfma_32x32x16(vf[t][s], pf[s], acc[b][t]);
                }
            }
        }
        if (tt + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
        key0 += KT;
    }

    if (active) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float l_tot = __shfl(acc[b][T - 1][LREG], j + 32 * LHALF);     // row D of O^T: lane (query j, half LHALF)
            const int qrow = q0 + 32 * b + j;
            if (qrow < p.Nq) {
                const float inv = 1.0f / l_tot;
                H* op = reinterpret_cast<H*>(p.o) + ((size_t)sf * p.Nq + qrow) * p.ldo + head * D;
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int dv = t * 32 + 8 * rg + 4 * h;
                        if (dv < D) {
                            hx4<H> o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (H)(acc[b][t][rg * 4 + e] * inv);
                            *reinterpret_cast<hx4<H>*>(op + dv) = o;
                        }
                    }
            }
        }
    }
}
#endif


// ---- software-pipelined form (d = 40) -------------------------------------------------------------------------------------------
// The kernel above still runs as phases -- a wave issues the score MFMAs of a key tile, WAITS for them, runs that tile's softmax on the
// vector ALU, then the PV MFMAs -- and leaves it to the second wave of the SIMD to fill the other pipe (measured: 884 TFLOP/s
// algorithmic at level 0 against an issue bound of ~1500).  Here a wave never waits for its own MFMAs: in iteration i the scores of key
// tile i + 1 (both query blocks: 6 MFMAs) are issued first and the vector ALU exponentiates tile i (whose scores the previous iteration
// produced) underneath them, then the PV MFMAs of tile i run while the maximum test of tile i + 1 and the K fragment reads of tile
// i + 2 issue.  One basic block per iteration: the ragged last key tile of a segment is a separate instance of the body, and moving a
// row's maximum is ONE rarely taken branch at the end of the iteration that repairs the already computed scores of tile i + 1.
//   * The running maximum rides in Q: d = 40 pads the third 16-deep k-step of K Q^T to 48, so column 40 of every K row in LDS holds
//     1.0 and slot 40 of a query's Q fragment holds -m: the MFMAs that compute the scores subtract the maximum, the 2 x 16 registers
//     of the -m accumulator block of the form above are gone.  m must be a bf16 number for that; a softmax may take ANY reference, so
//     the reference is the row maximum rounded to bf16 (the same P and the same denominator see it).
//   * Three LDS stages of 64 keys: during stage s the waves read V of stage s, K of stage s + 1 (scores run one tile ahead) and, at
//     its end, write stage s + 2 -- one barrier per 64 keys.
//   * P is bounded as before: a reference moves when a score tops it by more than 2^8.
// (Pinning the interleave of an iteration with sched_group_barrier -- one MFMA, 3 v_exp_f32 + 1-2 conversions, ... -- made the
// scheduler give up and bunch eight MFMAs back to back; the compiler's own interleave of the single basic block is the one kept.)
template <typename H, int NW>      // H: bf16 / fp16 (h16.h)
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void flash_attn_b16q64p_kernel(const AttnArgs p) {
    constexpr int D = 40;
    typedef Q64Layout<D> L;
    static_assert(L::DP > D, "the maximum rides in a spare k slot");
    constexpr int NT = 64 * NW, QB = 64 * NW, NBUF = 3;
    constexpr int KS = L::KS, T = L::T, KT = L::KT, KROW = L::KROW, VROW = L::VROW, KBYTES = L::KBYTES, STAGE = L::STAGE;
    constexpr int LROW = D % 32, LREG = 4 * (LROW / 8) + (LROW & 3), LHALF = (LROW >> 2) & 1;      // where row D of O^T lives
    constexpr int C8 = D / 8;
    constexpr int NP = KT * C8;
    constexpr int LPT = (NP + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem_q[];      // [3][K rows | V rows]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const Q64Block blk = q64_block(p, QB);
    if (!blk.valid) return;
    const int sf = blk.sf;
    const int smp = sf / p.F, f = sf - smp * p.F;
    const int head = blk.head;
    const int q0 = blk.qb * QB + wave * 64;
    const bool active = q0 < p.Nq;
    const H* __restrict__ Q = reinterpret_cast<const H*>(p.q);
    const H* __restrict__ K = reinterpret_cast<const H*>(p.k);
    const H* __restrict__ V = reinterpret_cast<const H*>(p.v);

    size_t kvbase[2];
    kvbase[0] = (size_t)(smp * p.F) * p.Nk;
    kvbase[1] = (size_t)(smp * p.F + (f > 0 ? f - 1 : 0)) * p.Nk;
    const int nseg = f >= 2 ? 2 : 1;            // frames 0 and 1 see [K0; K0]: softmax over a duplicated key set = softmax over the set
    const int tps = (p.Nk + KT - 1) / KT;
    const int NS = nseg * tps;                  // stages of 64 keys

    for (int i = tid * 16; i < NBUF * STAGE; i += NT * 16)               // pad columns are never rewritten: keep them finite
        *reinterpret_cast<qf32x4*>(smem_q + i) = qf32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int i = tid; i < NBUF * KT; i += NT) {
        char* st = smem_q + (i / KT) * STAGE;
        *reinterpret_cast<H*>(st + KBYTES + (i % KT) * VROW + D * 2) = (H)1.0f;      // V column D: row D of O^T = the denominator
    }

    hx8<H> qf[2][KS];
    {
        const float qs = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int qrow = min(q0 + 32 * b + j, p.Nq - 1);
            const H* qp = Q + ((size_t)sf * p.Nq + qrow) * p.ldq + head * D;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int k0 = 16 * s + 8 * h;
                hx8<H> a;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = (H)0.f;
                if (k0 < D) a = *reinterpret_cast<const hx8<H>*>(qp + k0);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = (H)((float)a[e] * qs);
                qf[b][s] = a;
            }
            if (h) qf[b][KS - 1][2] = (H)(-H16Traits<H>::mask_marker);      // slot D + 2: times the marker column of a key past its segment's end
        }
    }

    constexpr unsigned OOB = 0x80000000u;
    auto rsrc_of = [](const void* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), (short)0, 0x7FFFFFF0,
                                                 0x00020000);
    };
    int ld_row[LPT], ld_c8[LPT];
    unsigned ld_off[LPT];
#pragma unroll
    for (int e = 0; e < LPT; ++e) {
        const int idx = tid + NT * e;
        const int row = idx / C8, c8 = idx - row * C8;
        ld_row[e] = row; ld_c8[e] = c8;
        ld_off[e] = idx < NP ? (unsigned)(row * p.ldkv + c8 * 8) * 2u : OOB;
    }
    qf32x4 kreg[LPT], vreg[LPT];
    auto load_stage = [&](const int s) {                        // stage s of the key list -> staging registers
        const int seg = s >= tps ? 1 : 0;
        const int key0 = (s - seg * tps) * KT;
        const size_t first = (kvbase[seg] + key0) * p.ldkv + head * D;
        const __amdgpu_buffer_rsrc_t rk = rsrc_of(K + first), rv = rsrc_of(V + first);
        const int left = p.Nk - key0;                           // keys this stage really has (uniform)
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            unsigned off = ld_off[e];
            if (left < KT) off = ld_row[e] < left ? off : OOB;  // ragged last stage of a segment only: zero rows
            kreg[e] = __builtin_bit_cast(qf32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0));
            vreg[e] = __builtin_bit_cast(qf32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0));
        }
    };
    // Keys past the end of a segment (its ragged last stage) are masked by the MATRIX pipe as well: column D + 2 of their K rows holds
    // a marker (0 in every real key's row) and slot D + 2 of every query's Q fragment holds minus that marker, so their scores come out
    // of the MFMA at -marker^2 -- -2^120 in bf16, -2^30 in fp16 (H16Traits: whatever the row's reference maximum is, nothing a real score
    // reaches) -- and exponentiate to zero: no select in the loop, no second instance of its body.  The thread that stores the
    // first piece of a K row rewrites the row's two marker columns with every stage (the ring reuses the buffer).
    auto store_stage = [&](const int buf, const int left) {     // left: valid keys from the stage's first on
        char* Kl = smem_q + buf * STAGE;
        char* Vl = Kl + KBYTES;
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            if (tid + NT * e < NP) {
                *reinterpret_cast<qf32x4*>(Kl + ld_row[e] * KROW + ld_c8[e] * 16) = kreg[e];
                *reinterpret_cast<qf32x4*>(Vl + ld_row[e] * VROW + ld_c8[e] * 16) = vreg[e];
                if (ld_c8[e] == 0) {                            // columns D, D + 1: 1.0 (the two maximum slots); D + 2: the marker; D + 3: 0
                    unsigned* mk = reinterpret_cast<unsigned*>(Kl + ld_row[e] * KROW + D * 2);
                    mk[0] = (unsigned)H16Traits<H>::max_marker_bits * 0x00010001u;
                    mk[1] = ld_row[e] < left ? 0u : (unsigned)H16Traits<H>::mask_marker_bits;
                }
            }
        }
    };
    auto stage_keys = [&](const int s) { return p.Nk - (s - (s >= tps ? tps : 0)) * KT; };      // valid keys from the stage's first on

    qf32x16 acc[2][T];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;
    float mref[2] = {0.f, 0.f};                   // the reference maximum of the lane's query (the sum of two bf16 numbers)

    load_stage(0);
    store_stage(0, stage_keys(0));
    if (NS > 1) { load_stage(1); store_stage(1, stage_keys(1)); }
    __syncthreads();

    const int ti = lane & 15;
    const int k_off = j * KROW + h * 16;                                              // K fragment: key row j, columns 16 s + 8 h ..
    const int v_off = KBYTES + (4 * h + (ti >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (ti & 3)) * 2;      // V^T fragment (tr read), as above
    constexpr int THRESH_BITS = 0x41000000;        // 8.0f
    const qf32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    hx8<H> kf[KS];
    auto read_k = [&](const int buf, const int sub) {
        const char* kp = smem_q + buf * STAGE + sub * 32 * KROW + k_off;
#pragma unroll
        for (int s = 0; s < KS; ++s) kf[s] = *reinterpret_cast<const hx8<H>*>(kp + s * 32);
    };
    auto scores = [&](qf32x16 (&S)[2]) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s = 0; s < KS; ++s) S[b] = mfma_32x32x16(kf[s], qf[b][s], s == 0 ? zero16 : S[b]);
    };
    // Move the reference maxima of the rows whose new scores S (computed against the old references) top them by more than 2^8
    // (FIRST: take the tile's maximum whatever it is): new reference = a bf16 number, S and the accumulators follow, -m goes into Q.
    auto move_maximum = [&](qf32x16 (&S)[2], const bool first) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float mt = S[b][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = __builtin_fmaxf(mt, S[b][r]);
            {
                float a = mt, c = mt;       // the other half-wave holds the other 16 keys of the same query
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c));
                mt = __builtin_fmaxf(a, c);
            }
            constexpr float DEFER = 8.0f;
            const float d = first ? mt : (mt > DEFER ? mt : 0.f);
            // the new reference as TWO bf16 numbers (16 mantissa bits: within 2^-16 of the wanted value at any magnitude, so that a row
            // of huge scores neither overflows its probabilities nor underflows them all)
            // (the marker columns of K hold MM: 1.0 for bf16; 2.0 for fp16, whose range a reference of 6e4 x log2 e would leave)
            constexpr float MM = H16Traits<H>::max_marker;
            const float want = (mref[b] + d) * (1.0f / MM);
            const H m_hi = (H)want;
            const H m_lo = (H)(want - (float)m_hi);
            const float mnew = MM * ((float)m_hi + (float)m_lo);
            const float delta = mnew - mref[b];
            mref[b] = mnew;
            const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
            for (int r = 0; r < 16; ++r) S[b][r] -= delta;
            if (!first) {
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[b][t][r] *= alpha;
            }
            qf[b][KS - 1][0] = h ? (H)(-(float)m_hi) : qf[b][KS - 1][0];      // k = D, D + 1 live in the upper half-wave's fragment of the last k-step
            qf[b][KS - 1][1] = h ? (H)(-(float)m_lo) : qf[b][KS - 1][1];
        }
    };
    // One iteration: scores of the NEXT tile into Sn (from the K fragment registers), softmax numerators and PV of THIS tile (scores
    // Sc, V rows at vl), then the K fragments of the tile after next.
    auto iteration = [&](qf32x16 (&Sc)[2], qf32x16 (&Sn)[2], const char* vl, const int kbuf, const int ksub, const bool has_next) {
        hx8<H> vf[T][2];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const char* vb = vl + t * 64;
                const hx4<H> lo = lds_read_tr16<H>((vb + (16 * s) * VROW));       // keys 16s + 4h + 0..3
                const hx4<H> hi = lds_read_tr16<H>((vb + (16 * s + 8) * VROW));   // keys 16s + 8 + 4h + 0..3
                vf[t][s] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        scores(Sn);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            hx8<H> pf[2];                       // P^T fragments: registers 8s..8s+7 are k-step s as they stand
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[s][e] = (H)__builtin_amdgcn_exp2f(Sc[b][8 * s + e]);
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) acc[b][t] = mfma_32x32x16(vf[t][s], pf[s], acc[b][t]);
        }
        read_k(kbuf, ksub);
        int mi = imax3(__float_as_int(Sn[0][0]), __float_as_int(Sn[0][1]), __float_as_int(Sn[0][2]));
#pragma unroll
        for (int r = 3; r < 15; r += 2) mi = imax3(mi, __float_as_int(Sn[0][r]), __float_as_int(Sn[0][r + 1]));
#pragma unroll
        for (int r = 0; r < 16; r += 2) mi = imax3(mi, __float_as_int(Sn[1][r]), __float_as_int(Sn[1][r + 1]));
        mi = max(mi, __float_as_int(Sn[0][15]));
        const bool tops = __any(mi > THRESH_BITS);
        if (tops & has_next) {
            asm volatile("" ::: "memory");          // (rare: keep it a branch)
            move_maximum(Sn, false);
        }
    };

    if (active) {                                   // scores of tile 0, its maxima = the first references
        qf32x16 S0[2], S1[2];
        read_k(0, 0);
        scores(S0);
        move_maximum(S0, true);
        read_k(0, 1);
        int b0 = 0;                                 // buffer of stage s
        for (int s = 0; s < NS; ++s) {
            if (s + 2 < NS) load_stage(s + 2);
            const int b1 = b0 == NBUF - 1 ? 0 : b0 + 1, b2 = b1 == NBUF - 1 ? 0 : b1 + 1;
            const char* vl = smem_q + b0 * STAGE + v_off;
            iteration(S0, S1, vl, b1, 0, true);
            iteration(S1, S0, vl + 32 * VROW, b1, 1, s + 1 < NS);
            if (s + 2 < NS) store_stage(b2, stage_keys(s + 2));
            __syncthreads();
            b0 = b1;
        }
    } else {                                        // a wave without queries only stages
        int b0 = 0;
        for (int s = 0; s < NS; ++s) {
            if (s + 2 < NS) load_stage(s + 2);
            const int b1 = b0 == NBUF - 1 ? 0 : b0 + 1, b2 = b1 == NBUF - 1 ? 0 : b1 + 1;
            if (s + 2 < NS) store_stage(b2, stage_keys(s + 2));
            __syncthreads();
            b0 = b1;
        }
    }

    if (active) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float l_tot = __shfl(acc[b][T - 1][LREG], j + 32 * LHALF);     // row D of O^T: lane (query j, half LHALF)
            const int qrow = q0 + 32 * b + j;
            if (qrow < p.Nq) {
                const float inv = 1.0f / l_tot;
                H* op = reinterpret_cast<H*>(p.o) + ((size_t)sf * p.Nq + qrow) * p.ldo + head * D;
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int dv = t * 32 + 8 * rg + 4 * h;
                        if (dv < D) {
                            hx4<H> o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (H)(acc[b][t][rg * 4 + e] * inv);
                            *reinterpret_cast<hx4<H>*>(op + dv) = o;
                        }
                    }
            }
        }
    }
}

template <int NW>
static void launch_q64p(const AttnArgs& a, hipStream_t s) {
    typedef Q64Layout<40> L;
    constexpr size_t smem = 3 * (size_t)L::STAGE;
    h16_dispatch(a.io_bf16, [&](auto h16_tag) {
        using H = decltype(h16_tag);
        E2V_KATTR((flash_attn_b16q64p_kernel<H, NW>), smem);
        E2V_KLAUNCH((flash_attn_b16q64p_kernel<H, NW>), dim3(q64_grid(a, 64 * NW)), dim3(64 * NW), smem, s, a);
    });
}

#ifdef E2V_AB
template <int D, int NW>
static void launch_q64(const AttnArgs& a, hipStream_t s) {
    typedef Q64Layout<D> L;
    constexpr size_t smem = 2 * (size_t)L::STAGE;
    h16_dispatch(a.io_bf16, [&](auto h16_tag) {
        using H = decltype(h16_tag);
        E2V_KATTR((flash_attn_b16q64_kernel<H, D, NW>), smem);
        E2V_KLAUNCH((flash_attn_b16q64_kernel<H, D, NW>), dim3(q64_grid(a, 64 * NW)), dim3(64 * NW), smem, s, a);
    });
}
#endif

// Which (if any) instance of the 64-queries-per-wave kernel serves the call: waves per workgroup, 0 = none (the caller falls back to
// flash_attn_b16io_kernel).  A rule of the SHAPE only (never of the batch: the two kernels round differently, and a clip's bits must
// not depend on how many clips run together).
int flash_attention_q64_waves(const AttnArgs& a) {
    // 0: every bf16 self-attention through flash_attn_b16io_kernel; 1: d = 40 here; 2: d = 80 as well (287+ registers: one wave per SIMD)
    static const int* const on = knob("E2V_ATTN_Q64", 1);
#ifdef E2V_AB
    const bool d80 = a.D == 80 && *on >= 2;
#else
    const bool d80 = false;
#endif
    if (!*on || !a.io_bf16 || a.mode != 0 || !(a.D == 40 || d80) || a.Nk <= 32) return 0;
    if (a.Nq < 128) return 0;
    static const int* const force_nw = E2V_AB_KNOB("E2V_ATTN_Q64_NW", 0);     // 2..4: that many waves per workgroup whatever the padding (A/B)
    if (*force_nw >= 2 && *force_nw <= 4) return *force_nw;
    int best = 0, waste = 1 << 30;
    for (int nw = 4; nw >= 2; --nw) {                              // least padding of the last query block; ties: the larger workgroup
        const int qb = 64 * nw, w = (a.Nq + qb - 1) / qb * qb - a.Nq;
        if (w < waste) { waste = w; best = nw; }
    }
    return best;
}

bool flash_attention_q64(const AttnArgs& a, hipStream_t s) {
    const int nw = flash_attention_q64_waves(a);
    if (!nw) return false;
    const double probs = (double)a.n * a.F * a.heads;
    std::string pname = a.io_bf16 == H16_FP16 ? "flash_attn_fp16_sparse_causal" : "flash_attn_bf16_sparse_causal";
    if (prof_detail()) pname += attn_shape_tag(a);
    ProfScope ps(pname.c_str(), 4.0 * probs * a.Nq * 2.0 * a.Nk * a.D, 2.0 * probs * a.D * (2.0 * a.Nq + 2.0 * a.Nk), s);
    static const int* const pipelined = E2V_AB_KNOB("E2V_ATTN_Q64P", 1);      // 0: the phase-by-phase form of the 64-query kernel
    dry_tag(std::string(a.D == 40 && *pipelined ? " -> flash_attn_b16q64p_kernel" : " -> flash_attn_b16q64_kernel") + " w" + std::to_string(nw));
    if (a.D == 40 && *pipelined) { if (nw == 4) launch_q64p<4>(a, s); else if (nw == 3) launch_q64p<3>(a, s); else launch_q64p<2>(a, s); }
#ifdef E2V_AB
    else if (a.D == 40) { if (nw == 4) launch_q64<40, 4>(a, s); else if (nw == 3) launch_q64<40, 3>(a, s); else launch_q64<40, 2>(a, s); }
    else           { if (nw == 4) launch_q64<80, 4>(a, s); else if (nw == 3) launch_q64<80, 3>(a, s); else launch_q64<80, 2>(a, s); }
#endif
    return true;
}

}  // namespace e2v
