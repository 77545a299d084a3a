// Fused fp32 attention for the three attentions of BasicTransformerBlock (attention.py:232-269).
//
// flash_attention: sparse-causal self-attention (attn1, attention.py:272-328) and cross-attention to
// the cond tokens (attn2).  The reference gathers and concatenates K/V of frame 0 and frame i-1
// (attention.py:292-301) and materialises a [48, N, 2N] score tensor per sample; here the two key
// segments are walked in place with an online softmax, nothing is concatenated or materialised.
// Frames 0 and 1 both see [K0; K0]; softmax over a duplicated key set equals softmax over the set,
// so they walk frame 0 once.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact fp32): one wave owns 32 queries.  S^T = K Q^T puts the
// QUERY on the lane (C/D column) and the 32 keys of the tile in the 16 accumulator registers of the
// two lane halves, so the row max / row sum of the online softmax are per-lane scalars (one
// cross-half exchange), and P^T is already in the B-operand layout of O^T = V^T P^T: no LDS round
// trip, no transposes.  K and V tiles (32 keys) are staged through LDS, double-buffered.
#include "h16.h"
#include "kernels.h"
#include "prof.h"
#include "runtime.h"
#include "act_io.h"

#include <cstdlib>
#include <type_traits>

namespace e2v {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Which (query block, head, sample-frame) a workgroup works on.  The grid is one-dimensional and blocks are dealt to the 8 XCDs
// round-robin (block b runs on XCD b % 8), each XCD with its own 4 MB L2: XCD x takes whole SAMPLES x, x + 8, ... (whole
// sample-frames when there are fewer than 8 samples) and walks them frame by frame, head by head, so that the K / V rows of a frame
// -- read by the 18 query blocks of all 8 heads of that frame and, as the "previous frame", of the next one -- are fetched into
// ONE L2 instead of all eight (PMC at level 0, B = 32, with the former (query block, head, sample-frame) 3-D grid: 30 GB through
// the fabric per launch against 2.3 GB of q, k, v, o; L2 hit rate 0.64).  Placement is a speed matter only.
// The unit dealt to an XCD is the coarsest one that still divides evenly over the 8 XCDs (round 5; it used to be "samples when n >= 8,
// else sample-frames", which at the reference's clip-by-clip operating point -- 2 samples x 6 frames = 12 units -- gave four XCDs two
// frames and four XCDs one: a third of the chip idle for the launch's last third): whole samples when n % 8 == 0, else sample-frames
// when n F % 8 == 0, else (sample-frame, head) pairs (8 heads: always even).
struct AttnBlock { int qb, head, sf; bool valid; };
__host__ __device__ __forceinline__ int attn_xcd_unit(const int n, const int F, const int heads) {
    return n % 8 == 0 ? 0 : (n * F) % 8 == 0 ? 1 : 2;
}
__device__ __forceinline__ AttnBlock attn_block_of(const AttnArgs& p, const int nqb) {
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int S = p.n * p.F;
    const int unit = attn_xcd_unit(p.n, p.F, p.heads);
    AttnBlock r;
    if (unit == 0) {                                 // units = samples
        const int per = p.F * p.heads * nqb;
        const int ul = idx / per, w = idx - ul * per;
        const int smp = ul * 8 + xcd;
        const int f = w / (p.heads * nqb), w2 = w - f * (p.heads * nqb);
        r.sf = smp * p.F + f; r.head = w2 / nqb; r.qb = w2 - r.head * nqb; r.valid = smp < p.n;
    } else if (unit == 1) {                          // units = sample-frames
        const int per = p.heads * nqb;
        const int ul = idx / per, w = idx - ul * per;
        r.sf = ul * 8 + xcd; r.head = w / nqb; r.qb = w - r.head * nqb; r.valid = r.sf < S;
    } else {                                         // units = (sample-frame, head) pairs
        const int ul = idx / nqb;
        const int pair = ul * 8 + xcd;
        r.qb = idx - ul * nqb; r.sf = pair / p.heads; r.head = pair - r.sf * p.heads; r.valid = pair < S * p.heads;
    }
    return r;
}
__device__ __forceinline__ AttnBlock attn_block(const AttnArgs& p) { return attn_block_of(p, (p.Nq + 127) / 128); }
static inline unsigned attn_grid_of(const AttnArgs& a, const unsigned nqb) {
    const int unit = attn_xcd_unit(a.n, a.F, a.heads);
    if (unit == 0) return 8u * (a.n / 8) * (unsigned)(a.F * a.heads) * nqb;
    if (unit == 1) return 8u * ((a.n * a.F) / 8) * (unsigned)a.heads * nqb;
    return 8u * ((a.n * a.F * a.heads + 7) / 8) * nqb;
}
static inline unsigned attn_grid(const AttnArgs& a) { return attn_grid_of(a, (a.Nq + 127) / 128); }

// shape tag of a launch for the detailed profile / the dispatch record
std::string attn_shape_tag(const AttnArgs& a) {
    return " D" + std::to_string(a.D) + " Nq" + std::to_string(a.Nq) + " Nk" + std::to_string(a.Nk) + " n" + std::to_string(a.n) + " F" + std::to_string(a.F) +
           " h" + std::to_string(a.heads);
}

template <int D>
__global__ __launch_bounds__(256) void flash_attn_kernel(const AttnArgs p) {
    constexpr int LD = D + 4;               // padded LDS row: conflict-free ds_read_b128 of 16 rows
    // O^T is built from 32-row MFMA tiles over the value columns.  When 8 columns are left over (d = 40: 32 + 8)
    // they are accumulated on the VALU from the same P registers (128 FMAs per key tile, on the vector pipe while the
    // matrix pipe runs the next MFMAs) instead of a second MFMA tile with 24 dead rows (16 of 52 MFMAs per tile).
    constexpr int TF = D / 32;
    constexpr bool VREM = (D - 32 * TF) == 8;
    constexpr int T = VREM ? TF : (D + 31) / 32;
    constexpr int TA = T > 0 ? T : 1;
    constexpr int G = D / 8;
    constexpr int DQ = D / 4;
    constexpr int KT = 32;                   // keys staged per barrier interval (64 measured no faster)
    constexpr int NF4 = KT * DQ;
    constexpr int LPT = (NF4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2 buf][K | V][KT][LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const AttnBlock blk = attn_block(p);
    if (!blk.valid) return;
    const int sf = blk.sf;
    const int smp = sf / p.F, f = sf - smp * p.F;
    const int head = blk.head;
    const int q0 = (blk.qb * 4 + wave) * 32;
    const bool active = q0 < p.Nq;

    int nseg = 1;
    size_t kvbase[2];
    if (p.mode == 0) {
        kvbase[0] = (size_t)(smp * p.F) * p.Nk;
        kvbase[1] = (size_t)(smp * p.F + (f > 0 ? f - 1 : 0)) * p.Nk;
        nseg = f >= 2 ? 2 : 1;
    } else {
        kvbase[0] = kvbase[1] = (size_t)smp * p.Nk;
    }
    const int tps = (p.Nk + KT - 1) / KT;
    const int ntiles = nseg * tps;

    // Q fragments, pre-multiplied by scale * log2(e): scores live in the exp2 domain
    f32x4 qf[G];
    {
        const int qrow = min(q0 + j, p.Nq - 1);
        const float* qp = p.q + ((size_t)sf * p.Nq + qrow) * p.ldq + head * D + h * 4;
        const float c = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            qf[g] = *reinterpret_cast<const f32x4*>(qp + g * 8);
            qf[g] *= c;
        }
    }

    // K/V staging: the (row, column) a thread fetches never changes, only the tile's first key does.  Loads go through
    // buffer descriptors rebuilt per tile from the (wave-uniform) address of that first key, so a lane's offset is a
    // loop-invariant 32-bit number and keys past the end of a ragged last tile are fetched with an offset outside the
    // descriptor window -- the buffer unit returns zeros, no per-element select (VALU time is not hidden behind the fp32
    // MFMA on gfx950: this took the staging from ~60 to ~10 vector instructions per tile).
    constexpr unsigned OOB = 0x80000000u;
    auto rsrc_of = [](const void* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), (short)0, 0x7FFFFFF0,
                                                 0x00020000);
    };
    int ld_row[LPT];
    unsigned ld_off[LPT];
#pragma unroll
    for (int e = 0; e < LPT; ++e) {
        const int idx = tid + 256 * e;
        const int row = idx / DQ, c4 = idx - row * DQ;
        ld_row[e] = row;
        ld_off[e] = idx < NF4 ? (unsigned)(row * p.ldkv + c4 * 4) * 4u : OOB;
    }
    f32x4 kreg[LPT], vreg[LPT];
    auto load_tile = [&](int tt) {
        const int seg = tt / tps;
        const int key0 = (tt - seg * tps) * KT;
        const size_t first = (kvbase[seg] + key0) * p.ldkv + head * D;
        const __amdgpu_buffer_rsrc_t rk = rsrc_of(p.k + first), rv = rsrc_of(p.v + first);
        const int left = p.Nk - key0;                           // keys this tile really has (uniform)
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            unsigned off = ld_off[e];
            if (left < KT) off = ld_row[e] < left ? off : OOB;  // ragged last tile of a segment only
            kreg[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0));
            vreg[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0));
        }
    };
    auto store_tile = [&](int buf) {
        float* Ks = smem + buf * (2 * KT * LD);
        float* Vs = Ks + KT * LD;
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            const int idx = tid + 256 * e;
            if (idx < NF4) {
                const int row = idx / DQ, c4 = idx - row * DQ;
                *reinterpret_cast<f32x4*>(Ks + row * LD + c4 * 4) = kreg[e];
                *reinterpret_cast<f32x4*>(Vs + row * LD + c4 * 4) = vreg[e];
            }
        }
    };

    f32x16 acc[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4 rem0 = {0.f, 0.f, 0.f, 0.f}, rem1 = {0.f, 0.f, 0.f, 0.f};     // value columns 32*TF+0..3 / +4..7
    float m_i = -INFINITY, l_i = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int tt = 0; tt < ntiles; ++tt) {
        const int buf = tt & 1;
        if (tt + 1 < ntiles) load_tile(tt + 1);
        const int seg_c = tt / tps;
        const int kbase = (tt - seg_c * tps) * KT;
        // one 32-key tile; `ragged` (compile-time) adds the key mask of a segment's last, partial tile
        auto tile_body = [&](const int key0, const int sub, auto ragged) {
            const float* Ks = smem + buf * (2 * KT * LD) + sub * 32 * LD;
            const float* Vs = smem + buf * (2 * KT * LD) + KT * LD + sub * 32 * LD;
            // S^T[key][query] = sum_d K[key][d] Q[query][d]
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 st;
            const float* kp = Ks + j * LD + h * 4;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + g * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[g][s], (g == 0 && s == 0) ? zero16 : st, 0, 0, 0);
            }
            // online softmax; register r of lane half h holds key  key0 + (r&3) + 8(r>>2) + 4h.
            // On gfx950 the fp32 MFMA runs at the vector rate and VALU instructions are NOT hidden behind it
            // (measured: kernel cycles = MFMA busy + VALU active), so the softmax is kept lean: key masking only on
            // the ragged last tile, raw v_exp_f32, and the accumulator rescale only when some row's max moved.
            if constexpr (decltype(ragged)::value) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (key >= p.Nk) st[r] = -INFINITY;
                }
            }
            float mt = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = __builtin_fmaxf(mt, st[r]);
            mt = __builtin_fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = __builtin_fmaxf(m_i, mt);
            const bool moved = __any(m_new > m_i);                  // wave-uniform
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(st[r] - m_new);
                ps += st[r];
            }
            float alpha = 1.0f;
            if (moved) {
                alpha = __builtin_amdgcn_exp2f(m_i - m_new);
                l_i *= alpha;
            }
            l_i += ps;
            m_i = m_new;
            // O^T[dv][query] += sum_key V[key][dv] P[query][key]
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (moved) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] *= alpha;
                }
                const int dv = min(t * 32 + j, D - 1);
                const float* vp = Vs + (4 * h) * LD + dv;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float vv = vp[((r & 3) + 8 * (r >> 2)) * LD];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, st[r], acc[t], 0, 0, 0);
                }
            }
            if constexpr (VREM) {      // value columns 32*TF .. 32*TF+7 for this lane's 16 keys
                if (moved) {
                    rem0 *= alpha;
                    rem1 *= alpha;
                }
                const float* vr = Vs + (4 * h) * LD + 32 * TF;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* row = vr + ((r & 3) + 8 * (r >> 2)) * LD;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(row);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(row + 4);
                    rem0 += v0 * st[r];
                    rem1 += v1 * st[r];
                }
            }
        };
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const int key0 = kbase + 32 * sub;
            if (active && key0 < p.Nk) {
                if (key0 + 32 <= p.Nk) tile_body(key0, sub, std::false_type{});
                else tile_body(key0, sub, std::true_type{});
            }
        }
        if (tt + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    // lanes l and l+32 hold the two key halves of the same query: fold the partial row sums
    const float l_tot = l_i + __shfl_xor(l_i, 32);
    if constexpr (VREM) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            rem0[e] += __shfl_xor(rem0[e], 32);
            rem1[e] += __shfl_xor(rem1[e], 32);
        }
    }
    if (active && q0 + j < p.Nq) {
        const float inv = 1.0f / l_tot;
        float* op = p.o + ((size_t)sf * p.Nq + q0 + j) * p.ldo + head * D;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int dv = t * 32 + 8 * rg + 4 * h;
                if (dv < D) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[t][rg * 4 + e] * inv;
                    *reinterpret_cast<f32x4*>(op + dv) = o;
                }
            }
        if constexpr (VREM) {          // half h writes columns 32*TF + 4h .. +3
            const f32x4 o = (h ? rem1 : rem0) * inv;
            *reinterpret_cast<f32x4*>(op + 32 * TF + 4 * h) = o;
        }
    }
}

typedef __bf16 abf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));

// ---- bf16 in HBM (bf16-activation mode, AttnArgs::io_bf16) ----------------------------------------------------
// Q, K, V arrive as bf16 rows and O leaves as bf16: nothing is converted on the way in.  K and V tiles go to LDS as they
// are (16-byte pieces, row-major, padded rows); the V^T fragment of O^T = V^T P^T comes out of the row-major V image
// through the transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads 4 keys x 16 value columns and gets them
// column-major), so there is no transposed staging.  The softmax scale rides in the exponent's FMA instead of being
// multiplied into a bf16 Q.  Scores, softmax and the O accumulator are fp32.
// FOLD: the running maximum is subtracted by the MATRIX pipe -- Q is pre-multiplied by scale * log2(e) (rounded to bf16 once), and the
// score chain K Q^T starts from an accumulator block that holds -m (16 registers, rewritten only when a row's maximum moves), so
// that the scores come out of the MFMA as exponents: the 16 fused multiply-adds per key tile of the plain form disappear from a loop
// that is bound by its vector instructions (16 v_exp_f32 + 8 conversions + the maximum are what is left).
// KT: keys per LDS stage (32 or 64: one barrier per 64 keys -- the four waves of a workgroup sit on four SIMDs, each shared with
// three other workgroups, and every barrier couples them).
template <typename H, int D, bool FOLD, int KT>      // H: bf16 / fp16 (h16.h)
__global__ __launch_bounds__(256) void flash_attn_b16io_kernel(const AttnArgs p) {
    constexpr int DP = (D + 15) / 16 * 16;      // head dim padded to the 16-deep MFMA step
    constexpr int KS = DP / 16;
    constexpr int T = (D + 31) / 32;            // 32-row tiles of O^T
    constexpr int KROW = DP * 2 + 16;           // bytes per K row: conflict-free ds_read_b128 of 16 rows
    // bytes per V row: T * 64 of data (the tr reads of the last tile may run past D: they stay inside the row), and a row stride of
    // 16 or 48 dwords mod 64, so that the four rows a half-wave's transposing read touches (2 x 8 dwords each) tile the 64 banks exactly;
    // with the former T * 64 + 16 the reads of rows q and q + 2 met on 8 banks (SQ_LDS_BANK_CONFLICT: 40 % of the LDS cycles)
    constexpr int VROW = ((T * 16) % 32 == 16) ? T * 64 : T * 64 + 64;
    constexpr int KBYTES = KT * KROW, VBYTES = KT * VROW;
    constexpr int STAGE = (KBYTES + VBYTES + 15) / 16 * 16;
    // Row sums on the matrix pipe: when D is not a multiple of 32 the last 32-row tile of O^T has spare rows, and a value
    // column of ONES at index D makes row D of O^T the softmax denominator (rescaled with the accumulator, summed over the
    // same bf16-rounded probabilities as the numerator) -- 16 vector adds per key tile less in a loop that is VALU-bound.
    constexpr bool SUMV = (D % 32) != 0;
    constexpr int LROW = D % 32, LREG = 4 * (LROW / 8) + (LROW & 3), LHALF = (LROW >> 2) & 1;
    constexpr int C8 = D / 8;                   // 16-byte pieces per row
    constexpr int NP = KT * C8;
    constexpr int LPT = (NP + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem_h[];      // [2][K rows | V rows]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const AttnBlock blk = attn_block(p);
    if (!blk.valid) return;
    const int sf = blk.sf;
    const int smp = sf / p.F, f = sf - smp * p.F;
    const int head = blk.head;
    const int q0 = (blk.qb * 4 + wave) * 32;
    const bool active = q0 < p.Nq;
    const H* __restrict__ Q = reinterpret_cast<const H*>(p.q);
    const H* __restrict__ K = reinterpret_cast<const H*>(p.k);
    const H* __restrict__ V = reinterpret_cast<const H*>(p.v);

    int nseg = 1;
    size_t kvbase[2];
    if (p.mode == 0) {
        kvbase[0] = (size_t)(smp * p.F) * p.Nk;
        kvbase[1] = (size_t)(smp * p.F + (f > 0 ? f - 1 : 0)) * p.Nk;
        nseg = f >= 2 ? 2 : 1;
    } else {
        kvbase[0] = kvbase[1] = (size_t)smp * p.Nk;
    }
    const int tps = (p.Nk + KT - 1) / KT;
    const int ntiles = nseg * tps;

    for (int i = tid * 16; i < 2 * STAGE; i += 256 * 16)                // pad columns are never rewritten: keep them finite
        *reinterpret_cast<f32x4*>(smem_h + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (SUMV) {
        __syncthreads();
        if (tid < 2 * KT)                                                   // value column D of every key row, both stages: 1.0
            *reinterpret_cast<H*>(smem_h + (tid / KT) * STAGE + KBYTES + (tid % KT) * VROW + D * 2) = (H)1.0f;
    }

    hx8<H> qf[KS];
    {
        const int qrow = min(q0 + j, p.Nq - 1);
        const H* qp = Q + ((size_t)sf * p.Nq + qrow) * p.ldq + head * D;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k0 = 16 * s + 8 * h;
            hx8<H> a;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (H)0.f;
            if (k0 < D) a = *reinterpret_cast<const hx8<H>*>(qp + k0);
            if constexpr (FOLD) {
                const float qs = p.scale * 1.44269504088896340736f;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = (H)((float)a[e] * qs);
            }
            qf[s] = a;
        }
    }
    const float sc = p.scale * 1.44269504088896340736f;                 // scores are exponentiated as exp2(sc * s - m)
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
        const int idx = tid + 256 * e;
        const int row = idx / C8, c8 = idx - row * C8;
        ld_row[e] = row; ld_c8[e] = c8;
        ld_off[e] = idx < NP ? (unsigned)(row * p.ldkv + c8 * 8) * 2u : OOB;
    }
    f32x4 kreg[LPT], vreg[LPT];
    auto load_tile = [&](const int seg, const int key0) {        // position of the tile, kept incrementally by the caller
        const size_t first = (kvbase[seg] + key0) * p.ldkv + head * D;
        const __amdgpu_buffer_rsrc_t rk = rsrc_of(K + first), rv = rsrc_of(V + first);
        const int left = p.Nk - key0;                           // keys this tile really has (uniform)
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            unsigned off = ld_off[e];
            if (left < KT) off = ld_row[e] < left ? off : OOB;  // ragged last tile of a segment only
            kreg[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0));
            vreg[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0));
        }
    };
    auto store_tile = [&](int buf) {
        char* Kl = smem_h + buf * STAGE;
        char* Vl = Kl + KBYTES;
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            if (tid + 256 * e < NP) {
                *reinterpret_cast<f32x4*>(Kl + ld_row[e] * KROW + ld_c8[e] * 16) = kreg[e];
                *reinterpret_cast<f32x4*>(Vl + ld_row[e] * VROW + ld_c8[e] * 16) = vreg[e];
            }
        }
    };

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;
    f32x16 negm;                                 // FOLD: -(running maximum) of the lane's query, in every register of the block
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;

    load_tile(0, 0);
    store_tile(0);
    __syncthreads();

    // transposing read of the V^T fragment: 16-lane group g = lane >> 4 covers value columns 16 (g & 1) .. + 15 of the tile
    // and keys 4 (g >> 1) .. + 3 of each 8-key half of the k-step; lane 4 q + p of the group supplies row q, columns 4 p ..
    const int ti = lane & 15;
    const int tr_off = (4 * h + (ti >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (ti & 3)) * 2;

    int key0 = 0;
    for (int tt = 0; tt < ntiles; ++tt) {
        const int buf = tt & 1;
        if (tt == tps) key0 = 0;                                  // second key segment
        if (tt + 1 < ntiles) {
            if (tt + 1 == tps) load_tile(1, 0); else load_tile(tt + 1 > tps ? 1 : 0, key0 + KT);
        }
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
        const int keyb = key0 + 32 * sub;                           // first key of this 32-key block
        if (active && keyb < p.Nk) {
            const char* Kl = smem_h + buf * STAGE + sub * 32 * KROW;
            const char* Vl = smem_h + buf * STAGE + KBYTES + sub * 32 * VROW;
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 st;
            const char* kp = Kl + j * KROW + h * 16;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const hx8<H> kf = *reinterpret_cast<const hx8<H>*>(kp + s * 32);
                st = mfma_32x32x16(kf, qf[s], s == 0 ? (FOLD ? negm : zero16) : st);
            }
            if (keyb + 32 > p.Nk) {
                asm volatile("" ::: "memory");                  // (rare path: keep it a branch -- if-converted, its 48 selects run on every tile)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = keyb + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (key >= p.Nk) st[r] = -INFINITY;
                }
            }
            float mt = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = __builtin_fmaxf(mt, st[r]);      // (a chain of v_max3_f32)
            float alpha = 1.0f;
            bool moved;
            if constexpr (FOLD) {
                // st = scaled score - m_i: a positive entry is a new maximum.  (First tile: m_i = 0 stands in for "no maximum yet";
                // the row takes the tile's true maximum whatever its sign.)  The other half-wave's maximum of the same query comes
                // through v_permlane32_swap (one instruction; __shfl_xor is a ds_bpermute with seven instructions of address arithmetic)
                {
                    // v_permlane32_swap a, b: lanes 32-63 of a <-> lanes 0-31 of b.  From two copies of mt: a = the lower half's value in
                    // both halves, b = the upper half's.  (Inline asm: given through the builtin, hipcc of ROCm 7.2 drops the second
                    // result.  The two wait states a VALU-written operand needs in front of a permlane are inside the string.)
                    float a = mt, b = mt;
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                    mt = __builtin_fmaxf(a, b);
                }
                // Deferred maximum: with 32 queries per wave SOME row sees a new maximum in most key tiles (on random data in half of them
                // after a thousand keys), and the wave-level rescale path -- 16 + 16 subtractions, 32 multiplies, an exp -- then runs
                // nearly always.  A row's reference maximum is therefore moved only when a score exceeds it by more than 2^8: the
                // probabilities stay below 256 (fp32 accumulators; bf16 rounding is relative), the normaliser is summed over the same
                // probabilities, and the result is the same softmax.
                constexpr float DEFER = 8.0f;
                const bool first = tt == 0 && sub == 0;
                const float d = first ? mt : (mt > DEFER ? mt : 0.f);
                moved = __any(d != 0.f);
                if (moved) {
                    asm volatile("" ::: "memory");              // (rare after the first tiles: a branch, not 32 subtractions of zero per tile)
                    alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-d);       // (first tile: nothing accumulated yet, and -d may be large)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { st[r] -= d; negm[r] -= d; }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = __builtin_amdgcn_exp2f(st[r]);
            } else {
                mt = __builtin_fmaxf(mt, __shfl_xor(mt, 32)) * sc;             // sc > 0: max commutes with the scaling
                const float m_new = __builtin_fmaxf(m_i, mt);
                moved = __any(m_new > m_i);
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], sc, -m_new));
                    if constexpr (!SUMV) ps += st[r];
                }
                if (moved) {
                    alpha = __builtin_amdgcn_exp2f(m_i - m_new);
                    if constexpr (!SUMV) l_i *= alpha;
                }
                if constexpr (!SUMV) l_i += ps;
                m_i = m_new;
            }
            if constexpr (FOLD && !SUMV) {
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) ps += st[r];
                l_i = l_i * alpha + ps;
            }
            hx8<H> pf[2];                       // P^T fragments: registers 8s..8s+7 are k-step s as they stand
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[s][e] = (H)st[8 * s + e];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (moved) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] *= alpha;
                }
                const char* vb = Vl + tr_off + t * 64;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const hx4<H> lo = lds_read_tr16<H>((vb + (16 * s) * VROW));       // keys 16s + 4h + 0..3
                    const hx4<H> hi = lds_read_tr16<H>((vb + (16 * s + 8) * VROW));   // keys 16s + 8 + 4h + 0..3
                    const hx8<H> vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[t] = mfma_32x32x16(vf, pf[s], acc[t]);
                }
            }
        }
        }
        if (tt + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
        key0 += KT;
    }

    float l_tot;
    if constexpr (SUMV) l_tot = __shfl(acc[T - 1][LREG], j + 32 * LHALF);     // row D of O^T: lane (query j, half LHALF)
    else l_tot = l_i + __shfl_xor(l_i, 32);
    if (active && q0 + j < p.Nq) {
        const float inv = 1.0f / l_tot;
        H* op = reinterpret_cast<H*>(p.o) + ((size_t)sf * p.Nq + q0 + j) * p.ldo + head * D;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int dv = t * 32 + 8 * rg + 4 * h;
                if (dv < D) {
                    hx4<H> o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (H)(acc[t][rg * 4 + e] * inv);
                    *reinterpret_cast<hx4<H>*>(op + dv) = o;
                }
            }
    }
}

// ---- cross-attention with the keys RESIDENT in LDS (bf16 rows; mode 1, Nk <= 96: the 77 conditioning tokens) -----------------------
// The K / V of a (sample, head) are the same 77 rows for every frame and every query of the sample, 6 KB at d = 40.  The kernel
// above stages them per 128-query workgroup (two barriers, a 37 KB LDS clear, a prologue and an epilogue for 0.4 us of MFMAs:
// 177 TFLOP/s, 2.4 TB/s of the Q / O rows it exists to stream).  Here a workgroup stages them ONCE and its four waves walk TPW
// 32-query tiles each of the sample's F * Nq rows with no barrier in the loop: all scores of a query against the 96 (padded) keys
// are in registers at once, so the softmax is one pass (no running maximum, no rescale), the next tile's Q rows are requested
// before this tile is computed, and the denominator is again row D of O^T.  The maximum is subtracted only when a row needs it
// (|max| > 8, wave-uniform branch): with scores pre-scaled into the exp2 domain by the bf16 Q, probabilities stay within
// [2^-8, 2^8] otherwise and the normalised result is the same.
template <typename H, int D>
__global__ __launch_bounds__(256) void cross_attn_resident_kernel(const AttnArgs p, const int tpw, const int chunks) {
    constexpr int DP = (D + 15) / 16 * 16;
    constexpr int KS = DP / 16;
    constexpr int T = (D + 31) / 32;
    constexpr int KROW = DP * 2 + 16;
    constexpr int VROW = ((T * 16) % 32 == 16) ? T * 64 : T * 64 + 64;
    constexpr int NKT = 3, NKEY = 32 * NKT;
    constexpr int KBYTES = NKEY * KROW;
    constexpr bool SUMV = (D % 32) != 0;
    constexpr int LROW = D % 32, LREG = 4 * (LROW / 8) + (LROW & 3), LHALF = (LROW >> 2) & 1;
    constexpr int C8 = D / 8;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];      // [K rows | V rows]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    // block -> (sample, head, chunk of the sample's rows); XCD b & 7 takes whole samples (the sample's K / V and its neighbours' Q lines in
    // one L2) when they divide over the 8 XCDs, else (sample, head) pairs (two samples used to put the whole launch on two XCDs)
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    int smp, head, chunk;
    if (p.n % 8 == 0) {
        const int per = p.heads * chunks;
        const int ul = idx / per, w = idx - ul * per;
        smp = ul * 8 + xcd;
        head = w / chunks; chunk = w - head * chunks;
    } else {
        const int ul = idx / chunks;
        const int pair = ul * 8 + xcd;
        chunk = idx - ul * chunks; smp = pair / p.heads; head = pair - smp * p.heads;
    }
    if (smp >= p.n) return;
    const int rows = p.F * p.Nq;                                        // query rows of the sample
    const H* __restrict__ Q = reinterpret_cast<const H*>(p.q) + (size_t)smp * rows * p.ldq + head * D;
    const H* __restrict__ K = reinterpret_cast<const H*>(p.k) + (size_t)smp * p.Nk * p.ldkv + head * D;
    const H* __restrict__ V = reinterpret_cast<const H*>(p.v) + (size_t)smp * p.Nk * p.ldkv + head * D;
    H* __restrict__ O = reinterpret_cast<H*>(p.o) + (size_t)smp * rows * p.ldo + head * D;

    for (int i = tid * 16; i < KBYTES + NKEY * VROW; i += 256 * 16)      // pad rows / columns: zeros
        *reinterpret_cast<f32x4*>(smem_c + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int i = tid; i < p.Nk * C8; i += 256) {
        const int row = i / C8, c8 = i - row * C8;
        *reinterpret_cast<f32x4*>(smem_c + row * KROW + c8 * 16) = *reinterpret_cast<const f32x4*>(K + (size_t)row * p.ldkv + c8 * 8);
        *reinterpret_cast<f32x4*>(smem_c + KBYTES + row * VROW + c8 * 16) = *reinterpret_cast<const f32x4*>(V + (size_t)row * p.ldkv + c8 * 8);
    }
    if constexpr (SUMV) {
        if (tid < NKEY) *reinterpret_cast<H*>(smem_c + KBYTES + tid * VROW + D * 2) = (H)(tid < p.Nk ? 1.0f : 0.0f);
    }
    __syncthreads();

    const float qs = p.scale * 1.44269504088896340736f;
    auto load_q = [&](const int row0, hx8<H> (&qf)[KS]) {
        const int qrow = min(row0 + j, rows - 1);
        const H* qp = Q + (size_t)qrow * p.ldq;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k0 = 16 * s + 8 * h;
            hx8<H> a;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (H)0.f;
            if (k0 < D) a = *reinterpret_cast<const hx8<H>*>(qp + k0);
            qf[s] = a;
        }
    };
    const int ti = lane & 15;
    const int tr_off = (4 * h + (ti >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (ti & 3)) * 2;
    const char* const Kl = smem_c;
    const char* const Vl = smem_c + KBYTES;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const int tile0 = chunk * 4 * tpw + wave;                            // this wave's tiles: tile0, tile0 + 4, ...
    hx8<H> qn[KS];
    if (tile0 * 32 < rows) load_q(tile0 * 32, qn);
    for (int it = 0; it < tpw; ++it) {
        const int row0 = (tile0 + 4 * it) * 32;
        if (row0 >= rows) break;
        hx8<H> qf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = (H)((float)qn[s][e] * qs);
        }
        if (it + 1 < tpw && row0 + 128 < rows) load_q(row0 + 128, qn);      // the next tile's rows: in flight under this tile

        f32x16 st[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const char* kp = Kl + (kt * 32 + j) * KROW + h * 16;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const hx8<H> kf = *reinterpret_cast<const hx8<H>*>(kp + s * 32);
                st[kt] = mfma_32x32x16(kf, qf[s], s == 0 ? zero16 : st[kt]);
            }
        }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (key >= p.Nk) st[kt][r] = -INFINITY;
            }
        float mt = st[0][0];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = __builtin_fmaxf(mt, st[kt][r]);
        {
            float a = mt, bb = mt;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(bb));
            mt = __builtin_fmaxf(a, bb);
        }
        if (__any(__builtin_fabsf(mt) > 8.0f)) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kt][r] -= mt;
        }
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[kt][r] = __builtin_amdgcn_exp2f(st[kt][r]);
                if constexpr (!SUMV) ps += st[kt][r];
            }
        f32x16 acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                hx8<H> pf[2];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[s][e] = (H)st[kt][8 * s + e];
                const char* vb = Vl + kt * 32 * VROW + tr_off + t * 64;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const hx4<H> lo = lds_read_tr16<H>((vb + (16 * s) * VROW));
                    const hx4<H> hi = lds_read_tr16<H>((vb + (16 * s + 8) * VROW));
                    const hx8<H> vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[t] = mfma_32x32x16(vf, pf[s], (kt == 0 && s == 0) ? zero16 : acc[t]);
                }
            }
        }
        float l_tot;
        if constexpr (SUMV) l_tot = __shfl(acc[T - 1][LREG], j + 32 * LHALF);
        else l_tot = ps + __shfl_xor(ps, 32);
        {
            // O rows leave as 16-byte pieces: a lane holds 4 consecutive value columns per register quad (8 rg + 4 h), the other half-wave
            // the next 4; v_permlane32_swap on the packed pairs of two adjacent quads (rg = 2k, 2k + 1) leaves lane (j, h) with the 8
            // columns 16 k + 8 h .. + 7 -- per row and instruction 32 contiguous bytes, half as many store instructions as 8-byte pieces
            const float inv = 1.0f / l_tot;
            const bool row_ok = row0 + j < rows;
            H* op = O + (size_t)min(row0 + j, rows - 1) * p.ldo;
            typedef unsigned cu32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned cu32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (32 * t + 16 * k >= D) continue;               // (compile-time: the pair lies beyond the head)
                    hx4<H> oa, ob;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        oa[e] = (H)(acc[t][(2 * k) * 4 + e] * inv);
                        ob[e] = (H)(acc[t][(2 * k + 1) * 4 + e] * inv);
                    }
                    const cu32x2 a = __builtin_bit_cast(cu32x2, oa), bq = __builtin_bit_cast(cu32x2, ob);
                    unsigned a0 = a[0], a1 = a[1], b0 = bq[0], b1 = bq[1];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
                    const int dv = 32 * t + 16 * k + 8 * h;
                    if (row_ok && dv + 8 <= D) *reinterpret_cast<cu32x4*>(op + dv) = cu32x4{a0, a1, b0, b1};
                }
        }
    }
}

template <int D>
static bool launch_cross_resident(const AttnArgs& a, hipStream_t s) {
    static const int* const on = knob("E2V_ATTN_CROSS_RESIDENT", 1);     // 0: cross-attention through the staged kernel
    if (!*on || a.mode != 1 || a.Nk > 96 || a.Nk < 1) return false;
    constexpr int DP = (D + 15) / 16 * 16, T = (D + 31) / 32;
    constexpr int VROW = ((T * 16) % 32 == 16) ? T * 64 : T * 64 + 64;
    const size_t smem = (size_t)96 * (DP * 2 + 16) + (size_t)96 * VROW;
    const int rows = a.F * a.Nq;
    const int tiles = (rows + 31) / 32;
    // tiles per wave: enough workgroups to fill the chip a few times over (n * heads * chunks >= ~2048), at most 16 tiles per wave
    int tpw = 16;
    while (tpw > 1 && (long)a.n * a.heads * ((tiles + 4 * tpw - 1) / (4 * tpw)) < 2048) tpw >>= 1;
    const int chunks = (tiles + 4 * tpw - 1) / (4 * tpw);
    const unsigned grid = a.n % 8 == 0 ? 8u * (a.n / 8) * (unsigned)(a.heads * chunks) : 8u * ((a.n * a.heads + 7) / 8) * (unsigned)chunks;
    const double probs = (double)a.n * a.F * a.heads;
    std::string pname = a.io_bf16 == H16_FP16 ? "flash_attn_fp16_cross" : "flash_attn_bf16_cross";
    if (prof_detail()) pname += attn_shape_tag(a);
    ProfScope ps(pname.c_str(), 4.0 * probs * a.Nq * a.Nk * D, 2.0 * probs * D * (2.0 * a.Nq + 2.0 * (double)a.Nk / a.F), s);
    dry_tag(" -> cross_attn_resident_kernel tpw" + std::to_string(tpw));
    h16_dispatch(a.io_bf16, [&](auto h16_tag) {
        using H = decltype(h16_tag);
        E2V_KATTR((cross_attn_resident_kernel<H, D>), smem);
        E2V_KLAUNCH((cross_attn_resident_kernel<H, D>), dim3(grid), dim3(256), smem, s, a, tpw, chunks);
    });
    return true;
}

template <int D>
static void launch_flash_b16io(const AttnArgs& a, hipStream_t s) {
    constexpr int DP = (D + 15) / 16 * 16, T = (D + 31) / 32;
    constexpr int VROW = ((T * 16) % 32 == 16) ? T * 64 : T * 64 + 64;      // as in the kernel
    auto stage_bytes = [](const int kt) { return ((size_t)kt * (DP * 2 + 16) + (size_t)kt * VROW + 15) / 16 * 16; };
    if (launch_cross_resident<D>(a, s)) return;
    if (flash_attention_q64(a, s)) return;               // 64 queries per wave (attn_q64.hip): d = 40 / 80 self-attention
    static const int* const fold = E2V_AB_KNOB("E2V_ATTN_FOLD", 1);     // 0: the plain form (scale and maximum applied by vector FMAs)
    static const int* const kt64 = knob("E2V_ATTN_KT64", 1);     // 0: 32-key stages (one barrier per 32 keys)
    dim3 grid(attn_grid(a), 1, 1);
    const double nk = a.mode == 0 ? 2.0 * a.Nk : (double)a.Nk;
    const double probs = (double)a.n * a.F * a.heads;
    std::string pname = std::string(a.io_bf16 == H16_FP16 ? "flash_attn_fp16" : "flash_attn_bf16") + (a.mode == 0 ? "_sparse_causal" : "_cross");
    if (prof_detail()) pname += attn_shape_tag(a);
    ProfScope ps(pname.c_str(), 4.0 * probs * a.Nq * nk * D,
                 2.0 * probs * D * (2.0 * a.Nq + 2.0 * (a.mode == 0 ? a.Nk : (double)a.Nk / a.F)), s);
    auto go = [&](auto kern, const int kt) {
        dry_tag(std::string(" -> flash_attn_b16io_kernel") + (*fold ? " fold" : "") + " kt" + std::to_string(kt));
        const size_t smem = 2 * stage_bytes(kt);
        E2V_KATTR(kern, (2 * stage_bytes(64)));
        E2V_KLAUNCH(kern, grid, dim3(256), smem, s, a);
    };
    // 64-key stages halve the barriers per key, but at D = 160 two such stages are 84 KB and only one block fits a CU (measured: 224 vs
    // 344 TFLOP/s at the 12x12 level) -- keep them to the head sizes where three blocks still fit
    const bool wide = *kt64 && a.Nk > 32 && 2 * stage_bytes(64) <= 52 * 1024;
    h16_dispatch(a.io_bf16, [&](auto h16_tag) {
        using H = decltype(h16_tag);
        if (*fold) { if (wide) go(flash_attn_b16io_kernel<H, D, true, 64>, 64); else go(flash_attn_b16io_kernel<H, D, true, 32>, 32); }
#ifdef E2V_AB              // the plain form (scale and maximum by vector FMAs): the other arm of the A/B that adopted the fold
        else       { if (wide) go(flash_attn_b16io_kernel<H, D, false, 64>, 64); else go(flash_attn_b16io_kernel<H, D, false, 32>, 32); }
#endif
    });
}

// =====================================================================================================
// f32x3 attention (opt-in, AttnArgs::x3; see igemm.hip / DESIGN 3.7): both matmuls as six bf16-piece MFMAs over operands
// split EXACTLY into three bf16 pieces by truncation -- Q once per wave, K and V on their way into LDS (three planes each),
// the probabilities in registers.  Scores and outputs carry fp32-level error; what changes is the pipe: 42 bf16 MFMAs
// (1344 cycles) per key tile instead of 36 fp32 ones (2304), on a pipe that leaves the vector ALUs to the softmax.
// Same data flow as flash_attn_bf16_kernel: S^T = K Q^T with the query on the lane, P^T registers are the B operand.
// Measured and not adopted (level 0, 8.0-8.6 ms as it stands vs 9.7 fp32): splitting K / V once in a pre-pass instead of in
// every query block (kernel 8.8 ms + 0.3 ms pre-pass: the split is not the bottleneck), and a three-stage software pipeline
// issuing the next tile's score MFMAs under this tile's softmax with sched_group_barrier interleaving (8.9 ms).
// =====================================================================================================
__device__ __forceinline__ void split3(float x, unsigned& b1, unsigned& b2, unsigned& b3) {
    b1 = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
    const float r1 = x - __builtin_bit_cast(float, b1);
    b2 = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
    b3 = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, b2));
}
typedef unsigned au32x2 __attribute__((ext_vector_type(2)));
typedef unsigned au32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(256) void flash_attn_x3_kernel(const AttnArgs p) {
    constexpr int DP = (D + 15) / 16 * 16;
    constexpr int KS = DP / 16;
    constexpr int KROW = DP * 2 + 16;
    constexpr int T = (D + 31) / 32;
    constexpr int VROW = 72;
    constexpr int KBYTES = 32 * KROW, VBYTES = T * 32 * VROW;       // one plane
    constexpr int STAGE = (3 * (KBYTES + VBYTES) + 15) / 16 * 16;
    constexpr int DQ = D / 4;
    constexpr int NF4 = 32 * DQ;
    constexpr int LPT = (NF4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem_x[];   // [2][3 x K rows | 3 x V^T rows]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const AttnBlock blk = attn_block(p);
    if (!blk.valid) return;
    const int sf = blk.sf;
    const int smp = sf / p.F, f = sf - smp * p.F;
    const int head = blk.head;
    const int q0 = (blk.qb * 4 + wave) * 32;
    const bool active = q0 < p.Nq;

    int nseg = 1;
    size_t kvbase[2];
    if (p.mode == 0) {
        kvbase[0] = (size_t)(smp * p.F) * p.Nk;
        kvbase[1] = (size_t)(smp * p.F + (f > 0 ? f - 1 : 0)) * p.Nk;
        nseg = f >= 2 ? 2 : 1;
    } else {
        kvbase[0] = kvbase[1] = (size_t)smp * p.Nk;
    }
    const int tps = (p.Nk + 31) / 32;
    const int ntiles = nseg * tps;

    for (int i = tid * 16; i < 2 * STAGE; i += 256 * 16)            // pad columns / rows are never rewritten
        *reinterpret_cast<f32x4*>(smem_x + i) = f32x4{0.f, 0.f, 0.f, 0.f};

    au32x4 qf[3][KS];                                               // three planes of the lane's 8 k of every 16-step
    {
        const int qrow = min(q0 + j, p.Nq - 1);
        const float* qp = p.q + ((size_t)sf * p.Nq + qrow) * p.ldq + head * D;
        const float c = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k0 = 16 * s + 8 * h;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            if (k0 < D) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(qp + k0) * c;
                const f32x4 b = *reinterpret_cast<const f32x4*>(qp + k0 + 4) * c;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned a1, a2, a3, b1, b2, b3;
                split3(v[2 * e], a1, a2, a3);
                split3(v[2 * e + 1], b1, b2, b3);
                qf[0][s][e] = (a1 >> 16) | (b1 & 0xFFFF0000u);
                qf[1][s][e] = (a2 >> 16) | (b2 & 0xFFFF0000u);
                qf[2][s][e] = (a3 >> 16) | (b3 & 0xFFFF0000u);
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
    int ld_row[LPT], ld_c4[LPT];
    unsigned ld_off[LPT];
#pragma unroll
    for (int e = 0; e < LPT; ++e) {
        const int idx = tid + 256 * e;
        const int row = idx / DQ, c4 = idx - row * DQ;
        ld_row[e] = row; ld_c4[e] = c4;
        ld_off[e] = idx < NF4 ? (unsigned)(row * p.ldkv + c4 * 4) * 4u : OOB;
    }
    f32x4 kreg[LPT], vreg[LPT];
    auto load_tile = [&](int tt) {
        const int seg = tt / tps;
        const int key0 = (tt - seg * tps) * 32;
        const size_t first = (kvbase[seg] + key0) * p.ldkv + head * D;
        const __amdgpu_buffer_rsrc_t rk = rsrc_of(p.k + first), rv = rsrc_of(p.v + first);
        const int left = p.Nk - key0;
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            unsigned off = ld_off[e];
            if (left < 32) off = ld_row[e] < left ? off : OOB;
            kreg[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0));
            vreg[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0));
        }
    };
    auto store_tile = [&](int buf) {
        char* Kl = smem_x + buf * STAGE;
        char* Vl = Kl + 3 * KBYTES;
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            if (tid + 256 * e < NF4) {
                const int row = ld_row[e], c4 = ld_c4[e];
                unsigned k1[4], k2[4], k3[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) split3(kreg[e][x], k1[x], k2[x], k3[x]);
                char* kd = Kl + row * KROW + c4 * 8;
                *reinterpret_cast<au32x2*>(kd) = au32x2{(k1[0] >> 16) | (k1[1] & 0xFFFF0000u), (k1[2] >> 16) | (k1[3] & 0xFFFF0000u)};
                *reinterpret_cast<au32x2*>(kd + KBYTES) = au32x2{(k2[0] >> 16) | (k2[1] & 0xFFFF0000u), (k2[2] >> 16) | (k2[3] & 0xFFFF0000u)};
                *reinterpret_cast<au32x2*>(kd + 2 * KBYTES) = au32x2{(k3[0] >> 16) | (k3[1] & 0xFFFF0000u), (k3[2] >> 16) | (k3[3] & 0xFFFF0000u)};
#pragma unroll
                for (int x = 0; x < 4; ++x) {                           // transpose: [value column][key]
                    unsigned v1, v2, v3;
                    split3(vreg[e][x], v1, v2, v3);
                    char* vd = Vl + (c4 * 4 + x) * VROW + row * 2;
                    *reinterpret_cast<unsigned short*>(vd) = (unsigned short)(v1 >> 16);
                    *reinterpret_cast<unsigned short*>(vd + VBYTES) = (unsigned short)(v2 >> 16);
                    *reinterpret_cast<unsigned short*>(vd + 2 * VBYTES) = (unsigned short)(v3 >> 16);
                }
            }
        }
    };

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    // (first piece index, second piece index) of the six products, smallest first
    constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
    for (int tt = 0; tt < ntiles; ++tt) {
        const int buf = tt & 1;
        if (tt + 1 < ntiles) load_tile(tt + 1);
        const int seg_c = tt / tps;
        const int key0 = (tt - seg_c * tps) * 32;
        if (active) {
            const char* Kl = smem_x + buf * STAGE;
            const char* Vl = Kl + 3 * KBYTES;
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            const char* kp = Kl + j * KROW + h * 16;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                abf16x8 kf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) kf[pl] = *reinterpret_cast<const abf16x8*>(kp + pl * KBYTES + s * 32);
#pragma unroll
                for (int t6 = 0; t6 < 6; ++t6)
                    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[TA[t6]], __builtin_bit_cast(abf16x8, qf[TB[t6]][s]), st, 0, 0, 0);
            }
            if (key0 + 32 > p.Nk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (key >= p.Nk) st[r] = -INFINITY;
                }
            }
            float mt = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = __builtin_fmaxf(mt, st[r]);
            mt = __builtin_fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = __builtin_fmaxf(m_i, mt);
            const bool moved = __any(m_new > m_i);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(st[r] - m_new);
                ps += st[r];
            }
            float alpha = 1.0f;
            if (moved) {
                alpha = __builtin_amdgcn_exp2f(m_i - m_new);
                l_i *= alpha;
            }
            l_i += ps;
            m_i = m_new;
            au32x4 pf[3][2];                     // P^T pieces: registers 8s..8s+7 are k-step s as they stand
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned a1, a2, a3, b1, b2, b3;
                    split3(st[8 * s + 2 * e], a1, a2, a3);
                    split3(st[8 * s + 2 * e + 1], b1, b2, b3);
                    pf[0][s][e] = (a1 >> 16) | (b1 & 0xFFFF0000u);
                    pf[1][s][e] = (a2 >> 16) | (b2 & 0xFFFF0000u);
                    pf[2][s][e] = (a3 >> 16) | (b3 & 0xFFFF0000u);
                }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (moved) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] *= alpha;
                }
                const char* vrow = Vl + min(t * 32 + j, D - 1) * VROW + h * 8;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    abf16x8 vf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const abf16x4 lo = *reinterpret_cast<const abf16x4*>(vrow + pl * VBYTES + s * 32);
                        const abf16x4 hi = *reinterpret_cast<const abf16x4*>(vrow + pl * VBYTES + s * 32 + 16);
                        vf[pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                    for (int t6 = 0; t6 < 6; ++t6)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[TA[t6]], __builtin_bit_cast(abf16x8, pf[TB[t6]][s]), acc[t], 0, 0, 0);
                }
            }
        }
        if (tt + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    const float l_tot = l_i + __shfl_xor(l_i, 32);
    if (active && q0 + j < p.Nq) {
        const float inv = 1.0f / l_tot;
        float* op = p.o + ((size_t)sf * p.Nq + q0 + j) * p.ldo + head * D;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int dv = t * 32 + 8 * rg + 4 * h;
                if (dv < D) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[t][rg * 4 + e] * inv;
                    *reinterpret_cast<f32x4*>(op + dv) = o;
                }
            }
    }
}
template <int D>
static void launch_flash_x3(const AttnArgs& a, hipStream_t s) {
    constexpr int DP = (D + 15) / 16 * 16;
    constexpr size_t stage = ((size_t)3 * (32 * (DP * 2 + 16) + ((D + 31) / 32) * 32 * 72) + 15) / 16 * 16;
    constexpr size_t smem = 2 * stage;
    E2V_KATTR(&flash_attn_x3_kernel<D>, smem);
    dim3 grid(attn_grid(a), 1, 1);
    const double nk = a.mode == 0 ? 2.0 * a.Nk : (double)a.Nk;
    const double probs = (double)a.n * a.F * a.heads;
    ProfScope ps(a.mode == 0 ? "flash_attn_f32x3_sparse_causal" : "flash_attn_f32x3_cross", 4.0 * probs * a.Nq * nk * D,
                 4.0 * probs * D * (2.0 * a.Nq + 2.0 * (a.mode == 0 ? a.Nk : (double)a.Nk / a.F)), s);
    E2V_KLAUNCH((flash_attn_x3_kernel<D>), grid, dim3(256), smem, s, a);
}

template <int D>
static void launch_flash(const AttnArgs& a, hipStream_t s) {
    constexpr size_t smem = (size_t)2 * 2 * 32 * (D + 4) * sizeof(float);
    E2V_KATTR(&flash_attn_kernel<D>, smem);
    dim3 grid(attn_grid(a), 1, 1);
    const double nk = a.mode == 0 ? 2.0 * a.Nk : (double)a.Nk;       // the reference attends to 2N concatenated keys
    const double probs = (double)a.n * a.F * a.heads;
    std::string pname = a.mode == 0 ? "flash_attn_sparse_causal" : "flash_attn_cross";
    if (prof_detail()) pname += attn_shape_tag(a);
    ProfScope ps(pname.c_str(), 4.0 * probs * a.Nq * nk * D,
                 4.0 * probs * D * (2.0 * a.Nq + 2.0 * (a.mode == 0 ? a.Nk : (double)a.Nk / a.F)), s);
    dry_tag(" -> flash_attn_kernel");
    E2V_KLAUNCH((flash_attn_kernel<D>), grid, dim3(256), smem, s, a);
}

bool flash_attention_supports(int D) {
    switch (D) {
        case 8: case 16: case 32: case 40: case 64: case 80: case 160: return true;
        default: return false;
    }
}

void flash_attention(const AttnArgs& a, hipStream_t s) {
    // every caller (graph walker and op wrapper alike) ends up here: an unsupported head dim must never fall through to
    // "nothing launched, output uninitialised"
    if (!flash_attention_supports(a.D))
        throw Error(E2V_EINVAL, "attention head dim " + std::to_string(a.D) + " has no kernel instance (supported: 8, 16, 32, 40, 64, 80, 160)");
    if (a.io_bf16) {
        if ((a.ldq | a.ldkv | a.ldo) & 7) throw Error(E2V_ESHAPE, "bf16 attention: row strides must be multiples of 8 elements");
        switch (a.D) {
            case 8: launch_flash_b16io<8>(a, s); return;
            case 16: launch_flash_b16io<16>(a, s); return;
            case 32: launch_flash_b16io<32>(a, s); return;
            case 40: launch_flash_b16io<40>(a, s); return;
            case 64: launch_flash_b16io<64>(a, s); return;
            case 80: launch_flash_b16io<80>(a, s); return;
            default: launch_flash_b16io<160>(a, s); return;
        }
    }
    if (a.x3) {
        switch (a.D) {
            case 8: launch_flash_x3<8>(a, s); return;
            case 16: launch_flash_x3<16>(a, s); return;
            case 32: launch_flash_x3<32>(a, s); return;
            case 40: launch_flash_x3<40>(a, s); return;
            case 64: launch_flash_x3<64>(a, s); return;
            case 80: launch_flash_x3<80>(a, s); return;
            default: launch_flash_x3<160>(a, s); return;
        }
    }
    switch (a.D) {
        case 8: launch_flash<8>(a, s); break;
        case 16: launch_flash<16>(a, s); break;
        case 32: launch_flash<32>(a, s); break;
        case 40: launch_flash<40>(a, s); break;
        case 64: launch_flash<64>(a, s); break;
        case 80: launch_flash<80>(a, s); break;
        default: launch_flash<160>(a, s); break;
    }
}

// ---- temporal attention: F x F per (pixel, head); tokens stay where they are --------------------------
// The reference transposes (b f) d c -> (b d) f c and back (attention.py:262,267); with channel-last
// rows (sample, frame, pixel) the F rows of one pixel are HW rows apart and are read in place.
static constexpr int FMAX = 8;
template <typename T>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const T* __restrict__ qkv, int ld, T* __restrict__ out,
                                                            int ldo, int n, int F, int HW, int heads, int D, float scale) {
    const size_t total = (size_t)n * HW * heads * F;
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int i = gid % F;
    const int head = (gid / F) % heads;
    const size_t ph = gid / ((size_t)F * heads);
    const int pix = ph % HW;
    const int smp = ph / HW;
    const int C = heads * D;
    const size_t row0 = (size_t)smp * F * HW + pix;          // frame 0 row; frame f is f*HW rows further
    const T* qp = qkv + (row0 + (size_t)i * HW) * ld + head * D;
    const T* kp = qkv + row0 * ld + C + head * D;
    const T* vp = qkv + row0 * ld + 2 * C + head * D;
    const size_t fstride = (size_t)HW * ld;
    float s[FMAX];
#pragma unroll
    for (int jf = 0; jf < FMAX; ++jf) s[jf] = 0.f;
    for (int c = 0; c < D; c += 4) {
        const f32x4 qv = ld4(qp + c);
#pragma unroll
        for (int jf = 0; jf < FMAX; ++jf)
            if (jf < F) {
                const f32x4 kv = ld4(kp + jf * fstride + c);
                s[jf] += qv[0] * kv[0] + qv[1] * kv[1] + qv[2] * kv[2] + qv[3] * kv[3];
            }
    }
    float m = -INFINITY;
#pragma unroll
    for (int jf = 0; jf < FMAX; ++jf)
        if (jf < F) {
            s[jf] *= scale;
            m = fmaxf(m, s[jf]);
        }
    float l = 0.f;
#pragma unroll
    for (int jf = 0; jf < FMAX; ++jf)
        if (jf < F) {
            s[jf] = expf(s[jf] - m);
            l += s[jf];
        }
    const float inv = 1.0f / l;
    T* op = out + (row0 + (size_t)i * HW) * ldo + head * D;
    for (int c = 0; c < D; c += 4) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jf = 0; jf < FMAX; ++jf)
            if (jf < F) {
                const f32x4 vv = ld4(vp + jf * fstride + c);
                o += vv * (s[jf] * inv);
            }
        st4(op + c, o);
    }
}

// LDS-staged version: a block stages q, k, v of PB pixels x F frames for a slab of CS channels (whole heads) with fully
// coalesced row reads, then one thread per (pixel, head, query frame) works out of LDS.  HBM sees every byte once.
template <typename T>
__global__ __launch_bounds__(128) void temporal_attn_lds_kernel(const T* __restrict__ qkv, int ld, T* __restrict__ out,
                                                                int ldo, int F, int HW, int C, int D, float scale, int PB,
                                                                int CS, int npg) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // [F][PB][q | k | v][CS]
    const int smp = blockIdx.x / npg, p0 = (blockIdx.x - smp * npg) * PB;
    const int c_base = blockIdx.y * CS;
    if constexpr (!std::is_same<T, float>::value) {       // bf16 rows: 16-byte (8-channel) pieces, widened on their way into LDS
        const int CO = CS / 8;
        const int total = F * PB * 3 * CO;
        for (int idx = threadIdx.x; idx < total; idx += 128) {
            const int c8 = idx % CO;
            int r = idx / CO;
            const int part = r % 3; r /= 3;
            const int pp = r % PB;
            const int f = r / PB;
            const int pix = p0 + pp;
            f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
            if (pix < HW) {
                const hx8<T> v = *reinterpret_cast<const hx8<T>*>(qkv + ((size_t)(smp * F + f) * HW + pix) * ld + part * C + c_base + c8 * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = (float)v[e]; hi[e] = (float)v[4 + e]; }
            }
            *reinterpret_cast<f32x4*>(sm + (size_t)idx * 8) = lo;
            *reinterpret_cast<f32x4*>(sm + (size_t)idx * 8 + 4) = hi;
        }
    } else {
        const int CQ = CS / 4;
        const int total = F * PB * 3 * CQ;
        for (int idx = threadIdx.x; idx < total; idx += 128) {
            const int c4 = idx % CQ;
            int r = idx / CQ;
            const int part = r % 3; r /= 3;
            const int pp = r % PB;
            const int f = r / PB;
            const int pix = p0 + pp;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pix < HW) v = ld4(qkv + ((size_t)(smp * F + f) * HW + pix) * ld + part * C + c_base + c4 * 4);
            *reinterpret_cast<f32x4*>(sm + (size_t)idx * 4) = v;
        }
    }
    __syncthreads();
    const int hs = CS / D;
    const int items = PB * hs * F;
    for (int item = threadIdx.x; item < items; item += 128) {
        const int i = item % F;
        const int hh = (item / F) % hs;
        const int pp = item / (F * hs);
        const int pix = p0 + pp;
        if (pix >= HW) continue;
        const size_t fs = (size_t)PB * 3 * CS;                          // frame stride in LDS
        const float* q = sm + ((size_t)(i * PB + pp) * 3) * CS + hh * D;
        const float* k = sm + ((size_t)pp * 3 + 1) * CS + hh * D;
        const float* v = sm + ((size_t)pp * 3 + 2) * CS + hh * D;
        float s[FMAX];
#pragma unroll
        for (int jf = 0; jf < FMAX; ++jf) s[jf] = 0.f;
        for (int c = 0; c < D; c += 4) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(q + c);
#pragma unroll
            for (int jf = 0; jf < FMAX; ++jf)
                if (jf < F) {
                    const f32x4 kv = *reinterpret_cast<const f32x4*>(k + jf * fs + c);
                    s[jf] += qv[0] * kv[0] + qv[1] * kv[1] + qv[2] * kv[2] + qv[3] * kv[3];
                }
        }
        float m = -INFINITY;
#pragma unroll
        for (int jf = 0; jf < FMAX; ++jf)
            if (jf < F) {
                s[jf] *= scale;
                m = fmaxf(m, s[jf]);
            }
        float l = 0.f;
#pragma unroll
        for (int jf = 0; jf < FMAX; ++jf)
            if (jf < F) {
                s[jf] = expf(s[jf] - m);
                l += s[jf];
            }
        const float inv = 1.0f / l;
        T* op = out + ((size_t)(smp * F + i) * HW + pix) * ldo + c_base + hh * D;
        if constexpr (!std::is_same<T, float>::value) {
            for (int c = 0; c < D; c += 8) {                    // D % 8 == 0: one 16-byte store per 8 channels
                f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
                for (int jf = 0; jf < FMAX; ++jf)
                    if (jf < F) {
                        const float pw = s[jf] * inv;
                        o0 += *reinterpret_cast<const f32x4*>(v + jf * fs + c) * pw;
                        o1 += *reinterpret_cast<const f32x4*>(v + jf * fs + c + 4) * pw;
                    }
                hx8<T> ob;
#pragma unroll
                for (int e = 0; e < 4; ++e) { ob[e] = (T)o0[e]; ob[4 + e] = (T)o1[e]; }
                *reinterpret_cast<hx8<T>*>(op + c) = ob;
            }
        } else {
            for (int c = 0; c < D; c += 4) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jf = 0; jf < FMAX; ++jf)
                    if (jf < F) {
                        const f32x4 vv = *reinterpret_cast<const f32x4*>(v + jf * fs + c);
                        o += vv * (s[jf] * inv);
                    }
                st4(op + c, o);
            }
        }
    }
}

// One pixel per heads x LH lanes, no LDS: lane (head, c) owns the 8 channels c of its head in every frame -- 16-byte loads (bf16; two for
// fp32 rows) that cover a pixel's row contiguously, every byte of q, k, v read once and all 3 F rows of a wave in flight together --,
// the F x F scores are partial dot products over those 8 channels all-reduced over the head's LH
// lanes with DPP adds (quad_perm, row_half_mirror, row_mirror: no LDS, no ds_bpermute below 32 lanes), every lane then runs the
// softmax of its head and the P V product of its own 8 channels.  LH = lanes per head = D / 8 rounded up to a power of two (D = 40:
// 5 of 8 lanes carry data, the loads of the other 3 are masked); the staged kernel below spent its time alternating a load phase
// and a compute phase in which 24 of a block's 128 threads worked (1.6 TB/s at level 0).
template <int LH>
__device__ __forceinline__ float head_allreduce(float x) {
    auto dpp = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});                       // quad_perm [1,0,3,2]
    x += dpp(x, std::integral_constant<int, 0x4E>{});                       // quad_perm [2,3,0,1]
    if constexpr (LH >= 8) x += dpp(x, std::integral_constant<int, 0x141>{});   // row_half_mirror: the other quad of the 8
    if constexpr (LH >= 16) x += dpp(x, std::integral_constant<int, 0x140>{});  // row_mirror: the other 8 of the 16
    if constexpr (LH >= 32) x += __shfl_xor(x, 16);
    return x;
}

template <typename T, int F, int D>
__global__ __launch_bounds__(256) void temporal_attn_wave_kernel(const T* __restrict__ qkv, int ld, T* __restrict__ out, int ldo, int HW,
                                                                 long npix, int heads, float scale) {
    constexpr int CH = D / 8;
    constexpr int LH = CH <= 4 ? 4 : CH <= 8 ? 8 : CH <= 16 ? 16 : 32;
    constexpr bool B16 = !std::is_same<T, float>::value;
    const int lpp = heads * LH;                                  // lanes per pixel (divides 256: launcher)
    const int t = threadIdx.x;
    const int lp = t % lpp;
    const int head = lp / LH, c = lp % LH;
    const long g = (long)blockIdx.x * (256 / lpp) + t / lpp;
    const bool on = c < CH && g < npix;
    const long gg = g < npix ? g : npix - 1;
    const long smp = gg / HW;
    const int pix = (int)(gg - smp * HW);
    const int C = heads * D;
    const T* base = qkv + ((size_t)(smp * F) * HW + pix) * ld + head * D + (c < CH ? c : 0) * 8;
    const size_t fs = (size_t)HW * ld;
    typedef float f8 __attribute__((ext_vector_type(8)));
    
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    auto ld16 = [&](const T* p) -> u4 { return on ? *reinterpret_cast<const u4*>(p) : u4{0u, 0u, 0u, 0u}; };
    auto ld8f = [&](const T* p) -> f8 {
        f8 r = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (on) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + 4);
            r = f8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        }
        return r;
    };
    float s[F][F];
    if constexpr (B16) {
        u4 q[F], k[F];
#pragma unroll
        for (int f = 0; f < F; ++f) { q[f] = ld16(base + f * fs); k[f] = ld16(base + f * fs + C); }
        // (bf16 -> fp32 is a shift: the element is the high half of the float.  v_dot2c_f32_bf16 would take the packed pairs as they
        // are, but its result feeding a DPP add came out wrong on gfx950 / ROCm 7.2 -- tools/micro/dot2_check.hip shows the
        // instruction itself is right -- and the kernel is bound by its loads either way)
        auto lo = [](unsigned u) { return h16_unpack_lo<T>(u); };
        auto hi = [](unsigned u) { return h16_unpack_hi<T>(u); };
#pragma unroll
        for (int i = 0; i < F; ++i)
#pragma unroll
            for (int j = 0; j < F; ++j) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a = __builtin_fmaf(lo(q[i][e]), lo(k[j][e]), a);
                    a = __builtin_fmaf(hi(q[i][e]), hi(k[j][e]), a);
                }
                s[i][j] = a;
            }
    } else {
        f8 q[F], k[F];
#pragma unroll
        for (int f = 0; f < F; ++f) { q[f] = ld8f(base + f * fs); k[f] = ld8f(base + f * fs + C); }
#pragma unroll
        for (int i = 0; i < F; ++i)
#pragma unroll
            for (int j = 0; j < F; ++j) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) a = __builtin_fmaf(q[i][e], k[j][e], a);
                s[i][j] = a;
            }
    }
    // the value rows: in flight under the reduction and the softmax
    u4 v16[B16 ? F : 1];
    f8 v32[B16 ? 1 : F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        if constexpr (B16) v16[f] = ld16(base + f * fs + 2 * C); else v32[f] = ld8f(base + f * fs + 2 * C);
    }
#pragma unroll
    for (int i = 0; i < F; ++i) {
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < F; ++j) {
            s[i][j] = head_allreduce<LH>(s[i][j]) * scale;
            m = fmaxf(m, s[i][j]);
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < F; ++j) {
            s[i][j] = __expf(s[i][j] - m);
            l += s[i][j];
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int j = 0; j < F; ++j) s[i][j] *= inv;
    }
    T* ob = out + ((size_t)(smp * F) * HW + pix) * ldo + head * D + (c < CH ? c : 0) * 8;
    const size_t ofs = (size_t)HW * ldo;
#pragma unroll
    for (int i = 0; i < F; ++i) {
        f8 o = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < F; ++j) {
            if constexpr (B16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {                    // bf16 -> fp32: the element IS the high half of the float
                    o[2 * e] = __builtin_fmaf(s[i][j], h16_unpack_lo<T>(v16[j][e]), o[2 * e]);
                    o[2 * e + 1] = __builtin_fmaf(s[i][j], h16_unpack_hi<T>(v16[j][e]), o[2 * e + 1]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(s[i][j], v32[j][e], o[e]);
            }
        }
        if (on) {
            if constexpr (B16) {
                hx8<T> ob8;
#pragma unroll
                for (int e = 0; e < 8; ++e) ob8[e] = (T)o[e];
                *reinterpret_cast<hx8<T>*>(ob + i * ofs) = ob8;
            } else {
                *reinterpret_cast<f32x4*>(ob + i * ofs) = f32x4{o[0], o[1], o[2], o[3]};
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ob + i * ofs) + 4) = f32x4{o[4], o[5], o[6], o[7]};
            }
        }
    }
}

template <typename T, int D>
static bool temporal_wave_launch(const T* qkv, int ld, T* out, int ldo, int n, int F, int HW, int heads, float scale, hipStream_t s) {
    constexpr int CH = D / 8, LH = CH <= 4 ? 4 : CH <= 8 ? 8 : CH <= 16 ? 16 : 32;
    const int lpp = heads * LH;
    if (F != 6 || lpp > 256 || 256 % lpp || (ld % 8) || (ldo % 8)) return false;
    const long npix = (long)n * HW;
    const int ppb = 256 / lpp;
    E2V_KLAUNCH((temporal_attn_wave_kernel<T, 6, D>), dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), 0, s, qkv, ld, out, ldo, HW, npix, heads,
                       scale);
    return true;
}

template <typename T>
static void temporal_attention_launch(const T* qkv, int ld, T* out, int ldo, int n, int F, int HW, int heads, int D, float scale,
                                      hipStream_t s);

void temporal_attention(const float* qkv, int ld, float* out, int ldo, int n, int F, int HW, int heads, int D, float scale,
                        hipStream_t s, int bf16) {
    const size_t total = (size_t)n * HW * heads * F;
    if (!total) return;
    std::string pname = "temporal_attn";
    if (prof_detail()) pname += " n" + std::to_string(n) + " F" + std::to_string(F) + " HW" + std::to_string(HW) + " h" + std::to_string(heads) + " D" + std::to_string(D);
    ProfScope ps(pname.c_str(), 4.0 * total * F * D, 4.0 * (bf16 ? 2.0 : 4.0) * (double)n * F * HW * heads * D, s);
    if (bf16)
        h16_dispatch(bf16, [&](auto h16_tag) {
            using H = decltype(h16_tag);
            temporal_attention_launch(reinterpret_cast<const H*>(qkv), ld, reinterpret_cast<H*>(out), ldo, n, F, HW, heads, D, scale, s);
        });
    else
        temporal_attention_launch(qkv, ld, out, ldo, n, F, HW, heads, D, scale, s);
}

template <typename T>
static void temporal_attention_launch(const T* qkv, int ld, T* out, int ldo, int n, int F, int HW, int heads, int D, float scale,
                                      hipStream_t s) {
    const size_t total = (size_t)n * HW * heads * F;
    const int C = heads * D;
    static const int* const wave_on = knob("E2V_TATTN_WAVE", 1);      // 0: the LDS-staged kernel (same-process A/B, the op test)
    if (*wave_on) {
        bool done = false;
        switch (D) {
            case 8: done = temporal_wave_launch<T, 8>(qkv, ld, out, ldo, n, F, HW, heads, scale, s); break;
            case 16: done = temporal_wave_launch<T, 16>(qkv, ld, out, ldo, n, F, HW, heads, scale, s); break;
            case 32: done = temporal_wave_launch<T, 32>(qkv, ld, out, ldo, n, F, HW, heads, scale, s); break;
            case 40: done = temporal_wave_launch<T, 40>(qkv, ld, out, ldo, n, F, HW, heads, scale, s); break;
            case 64: done = temporal_wave_launch<T, 64>(qkv, ld, out, ldo, n, F, HW, heads, scale, s); break;
            case 80: done = temporal_wave_launch<T, 80>(qkv, ld, out, ldo, n, F, HW, heads, scale, s); break;
            case 160: done = temporal_wave_launch<T, 160>(qkv, ld, out, ldo, n, F, HW, heads, scale, s); break;
            default: break;
        }
        if (done) { dry_tag(" -> temporal_attn_wave_kernel"); return; }
    }
    dry_tag(" -> temporal_attn_lds_kernel");
    // slab of whole heads and pixel count such that the staged q/k/v (fp32 in LDS whatever the storage type) fit 16 KB: with
    // 48 KB (3 blocks of 2 waves per CU) the load phase had too little in flight -- 0.60 -> 0.33 ms at level 0
    static const size_t budget = [] { const char* e = std::getenv("E2V_TATTN_LDS_KB"); return (size_t)(e ? std::atoi(e) : 16) * 1024; }();
    int hs = heads;
    while (hs > 1 && (size_t)F * 3 * hs * D * 4 > budget) hs = (hs + 1) / 2;
    while (heads % hs) --hs;
    const int CS = hs * D;
    int PB = (int)(budget / ((size_t)F * 3 * CS * 4));
    PB = PB < 1 ? 1 : (PB > 8 ? 8 : PB);
    const size_t smem = (size_t)F * PB * 3 * CS * 4;
    if (smem > 64 * 1024) {           // does not fit the default LDS window: per-thread global version
        E2V_KLAUNCH(temporal_attn_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, qkv, ld, out, ldo, n, F, HW,
                           heads, D, scale);
        return;
    }
    const int npg = (HW + PB - 1) / PB;
    E2V_KLAUNCH(temporal_attn_lds_kernel<T>, dim3((unsigned)(n * npg), C / CS), dim3(128), smem, s, qkv, ld, out, ldo, F, HW, C,
                       D, scale, PB, CS, npg);
}

// ---- row softmax in place (single-head VAE attention, 2304 keys, fp32 as the dep computes it) ----------
template <typename H>
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int ld, int rows, int cols, H* __restrict__ out16) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    float* xr = x + (size_t)row * ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, xr[c]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float l = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float e = expf(xr[c] - m);
        xr[c] = e;
        l += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off);
    if (lane == 0) red[wave] = l;
    __syncthreads();
    l = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / l;
    if (out16) {                      // bf16-activation mode: the probabilities feed a bf16 GEMM
        H* orow = out16 + (size_t)row * ld;
        for (int c = threadIdx.x; c < cols; c += 256) orow[c] = (H)(xr[c] * inv);
    } else {
        for (int c = threadIdx.x; c < cols; c += 256) xr[c] *= inv;
    }
}

void softmax_rows(float* x, int ld, int rows, int cols, hipStream_t s, void* out_bf16, int out_mode) {
    if (rows <= 0) return;
    ProfScope ps("softmax_rows", 8.0 * rows * cols, (out_bf16 ? 10.0 : 12.0) * rows * (double)cols, s);
    h16_dispatch(out_mode, [&](auto h16_tag) {
        using H = decltype(h16_tag);
        E2V_KLAUNCH(softmax_rows_kernel<H>, dim3(rows), dim3(256), 0, s, x, ld, rows, cols, static_cast<H*>(out_bf16));
    });
}

}  // namespace e2v
