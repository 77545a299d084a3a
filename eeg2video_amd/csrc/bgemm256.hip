// bf16 implicit-GEMM convolution / linear on 256-row x 256- or 320-column workgroup tiles: the deep-pipelined form of bgemm.hip.
//
// Same contract as bgemm.hip (out[m][n] = epilogue(sum_{tap,c} A[src(m,tap)][c] W[n][k(tap,c)]), bf16 rows in HBM, LDS-DMA staging with
// the gather in the per-lane SOURCE address, XOR-swizzled 128-byte LDS rows, fp32 accumulation) -- what changes is the tile and the
// schedule.  bgemm.hip's 64 x 64 wave tiles read 16 LDS fragments per 16 MFMAs, move 48 KB into LDS per 256 x 128 x 64 tile step and
// march eight waves in lockstep through one barrier per step (PMC: matrix pipe 0.41 busy over a pass, waves 37 % in waits, 2.35x
// the algorithmic bytes through L2).  Here:
//   * workgroup tile 256 x (64 NT) with NT = 4 or 5: 8 waves as 2 (M) x 4 (N), wave tile 128 rows x 16 NT columns (64 or 80), i.e.
//     8 x NT tiles of v_mfma_f32_16x16x32_bf16 = 128 / 160 accumulator registers.  N = 320, 640, 960, 1280, 1920, 2560 ... of the UNet
//     are all multiples of 320, so the 256 x 320 tile wastes no column; GEGLU (value / gate interleaved per 64 columns) and the
//     VAE's 256 / 512 channels take 256 x 256.  Per 64-deep K step a wave reads 16 + 2 NT fragments for 64 / 80 MFMAs and the
//     workgroup moves 64 / 72 KB into LDS: 0.67x / 0.6x the L2 -> LDS bytes per flop of the 256 x 128 tile, a third of its LDS reads;
//   * two K-step buffers (all of the tile's LDS: 128 / 144 KB), each restaged PIECEWISE while it is still being consumed: a K step
//     is four phases -- quadrant (rows 0-63 | 64-127 of the wave) x (first | second group of column tiles), K = 64 each -- and the
//     half of X / of W a phase has read for the last time is refilled, for the K step after next, one phase later (units q0, n1,
//     q1, n0 behind phases 0, 1, 2, 3).  Three such units (6 LDS-DMA instructions per wave) stay in flight behind the counted
//     `s_waitcnt vmcnt(6)` that ends a K step; vmcnt is never 0 in the loop.  (The first column group's fragments are read again in
//     phase 3 instead of being kept over phases 1-2: 160 accumulators + 56 fragment registers is what fits 256 VGPRs at NT = 5);
//   * the two wave rows (waves 0-3 | 4-7: one of each per SIMD) run one barrier apart: between two barriers one of them issues its
//     LDS reads and its DMA while the other one owns the matrix pipe with 16 / 24 back-to-back MFMAs (`s_setprio` around the
//     cluster keeps the compiler from spreading it over the barriers).
// RAW: a unit's DMA is retired by its wave's counted vmcnt in front of a barrier that every reader passes before the phase
// that reads it.  WAR: every phase's fragment reads are retired (`lgkmcnt(0)`) in front of the barrier that precedes the
// earliest refill of what they read.  (cdna_hip_programming.md, "The 256^2 8-phase template"; the schedule here is this file's.)
//
// The arithmetic is v_mfma_f32_16x16x32_bf16 (bgemm.hip: 32x32x16).  Both walk k in the same order with one fp32 accumulator per
// output, and they agree BIT FOR BIT on every shape of tests/test_hip_ops.py::test_bf16_t256_* and on the whole UNet + VAE decode
// (tests/test_hip_full.py::test_bf16_persistent_gemm_bit_identical_to_the_tile_kernels), so which kernel a layer takes may depend on
// M (few-tile launches stay on the 128-row grid) without a clip's result depending on its batch.
#include "igemm_epi.h"
#include "prof.h"
#include "runtime.h"

#include <string>

namespace e2v {

typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned T256_OOB = 0x80000000u;      // beyond every descriptor window: the buffer unit returns zeros / drops the store

__device__ __forceinline__ __amdgpu_buffer_rsrc_t t256_rsrc(const void* ptr, const int records) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), (short)0, records, 0x00020000);
}

template <typename H, int NT, bool LIN, bool RB = false>      // RB: the launch leaves row-block sums for a following GroupNorm (IgemmArgs::rbsum)
__device__ __forceinline__ void t256_body(const IgemmArgs& p, const int vblock, const int col_off) {
    constexpr int BM = 256, WN = 16 * NT, BN = 4 * WN;
    constexpr int N0 = (NT + 1) / 2, N1 = NT / 2;            // column tiles of a wave: first group (phases 0, 3), second group (phases 1, 2)
    constexpr int ROWB = 128;                                // bytes per LDS row = one 64-deep K step of bf16
    constexpr int XBYTES = BM * ROWB, WBYTES = BN * ROWB;    // one K step of X rows / of W rows
    // LDS: [X, buffer 0][X, buffer 1][W, buffer 0][W, buffer 1] -- both buffers of an operand within the 16-bit immediate offset of
    // ONE base register per 32-deep k step
    constexpr int WREG = 2 * XBYTES;
    constexpr int INFLIGHT = 2 + N1 + 2;                    // DMA instructions of the three units (q0, n1, q1) that stay in flight over a K-step boundary
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- tile of this workgroup: XCD x (= blockIdx % 8, blocks are dealt round-robin) owns a contiguous run of row blocks, column
    // tiles of one row block adjacent (they share the gathered X rows in that XCD's L2)
    // (a partial launch -- the tail split of bgemm_t256_launch -- starts at row block rb0 and takes column tiles col_off + k col_stride)
    const int nct = p.nct_l ? p.nct_l : (p.N + BN - 1) / BN;
    const int x = vblock & 7, loc = vblock >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    if (loc >= (rb_hi - rb_lo) * nct) return;
    const int bm = p.rb0 + rb_lo + loc / nct;
    const int n0 = col_off + (loc % nct) * (p.col_stride ? p.col_stride : BN);

    const int steps0 = p.c0 >> 6, steps1 = p.c1 >> 6;        // channel counts are multiples of 64 (launcher)
    const int nk = p.taps * (steps0 + steps1);

    // ---- DMA geometry.  A 1-KB piece = 8 LDS rows; lane -> (row r8 of the piece, 16-byte position pp); the chunk FETCHED for position
    // pp of tile row r is pp ^ ((r >> 1) & 7) (source-side swizzle; the fragment reads apply the same XOR).  A wave's pieces of
    // one operand are 16 rows apart, so that they share the lane's swizzle and ONE offset register: the displacement of a piece is
    // wave-uniform (a scalar add).
    const int r8 = lane >> 3, pp = lane & 7;
    // X: unit q0 = the first 64 rows of both wave rows (tile rows 0-63, 128-191), unit q1 = the other 64 (64-127, 192-255); a wave
    // moves rows xb + {0, 16} (+ 64 for q1) + r8
    const int xb = (wave < 4 ? 0 : 128) + 32 * ((wave & 3) >> 1) + 8 * (wave & 1);
    // W: wave column wave >> 1, rows wb + 16 i + r8, i < NT: the first N0 pieces are unit n0 (the first 16 N0 columns of the wave
    // column), the other two unit n1
    const int wb = (wave >> 1) * WN + 8 * (wave & 1);

    const char* wsrc = reinterpret_cast<const char*>(p.w16) + (size_t)n0 * p.ldw * 2;
    const __amdgpu_buffer_rsrc_t rw_on = t256_rsrc(wsrc, 0x7FFFFFF0);      // (N is a multiple of the tile width: no column masks)
    const __amdgpu_buffer_rsrc_t r_off = t256_rsrc(wsrc, 0);               // zero records: every load returns zeros
    const unsigned ldw2 = (unsigned)p.ldw * 2u;
    const unsigned w_off = (unsigned)(wb + r8) * ldw2 + (unsigned)((pp ^ (((wb + r8) >> 1) & 7)) * 16);

    // X source offsets.  LIN: row * row stride + chunk for the lane's first row, the other three a scalar displacement further; rows
    // beyond M fall out of the descriptor window (the range check covers the vector offset; the K position rides in the scalar
    // offset, outside it).  Conv: the PIXEL under tap (0, 0) of the lane's four rows (tile rows xb + r8 + {0, 16, 64, 80}), relative to
    // a descriptor based pad rows + pad pixels BEFORE the tile's first image, so that it is never negative (times the source's row
    // stride at issue: one 24-bit multiply-add); the tap's displacement (ky Ws + kx) x stride rides in the scalar offset; a 9-bit
    // mask per row says which taps fall inside the image (zero padding, rows beyond M: out of window).
    constexpr int NXO = LIN ? 1 : 4;
    unsigned x_off[NXO];                                     // conv: pixel index (20 bits) | tap mask << 20
    const int hw_out = p.Ho * p.Wo, hw_in = p.Hs * p.Ws;
    const int img0 = LIN ? 0 : (bm * BM) / hw_out;
    const H* const a0 = reinterpret_cast<const H*>(p.a0);
    const H* const a1 = reinterpret_cast<const H*>(p.a1);
    const int kw = p.kw, pad_x = p.pad_x < 0 ? p.pad : p.pad_x;       // (3x3: kw = 3, pad_x = pad; the sub-pixel 2x2 kernels: see bgemm_up2x)
    const long xrow0 = LIN ? (long)bm * BM : (long)img0 * hw_in - (long)(p.pad * p.Ws + pad_x);
    const int xrows = min(BM, p.M - bm * BM);                // valid rows of this tile
    int k_src = 0, k_chunk = 0, k_tap = 0;                   // issue side: source, 64-channel chunk within it, tap
    unsigned ld2 = (unsigned)p.lda0 * 2u;
    const unsigned x_kcb = (unsigned)((pp ^ (((xb + r8) >> 1) & 7)) * 16);
    auto x_setup = [&](const unsigned l2) {
        if constexpr (LIN) {
            x_off[0] = (unsigned)(xb + r8) * l2 + x_kcb;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = xb + (j & 1) * 16 + (j >> 1) * 64 + r8;
                const int m = bm * BM + r;
                unsigned mask = 0, off = 0;
                if (m < p.M) {
                    const int img = m / hw_out;
                    const int rem = m - img * hw_out;
                    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                    off = (unsigned)(((img - img0) * p.Hs + oy * p.stride) * p.Ws + ox * p.stride);      // a PIXEL index: x row stride at issue
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int ky = kw == 2 ? t >> 1 : t / 3, kx = t - kw * ky;
                        const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - pad_x + kx;
                        mask |= (t < p.taps && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) ? (1u << t) : 0u;
                    }
                }
                x_off[j] = off | (mask << 20);
            }
        }
    };
    x_setup(ld2);
    bool x_live = true;
    auto x_rsrc = [&]() {
        const H* base = (k_src ? a1 : a0) + xrow0 * (long)(k_src ? p.lda1 : p.lda0);
        return t256_rsrc(base, !x_live ? 0 : LIN ? (int)((unsigned)xrows * ld2) : 0x7FFFFFF0);
    };
    __amdgpu_buffer_rsrc_t rx = x_rsrc();

    // One unit of a K step into buffer `buf`.  Units are issued in the order q0, n1, q1, n0 of a K step (X's position advances behind
    // q1, W's behind n0); a unit past the last K step is issued all the same, against a zero-record descriptor (it writes zeros where
    // nothing reads any more): every phase then has the same number of DMA instructions in front of the counted waits.
    unsigned x_so = 0;                                       // scalar offset of the K step being fetched from X
    auto issue_x = [&](const int half, const int b) {          // half 0: unit q0, 1: unit q1
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = half * 2 + i;
            unsigned off;
            if constexpr (LIN) {
                off = x_off[0] + (unsigned)(16 * i + 64 * half) * ld2;
            } else {
                off = ((x_off[j] >> (20 + k_tap)) & 1u) ? __umul24(x_off[j] & 0xFFFFFu, ld2) + x_kcb : T256_OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + b * XBYTES + (xb + 16 * i + 64 * half) * ROWB), 16, off, x_so, 0, 0);
        }
    };
    auto advance_x = [&]() {                                 // to the next K step: tap fastest, then chunk, then source
        if (!x_live) return;
        if (++k_tap == p.taps) {
            k_tap = 0;
            if (++k_chunk == (k_src ? steps1 : steps0)) {
                k_chunk = 0;
                if (k_src == 0 && steps1 > 0) {
                    k_src = 1;
                    ld2 = (unsigned)p.lda1 * 2u;
                    if constexpr (LIN) x_setup(ld2);         // (conv: the pixel indices and masks do not depend on the source)
                } else {
                    x_live = false;
                }
                rx = x_rsrc();
            }
        }
        const int dy = LIN ? 0 : kw == 2 ? k_tap >> 1 : k_tap / 3, dx = LIN ? 0 : k_tap - kw * dy;
        x_so = (unsigned)k_chunk * 128u + (unsigned)(dy * p.Ws + dx) * ld2;
    };
    int w_kt = 0;                                            // K step the W units fetch next (n1 first, n0 advances it)
    auto issue_w = [&](const int grp, const int b) {
        const __amdgpu_buffer_rsrc_t rw = w_kt < nk ? rw_on : r_off;
        const unsigned so = (unsigned)w_kt * 128u;
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < N0; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + WREG + b * WBYTES + (wb + 16 * i) * ROWB), 16, w_off, so + (unsigned)(16 * i) * ldw2, 0, 0);
            ++w_kt;
        } else {
#pragma unroll
            for (int i = N0; i < NT; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + WREG + b * WBYTES + (wb + 16 * i) * ROWB), 16, w_off, so + (unsigned)(16 * i) * ldw2, 0, 0);
        }
    };

    // ---- fragments: v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the first operand: lane -> (row fl of a 16-row tile, k chunk
    // fq of the 32-deep step); the result holds pixel fl, channels 4 fq + {0..3} of the 16 x 16 tile in its four registers
    const int fl = lane & 15, fq = lane >> 4;
    const int fsw = (fl >> 1) & 7;                           // tiles start at multiples of 16 rows: they do not enter (row >> 1) & 7
    const unsigned fo0 = (unsigned)(((0 + fq) ^ fsw) * 16), fo1 = (unsigned)(((4 + fq) ^ fsw) * 16);
    // LDS addresses of the fragment reads: per (buffer, 32-deep k step) ONE base register for X and one for W, every tile an immediate
    // offset below 64 KB.  (Made opaque to the compiler: with the second buffer beyond the 16-bit offset field it otherwise
    // materialises one address per read, parks them in scratch and reloads them -- behind vmcnt(0) -- in the loop.)
    typedef __attribute__((address_space(3))) const hx8<H>* lds_frag;
    typedef __attribute__((address_space(3))) const char* lds_cptr;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_cptr)smem;
    unsigned xa[2], wa[2];
    xa[0] = lds0 + (wr * 128 + fl) * ROWB + fo0;
    xa[1] = lds0 + (wr * 128 + fl) * ROWB + fo1;
    wa[0] = lds0 + WREG + (wc * WN + fl) * ROWB + fo0;
    wa[1] = lds0 + WREG + (wc * WN + fl) * ROWB + fo1;
    asm volatile("" : "+v"(xa[0]), "+v"(xa[1]), "+v"(wa[0]), "+v"(wa[1]));
    auto ldsf = [](const unsigned addr) { return *(lds_frag)(lds_cptr)(unsigned long long)addr; };
    hx8<H> xf[4][2], wf0[N0][2], wf1[N1][2];
    f32x4v acc[2][4][NT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[h][mt][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};

    auto read_x = [&](const int b, const int half) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            xf[mt][0] = ldsf(xa[0] + b * XBYTES + (half * 64 + mt * 16) * ROWB);
            xf[mt][1] = ldsf(xa[1] + b * XBYTES + (half * 64 + mt * 16) * ROWB);
        }
    };
    auto read_w0 = [&](const int b) {
#pragma unroll
        for (int nt = 0; nt < N0; ++nt) {
            wf0[nt][0] = ldsf(wa[0] + b * WBYTES + nt * 16 * ROWB);
            wf0[nt][1] = ldsf(wa[1] + b * WBYTES + nt * 16 * ROWB);
        }
    };
    auto read_w1 = [&](const int b) {
#pragma unroll
        for (int nt = 0; nt < N1; ++nt) {
            wf1[nt][0] = ldsf(wa[0] + b * WBYTES + (N0 + nt) * 16 * ROWB);
            wf1[nt][1] = ldsf(wa[1] + b * WBYTES + (N0 + nt) * 16 * ROWB);
        }
    };
    auto mma0 = [&](const int half) {                        // quadrant (half, first column group)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < N0; ++nt)
                    acc[half][mt][nt] = mfma_16x16x32(wf0[nt][ks], xf[mt][ks], acc[half][mt][nt]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto mma1 = [&](const int half) {                        // quadrant (half, second column group)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < N1; ++nt)
                    acc[half][mt][N0 + nt] = mfma_16x16x32(wf1[nt][ks], xf[mt][ks], acc[half][mt][N0 + nt]);
        __builtin_amdgcn_s_setprio(0);
    };
    // the two halves of a phase: [reads + DMA of this wave] | barrier | [MFMAs of this wave] | barrier; the fragment reads are
    // retired in front of the first barrier (WAR, see the header), the compiler may move nothing across either
    auto phase_sync_a = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0) alone (vmcnt / expcnt fields at their maxima)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto phase_sync_b = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: K step 0 whole, three units (q0, n1, q1) of K step 1
    issue_x(0, 0);                                           // (the X position starts at source 0, chunk 0, tap 0: scalar offset 0)
    issue_w(1, 0);
    issue_x(1, 0);
    advance_x();
    issue_w(0, 0);
    issue_x(0, 1);
    issue_w(1, 1);
    issue_x(1, 1);
    advance_x();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
    __builtin_amdgcn_s_barrier();                            // K step 0 has landed for every wave
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();               // the second wave row runs one barrier behind the first

    // K step t in buffer B: phases 0..3.  DMA issued meanwhile: n0 of step t + 1 (other buffer), then q0, n1, q1 of
    // step t + 2 (THIS buffer, each one phase behind its last readers).
    auto kstep = [&](auto Bc) {
        constexpr int B = decltype(Bc)::value;
        constexpr int mine = B, other = 1 - B;
        // phase 0: quadrant (rows 0-63, first column group)
        read_w0(B);
        read_x(B, 0);
        issue_w(0, other);                                   // n0 of the next K step (the other buffer's was read for the last time in its phase 3)
        phase_sync_a();
        mma0(0);
        phase_sync_b();
        // phase 1: (rows 0-63, second column group)
        read_w1(B);
        issue_x(0, mine);                                    // q0 of the step after next: rows 0-63 were read for the last time in phase 0
        phase_sync_a();
        mma1(0);
        phase_sync_b();
        // phase 2: (rows 64-127, second column group)
        read_x(B, 1);
        issue_w(1, mine);                                    // n1: read for the last time in phase 1
        phase_sync_a();
        mma1(1);
        phase_sync_b();
        // phase 3: (rows 64-127, first column group); the K-step boundary
        read_w0(B);
        issue_x(1, mine);                                    // q1: rows 64-127 were read for the last time in phase 2
        advance_x();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");   // the next K step has landed; q0, n1, q1 of the one after stay in flight
        phase_sync_a();
        mma0(1);
        phase_sync_b();
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
    }
    if (kt < nk) kstep(std::integral_constant<int, 0>{});
    if (wr == 0) __builtin_amdgcn_s_barrier();               // the first wave row waits for the second one's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the out-of-window units behind the last K step
    __builtin_amdgcn_s_barrier();                            // LDS becomes the epilogue's staging area
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue.  A lane holds (pixel fl, channels 4 fq + e) of each 16 x 16 tile; the wave transposes 32 rows at a time through
    // LDS (32 x WN floats, rows padded by 4 floats) so that a lane ends up with 8 consecutive channels of one row: bias and
    // time-embedding rows in fp32, the residual as 8 bf16, one rounding, 16-byte stores of whole row segments.
    constexpr int SLD = WN + 4;
    float* const st = reinterpret_cast<float*>(smem) + wave * 32 * SLD;
    constexpr int WNO = LIN ? WN : WN;                       // (GEGLU halves the output width below)
    const bool geglu = p.geglu != 0;
    const int wout = geglu ? WN / 2 : WNO;                   // output columns of this wave
    const int lpr = wout / 8;                                // lanes per output row (8 columns each)
    const int rpi = 64 / lpr;                                // rows per pass of the wave (NT = 5: 6 rows, 60 lanes)
    const int ocol = (lane % lpr) * 8;
    const int orow = lane / lpr;
    const int ncol0 = geglu ? (n0 + wc * WN) / 2 : n0 + wc * WN;   // first output column of the wave
    const int Nout = geglu ? p.N / 2 : p.N;
    const int nn = ncol0 + ocol;
    const bool lane_on = orow < rpi && nn < Nout;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
    if (!geglu && p.bias && lane_on) {
        b0 = *reinterpret_cast<const f32x4*>(p.bias + nn);
        b1 = *reinterpret_cast<const f32x4*>(p.bias + nn + 4);
    }
    // The rows a lane finishes in one 32-row chunk (orow, orow + rpi, ...: at most ITER_N of them) read the residual and / or the
    // time-embedding row from global memory.  Inside the row loop each was a load -> wait -> store round of its own: 24 exposed memory
    // latencies per wave and tile (the conv tiles ran 183 us for 129 us of MFMAs).  Now the bf16 residual rows of a chunk (16 bytes
    // per row) are ALL requested before the chunk is staged, and the time-embedding row is read once per chunk when the chunk's rows
    // belong to one sample (they nearly always do).  The fp32 residual of the test entry points stays in the loop.
    constexpr int RPI_N = 64 / (2 * NT), ITER_N = (32 + RPI_N - 1) / RPI_N;      // (GEGLU: 16 rows per pass, 2 passes <= ITER_N)
    const bool pf_resid = !geglu && p.resid != nullptr && p.resid_bf16;
    auto out_row = [&](const int m) -> size_t {              // (the sub-pixel kernels scatter their pixels over the 2x map)
        if constexpr (!LIN) {
            if (p.osy) {
                const int img = m / hw_out, rem = m - img * hw_out;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                return ((size_t)img * p.Ho * p.osy + (size_t)(oy * p.osy + p.ooy)) * (size_t)(p.Wo * p.osx) + (size_t)(ox * p.osx + p.oox);
            }
        }
        return (size_t)m;
    };
    // Four 32-row chunks c = (half, pair of row tiles).  Order: stage(c) frees the chunk's 8 NT accumulator registers; the loads of
    // chunk c + 1 are issued before the rows of chunk c are finished (two prefetch sets alternate), so that only the first chunk's
    // loads are exposed -- and never while all accumulators are still live: requested in front of stage(0), the prefetch registers
    // pushed the allocator into spilling accumulators INSIDE the K loop.
    f32x4 pfr[2][ITER_N], rbc0[2], rbc1[2];
    bool rb_one[2];
    auto chunk_m0 = [&](const int c) { return bm * BM + wr * 128 + (c >> 1) * 64 + (c & 1) * 32; };
    auto issue_loads = [&](const int c, const int set) {
        const int m0 = chunk_m0(c);
#pragma unroll
        for (int it = 0; it < ITER_N; ++it) {
            const int r = orow + it * rpi;
            pfr[set][it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (pf_resid && lane_on && r < 32 && m0 + r < p.M)
                pfr[set][it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const H*>(p.resid) + out_row(m0 + r) * p.ldr + nn);      // 8 bf16, raw
        }
        // one time-embedding row for the chunk if its first and last row belong to the same sample (wave-uniform test)
        const int m_last = min(m0 + 31, p.M - 1);
        rb_one[set] = !geglu && p.rowbias != nullptr && m0 < p.M && m0 / p.rows_per_sample == m_last / p.rows_per_sample;
        rbc0[set] = f32x4{0.f, 0.f, 0.f, 0.f};
        rbc1[set] = rbc0[set];
        if (rb_one[set] && lane_on) {
            const float* rb = p.rowbias + (size_t)(m0 / p.rows_per_sample) * p.rb_ld + nn;
            rbc0[set] = *reinterpret_cast<const f32x4*>(rb);
            rbc1[set] = *reinterpret_cast<const f32x4*>(rb + 4);
        }
    };
    auto stage = [&](const int c) {
        const int h = c >> 1, pr = c & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int mt = pr * 2 + t;
            if (geglu) {
                if constexpr (NT == 4) {                 // value tiles 0, 1 | gate tiles 2, 3 of the wave's 64 packed columns
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int cb = n0 + wc * WN + nt * 16 + 4 * fq;
                        const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
                        const f32x4 bg = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + cb + 32) : f32x4{0.f, 0.f, 0.f, 0.f};
                        f32x4 y;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            y[e] = (acc[h][mt][nt][e] * p.alpha + bv[e]) * gelu_gate16<H>(acc[h][mt][nt + 2][e] * p.alpha + bg[e]);
                        *reinterpret_cast<f32x4*>(st + (t * 16 + fl) * SLD + nt * 16 + 4 * fq) = y;
                    }
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = acc[h][mt][nt][e];
                    *reinterpret_cast<f32x4*>(st + (t * 16 + fl) * SLD + nt * 16 + 4 * fq) = y;
                }
            }
        }
    };
    // (the staging block is private to the wave: its own LDS writes are in order with its own reads)
    auto rows = [&](const int c, const int set) {
        const int m0 = chunk_m0(c);
#pragma unroll
        for (int it = 0; it < ITER_N; ++it) {
            const int r = orow + it * rpi;
            const int m = m0 + r;
            if (!lane_on || r >= 32 || m >= p.M) continue;
            const size_t orw = out_row(m);
            f32x4 y0 = *reinterpret_cast<const f32x4*>(st + r * SLD + ocol);
            f32x4 y1 = *reinterpret_cast<const f32x4*>(st + r * SLD + ocol + 4);
            if (!geglu) {
                if (p.alpha != 1.0f) { y0 *= p.alpha; y1 *= p.alpha; }
                y0 += b0; y1 += b1;
                if (rb_one[set]) {
                    y0 += rbc0[set]; y1 += rbc1[set];
                } else if (p.rowbias) {
                    const float* rb = p.rowbias + (size_t)(m / p.rows_per_sample) * p.rb_ld + nn;
                    y0 += *reinterpret_cast<const f32x4*>(rb);
                    y1 += *reinterpret_cast<const f32x4*>(rb + 4);
                }
                if (pf_resid) {
                    const hx8<H> rr = __builtin_bit_cast(hx8<H>, pfr[set][it]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { y0[e] += (float)rr[e]; y1[e] += (float)rr[4 + e]; }
                } else if (p.resid) {
                    if (p.resid_bf16) {
                        const hx8<H> rr = *reinterpret_cast<const hx8<H>*>(reinterpret_cast<const H*>(p.resid) + orw * p.ldr + nn);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { y0[e] += (float)rr[e]; y1[e] += (float)rr[4 + e]; }
                    } else {
                        y0 += *reinterpret_cast<const f32x4*>(p.resid + orw * p.ldr + nn);
                        y1 += *reinterpret_cast<const f32x4*>(p.resid + orw * p.ldr + nn + 4);
                    }
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { y0[e] = fmaxf(y0[e], 0.f); y1[e] = fmaxf(y1[e], 0.f); }
                }
            }
            if (p.out_f32) {
                float* o = p.out + orw * p.ldc + nn;
                *reinterpret_cast<f32x4*>(o) = y0;
                *reinterpret_cast<f32x4*>(o + 4) = y1;
            } else {
                hx8<H> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (H)y0[e]; o[4 + e] = (H)y1[e]; }
                *reinterpret_cast<hx8<H>*>(reinterpret_cast<H*>(p.out) + orw * p.ldc + nn) = o;
            }
        }
    };
    stage(0);
    issue_loads(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c + 1 < 4) issue_loads(c + 1, (c + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        rows(c, c & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < 4) stage(c + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- row-block sums for a following GroupNorm (IgemmArgs::rbsum; RB instances only).  Taken AFTER the epilogue, from the stored
    // tile: every lane reads back the 8-column pieces IT has just written (its own stores, still in the L2 of its XCD) and adds them
    // in the canonical order of rowblock_sums (norm.hip) -- per 32-row chunk pass by pass (plain add; squares by fma), chunk 0 + chunk
    // 1 of a 64-row half, then the rpi lanes of a piece in lane order through LDS; the first of them stores the piece's 16 numbers.
    // Adding them inside rows() instead cost nothing in time but 16 live values at the epilogue's register peak, and the allocator
    // answered with reloads -- each a vmcnt(0) behind the DMA -- INSIDE the K loop; here every accumulator is dead and nothing
    // new lives across the loop (the lane's values are rebuilt from an opaque copy of the lane id).
    if constexpr (RB) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tile's stores have left the wave
        int l2 = lane;
        asm volatile("" : "+v"(l2));
        const int lpr2 = WN / 8, rpi2 = 64 / lpr2;
        const int ocol2 = (l2 % lpr2) * 8, orow2 = l2 / lpr2;
        const int nn2 = n0 + wc * WN + ocol2;
        const bool on2 = orow2 < rpi2 && nn2 < p.N;
        float* const rbl = reinterpret_cast<float*>(smem) + wave * 32 * SLD + l2 * 16;   // inside the wave's OWN staging block (dead now; the other waves' may not be)
        const H* const outp = reinterpret_cast<const H*>(p.out);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 hs0 = {0.f, 0.f, 0.f, 0.f}, hs1 = hs0, hq0 = hs0, hq1 = hs0;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int m0 = bm * BM + wr * 128 + half * 64 + cc * 32;
                hx8<H> v[ITER_N];
#pragma unroll
                for (int it = 0; it < ITER_N; ++it) {
                    const int r = orow2 + it * rpi2;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[it][e] = (H)0.f;
                    if (on2 && r < 32 && m0 + r < p.M) v[it] = *reinterpret_cast<const hx8<H>*>(outp + (size_t)(m0 + r) * p.ldc + nn2);
                }
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, q0 = s0, q1 = s0;
#pragma unroll
                for (int it = 0; it < ITER_N; ++it) {
                    const int r = orow2 + it * rpi2;
                    if (on2 && r < 32 && m0 + r < p.M) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v0 = (float)v[it][e], v1 = (float)v[it][4 + e];
                            s0[e] += v0; s1[e] += v1;
                            q0[e] = __builtin_fmaf(v0, v0, q0[e]); q1[e] = __builtin_fmaf(v1, v1, q1[e]);
                        }
                    }
                }
                if (cc == 0) { hs0 = s0; hs1 = s1; hq0 = q0; hq1 = q1; }
                else { hs0 += s0; hs1 += s1; hq0 += q0; hq1 += q1; }
            }
            asm volatile("" ::: "memory");                   // (the other lanes' reads of the previous half are done: LDS is in order per wave)
            *reinterpret_cast<f32x4*>(rbl) = hs0; *reinterpret_cast<f32x4*>(rbl + 4) = hs1;
            *reinterpret_cast<f32x4*>(rbl + 8) = hq0; *reinterpret_cast<f32x4*>(rbl + 12) = hq1;
            // the reads below fetch what OTHER lanes of the wave have just written: the compiler sees one base pointer with disjoint
            // offsets and would hoist them over the stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const long long mh = (long long)bm * BM + wr * 128 + half * 64;
            if (orow2 == 0 && nn2 < p.N && mh < p.M) {
                f32x4 t0 = hs0, t1 = hs1, u0 = hq0, u1 = hq1;
                for (int jj = 1; jj < rpi2; ++jj) {
                    const float* o = rbl + jj * lpr2 * 16;
                    t0 += *reinterpret_cast<const f32x4*>(o); t1 += *reinterpret_cast<const f32x4*>(o + 4);
                    u0 += *reinterpret_cast<const f32x4*>(o + 8); u1 += *reinterpret_cast<const f32x4*>(o + 12);
                }
                float* d = p.rbsum + ((size_t)(mh >> 6) * p.N + nn2) * 2;
                *reinterpret_cast<f32x4*>(d) = f32x4{t0[0], u0[0], t0[1], u0[1]};
                *reinterpret_cast<f32x4*>(d + 4) = f32x4{t0[2], u0[2], t0[3], u0[3]};
                *reinterpret_cast<f32x4*>(d + 8) = f32x4{t1[0], u1[0], t1[1], u1[1]};
                *reinterpret_cast<f32x4*>(d + 12) = f32x4{t1[2], u1[2], t1[3], u1[3]};
            }
        }
    }
}

template <typename H, int NT, bool LIN, bool RB = false>
__global__ __launch_bounds__(512) void bgemm_t256_kernel(const IgemmArgs p) {
    t256_body<H, NT, LIN, RB>(p, blockIdx.x, p.col_off);
}

#ifdef E2V_AB          // measured, not adopted
// The tail split's two tile widths in ONE launch (two launches on a stream run one after the other, each a third of the chip):
// groups of eight workgroups alternate between 256 x 192 tiles at column offset col_off and 256 x 128 tiles at col_off2; the low
// three bits of the block index -- the XCD -- stay what they are.
template <typename H, bool LIN>
__global__ __launch_bounds__(512) void bgemm_t256_tail_kernel(const IgemmArgs p) {
    const int vblock = (int)(((blockIdx.x >> 4) << 3) | (blockIdx.x & 7));
    if ((blockIdx.x >> 3) & 1) t256_body<H, 2, LIN>(p, vblock, p.col_off2);
    else t256_body<H, 3, LIN>(p, vblock, p.col_off);
}
#endif

// ---- persistent form for the linears (taps = 1) ----------------------------------------------------------------------------------
// A short-K tile (K = 320: five K steps, ~6 us of MFMAs) of the kernel above pays, around them: the workgroup launch, the latency
// of its first K step with nothing in flight before it (~2-3 us), and an epilogue during which the CU fetches nothing; measured at
// M = 884 736, K = 320: 17-21 us per tile, 2.6 TB/s of HBM traffic.  Here a workgroup stays on its CU and walks the tile list of its
// XCD (slot s of 32 takes tiles s, s + 32, ...; column tiles of a row block adjacent in time and place, as above), and the DMA
// stream never stops: the units behind a tile's last K steps are the next tile's first ones, so that the epilogue -- straight
// from the accumulator registers, no LDS: both K-step buffers are being refilled -- runs with two K steps of the next tile in
// flight, and its stores are never waited for.  The two wave rows re-align for the epilogue (one barrier) and split again
// behind it.  Same K loop, same k order, same arithmetic: bit-identical to the kernel above.
// BLDS: the bias vector ([N] floats, at most the LDS the operand buffers leave free) is copied into LDS once per workgroup and the
// epilogue reads it from there.  As buffer loads at the head of the epilogue the NT bias quads were the youngest entries of the wave's
// in-order memory queue: waiting for them drained the DMA of the next tile's first K steps (~0.7 us per tile; the residual-free
// K = 320 / 640 projections -- QKV, GEGLU -- have no other load there).
template <typename H, int NT, bool GEGLU, bool F32IO, bool BLDS>      // F32IO: fp32 output and residual (the op-level test entry points); else bf16 (the graph)
__global__ __launch_bounds__(512) void bgemm_t256p_kernel(const IgemmArgs p) {
    constexpr int BM = 256, WN = 16 * NT, BN = 4 * WN;
    constexpr int N0 = (NT + 1) / 2, N1 = NT / 2;
    constexpr int ROWB = 128;
    constexpr int XBYTES = BM * ROWB, WBYTES = BN * ROWB;
    constexpr int WREG = 2 * XBYTES;
    constexpr int INFLIGHT = 2 + N1 + 2;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nct = p.N / BN;
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int rb_lo = (int)(((long)x * p.nbm) >> 3), rb_hi = (int)(((long)(x + 1) * p.nbm) >> 3);
    const int ntx = (rb_hi - rb_lo) * nct;                   // tiles of this XCD
    if (slot >= ntx) return;
    float* const lds_bias = reinterpret_cast<float*>(smem + 2 * (XBYTES + WBYTES));      // behind the operand buffers
    if constexpr (BLDS) {
        for (int i = tid * 4; i < p.N; i += 512 * 4)         // (N is a multiple of 64)
            *reinterpret_cast<f32x4*>(lds_bias + i) = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                      // (vmcnt(0) + lgkmcnt(0) + barrier: nothing of this is pending when the DMA ring starts)
    }
    const int steps0 = p.c0 >> 6, steps1 = p.c1 >> 6;
    const int nk = steps0 + steps1;

    const int r8 = lane >> 3, pp = lane & 7;
    const int xb = (wave < 4 ? 0 : 128) + 32 * ((wave & 3) >> 1) + 8 * (wave & 1);
    const int wb = (wave >> 1) * WN + 8 * (wave & 1);
    const unsigned ldw2 = (unsigned)p.ldw * 2u;
    const unsigned w_off = (unsigned)(wb + r8) * ldw2 + (unsigned)((pp ^ (((wb + r8) >> 1) & 7)) * 16);
    const unsigned x_kcb = (unsigned)((pp ^ (((xb + r8) >> 1) & 7)) * 16);
    const H* const a0 = reinterpret_cast<const H*>(p.a0);
    const H* const a1 = reinterpret_cast<const H*>(p.a1);

    // ---- issue side: an X cursor and a W cursor, each (tile, K step), running up to two K steps ahead of the compute side and
    // across tile boundaries; past the XCD's last tile they fetch through zero-record descriptors (zeros into rows nobody reads)
    int xt = slot, x_src = 0, x_chunk = 0;
    unsigned ld2 = (unsigned)p.lda0 * 2u;
    unsigned x_off = (unsigned)(xb + r8) * ld2 + x_kcb;
    auto x_rsrc = [&]() {
        const int bm = rb_lo + xt / nct;
        const bool live = xt < ntx;
        const int rows = live ? min(BM, p.M - bm * BM) : 0;
        const H* base = (x_src ? a1 : a0) + (live ? (long)bm * BM * (x_src ? p.lda1 : p.lda0) : 0L);
        return t256_rsrc(base, (int)((unsigned)rows * ld2));
    };
    __amdgpu_buffer_rsrc_t rx = x_rsrc();
    auto issue_x = [&](const int half, const int b) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + b * XBYTES + (xb + 16 * i + 64 * half) * ROWB), 16,
                                                     x_off + (unsigned)(16 * i + 64 * half) * ld2, (unsigned)x_chunk * 128u, 0, 0);
    };
    auto advance_x = [&]() {
        if (++x_chunk < (x_src ? steps1 : steps0)) return;
        x_chunk = 0;
        if (x_src == 0 && steps1 > 0) {
            x_src = 1;
        } else {
            x_src = 0;
            xt += nslots;
        }
        const unsigned l2 = (unsigned)(x_src ? p.lda1 : p.lda0) * 2u;
        if (l2 != ld2) {                                     // (wave-uniform: the two sources of a concat have their own row strides)
            ld2 = l2;
            x_off = (unsigned)(xb + r8) * ld2 + x_kcb;
        }
        rx = x_rsrc();
    };
    int wt = slot, w_k = 0;
    auto w_rsrc = [&]() {
        const bool live = wt < ntx;
        const char* base = reinterpret_cast<const char*>(p.w16) + (live ? (size_t)((wt % nct) * BN) * p.ldw * 2 : (size_t)0);
        return t256_rsrc(base, live ? 0x7FFFFFF0 : 0);
    };
    __amdgpu_buffer_rsrc_t rw = w_rsrc();
    auto issue_w = [&](const int grp, const int b) {
        const unsigned so = (unsigned)w_k * 128u;
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < N0; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + WREG + b * WBYTES + (wb + 16 * i) * ROWB), 16, w_off, so + (unsigned)(16 * i) * ldw2, 0, 0);
            if (++w_k == nk) {
                w_k = 0;
                wt += nslots;
                rw = w_rsrc();
            }
        } else {
#pragma unroll
            for (int i = N0; i < NT; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + WREG + b * WBYTES + (wb + 16 * i) * ROWB), 16, w_off, so + (unsigned)(16 * i) * ldw2, 0, 0);
        }
    };

    const int fl = lane & 15, fq = lane >> 4;
    const int fsw = (fl >> 1) & 7;
    const unsigned fo0 = (unsigned)(((0 + fq) ^ fsw) * 16), fo1 = (unsigned)(((4 + fq) ^ fsw) * 16);
    typedef __attribute__((address_space(3))) const hx8<H>* lds_frag;
    typedef __attribute__((address_space(3))) const char* lds_cptr;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_cptr)smem;
    unsigned xa[2], wa[2];
    xa[0] = lds0 + (wr * 128 + fl) * ROWB + fo0;
    xa[1] = lds0 + (wr * 128 + fl) * ROWB + fo1;
    wa[0] = lds0 + WREG + (wc * WN + fl) * ROWB + fo0;
    wa[1] = lds0 + WREG + (wc * WN + fl) * ROWB + fo1;
    asm volatile("" : "+v"(xa[0]), "+v"(xa[1]), "+v"(wa[0]), "+v"(wa[1]));
    auto ldsf = [](const unsigned addr) { return *(lds_frag)(lds_cptr)(unsigned long long)addr; };
    hx8<H> xf[4][2], wf0[N0][2], wf1[N1][2];
    f32x4v acc[2][4][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[h][mt][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    // (the buffer is a RUN-TIME parity here -- one K-step body for the whole kernel, whatever the parity a tile starts on: the
    // fragment bases of the K step are two vector adds away, the tiles stay immediate offsets)
    unsigned xc[2], wcb[2];                                  // fragment bases of the current K step's buffer
    auto set_buffer = [&](const int b) {
        xc[0] = xa[0] + (unsigned)b * XBYTES; xc[1] = xa[1] + (unsigned)b * XBYTES;
        wcb[0] = wa[0] + (unsigned)b * WBYTES; wcb[1] = wa[1] + (unsigned)b * WBYTES;
    };
    auto read_x = [&](const int half) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            xf[mt][0] = ldsf(xc[0] + (half * 64 + mt * 16) * ROWB);
            xf[mt][1] = ldsf(xc[1] + (half * 64 + mt * 16) * ROWB);
        }
    };
    auto read_w0 = [&]() {
#pragma unroll
        for (int nt = 0; nt < N0; ++nt) {
            wf0[nt][0] = ldsf(wcb[0] + nt * 16 * ROWB);
            wf0[nt][1] = ldsf(wcb[1] + nt * 16 * ROWB);
        }
    };
    auto read_w1 = [&]() {
#pragma unroll
        for (int nt = 0; nt < N1; ++nt) {
            wf1[nt][0] = ldsf(wcb[0] + (N0 + nt) * 16 * ROWB);
            wf1[nt][1] = ldsf(wcb[1] + (N0 + nt) * 16 * ROWB);
        }
    };
    // (FIRST: a tile's first K step starts its accumulators from the MFMA's inline zero instead of from registers that a run of
    // 128 / 160 v_mov had to clear after every epilogue -- 5 % of a K = 320 tile)
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
    auto mma0 = [&](const int half, auto firstc) {
        constexpr bool FIRST = decltype(firstc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < N0; ++nt)
                    acc[half][mt][nt] = mfma_16x16x32(wf0[nt][ks], xf[mt][ks], (FIRST && ks == 0) ? zero4 : acc[half][mt][nt]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto mma1 = [&](const int half, auto firstc) {
        constexpr bool FIRST = decltype(firstc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < N1; ++nt)
                    acc[half][mt][N0 + nt] = mfma_16x16x32(wf1[nt][ks], xf[mt][ks], (FIRST && ks == 0) ? zero4 : acc[half][mt][N0 + nt]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto phase_sync_a = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto phase_sync_b = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto kstep = [&](const int mine, auto firstc) {
        const int other = 1 - mine;
        set_buffer(mine);
        read_w0();
        read_x(0);
        issue_w(0, other);
        phase_sync_a();
        mma0(0, firstc);
        phase_sync_b();
        read_w1();
        issue_x(0, mine);
        phase_sync_a();
        mma1(0, firstc);
        phase_sync_b();
        read_x(1);
        issue_w(1, mine);
        phase_sync_a();
        mma1(1, firstc);
        phase_sync_b();
        read_w0();
        issue_x(1, mine);
        advance_x();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        phase_sync_a();
        mma0(1, firstc);
        phase_sync_b();
    };

    // ---- epilogue of tile (bm, n0), straight from the registers: a lane holds pixel fl, channels 4 fq + {0..3} of each 16 x 16 tile.
    // Buffer loads / stores against descriptors based at the tile's first row: one offset register per lane, the (half, row tile)
    // displacement one vector add, the column tile in the scalar offset; rows beyond M fall out of the window.
    // Every load is UNCONDITIONAL (an absent bias / residual is a zero-record descriptor: the buffer unit returns zeros) and every
    // loaded value is consumed: a load under a branch leaves the compiler's vmcnt bookkeeping with a "maybe pending" register at
    // the head of the K loop, and it drains the queue -- the DMA of two K steps -- there.
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    constexpr unsigned osz = F32IO ? 4u : 2u;
    auto epilogue = [&](const int bm, const int n0) {
        // (everything lane-dependent is recomputed here from an opaque copy of the lane id: hoisted out of the tile loop as loop
        // invariants, these values stay live across the K loop and the allocator spills them -- a scratch reload in the loop is a
        // vmcnt(0) behind the DMA)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int fl = lane_e & 15, fq = lane_e >> 4;
        const int rows = min(BM, p.M - bm * BM);
        const int ocol0 = GEGLU ? (n0 + wc * WN) / 2 : n0 + wc * WN;
        const __amdgpu_buffer_rsrc_t ro = t256_rsrc(reinterpret_cast<const char*>(p.out) + (size_t)bm * BM * p.ldc * osz, (int)((unsigned)rows * p.ldc * osz));
        const __amdgpu_buffer_rsrc_t rr = t256_rsrc(reinterpret_cast<const char*>(p.resid) + (p.resid ? (size_t)bm * BM * p.ldr * osz : (size_t)0),
                                                    p.resid ? (int)((unsigned)rows * p.ldr * osz) : 0);
        const __amdgpu_buffer_rsrc_t rb = t256_rsrc(p.bias, p.bias ? p.N * 4 : 0);
        const unsigned vo0 = ((unsigned)(wr * 128 + fl) * p.ldc + ocol0 + 4 * fq) * osz;
        const unsigned vr0 = ((unsigned)(wr * 128 + fl) * p.ldr + n0 + wc * WN + 4 * fq) * osz;
        f32x4 bias[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (BLDS) bias[nt] = *reinterpret_cast<const f32x4*>(lds_bias + n0 + wc * WN + nt * 16 + 4 * fq);
            else bias[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)(n0 + wc * WN + nt * 16 + 4 * fq) * 4u, 0, 0));
        }
        // Two adjacent 16-column tiles leave the wave as 16-byte stores: a lane holds channels 4 fq .. + 3 of BOTH tiles; after
        // v_permlane16_swap (odd 16-lane rows of the first operand <-> even rows of the second) an even-fq lane holds channels
        // 8 (fq >> 1) .. + 7 of the first tile and an odd-fq lane the same channels of the second one -- every row of the wave's 16
        // then gets 64 contiguous bytes per instruction instead of 2 x 32 (the 8-byte stores of the accumulator layout moved 32 bytes
        // per cache line touched).  (Inline asm with the two wait states a VALU-written operand needs in front of a permlane.)
        const unsigned vop0 = ((unsigned)(wr * 128 + fl) * p.ldc + ocol0 + (fq & 1) * 16 + (fq >> 1) * 8) * 2u;
        const unsigned vrp0 = ((unsigned)(wr * 128 + fl) * p.ldr + n0 + wc * WN + (fq & 1) * 16 + (fq >> 1) * 8) * 2u;      // the residual, paired likewise
        auto store_pair = [&](const f32x4 ya, const f32x4 yb, const unsigned vop, const int nt) {      // tiles nt, nt + 1 (bf16 output)
            hx4<H> oa, ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) { oa[e] = (H)ya[e]; ob[e] = (H)yb[e]; }
            u32x2 a = __builtin_bit_cast(u32x2, oa), b = __builtin_bit_cast(u32x2, ob);
            unsigned a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
            const u32x4 o = {a0, a1, b0, b1};
            __builtin_amdgcn_raw_buffer_store_b128(o, ro, vop, (unsigned)(nt * 16) * 2u, 0);
        };
        auto store4 = [&](const f32x4 y, const unsigned vo, const int nt) {
            if constexpr (F32IO) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), ro, vo, (unsigned)(nt * 16) * 4u, 0);
            } else {
                hx4<H> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (H)y[e];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ro, vo, (unsigned)(nt * 16) * 2u, 0);
            }
        };
        if constexpr (GEGLU) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const unsigned vo = vo0 + (unsigned)(h * 64 + mt * 16) * p.ldc * osz;
                    static_assert(!GEGLU || NT == 4, "GEGLU: 256 x 256 tiles");      // value tiles 0, 1 | gate tiles 2, 3 of the wave's 64 packed columns
                    f32x4 yg[2];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            yg[nt][e] = (acc[h][mt][nt][e] * p.alpha + bias[nt][e]) * gelu_gate16<H>(acc[h][mt][nt + 2][e] * p.alpha + bias[nt + 2][e]);
                    if constexpr (F32IO) { store4(yg[0], vo, 0); store4(yg[1], vo, 1); }
                    else store_pair(yg[0], yg[1], vop0 + (unsigned)(h * 64 + mt * 16) * p.ldc * 2u, 0);
                    __builtin_amdgcn_sched_barrier(0);       // one (half, row tile) at a time
                }
        } else {
            // The residual of a whole half (64 rows of the wave: 4 x NT loads) is requested before the first of its values is needed:
            // the K loop's fragment registers are dead here, which is room for 40 dwords of bf16 residual -- one exposed memory
            // latency per half instead of one per 16-row tile (eight per tile; measured on 640 -> 640 at M = 221 184: 40 us per tile
            // against 14 us of MFMAs and a 23 us HBM floor).
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 res[4][NT];
                hx4<H> rb4[4][NT];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const unsigned vr = vr0 + (unsigned)(h * 64 + mt * 16) * p.ldr * osz;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if constexpr (BLDS) {                        // (BLDS launches have no residual: nothing enters the memory queue here)
                            res[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                        } else if constexpr (F32IO) res[mt][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, vr, (unsigned)(nt * 16) * 4u, 0));
                        else if ((nt & 1) == 0 && nt + 1 < NT) {      // two adjacent tiles: 16 bytes per lane (channels 8 (fq >> 1) .. + 7 of tile nt + (fq & 1)), un-paired below
                            const u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(rr, vrp0 + (unsigned)(h * 64 + mt * 16) * p.ldr * 2u, (unsigned)(nt * 16) * 2u, 0);
                            rb4[mt][nt] = __builtin_bit_cast(hx4<H>, u32x2{l[0], l[1]});
                            rb4[mt][nt + 1] = __builtin_bit_cast(hx4<H>, u32x2{l[2], l[3]});
                        } else if (nt == NT - 1 && (NT & 1)) {
                            rb4[mt][nt] = __builtin_bit_cast(hx4<H>, __builtin_amdgcn_raw_buffer_load_b64(rr, vr, (unsigned)(nt * 16) * 2u, 0));
                        }
                    }
                    if constexpr (F32IO) __builtin_amdgcn_sched_barrier(0);      // (fp32 residual, test entry points: 16 dwords per row tile, one row tile at a time)
                    if constexpr (F32IO) {
                        const unsigned vo = vo0 + (unsigned)(h * 64 + mt * 16) * p.ldc * osz;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            f32x4 y;
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[e] = acc[h][mt][nt][e];
                            if (p.alpha != 1.0f) y *= p.alpha;
                            y += bias[nt];
                            if constexpr (!BLDS) y += res[mt][nt];
                            if (p.relu) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
                            }
                            store4(y, vo, nt);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (!F32IO) {
                    __builtin_amdgcn_sched_barrier(0);       // every load of the half is in flight before the first conversion
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const unsigned vo = vo0 + (unsigned)(h * 64 + mt * 16) * p.ldc * osz;
                        const unsigned vop = vop0 + (unsigned)(h * 64 + mt * 16) * p.ldc * 2u;
#pragma unroll
                        for (int nt = 0; nt + 1 < NT && !BLDS; nt += 2) {     // (swap of the first dwords, then of the second ones: see store_pair)
                            u32x2 a = __builtin_bit_cast(u32x2, rb4[mt][nt]), b = __builtin_bit_cast(u32x2, rb4[mt][nt + 1]);
                            unsigned l0 = a[0], l1 = a[1], l2 = b[0], l3 = b[1];
                            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(l0), "+v"(l2), "+v"(l1), "+v"(l3));
                            rb4[mt][nt] = __builtin_bit_cast(hx4<H>, u32x2{l0, l1});
                            rb4[mt][nt + 1] = __builtin_bit_cast(hx4<H>, u32x2{l2, l3});
                        }
                        f32x4 y[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[nt][e] = acc[h][mt][nt][e];
                            if (p.alpha != 1.0f) y[nt] *= p.alpha;
                            y[nt] += bias[nt];
                            if constexpr (!BLDS) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) y[nt][e] += (float)rb4[mt][nt][e];
                            }
                            if (p.relu) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) y[nt][e] = fmaxf(y[nt][e], 0.f);
                            }
                        }
#pragma unroll
                        for (int nt = 0; nt + 1 < NT; nt += 2) store_pair(y[nt], y[nt + 1], vop, nt);
                        if constexpr (NT & 1) store4(y[NT - 1], vo, NT - 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    };

    // ---- prologue: K step 0 of the first tile whole, q0, n1, q1 of the K step after it (the next tile's first one if nk = 1)
    issue_x(0, 0);
    issue_w(1, 0);
    issue_x(1, 0);
    advance_x();
    issue_w(0, 0);
    issue_x(0, 1);
    issue_w(1, 1);
    issue_x(1, 1);
    advance_x();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int g = 0;                                               // K steps consumed so far: buffer g & 1
    for (int ct = slot; ct < ntx; ct += nslots) {
        if (wr == 1) __builtin_amdgcn_s_barrier();           // the second wave row runs one barrier behind the first
        kstep(g & 1, std::true_type{});
        ++g;
        for (int k = 1; k < nk; ++k, ++g) kstep(g & 1, std::false_type{});
        if (wr == 0) __builtin_amdgcn_s_barrier();           // the first wave row waits for the second one's last phase: both write out together
        __builtin_amdgcn_sched_barrier(0);
        epilogue(rb_lo + ct / nct, (ct % nct) * BN);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the zero-record units behind the last tile: nothing may land after the workgroup
}

}  // namespace

// Which layers take the 256-row deep-pipelined tiles: linears and stride-1 / stride-2 3x3 convs without the nearest resize, channel
// counts in whole 64-deep K steps, output width a multiple of 320 (256 x 320 tiles) or -- GEGLU, VAE -- of 256.  Same-box A/B over
// every GEMM shape of a B = 32 UNet step (tools/shape_ab.py, profiles/r03_shape_ab_*.log): 3x3 convs -14..-32 % (1300-1570
// TFLOP/s against 1060-1250), linears with K >= 960 -5..-42 %, the K = 320 / 640 projections WITHOUT a residual -11..-23 %; the ones with
// a residual to read at K <= 640 were 5-23 % faster on bgemm.hip's persistent kernel until the persistent form below got its
// second epilogue (now -3..-10 % here); launches of fewer than E2V_BGEMM_T256_MINTILES tiles stay on the 128-row grid (the
// cross-attention K / V projections: 80 tiles of 256 x 320 leave two thirds of the chip idle).  E2V_BGEMM_T256 = 0: off (A/B).
// (v_mfma_f32_16x16x32_bf16 here, 32x32x16 there: both walk k in the same order and -- measured on every shape of
// tests/test_hip_ops.py::test_bf16_t256_linear -- agree bit for bit, so the choice of kernel does not change a result.)
static int t256_tile_cols(const IgemmArgs& a, const bool force = false) {      // force: the sub-pixel convs (their arithmetic differs from the
                                                                              // gather path's, so the choice must not depend on M or on a switch of the 3x3 path)
    static const int* const on = knob("E2V_BGEMM_T256", 1);
    static const int* const mink = E2V_AB_KNOB("E2V_BGEMM_T256_MINK", 320);
    static const int* const mintiles = E2V_AB_KNOB("E2V_BGEMM_T256_MINTILES", 160);
    if ((!*on && !force) || !a.a_bf16 || a.batch != 1 || a.upsample || (a.taps != 1 && a.taps != 9 && !(a.taps == 4 && a.kw == 2))) return 0;
    if (a.taps != 4 && (a.kw != 3 || a.pad_x >= 0 || a.osy)) return 0;        // (kernel width / split pad / scatter: the sub-pixel kernels only)
    const int Kc = a.c0 + a.c1;
    if (a.c0 <= 0 || a.c0 % 64 || a.c1 % 64 || (!force && a.taps * Kc < *mink)) return 0;
    int cols = 0;
    if (a.geglu) cols = a.N % 256 == 0 ? 256 : 0;
    else if ((a.N | a.ldc | a.ldr) & 7 || (a.rb_ld & 3)) cols = 0;
    else if (a.N % 320 == 0) cols = 320;
    else if (a.N % 256 == 0) cols = 256;
    if (!cols) return 0;
    if (a.taps != 1) {          // a tile's rows span at most ceil(256 / (Ho Wo)) + 1 images: their pixel indices must fit 20 bits
        const long span = (255 / ((long)a.Ho * a.Wo) + 2) * (long)a.Hs * a.Ws;
        if (span >= (1L << 20)) return 0;
        // ... and the gather's 32-bit BYTE offset (pixel x row pitch) must stay below the out-of-window marker (0x80000000) and inside
        // the descriptor's 0x7FFFFFF0-byte record: a wide source (lda * 2 bytes per pixel) on a large map would wrap or read zeros
        const long ldmax = a.lda0 > a.lda1 ? a.lda0 : a.lda1;
        if (span * ldmax * 2 >= (1L << 31) - 16) return 0;
    }
    if (!force && *on != 2 && (long)((a.M + 255) / 256) * (a.N / cols) < *mintiles) return 0;
    return cols;
}

// One launch of `nbm` row blocks from row block rb0 on (whole width unless a column tiling is given): the grid and the kernel instance
static void t256_launch_part(IgemmArgs a, const int cols, const int rb0, const int nbm, const int col_off, const int col_stride, const int nct_l,
                             const int nt, const bool allow_persistent, hipStream_t s) {
    a.nbm = nbm; a.rb0 = rb0; a.col_off = col_off; a.col_stride = col_stride; a.nct_l = nct_l;
    int per_xcd = 0;
    for (int x = 0; x < 8; ++x) {
        const int nrb = (int)(((long)(x + 1) * a.nbm) >> 3) - (int)(((long)x * a.nbm) >> 3);
        per_xcd = nrb > per_xcd ? nrb : per_xcd;
    }
    const int nct = nct_l ? nct_l : a.N / cols;
    const int grid = 8 * per_xcd * nct * (nt == 0 ? 2 : 1);
    const size_t smem = (size_t)2 * (256 + 64 * (nt == 0 ? 3 : nt)) * 128;
    const bool lin = a.taps == 1;
    if (nt == 0) a.col_off2 = col_off + 192;
    h16_dispatch(a.a_bf16, [&](auto h16_tag) {          // the launch's 16-bit type: bf16 / fp16 instances of the same kernels
        using H = decltype(h16_tag);
        auto go = [&](auto kern) {
            E2V_KATTR(kern, ((size_t)2 * (256 + 320) * 128));
            E2V_KLAUNCH(kern, dim3(grid), dim3(512), smem, s, a);
        };
        // linears: the persistent form (E2V_BGEMM_T256P: 0 never, 2 every linear, 1 launches of at least E2V_BGEMM_T256P_MINTILES tiles).
        // Same-box A/B over the linears of a B = 32 UNet step (tools/shape_ab.py): with the first register epilogue (8-byte stores, the
        // residual fetched one 16-row tile at a time) only the GEGLU projections and the K >= 1280 ones with a residual gained
        // (profiles/r03_shape_ab_t256p.log); with 16-byte paired stores and the residual of a half requested at once
        // (profiles/r03_shape_ab_t256p_epilogue.log) every linear of >= 256 tiles does: 640 -> 640 with a residual -10 %, 320 -> 960 -6 %,
        // 320 -> 320 -3 % against bgemm.hip's persistent 128-row kernel, which the K <= 640 projections with a residual used to stay on.
        static const int* const persp = knob("E2V_BGEMM_T256P", 1);
        static const int* const pmaxk = E2V_AB_KNOB("E2V_BGEMM_T256P_MAXK", 1 << 30);
        static const int* const pmint = E2V_AB_KNOB("E2V_BGEMM_T256P_MINTILES", 256);
        const long ntiles = (long)a.nbm * nct;
        // (the epilogue reads the residual in the output's type: fp32 with fp32 at the test entry points, bf16 with bf16 in the graph)
        const bool io_ok = !a.resid || (a.out_f32 ? !a.resid_bf16 : a.resid_bf16 != 0);
        // (the persistent epilogue adds no per-sample row bias: such a launch -- none in the graph today -- stays on the tile kernel)
        if (allow_persistent && lin && io_ok && !a.rowbias && !a.rbsum && *persp && (*persp == 2 || (a.c0 + a.c1 <= *pmaxk && ntiles >= *pmint))) {
            // bias in LDS where it fits behind the operand buffers (160 KB per CU) and the epilogue loads nothing else (no residual)
            static const int* const bldsp = knob("E2V_BGEMM_T256P_BIAS_LDS", 1);
            const bool blds = *bldsp && !a.resid && smem + (size_t)a.N * 4 <= (size_t)160 * 1024;      // (no bias: the LDS copy is zeros)
            auto gop = [&](auto kern) {
                E2V_KATTR(kern, 160 * 1024);
                const int per = per_xcd * nct;                   // tiles of the largest XCD share
                const int slots = per < 32 ? per : 32;
                dry_tag(std::string(" -> bgemm_t256p_kernel 256x") + std::to_string(cols) + (blds ? " bias-lds" : ""));
                E2V_KLAUNCH(kern, dim3(8 * slots), dim3(512), smem + (blds ? (size_t)a.N * 4 : (size_t)0), s, a);
            };
            const bool f32io = a.out_f32 != 0;
            auto pick = [&](auto ntc, auto gc) {
                constexpr int NTc = decltype(ntc)::value;
                constexpr bool Gc = decltype(gc)::value;
                if (f32io) { if (blds) gop(bgemm_t256p_kernel<H, NTc, Gc, true, true>); else gop(bgemm_t256p_kernel<H, NTc, Gc, true, false>); }
                else       { if (blds) gop(bgemm_t256p_kernel<H, NTc, Gc, false, true>); else gop(bgemm_t256p_kernel<H, NTc, Gc, false, false>); }
            };
            if (cols == 320) pick(std::integral_constant<int, 5>{}, std::false_type{});
            else if (a.geglu) pick(std::integral_constant<int, 4>{}, std::true_type{});
            else pick(std::integral_constant<int, 4>{}, std::false_type{});
            return;
        }
        dry_tag(nt == 0 ? std::string(" -> bgemm_t256_tail_kernel 256x192+256x128") : std::string(" -> bgemm_t256_kernel 256x") + std::to_string(64 * nt));
        switch (nt) {
#ifdef E2V_AB              // (the instances that leave row-block sums: measured and not adopted)
            case 5: if (lin) go(bgemm_t256_kernel<H, 5, true>); else if (a.rbsum) go(bgemm_t256_kernel<H, 5, false, true>); else go(bgemm_t256_kernel<H, 5, false>); break;
            case 4: if (lin) go(bgemm_t256_kernel<H, 4, true>); else if (a.rbsum) go(bgemm_t256_kernel<H, 4, false, true>); else go(bgemm_t256_kernel<H, 4, false>); break;
#else
            case 5: if (lin) go(bgemm_t256_kernel<H, 5, true>); else go(bgemm_t256_kernel<H, 5, false>); break;
            case 4: if (lin) go(bgemm_t256_kernel<H, 4, true>); else go(bgemm_t256_kernel<H, 4, false>); break;
#endif
#ifdef E2V_AB
            default: if (lin) go(bgemm_t256_tail_kernel<H, true>); else go(bgemm_t256_tail_kernel<H, false>); break;
#else
            default: throw Error(E2V_EINVAL, "bgemm_t256: no kernel instance for this tile width");
#endif
        }
    });
}

// Will bgemm_t256_launch(a) leave the row-block sums?  Only the staged epilogue does: bf16 output, no GEGLU, no scatter, whole 64-row
// blocks, and the launch must be one this file takes at all.
bool bgemm_t256_writes_rbsum(const IgemmArgs& a) {
#ifndef E2V_AB
    if (a.rbsum) return false;       // (the shipped build has no instance that writes them)
#endif
    return a.rbsum != nullptr && a.taps == 9 && a.a_bf16 && !a.out_f32 && !a.geglu && !a.osy && a.M % 64 == 0 && t256_tile_cols(a, false) != 0;
}

bool bgemm_t256_launch(const IgemmArgs& a_in, hipStream_t s) {
    const int cols = t256_tile_cols(a_in, a_in.osy != 0);
    if (!cols) return false;
    const IgemmArgs& a = a_in;
    const int nbm = (a.M + 255) / 256;
    const int nct = a.N / cols;
    const bool lin = a.taps == 1;
    const double K = (double)a.taps * (a.c0 + a.c1);
    const double rows_in = lin ? (double)a.M : (double)a.M * a.Hs * a.Ws / ((double)a.Ho * a.Wo);
    std::string pname = a.a_bf16 == H16_FP16 ? "igemm_fp16" : "igemm_bf16";
    if (prof_detail())
        pname += " M" + std::to_string(a.M) + " N" + std::to_string(a.N) + " K" + std::to_string((long)K) + " t" + std::to_string(a.taps) +
                 (a.stride > 1 ? " s2" : "") + (a.c1 ? " cat" : "") + (a.geglu ? " geglu" : "") + " T" + std::to_string(cols);
    const double out_b = a.out_f32 ? 4.0 : 2.0;
    ProfScope ps(pname.c_str(), 2.0 * a.M * a.N * K,
                 2.0 * rows_in * (a.c0 + a.c1) + 2.0 * a.N * K + out_b * a.M * (a.geglu ? a.N / 2 : a.N), s);
    // Tail split (E2V_BGEMM_T256_TAIL = 1; OFF by default).  256 CUs take whole tiles: a launch of 3.375 rounds (level 2 at B = 32: 216
    // row blocks x 4 column tiles) runs four, the last one three eighths full.  With the switch on, when the last round would be at
    // most half full its row blocks go into a second launch as 256 x 192 and 256 x 128 tiles (the same kernel body with NT = 3 / 2:
    // two tiles per 320 columns, twice as many workgroups).  Same k order per output: bit-identical
    // (tests/test_hip_ops.py::test_bf16_t256_tail_split).  MEASURED (same-process A/B over a B = 32 UNet step): the level-2 convs -3 %
    // (not the -9 % of the round count: the narrow tiles' phases are 8-12 MFMAs long and run at half the efficiency), 320 -> 320 at
    // level 0 +3 % (its persistent launch loses 128 tiles to a non-persistent one), 198.6 -> 198.4 ms per step: not worth a default.
    static const int* const tailp = E2V_AB_KNOB("E2V_BGEMM_T256_TAIL", 0);
    const long T = (long)nbm * nct, full = T / 256, tail = T - 256 * full;
    if (*tailp && cols == 320 && !a.geglu && !a.rbsum && full >= 1 && tail > 0 && 2 * tail <= 256 && (256 * full) % nct == 0) {
        const int r1 = (int)(256 * full / nct);
        t256_launch_part(a, cols, 0, r1, 0, 0, 0, 5, true, s);
        t256_launch_part(a, cols, r1, nbm - r1, 0, 320, nct, 0, false, s);      // nt = 0: the tail kernel (192- and 128-column tiles)
        return true;
    }
    t256_launch_part(a, cols, 0, nbm, 0, 0, 0, cols / 64, true, s);
    return true;
}

// ---- sub-pixel form of `Upsample3D` (resnet.py:30-62: F.interpolate(scale_factor = [1, 2, 2], mode = "nearest") + 3x3 conv) -----------
// With an exact 2x nearest resize, output pixel (2y + a, 2x + b) reads source rows {y - 1, y} (a = 0) or {y, y + 1} (a = 1) and
// likewise columns: taps ky = {0 | 1, 2} (a = 0) or {0, 1 | 2} (a = 1) fall on the same source pixel.  So the layer is four 2x2
// convs on the SOURCE map, one per output parity (a, b), whose weights are the sums over those tap sets (summed in fp32, rounded to
// bf16 once), with top / left pad 1 - a / 1 - b and zero padding of the source map exactly where the resized map was padded;
// each writes the output pixels of its parity.  4 / 9 of the multiplies, no resize arithmetic in the gather.
__global__ void pack_conv_up2x_kernel(const float* __restrict__ w, float* __restrict__ o, int cout, int cin) {
    const int nq = (cin + 63) / 64;
    const size_t per = (size_t)nq * 4 * 64;
    const size_t total = 4 * (size_t)cout * per;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int par = (int)(i / ((size_t)cout * per));
        const size_t r0 = i - (size_t)par * cout * per;
        const int oc = (int)(r0 / per);
        const size_t r = r0 - (size_t)oc * per;
        const int e = (int)(r % 64), t = (int)((r / 64) % 4), q = (int)(r / 256);
        const int c = q * 64 + e;
        const int a = par >> 1, b = par & 1, ty = t >> 1, tx = t & 1;
        // tap sets: parity 0: {0} | {1, 2}; parity 1: {0, 1} | {2}
        const int ky0 = a == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = a == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
        const int kx0 = b == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = b == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
        float sum = 0.f;
        if (c < cin) {
            const float* wp = w + ((size_t)oc * cin + c) * 9;
            for (int ky = ky0; ky <= ky1; ++ky)
                for (int kx = kx0; kx <= kx1; ++kx) sum += wp[ky * 3 + kx];
        }
        o[i] = sum;
    }
}
int conv_up2x_packed_ld(int cin) { return (cin + 63) / 64 * 4 * 64; }
void pack_conv_up2x(const float* w, float* o, int cout, int cin, hipStream_t s) {
    const size_t total = 4 * (size_t)cout * conv_up2x_packed_ld(cin);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    E2V_KLAUNCH(pack_conv_up2x_kernel, dim3(blocks), dim3(256), 0, s, w, o, cout, cin);
}

static IgemmArgs up2x_parity_args(const IgemmArgs& g, const int a, const int b) {
    IgemmArgs q = g;
    q.upsample = 0; q.ups_h = q.ups_w = 1.f;
    q.taps = 4; q.kw = 2; q.pad = 1 - a; q.pad_x = 1 - b;
    q.Hi = g.Hs; q.Wi = g.Ws; q.Ho = g.Hs; q.Wo = g.Ws;
    q.M = g.M / 4;
    q.osy = 2; q.osx = 2; q.ooy = a; q.oox = b;
    q.ldw16 = conv_up2x_packed_ld(g.c0);
    q.ldw = q.ldw16;
    return q;
}

bool bgemm_up2x_applies(const IgemmArgs& g) {
    static const int* const on = knob("E2V_BGEMM_UP2X", 1);
    if (!*on || !g.a_bf16 || !g.upsample || g.taps != 9 || g.stride != 1 || g.pad != 1 || g.c1 != 0 || g.batch != 1) return false;
    if (g.Hi != 2 * g.Hs || g.Wi != 2 * g.Ws || g.Ho != g.Hi || g.Wo != g.Wi || g.resid || g.rowbias || g.geglu) return false;
    if (g.M != (g.M / (g.Ho * g.Wo)) * g.Ho * g.Wo) return false;
    return t256_tile_cols(up2x_parity_args(g, 0, 0), true) != 0;
}

void bgemm_up2x_launch(const IgemmArgs& g, const void* w16_up2, hipStream_t s) {
    // this route does not pass igemm()'s operand checks: the LDS-DMA moves 16-byte pieces
    if ((g.lda0 & 7) || (g.ldc & 7) || (reinterpret_cast<uintptr_t>(g.a0) & 15) || (reinterpret_cast<uintptr_t>(g.out) & 15))
        throw Error(E2V_ESHAPE, "bf16 sub-pixel conv: row strides must be multiples of 8 elements and rows 16-byte aligned");
    const size_t per = (size_t)g.N * conv_up2x_packed_ld(g.c0);
    for (int par = 0; par < 4; ++par) {
        IgemmArgs q = up2x_parity_args(g, par >> 1, par & 1);
        q.w16 = static_cast<const char*>(w16_up2) + (size_t)par * per * 2;
        q.w = nullptr;
        if (!bgemm_t256_launch(q, s)) throw Error(E2V_EINVAL, "bgemm_up2x_launch: the layer is not eligible (bgemm_up2x_applies first)");
    }
}

}  // namespace e2v
