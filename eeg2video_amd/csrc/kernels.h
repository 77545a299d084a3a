// Launchers of the hand-written gfx950 kernels of the EEG2Video generation hot path.
// All activations are CHANNEL-LAST fp32: a tensor [n, F, H, W, C] is a row-major matrix
// [rows = n*F*H*W][C]; every kernel takes explicit row strides so that slices of a wider
// buffer (fused QKV output, concatenated skip) are read in place.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>

namespace e2v {

// ---------------------------------------------------------------------------------------
// implicit-GEMM convolution / linear  (igemm.hip)
//   out[m][n] = epilogue( alpha * sum_{tap, c} A[src(m, tap)][c] * W[n][tap*Ctot + c] )
// A comes from up to two channel-last sources (the reference's torch.cat([h, skip], dim=1),
// unet_blocks.py:487,570, is never materialised); src() folds zero padding, stride and the
// nearest-neighbour resize of Upsample3D (resnet.py:58-61) into the gather.
// ---------------------------------------------------------------------------------------
struct IgemmArgs {
    const float* a0 = nullptr; const float* a1 = nullptr;
    int c0 = 0, c1 = 0;            // channels taken from a0 / a1 (multiples of 4)
    int lda0 = 0, lda1 = 0;        // row strides (floats)
    const float* w = nullptr;      // linear: [N][K]; 3x3: [N][chunk of 32][tap][32] (pack_conv3x3)
    int ldw = 0;
    int ldw16 = 0;                 // row length of w16 (3x3: chunks of 64)
    float* out = nullptr; int ldc = 0;
    const float* bias = nullptr;       // [N]
    const float* rowbias = nullptr;    // [samples][rb_ld] (time embedding added after conv1, resnet.py:186)
    int rb_ld = 0; int rows_per_sample = 1;
    const float* resid = nullptr; int ldr = 0;
    int M = 0, N = 0;
    int taps = 1;                  // 1 (linear / 1x1) or 9 (3x3)
    int Ho = 1, Wo = 1;            // output map
    int Hi = 1, Wi = 1;            // logical input map (after nearest resize)
    int Hs = 1, Ws = 1;            // physical source map
    int stride = 1, pad = 0;       // pad = rows/cols of zeros above/left (below/right is implied by Hi/Wi)
    // bgemm256.hip only (the sub-pixel form of a 2x nearest resize + 3x3 conv, bgemm_up2x): kernel width (taps = kh x kw, tap t =
    // (t / kw, t % kw)), the left pad if it differs from the top pad `pad` (-1: the same), and an output scatter -- output pixel
    // (oy, ox) of image i is written to row ((i Ho osy + oy osy + ooy) Wo osx + ox osx + oox) instead of row m (osy = 0: off)
    int kw = 3, pad_x = -1;
    int osy = 0, osx = 0, ooy = 0, oox = 0;
    int upsample = 0; float ups_h = 1.f, ups_w = 1.f;
    float alpha = 1.f;
    int relu = 0;                  // out = max(out, 0) after bias (semantic-predictor MLP)
    int geglu = 0;                 // W rows packed [32 value | 32 gate] per 64: out[m][n/2] = v * gelu(g)
    int batch = 1;                 // blockIdx.z; strides in floats
    long long sa0 = 0, sw = 0, sout = 0;
    const void* w16 = nullptr;     // the same packed weights rounded to bf16 (a_bf16 mode)
    int x3 = 0;                    // 1: fp32 products from three bf16 pieces per operand on the bf16 MFMA (igemm_tile_x3); needs w3
    const void* w3 = nullptr;      // three bf16 planes of w ([N][K] each, w3_plane elements apart; batch entries sw apart)
    long long w3_plane = 0;
    // bf16-ACTIVATION mode (bgemm.hip; e2v_set_compute_dtype(E2V_BF16)): a0 / a1 point at bf16 rows (lda in elements), the weights
    // are w16, the product is v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  out is bf16 unless out_f32; resid is bf16 iff
    // resid_bf16 (it is an activation) -- bias and rowbias stay fp32.
    int a_bf16 = 0, out_f32 = 0, resid_bf16 = 0;
    int bm256 = 0;                 // filled by the launcher: 256-row tiles (bgemm256_kernel)
    int ablate = 0;                // timing experiments only (E2V_BGEMM_ABLATE): 1 = no output stores, 2 = A loads read zeros, 3 = both
    int rb1 = 0, w1 = 0, s1 = 0, s2 = 0, nbm = 0, nbm_per = 0, tail_rb = 0;   // filled by the launcher: tile schedule (see igemm_kernel)
    int rb0 = 0, col_off = 0, col_off2 = 0, col_stride = 0, nct_l = 0;   // filled by bgemm256.hip's launcher: first row block / column tiling of a partial launch (tail split)
    // GroupNorm statistics from the PRODUCER (bf16 mode): rbsum != null asks the launch to leave, beside the tensor, the sums a
    // following GroupNorm needs -- rbsum[m / 64][n][2] = (sum, sum of squares) of the STORED (bf16-rounded) outputs of rows
    // 64 b .. 64 b + 63, column n, in the canonical order of rowblock_sums (norm.hip).  Only the staged epilogue of bgemm_t256_kernel
    // writes it (igemm_writes_rbsum says whether a launch will); anything else leaves it untouched and the caller runs
    // rowblock_sums over the tensor instead -- same sums, bit for bit, so that which kernel serves a layer (a function of the
    // batch) never shows in a result.
    float* rbsum = nullptr;
    // Split-K (bgemm.hip: bgemm_splitk_kernel + splitk_reduce_kernel; 16-bit modes, the SMALL-BATCH dispatch family -- the reference
    // generates clip by clip, inference_eeg2video.py:90-100: two UNet samples leave a deep-level 3x3 conv 40 tiles for 256 CUs).
    // sk >= 2 and sk_ws != null: the K range is cut into sk runs of whole 64-channel chunks (each inside one source of a concat), run
    // z = blockIdx.y leaves its fp32 partial tile in sk_ws[z][M][N] and a second kernel adds the runs IN ORDER (run 0 first) and applies
    // the epilogue -- deterministic, but a different summation order than the unsplit kernels: results are equal to theirs up to fp32
    // rounding, not bit for bit (DESIGN: bit-identity across batch sizes holds within a dispatch family).  splitk_plan() sizes it.
    int sk = 0;
    float* sk_ws = nullptr;
    int sk_s0 = 0, sk_q0 = 0, sk_q1 = 0;      // filled by the launcher: runs inside source 0, chunks per run in source 0 / source 1
};
// The split a launch of `a` would take for a request of `want` runs: 0 = none (the launch is not eligible), else the number of runs
// actually used (<= want: whole chunks, every run non-empty); the caller then provides sk_ws of that many [M][N] fp32 planes.
int splitk_plan(const IgemmArgs& a, int want);
void igemm(const IgemmArgs& a, hipStream_t s);
bool igemm_writes_rbsum(const IgemmArgs& a);       // will igemm(a) fill a.rbsum?  (same rules as the launch itself)
// rows per lane pass of the canonical order: the staged epilogue of a 256 x 320 tile finishes 6 rows per pass, of a 256 x 256 tile 8
inline int rbsum_rows_per_pass(int N) { return N % 320 == 0 ? 6 : 8; }
// could a tensor [M][N] carry row-block sums at all (a property of the layer, never of the batch)?
inline bool rbsum_capable(long long M, int N) { return M % 64 == 0 && (N % 320 == 0 || N % 256 == 0); }
// Run-time switches: an int per name, initialised from the environment variable of that name on first use
// and settable through e2v_op_set_knob for A/B comparisons inside one process.  The returned pointer stays valid.
int* knob(const char* name, int dflt);
bool set_knob(const char* name, int value);
// Switches of variants that were MEASURED AND NOT ADOPTED (or are kept only as the other arm of a same-process A/B) exist in an
// `make AB=1` build (-DE2V_AB) only: the shipped library carries neither their kernels nor their switches (set_knob refuses the
// names, the environment variables are not read), and the launch rules read the default.
template <int V> inline const int* ab_const() { static const int v = V; return &v; }
#ifdef E2V_AB
#define E2V_AB_KNOB(name, dflt) ::e2v::knob(name, dflt)
#else
#define E2V_AB_KNOB(name, dflt) ::e2v::ab_const<dflt>()
#endif      // false: no kernel has asked for a knob of that name yet and it is not a known one
void bgemm_launch(const IgemmArgs& a, int ntiles, hipStream_t s);
void bgemm_splitk_launch(const IgemmArgs& a, hipStream_t s);            // the split-K form (a.sk, a.sk_ws; schedule filled by igemm())
bool bgemm_all_n64(const IgemmArgs& a);
bool bgemm_t256_writes_rbsum(const IgemmArgs& a);                         // bgemm256.hip: would that launch fill a.rbsum?
bool bgemm_t256_launch(const IgemmArgs& a, hipStream_t s);                 // bgemm256.hip: true = the layer was eligible and has been launched
// Exact 2x nearest resize in front of a stride-1, pad-1 3x3 conv as four 2x2 convs on the SOURCE map (one per output parity; weights
// summed over the taps that read the same source pixel: 4 / 9 of the multiplies).  `g` describes the conv as igemm() takes it.
bool bgemm_up2x_applies(const IgemmArgs& g);
int conv_up2x_packed_ld(int cin);                                        // row length of one parity's [O][chunk64][4 taps][64] layout
void pack_conv_up2x(const float* w_oihw, float* w_packed, int cout, int cin, hipStream_t s);    // [4 parities][O][ld], fp32
void bgemm_up2x_launch(const IgemmArgs& g, const void* w16_up2, hipStream_t s);                // w16_up2: the packed layout as bf16
bool bgemm_use_256(const IgemmArgs& a);                                  // schedule hint: 256-row tiles pay for this launch

// weight re-layout helpers (one-off, at finalize)
// [O][I][3][3] -> [O][ceil(I/bke)][9][bke] (bke = 32 for the fp32 kernel, 64 for the bf16 one); row length below
int conv3x3_packed_ld(int cin, int bke);
void pack_conv3x3(const float* w_oihw, float* w_packed, int cout, int cin, int bke, hipStream_t s);
void copy_rows(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, hipStream_t s);
void to_bf16(const float* in, void* out_bf16, size_t n, hipStream_t s);
void to_h16(const float* in, void* out16, size_t n, int mode, hipStream_t s);      // mode: 1 bf16 / 2 fp16 (h16.h), round to nearest even
void split_bf16x3(const float* in, void* out_planes, size_t n, size_t plane_stride, hipStream_t s);   // exact 3-way truncation split

// ---------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) for the stride-1, pad-1 3x3 convolutions (wino.hip); output map = logical input map
// ---------------------------------------------------------------------------------------
struct WinoArgs {
    const float* x0 = nullptr; const float* x1 = nullptr;      // channel-last sources (x1: concatenated skip)
    int c0 = 0, c1 = 0, ld0 = 0, ld1 = 0;
    int nimg = 0, Hs = 1, Ws = 1;                              // physical source map
    int Ho = 1, Wo = 1;                                        // output map (= source map, or 2x with upsample)
    int upsample = 0; float ups_h = 1.f, ups_w = 1.f;
    const float* gn_scsh = nullptr; int gn_P = 1; int gn_silu = 0;   // fused GroupNorm affine (+ SiLU) on the way in
    int m = 2;                                                 // output tile: 2 = F(2x2,3x3), 4 = F(4x4,3x3)
    const float* U = nullptr;                                  // [(m+2)^2][N][c0+c1] (wino_pack_weights)
    const void* U3 = nullptr;                                  // its three-plane bf16 split (f32x3 mode), planes (m+2)^2*N*C apart
    int N = 0;
    float* out = nullptr; int ldc = 0;
    const float* bias = nullptr;
    const float* rowbias = nullptr; int rb_ld = 0; int rows_per_sample = 1;
    const float* resid = nullptr; int ldr = 0;
};
void wino_pack_weights(const float* w_oihw, float* U, int cout, int cin, int m, hipStream_t s);
size_t wino_workspace_floats(const WinoArgs& a, int nimg);       // V + M for `nimg` images
int wino_chunk_images(const WinoArgs& a, size_t max_floats);     // images per pass so that the workspace fits
void wino_conv3x3(const WinoArgs& a, float* workspace, int chunk_images, hipStream_t s);

// ---------------------------------------------------------------------------------------
// normalisation (norm.hip)
// ---------------------------------------------------------------------------------------
// GroupNorm over `samples` slabs of P rows.  Statistics are taken over (C/groups) channels x P rows
// (5-D GroupNorm of ResnetBlock3D: P = F*H*W, resnet.py:177; per-frame GroupNorm of
// Transformer3DModel: samples = n*F, P = H*W, attention.py:93,99).  Two sources = channel concat.
struct GroupNormArgs {
    const float* x0 = nullptr; const float* x1 = nullptr;
    int c0 = 0, c1 = 0, ld0 = 0, ld1 = 0;
    const float* gamma = nullptr; const float* beta = nullptr;   // [c0+c1]
    float* out = nullptr; int ldo = 0;                           // [samples*P][c0+c1]
    int samples = 0, P = 0, groups = 32;
    float eps = 1e-5f; int silu = 0;
    int bf16 = 0;                  // 1: x0 / x1 / out are bf16 (statistics, scale / shift and the arithmetic stay fp32)
    float* ws_part = nullptr;      // workspace: samples*chunks*(c0+c1)*2 floats
    float* ws_scale = nullptr;     // workspace: samples*(c0+c1)*2 floats
    // row-block sums that came with a source tensor (IgemmArgs::rbsum / rowblock_sums: [rows / 64][c_i][2]): the statistics pass
    // over that source is skipped and the fold reads them (needs P % 64 == 0; bf16 mode)
    const float* rb0 = nullptr; const float* rb1 = nullptr;
    // small-batch dispatch family (16-bit modes): where a (sample, group) slice fits LDS, ONE kernel reads it once, folds and applies
    // (gn_fused_small_kernel) instead of the three launches -- another summation order, so the family picks it, never the batch
    int fused_small = 0;
    int small_chunks = 0;          // the family's 64-row statistics / apply chunks at every level (two samples: 108 chunks of 256 rows for 256 CUs)
};
// the canonical row-block sums of a stored bf16 tensor x[rows][C] (rows % 64 == 0): out[rows / 64][C][2], bit-identical to what the
// staged epilogue of bgemm_t256_kernel leaves for a tensor of the same width (rpp = rbsum_rows_per_pass(C))
void rowblock_sums(const void* x_bf16, int ld, int C, long long rows, int rpp, float* out, hipStream_t s);
int  groupnorm_chunks(int P);
void groupnorm(const GroupNormArgs& a, hipStream_t s);
// statistics only: leaves per-(slab, channel) (scale, shift) pairs in a.ws_scale for a consumer that applies the affine
// (+ SiLU) itself while it reads the tensor anyway (the Winograd input transform)
void groupnorm_stats(const GroupNormArgs& a, hipStream_t s);
void layernorm(const float* x, int ldx, const float* gamma, const float* beta, float* out, int ldo,
               int rows, int C, float eps, hipStream_t s, int bf16 = 0);      // bf16: x / out are bf16 rows

// ---------------------------------------------------------------------------------------
// attention (attn.hip)
// ---------------------------------------------------------------------------------------
struct AttnArgs {
    const float* q = nullptr; int ldq = 0;
    const float* k = nullptr; const float* v = nullptr; int ldkv = 0;
    float* o = nullptr; int ldo = 0;
    int n = 0, F = 1, heads = 8, D = 40;
    int Nq = 0, Nk = 0;
    int mode = 0;      // 0: sparse-causal self-attention, keys = [frame 0 ; frame max(f-1,0)] (attention.py:292-301)
                       // 1: keys shared by all frames of a sample (cross-attention to the 77 cond tokens)
    float scale = 1.f;
    int x3 = 0;        // 1: fp32-equivalent QK^T and PV from exactly split bf16 pieces (six MFMAs per product, f32x3 mode)
    int io_bf16 = 0;   // 1: q / k / v / o are bf16 rows (strides in elements, multiples of 8); implies the bf16 MFMA
};
void flash_attention(const AttnArgs& a, hipStream_t s);      // throws Error(E2V_EINVAL) for a head dim without a kernel instance
bool flash_attention_supports(int D);
// attn_q64.hip: bf16 sparse-causal attention with 64 queries per wave (head dims 40 / 80).  _waves: waves per workgroup of the instance
// that would serve the call, 0 = not served (flash_attention falls back to flash_attn_b16io_kernel); the second launches it.
std::string attn_shape_tag(const AttnArgs& a);                // " D.. Nq.. Nk.. n.. F.. h.." for profile names
int flash_attention_q64_waves(const AttnArgs& a);
bool flash_attention_q64(const AttnArgs& a, hipStream_t s);
// temporal self-attention over the F frames of every pixel (attention.py:261-267), qkv = [n*F*HW][3C]
void temporal_attention(const float* qkv, int ld, float* out, int ldo, int n, int F, int HW, int heads, int D,
                        float scale, hipStream_t s, int bf16 = 0);          // bf16: qkv / out are bf16 rows
// in place (VAE attention); out_bf16 != null: the normalised probabilities are written there as bf16 ([rows][ld]) instead
void softmax_rows(float* x, int ld, int rows, int cols, hipStream_t s, void* out_bf16 = nullptr, int out_mode = 1);   // out_mode: 1 bf16 / 2 fp16

// ---------------------------------------------------------------------------------------
// element-wise / layout (misc.hip)
// ---------------------------------------------------------------------------------------
void ncfhw_to_cl(const float* in, float* out, int n, int C, int Cpad, int FHW, float scale, hipStream_t s, int out_bf16 = 0);
void cl_to_ncfhw(const float* in, int ld, float* out, int n, int C, int FHW, float mul, float add, int clamp, float lo,
                 float hi, hipStream_t s);
void nchw_frames_to_ncfhw(const float* in, int ld, float* out, int n, int F, int C, int HW, float mul, float add,
                          int clamp01, hipStream_t s);
void timestep_sinusoid(const long long* t, int nt, float* out, int n, int dim, int flip_sin_to_cos, float freq_shift,
                       hipStream_t s, int t_is_f32 = 0);      // t_is_f32: `t` points at nt floats instead
void silu(const float* in, float* out, long long count, hipStream_t s);
void transpose2d(const float* in, int ld_in, float* out, int ld_out, int rows, int cols, int batch,
                 long long sb_in, long long sb_out, hipStream_t s, int bf16 = 0);
// eps = eps_u + g (eps_c - eps_u); DDIM eta = 0 update (pipeline_tuneeeg2video.py:320-325)
void ddim_cfg_step(const float* eps_u, const float* eps_c, const float* x, float* x_out, long long count,
                   float guidance, float sqrt_a_t, float sqrt_1m_a_t, float sqrt_a_p, float sqrt_1m_a_p,
                   hipStream_t s);

// out = sum_{i<n} coefs[i] xs[i] (n <= 5) / out = eu + g (ec - eu): schedulers other than DDIM (PNDM) and the guidance line
void lincomb(int n, const float* const* xs, const float* coefs, float* out, long long count, hipStream_t s);
void cfg_combine(const float* eu, const float* ec, float g, float* out, long long count, hipStream_t s);

// (f) rows -------------------------------------------------------------------------------------------
// DANA noise (EEG2Video/models/DANA_module.py:52-72) fused with the caller's layout fix 'a b c d e -> a c b d e'
// (inference_eeg2video.py:77,82): x0, eps_div [B,F,C,HW], eps_same [B,1,C,HW] -> out [B,C,F,HW];
// coef [B][2] = (sqrt(abar_t), sqrt(1 - abar_t)) per clip (device).
void dana_noise(const float* x0, const float* eps_div, const float* eps_same, const float* coef, float sqrt_1m_beta,
                float sqrt_beta, float* out, int B, int F, int C, int HW, hipStream_t s);
// (x * 255) truncated to uint8, as save_videos_grid does (tuneavideo/util.py:29)
void frames_to_u8(const float* in, unsigned char* out, long long count, hipStream_t s);
void pad_cols(const float* in, int cols, float* out, int cols_pad, long long rows, hipStream_t s, int out_bf16 = 0);   // fp32 in; out fp32 or bf16

// weight-streaming GEMV (gemv.hip): out[b][n] = act(x[b] . W[n] + bias[n]) for B <= 16 rows, x / out fp32 rows, W [N][ldw] fp32 or
// bf16 (rows zero-padded to ldw); the Semantic Predictor at the reference's batch sizes
bool gemv_rows_supported(int B, int K, int w_bf16);
void gemv_rows(const float* x, int ldx, const void* w, int ldw, int w_bf16, const float* bias, float* out, int ldo, int B, int N, int K,
               int relu, hipStream_t s);

// strided row copy between storage types (fp32 <-> bf16), zero-filling columns cols .. cols_out-1 of the output
void cvt_rows(const void* in, int ld_in, int in_bf16, void* out, int ld_out, int out_bf16, long long rows, int cols, int cols_out,
              hipStream_t s);

}  // namespace e2v
