// The model graphs of the hot path, run as sequences of gfx950 kernel launches on one stream:
// UNet3DConditionModel.forward (unet.py:278-413), AutoencoderKL encode/decode (diffusers 0.11.1,
// SURVEY App. C.5) and the DDIM denoising loop of TuneAVideoPipeline.__call__
// (pipeline_tuneeeg2video.py:287-334).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"
#include "runtime.h"

namespace e2v {

struct NormW { const float* g = nullptr; const float* b = nullptr; int c = 0; };
struct LinW { const float* w = nullptr; const float* b = nullptr; int in = 0, out = 0; const void* w16 = nullptr;
              const void* w3 = nullptr;         // three bf16 planes of w (out*in elements apart) for the f32x3 mode
              int in16 = 0;                     // row length of w16: `in` rounded up to 8 (zero columns; 16-byte DMA pieces)
              int part = 0;                     // which finalize() group owns the lazily built fp16 form
              mutable const void* w16h = nullptr; };   // the same matrix as IEEE half (fp16 mode), built on first use (e2v_ctx::lin_f16)
// A 3x3 conv keeps its torch-layout weight and builds the kernel layouts ON FIRST USE (e2v_ctx::conv_form): which of them a
// layer ever needs depends on the arithmetic mode and on the map size it is called with -- fp32 direct-packed, F(2x2) / F(4x4)
// Winograd-domain (x2.25 / x4 the raw size), bf16 direct-packed -- and building all of them cost 18 GiB where one mode uses 4-9.
struct ConvW {
    const float* raw = nullptr;                  // [O][I][3][3] fp32, as uploaded
    const float* b = nullptr;
    int cin = 0, cin_pad = 0, cout = 0;
    int ldw = 0, ldw16 = 0;                      // row lengths of the fp32 (32-channel chunks) / bf16 (64-channel chunks) packed forms
    int cin_pad16 = 0;                           // channels of the bf16 input: cin rounded up to 8
    bool stride1 = true;                         // a Winograd form exists (stride-1 conv, channel counts multiples of 4)
    int part = 0;                                // 0 UNet, 1 VAE: which finalize() group owns the lazily built forms
    mutable const float* w = nullptr;            // [O][chunk32][tap][32]
    mutable const void* w16 = nullptr;           // [O][chunk64][tap][64] bf16
    mutable const void* w16_up2 = nullptr;       // [4 parities][O][chunk64][4 taps][64] bf16: the sub-pixel form of resize + conv (bgemm_up2x)
    mutable const void* w16h = nullptr;          // the two 16-bit layouts as IEEE half (fp16 mode)
    mutable const void* w16h_up2 = nullptr;
    mutable const float* wino = nullptr;         // [16][O][I]
    mutable const float* wino4 = nullptr;        // [36][O][I]
    mutable const void* wino_x3 = nullptr; mutable const void* wino4_x3 = nullptr;   // three-plane bf16 splits (f32x3 mode)
};

struct ResW {
    NormW n1, n2;
    ConvW c1, c2;
    LinW temb;        // w == nullptr: no time embedding (VAE)
    LinW sc;          // w == nullptr: identity shortcut
    int cin = 0, cout = 0;
};

struct TransW {
    NormW norm, ln1, ln2, ln3, lnt;
    LinW proj_in, proj_out;
    LinW a1_qkv, a1_out;            // sparse-causal self-attention, fused [3C][C]
    LinW a2_q, a2_kv, a2_out;       // cross-attention, fused [2C][cross]
    LinW ff1, ff2;                  // ff1 rows interleaved [32 value | 32 gate] for the GEGLU epilogue
    LinW at_qkv, at_out;            // temporal attention
    int C = 0;
};

struct UNetW {
    ConvW conv_in, conv_out;
    LinW te1, te2;
    NormW norm_out;
    struct Block {
        std::vector<ResW> res;
        std::vector<TransW> attn;       // empty for DownBlock3D / UpBlock3D
        bool resample = false;
        ConvW rs;                       // downsampler / upsampler conv
    };
    std::vector<Block> down, up;
    ResW mid_r0, mid_r1;
    TransW mid_attn;
};

struct VAEAttnW { NormW norm; LinW qkv, proj; int C = 0; };
struct VAEW {
    struct Block { std::vector<ResW> res; bool resample = false; ConvW rs; };
    // decoder
    LinW post_quant;
    ConvW dec_in, dec_out;
    ResW dec_mid0, dec_mid1;
    VAEAttnW dec_attn;
    std::vector<Block> dec_up;
    NormW dec_norm_out;
    // encoder
    LinW quant;
    ConvW enc_in, enc_out;
    ResW enc_mid0, enc_mid1;
    VAEAttnW enc_attn;
    std::vector<Block> enc_down;
    NormW enc_norm_out;
};

}  // namespace e2v

struct e2v_ctx {
    e2v_config cfg;
    int device = 0;
    std::string err;
    std::vector<std::string> keys;                               // expected keys, in state-dict order
    std::unordered_map<std::string, e2v::WTensor> raw;           // uploaded tensors (torch layout)
    std::vector<float*> owned;                                   // packed weights + misc device blocks
    size_t weight_bytes = 0;
    e2v::Pool pool;
    e2v::UNetW unet;
    e2v::VAEW vae;
    bool unet_ready = false, vae_ready = false, sem_ready = false;
    int conv_algo = 0;                                           // e2v_set_conv_algo: 0 auto, 1 direct, 2 / 3 Winograd F(2x2) / F(4x4) wherever it applies
    bool wino_f4 = true;                                         // auto: F(4x4,3x3) where min(Cin, Cout) >= wino4_min_c and the map, padded to
    int wino4_min_c = 128;                                       //   multiples of 4, grows by <= wino_f4_pad (E2V_WINO_F4 / _F4_MIN_C / _F4_PAD)
    double wino_f4_pad = 1.7;
    int wino_min_c = 256;                                        // auto: F(2x2,3x3) when min(Cin, Cout) >= this (E2V_WINO_MIN_C)
    size_t wino_ws_floats = (size_t)1 << 30;                     // workspace cap per pass (E2V_WINO_WS_MB)
    bool x3_compute = false;                                     // E2V_F32X3: fp32 products from split bf16 pieces on the bf16 MFMA
    bool bf16_compute = false;                                   // e2v_set_compute_dtype: a 16-bit activation mode (16-bit MFMA, 16-bit rows in HBM) ...
    int h16_mode = 0;                                            // ... and which: 0 none, 1 bf16 (E2V_BF16), 2 fp16 (E2V_F16) -- the flag every 16-bit launch carries (h16.h)
    void set_h16_mode(int m) { h16_mode = m; bf16_compute = m != 0; }
    // The SMALL-BATCH dispatch family (16-bit modes): set per call when the graph runs at most 8 UNet samples (B <= 4 clips with their
    // guidance pairs: the reference's clip-by-clip loop, inference_eeg2video.py:90-100, and a caller that batches a few clips; the boundary
    // was measured -- B = 3 / 4 gain 7 / 4 %, B = 5 / 6 / 8 2.4 / 1.7 / 0.5 %: profiles/r05_family_boundary.json).  Launches that would leave most of the chip idle
    // then take split-K (kernels.h: IgemmArgs::sk) -- another summation order, so bit-identity across batch sizes holds WITHIN a family
    // (B >= 5: every kernel choice is a bit-identical alternative; B <= 4: held to the oracle bounds).
    bool small_family = false;
    std::vector<e2v::LinW> sem;                                  // semantic predictor layers (first one K-padded to 4)
    int sem_in_pad = 0;
    std::vector<float> alphas;                                   // host alpha-bar table
    float* gn_part = nullptr; size_t gn_part_floats = 0;         // GroupNorm workspaces (grown on demand)
    float* gn_scale = nullptr; size_t gn_scale_floats = 0;
    long long* d_timesteps = nullptr; int d_timesteps_cap = 0;
    // per-generate caches of what does not depend on the latents: time_emb_proj(SiLU(emb(t_i))) of every resnet and step
    // ([steps][cout] each, call order) and to_k / to_v of the conditioning of every transformer ([N*T][2C] each)
    bool step_cache_on = false; int step_cache_step = 0;
    std::vector<e2v::Act> temb_cache, kv_cache;
    void build_step_caches(const int64_t* ts, int steps, const float* cond, int N, int T, hipStream_t s);

    // stream of the previous call (workspace reuse is stream-ordered): see enter_stream
    hipStream_t last_stream = nullptr; bool has_last_stream = false; hipEvent_t stream_ev = nullptr;
    void enter_stream(hipStream_t s);
    // device blocks made by finalize(), per part (bit index of `which`), so that finalizing a part again frees what it replaces
    std::vector<float*> owned_part[3];
    std::unordered_map<void*, size_t> owned_bytes;
    void* comm = nullptr; int comm_rank = 0, comm_world = 0;     // RCCL communicator of e2v_comm_init (comm.cpp)
    // e2v_op_unet_forward_taps (test aid): while set, unet_forward_cl copies the tensors the oracle exposes (emb, down0..3, mid,
    // up0..3) out as fp32 NCFHW, back to back; shapes = {n, C, F, H, W} per tap
    struct TapSink { float* buf = nullptr; int64_t cap = 0, used = 0; int count = 0; int64_t shapes[16][5]; };
    TapSink* tap_sink = nullptr;
    int alloc_part = -1;                                         // >= 0: dev_alloc files the block under owned_part[alloc_part]
    void free_part(int part);

    float* dev_alloc(size_t floats);
    enum ConvForm { FORM_DIRECT32, FORM_BF16, FORM_WINO2, FORM_WINO4, FORM_BF16_UP2, FORM_F16, FORM_F16_UP2 };
    const void* lin_f16(const e2v::LinW& w, hipStream_t s);              // the fp16 copy of a linear's weight (built on first use)
    void conv_form(const e2v::ConvW& w, ConvForm f, hipStream_t s);     // build the layout if this is its first use
    bool conv_has_wino(const e2v::ConvW& w, int m) const;               // would the policy allow the F(m x m) form for this layer?
    void expected_keys();
    void finalize(int which);
    // graphs (channel-last in/out); see model.cpp
    // cfg_pair: sample_cl holds N / 2 samples standing for the [x ; x] input of a classifier-free-guidance step
    // host_tf != null: fractional timesteps (fp32) instead of host_t
    e2v::Act unet_forward_cl(const float* sample_cl, const int64_t* host_t, int n_t, const float* cond, int N, int F,
                             int H, int W, int T, hipStream_t s, bool cfg_pair = false, const float* host_tf = nullptr);
    void vae_decode_frames(const float* z_cl, int nframes, int h, int w, float* out_cl3, hipStream_t s, bool small_family_call = false);
    void vae_encode_frames(const float* img_cl4, int n, int H, int W, float* moments_cl8, hipStream_t s);
};
