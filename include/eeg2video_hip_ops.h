/* eeg2video_hip_ops.h -- kernel-level entry points of libeeg2video_hip.so.
 *
 * These expose the individual gfx950 kernels behind the model-level ABI of eeg2video_hip.h so that each
 * can be checked against the oracle and profiled on its own.  Activations are CHANNEL-LAST fp32 device
 * tensors: [n, F, H, W, C] is a row-major matrix [n*F*H*W][C].  Weights are device pointers in the
 * torch layouts the reference's modules hold (Conv2d [Cout,Cin,3,3], Linear [out,in]).
 * Reference op each one stands for is named per function (paths relative to the reference repo).
 */
#ifndef EEG2VIDEO_HIP_OPS_H
#define EEG2VIDEO_HIP_OPS_H

#include "eeg2video_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* InflatedConv3d 3x3 (EEG2Video/models/resnet.py:10-18) over n_img frames of Hs x Ws, optionally after a
 * nearest resize to (Hi, Wi) (Upsample3D, resnet.py:58-61; Hi = Hs, Wi = Ws: none).  Input = channel concat
 * of x0 (c0) and x1 (c1, may be NULL/0).  Output map (Ho, Wo) = floor((Hi + pad_lo + pad_hi - 3)/stride) + 1
 * with pad_lo rows/cols of zeros above/left (pad_hi is implied by Ho).  Epilogue: + bias[cout]
 * + rowbias[row / rows_per_sample][cout] (time embedding, resnet.py:186) + resid[row][cout]. */
e2v_status e2v_op_conv3x3(e2v_ctx* ctx, const float* x0, int c0, const float* x1, int c1, int n_img, int Hs, int Ws,
                          int Hi, int Wi, int Ho, int Wo, int stride, int pad_lo, const float* w_oihw, const float* bias,
                          int cout, const float* rowbias, int rows_per_sample, const float* resid, float* out,
                          e2v_stream stream);

/* nn.Linear / 1x1 conv: out[M][N] = x[M][K] w[N][K]^T + bias (+ resid).  geglu != 0: w is the GEGLU
 * projection [2*N2][K] in torch row order (value rows then gate rows, attention.py:189 via diffusers GEGLU);
 * out[M][N2] = (x w_v^T + b_v) * gelu_erf(x w_g^T + b_g). */
e2v_status e2v_op_linear(e2v_ctx* ctx, const float* x, int ldx, int64_t M, int K, const float* w, const float* bias,
                         int N, const float* resid, int geglu, float* out, e2v_stream stream);

/* nn.GroupNorm (+ SiLU) with statistics over (C/groups channels) x (P rows) per slab; slabs = samples.
 * 5-D GroupNorm of ResnetBlock3D (resnet.py:177): samples = n, P = F*H*W.  Per-frame GroupNorm of
 * Transformer3DModel (attention.py:99): samples = n*F, P = H*W. */
e2v_status e2v_op_groupnorm(e2v_ctx* ctx, const float* x0, int c0, const float* x1, int c1, int samples, int P,
                            int groups, float eps, const float* gamma, const float* beta, int silu, float* out,
                            e2v_stream stream);

/* nn.LayerNorm over the last dim (attention.py:167,184,190,202) */
e2v_status e2v_op_layernorm(e2v_ctx* ctx, const float* x, int64_t rows, int C, const float* gamma, const float* beta,
                            float eps, float* out, e2v_stream stream);

/* softmax(q k^T * scale) v per (sample, frame, head).
 * mode 0: SparseCausalAttention (attention.py:272-328): q, k, v are [n*F*Nq][ld]; keys of frame f are
 *         [frame 0 ; frame max(f-1, 0)].  mode 1: keys shared by the F frames of a sample, k, v [n*Nk][ldkv]. */
e2v_status e2v_op_attention(e2v_ctx* ctx, const float* q, int ldq, const float* k, const float* v, int ldkv, float* o,
                            int ldo, int n, int F, int heads, int D, int Nq, int Nk, int mode, float scale,
                            e2v_stream stream);

/* attn_temp (attention.py:261-267): self-attention over the F frames of every pixel; qkv [n*F*HW][3C]. */
e2v_status e2v_op_temporal_attention(e2v_ctx* ctx, const float* qkv, float* out, int n, int F, int HW, int heads, int D,
                                     float scale, e2v_stream stream);

/* layout conversion at the boundary: [n][C][FHW] <-> [n][FHW][Cpad] */
e2v_status e2v_op_to_channels_last(e2v_ctx* ctx, const float* in, float* out, int n, int C, int Cpad, int FHW,
                                   e2v_stream stream);
e2v_status e2v_op_from_channels_last(e2v_ctx* ctx, const float* in, int ld, float* out, int n, int C, int FHW,
                                     e2v_stream stream);

/* Test aid: UNet3DConditionModel.forward (EEG2Video/models/unet.py:278-413, as e2v_unet_forward) that also copies out the
 * intermediate tensors oracle/unet3d.py exposes as `taps`, in this order: "emb" (time embedding before the resnets' SiLU, unet.py:345,
 * [N, 1280, 1, 1, 1]), "down0".."down3" (after each down block incl. its downsampler, unet.py:362-373), "mid" (unet.py:376-378),
 * "up0".."up3" (after each up block incl. its upsampler, unet.py:381-404).  Each is written to `taps` (device, fp32) as a contiguous
 * NCFHW tensor, back to back; shapes[5 i ..] = {n, C, F, H, W} of tap i (host, room for 16 taps = 80 entries), *n_taps their number.
 * taps_cap = capacity of `taps` in floats (E2V_EINVAL when too small).  In the bf16 mode the taps are the bf16 tensors widened. */
e2v_status e2v_op_unet_forward_taps(e2v_ctx* ctx, const float* sample, const int64_t* host_t, int n_t, const float* cond, int N,
                                    int F, int H, int W, int T, float* out, float* taps, int64_t taps_cap, int64_t* shapes,
                                    int* n_taps, e2v_stream stream);

/* Test aid: the row-block sums that a conv of the bf16 mode leaves with its output for the GroupNorm that follows (resnet.py:177,188:
 * the statistics pass over the tensor is then skipped): x [rows][C] (device fp32, rounded to bf16 first; rows % 64 == 0, C % 8 == 0)
 * -> out[rows / 64][C][2] = (sum, sum of squares) over each 64-row block, in the library's canonical summation order. */
e2v_status e2v_op_rowblock_sums(e2v_ctx* ctx, const float* x, int64_t rows, int C, float* out, e2v_stream stream);

/* Which kernel and tile would every launch of a configuration take?  On a HOST-ONLY context (e2v_create(cfg, -1, &ctx): no GPU, no
 * weights) this runs e2v_generate -- one guided DDIM step + VAE decode of B clips of [4, F, h, w] latents with T conditioning tokens, the
 * walk of UNet3DConditionModel.forward (EEG2Video/models/unet.py:278-413) and AutoencoderKL.decode -- as a dry run: every launch rule
 * executes, nothing is launched, and each launch leaves a record "class shape -> kernel tile".  `buf` receives one line per distinct
 * record in first-occurrence order, "<count>x <record>\n" (NUL-terminated); *needed = bytes required (call with cap = 0 to size).
 * dtype: E2V_F32 or E2V_BF16.  E2V_ESTATE on a context that owns a device.  No reference counterpart. */
e2v_status e2v_op_describe_dispatch(e2v_ctx* ctx, int dtype, int B, int F, int h, int w, int T, char* buf, int64_t cap, int64_t* needed);

/* Test / profiling aid: set one of the run-time switches of DESIGN.md section 10 (the integer an environment variable of the
 * same name would give it at first use), for same-process A/B comparisons of kernel variants -- e.g. "E2V_BGEMM_PERS" 0/1.
 * Process-wide; E2V_EINVAL for an unknown name.  No reference counterpart. */
e2v_status e2v_op_set_knob(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif
