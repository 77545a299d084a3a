/* eeg2video_hip.h -- C ABI of libeeg2video_hip.so: the MI355X (gfx950) implementation of the
 * EEG2Video generation hot path (Tune-A-Video denoising loop + Stable-Diffusion VAE).
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository gaspachoo/EEG2Video, snapshot 2025-07-11).  The reference is pure Python; its FFI for
 * this path would be a ctypes binding, shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns an e2v_status (0 = ok, < 0 = error); e2v_last_error() gives the text.
 *   - tensors are plain pointers + sizes.  Activations at the boundary are fp32, contiguous, in the
 *     reference's own layouts (latents NCFHW, conditioning [N,T,D], videos NCFHW).  Unless a
 *     parameter says "host", pointers are DEVICE pointers of the ctx's device, owned by the caller.
 *   - the library owns weights and workspace inside the ctx; after e2v_finalize_weights() and one
 *     warm-up call of a given shape no further device allocation happens (workspace is cached).
 *   - one ctx per device; calls on one ctx must be externally serialised (the reference is a single
 *     Python thread on the default stream, @torch.no_grad()).  `stream` is a hipStream_t (NULL = the
 *     default stream); all work of a call is enqueued on it and the call does not synchronise unless
 *     stated.
 */
#ifndef EEG2VIDEO_HIP_H
#define EEG2VIDEO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct e2v_ctx e2v_ctx;
typedef void* e2v_stream;          /* hipStream_t */

typedef enum {
    E2V_OK = 0,
    E2V_EINVAL = -1,     /* bad argument (ValueError in the reference) */
    E2V_ESHAPE = -2,     /* shape mismatch ("Unexpected latents shape", pipeline_tuneeeg2video.py:239-240) */
    E2V_ENOWEIGHT = -3,  /* missing / unknown / mis-shaped state-dict key (RuntimeError, unet.py:442-448) */
    E2V_EHIP = -4,       /* HIP runtime error */
    E2V_ESTATE = -5      /* call order (weights not finalized, ...) */
} e2v_status;

typedef enum { E2V_F32 = 0, E2V_F16 = 1, E2V_BF16 = 2, E2V_F32X3 = 3 } e2v_dtype;

/* Mirror of the UNet3DConditionModel ctor kwargs the path uses (EEG2Video/models/unet.py:41-78) and of
 * the AutoencoderKL config (diffusers 0.11.1 vae/config.json).  e2v_default_config()
 * fills the Stable-Diffusion v1-4 values. */
typedef struct {
    int in_channels, out_channels;          /* 4, 4 */
    int block_out_channels[4];              /* 320, 640, 1280, 1280 */
    int layers_per_block;                   /* 2 */
    int cross_attention_dim;                /* 768 */
    int attention_heads;                    /* `attention_head_dim` = 8 = NUMBER of heads (unet_blocks.py:257-259) */
    int norm_num_groups;                    /* 32 */
    float norm_eps;                         /* 1e-5 */
    int flip_sin_to_cos;                    /* 1 */
    float freq_shift;                       /* 0 */
    /* VAE */
    int vae_in_channels, vae_latent_channels;   /* 3, 4 */
    int vae_block_out_channels[4];          /* 128, 256, 512, 512 */
    int vae_layers_per_block;               /* 2 */
    int vae_norm_num_groups;                /* 32 */
    float vae_norm_eps;                     /* 1e-6 */
    double vae_scaling_factor;              /* 0.18215 (pipeline_tuneeeg2video.py:177) */
    /* DDIM (SD-v1-4 scheduler config) */
    int num_train_timesteps;                /* 1000 */
    double beta_start, beta_end;            /* 0.00085, 0.012, "scaled_linear" */
    int steps_offset;                       /* 1 (forced by pipeline_tuneeeg2video.py:59-71) */
    /* Semantic Predictor MLP (EEG2Video/models/train_semantic_predictor.py:11-32):
     * in -> hidden -> hidden -> hidden -> hidden -> sem_tokens * cross_attention_dim, ReLU between */
    int sem_in_features;                    /* 310 */
    int sem_hidden;                         /* 10000 */
    int sem_tokens;                         /* 77 */
    /* Per-block head counts (`attention_head_dim` given as a tuple, unet.py:71,110-111: down block i takes [i] (:131), the mid block
     * [3] (:151), up block i the reversed list's [i] (:165,194); the SD-2.x UNet's 5 / 10 / 20 / 20).  All zero (the default): every
     * block has `attention_heads` heads.  Appended in round 5: e2v_config_size() tells a binding which layout a library has. */
    int attention_heads_per_block[4];
} e2v_config;

void e2v_default_config(e2v_config* cfg);
/* sizeof(e2v_config) as THIS build of the library sees it: a binding that declares the struct itself (ctypes, cgo, JNA ...) asserts
 * its own size against this before the first e2v_default_config / e2v_create, so that a stale field list fails loudly instead
 * of having the library write past the caller's object. */
int64_t e2v_config_size(void);
const char* e2v_version(void);

/* ---- context ------------------------------------------------------------------------------------ */
/* replaces: UNet3DConditionModel.__init__ / TuneAVideoPipeline.__init__ + .to("cuda")
 * (unet.py:41-207, pipeline_tuneeeg2video.py:43-113, inference_eeg2video.py:69-70) */
/* device = -1 creates a HOST-ONLY context: it serves the key scheme and the DDIM schedule (no HIP call is
 * made, so it works on a machine without a GPU); every device entry point then returns E2V_ESTATE. */
e2v_status e2v_create(const e2v_config* cfg, int device, e2v_ctx** out);
void e2v_destroy(e2v_ctx* ctx);
const char* e2v_last_error(const e2v_ctx* ctx);      /* ctx may be NULL: last error of a failed e2v_create */

/* ---- weights ------------------------------------------------------------------------------------ */
/* replaces: ModelMixin.from_pretrained / load_state_dict (unet.py:415-449, inference_eeg2video.py:69).
 * `key` is the reference state-dict key; UNet keys as they are ("conv_in.weight", "down_blocks.0....",
 * SURVEY App. D), VAE keys prefixed "vae." ("vae.decoder.conv_in.weight").  `data` is a HOST pointer in
 * the torch layout (Conv2d [Cout,Cin,kh,kw], Linear [out,in]); the shape is checked against the config. */
e2v_status e2v_load_tensor(e2v_ctx* ctx, const char* key, const void* host_data, e2v_dtype dtype,
                           const int64_t* shape, int ndim);
/* number of keys the config expects / a key by index (for loaders that iterate) */
int64_t e2v_num_expected_keys(const e2v_ctx* ctx);
const char* e2v_expected_key(const e2v_ctx* ctx, int64_t i, int64_t* shape4, int* ndim);
/* re-layout to the kernels' formats (tap-major convs, fused QKV / KV, GEGLU row interleave).
 * which: bit 0 = UNet, bit 1 = VAE, bit 2 = semantic predictor (keys "semantic.mlp.{0,2,4,6,8}.{weight,bias}");
 * every expected key of the selected parts must have been loaded. */
e2v_status e2v_finalize_weights(e2v_ctx* ctx, int which);

/* ---- schedule (host, integer-exact) ---------------------------------------------------------------- */
/* replaces: DDIMScheduler.set_timesteps (call site pipeline_tuneeeg2video.py:287-288).
 * t_i = (i * (T // n))[::-1] + steps_offset, int64; n=50 -> 981, 961, ..., 21, 1. */
e2v_status e2v_ddim_timesteps(const e2v_ctx* ctx, int num_inference_steps, int64_t* host_out);
/* alpha-bar table (fp32 cumprod of 1 - linspace(sqrt(b0), sqrt(b1), T)^2), host copy of T floats.
 * The library computes it with scalar fp32 arithmetic; torch.linspace's vectorised CPU kernel can differ in
 * the last bit depending on the host's SIMD width, so a host that wants the very table its own diffusers
 * install would build hands it in with e2v_set_alphas_cumprod (the Python mirror does). */
e2v_status e2v_ddim_alphas_cumprod(const e2v_ctx* ctx, float* host_out);
e2v_status e2v_set_alphas_cumprod(e2v_ctx* ctx, const float* host_table, int n);
/* replaces: the scheduler config the pipeline reads (scheduler.config.steps_offset / num_train_timesteps,
 * pipeline_tuneeeg2video.py:59-71; tuneavideo/util.py:58).  Hands the ctx the whole schedule of the scheduler object the
 * caller holds -- table of n = num_train_timesteps alpha-bars and steps_offset -- so that the fused loop (e2v_generate,
 * e2v_ddim_invert) and the caller's own stepped loop walk the same timesteps with the same coefficients. */
e2v_status e2v_set_ddim_schedule(e2v_ctx* ctx, const float* host_alphas_cumprod, int n, int steps_offset);

/* ---- the hot path --------------------------------------------------------------------------------- */
/* replaces: UNet3DConditionModel.forward (unet.py:278-413).
 * sample [N,C,F,H,W], timesteps int64 HOST array of length n_t (1 = broadcast, else N),
 * cond [N,T,cross_attention_dim], out [N,out_channels,F,H,W]. */
e2v_status e2v_unet_forward(e2v_ctx* ctx, const float* sample, const int64_t* host_timesteps, int n_t,
                            const float* cond, int N, int F, int H, int W, int T, float* out, e2v_stream stream);

/* The same with FRACTIONAL timesteps, as the sigma-space schedulers the pipeline's constructor accepts produce them
 * (pipeline_tuneeeg2video.py:48-55: LMSDiscrete / EulerDiscrete / EulerAncestralDiscrete: timesteps = linspace(0, T-1, n)[::-1];
 * unet.py:324-339 turns a float timestep into the fp32 argument of the sinusoid).  host_timesteps: n_t floats. */
e2v_status e2v_unet_forward_ft(e2v_ctx* ctx, const float* sample, const float* host_timesteps, int n_t, const float* cond,
                               int N, int F, int H, int W, int T, float* out, e2v_stream stream);

/* replaces: noise_pred chunk + guidance + scheduler.step (pipeline_tuneeeg2video.py:320-325), eta = 0.
 * eps_cond may be NULL (guidance off).  t, t_prev are train timesteps (t_prev < 0 -> final alpha = abar[0]). */
e2v_status e2v_ddim_cfg_step(e2v_ctx* ctx, const float* eps_uncond, const float* eps_cond, const float* x,
                             float* x_out, int64_t count, float guidance_scale, int64_t t, int64_t t_prev,
                             e2v_stream stream);

/* replaces: the guidance line noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)
 * (pipeline_tuneeeg2video.py:320-322) for schedulers whose update is not fused with it (everything but DDIM). */
e2v_status e2v_cfg_combine(e2v_ctx* ctx, const float* eps_uncond, const float* eps_cond, float guidance_scale, float* out,
                           int64_t count, e2v_stream stream);

/* out = sum_{i<n} coefs[i] * xs[i], 1 <= n <= 5 (host arrays of n device pointers / n coefficients): the arithmetic of the
 * linear multistep schedulers the pipeline's constructor accepts (pipeline_tuneeeg2video.py:48-55) -- PNDM/PLMS's
 * (55 e1 - 59 e2 + 37 e3 - 9 e4) / 24 and its sample update; host side in eeg2video_amd/scheduler.py: PNDMScheduler. */
e2v_status e2v_lincomb(e2v_ctx* ctx, int n, const float* const* xs, const float* coefs, float* out, int64_t count,
                       e2v_stream stream);

/* replaces: next_step (EEG2Video_New/Generation/tuneavideo/util.py:56-66), the deterministic DDIM update run towards
 * noise: alpha_t = abar[min(t - T/n, 999)] (final alpha when negative), alpha_next = abar[t];
 * x_next = sqrt(alpha_next) (x - sqrt(1-alpha_t) eps) / sqrt(alpha_t) + sqrt(1-alpha_next) eps. */
e2v_status e2v_ddim_next_step(e2v_ctx* ctx, const float* eps, const float* x, float* x_out, int64_t count, int64_t t,
                              int num_inference_steps, e2v_stream stream);

/* replaces: ddim_loop / ddim_inversion (util.py:74-101; caller train_finetune_videodiffusion.py:326-328): num_inv_steps
 * times { eps = unet(latent, t_i, cond); latent = next_step(eps, t_i, latent) } over the ascending DDIM timesteps, no
 * guidance.  latents [B,4,F,h,w], cond [B,T,cross_attention_dim]; all_latents (may be NULL) receives the reference's
 * list [latent_0 .. latent_n] as [n+1][B,4,F,h,w]; final_latent (may be NULL) = all_latents[-1], what the caller feeds
 * back as `latents=`. */
e2v_status e2v_ddim_invert(e2v_ctx* ctx, const float* latents, const float* cond, int B, int F, int h, int w, int T,
                           int num_inv_steps, float* all_latents, float* final_latent, e2v_stream stream);

/* postprocess = 1 replaces TuneAVideoPipeline.decode_latents (pipeline_tuneeeg2video.py:175-184) without the
 * D2H copy: latents [B,4,F,h,w] -> (vae.decode(latents / 0.18215).sample / 2 + 0.5).clamp(0,1) as videos
 * [B,3,F,8h,8w].  postprocess = 0 replaces AutoencoderKL.decode (call site :179): z [B,4,F,h,w] is decoded as
 * given (no 1/0.18215, no clamp); the (b f) frames of the reference are B = n, F = 1. */
e2v_status e2v_vae_decode(e2v_ctx* ctx, const float* latents, int B, int F, int h, int w, int postprocess,
                          float* videos, e2v_stream stream);
/* replaces: AutoencoderKL.encode(...).latent_dist (train_finetune_videodiffusion.py:264,
 * EEG2Video_New/Seq2Seq/generate_1200_latent.py:38): images [n,3,H,W] -> mean, logvar [n,4,H/8,W/8]
 * (logvar clamped to [-30,20]); no 0.18215 factor (that is the caller's, as in the reference). */
e2v_status e2v_vae_encode(e2v_ctx* ctx, const float* images, int n, int H, int W, float* mean, float* logvar,
                          e2v_stream stream);

/* replaces: the body of TuneAVideoPipeline.__call__ after _encode_eeg (pipeline_tuneeeg2video.py:287-334):
 * set_timesteps, the denoising loop with classifier-free guidance (guidance_scale > 1) and the VAE decode,
 * all enqueued on `stream` without a host round trip per step.
 * latents [B,4,F,h,w] (already scaled by init_noise_sigma = 1), cond [B,T,D], uncond [Bu,T,D] with Bu = 1
 * (broadcast, the reference's negative.npy) or B; videos [B,3,F,8h,8w] (may be NULL: skip decode);
 * latents_out [B,4,F,h,w] (may be NULL). */
e2v_status e2v_generate(e2v_ctx* ctx, const float* latents, const float* cond, const float* uncond, int Bu,
                        int B, int F, int h, int w, int T, int num_inference_steps, float guidance_scale,
                        float eta, float* videos, float* latents_out, e2v_stream stream);

/* ---- the steps either side of the path -------------------------------------------------------- */
/* rank 1 -- replaces CLIP.forward of the Semantic Predictor (EEG2Video/models/train_semantic_predictor.py:11-32, called
 * at pipeline_tuneeeg2video.py:149): eeg [B, sem_in_features] -> embeddings [B, sem_tokens * cross_attention_dim]
 * (the caller reshapes to [B,77,768], :150), all on device. */
e2v_status e2v_semantic_predict(e2v_ctx* ctx, const float* eeg, int B, float* out, e2v_stream stream);

/* rank 2 -- replaces Diffusion.forward of DANA (EEG2Video/models/DANA_module.py:52-72) given its random draws, fused with
 * the latent layout fix 'a b c d e -> a c b d e' of inference_eeg2video.py:77,82:
 *   out[b,c,f] = sqrt(abar_t_b) x0[b,f,c] + sqrt(1 - abar_t_b) (sqrt(1 - beta) eps_div[b,f,c] + sqrt(beta) eps_same[b,0,c])
 * x0, eps_div [B,F,C,H,W]; eps_same [B,1,C,H,W]; host_t int64 [B] in [0, time_steps); linear betas 1e-4 .. 0.02 over
 * time_steps (:42-52); out [B,C,F,H,W] (the pipeline's latent layout). */
e2v_status e2v_dana_noise(e2v_ctx* ctx, const float* x0, const float* eps_div, const float* eps_same,
                          const int64_t* host_t, int time_steps, float dynamic_beta, int B, int F, int C, int H, int W,
                          float* out, e2v_stream stream);

/* rank 3 -- replaces `(x * 255).numpy().astype(np.uint8)` of save_videos_grid (EEG2Video_New/Generation/tuneavideo/
 * util.py:29) on the device, so that frames cross xGMI / PCIe as 1 byte per sample: videos in [0,1] -> uint8. */
e2v_status e2v_frames_to_uint8(e2v_ctx* ctx, const float* videos, uint8_t* out, int64_t count, e2v_stream stream);

/* ---- multi-GPU: the one exchange of the path ------------------------------------------------------------------
 * replaces: nothing in the reference (it generates its 200 clips serially on one GPU, inference_eeg2video.py:90); clips shard
 * over one process per GPU with no data-path exchange, and the decoded frames of all ranks are gathered at the end.  The
 * library opens its own RCCL communicator: rank 0 draws an id (128 bytes, HOST memory), the host side ships it to the other
 * ranks by any channel it has, every rank calls e2v_comm_init (collective: returns when all `world` ranks have called it).
 * RCCL is resolved at run time from the librccl already in the process (else the ROCm one): E2V_ESTATE if there is none. */
e2v_status e2v_comm_unique_id(void* id128_host);
e2v_status e2v_comm_init(e2v_ctx* ctx, const void* id128_host, int rank, int world);
int e2v_comm_world(const e2v_ctx* ctx);                 /* 0: no communicator */
/* `frames`: this rank's `count` floats (e.g. [b,3,F,H,W], the same count on every rank); `out`: world * count elements, rank r's
 * shard at offset r * count -- fp32, or (as_uint8) the (x * 255) truncation save_videos_grid applies (tuneavideo/util.py:29), a
 * quarter of the xGMI bytes.  One ncclAllGather on `stream`; does not synchronise. */
e2v_status e2v_allgather_frames(e2v_ctx* ctx, const float* frames, int64_t count, int as_uint8, void* out, e2v_stream stream);
e2v_status e2v_comm_destroy(e2v_ctx* ctx);              /* also done by e2v_destroy */

/* Per-kernel-class timing with HIP events on the launch stream (used by bench.py for the roofline figures).
 * Between begin and end every kernel launch of the library is bracketed by an event pair and tagged with its
 * algorithmic flops / bytes.  e2v_profile_end synchronises and writes a JSON object
 * {"<kernel class>": {"launches": n, "ms": total, "flops": total, "bytes": total}, ...} into `json`
 * (NUL-terminated, truncated to `cap`); it returns the untruncated length or -1. */
e2v_status e2v_profile_begin(e2v_ctx* ctx);
int64_t e2v_profile_end(e2v_ctx* ctx, char* json, int64_t cap);

/* Arithmetic and storage of the graph's tensors: E2V_F32 (default; fp32 MFMA, fp32 activations, the parity configuration of
 * BASELINE configs[1]) or E2V_BF16 (BASELINE configs[2]: every tensor the graph stores between kernels is bf16 in HBM, bf16
 * MFMA with fp32 accumulation; what stays fp32: the latents / DDIM state, eps, the decoded frames, VAE moments and attention
 * scores, GroupNorm / LayerNorm statistics, softmax, the time-embedding MLP) or E2V_F16 (the same kernels on IEEE half --
 * v_mfma_f32_*_f16, fp16 rows in HBM, fp32 everything listed above: the reference's own inference arithmetic,
 * EEG2Video/inference_eeg2video.py:69-70,76,81 `torch_dtype=torch.float16`, pipelines/pipeline_tuneeeg2video.py:150; three more
 * mantissa bits than bf16 -- every block within 2e-3 of the fp32 oracle where bf16 sits at 1.6e-2 -- at fp16's range: a stored
 * activation beyond 65 504 becomes inf exactly where the reference's .half() run would).  The boundary tensors of this header are
 * fp32 in every mode.  Takes effect for the following calls (the fp16 weight copies are derived on first use).
 * E2V_F32X3 (opt-in, experimental): fp32 results from the bf16 matrix pipe -- every operand of a linear / Winograd-domain
 * GEMM is split exactly into three bf16 pieces and the six significant piece products are accumulated in fp32 (error at
 * the level of the fp32 FMA chain); must be selected BEFORE e2v_finalize_weights (the weights are split there), else
 * E2V_ESTATE.  Also selectable with E2V_F32X3=1 in the environment at e2v_create. */
e2v_status e2v_set_compute_dtype(e2v_ctx* ctx, int dtype);

/* Algorithm of the stride-1 3x3 convolutions in fp32 arithmetic.  E2V_CONV_AUTO (default): Winograd F(4x4,3x3) -- 4x
 * fewer matrix-core flops than the direct implicit GEMM -- where min(Cin, Cout) >= 128 and the map padded to multiples of
 * 4 grows by at most 1.7x, else F(2x2,3x3) (2.25x fewer) where min(Cin, Cout) >= 256, else direct.  Measured deviation
 * of a full 50-step B = 8 generate from the all-direct result: frames 1.4e-5 (F(2x2) only: 5.8e-6), against the 1e-3
 * parity tolerance.  E2V_CONV_DIRECT / E2V_CONV_WINOGRAD / E2V_CONV_WINOGRAD4 force one form wherever it applies.
 * The Winograd-domain weights are made by e2v_finalize_weights, so for the graph entry points call this BEFORE finalizing;
 * the kernel-level e2v_op_conv3x3 follows it immediately.  (The reference leaves this choice to cuDNN: F.conv2d,
 * resnet.py:24-31.) */
enum { E2V_CONV_AUTO = 0, E2V_CONV_DIRECT = 1, E2V_CONV_WINOGRAD = 2, E2V_CONV_WINOGRAD4 = 3 };
e2v_status e2v_set_conv_algo(e2v_ctx* ctx, int algo);

/* bytes of device memory currently held by the ctx (weights + cached workspace) */
int64_t e2v_device_bytes(const e2v_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* EEG2VIDEO_HIP_H */
