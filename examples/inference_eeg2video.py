"""End-to-end drop-in example: the flow of the reference's ``inference_eeg2video.py`` (EEG2Video/inference_eeg2video.py:60-98,
EEG2Video_New/Generation/inference_eeg2video.py) on this library, with synthetic EEG features, synthetic Seq2Seq latents
and random-init weights (there are no datasets or checkpoints in this environment).

    EEG features [B,310] --semantic predictor (HIP)--> cond [B,77*768]
    Seq2Seq latents [B,F,4,h,w] --DANA noise + layout fix (HIP)--> start latents [B,4,F,h,w]
    pipe(model, eeg, latents=..., video_length=6, height=288, width=512, num_inference_steps=.., guidance_scale=12.5).videos
    (videos * 255) -> uint8 on the device, ready for the GIF writer of tuneavideo/util.py

``--tiny`` runs the same code on the tiny test configuration in a second or two.
"""
import argparse, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--mode", default="full", choices=["full", "woDANA", "woSeq2Seq"])
    args = ap.parse_args()

    from eeg2video_amd.engine import Engine
    from eeg2video_amd.pipeline import TuneAVideoPipeline
    from eeg2video_amd.scheduler import DDIMScheduler
    from eeg2video_amd.semantic import CLIP
    from eeg2video_amd.unet import UNet3DConditionModel
    from eeg2video_amd.vae import AutoencoderKL
    from eeg2video_amd.weights import (TINY_UNET, TINY_VAE, SemanticConfig, UNetConfig, VAEConfig, counter_normal)

    # the pipeline reshapes the predictor's output to 77 tokens like the reference (pipeline_tuneeeg2video.py:150)
    ucfg, vcfg, scfg = ((TINY_UNET, TINY_VAE, SemanticConfig(in_features=22, hidden=96, tokens=77)) if args.tiny
                        else (UNetConfig(), VAEConfig(), SemanticConfig()))
    F, h, w = (3, 4, 6) if args.tiny else (6, 36, 64)
    eng = Engine(ucfg, vcfg, 0, sem_cfg=scfg)
    unet = UNet3DConditionModel(sample_size=ucfg.sample_size, in_channels=ucfg.in_channels, out_channels=ucfg.out_channels,
                                block_out_channels=ucfg.block_out_channels, layers_per_block=ucfg.layers_per_block,
                                cross_attention_dim=ucfg.cross_attention_dim, attention_head_dim=ucfg.attention_head_dim,
                                norm_num_groups=ucfg.norm_num_groups, norm_eps=ucfg.norm_eps, engine=eng)
    vae = AutoencoderKL(vcfg, engine=eng)
    unet.init_synthetic(42, "reference_init")
    vae.init_synthetic(43, "reference_init")
    model = CLIP(scfg, engine=eng).init_synthetic(44)                       # the reference's semantic predictor (`model`)
    pipe = TuneAVideoPipeline(vae=vae, tokenizer=None, unet=unet, scheduler=DDIMScheduler(engine=eng))
    pipe.set_progress_bar_config(disable=True)

    B = args.clips
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    eeg_test = t(counter_normal(1, "eeg", (B, scfg.in_features))).cuda()     # StandardScaler-ed EEG features
    negative = t(counter_normal(2, "neg", (1, scfg.tokens, ucfg.cross_attention_dim))).cuda()
    seq2seq = t(counter_normal(3, "seq2seq", (B, F, 4, h, w)))                # latents predicted by the Seq2Seq model
    latents = None
    if args.mode == "woDANA":
        latents = seq2seq.permute(0, 2, 1, 3, 4).contiguous().cuda()           # 'a b c d e -> a c b d e'
    elif args.mode == "full":                                                 # DANA: dynamic-aware noise, layout fix fused
        g = torch.Generator().manual_seed(0)
        latents = eng.dana_noise(seq2seq, torch.randn(seq2seq.shape, generator=g), torch.randn((B, 1, 4, h, w), generator=g),
                                 t=[499] * B, dynamic_beta=0.3)
    t0 = time.perf_counter()
    videos = []
    for i in range(B):                                                        # the reference generates clip by clip (:90-98)
        lat = None if latents is None else latents[i:i + 1]
        videos.append(pipe(model, eeg_test[i:i + 1], latents=lat, video_length=F, height=8 * h, width=8 * w,
                           num_inference_steps=args.steps, guidance_scale=12.5, negative_prompt=negative).videos)
    video = torch.cat(videos)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    u8 = eng.frames_to_uint8(video.cuda())
    assert video.shape == (B, 3, F, 8 * h, 8 * w) and torch.isfinite(video).all() and u8.dtype == torch.uint8
    print(f"{B} clips of {F}x{8 * h}x{8 * w} in {dt:.2f} s ({args.mode}, {args.steps} steps); frames in [{float(video.min()):.3f}, "
          f"{float(video.max()):.3f}]; uint8 checksum {int(u8.sum())}")
    return video


if __name__ == "__main__":
    main()
