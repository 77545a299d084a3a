#!/usr/bin/env python3
"""Headline benchmark of the hot path: 6-frame 288x512 clips/sec for 50-step DDIM with classifier-free guidance
and VAE decode, fp32 (BASELINE.json configs[1]: "1xMI355X: batch=8 synthetic latents, 50-step DDIM, 288x512x6
decode, fp32"), at N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: ``e2v_generate`` (50 x [UNet3D on 2B samples, CFG + DDIM
update] + VAE decode of B clips), inputs resident in HBM, frames left in HBM; for N > 1 each rank runs its own B
clips (weak scaling, no data-path collective) and the decoded frames are all-gathered over RCCL inside the step.
Rank 0 prints ONE JSON line.  Weights are random-init of the SD-v1-4 architecture from the counter RNG, inputs
are synthetic normals (no checkpoints / datasets offline).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TFLOP_UNET_SAMPLE = 2.962      # SURVEY.md App. B (2 x MAC)
TFLOP_VAE_CLIP = 8.448
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense (the 5 PF headline includes 2:1 sparsity)
PEAK_HBM_GBPS = 8000.0
HBM_BOUND = {"groupnorm", "groupnorm_silu", "layernorm", "ddim_cfg_step", "softmax_rows", "temporal_attn"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(usd, vsd, gpu_unet_out, x, cond):
    """The oracle (CPU restatement of the reference's op sequence) timed on this box's host cores, on a bounded
    sample of the same workload: one UNet3D sample forward and one VAE frame decode, fp32."""
    from eeg2video_amd.weights import TINY_UNET, UNetConfig, VAEConfig, counter_normal, synth_state_dict, unet_param_spec
    from oracle import unet3d_forward, vae_decode
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    # the box's CPU share (16 host cores per GPU), not os.cpu_count() of the whole host: oversubscribing the
    # cgroup quota stalls the intra-op pool
    cores = int(os.environ.get("E2V_CPU_THREADS", "0")) or min(torch.get_num_threads(), 16)
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads (os.cpu_count() = {os.cpu_count()}, affinity = {len(os.sched_getaffinity(0))})")
    with torch.no_grad():
        tiny = {k: t(v) for k, v in synth_state_dict(unet_param_spec(TINY_UNET), seed=1).items()}
        unet3d_forward(tiny, TINY_UNET, torch.zeros(1, 4, 2, 8, 8), 1, torch.zeros(1, 3, TINY_UNET.cross_attention_dim))
        usd_t = {k: t(v) for k, v in usd.items()}
        t0 = time.perf_counter()
        ref = unet3d_forward(usd_t, UNetConfig(), x, 501, cond)
        t_unet = time.perf_counter() - t0
        del usd_t
        vsd_t = {k: t(v) for k, v in vsd.items()}
        z = t(counter_normal(77, "z", (1, 4, 36, 64)))
        t0 = time.perf_counter()
        vae_decode(vsd_t, VAEConfig(), z)
        t_vae = time.perf_counter() - t0
    clip_s = 100.0 * t_unet + 6.0 * t_vae
    err = ((gpu_unet_out.cpu().double() - ref.double()).abs().max() / ref.double().abs().max()).item()
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {
        "value": 1.0 / clip_s, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": (f"oracle (torch {torch.__version__} CPU fp32, {model}): 1 UNet3D sample forward [1,4,6,36,64] = "
                   f"{t_unet:.2f} s (2.96 TFLOP) + 1 VAE frame decode 36x64->288x512 = {t_vae:.2f} s (1.41 TFLOP); "
                   "extrapolated to one 50-step CFG clip = 100 UNet samples + 6 frames"),
        "unet_sample_s": t_unet, "vae_frame_s": t_vae,
    }, err


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU (BASELINE configs[1]: 8)")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--guidance", type=float, default=12.5)
    ap.add_argument("--profile-ddim-steps", type=int, default=0,
                    help="DDIM steps of the event-instrumented pass (0 = --ddim-steps: the instrumented pass is the timed workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-table", default="", help="write the per-kernel-class table (JSON) here")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "f32x3"],
                    help="fp32 = BASELINE configs[1] (default, the metric's configuration); bf16 = configs[2]: bf16 MFMA "
                         "(fp32 accumulate) for convs / linears, fp32 activations, GroupNorm, LayerNorm, softmax")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank path on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "E2V_FORCE_DEVICE" in os.environ:      # rehearsal only: several ranks on one GPU (with --backend gloo)
        local = int(os.environ["E2V_FORCE_DEVICE"])
    if args.gpus != world:
        if args.gpus > 1:
            log(f"--gpus {args.gpus} needs a {args.gpus}-rank launch (python -m torch.distributed.run --nproc-per-node {args.gpus} ...)")
            return 2
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    from eeg2video_amd.dist import all_gather_frames
    from eeg2video_amd.pipeline import build_pipeline
    from eeg2video_amd.weights import (UNetConfig, VAEConfig, counter_normal, synth_state_dict, unet_param_spec,
                                       vae_param_spec)

    t_setup = time.perf_counter()
    if args.dtype == "f32x3":                      # opt-in, experimental: the weights are split at finalize, so select it first
        os.environ["E2V_F32X3"] = "1"
    ucfg, vcfg = UNetConfig(), VAEConfig()
    usd = synth_state_dict(unet_param_spec(ucfg), seed=42, mode="reference_init")
    vsd = synth_state_dict(vae_param_spec(vcfg), seed=43, mode="reference_init")
    pipe = build_pipeline(ucfg, vcfg, device=local, unet_sd=usd, vae_sd=vsd)
    eng = pipe.unet.engine
    if args.dtype != "f32x3":
        eng.set_compute_dtype(args.dtype)
    dev = eng.device
    B = args.batch
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    # synthetic inputs, resident in HBM before the timed region (clip k of rank r uses seed 1234 + r*B + k)
    lat = torch.stack([t(counter_normal(1234 + rank * B + k, "latent", (4, 6, 36, 64))) for k in range(B)]).to(dev)
    cond = torch.stack([t(counter_normal(1235 + 7919 * (rank * B + k), "cond", (77, 768))) for k in range(B)]).to(dev)
    unc = t(counter_normal(1236, "uncond", (1, 77, 768))).to(dev)
    if rank == 0:
        log(f"setup {time.perf_counter() - t_setup:.1f} s; device memory held {eng.device_bytes() / 2**30:.2f} GiB")

    def step():
        frames = eng.generate(lat, cond, unc, args.ddim_steps, args.guidance, 0.0, decode=True)
        if world > 1:
            frames = all_gather_frames(frames)
        return frames

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    finite = bool(torch.isfinite(out).all().item())
    clips = world * B * args.steps
    value = clips / elapsed

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel: HIP events around every launch of an instrumented pass --------
        if args.profile_ddim_steps <= 0:
            args.profile_ddim_steps = args.ddim_steps
        eng.profile_begin()
        eng.generate(lat, cond, unc, args.profile_ddim_steps, args.guidance, 0.0, decode=True)
        table = eng.profile_end()
        tot_ms = sum(v["ms"] for v in table.values())
        for k, v in table.items():
            v["avg_us"] = 1e3 * v["ms"] / max(v["launches"], 1)
            v["tflops"] = v["flops"] / (v["ms"] * 1e9) if v["ms"] > 0 else 0.0
            v["gbps"] = v["bytes"] / (v["ms"] * 1e6) if v["ms"] > 0 else 0.0
            v["share"] = v["ms"] / tot_ms if tot_ms > 0 else 0.0
        dom = max(table, key=lambda k: table[k]["ms"])
        d = table[dom]
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")     # written from a rocprofv3 --pmc pass
        if os.path.isfile(pmc):
            try:
                traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch") if (B == 8 and args.dtype == "fp32") else None
            except Exception:
                traffic = None
        if dom in HBM_BOUND:
            roof = {"bound": "hbm", "achieved": d["gbps"], "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": d["gbps"] / PEAK_HBM_GBPS}
        else:
            # f32x3 executes 6 bf16 MFMA flops per fp32 flop counted: its fp32-equivalent ceiling is the bf16 peak / 6
            peak = PEAK_BF16_MFMA_TFLOPS / 6 if "f32x3" in dom else PEAK_BF16_MFMA_TFLOPS if "bf16" in dom else PEAK_F32_MFMA_TFLOPS
            roof = {"bound": "mfma", "achieved": d["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": d["tflops"] / peak}
        roof.update({"traffic": traffic, "kernel": dom, "launches": d["launches"], "avg_launch_us": d["avg_us"],
                     "share_of_gpu_time": d["share"],
                     "sample": f"HIP events around every launch of one e2v_generate pass ({args.profile_ddim_steps} DDIM steps + decode, B={B})",
                     "whole_path_direct_conv_flops_over_f32_mfma_peak": (value / world) * (2 * args.ddim_steps * TFLOP_UNET_SAMPLE + TFLOP_VAE_CLIP) / PEAK_F32_MFMA_TFLOPS})
        if args.kernel_table:
            os.makedirs(os.path.dirname(os.path.abspath(args.kernel_table)), exist_ok=True)
            json.dump(table, open(args.kernel_table, "w"), indent=1, sort_keys=True)
        log("kernel classes (instrumented pass): " + ", ".join(
            f"{k}: {v['share'] * 100:.1f}% {v['tflops']:.1f}TF {v['gbps']:.0f}GB/s" for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"])))

        cpu, parity = None, None
        if world == 1 and not args.no_cpu_baseline:
            x1 = t(counter_normal(1234, "latent", (1, 4, 6, 36, 64)))
            c1 = t(counter_normal(1235, "cond", (1, 77, 768)))
            y_gpu = pipe.unet(x1.to(dev), 501, c1.to(dev)).sample
            torch.cuda.synchronize()
            cpu, err = cpu_baseline(usd, vsd, y_gpu, x1, c1)
            parity = {"unet_sample_max_abs_over_max_ref": err, "tolerance": 5e-2 if args.dtype == "bf16" else 1e-3}
        result = {
            "metric": "6-frame 288x512 clips/sec (50-step DDIM)", "value": value, "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 multiply / f32 accumulate, f32 activations",
                      "f32x3": "f32 products from 3-way split bf16 operands (6 bf16 MFMAs), f32 accumulate"}[args.dtype], "data": "synthetic",
            "config": {"workload": (f"{world}xMI355X: batch={B}/GPU synthetic latents [B,4,6,36,64] + [B,77,768] cond, "
                                    f"{args.ddim_steps}-step DDIM, CFG {args.guidance}, 288x512x6 VAE decode, "
                                    + {"fp32": "fp32 (BASELINE configs[1])", "bf16": "bf16 MFMA with fp32 norms (BASELINE configs[2])",
                                       "f32x3": "fp32-equivalent via split bf16 (experimental, opt-in)"}[args.dtype]),
                       "clips_per_gpu": B, "ddim_steps": args.ddim_steps, "guidance_scale": args.guidance,
                       "unet_samples_per_ddim_step": 2 * B, "weights": "random-init SD-v1-4 architecture, counter RNG seed 42/43",
                       "collective": "RCCL all-gather of decoded frames" if world > 1 else "none"},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "output_finite": finite,
            "gpu_over_cpu": (value / cpu["value"]) if cpu else None,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if finite else 1


if __name__ == "__main__":
    sys.exit(main())
