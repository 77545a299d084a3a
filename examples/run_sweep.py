"""BASELINE configs[4]: "End-to-end: GLMNet + Seq2Seq (PyTorch-ROCm) -> HIP UNet3D 50-step -> VAE decode, sub1 synthetic EEG,
40-concept sweep" -- the whole generation flow of the reference (EEG2Video/inference_eeg2video.py:60-98 with the Seq2Seq / DANA
latents of EEG2Video_New) over 40 concepts x 5 clips, batched and shardable over ranks:

    raw EEG [B,62,200] ----GLMNet (host torch)------> fast / slow label -> DANA's dynamic_beta (0.3 / 0.2)
    DE features [B,310] ---Semantic Predictor (HIP)--> cond [B,77,768]
    EEG windows [B,7,62,100] --Seq2Seq (host torch)--> latents [B,6,4,36,64] --DANA noise + layout fix (HIP)--> [B,4,6,36,64]
    e2v_generate (HIP: 50 x [UNet3D on 2B samples, CFG, DDIM] + VAE decode) -> frames -> uint8 (HIP) -> GIF / NPY writer (host)

Synthetic EEG and random-init weights throughout (no datasets / checkpoints offline).  Prints one JSON line: clips/s of the
whole flow and the share of each stage.  ``--tiny`` runs the same code on the tiny test configuration in seconds.

    python examples/run_sweep.py --dtype bf16 --batch 32            # 200 clips on one GPU
    python -m torch.distributed.run --nproc-per-node 8 examples/run_sweep.py   # concepts sharded over ranks, frames gathered
"""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main(argv=None, engine=None):
    """``engine``: an already built full-size ``Engine`` (tests reuse one; its semantic-predictor weights are (re)loaded here)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--concepts", type=int, default=40)
    ap.add_argument("--per-concept", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per e2v_generate call")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "fp16"])
    ap.add_argument("--out", default="", help="directory for <concept>_<k>.gif (empty: frames are produced but not written)")
    ap.add_argument("--npy", action="store_true", help="write .npy instead of .gif")
    args = ap.parse_args(argv)

    import torch.distributed as dist
    from eeg2video_amd.dist import all_gather_frames, shard_range
    from eeg2video_amd.engine import Engine
    from eeg2video_amd.host_models import GLMNet, Seq2SeqLatents
    from eeg2video_amd.semantic import CLIP
    from eeg2video_amd.util import save_videos_grid
    from eeg2video_amd.weights import (TINY_UNET, TINY_VAE, SemanticConfig, UNetConfig, VAEConfig, counter_normal, synth_state_dict,
                                       unet_param_spec, vae_param_spec)

    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    ucfg, vcfg, scfg = ((TINY_UNET, TINY_VAE, SemanticConfig(in_features=22, hidden=96, tokens=77)) if args.tiny
                        else (UNetConfig(), VAEConfig(), SemanticConfig()))
    F, h, w = (3, 4, 6) if args.tiny else (6, 36, 64)
    t_setup = time.perf_counter()
    if engine is not None:
        eng = engine
    else:
        eng = Engine(ucfg, vcfg, local, sem_cfg=scfg)
        eng.load_state_dict(synth_state_dict(unet_param_spec(ucfg), seed=42, mode="reference_init"))
        eng.load_state_dict(synth_state_dict(vae_param_spec(vcfg), seed=43, mode="reference_init"), prefix="vae.")
        eng.finalize(Engine.UNET | Engine.VAE)
    sem = CLIP(scfg, engine=eng)
    if getattr(eng, "_sweep_sem_seed", None) != 44:                 # (a reused engine keeps the 0.89 G synthetic parameters)
        sem.init_synthetic(44)
        eng._sweep_sem_seed = 44
    eng.set_compute_dtype(args.dtype)
    dev = eng.device
    torch.manual_seed(0)
    glm = GLMNet(out_dim=2, emb_dim=64, C=62, T=200).to(dev).eval()
    s2s = Seq2SeqLatents(d_model=512 if not args.tiny else 64, latent_shape=(4, h, w), frames=F).to(dev).eval()
    t_setup = time.perf_counter() - t_setup

    total = args.concepts * args.per_concept
    lo, hi = shard_range(total, rank, world)                       # clips partition over ranks, no data-path exchange
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    neg = t(counter_normal(2, "neg", (1, scfg.tokens, ucfg.cross_attention_dim))).to(dev)
    stage = {"glmnet": 0.0, "seq2seq": 0.0, "semantic": 0.0, "dana": 0.0, "generate": 0.0, "uint8_d2h": 0.0, "write": 0.0}
    mine = []
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=max(1, min(8, len(os.sched_getaffinity(0)) - 2)))
    pending = []

    def tick(name, t0):
        torch.cuda.synchronize()
        stage[name] += time.perf_counter() - t0

    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for b0 in range(lo, hi, args.batch):
        ids = list(range(b0, min(b0 + args.batch, hi)))
        B = len(ids)
        # synthetic "sub1" recordings of these clips (seeded by the clip id: any rank reproduces any clip)
        raw = torch.stack([t(counter_normal(1000 + k, "raw", (62, 200))) for k in ids]).to(dev)
        win = torch.stack([t(counter_normal(2000 + k, "win", (7, 62, 100))) for k in ids]).to(dev)
        de = torch.stack([t(counter_normal(3000 + k, "de", (scfg.in_features,))) for k in ids]).to(dev)
        t0 = time.perf_counter()
        with torch.no_grad():
            fast = glm(raw[:, None]).argmax(1).bool()               # fast / slow optical-flow class -> dynamic_beta
        tick("glmnet", t0)
        t0 = time.perf_counter()
        lat_s2s = s2s(win).float()                                  # [B, F, 4, h, w]
        tick("seq2seq", t0)
        t0 = time.perf_counter()
        cond = sem(de).reshape(B, scfg.tokens, -1)
        tick("semantic", t0)
        t0 = time.perf_counter()
        e_div, e_same = torch.empty(lat_s2s.shape), torch.empty((B, 1) + tuple(lat_s2s.shape[2:]))
        for j, k in enumerate(ids):                                 # DANA's two draws, seeded per clip: batching never changes a clip
            g = torch.Generator().manual_seed(4000 + k)
            e_div[j] = torch.randn(lat_s2s.shape[1:], generator=g)
            e_same[j] = torch.randn((1,) + tuple(lat_s2s.shape[2:]), generator=g)
        lat = torch.empty((B, 4, F, h, w), device=dev)
        for flag, beta in ((True, 0.3), (False, 0.2)):             # DANA's two dynamic_beta values, one kernel call per group
            sel = (fast == flag).nonzero().flatten()
            if sel.numel():
                cpu = sel.cpu()
                lat[sel] = eng.dana_noise(lat_s2s[sel], e_div[cpu], e_same[cpu], t=[499] * int(sel.numel()), dynamic_beta=beta)
        tick("dana", t0)
        t0 = time.perf_counter()
        frames = eng.generate(lat, cond, neg, args.steps, 12.5, 0.0, decode=True)
        tick("generate", t0)
        t0 = time.perf_counter()
        u8 = eng.frames_to_uint8(frames).cpu()
        tick("uint8_d2h", t0)
        mine.append(u8)
        if args.out:
            # the GIF encoder (Pillow's quantiser, ~0.45 s per clip on one core) runs on a worker pool while the GPU generates
            # the next batch; "write" = the time the main thread spent waiting for it at the end
            for j, k in enumerate(ids):
                name = f"{k // args.per_concept:02d}_{k % args.per_concept}" + (".npy" if args.npy else ".gif")
                pending.append(pool.submit(save_videos_grid, u8[j:j + 1], os.path.join(args.out, name)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in pending:
        f.result()
    stage["write"] += time.perf_counter() - t0
    pool.shutdown()
    elapsed = time.perf_counter() - t_all
    u8_all = torch.cat(mine) if mine else torch.zeros((0, 3, F, 8 * h, 8 * w), dtype=torch.uint8)
    if world > 1:
        u8_all = all_gather_frames(u8_all.to(dev).float() / 255.0, as_uint8=True).cpu()
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    if rank == 0:
        assert u8_all.shape == (total, 3, F, 8 * h, 8 * w) and u8_all.dtype == torch.uint8
        host = stage["glmnet"] + stage["seq2seq"]
        print(json.dumps({"workload": f"{args.concepts} concepts x {args.per_concept} clips, {args.steps}-step DDIM, CFG 12.5, {F}x{8 * h}x{8 * w}, {args.dtype}, batch {args.batch}",
                          "clips": total, "n_gpus": world, "seconds": elapsed, "clips_per_s": total / elapsed, "setup_s": t_setup,
                          "stage_seconds_rank0": stage, "host_model_share": host / max(elapsed, 1e-9),
                          "generate_share": stage["generate"] / max(elapsed, 1e-9), "uint8_checksum": int(u8_all.long().sum())}))
    if world > 1:
        dist.destroy_process_group()
    return u8_all


if __name__ == "__main__":
    main()
