#!/usr/bin/env python3
"""Headline benchmark of the hot path: 6-frame 288x512 clips/sec for 50-step DDIM with classifier-free guidance
and VAE decode (BASELINE.json configs[1]: "1xMI355X: batch=8 synthetic latents, 50-step DDIM, 288x512x6 decode,
fp32"), at N GPUs of one node.

    python bench.py --gpus N --steps K --warmup W            # N > 1: starts its own N-rank launch (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               # what the parent runs / what a driver may run itself

A "step" is one pass of the hot path over one batch, boundary to boundary as the reference has it
(pipeline_tuneeeg2video.py:287-334 + the .cpu() of :183): ``e2v_generate`` (50 x [UNet3D on 2B samples, CFG + DDIM
update] + VAE decode of B clips) with inputs resident in HBM, then -- for N > 1 -- the RCCL all-gather of the decoded
frames, then the D2H copy of the frames (fp32, pinned host buffer; rank 0 receives all N*B clips).  Each rank runs its
own B clips (weak scaling, no data-path collective).  Rank 0 prints ONE JSON line.  Weights are random-init of the
SD-v1-4 architecture from the counter RNG, inputs are synthetic normals (no checkpoints / datasets offline).

``--dtype bf16`` is BASELINE configs[2] (bf16 UNet3D / VAE with fp32 GroupNorm statistics; default batch 32).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T_PROC = time.time()           # (process age for the budget guard; refined from /proc below)
TFLOP_UNET_SAMPLE = 2.962      # SURVEY.md App. B (2 x MAC, direct-conv count)
TFLOP_VAE_CLIP = 8.448
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense (the 5 PF headline includes 2:1 sparsity)
PEAK_HBM_GBPS = 8000.0
HBM_BOUND = {"groupnorm", "groupnorm_silu", "groupnorm_stats", "layernorm", "ddim_cfg_step", "softmax_rows", "temporal_attn",
             "wino_in", "wino_out"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (default: 8 = BASELINE configs[1]; 32 with --dtype bf16 = configs[2])")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--guidance", type=float, default=12.5)
    ap.add_argument("--profile-ddim-steps", type=int, default=0,
                    help="DDIM steps of the event-instrumented pass (0 = --ddim-steps: the instrumented pass is the timed workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the event-instrumented pass (A/B timing runs)")
    ap.add_argument("--kernel-table", default="", help="write the per-kernel-class table (JSON) here")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "fp16", "f32x3"],
                    help="fp32 = BASELINE configs[1] (default, the metric's configuration); bf16 = configs[2]; fp16 = the reference's own "
                         "inference dtype (inference_eeg2video.py:69-70: torch_dtype=torch.float16)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--gather", default="fp32", choices=["fp32", "uint8"],
                    help="what the ranks exchange and rank 0 copies to the host: the fp32 frames (the reference's .cpu().float(), default) or "
                         "their uint8 form (save_videos_grid's (x * 255) truncation, done on the GPU: 4x fewer xGMI / PCIe bytes)")
    ap.add_argument("--gather-impl", default="torch", choices=["torch", "cabi"],
                    help="the all-gather through torch.distributed (default) or below the C ABI (e2v_allgather_frames on the library's own "
                         "RCCL communicator; torch.distributed then only ships the 128-byte id)")
    ap.add_argument("--no-configs2", action="store_true",
                    help="skip the BASELINE configs[2] leg (bf16, B = 32) that a default fp32 run at N = 1 attaches to its JSON line")
    ap.add_argument("--configs2-steps", type=int, default=4, help="timed passes of the configs[2] leg")
    ap.add_argument("--configs2-budget-s", type=float, default=455.0,
                    help="run the configs[2] leg only if the process is younger than this when the fp32 part is done (the driver's limit "
                         "is 600 s; the leg takes ~50 s, the CPU baseline ~65 s runs beside it on a thread)")
    ap.add_argument("--dist-single", action="store_true",
                    help="N = 1 rehearsal of the collective: initialise the process group (world size 1) and run the frame "
                         "all-gather through it, so that the RCCL code path executes on a one-GPU box")
    return ap.parse_args(argv)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv) -> int:
    """``python bench.py --gpus N`` outside a launcher: start the N-rank job as CHILD processes and pass its exit code on.
    Nothing here touches the GPU (no torch import in this process), and nothing re-execs."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = os.environ.copy()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    log("bench.py: launching", " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


def host_cores() -> dict:
    """CPU share this process may really use: affinity mask capped by the cgroup CPU quota (a container with 256 visible
    cores and a 16-core quota stalls an intra-op pool sized for 256)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    raw = ""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            raw = open(path).read().strip()
        except OSError:
            continue
        try:
            if path.endswith("cpu.max"):
                q, p = raw.split()
                quota = None if q == "max" else float(q) / float(p)
            else:
                q = float(raw)
                p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                quota = None if q <= 0 else q / p
        except Exception:
            quota = None
        break
    cores = aff if quota is None else max(1, min(aff, int(quota + 0.999)))
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"cores": cores, "affinity": aff, "os_cpu_count": os.cpu_count(), "cgroup_cpu_max": raw or None, "cpu_model": model}


def cpu_oracle_run(usd, vsd, ucfg, vcfg, guidance):
    """BASELINE configs[0] for real, CPU part: the oracle (CPU restatement of the reference's op sequence, fp32) generates ONE clip
    -- random [1,4,6,36,64] latent, [1,77,768] cond, 4-step DDIM with classifier-free guidance, UNet3D + VAE decode -- on this box's
    host cores.  Touches no GPU state: the default run executes it on a thread beside the untimed warm-up steps (whose host thread
    sleeps in the HIP runtime) and joins it before the timed region starts."""
    import numpy as np
    import torch
    from eeg2video_amd.weights import TINY_UNET, counter_normal, synth_state_dict, unet_param_spec
    from oracle import generate, unet3d_forward
    from oracle.pipeline import decode_latents
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    hc = host_cores()
    cores = int(os.environ.get("E2V_CPU_THREADS", "0")) or hc["cores"]
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads ({hc})")
    lat = t(counter_normal(1234, "latent", (1, 4, 6, 36, 64)))
    cond = t(counter_normal(1235, "cond", (1, 77, 768)))
    unc = t(counter_normal(1236, "uncond", (1, 77, 768)))
    n_cpu = 4
    with torch.no_grad():
        tiny = {k: t(v) for k, v in synth_state_dict(unet_param_spec(TINY_UNET), seed=1).items()}     # warm the thread pool
        unet3d_forward(tiny, TINY_UNET, torch.zeros(1, 4, 2, 8, 8), 1, torch.zeros(1, 3, TINY_UNET.cross_attention_dim))
        usd_t = {k: t(v) for k, v in usd.items()}
        trace = {}
        t0 = time.perf_counter()
        x = generate(usd_t, ucfg, None, None, lat, cond, unc, n_cpu, guidance, trace=trace, decode=False)
        t_loop = time.perf_counter() - t0
        del usd_t
        vsd_t = {k: t(v) for k, v in vsd.items()}
        t0 = time.perf_counter()
        ref = decode_latents(vsd_t, vcfg, x)
        t_vae = time.perf_counter() - t0
        del vsd_t
    return {"hc": hc, "cores": cores, "n_cpu": n_cpu, "t_loop": t_loop, "t_vae": t_vae, "lat": lat, "cond": cond, "unc": unc, "x": x, "ref": ref,
            "torch": torch.__version__}


def cpu_baseline(run, pipe_gen, ddim_steps, guidance, overlapped_with=None):
    """The `cpu_baseline` and `parity` objects from a finished cpu_oracle_run: BASELINE.md section 4's extrapolation gives the 50-step
    rate, 1 / (t_vae + ddim_steps * (t_total - t_vae) / 4); the same clip goes through the HIP path for the parity figure."""
    hc, cores, n_cpu, t_loop, t_vae = run["hc"], run["cores"], run["n_cpu"], run["t_loop"], run["t_vae"]
    t_total = t_loop + t_vae
    clip_s = t_vae + ddim_steps * t_loop / n_cpu
    vid, lat_gpu = pipe_gen(run["lat"], run["cond"], run["unc"], n_cpu)
    ref, x = run["ref"], run["x"]
    frames_err = (vid.cpu().double() - ref.double()).abs().max().item()
    lat_err = ((lat_gpu.cpu().double() - x.double()).abs().max() / x.double().abs().max()).item()
    return {
        "value": 1.0 / clip_s, "unit": "clips/s", "cores": cores, "kind": "port",
        "sample": (f"BASELINE configs[0] run in full: oracle (torch {run['torch']} CPU fp32) on {hc['cpu_model']}, {cores} threads "
                   f"(affinity {hc['affinity']}, cgroup cpu.max '{hc['cgroup_cpu_max']}'): 1 clip, {n_cpu}-step DDIM, CFG {guidance}, "
                   f"UNet3D loop {t_loop:.1f} s + VAE decode of 6 frames {t_vae:.1f} s = {t_total:.1f} s; "
                   f"{ddim_steps}-step rate = 1 / (t_vae + {ddim_steps} * t_loop / {n_cpu}) (BASELINE.md section 4)"),
        "wall_s": t_total, "unet_loop_s": t_loop, "vae_decode_s": t_vae, "vae_share": t_vae / t_total,
        "cpu_model": hc["cpu_model"], "affinity": hc["affinity"], "cgroup_cpu_max": hc["cgroup_cpu_max"],
        "overlapped_with": overlapped_with,
    }, {"config": "BASELINE configs[0]: 1 clip, seeds 1234/1235/1236, 4-step DDIM (751,501,251,1), CFG, UNet3D + VAE decode",
        "frames_max_abs": frames_err, "final_latents_max_abs_over_max_ref": lat_err}


def process_age_s() -> float:
    try:
        import psutil
        return time.time() - psutil.Process().create_time()
    except Exception:
        return time.time() - T_PROC


def instrumented_pass(eng, lat, cond, unc, ddim_steps, guidance, B, dtype, value_per_gpu, kernel_table_path=""):
    """Roofline of the dominant kernel class: HIP events (on the launch stream, inside the library) around every launch of one
    e2v_generate pass that is the timed workload itself.  Returns (roofline object, per-class table)."""
    eng.profile_begin()
    eng.generate(lat, cond, unc, ddim_steps, guidance, 0.0, decode=True)
    table = eng.profile_end()
    return roofline_of(table, ddim_steps, B, dtype, value_per_gpu, kernel_table_path)


def roofline_of(table, ddim_steps, B, dtype, value_per_gpu, kernel_table_path="", where="one e2v_generate pass"):
    """The `roofline` object from the per-class event table of an instrumented pass."""
    tot_ms = sum(v["ms"] for v in table.values())
    for k, v in table.items():
        v["avg_us"] = 1e3 * v["ms"] / max(v["launches"], 1)
        v["tflops"] = v["flops"] / (v["ms"] * 1e9) if v["ms"] > 0 else 0.0
        v["gbps"] = v["bytes"] / (v["ms"] * 1e6) if v["ms"] > 0 else 0.0
        v["share"] = v["ms"] / tot_ms if tot_ms > 0 else 0.0
    dom = max(table, key=lambda k: table[k]["ms"])
    d = table[dom]
    # HBM-side traffic of the dominant kernel: PMC counters cannot be read from inside the process, so this is the
    # figure of the committed rocprofv3 --pmc passes over this very command (tools/profile_round.sh), used only when
    # the configuration matches the one those passes ran
    traffic, traffic_source = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")
    if os.path.isfile(pmc):
        try:
            rec = json.load(open(pmc))
            ent = rec.get(dom, {})
            if ent and ent.get("batch", 8) == B and ent.get("dtype", "fp32") == dtype:
                traffic = ent.get("hbm_bytes_per_launch")
                traffic_source = f"profiles/pmc_dominant_kernel.json (static: {rec.get('_source', 'rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes')})"
        except Exception:
            traffic = None
    # (v_mfma_f32_*_f16 and _bf16 run at the same dense rate: one 16-bit peak)
    mfma_peak = PEAK_BF16_MFMA_TFLOPS / 6 if dtype == "f32x3" else PEAK_BF16_MFMA_TFLOPS if dtype in ("bf16", "fp16") else PEAK_F32_MFMA_TFLOPS
    if dom in HBM_BOUND:
        roof = {"bound": "hbm", "achieved": d["gbps"], "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": d["gbps"] / PEAK_HBM_GBPS}
    else:
        # f32x3 executes 6 bf16 MFMA flops per fp32 flop counted: its fp32-equivalent ceiling is the bf16 peak / 6
        peak = PEAK_BF16_MFMA_TFLOPS / 6 if "f32x3" in dom else PEAK_BF16_MFMA_TFLOPS if ("bf16" in dom or "fp16" in dom) else PEAK_F32_MFMA_TFLOPS
        roof = {"bound": "mfma", "achieved": d["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": d["tflops"] / peak}
    roof.update({"traffic": traffic, "traffic_source": traffic_source, "kernel": dom, "launches": d["launches"],
                 "avg_launch_us": d["avg_us"], "flops_per_launch": d["flops"] / max(d["launches"], 1),
                 "bytes_per_launch": d["bytes"] / max(d["launches"], 1), "share_of_gpu_time": d["share"],
                 "sample": f"HIP events around every launch of {where} ({ddim_steps} DDIM steps + decode, B={B})",
                 "whole_path_direct_conv_flops_over_f32_mfma_peak": value_per_gpu * (2 * ddim_steps * TFLOP_UNET_SAMPLE + TFLOP_VAE_CLIP) / PEAK_F32_MFMA_TFLOPS,
                 # the path as a whole against the MFMA peak of the arithmetic type, counted in ALGORITHMIC flops per clip
                 # (direct convolutions, every key of the reference's attention) -- Winograd / sub-pixel convs and the shared frame-0 keys
                 # execute fewer, which is why the fp32 figure can exceed 1
                 "whole_path": {"algorithmic_tflop_per_clip": 2 * ddim_steps * TFLOP_UNET_SAMPLE + TFLOP_VAE_CLIP,
                                "achieved_tflops": value_per_gpu * (2 * ddim_steps * TFLOP_UNET_SAMPLE + TFLOP_VAE_CLIP), "peak": mfma_peak,
                                "frac": value_per_gpu * (2 * ddim_steps * TFLOP_UNET_SAMPLE + TFLOP_VAE_CLIP) / mfma_peak}})
    if kernel_table_path:
        os.makedirs(os.path.dirname(os.path.abspath(kernel_table_path)), exist_ok=True)
        json.dump(table, open(kernel_table_path, "w"), indent=1, sort_keys=True)
    log(f"kernel classes (instrumented pass, {dtype} B={B}): " + ", ".join(
        f"{k}: {v['share'] * 100:.1f}% {v['tflops']:.1f}TF {v['gbps']:.0f}GB/s" for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"])))
    return roof, table


def configs2_leg(eng, args, host_frames_u8=None):
    """BASELINE configs[2] inside the default run: "1xMI355X: bf16 UNet3D with fp32 GroupNorm, batch=32" -- the same engine switched
    to the bf16-activation mode, B = 32 fresh synthetic clips resident in HBM, one event-instrumented full pass that doubles as the
    warm-up (builds the bf16 weight forms), then --configs2-steps (4) timed passes (e2v_generate + D2H of the fp32 frames, as the
    headline)."""
    import numpy as np
    import torch
    from eeg2video_amd.weights import counter_normal
    B = 32
    dev = eng.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    lat = torch.stack([t(counter_normal(1234 + k, "latent", (4, 6, 36, 64))) for k in range(B)]).to(dev)
    cond = torch.stack([t(counter_normal(1235 + 7919 * k, "cond", (77, 768))) for k in range(B)]).to(dev)
    unc = t(counter_normal(1236, "uncond", (1, 77, 768))).to(dev)
    host = torch.empty((B, 3, 6, 288, 512), dtype=torch.float32).pin_memory()
    eng.set_compute_dtype("bf16")
    try:
        # 4 timed passes where the driver's 600 s budget has room for them; the decision is taken HERE, before the leg's first pass
        # (instrumented pass ~11 s + 4 x ~9.5 s + the parity clip: a leg that starts before 420 s ends before 480 s); later, 3
        n = args.configs2_steps if args.configs2_steps != 4 or process_age_s() < 420.0 else 3
        # the event-instrumented pass comes FIRST and is the leg's warm-up (it builds the bf16 weight forms; event timings are per
        # kernel, so the one-off packing launches only add their own small classes)
        eng.profile_begin()
        eng.generate(lat, cond, unc, args.ddim_steps, args.guidance, 0.0, decode=True)
        table = eng.profile_end()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            frames = eng.generate(lat, cond, unc, args.ddim_steps, args.guidance, 0.0, decode=True)
            host.copy_(frames, non_blocking=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        finite = bool(torch.isfinite(frames).all().item())
        value = B * n / el
        roof, _ = roofline_of(table, args.ddim_steps, B, "bf16", value, (args.kernel_table + ".configs2.json") if args.kernel_table else "",
                              where="the leg's first (warm-up) e2v_generate pass")
    finally:
        eng.set_compute_dtype("fp32")
    return {"value": value, "unit": "clips/s", "ms_per_step": 1e3 * el / n, "steps": n, "warmup": "one full pass (the event-instrumented one)",
            "dtype": "bf16", "output_finite": finite,
            "config": {"workload": (f"1xMI355X: batch={B} synthetic latents [B,4,6,36,64] + [B,77,768] cond, {args.ddim_steps}-step DDIM, CFG "
                                    f"{args.guidance}, 288x512x6 VAE decode, bf16 MFMA, bf16 activations in HBM, fp32 accumulate / norm "
                                    "statistics / softmax (BASELINE configs[2])"),
                       "clips_per_gpu": B, "unet_samples_per_ddim_step": 2 * B,
                       "timed_region": "e2v_generate (inputs in HBM) + D2H of the fp32 frames into pinned host memory"},
            "roofline": roof}


def synthetic_inputs(rank, B, dev, latent_shape=(4, 6, 36, 64), cond_shape=(77, 768)):
    """Synthetic inputs of one rank, resident on `dev` before the timed region: clip k of rank r is global clip r*B + k (latent seed
    1234 + r*B + k, conditioning seed 1235 + 7919 (r*B + k)), so that the ranks of an N-GPU job generate N*B DIFFERENT clips."""
    import numpy as np
    import torch
    from eeg2video_amd.weights import counter_normal
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    lat = torch.stack([t(counter_normal(1234 + rank * B + k, "latent", latent_shape)) for k in range(B)]).to(dev)
    cond = torch.stack([t(counter_normal(1235 + 7919 * (rank * B + k), "cond", cond_shape)) for k in range(B)]).to(dev)
    unc = t(counter_normal(1236, "uncond", (1,) + tuple(cond_shape))).to(dev)
    return lat, cond, unc


def rank_body(args, rank, world, eng, *, use_dist, B, frame_shape=(3, 6, 288, 512), latent_shape=(4, 6, 36, 64), cond_shape=(77, 768),
              sync=None, pin=True, cpu_leg=None, configs2=None):
    """What ONE rank of the benchmark does once its engine exists: inputs -> warm-up -> barrier -> K timed steps -> barrier -> max over
    ranks -> (rank 0) the JSON object.  Factored out of main() so that the N > 1 bookkeeping -- per-rank seeds, rank 0's host buffer
    holding world x B clips in rank order, per-rank clocks, the line's contract -- runs under a world-size-2 gloo test with a stub
    engine (tests/test_bench_ranks_gloo.py); main() passes the HIP engine, `torch.cuda.synchronize` and the CPU-baseline / configs[2]
    legs.  `eng` needs: device, generate(lat, cond, unc, ddim_steps, guidance, eta, decode=True), frames_to_uint8, profile_begin /
    profile_end, device_bytes.  Returns {"result": the line's object (rank 0) or None, "finite": bool, "host_frames": rank 0's buffer}."""
    import torch
    import torch.distributed as dist
    from eeg2video_amd.dist import all_gather_frames
    sync = sync or torch.cuda.synchronize
    dev = eng.device
    lat, cond, unc = synthetic_inputs(rank, B, dev, latent_shape, cond_shape)
    # where the frames end up: the reference's `.cpu().float().numpy()` (pipeline_tuneeeg2video.py:183); rank 0 is the consumer
    as_u8 = args.gather == "uint8"
    host_frames = None
    if rank == 0:
        host_frames = torch.empty((world * B,) + tuple(frame_shape), dtype=torch.uint8 if as_u8 else torch.float32)
        if pin:
            host_frames = host_frames.pin_memory()
    gather_engine = eng if args.gather_impl == "cabi" else None

    def step():
        frames = eng.generate(lat, cond, unc, args.ddim_steps, args.guidance, 0.0, decode=True)
        if use_dist:       # every rank runs B clips: no count exchange, no host sync in front of the collective
            frames = all_gather_frames(frames, as_uint8=as_u8, force_collective=True, engine=gather_engine, uniform=True)
        elif as_u8:
            frames = eng.frames_to_uint8(frames)
        if host_frames is not None:
            host_frames.copy_(frames, non_blocking=True)
        return frames

    # The CPU baseline (the oracle on the host cores, ~1 min) runs on a thread BESIDE THE WARM-UP STEPS and is joined before the timed
    # region starts: the warm-up is untimed GPU work whose host thread sleeps in the HIP runtime, so the two share nothing, the timed
    # region never sees the oracle's threads, and the default run is a minute shorter than with the leg at its end.  Without a warm-up
    # to hide it under, the leg runs ALONE at the very end (after the configs[2] leg: its threads never sit beside a timed region).
    cpu_wanted = cpu_leg is not None and rank == 0 and world == 1 and not args.no_cpu_baseline
    cpu_overlap = None
    if cpu_wanted and args.warmup > 0:
        cpu_leg.start()
        cpu_overlap = f"the {args.warmup} untimed warm-up step(s); joined before the timed region started"

    # The LAST warm-up step of rank 0 carries the HIP-event instrumentation (the roofline's table): it is a full pass of the timed
    # workload either way, and it saves the default run a separate 14 s pass.  (No warm-up, or a different --profile-ddim-steps:
    # a separate instrumented pass after the timed region, as before.)
    warm_table = None
    instr_in_warmup = (rank == 0 and not args.no_roofline and args.warmup > 0 and args.profile_ddim_steps in (0, args.ddim_steps))
    for i in range(args.warmup):
        if instr_in_warmup and i == args.warmup - 1:
            eng.profile_begin()
            step()
            warm_table = eng.profile_end()
        else:
            step()
    sync()
    if cpu_wanted and args.warmup > 0:
        t_join = time.perf_counter()
        cpu_leg.join()
        log(f"cpu_baseline thread joined {time.perf_counter() - t_join:.1f} s after the warm-up ended")
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    t_own = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    own_ms = None
    if use_dist:
        # a rank's OWN time for its K steps (up to the sync, before the closing barrier): a straggler shows as a low per-rank rate
        tt = torch.tensor([t_own], device=dev, dtype=torch.float64)
        alls = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(alls, tt)
        own_ms = [1e3 * float(a.item()) / args.steps for a in alls]
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    finite = bool(torch.isfinite(out.float()).all().item())
    clips = world * B * args.steps
    value = clips / elapsed
    host_snapshot = host_frames.clone() if host_frames is not None and not pin else host_frames     # (tests read the timed steps' buffer)

    # the exchange and the D2H on their own (both are inside the timed step above)
    gather_ms = d2h_ms = None
    frames1 = eng.generate(lat, cond, unc, 1, args.guidance, 0.0, decode=True)
    sync()
    if use_dist:
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            g = all_gather_frames(frames1, as_uint8=as_u8, force_collective=True, engine=gather_engine, uniform=True)
        sync()
        gather_ms = 1e3 * (time.perf_counter() - t1) / 3
    else:
        g = eng.frames_to_uint8(frames1) if as_u8 else frames1
    if host_frames is not None:
        t1 = time.perf_counter()
        for _ in range(3):
            host_frames.copy_(g, non_blocking=True)
        sync()
        d2h_ms = 1e3 * (time.perf_counter() - t1) / 3
    rccl_ranks = dist.get_world_size() if use_dist else 0
    del g, frames1

    result = None
    if rank == 0:
        roof = None
        if not args.no_roofline:
            if args.profile_ddim_steps <= 0:
                args.profile_ddim_steps = args.ddim_steps
            if warm_table is not None:
                roof, _ = roofline_of(warm_table, args.ddim_steps, B, args.dtype, value / world, args.kernel_table,
                                      where="the last warm-up step (one e2v_generate pass of the timed workload)")
            else:
                roof, _ = instrumented_pass(eng, lat, cond, unc, args.profile_ddim_steps, args.guidance, B, args.dtype, value / world,
                                            args.kernel_table)

        # ---- BASELINE configs[2] (bf16, B = 32) as a second leg of the default run, inside the driver's wall-clock budget ----
        c2 = None
        if configs2 is not None and world == 1 and args.dtype == "fp32" and not args.no_configs2 and not args.no_roofline and not use_dist:
            age = process_age_s()
            if age < args.configs2_budget_s:
                log(f"configs[2] leg: process age {age:.0f} s < {args.configs2_budget_s:.0f} s, running")
                try:
                    c2 = configs2(eng, args)
                except Exception as e:      # the headline line must survive a failure of the extra leg; the reason is reported
                    c2 = {"skipped": f"failed: {type(e).__name__}: {e}"}
            else:
                c2 = {"skipped": f"process age {age:.0f} s >= budget {args.configs2_budget_s:.0f} s (the driver's limit is 600 s)"}

        cpu, parity = None, None
        if cpu_wanted:
            if args.warmup <= 0:            # no warm-up to hide it under: alone, after every timed region of the run
                cpu_leg.start()
                cpu_leg.join()
                cpu_overlap = "nothing: run alone after the timed regions (no warm-up steps to hide it under)"
            cpu, parity = cpu_leg.objects(eng, args, cpu_overlap)
        dtype_name = {"fp32": "f32", "bf16": "bf16", "fp16": "f16",
                      "f32x3": "f32 products from 3-way split bf16 operands (6 bf16 MFMAs), f32 accumulate"}[args.dtype]
        mode_text = {"fp32": "fp32 (BASELINE configs[1])",
                     "bf16": "bf16 MFMA, bf16 activations in HBM, fp32 accumulate / norm statistics / softmax (BASELINE configs[2])",
                     "fp16": "fp16 MFMA, fp16 activations in HBM, fp32 accumulate / norm statistics / softmax (the reference's own inference dtype)",
                     "f32x3": "fp32-equivalent via split bf16 (experimental, opt-in)"}[args.dtype]
        result = {
            "metric": "6-frame 288x512 clips/sec (50-step DDIM)", "value": value, "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": (f"{world}xMI355X: batch={B}/GPU synthetic latents [B,4,6,36,64] + [B,77,768] cond, "
                                    f"{args.ddim_steps}-step DDIM, CFG {args.guidance}, 288x512x6 VAE decode, " + mode_text),
                       "clips_per_gpu": B, "ddim_steps": args.ddim_steps, "guidance_scale": args.guidance,
                       "unet_samples_per_ddim_step": 2 * B, "weights": "random-init SD-v1-4 architecture, counter RNG seed 42/43",
                       "timed_region": "e2v_generate (inputs in HBM)" + (" + all-gather of frames" if use_dist else "") +
                                       f" + D2H of the {args.gather} frames into pinned host memory (rank 0)",
                       "collective": (f"{args.backend} all-gather of decoded frames ({args.gather}) over {rccl_ranks} rank(s)" if use_dist else "none")},
            "gather_ms": gather_ms, "d2h_ms": d2h_ms, "rccl_ranks": rccl_ranks if args.backend == "nccl" else 0,
            "gather": {"dtype": args.gather, "impl": ("e2v_allgather_frames (C ABI, library-owned RCCL communicator)" if args.gather_impl == "cabi"
                                                      else f"torch.distributed ({args.backend})") if use_dist else "none"},
            "per_rank_clips_per_s": ({"min": B * 1e3 / max(own_ms), "max": B * 1e3 / min(own_ms), "ms_per_step": own_ms} if own_ms else None),
            "roofline": roof, "configs2": c2, "cpu_baseline": cpu, "parity": parity, "output_finite": finite,
            "gpu_over_cpu": (value / cpu["value"]) if cpu and "value" in cpu else None,
        }
    return {"result": result, "finite": finite, "host_frames": host_snapshot}


class CpuLeg:
    """The CPU-oracle baseline as a thread (cpu_oracle_run touches no GPU state) + the objects the line carries once it is done."""

    def __init__(self, usd, vsd, ucfg, vcfg, guidance):
        self.a = (usd, vsd, ucfg, vcfg, guidance)
        self.out, self.th = {}, None

    def start(self):
        import threading

        def _leg():
            try:
                self.out["run"] = cpu_oracle_run(*self.a)
            except BaseException as e:          # reported in the line instead of killing the run
                self.out["error"] = f"{type(e).__name__}: {e}"
        self.th = threading.Thread(target=_leg, name="cpu_baseline", daemon=True)
        self.th.start()

    def join(self):
        if self.th is not None:
            self.th.join()

    def objects(self, eng, args, overlapped_with):
        self.join()
        if "run" not in self.out:
            return {"skipped": self.out.get("error", "no result")}, None
        dev = eng.device

        def gpu_gen(l, c, u, n):
            return eng.generate(l.to(dev), c.to(dev), u.to(dev), n, args.guidance, 0.0, decode=True, return_latents=True)
        cpu, parity = cpu_baseline(self.out["run"], gpu_gen, args.ddim_steps, args.guidance, overlapped_with)
        parity["tolerance_frames_max_abs"] = {"bf16": 1e-1, "fp16": 2.5e-2}.get(args.dtype, 1e-3)
        return cpu, parity


def main() -> int:
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, argv)

    # stdout carries ONE JSON line and nothing else: libraries below write banners to fd 1 (RCCL prints its version block there
    # on init), so fd 1 is pointed at stderr for the run and the line goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "E2V_FORCE_DEVICE" in os.environ:      # rehearsal only: several ranks on one GPU (with --backend gloo)
        local = int(os.environ["E2V_FORCE_DEVICE"])
    if args.gpus != world:
        log(f"--gpus {args.gpus} but WORLD_SIZE = {world}: the launcher's rank count and --gpus must agree")
        return 2
    if local >= torch.cuda.device_count():
        log(f"rank {rank}: local rank {local} has no GPU (device_count = {torch.cuda.device_count()})")
        return 2
    use_dist = world > 1 or args.dist_single
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from eeg2video_amd.pipeline import build_pipeline
    from eeg2video_amd.weights import UNetConfig, VAEConfig, synth_state_dict, unet_param_spec, vae_param_spec

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
    B = args.batch or (32 if args.dtype in ("bf16", "fp16") else 8)
    if rank == 0:
        log(f"setup {time.perf_counter() - t_setup:.1f} s; device memory held {eng.device_bytes() / 2**30:.2f} GiB")
        if os.environ.get("E2V_LOG_MAPS"):       # profiling aid: where the runtime / tool libraries sit, so that a native backtrace can be attributed
            seen = set()
            for line in open("/proc/self/maps"):
                f = line.split()
                if len(f) >= 6 and "x" in f[1] and any(k in f[5] for k in ("libamdhip64", "libhsa-runtime", "rocprofiler", "libeeg2video_hip", "libroctracer", "librocprofiler")):
                    key = os.path.basename(f[5])
                    if key not in seen:
                        seen.add(key)
                        log(f"map {f[0]} {f[5]}")

    r = rank_body(args, rank, world, eng, use_dist=use_dist, B=B, cpu_leg=CpuLeg(usd, vsd, ucfg, vcfg, args.guidance), configs2=configs2_leg)
    if r["result"] is not None:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(r["result"]) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if r["finite"] else 1


if __name__ == "__main__":
    sys.exit(main())
