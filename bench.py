#!/usr/bin/env python3
"""Headline benchmark: frames/sec of the hot path at 640x360, batch 16 per GPU.

Workload (BASELINE.json configs[1]): ``Reconstructor.predict(x, consistency=False,
project_poi=False)`` = UNet segmentation + ResNet34-STN + nearest homography warp of the
court template, on synthetic uint8-derived frames already resident in HBM.  Arithmetic: the
default "bf16x6" mode (fp32-equivalent split-bf16 contraction, see csrc/conv_s3.hip) or fp32 MFMA
throughout with SFH_PRECISION=fp32.
One "step" = one batch of 16 frames per GPU.  With N > 1 (launched by torch.distributed.run,
one process per GPU) every rank processes its own 16 frames (weak scaling) and the 3x3
thetas are all-gathered over RCCL each step.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     - live HIP-event timing of the DoubleConv 3x3 MFMA launches vs fp32 matrix peak
  cpu_baseline - the CPU oracle (oracle/torch_ref.predict) timed on this box's host cores
                 (rank 0, N == 1 only; reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_*_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
# bf16x6 mode: one fp32-accurate product = 6 bf16 MFMA products, so the MFMA roofline of the
# ALGORITHMIC (fp32-equivalent) work is the bf16 peak / 6
BF16X6_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def pmc_traffic(tag):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json, produced by profiles/collect_pmc.sh); None if absent."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return d[tag]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def train_bench(args):
    """Config 3 (train.py:155-237): forward under net.train(), CE + SmoothL1 + RRMSE + consistency CE,
    backward, clip_grad_value_(0.1), RMSprop(momentum 0.9).  The model's forward/backward run on the HIP
    training kernels; losses and optimizer are the caller's torch ops, as in the reference."""
    import torch
    import torch.nn.functional as F
    from sfh_amd import synth
    from sfh_amd.reconstructor import Reconstructor

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, W, H = args.batch, args.width, args.height
    court = synth.load_court_template("ncaa_nc4_640x360" if (W, H) == (640, 360) else "pitch_v3_nc4_1280x720", 4, B).to(dev)
    poi = synth.load_court_poi("pitch", B).to(dev)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
    net.to(dev).train()
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-5, weight_decay=1e-8, momentum=0.9)
    g = torch.Generator().manual_seed(0)
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
    mask = torch.randint(0, 4, (B, H, W), generator=g).to(dev)
    weight = torch.ones(B, device=dev)
    gt_poi = torch.rand(B, poi.shape[1], 2, generator=g).to(dev)
    nonzeros = torch.ones(B, poi.shape[1], device=dev)
    num_nonzero = nonzeros.sum(1)

    from sfh_amd import training
    ts = None if args.train_autograd else training.TrainStep(net, lr=1e-5, weight_decay=1e-8, seg_lambda=1.0,
                                                             rec_lambda=1.0, reproj_lambda=1.0, consist_lambda=1.0)
    batch = {"mask": mask, "weight": weight, "poi": gt_poi, "nonzeros": nonzeros, "num_nonzero": num_nonzero}

    def step_hip():
        return ts.step(x, batch).sum()

    def step():
        if ts is not None:
            return step_hip()
        preds = net(x)
        seg = (F.cross_entropy(preds["logits"], mask, reduction="none").mean(dim=(1, 2)) * weight).mean()
        rec = (F.smooth_l1_loss(preds["warp_mask"], mask.float() / 4.0, reduction="none").mean(dim=(1, 2)) * weight).mean()
        dist = torch.sqrt(torch.sum((gt_poi - preds["poi"]) ** 2, dim=2))
        reproj = torch.mean(torch.sum(dist * nonzeros, dim=1) / num_nonzero)
        cons = F.cross_entropy(preds["logits"], (preds["warp_mask"] * 4).to(torch.long))
        loss = seg + rec + reproj + cons
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_value_(net.parameters(), 0.1)
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({
        "metric": "training steps: frames/sec at %dx%d batch=%d (forward + losses + backward + clip + RMSprop)" % (W, H, B),
        "value": round(B * args.steps / el, 2), "unit": "frames/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 2), "higher_is_better": True,
        "dtype": "bf16x6->f32 convs (fp32 MFMA backward-filter)", "data": "synthetic", "final_loss": float(loss.detach()),
        "losses_and_optimizer": "torch ops (caller side)" if args.train_autograd else "HIP kernels (training.TrainStep)",
        "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2**30, 2),
        "config": {"workload": "BASELINE config 3: Reconstructor training step, CE + SmoothL1 + RRMSE + consistency CE"}}),
        flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="frames per GPU per step")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=16,
                    help="frames of the same workload timed on the host cores for cpu_baseline (about 10 s)")
    ap.add_argument("--consistency", action="store_true", help="also compute consist_score + poi")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE config 3 instead of the headline: one training step (forward, losses, "
                         "backward, clip, RMSprop) per batch; prints its own JSON line")
    ap.add_argument("--train-autograd", action="store_true",
                    help="with --train: losses, clip and RMSprop as the caller's torch ops around the model's "
                         "autograd node (the reference's train.py structure) instead of the all-HIP TrainStep")
    args = ap.parse_args()
    if args.train or args.train_autograd:
        return train_bench(args)

    import torch
    import torch.distributed as dist
    from sfh_amd import synth, engine, sharding
    from sfh_amd.reconstructor import Reconstructor

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    B, W, H = args.batch, args.width, args.height
    tmpl_name = "ncaa_nc4_640x360" if (W, H) == (640, 360) else "pitch_v3_nc4_1280x720"
    court = synth.load_court_template(tmpl_name, 4, B).to(dev)
    if tuple(court.shape[2:]) != (H, W):
        raise SystemExit(f"no court template fixture for {W}x{H}")
    poi = synth.load_court_poi("pitch", B).to(dev)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H),
                        warp_with_nearest=True)
    sd = synth.synth_state_dict(net.state_dict(), 0)
    net.load_state_dict(sd)
    net.to(dev).eval()

    # distinct synthetic batches, resident in HBM before the timed region (rank r, step k -> seed)
    nbatches = 2
    frames = [synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=1000 * rank + k)).to(dev)
              for k in range(nbatches)]
    gbuf = [torch.empty((B, 10), device=dev) for _ in range(world)] if world > 1 else None

    def step(k):
        out = net.predict(frames[k % nbatches], consistency=args.consistency, project_poi=args.consistency)
        if world > 1:  # the one exchange step of the sharded path: theta (+score) rows to every rank
            dist.all_gather(gbuf, sharding.pack_results(out["theta"], out.get("consist_score")))
        return out

    with torch.no_grad():
        for k in range(args.warmup):
            step(k)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        timer = engine.ConvTimer()
        engine.PackedConv.timer = timer
        t0 = time.perf_counter()
        for k in range(args.steps):
            out = step(k)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        engine.PackedConv.timer = None

    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = el.item()
    frames_total = B * args.steps * world
    fps = frames_total / elapsed

    summ = timer.summary()
    n, fl, ms = summ.get("doubleconv3x3", (0, 0.0, 1.0))
    achieved = fl / (ms * 1e-3) / 1e12 if n else 0.0
    x6 = net.precision == "bf16x6"
    peak = BF16X6_PEAK_TFLOPS if x6 else FP32_MFMA_PEAK_TFLOPS
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": pmc_traffic("doubleconv3x3") if (W, H, B) == (640, 360, 16) else None,
                "peak_basis": ("2500 TFLOP/s dense bf16 MFMA / 6 bf16 products per fp32-accurate product; the "
                               "launches execute 6x the algorithmic FLOPs on the bf16 matrix cores "
                               f"(= {achieved * 6:.0f} TFLOP/s of bf16 MFMA work, {achieved * 6 / BF16_MFMA_PEAK_TFLOPS:.1%} of 2.5 PFLOP/s)"
                               if x6 else "157.3 TFLOP/s dense fp32 MFMA (v_mfma_f32_16x16x4_f32)"),
                "kernel": ("conv_s3_kernel<3x3> (DoubleConv, split-bf16)" if x6 else "conv_mfma_kernel<3x3,s1> (DoubleConv)"),
                "launches": n,
                "avg_launch_ms": round(ms / max(n, 1), 4),
                "algorithmic_gflop_per_launch": round(fl / max(n, 1) / 1e9, 2),
                "share_of_step_time": round(ms * 1e-3 / elapsed, 4)}
    other = {t: {"launches": v[0], "tflops": round(v[1] / (v[2] * 1e-3) / 1e12, 2), "ms_per_step": round(v[2] / args.steps, 3)}
             for t, v in summ.items()}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_ref
        ncpu = min(usable_cores(), int(os.environ.get("SFH_CPU_THREADS", "64")))
        torch.set_num_threads(ncpu)
        nf = min(args.cpu_frames, B)
        xc = frames[0][:nf].cpu()
        sd_cpu = {k: v.cpu() for k, v in sd.items()}
        court_c, poi_c = court.cpu(), poi.cpu()
        with torch.no_grad():
            torch_ref.predict(xc[:1], sd_cpu, court_c, poi_c, warp_size=(W, H), unet_size=(W, H),
                              target_size=(W, H), consistency=False)  # warm-up (1 frame)
            t0 = time.perf_counter()
            ref = torch_ref.predict(xc, sd_cpu, court_c, poi_c, warp_size=(W, H), unet_size=(W, H),
                                    target_size=(W, H), consistency=False)
            cpu_s = time.perf_counter() - t0
        with torch.no_grad():
            got = net.predict(frames[0][:nf], consistency=False)
        dth = (got["theta"].cpu() - ref["theta"]).abs().max().item()
        model = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        cpu_baseline = {"value": round(nf / cpu_s, 4), "unit": "frames/s", "cores": ncpu, "kind": "port",
                        "sample": f"{nf} frames of the same {W}x{H} workload, 1 warm-up frame, "
                                  f"torch {torch.__version__} CPU fp32, {ncpu} threads, {model}",
                        "max_abs_dtheta_gpu_vs_cpu": dth}

    if rank == 0:
        line = {
            "metric": "frames/sec at 640x360 batch=16 (1/2/4/8 GPU) + homography L1 vs ref",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x6->f32 (3-way bf16 split operands, 6 bf16 MFMA products, fp32 accumulate; fp32-equivalent)"
                     if x6 else "f32",
            "data": "synthetic",
            "config": {"workload": f"predict(): UNet seg + ResNet34-STN + nearest warp, {W}x{H}, "
                                   f"batch {B}/GPU, req_outputs=theta,warp_mask"
                                   + (",consistency,poi" if args.consistency else ""),
                       "frames_per_gpu_per_step": B, "global_batch": B * world,
                       "parallelism": f"frame-sharded x{world}, all_gather(theta) over RCCL" if world > 1 else "single GPU"},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "kernel_groups": other,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
