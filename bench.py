#!/usr/bin/env python3
"""Headline benchmark: frames/sec of the hot path at 640x360, batch 16 per GPU.

Workload (BASELINE.json configs[1]): ``Reconstructor.predict(x, consistency=False,
project_poi=False)`` = UNet segmentation + ResNet34-STN + nearest homography warp of the
court template, on synthetic uint8-derived frames already resident in HBM.  Arithmetic: the
default "f16x3" mode (two fp16 planes per operand, three fp16 MFMAs per product, fp32 accumulation; see
csrc/conv_s3.hip and DESIGN.md section 2), "bf16x6" (three bf16 planes, six MFMAs) with SFH_PRECISION=bf16x6, or
fp32 MFMA throughout with SFH_PRECISION=fp32.
One "step" = one batch of 16 frames per GPU.  With N > 1 (launched by torch.distributed.run,
one process per GPU) every rank processes its own 16 frames (weak scaling) and the 3x3
thetas are all-gathered over RCCL each step.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     - live HIP-event timing of the DoubleConv 3x3 MFMA launches vs the matrix peak of the mode
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
# f16x3 mode (default): two fp16 planes per operand, 3 fp16 MFMA products per product (fp16 and bf16 MFMA have the
# same dense peak)
F16X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0
HBM_PEAK_TBS = 8.0              # MI355X_MICROARCH.md: HBM3E peak
# algorithmic work of one frame of the hot path at 640x360 (UNet 338.1 + ResNet34-STN 35.75)
STEP_GFLOP_PER_FRAME_640x360 = 338.1 + 35.75


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
    """(HBM bytes per launch of the dominant kernel, where the figure comes from): NOT measured in this run - read from
    the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, produced by profiles/collect_pmc.sh, which runs this
    same command under the counters in separate passes); (None, None) if absent."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        src = "profiles/pmc_traffic.json (%s, committed; rocprofv3 --pmc passes of this command, not this run)" % d.get("round", "r03")
        return d[tag]["hbm_bytes_per_launch"], src
    except (OSError, KeyError, ValueError):
        return None, None


class PowerSampler:
    """Socket power of one GPU while a region runs: a thread reads the amdgpu hwmon file (micro-watts; a plain file read,
    no subprocess) every `period` seconds.  None everywhere if the box exposes no such file for this device."""

    def __init__(self, pci_bus_id=None, period=0.02):
        import glob
        self.path, self.samples, self.period, self._stop, self._thr = None, [], period, False, None
        cands = []
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            for name in ("power1_average", "power1_input"):
                f = os.path.join(hw, name)
                if os.path.exists(f):
                    cands.append((os.path.realpath(os.path.join(hw, "..", "..")), f))
                    break
        if pci_bus_id is not None:
            want = str(pci_bus_id).lower()
            cands = [c for c in cands if os.path.basename(c[0]).lower().endswith(want)] or cands[:0]
        if len(cands) >= 1:
            self.path = cands[0][1]

    def _read(self):
        try:
            return int(open(self.path).read().strip()) * 1e-6
        except (OSError, ValueError):
            return None

    def __enter__(self):
        if self.path is None:
            return self
        import threading

        def loop():
            while not self._stop:
                v = self._read()
                if v is not None:
                    self.samples.append(v)
                time.sleep(self.period)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thr is not None:
            self._thr.join(timeout=1.0)
        return False

    def summary(self):
        if not self.samples:
            return None
        v = sorted(self.samples)
        return {"samples": len(v), "mean_w": round(sum(v) / len(v), 1), "max_w": round(v[-1], 1), "min_w": round(v[0], 1),
                "source": self.path}


def device_calibration(dev, ms_target=30.0):
    """What THIS device sustains, so that lines from different devices of the pool compare (they differ by up to 4 % in the
    clock they hold under an MFMA-dense load): a register-resident v_mfma_f32_16x16x32_f16 loop over every CU for about
    `ms_target` ms (sfh_probe_mfma_f16: random fp16 mantissas, two workgroups per CU, four rounds) -> TFLOP/s of fp16 MFMA
    work, the clock the chip held INSIDE the kernel (s_memtime / s_memrealtime) and the socket power while it ran."""
    import ctypes
    import torch
    from sfh_amd import _lib
    lib = _lib.load()
    props = torch.cuda.get_device_properties(dev)
    wgs = 2 * 4 * int(props.multi_processor_count)
    out = torch.empty(wgs * 256, dtype=torch.float32, device=dev)
    clk = torch.zeros(2, dtype=torch.int64, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())

    def run(iters):
        clk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.sfh_probe_mfma_f16(int(iters), wgs, p(out), p(clk), st), "probe_mfma_f16")
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)
    ms = run(300)                                   # warm-up, and the rate to size the timed launch from
    iters = max(300, min(1 << 18, int(300 * ms_target / max(ms, 1e-3))))
    bus = None
    try:
        bus = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    except AttributeError:
        pass
    # five launches back to back: the rate of the best one; the hwmon power reading is a moving average that lags by tens of
    # milliseconds, so it is sampled over all five (its maximum is the figure to read)
    with PowerSampler(bus, period=0.004) as ps:
        ms = min(run(iters) for _ in range(5))
    c = clk.cpu().tolist()
    flops = wgs * 4.0 * iters * 64 * 2 * 16 * 16 * 32
    return {"kernel": "register-resident v_mfma_f32_16x16x32_f16 loop, random fp16 mantissas, %d workgroups of 4 waves" % wgs,
            "ms": round(ms, 3), "launches": 5, "mfma_f16_tflops": round(flops / (ms * 1e-3) / 1e12, 1),
            "frac_of_2500": round(flops / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
            "in_kernel_clock_ghz": round(0.1 * c[0] / c[1], 3) if c[1] else None,
            "power": ps.summary(), "device": props.name, "compute_units": int(props.multi_processor_count),
            "pci": bus}


def c2_parity(net, dev):
    """The headline workload's parity figures IN the bench line (N = 1, 640x360, batch 16): predict() on the seed-0 batch
    against the golden vector of the reference's own classes (tests/golden/c2_640x360_b16.npz, data only): arg-max pixels
    that differ and the largest golden top-2 margin among them, max |d theta|, sub-sampled logits."""
    import numpy as np
    import torch
    from sfh_amd import synth
    path = os.path.join(ROOT, "tests", "golden", "c2_640x360_b16.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    B, H, W = 16, 360, 640
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
    with torch.no_grad():
        out = net.predict(x, consistency=True, project_poi=True)
    logits = out["logits"].cpu()
    am = logits.argmax(1).numpy().astype(np.uint8)
    bits = np.unpackbits(g["argmax_2bit"], axis=-1).reshape(B, H, W, 2)
    am_ref = (bits[..., 0] * 2 + bits[..., 1]).astype(np.uint8)
    diff = np.argwhere(am != am_ref)
    margin = np.full((B, H * W), np.inf, np.float32)
    margin[g["low_margin_frame"], g["low_margin_pixel"]] = g["low_margin_value"]
    margin = margin.reshape(B, H, W)
    dm = margin[diff[:, 0], diff[:, 1], diff[:, 2]] if len(diff) else np.zeros(0, np.float32)
    return {"against": "tests/golden/c2_640x360_b16.npz (the reference's own UNet / ResNetSTN classes on torch CPU fp32, 16 frames)",
            "argmax_pixels": int(B * H * W), "argmax_differ": int(len(diff)),
            "largest_golden_top2_margin_among_differing": float(dm.max()) if len(diff) else 0.0,
            "golden_pixels_inside_softmax_tie_margin_1.2e-7": int((g["low_margin_value"] < 1.2e-7).sum()),
            "max_abs_dtheta": float((out["theta"].cpu() - torch.from_numpy(g["theta"])).abs().max()),
            "max_abs_dlogits_every_16th_pixel": float((logits[:, :, 4::16, 4::16] - torch.from_numpy(g["logits_sub"])).abs().max()),
            "max_abs_dpoi": float((out["poi"].cpu() - torch.from_numpy(g["poi"])).abs().max()),
            "max_abs_dconsist": float((out["consist_score"].cpu() - torch.from_numpy(g["consist"])).abs().max()),
            "note": "north_star asks for bit-exact arg-max: met up to summation order - every differing pixel is a near-tie of "
                    "the golden logits (margin above), the fp32-MFMA mode flips the same pixels (DESIGN.md section 5)"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) through
    torch.distributed.run as a CHILD process and relay its output.  This process never touches the GPU
    (no torch import, no HIP call), so nothing is re-executed under an initialised device."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


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

    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()      # peak_mem_gib is this config's own (other configs ran in this process before)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return ({
        "metric": "training steps: frames/sec at %dx%d batch=%d (forward + losses + backward + clip + RMSprop)" % (W, H, B),
        "value": round(B * args.steps / el, 2), "unit": "frames/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 2), "higher_is_better": True,
        "dtype": {"f16x3": "f16x3->f32 convs: forward, backward-data and 3x3 backward-filter on the fp16 matrix cores (two-plane "
                           "fp16 operands, three products, fp32 accumulation; gradients carried at a power-of-two scale)",
                  "bf16x6": "bf16x6->f32 convs, forward, backward-data and 3x3 backward-filter on the bf16 matrix cores (fp32-equivalent)",
                  "fp32": "f32 (fp32 MFMA)"}[os.environ.get("SFH_TRAIN_PRECISION", "f16x3")],
        "range_fallbacks": int(getattr(ts, "range_fallbacks", 0)) if ts is not None else None,
        "range_rescales": int(getattr(ts, "range_rescales", 0)) if ts is not None else None,
        "grad_scale_shift": int(getattr(ts, "grad_scale_shift", 0)) if ts is not None else None,
        "data": "synthetic", "final_loss": float(loss.detach()),
        "losses_and_optimizer": "torch ops (caller side)" if args.train_autograd else "HIP kernels (training.TrainStep)",
        "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2**30, 2),
        "config": {"workload": "BASELINE config 3: Reconstructor training step, CE + SmoothL1 + RRMSE + consistency CE"}})


def _timed_predicts(net, x, n, warm, pipeline, **kw):
    """n timed predict() calls on one resident batch (pipeline: predict_async with two batches in flight) -> seconds"""
    import torch

    def run(m):
        if not pipeline:
            for _ in range(m):
                net.predict(x, **kw)
            return
        prev = None
        for _ in range(m):
            h = net.predict_async(x, **kw)
            if prev is not None:
                prev.result()
            prev = h
        prev.result()
    with torch.no_grad():
        run(warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        return time.perf_counter() - t0


def e2e_bench(net, B, W, H, src_wh, steps, warmup):
    """predict.py's loop shape end to end (predict.py:57-122): uint8 HWC frames in PINNED HOST memory -> H2D on a copy
    stream -> /255 + HWC->CHW (+ integer INTER_AREA downscale) on the GPU -> predict_async -> uint8 arg-max mask, uint8
    warp mask, theta, consistency score -> D2H into pinned host arrays, two batches in flight (sfh_amd.pipeline).
    -> (frames/s, ms per batch): PCIe-inclusive, never the headline `value`."""
    import torch
    from sfh_amd import synth
    from sfh_amd.pipeline import FramePipeline
    sw, sh = src_wh
    pipe = FramePipeline(net, B, (sh, sw), req_outputs=("theta", "warp_mask", "segm_mask"), consistency=True)
    host = [torch.from_numpy(synth.synth_frames_u8(B, sh, sw, seed=500 + k)).pin_memory() for k in range(2)]

    def run(n):
        got = 0
        for res in pipe.run(host[k % 2] for k in range(n)):
            got += res["theta"].shape[0]
        return got
    with torch.no_grad():
        run(warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = run(steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    assert got == B * steps
    return B * steps / el, el / steps * 1e3


def extra_configs(args):
    """BASELINE configs 3 and 5 on this GPU, a few steps each (driver-observed numbers for every
    single-GPU config in one default run): C5 = predict() with consistency + POI on 16 frames of
    1280x720 against the 4-class pitch template; C3 = one training step per batch of 16 at 640x360; and C2 again
    in the two other arithmetic modes (exact three-plane bf16 operands, fp32 MFMA)."""
    import copy
    import torch
    from sfh_amd import synth
    from sfh_amd.reconstructor import Reconstructor
    res = {}
    dev = torch.device("cuda", 0)
    # C2 (the headline workload) in the other two arithmetic modes: "bf16x6" carries every fp32 operand exactly
    # (six bf16 MFMA products), "fp32" is the fp32 MFMA throughout
    B, W, H = 16, 640, 360
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev)
    poi = synth.load_court_poi("pitch", B).to(dev)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
    net.to(dev).eval()
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
    # the headline workload END TO END (host uint8 frames in, host results out; PCIe inclusive), next to the same loop
    # device to device measured right before it on this device
    n = 10
    d2d = _timed_predicts(net, x, n, 3, True, consistency=True)
    for tag, src in (("E2E_640x360_batch16", (W, H)), ("E2E_1920x1080_to_640x360_batch16", (3 * W, 3 * H))):
        fps, ms = e2e_bench(net, B, W, H, src, n, 3)
        res[tag] = {"value": round(fps, 2), "unit": "frames/s", "ms_per_step": round(ms, 3), "steps": n, "warmup": 3,
                    "device_to_device_frames_per_s": round(B * n / d2d, 2), "fraction_of_device_to_device": round(fps * d2d / (B * n), 4),
                    "workload": ("uint8 HWC %dx%d frames in pinned host memory -> H2D (copy stream) -> /255 + CHW%s -> predict_async("
                                 "consistency=True) -> uint8 arg-max mask + uint8 warp mask + theta + score -> D2H (pinned), two batches "
                                 "in flight (sfh_amd.pipeline.FramePipeline; predict.py:57-122)")
                                % (src[0], src[1], "" if src == (W, H) else " + 3x3 INTER_AREA downscale on the GPU")}
    # predict.py's default geometry (predict.py:151-155): UNet at 640x360, court / warp raised to out_size 1280x720, the
    # consistency CE through the nearest-resized mask, 33-point POI
    court_hd = synth.load_court_template("ncaa_nc4_1280x720", 4, B).to(dev)
    net_d = Reconstructor(court_hd, poi, target_size=(W, H), unet_size=(W, H), warp_size=(2 * W, 2 * H), warp_with_nearest=True)
    net_d.load_state_dict(synth.synth_state_dict(net_d.state_dict(), 0))
    net_d.to(dev).eval()
    n = 6
    el = _timed_predicts(net_d, x, n, 2, not args.no_pipeline, consistency=True, project_poi=True)
    res["C2d_unet640x360_warp1280x720_batch16"] = {
        "value": round(B * n / el, 2), "unit": "frames/s", "ms_per_step": round(el / n * 1e3, 3), "steps": n, "warmup": 2,
        "workload": "predict.py's default geometry: predict(consistency=True, project_poi=True), UNet 640x360, court / warp 1280x720 "
                    "(NCAA template resized NEAREST), consistency through the nearest-resized warp mask, batch 16"}
    del net_d, court_hd
    torch.cuda.empty_cache()
    # small batches (the reference's default batchsize is 8, utils/config.py:30; one frame = the latency case): predict() per
    # batch.  These are GPU-bound, not host-bound: at one frame the ~90 dependent launches keep the GPU busy 95 % of the batch
    # (profiles/r06_batch1_launch_table.txt), so a launch list / HIP graph has at most 0.1-0.2 ms to return
    for bs in (1, 8):
        n = 40 if bs == 1 else 12
        xb = x[:bs].contiguous()
        el = _timed_predicts(net, xb, n, 4, False, consistency=False)
        with torch.no_grad():
            for _ in range(4):
                net.predict_replay(xb, consistency=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                net.predict_replay(xb, consistency=False)
            torch.cuda.synchronize()
            el_g = time.perf_counter() - t0
        el_a = _timed_predicts(net, xb, n, 4, True, consistency=False)
        res[f"C2_640x360_batch{bs}"] = {
            "value": round(bs * n / el, 2), "unit": "frames/s", "ms_per_step": round(el / n * 1e3, 3), "steps": n, "warmup": 4,
            "ms_per_step_graph_replay": round(el_g / n * 1e3, 3),
            "ms_per_step_async": round(el_a / n * 1e3, 3), "value_async": round(bs * n / el_a, 2),
            "workload": f"the headline workload at batch {bs}: predict() per batch (no pipelining), 640x360, theta + warp_mask"}
    for prec in ("bf16x6", "fp32"):
        net.precision = prec
        n = 4
        el = _timed_predicts(net, x, n, 2, not args.no_pipeline, consistency=False)
        res[f"C2_640x360_batch16_{prec}"] = {
            "value": round(B * n / el, 2), "unit": "frames/s", "ms_per_step": round(el / n * 1e3, 2), "steps": n, "warmup": 2,
            "precision": prec, "workload": "the headline workload (predict(), 640x360, batch 16, theta + warp_mask) in this arithmetic mode"}
        net.invalidate_engines()
        torch.cuda.empty_cache()
    del net, x, court, poi
    torch.cuda.empty_cache()
    B, W, H = 16, 1280, 720
    court = synth.load_court_template("pitch_v3_nc4_1280x720", 4, B).to(dev)
    poi = synth.load_court_poi("pitch", B).to(dev)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
    net.to(dev).eval()
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
    n = 4
    el = _timed_predicts(net, x, n, 2, not args.no_pipeline, consistency=True, project_poi=True)
    res["C5_1280x720_batch16_pitch_template_poi"] = {
        "value": round(B * n / el, 2), "unit": "frames/s", "ms_per_step": round(el / n * 1e3, 2), "steps": n, "warmup": 2,
        "workload": "predict(consistency=True, project_poi=True), 1280x720, batch 16, pitch_mask_v3_nc4_hd template, 33-point POI"}
    # the homography warp at this size against HBM (SURVEY 8d algorithmic bytes), launches alone on the chip
    from sfh_amd import engine
    tm = engine.ConvTimer()
    engine.PackedConv.timer = tm
    with torch.no_grad():
        for _ in range(3):
            net.predict(x, consistency=True, project_poi=True)
    torch.cuda.synchronize()
    engine.PackedConv.timer = None
    for wtag in ("warp", "warp+ce"):
        wv = tm.summary().get(wtag)
        if wv:
            tbs = wv[1] / (wv[2] * 1e-3) / 1e12
            res["C5_1280x720_batch16_pitch_template_poi"][wtag] = {
                "launches": wv[0], "bound": "hbm", "us_per_launch": round(wv[2] * 1e3 / wv[0], 2),
                "algorithmic_bytes_per_launch": int(wv[1] / wv[0]), "tb_per_s": round(tbs, 3), "frac": round(tbs / HBM_PEAK_TBS, 4),
                "kernel": ("warpce_kernel: nearest warp + consistency CE in one launch (logits + mask + template bytes)"
                           if wtag == "warp+ce" else "warp2_kernel")}
    del net, x, court, poi
    torch.cuda.empty_cache()
    a = copy.copy(args)
    a.batch, a.width, a.height, a.steps, a.warmup, a.train_autograd = 16, 640, 360, 4, 2, False
    t = train_bench(a)
    res["C3_train_step_640x360_batch16"] = {k: t[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype",
                                                              "losses_and_optimizer", "peak_mem_gib", "final_loss",
                                                              "range_fallbacks", "range_rescales")}
    res["C3_train_step_640x360_batch16"]["workload"] = t["config"]["workload"]
    return res


def main():
    """_run() with the process group torn down on EVERY exit path (an exception or SystemExit after
    init_process_group must not leave RCCL's communicator and its proxy thread behind)"""
    try:
        _run()
    finally:
        if "torch.distributed" in sys.modules:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                dist.destroy_process_group()


def _run():
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
    ap.add_argument("--consistency", action="store_true",
                    help="also compute consist_score + poi (N > 1 always computes consist_score: BASELINE config 4 gathers it)")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE config 3 instead of the headline: one training step (forward, losses, "
                         "backward, clip, RMSprop) per batch; prints its own JSON line")
    ap.add_argument("--train-autograd", action="store_true",
                    help="with --train: losses, clip and RMSprop as the caller's torch ops around the model's "
                         "autograd node (the reference's train.py structure) instead of the all-HIP TrainStep")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend (nccl = RCCL over xGMI; gloo only to rehearse the multi-rank path)")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the exchange step's collective even with ONE rank: `--gpus 1 --force-collective` initialises a "
                         "one-rank process group on --dist-backend (nccl = RCCL) and gathers theta + consist_score through the "
                         "real all_gather_into_tensor on the side stream, exactly as N > 1 does; gather_check is then non-null")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --dist-backend gloo; the value "
                         "then measures nothing)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="call predict() per step instead of predict_async() (two batches in flight: the ResNet-STN + warp of "
                         "batch k on a side stream under the UNet of batch k + 1)")
    ap.add_argument("--no-alone-pass", action="store_true",
                    help="skip the unpipelined pass behind the timed region (kernel_groups then come from the pipelined region): "
                         "for a profiler trace of the pipelined region only")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu: rehearsal of the multi-rank plumbing in the CPU tests (tests/test_sharding.py, gloo, a stand-in "
                         "model); the printed value then measures nothing")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short C3 (training step) and C5 (1280x720) measurements appended at N=1")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    if args.train or args.train_autograd:
        print(json.dumps(train_bench(args)), flush=True)
        return

    import torch
    import torch.distributed as dist
    from sfh_amd import synth, engine, sharding
    from sfh_amd.reconstructor import Reconstructor

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.share_gpu:
        if args.dist_backend != "gloo":
            raise SystemExit("--share-gpu is a rehearsal mode and needs --dist-backend gloo (RCCL wants one GPU per rank)")
        local_rank = 0
    on_gpu = args.device == "cuda"
    if not on_gpu and args.dist_backend != "gloo":
        raise SystemExit("--device cpu is a rehearsal mode and needs --dist-backend gloo")
    if on_gpu:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    use_pg = world > 1 or args.force_collective
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:        # `--gpus 1 --force-collective` without a launcher: a one-rank world of our own
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(so.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    B, W, H = args.batch, args.width, args.height
    # BASELINE config 4 (N > 1) names theta + consistency: the sharded run computes the score that its gather carries
    cons = bool(args.consistency or use_pg)
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
    # the one exchange step of the sharded path: theta (+score) rows to every rank, all-gathered on a side stream
    # while the next batch's kernels run (sharding.ResultGather; the final device synchronize covers it)
    gather = sharding.ResultGather(world, B, dev, depth=2, force_collective=args.force_collective) if use_pg else None

    pending = []    # predict_async handles: at most two batches in flight
    last = {}       # slot + theta of the newest gathered step (checked after the timed region)

    def exchange(out):
        if gather is not None:
            last["slot"], last["theta"] = gather.submit(out["theta"], out.get("consist_score")), out["theta"]

    def finish(h):
        out = h.result()
        exchange(out)
        return out

    def step(k, last=False):
        """one batch through the hot path.  Pipelined (default): submit batch k, then take the result of batch k - 1, so
        that the small launches behind the UNet (ResNet-STN, warp, CE) run under the next batch's UNet; `last` drains."""
        x = frames[k % nbatches]
        if args.no_pipeline:
            out = net.predict(x, consistency=cons, project_poi=args.consistency)
            exchange(out)
            return out
        pending.append(net.predict_async(x, consistency=cons, project_poi=args.consistency))
        out = None
        while len(pending) > (0 if last else 1):
            out = finish(pending.pop(0))
        return out

    calib = device_calibration(dev) if on_gpu else None
    bus = calib.get("pci") if calib else None
    with torch.no_grad():
        for k in range(args.warmup):
            step(k, last=(k == args.warmup - 1))
        sync()
        if use_pg:
            dist.barrier()
        # live HIP events around the launches of the dominant kernel (roofline); the other kernel groups are timed in the
        # unpipelined pass behind the region when there is one (two event records per launch are not free: all 59 timed
        # launches of a step cost about 0.2 ms of the step)
        will_time_alone = not args.no_pipeline and not args.no_alone_pass
        timer = engine.ConvTimer(only={"doubleconv3x3"} if will_time_alone else None)
        engine.PackedConv.timer = timer
        power = PowerSampler(bus) if on_gpu else None
        t0 = time.perf_counter()
        if power is not None:
            power.__enter__()
        for k in range(args.steps):
            out = step(k, last=(k == args.steps - 1))      # the K-th call drains the pipeline: exactly K batches are timed
        sync()
        own_elapsed = time.perf_counter() - t0             # this rank's own K steps (before it waits for the others)
        if use_pg:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if power is not None:
            power.__exit__()
        engine.PackedConv.timer = None

    # the exchange step, checked once outside the timed region: every rank holds world * B gathered rows, and its own
    # rows are the theta it computed in the last step
    gather_check = None
    if gather is not None:
        th_all, _ = gather.result(last["slot"])
        ok = torch.tensor([int(tuple(th_all.shape) == (world * B, 1, 3, 3)
                               and torch.equal(th_all[rank * B:(rank + 1) * B], last["theta"]))], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        gather_check = {"rows_per_rank": int(th_all.shape[0]), "own_rows_equal_own_theta_on_every_rank": bool(ok.item()),
                        "bytes_per_step_per_rank": 40 * B, "backend": dist.get_backend(),
                        "collective": "all_gather_into_tensor on a side stream, one per step",
                        "collectives_run": int(gather.collectives_run),
                        "forced_on_one_rank": bool(world == 1)}

    # the same K steps once more WITHOUT the pipeline: per-kernel durations of launches that run alone on the chip (under
    # the pipeline the ResNet-STN launches of batch k share the CUs with the UNet launches of batch k + 1, so every launch
    # of the timed region above takes longer than it would alone although the batch takes less)
    alone = None
    if not args.no_pipeline and not args.no_alone_pass:
        with torch.no_grad():
            # (a) the drop-in predict() rate: no event timers in this pass (two records per launch on 59 launches cost 0.2 ms)
            t1 = time.perf_counter()
            for k in range(args.steps):
                net.predict(frames[k % nbatches], consistency=cons, project_poi=args.consistency)
            sync()
            el2 = time.perf_counter() - t1
            # (b) the same steps with every launch timed: per-kernel durations alone on the chip
            tm2 = engine.ConvTimer()
            engine.PackedConv.timer = tm2
            for k in range(args.steps):
                net.predict(frames[k % nbatches], consistency=cons, project_poi=args.consistency)
            sync()
            engine.PackedConv.timer = None
        alone = (tm2.summary(), el2)
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if use_pg:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = el.item()
    # per-rank figures: each rank's own step time and what its device sustains - a 4 % spread between the pool's devices
    # would otherwise read as a scaling loss (value = all frames / the SLOWEST rank's time)
    mine = {"rank": rank, "ms_per_step": round(own_elapsed / args.steps * 1e3, 3),
            "device": calib["device"] if calib else str(dev),
            "mfma_f16_tflops": calib["mfma_f16_tflops"] if calib else None,
            "in_kernel_clock_ghz": calib["in_kernel_clock_ghz"] if calib else None,
            "power_w_timed_region": (power.summary() or {}).get("mean_w") if power is not None else None}
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    rms = sorted(r["ms_per_step"] for r in ranks)
    per_rank = {"ms_per_step_min": rms[0], "ms_per_step_median": rms[len(rms) // 2] if len(rms) % 2 else
                round(0.5 * (rms[len(rms) // 2 - 1] + rms[len(rms) // 2]), 3), "ms_per_step_max": rms[-1], "ranks": ranks}
    frames_total = B * args.steps * world
    fps = frames_total / elapsed

    summ = timer.summary()
    n, fl, ms = summ.get("doubleconv3x3", (0, 0.0, 1.0))
    achieved = fl / (ms * 1e-3) / 1e12 if n else 0.0
    prec = net.precision
    nprod = {"bf16x6": 6, "f16x3": 3}.get(prec)
    peak = {"bf16x6": BF16X6_PEAK_TFLOPS, "f16x3": F16X3_PEAK_TFLOPS}.get(prec, FP32_MFMA_PEAK_TFLOPS)
    x6 = nprod is not None
    half = "fp16" if prec == "f16x3" else "bf16"
    traffic, traffic_src = pmc_traffic("doubleconv3x3") if (W, H, B) == (640, 360, 16) else (None, None)
    # frac = ALGORITHMIC FLOPs / launch time / the dense MFMA peak of the dtype the launches execute in.  The
    # split-operand modes execute nprod MFMA products per algorithmic product: that executed-work share of the matrix
    # pipe is `mfma_utilisation` (north_star's "MFMA util"), never `frac`.
    mfma_peak = BF16_MFMA_PEAK_TFLOPS if x6 else FP32_MFMA_PEAK_TFLOPS
    talg = timer.traffic().get("doubleconv3x3")
    talg = talg / max(n, 1) if talg else None
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(mfma_peak, 1),
                "unit": "TFLOP/s", "frac": round(achieved / mfma_peak, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "traffic_algorithmic": round(talg) if talg else None,
                "traffic_ratio": round(traffic / talg, 3) if (traffic and talg) else None,
                "traffic_algorithmic_basis": "per launch: source tensor(s) read once at 4 B per element (two fp16 planes), result (+ "
                                             "pooled copy) written once, packed weights read once; averaged over the timed launches",
                "mfma_utilisation": round(achieved * (nprod or 1) / mfma_peak, 4),
                "fp32_grade_peak": round(peak, 1), "frac_of_fp32_grade_peak": round(achieved / peak, 4),
                "peak_basis": (f"2500 TFLOP/s dense {half} MFMA (MI355X_MICROARCH.md).  Every fp32-grade product is {nprod} {half} MFMA "
                               f"products, so the launches execute {nprod}x the algorithmic FLOPs: {achieved * (nprod or 1):.0f} TFLOP/s of "
                               f"{half} MFMA work = mfma_utilisation; the most algorithmic work this arithmetic could reach is "
                               f"2500 / {nprod} = fp32_grade_peak"
                               if x6 else "157.3 TFLOP/s dense fp32 MFMA (v_mfma_f32_16x16x4_f32)"),
                "kernel": ((f"conv_s3_kernel<3x3, {'2 fp16' if prec == 'f16x3' else '3 bf16'} planes> (DoubleConv, split operands)")
                           if x6 else "conv_mfma_kernel<3x3,s1> (DoubleConv)"),
                "frac_meaning": "frac = achieved (algorithmic 2 x MAC of the reference conv / launch time) / peak: SURVEY 8(d)",
                "launches": n,
                "avg_launch_ms": round(ms / max(n, 1), 4),
                "algorithmic_gflop_per_launch": round(fl / max(n, 1) / 1e9, 2),
                "share_of_step_time": round(ms * 1e-3 / elapsed, 4)}
    if alone is not None:
        n2, fl2, ms2 = alone[0].get("doubleconv3x3", (0, 0.0, 1.0))
        a2 = fl2 / (ms2 * 1e-3) / 1e12 if n2 else 0.0
        roofline["note"] = ("timed region = predict_async(): launches of two batches share the chip, so a launch's duration here "
                            "includes what it gives to the other batch's launches; `unpipelined` = the same kernel timed over "
                            "the same number of predict() steps right after, alone on the chip")
        roofline["unpipelined"] = {"achieved": round(a2, 2), "frac": round(a2 / mfma_peak, 4),
                                   "mfma_utilisation": round(a2 * (nprod or 1) / mfma_peak, 4), "avg_launch_ms": round(ms2 / max(n2, 1), 4),
                                   "ms_per_step": round(alone[1] / args.steps * 1e3, 3),
                                   "frames_per_s": round(B * args.steps / alone[1], 2)}
    # every timed kernel group against the roofline that bounds it: conv groups against the matrix peak of the mode
    # (algorithmic FLOPs), the homography warp against HBM (algorithmic bytes, SURVEY 8d: B*h*w*4 + Ht*Wt*4 + 36*B)
    other = {}
    executed = (tm2 if alone is not None else timer).executed()
    for t, v in (alone[0] if alone is not None else summ).items():      # (pipelined run: from the unpipelined pass, see above)
        if t in ("warp", "warp+ce"):
            tbs = v[1] / (v[2] * 1e-3) / 1e12
            other[t] = {"launches": v[0], "bound": "hbm", "us_per_launch": round(v[2] * 1e3 / v[0], 2),
                        "algorithmic_bytes_per_launch": int(v[1] / v[0]), "tb_per_s": round(tbs, 3),
                        "frac": round(tbs / HBM_PEAK_TBS, 4), "ms_per_step": round(v[2] / args.steps, 4)}
            if t == "warp":
                # north_star's >= 60 % HBM roofline on grid_sample, stated where it can and cannot hold
                other[t].update({
                    "ceiling_frac": 0.47 if (W, H, B) == (640, 360, 16) else None,
                    "ceiling_basis": "a STORE-ONLY kernel of this launch's shape reaches 0.47 of 8 TB/s at 640x360 x 16 (15.7 MB = 2 us of "
                                     "traffic behind ~6 us of launch and ramp: profiles/r02_warp_variants.txt); the warp itself is vector-issue "
                                     "bound at 32-40 instructions per pixel (0.51-0.58 at batch 1024, profiles/r04_warp_pmc.txt)",
                    "target_met_as": "warp + consistency CE fused (sfh_warp_consistency_fwd, predict(consistency=True)): 0.64 of 8 TB/s at "
                                     "640x360 x 16 and 0.75 at 1280x720 x 16 for the fused kernel alone, 0.50 / 0.64 with its one-wave-per-frame "
                                     "final launch (profiles/r05_warpce_pmc.txt); live figure of this run: other_configs.C5...['warp+ce']"})
        else:
            tf = v[1] / (v[2] * 1e-3) / 1e12
            other[t] = {"launches": v[0], "bound": "mfma", "tflops": round(tf, 2), "frac": round(tf / mfma_peak, 4),
                        "mfma_utilisation": round(tf * (nprod or 1) / mfma_peak, 4), "ms_per_step": round(v[2] / args.steps, 3)}
            if t in executed:      # credited with the reference's work, executes less (composed 2x2 Up conv: 8/9)
                ex = executed[t] / (v[2] * 1e-3) / 1e12
                basis = ("credited = the u-half of the reference's 3x3 conv (9 taps x C; the ConvTranspose2d it also replaces is not "
                         "credited); executed = 4 taps x 2C on the low-resolution tensor") if t != "upfused" else \
                        ("single-kernel Up block (csrc/conv_upfused.hip: u3, u4): credited = the reference's whole 3x3 conv over skip + "
                         "up-sampled channels (9 taps x 2C; its ConvTranspose2d is not credited); executed = 9 taps x C on the skip "
                         "tensor + 4 taps x 2C on the low-resolution tensor; these launches are not in `doubleconv3x3` / `fusedup2x2`")
                other[t].update({"frac_basis": basis,
                                 "executed_tflops": round(ex, 2), "mfma_utilisation": round(ex * (nprod or 1) / mfma_peak, 4)})
    step_gflop = STEP_GFLOP_PER_FRAME_640x360 * (W * H) / (640.0 * 360.0) * B
    whole_tf = step_gflop * 1e9 * args.steps / elapsed / 1e12    # per GPU: every rank runs its own batch per step

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_ref
        ncpu = min(usable_cores(), int(os.environ.get("SFH_CPU_THREADS", "64")))
        torch.set_num_threads(ncpu)
        nf = min(args.cpu_frames, B)
        sd_cpu = {k: v.cpu() for k, v in sd.items()}
        court_c, poi_c = court.cpu(), poi.cpu()
        samples, dth = [], 0.0
        with torch.no_grad():
            torch_ref.predict(frames[0][:1].cpu(), sd_cpu, court_c, poi_c, warp_size=(W, H), unet_size=(W, H),
                              target_size=(W, H), consistency=False)  # warm-up (1 frame)
            for k in range(min(2, nbatches)):       # two different batches, reported separately and together
                xc = frames[k][:nf].cpu()
                t0 = time.perf_counter()
                ref = torch_ref.predict(xc, sd_cpu, court_c, poi_c, warp_size=(W, H), unet_size=(W, H),
                                        target_size=(W, H), consistency=False)
                samples.append(time.perf_counter() - t0)
                got = net.predict(frames[k][:nf], consistency=False)
                dth = max(dth, (got["theta"].cpu() - ref["theta"]).abs().max().item())
        model = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        cpu_baseline = {"value": round(nf * len(samples) / sum(samples), 4), "unit": "frames/s", "cores": ncpu, "kind": "port",
                        "cores_note": f"{ncpu} = every core this process may use (affinity mask capped by the cgroup CPU quota) of the "
                                      f"host's {os.cpu_count()} logical CPUs: the one-GPU lease exposes that share, not the socket",
                        "sample": f"{len(samples)} batches of {nf} frames of the same {W}x{H} workload, 1 warm-up frame, "
                                  f"torch {torch.__version__} CPU fp32, {ncpu} threads, {model}",
                        "per_batch_frames_per_s": [round(nf / t, 4) for t in samples],
                        "max_abs_dtheta_gpu_vs_cpu": dth}

    # the other single-GPU BASELINE configs, a few steps each, AFTER the headline region (N = 1 only): the
    # headline fields above are not touched by them
    parity = None
    if rank == 0 and world == 1 and on_gpu and (W, H, B) == (640, 360, 16) and prec in ("f16x3", "bf16x6", "fp32"):
        parity = c2_parity(net, dev)
    other_configs = None
    if rank == 0 and world == 1 and not args.no_extra_configs and (W, H, B) == (640, 360, 16):
        del frames, out
        net.invalidate_engines()
        torch.cuda.empty_cache()
        other_configs = extra_configs(args)

    if rank == 0:
        line = {
            "metric": "frames/sec at 640x360 batch=16 (1/2/4/8 GPU) + homography L1 vs ref",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "value_predict_sync": (round(B * args.steps / alone[1], 2) if alone is not None else round(fps, 2)) if world == 1 else None,
            "value_exact_operands": ((other_configs or {}).get("C2_640x360_batch16_bf16x6") or {}).get("value"),
            "value_exact_operands_note": "the same workload with every fp32 operand carried EXACTLY (bf16x6: three bf16 planes, six MFMA "
                                         "products); `value` is the default two-plane fp16 emulation (22 significand bits, DESIGN.md section 2)",
            "value_per_calibrated_pflop": (round(fps / world / (calib["mfma_f16_tflops"] / 1000.0), 2)
                                           if calib and calib.get("mfma_f16_tflops") else None),
            "value_per_calibrated_pflop_note": "frames/s per GPU / (device_calibration.mfma_f16_tflops / 1000): comparable across the "
                                               "pool's devices, which hold different clocks under an MFMA-dense load",
            "value_predict_sync_note": "frames/s of the drop-in predict() called per batch (no predict_async pipelining), "
                                       "same kernels, timed right after the headline region; N = 1 only",
            "dtype": {"bf16x6": "bf16x6->f32 (3-way bf16 split operands, 6 bf16 MFMA products, fp32 accumulate; fp32-equivalent)",
                      "f16x3": "f16x3->f32 (2-way fp16 split operands = 22 significand bits, 3 fp16 MFMA products, fp32 "
                               "accumulate; end-to-end error at the level of an fp32 run, see DESIGN.md section 2)"}.get(prec, "f32"),
            "data": "synthetic",
            "config": {"workload": f"predict(): UNet seg + ResNet34-STN + nearest warp, {W}x{H}, "
                                   f"batch {B}/GPU, req_outputs=theta,warp_mask"
                                   + (",consistency,poi" if args.consistency else ",consistency" if cons else ""),
                       "frames_per_gpu_per_step": B, "global_batch": B * world,
                       "pipeline": ("none: predict() per step" if args.no_pipeline else
                                    "predict_async(): two batches in flight, ResNet-STN + warp of batch k on a side stream under the UNet of batch k + 1"),
                       "precision": prec, "range_fallbacks": int(getattr(net, "range_fallbacks", 0)),
                       "range_rescales": int(getattr(net, "range_rescales", 0)),
                       "range_raises": int(getattr(net, "range_raises", 0)),
                       "parallelism": (f"frame-sharded x{world}, all_gather_into_tensor(theta + consist_score) over "
                                       + ("RCCL" if args.dist_backend == "nccl" else "gloo (REHEARSAL, ranks share a GPU)" if args.share_gpu else "gloo")
                                       if world > 1 else
                                       ("single GPU; exchange step forced through a one-rank " +
                                        ("RCCL" if args.dist_backend == "nccl" else "gloo") + " process group (--force-collective)")
                                       if use_pg else "single GPU")},
            "roofline": roofline,
            "whole_step": {"algorithmic_gflop_per_step_per_gpu": round(step_gflop, 1), "tflops_per_gpu": round(whole_tf, 2),
                           "frac": round(whole_tf / mfma_peak, 4), "mfma_utilisation": round(whole_tf * (nprod or 1) / mfma_peak, 4),
                           "note": "UNet + ResNet34-STN algorithmic FLOPs of a batch / ms_per_step; frac against the dense MFMA peak of the "
                                   "dtype executed (2500), mfma_utilisation = executed MFMA work (x products per element) / that peak"},
            "cpu_baseline": cpu_baseline,
            "parity": parity,
            "device_calibration": dict(calib, power_timed_region=(power.summary() if power is not None else None)) if calib else None,
            "per_rank": per_rank,
            "gather_check": gather_check,
            "kernel_groups": other,
            "kernel_groups_measured_in": ("the unpipelined pass after the timed region (launches alone on the chip)"
                                          if alone is not None else "the timed region"),
            "other_configs": other_configs,
        }
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
