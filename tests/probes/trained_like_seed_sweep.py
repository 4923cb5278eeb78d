"""Full-size parity over several TRAINED-LIKE checkpoints (synth.trained_like_state_dict: BatchNorm variances over six decades,
negative / dead gammas, weight outliers, one layer scaled 2^+-12, the regression head at trained strength): for each seed
4 frames of 640x360, HIP predict() in f16x3 / bf16x6 / fp32 against the CPU restatement.
Prints, per seed and mode: max |dtheta|, max |dlogits|, arg-max pixels that differ and the largest top-2 margin among
them, POI pixels that differ, nearest warp vs oracle-warp(GPU theta).   (test infrastructure: imports the oracle)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import torch_ref, warp_ref  # noqa: E402
from sfh_amd import synth  # noqa: E402
from sfh_amd.reconstructor import Reconstructor  # noqa: E402

B, W, H = 4, 640, 360
seeds = [int(a) for a in sys.argv[1:]] or [201, 202, 203, 204, 205, 206, 207, 208]
court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
poi = synth.load_court_poi("pitch", B)
sys.path.insert(0, ROOT)
import bench  # noqa: E402   (usable_cores: the cgroup's CPU quota, not the host's core count)
torch.set_num_threads(min(bench.usable_cores(), 16))
OUT = open(os.path.join(ROOT, "gpurun_out", "r06_trained_like_seed_sweep.txt"), "w")


def emit(line):
    print(line, flush=True)
    OUT.write(line + "\n")
    OUT.flush()


emit(f"# {B} frames of {W}x{H} per seed; columns: mode, max|dtheta|, max|dlogits|, argmax pixels differing of {B * H * W} "
      "(largest top-2 margin among them), POI pixels differing, warp pixels differing vs oracle-warp(GPU theta)")
for seed in seeds:
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
    sd, info = synth.trained_like_state_dict(net.state_dict(), seed, return_info=True)
    net.load_state_dict(sd)
    net.range_rescales = net.range_raises = 0
    net.cuda().eval()
    x = torch.cat([synth.frames_to_float(synth.synth_frames_u8(2, H, W, seed=seed)), synth.smooth_frames(2, H, W, seed=seed)], 0)
    with torch.no_grad():
        want = torch_ref.predict(x, sd, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H), project_poi=True)
    top2 = want["logits"].topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    ref_arg = want["logits"].argmax(1)
    for prec in ("f16x3", "bf16x6", "fp32"):
        net.precision = prec
        with torch.no_grad():
            got = net.predict(x.cuda(), consistency=False, project_poi=True)
        th, lg = got["theta"].cpu(), got["logits"].cpu()
        diff = lg.argmax(1) != ref_arg
        nd = int(diff.sum())
        mm = float(margin[diff].max()) if nd else 0.0
        ppx = (torch.round(got["poi"].cpu() * W) != torch.round(want["poi"] * W)).sum().item()
        wm = (warp_ref.homography_warp(th, court, H, W, "nearest") * 4).to(torch.int32)
        wd = int((got["warp_mask"].cpu() != wm).sum())
        cond = float(torch.linalg.cond(want["theta"].reshape(-1, 3, 3).double()).max())
        emit(f"seed {seed} ({info['scaled_layer'].replace('.weight', '')} x 2^{info['scaled_layer_exp']:+d}, cond(theta) {cond:.0f}) {prec:7s} dtheta {float((th - want['theta']).abs().max()):.2e}  dlogits {float((lg - want['logits']).abs().max()):.2e}  "
              f"argmax differ {nd} (margin {mm:.1e})  poi px {ppx}  warp px {wd}  rescales / raises / fallbacks {net.range_rescales} / {net.range_raises} / {net.range_fallbacks}")
