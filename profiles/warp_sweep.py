#!/usr/bin/env python3
"""Homography warp (SURVEY.md §8 row A7+A7p) against the HBM roofline: batch sweep at C2/C5 sizes.

Algorithmic bytes per launch (row D): B*h*w*4 (int32 mask write) + ht*wt*4 (the one shared
template, read once) + B*36 (theta).  Timed with HIP events over `--iters` back-to-back launches
on the launch stream.  Usage: python profiles/warp_sweep.py [--iters 50]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    import torch
    from sfh_amd import engine, synth

    dev = torch.device("cuda", 0)
    rows = []
    for (name, W, H) in (("ncaa_nc4_640x360", 640, 360), ("pitch_v3_nc4_1280x720", 1280, 720)):
        tmpl = synth.load_court_template(name, 4, 1).to(dev)
        for B in (16, 128, 1024):
            if B * H * W * 4 > 8 << 30:
                continue
            g = torch.Generator().manual_seed(B)
            idx = torch.randint(0, len(synth.REALISTIC_THETAS), (B,), generator=g)
            theta = torch.tensor(synth.REALISTIC_THETAS, dtype=torch.float32)[idx]
            theta = (theta + 1e-3 * torch.randn(B, 3, 3, generator=g)).to(dev)
            for mode, nearest in (("nearest->i32", True), ("bilinear->f32", False)):
                kw = dict(nearest=nearest, scale=4.0 if nearest else None, want_f32=not nearest,
                          want_i32=nearest, shared_template=True)
                for _ in range(3):
                    engine.homography_warp(theta, tmpl, H, W, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    engine.homography_warp(theta, tmpl, H, W, **kw)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.iters
                nbytes = B * H * W * 4 + H * W * 4 + B * 36
                gbs = nbytes / (us * 1e-6) / 1e9
                rows.append({"size": f"{W}x{H}", "batch": B, "mode": mode, "us_per_launch": round(us, 2),
                             "algorithmic_bytes": nbytes, "GB/s": round(gbs, 1),
                             "frac_of_8TB/s": round(gbs / HBM_PEAK_GBS, 4)})
                print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
