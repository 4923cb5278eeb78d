#!/usr/bin/env python3
"""Nearest warp + consistency CE: the fused kernel (sfh_warp_consistency_fwd, one launch) against the separate kernels
(sfh_homography_warp_fwd + sfh_consistency_ce_fwd: warp, CE partial, CE final) at the C2 / C5 sizes.

Algorithmic bytes of the FUSED launch: logits B*4*h*w*4 (read once) + mask B*h*w*4 (written once) + one template + theta.
Timed with HIP events over `--iters` back-to-back launches (includes the launch gaps; the rocprofv3 kernel trace of the same
command gives the kernel-only durations).  Usage: python profiles/warpce_sweep.py [--iters 50]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    import torch
    from sfh_amd import engine, synth
    dev = torch.device("cuda", 0)
    for (name, W, H) in (("ncaa_nc4_640x360", 640, 360), ("pitch_v3_nc4_1280x720", 1280, 720)):
        tmpl = synth.load_court_template(name, 4, 1).to(dev)
        for B in (16, 128):
            g = torch.Generator().manual_seed(B)
            idx = torch.randint(0, len(synth.REALISTIC_THETAS), (B,), generator=g)
            theta = torch.tensor(synth.REALISTIC_THETAS, dtype=torch.float32)[idx]
            theta = (theta + 1e-3 * torch.randn(B, 3, 3, generator=g)).to(dev)
            logits = torch.randn(B, 4, H, W, device=dev) * 3

            def fused():
                return engine.warp_consistency(theta, tmpl, logits, 4.0, shared_template=True)

            def separate():
                _, wm = engine.homography_warp(theta, tmpl, H, W, True, scale=4.0, want_f32=False, want_i32=True, shared_template=True)
                return wm, engine.consistency_ce(logits, wm)
            res = {}
            for tag, fn in (("fused", fused), ("separate", separate)):
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[tag] = e0.elapsed_time(e1) * 1e3 / args.iters
            a, b = fused(), separate()
            assert torch.equal(a[0], b[0]) and float((a[1] - b[1]).abs().max()) < 1e-5
            nbytes = B * H * W * 4 * 5 + H * W * 4 + B * 36
            tbs = nbytes / (res["fused"] * 1e-6) / 1e12
            print(json.dumps({"size": f"{W}x{H}", "batch": B, "fused_us": round(res["fused"], 2), "separate_us": round(res["separate"], 2),
                              "algorithmic_bytes_fused": nbytes, "TB/s": round(tbs, 3), "frac_of_8TB/s": round(tbs / 8.0, 4)}), flush=True)


if __name__ == "__main__":
    main()
