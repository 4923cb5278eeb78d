"""Summarise the rocprofv3 passes collected by profiles/collect_pmc.sh into
profiles/<tag>_*.txt and profiles/pmc_traffic.json (read by bench.py's roofline.traffic).

HBM bytes per launch follow MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB,
collected in separate passes; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane)
reads, so the read side is doubled:  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.

usage: python profiles/summarize_pmc.py r01 [gpurun_out]
"""
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
here = os.path.dirname(os.path.abspath(__file__))


def rows(kind):
    """counter rows in dispatch order, each tagged with `_resnet` = launched after the step's
    space_to_depth2 (the ResNetSTN part of predict) and before the next step's nchw_to_nhwc"""
    f = sorted(glob.glob(os.path.join(src, f"{tag}_{kind}", "*", "*_counter_collection.csv")), key=os.path.getmtime)
    rr = list(csv.DictReader(open(f[-1]))) if f else []
    rr.sort(key=lambda r: int(r["Start_Timestamp"]))
    in_resnet = False
    for r in rr:
        if "nchw_to_nhwc" in r["Kernel_Name"] or "frame_to_h2" in r["Kernel_Name"]:   # first kernel of a step
            in_resnet = False
        elif "space_to_depth" in r["Kernel_Name"] or "stem7x7" in r["Kernel_Name"]:
            in_resnet = True
        r["_resnet"] = in_resnet
    return rr


def group(name, resnet=False):
    m = re.search(r"ConvCfg<(\d+), (\d+)", name)
    if "conv_mfma" in name and m:
        return {("3", "1"): "fp32_conv3x3", ("1", "1"): "fp32_conv1x1"}.get((m.group(1), m.group(2)), "fp32_other_conv")
    m = re.search(r"S3Cfg<(\d+), (\d+)", name)
    if "conv_s3" in name and m:
        if resnet:
            return "s3_resnet"
        return {"3": "s3_conv3x3", "2": "s3_up2x2"}.get(m.group(1), "s3_conv1x1")
    if "conv_small_kernel" in name:      # round 5: the small-map 3x3 kernel (ResNet layer3 / layer4 at batch 16)
        return "s3_resnet" if resnet else "s3_conv3x3"
    if "conv_upfused_kernel" in name:    # round 5: composed 2x2 + skip-half 3x3 of a fused Up block in one kernel (u3, u4)
        return "s3_upfused"
    if "stem7x7" in name:
        return "s3_stem7x7"
    if "conv3x3_c4" in name:
        return "first_layer_c4"
    for k in ("warp_kernel", "outconv", "maxpool", "avgpool", "space_to_depth", "nchw_to_nhwc", "pack_weights", "fold_bn", "ce_"):
        if k in name:
            return k
    return None


def per_group(kind, counter):
    acc = {}
    for r in rows(kind):
        if r["Counter_Name"] != counter:
            continue
        g = group(r["Kernel_Name"], r["_resnet"])
        if g is None:
            continue
        n, v, t = acc.get(g, (0, 0.0, 0.0))
        acc[g] = (n + 1, v + float(r["Counter_Value"]), t + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return acc


fetch = per_group("fetch", "FETCH_SIZE")
write = per_group("write", "WRITE_SIZE")
out = {}
lines = [f"# {tag}: HBM traffic per launch from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes)",
         "# hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
         f"{'kernel group':22s} {'launches':>8s} {'FETCH KiB/launch':>18s} {'WRITE KiB/launch':>18s} {'HBM MB/launch':>14s} {'GB/s (pmc-run time)':>20s}"]
for g in sorted(fetch):
    n, fv, ft = fetch[g]
    nw, wv, wt = write.get(g, (n, 0.0, ft))
    hbm = (2 * fv / n + wv / max(nw, 1)) * 1024
    out[g] = {"launches": n, "fetch_kib_per_launch": fv / n, "write_kib_per_launch": wv / max(nw, 1),
              "hbm_bytes_per_launch": hbm, "avg_launch_ns_in_pmc_run": ft / n}
    lines.append(f"{g:22s} {n:8d} {fv/n:18.1f} {wv/max(nw,1):18.1f} {hbm/1e6:14.2f} {hbm/(ft/n):20.1f}")
open(os.path.join(here, f"{tag}_pmc_hbm_traffic.txt"), "w").write("\n".join(lines) + "\n")
if "s3_conv3x3" in out:   # the DoubleConv launches of the default (bf16x6) mode
    out["doubleconv3x3"] = out["s3_conv3x3"]
elif "fp32_conv3x3" in out:
    out["doubleconv3x3"] = out["fp32_conv3x3"]
out["round"] = tag
json.dump(out, open(os.path.join(here, "pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines))

# MFMA busy
mf = {}
for r in rows("mfma"):
    g = group(r["Kernel_Name"], r["_resnet"])
    if g is None:
        continue
    d = mf.setdefault(g, {})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    d["_n"] = d.get("_n", 0) + (1 if r["Counter_Name"] == "SQ_BUSY_CYCLES" else 0)
    if r["Counter_Name"] == "SQ_BUSY_CYCLES":
        d["_ns"] = d.get("_ns", 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
ml = [f"# {tag}: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE (sums over launches)",
      "# mfma_busy_per_simd_cycle = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs): fraction of SIMD-cycles with the matrix pipe busy",
      f"{'kernel group':22s} {'launches':>8s} {'MFMA_BUSY':>16s} {'GRBM_GUI_ACTIVE':>16s} {'eff. clock GHz':>14s} {'mfma busy frac':>14s}"]
for g, d in sorted(mf.items()):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
        continue
    gui = d.get("GRBM_GUI_ACTIVE", 0.0)
    clk = gui / 8 / max(d.get("_ns", 1), 1)
    frac = d["SQ_VALU_MFMA_BUSY_CYCLES"] / max(gui / 8 * 1024, 1)
    ml.append(f"{g:22s} {d['_n']:8d} {d['SQ_VALU_MFMA_BUSY_CYCLES']:16.4g} {gui:16.4g} {clk:14.3f} {frac:14.3f}")
open(os.path.join(here, f"{tag}_pmc_mfma_busy.txt"), "w").write("\n".join(ml) + "\n")
print("\n".join(ml))

# kernel stats summary copies: the default command (pipelined region + the unpipelined pass behind it), and the
# pipelined region alone (--no-alone-pass: two batches in flight for every launch counted)
for sub, name, cmd in (("trace", "kernel_stats", "--steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs"),
                       ("trace_pipe", "kernel_stats_pipelined", "--steps 5 --warmup 1 --no-cpu-baseline --no-extra-configs --no-alone-pass")):
    ks = sorted(glob.glob(os.path.join(src, f"{tag}_{sub}", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if ks:
        rr = list(csv.DictReader(open(ks[-1])))
        sl = [f"# {tag}: rocprofv3 --kernel-trace --stats -- python3 bench.py {cmd}",
              f"{'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}  kernel"]
        for r in rr:
            sl.append(f"{r['Calls']:>6s} {int(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}  {r['Name'][:150]}")
        open(os.path.join(here, f"{tag}_{name}.txt"), "w").write("\n".join(sl) + "\n")
