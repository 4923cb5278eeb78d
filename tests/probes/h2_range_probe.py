import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
from sfh_amd import engine as E, modules, synth
import test_gpu_parity as T
g = np.load("/root/repo/tests/golden/blocks.npz")
rn, _ = T._mods_to_cuda(modules.ResNetSTN("resnet34", 7), 17)
x = torch.from_numpy(g["resnet34_7.x"])
y = E.nchw_to_nhwc(x.cuda(), 8)
res = {}
for prec in ("bf16x6", "f16x3"):
    eng = E.ResNetEngine(rn, 7, torch.device("cuda"), prec)
    th = eng.run(y, 2, 72, 128)
    torch.cuda.synchronize()
    res[prec] = {k: E.s3_to_f32(v[1], eng.ranges.exp("rn." + k)).float().cpu() if v[1].dtype != torch.float32 else v[1].cpu()
                 for k, v in eng.ws.bufs.items()}
    print(prec, th.flatten()[:4].tolist(), "headroom", getattr(eng.ranges, "headroom", dict)() if prec == "f16x3" and eng.ranges.read() is not None else "")
for k in res["bf16x6"]:
    a, b = res["bf16x6"][k], res["f16x3"][k]
    print(f"{k:24s} max|a| {a.abs().max().item():10.4g}  max|a-b| {(a-b).abs().max().item():10.4g}")
