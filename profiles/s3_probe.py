"""Probe of the split-bf16 (S3) conv kernel against the fp32 MFMA kernel: accuracy vs an fp64
CPU reference on a small case and throughput on the big DoubleConv shapes.  GPU box only."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sfh_amd import _lib, engine as E  # noqa: E402
from sfh_amd._lib import ConvDesc  # noqa: E402

lib = _lib.load()
st = E._stream


def to_s3(x):  # (B,H,W,C) f32 -> (B,H,W,3,C) bf16
    return E.f32_to_s3(x)


class S3Conv:
    def __init__(self, w, bias, bn, ks, c0, c1=0):
        cout = w.shape[0]
        n = lib.sfh_packed_s3_weight_bytes(ks, c0, c1, cout)
        assert n > 0
        self.wp = torch.empty(n, dtype=torch.uint8, device=w.device)
        _lib.check(lib.sfh_pack_s3_weights(E._ptr(w), E._ptr(self.wp), ks, c0, c1, cout, 0, 0, st()), "pack_s3")
        self.scale = torch.empty(cout, device=w.device)
        self.shift = torch.empty(cout, device=w.device)
        _lib.check(lib.sfh_fold_bn(E._ptr(bias), E._ptr(bn.weight), E._ptr(bn.bias), E._ptr(bn.running_mean),
                                   E._ptr(bn.running_var), 1e-5, cout, 1, E._ptr(self.scale), E._ptr(self.shift), st()), "fold")
        self.ks, self.c0, self.c1, self.cout = ks, c0, c1, cout

    def run(self, xs3, B, H, W, dst, dst_fmt=0, pool=None, tile=None):
        d = ConvDesc()
        d.src0 = xs3.data_ptr(); d.c0 = self.c0; d.cs0 = E._chan(xs3); d.h0 = H; d.w0 = W
        d.batch, d.H, d.W, d.ksize, d.stride = B, H, W, self.ks, 1
        d.tile = E.choose_tile_s3(B, H, W, 1, 2, self.cout // 64) if tile is None else tile
        d.wpacked, d.scale, d.shift = self.wp.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr()
        d.cout, d.relu = self.cout, 1
        d.dst = dst.data_ptr(); d.dst_cs = E._chan(dst); d.src_fmt = 1; d.dst_fmt = dst_fmt
        if pool is not None:
            d.dst_pool = pool.data_ptr(); d.pool_cs = E._chan(pool)
        _lib.check(lib.sfh_conv_s3_fwd(ctypes.byref(d), st()), "conv_s3")


def bench(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def layer(cin, cout, h, w, B):
    torch.manual_seed(0)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device="cuda") * 0.1
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    with torch.no_grad():
        bn.running_mean.uniform_(-0.1, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.1, 0.1)
    x = torch.relu(torch.randn(B, h, w, cin, device="cuda")) * 1.3
    return wt, bias, bn, x


def accuracy():
    B, h, w, cin, cout = 2, 45, 80, 512, 64
    wt, bias, bn, x = layer(cin, cout, h, w, B)
    s3 = S3Conv(wt, bias, bn, 3, cin)
    y3 = torch.empty(B, h, w, cout, device="cuda")
    s3.run(to_s3(x), B, h, w, y3)
    f32 = E.PackedConv(wt, bias, bn, 3, cin)
    yf = torch.empty(B, h, w, cout, device="cuda")
    f32.run(x, B, h, w, yf)
    torch.cuda.synchronize()
    xd = x.permute(0, 3, 1, 2).double().cpu()
    ref = torch.nn.functional.conv2d(xd, wt.double().cpu(), bias.double().cpu(), padding=1)
    a = (bn.weight / torch.sqrt(bn.running_var + 1e-5)).double().cpu().view(1, -1, 1, 1)
    ref = torch.relu((ref - bn.running_mean.double().cpu().view(1, -1, 1, 1)) * a + bn.bias.double().cpu().view(1, -1, 1, 1))
    ref = ref.permute(0, 2, 3, 1)
    for name, y in (("s3 (bf16x6)", y3), ("fp32 mfma", yf)):
        e = (y.double().cpu() - ref).abs()
        print(f"accuracy K={9*cin}: {name:12s} max abs err {e.max().item():.3e} mean {e.mean().item():.3e} (|ref| mean {ref.abs().mean().item():.3f})")
    # S3 output + fused pool
    ys3 = E.s3_empty(B, h, w, cout, "cuda")
    yp = E.s3_empty(B, h // 2, w // 2, cout, "cuda")
    s3.run(to_s3(x), B, h, w, ys3, dst_fmt=1, pool=yp)
    torch.cuda.synchronize()
    rec = E.s3_to_f32(ys3)
    print("s3 output planes reconstruct fp32 output exactly:", torch.equal(rec, y3))
    pooled = torch.nn.functional.max_pool2d(y3.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    print("fused pool == maxpool(fp32 out):", torch.equal(E.s3_to_f32(yp), pooled))


def speed():
    for name, cin, cout, h, w in (("64->64 360x640", 64, 64, 360, 640), ("128->128 180x320", 128, 128, 180, 320),
                                  ("256->256 90x160", 256, 256, 90, 160), ("512->512 45x80", 512, 512, 45, 80),
                                  ("1024->1024 22x40", 1024, 1024, 22, 40)):
        B = 16
        wt, bias, bn, x = layer(cin, cout, h, w, B)
        s3 = S3Conv(wt, bias, bn, 3, cin)
        xs = to_s3(x)
        y = E.s3_empty(B, h, w, cout, "cuda")
        f32 = E.PackedConv(wt, bias, bn, 3, cin)
        yf = torch.empty(B, h, w, cout, device="cuda")
        fl = 2.0 * B * h * w * cout * 9 * cin
        t3 = bench(lambda: s3.run(xs, B, h, w, y, dst_fmt=1))
        t3f = bench(lambda: s3.run(xs, B, h, w, yf, dst_fmt=0))
        print(f"   (fp32 destination instead of S3: {t3f:7.3f} ms {fl/t3f/1e9:7.1f} TF-equiv)")
        tf = bench(lambda: f32.run(x, B, h, w, yf))
        print(f"{name:18s} s3 {t3:7.3f} ms {fl/t3/1e9:7.1f} TF-equiv | fp32 {tf:7.3f} ms {fl/tf/1e9:7.1f} TF | speedup {tf/t3:4.2f}x", flush=True)


if __name__ == "__main__":
    accuracy()
    speed()
