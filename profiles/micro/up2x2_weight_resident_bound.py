"""the composed 2x2 launches of up1 / up2 (`fusedup2x2`) fetch 6x their algorithmic bytes - every pixel tile
streams ALL cout blocks' weights (33.5 MB at up1) through its XCD's 4 MB L2 (profiles/r04_pmc_per_launch.txt: 1023 MiB raw FETCH
for 58 MB of input + 34 MB of weights).  A weight-stationary order (a few cout blocks resident per XCD, all pixel tiles streamed
past them) would cut that stream.  UPPER BOUND of what such an order can return, without building it: the same kernel on a
launch with the SAME MFMA work whose weights already fit the L2s - 1/8 of the couts (4 blocks = 4.2 MB at up1) over 8x the
frames - against the real launch.  If the bound is not faster by the kill line (0.1 ms per batch), the order is not built.
GPU box only.  usage: python profiles/micro/up2x2_weight_resident_bound.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sfh_amd import engine as E  # noqa: E402


def bench(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def case(tag, cx, cout, B, hy, wy):
    """composed 2x2 conv over a (B, hy, wy, cx) H2 tensor -> fp32 partial (B, 2hy, 2wy, cout), as UNetEngine runs it"""
    torch.manual_seed(0)
    c0 = cout                                            # skip channels of the 3x3 conv (not read by this launch)
    conv = torch.nn.Conv2d(c0 + cout, cout, 3, padding=1).cuda()
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    up = torch.nn.ConvTranspose2d(cx, cout, 2, 2).cuda()
    fu = E.PackedConv.fused_up(conv, bn, up, c0, fmt="h2")
    fu.relu = False
    y = E.f32_to_h2(torch.relu(torch.randn(B, hy, wy, cx, device="cuda")))
    part = torch.empty((B, 2 * hy, 2 * wy, cout), dtype=torch.float32, device="cuda")
    ms = bench(lambda: fu.run(y, B, hy, wy, part))
    # socket power while the launch repeats for about half a second (the hwmon reading is a moving average)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_power", os.path.join(ROOT, "bench.py"))
    bench_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench_mod)
    props = torch.cuda.get_device_properties(0)
    bus = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    with bench_mod.PowerSampler(bus, period=0.01) as ps:
        for _ in range(int(500.0 / ms)):
            fu.run(y, B, hy, wy, part)
        torch.cuda.synchronize()
    pw = ps.summary() or {}
    executed = 2.0 * B * hy * wy * (4 * cout) * 4 * cx
    wmb = 4 * cout * cx * 4 * 4 / 2 ** 20
    print(f"{tag:34s} cx {cx:5d} -> 4 x {cout:4d} couts, {B:4d} frames of {hy}x{wy}: {ms * 1e3:8.1f} us  "
          f"{executed / ms / 1e9:7.1f} TFLOP/s executed   weights {wmb:6.1f} MB, input {B * hy * wy * cx * 4 / 2 ** 20:7.1f} MB   "
          f"socket power mean {pw.get('mean_w')} W, max {pw.get('max_w')} W over 0.5 s of this launch", flush=True)
    return ms


if __name__ == "__main__":
    for rep in range(2):
        a = case("up1 as the model runs it", 1024, 512, 16, 22, 40)
        b = case("up1 bound: weights L2-resident", 1024, 64, 128, 22, 40)
        c = case("up2 as the model runs it", 512, 256, 16, 45, 80)
        d = case("up2 bound: weights L2-resident", 512, 64, 64, 45, 80)
        print(f"  round {rep}: upper bound of a weight-stationary order: up1 {1e3 * (a - b):+.1f} us, up2 {1e3 * (c - d):+.1f} us per batch")
