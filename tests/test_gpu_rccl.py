"""The N > 1 code on REAL RCCL with the one GPU a test box has (BASELINE config 4).

Every multi-rank test of this repo is gloo on CPU tensors (tests/test_sharding.py); the first 8-GPU run must not be the
first time librccl is loaded beside this library's streams.  Here a process group of ONE rank on backend "nccl" (= RCCL
on ROCm) is initialised in the test process and `force_collective=True` takes every exchange entry point past its
one-rank shortcut: the same `all_gather_into_tensor` on the side stream, the same events, the same `all_reduce` on the
flat 209 MB gradient buffer as at N = 8 - only the peer count differs.  Matches /root/reference/predict.py:57-77 (one
process per device; results cross to the collecting process)."""
import os
import socket

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture(scope="module")
def rccl_world1():
    import torch.distributed as dist
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    assert not dist.is_initialized(), "another test left a process group behind"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        yield dev
    finally:
        dist.destroy_process_group()
    assert not dist.is_initialized()


def _small_net(dev, B, w=112, h=90, seed=23, nearest=True):
    from sfh_amd import synth
    from sfh_amd.reconstructor import Reconstructor
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :h, :w].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.to(dev), poi.to(dev), target_size=(w, h), unet_size=(w, h), warp_size=(w, h),
                        warp_with_nearest=nearest)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed))
    return net.to(dev).eval()


def test_result_gather_runs_the_real_collective_for_three_pipelined_batches(rccl_world1):
    """ResultGather past the world == 1 shortcut: three predict_async batches, each gathered on the side stream by RCCL
    while the next batch's kernels are already enqueued; the gathered rows are the rank's own theta / score."""
    from sfh_amd import sharding, synth
    dev = rccl_world1
    B, w, h = 4, 112, 90
    net = _small_net(dev, B)
    frames = [synth.smooth_frames(B, h, w, seed=40 + k).to(dev) for k in range(3)]
    g = sharding.ResultGather(1, B, dev, depth=2, force_collective=True)
    assert g.collective
    pend, got, want = [], [], []
    with torch.no_grad():
        for k in range(3):
            pend.append(net.predict_async(frames[k], consistency=True))
            if len(pend) > 1:
                out = pend.pop(0).result()
                want.append((out["theta"].clone(), out["consist_score"].clone()))
                got.append(g.submit(out["theta"], out["consist_score"]))
                if len(got) == 2:      # the ring has two slots: read the older one before the third submit reuses it
                    got[0] = g.result(got[0])
        out = pend.pop(0).result()
        want.append((out["theta"].clone(), out["consist_score"].clone()))
        got.append(g.submit(out["theta"], out["consist_score"]))
        got[1:] = [g.result(s) for s in got[1:]]
    torch.cuda.synchronize()
    assert g.collectives_run == 3
    for (th, sc), (wth, wsc) in zip(got, want):
        assert tuple(th.shape) == (B, 1, 3, 3) and torch.equal(th, wth) and torch.equal(sc, wsc)
    # three different batches gave three different thetas (the comparison above is not vacuous)
    assert not torch.equal(want[0][0], want[1][0]) and not torch.equal(want[1][0], want[2][0])
    # and the pipelined results are what the drop-in predict() gives for the same frames
    with torch.no_grad():
        ref = net.predict(frames[2], consistency=True)
    assert torch.equal(ref["theta"], want[2][0]) and torch.equal(ref["consist_score"], want[2][1])


def test_gather_results_and_predict_sharded_on_rccl(rccl_world1):
    from sfh_amd import sharding, synth
    dev = rccl_world1
    B = 3
    net = _small_net(dev, B)
    x = synth.smooth_frames(B, 90, 112, seed=77).to(dev)
    with torch.no_grad():
        out = sharding.predict_sharded(net, x, consistency=True, force_collective=True)
    assert out["shard"] == (0, B)
    assert torch.equal(out["theta_all"], out["theta"]) and torch.equal(out["consist_score_all"], out["consist_score"])
    assert out["theta_all"].data_ptr() != out["theta"].data_ptr()      # it came out of the receive buffer, not the shortcut
    th, sc = sharding.gather_results(out["theta"], None, force_collective=True)
    assert torch.equal(th, out["theta"]) and float(sc.abs().max()) == 0.0
    # without the switch a one-rank world takes the local path (same values)
    th2, sc2 = sharding.gather_results(out["theta"], out["consist_score"])
    assert th2.data_ptr() == out["theta"].data_ptr() and torch.equal(sc2, out["consist_score"])


def test_gradient_allreduce_of_the_flat_209mb_buffer_on_rccl(rccl_world1):
    """the data-parallel training exchange: ONE all-reduce over every gradient of the default model (52.3 M fp32)"""
    from sfh_amd import sharding
    from sfh_amd.reconstructor import Reconstructor
    dev = rccl_world1
    court = torch.zeros((1, 1, 36, 64))
    net = Reconstructor(court, torch.zeros((1, 4, 2)), target_size=(64, 36), unet_size=(64, 36), warp_size=(64, 36))
    shapes = [tuple(p.shape) for p in net.parameters()]
    flat, views = sharding.flat_views(shapes, dev)
    assert flat.numel() * 4 > 200e6
    gen = torch.Generator(device=dev).manual_seed(5)
    flat.copy_(torch.randn(flat.numel(), device=dev, generator=gen))
    before = flat.clone()
    assert sharding.allreduce_gradients(flat) == 1.0                      # shortcut: untouched
    assert torch.equal(flat, before)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                         # beside work on the default stream
        scale = sharding.allreduce_gradients(flat, force_collective=True)
    busy = before * 2.0                                                   # (default stream keeps working meanwhile)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert scale == 1.0
    assert torch.equal(flat, before)                                      # the sum over one rank, through RCCL
    assert torch.equal(busy, before * 2.0)
    assert all(v.data_ptr() >= flat.data_ptr() and v.data_ptr() < flat.data_ptr() + flat.numel() * 4 for v in views)


def test_train_step_with_forced_allreduce_gives_the_same_update(rccl_world1):
    """TrainStep.step with the all-reduce forced through RCCL (world 1: sum = own gradient, scale 1.0): losses and the
    flat gradient buffer the optimizer kernel reads are those of a step without the collective (up to the run-to-run
    rounding of the step's fp64 atomics - two plain steps differ by as much), and the update happened."""
    from sfh_amd import synth
    from sfh_amd.training import TrainStep
    dev = rccl_world1
    B, w, h = 2, 64, 48
    results = []
    for force in (False, True):
        net = _small_net(dev, B, w, h, seed=31, nearest=False)
        net.train()
        ts = TrainStep(net)
        ts.force_collective = force
        g = torch.Generator().manual_seed(9)
        x = synth.smooth_frames(B, h, w, seed=3).to(dev)
        npoi = net.court_poi.shape[1]
        batch = {"mask": torch.randint(0, 4, (B, h, w), generator=g).to(dev),
                 "weight": torch.ones(B, device=dev),
                 "poi": torch.rand((B, npoi, 2), generator=g).to(dev),
                 "nonzeros": torch.ones((B, npoi), device=dev),
                 "num_nonzero": torch.full((B,), float(npoi), device=dev)}
        p_before = [p.detach().clone() for p in net.parameters()]
        losses = ts.step(x, batch)
        torch.cuda.synchronize()
        moved = sum(int((a != b.detach()).sum()) for a, b in zip(p_before, net.parameters()))
        results.append((losses.clone(), ts.gflat.clone(), moved))
    (l0, g0, m0), (l1, g1, m1) = results
    assert torch.isfinite(l0).all() and torch.isfinite(g1).all()
    assert float((l0 - l1).abs().max()) <= 1e-9 * float(l0.abs().max())
    assert float((g0 - g1).abs().max()) <= 1e-5 * float(g0.abs().max())
    assert m0 > 0.9 * g0.numel() and m1 > 0.9 * g1.numel()        # RMSprop moved (nearly) every weight in both runs
