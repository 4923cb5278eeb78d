"""World-size-2 gloo tests (CPU) of the frame-sharding / gather logic that the multi-GPU
bench path uses with RCCL."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sfh_amd import sharding


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 16, 128, 129):
        for w in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


class _FakeNet:
    """predict() stand-in: theta and score are pure functions of the frame content."""

    def predict(self, x, consistency=True):
        n = x.shape[0]
        key = x.reshape(n, -1)[:, 0]
        theta = (key.reshape(n, 1, 1, 1) + torch.arange(9.0).reshape(1, 1, 3, 3)).float()
        out = {"theta": theta}
        if consistency:
            out["consist_score"] = key * 0.5
        return out


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames = torch.arange(float(n_frames)).reshape(n_frames, 1, 1, 1).expand(n_frames, 3, 2, 2).contiguous()
        out = sharding.predict_sharded(_FakeNet(), frames, consistency=True)
        q.put((rank, out["shard"], out["theta_all"].tolist(), out["consist_score_all"].tolist()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [8, 5, 1])
def test_predict_sharded_gloo_world2(n_frames):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_frames
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_theta = (torch.arange(float(n_frames)).reshape(-1, 1, 1, 1) + torch.arange(9.0).reshape(1, 1, 3, 3))
    spans = sorted(r[1] for r in res)
    assert spans[0][0] == 0 and spans[-1][1] == n_frames and spans[0][1] == spans[1][0]
    for _, _, theta_all, score_all in res:
        assert torch.equal(torch.tensor(theta_all), want_theta)
        assert torch.equal(torch.tensor(score_all), torch.arange(float(n_frames)) * 0.5)


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes = [(4, 3, 3, 3), (4,), (9, 8), (1,)]
        flat, views = sharding.flat_views(shapes, "cpu")
        for i, v in enumerate(views):
            v.copy_(torch.full(shapes[i], float((rank + 1) * (i + 1))))
        scale = sharding.allreduce_gradients(flat)
        q.put((rank, scale, [float(v.flatten()[0]) for v in views], flat.numel(),
               all(v.data_ptr() >= flat.data_ptr() for v in views)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 77
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, scale, firsts, n, inside in res:
        assert scale == 0.5 and n == 4 * 27 + 4 + 72 + 1 and inside
        # rank r filled tensor i with (r+1)*(i+1): the sum over both ranks is 3*(i+1), the mean 1.5*(i+1)
        assert firsts == [3.0 * (i + 1) for i in range(4)]


def test_allreduce_without_process_group_is_identity():
    flat, views = sharding.flat_views([(2, 2), (3,)], "cpu")
    views[0].fill_(2.0)
    assert sharding.allreduce_gradients(flat) == 1.0 and float(flat.sum()) == 8.0


def test_bench_self_launch_builds_one_child_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts torch.distributed.run as a CHILD (this process never
    touches the GPU) on 127.0.0.1 with N ranks and hands its own arguments through."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location(
        "bench_under_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                  # the child's exit code is this process's
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch.cuda" not in "".join(m for m in sys.modules if m.startswith("bench_under_test"))


def _gather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = sharding.ResultGather(world, 3, "cpu", depth=2)
        got = []
        slots = []
        for step in range(5):     # more steps than slots: the ring wraps
            theta = (torch.arange(27.0).reshape(3, 1, 3, 3) + 100 * rank + 1000 * step)
            slots.append(g.submit(theta, torch.full((3,), float(10 * rank + step))))
            th, sc = g.result(slots[-1])
            got.append((th[:, 0, 0, 0].tolist(), sc.tolist()))
        q.put((rank, got))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_result_gather_ring_gloo_world2():
    """The side-stream gather of bench.py's sharded step (synchronous on CPU tensors): every rank sees every rank's
    rows of the step it asked for, in rank order, also after the slot ring has wrapped."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 77
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res[0] == res[1]
    for step, (th00, sc) in enumerate(res[0]):
        assert th00 == [1000.0 * step + 9 * i for i in range(3)] + [1000.0 * step + 100 + 9 * i for i in range(3)]
        assert sc == [float(step)] * 3 + [10.0 + step] * 3


def test_result_gather_single_process():
    g = sharding.ResultGather(1, 2, "cpu")
    th, sc = g.result(g.submit(torch.ones(2, 1, 3, 3), torch.tensor([3.0, 4.0])))
    assert th.shape == (2, 1, 3, 3) and sc.tolist() == [3.0, 4.0]


class _StandInReconstructor(torch.nn.Module):
    """Takes Reconstructor's place in bench.main() on the CPU: theta is a pure function of the frames (their mean per
    frame), so that the gathered rows can be checked against what each rank computed."""
    precision = "f16x3"
    range_fallbacks = range_rescales = range_raises = 0

    def __init__(self, court_img, court_poi, **kw):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))

    def predict(self, x, consistency=True, project_poi=False):
        n = x.shape[0]
        theta = x.reshape(n, -1).mean(1).reshape(n, 1, 1, 1) + torch.arange(9.0).reshape(1, 1, 3, 3)
        out = {"theta": theta.float(), "logits": torch.zeros(n, 4, 2, 2)}
        if consistency:
            out["consist_score"] = theta.reshape(n, 9)[:, 0] * 0.5
        return out

    def predict_async(self, x, consistency=True, project_poi=False):
        out = self.predict(x, consistency, project_poi)

        class _H:
            def result(self_inner):
                return out
        return _H()

    def invalidate_engines(self):
        pass


def _bench_main_worker(rank, world, port, q):
    import contextlib
    import importlib.util
    import io
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import sfh_amd.reconstructor as R
    from sfh_amd import synth
    R.Reconstructor = _StandInReconstructor
    # small frames: the plumbing is what runs here, not the workload
    real = synth.synth_frames_u8
    synth.synth_frames_u8 = lambda B, H, W, seed=0: real(B, 36, 64, seed=seed)
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1", "--device", "cpu", "--dist-backend", "gloo",
                "--no-cpu-baseline", "--no-extra-configs"] + (["--force-collective"] if world == 1 else [])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("{")]
    import torch.distributed as dist
    q.put((rank, json.loads(lines[-1]) if lines else None, dist.is_initialized()))


def test_bench_main_two_ranks_gloo_reports_the_whole_job():
    """bench.py's own main() under WORLD_SIZE=2 (gloo, CPU, a stand-in model): ONE JSON line, from rank 0, with
    n_gpus 2, a global batch of 32 frames, weak scaling, value = frames of BOTH ranks per second, and the exchange
    step verified inside the run - every rank holds 2 x 16 gathered rows and its own rows equal its own theta."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 131
    procs = [ctx.Process(target=_bench_main_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert not any(g[2] for g in got)              # main() tore the process group down
    res = {g[0]: g[1] for g in got}
    assert res[1] is None                          # only rank 0 prints
    line = res[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 32 and line["config"]["frames_per_gpu_per_step"] == 16
    assert "frame-sharded x2" in line["config"]["parallelism"] and "gloo" in line["config"]["parallelism"]
    assert line["metric"].startswith("frames/sec at 640x360 batch=16") and line["unit"] == "frames/s"
    # whole-job throughput: 2 ranks x 16 frames x 3 steps over the (max over ranks) elapsed time
    assert abs(line["value"] - 2 * 16 * 3 / (line["ms_per_step"] * 3e-3)) < 0.02 * line["value"]
    assert line["gather_check"] == {"rows_per_rank": 32, "own_rows_equal_own_theta_on_every_rank": True,
                                    "bytes_per_step_per_rank": 640, "backend": "gloo",
                                    "collective": "all_gather_into_tensor on a side stream, one per step",
                                    "collectives_run": 4, "forced_on_one_rank": False}
    assert line["cpu_baseline"] is None and line["other_configs"] is None and line["vs_baseline"] is None
    # round 5: BASELINE config 4 gathers theta + consistency - with N > 1 the score is computed without --consistency
    assert "consistency" in line["config"]["workload"] and "all_gather_into_tensor" in line["config"]["parallelism"]
    # per-rank figures next to the whole-job value: own step time of every rank (the value uses the slowest)
    pr = line["per_rank"]
    assert [r["rank"] for r in pr["ranks"]] == [0, 1]
    assert pr["ms_per_step_min"] <= pr["ms_per_step_median"] <= pr["ms_per_step_max"] <= line["ms_per_step"] * 1.001
    assert all(set(r) >= {"ms_per_step", "device", "mfma_f16_tflops", "in_kernel_clock_ghz", "power_w_timed_region"} for r in pr["ranks"])
    assert {"frac", "mfma_utilisation", "traffic_algorithmic", "traffic_ratio"} <= set(line["roofline"])
    assert "device_calibration" in line and "parity" in line
    assert {"value_predict_sync", "value_exact_operands", "value_per_calibrated_pflop"} <= set(line)
    assert line["config"]["range_raises"] == 0



def test_bench_main_one_rank_force_collective_gloo():
    """`bench.py --gpus 1 --force-collective`: a one-rank process group of its own (no launcher), the exchange step
    through the real collective, a non-null gather_check, and the group destroyed when main() returns."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    p = ctx.Process(target=_bench_main_one_rank_worker, args=(q,))
    p.start()
    rank, line, still_up = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0 and not still_up
    assert line["n_gpus"] == 1 and line["config"]["global_batch"] == 16
    gc = line["gather_check"]
    assert gc is not None and gc["forced_on_one_rank"] and gc["own_rows_equal_own_theta_on_every_rank"]
    assert gc["rows_per_rank"] == 16 and gc["collectives_run"] == 4 and gc["backend"] == "gloo"
    assert "--force-collective" in line["config"]["parallelism"] and "consistency" in line["config"]["workload"]


def _bench_main_one_rank_worker(q):
    import contextlib
    import importlib.util
    import io
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    import sfh_amd.reconstructor as R
    from sfh_amd import synth
    R.Reconstructor = _StandInReconstructor
    real = synth.synth_frames_u8
    synth.synth_frames_u8 = lambda B, H, W, seed=0: real(B, 36, 64, seed=seed)
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sys.argv = ["bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--device", "cpu", "--dist-backend", "gloo",
                "--no-cpu-baseline", "--no-extra-configs", "--force-collective"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("{")]
    import torch.distributed as dist
    q.put((0, json.loads(lines[-1]) if lines else None, dist.is_initialized()))


def _bench_main_raises_worker(q):
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    import sfh_amd.reconstructor as R

    class _Boom(_StandInReconstructor):
        def predict_async(self, *a, **k):
            raise RuntimeError("boom inside the timed region")
    R.Reconstructor = _Boom
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sys.argv = ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--device", "cpu", "--dist-backend", "gloo",
                "--no-cpu-baseline", "--no-extra-configs", "--force-collective"]
    import torch.distributed as dist
    try:
        bench.main()
        q.put(("no error", dist.is_initialized()))
    except RuntimeError as e:
        q.put((str(e), dist.is_initialized()))


def test_bench_main_destroys_the_process_group_when_a_step_raises():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_bench_main_raises_worker, args=(q,))
    p.start()
    msg, still_up = q.get(timeout=240)
    p.join(timeout=60)
    assert "boom" in msg and not still_up


def _force_worker(q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + (os.getpid() % 2000) + 211), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        theta = torch.arange(36.0).reshape(4, 1, 3, 3)
        score = torch.arange(4.0)
        a = sharding.gather_results(theta, score)                              # one rank: the local shortcut
        b = sharding.gather_results(theta, score, force_collective=True)       # the collective anyway
        g0 = sharding.ResultGather(1, 4, "cpu")
        g1 = sharding.ResultGather(1, 4, "cpu", force_collective=True)
        r0 = g0.result(g0.submit(theta, score))
        r1 = g1.result(g1.submit(theta, score))
        flat, _ = sharding.flat_views([(3, 3), (5,)], "cpu")
        flat.fill_(2.0)
        s0 = sharding.allreduce_gradients(flat)
        s1 = sharding.allreduce_gradients(flat, force_collective=True)
        q.put((a[0].data_ptr() == theta.data_ptr(), b[0].data_ptr() != theta.data_ptr(),
               torch.equal(b[0], theta) and torch.equal(b[1], score),
               g0.collectives_run, g1.collectives_run, torch.equal(r0[0], r1[0]) and torch.equal(r1[0], theta)
               and torch.equal(r1[1], score), s0, s1, float(flat.sum())))
    finally:
        dist.destroy_process_group()


def test_force_collective_takes_a_one_rank_world_through_the_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_force_worker, args=(q,))
    p.start()
    res = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert res == (True, True, True, 0, 1, True, 1.0, 1.0, 28.0)


def test_more_than_one_rank_without_a_process_group_raises():
    """world > 1 with torch.distributed uninitialised used to copy (n, 10) rows into a (world * n, 10)
    buffer (a broadcast shape error); it is a clear error now"""
    assert not dist.is_initialized()
    with pytest.raises(RuntimeError, match="not initialised"):
        sharding.ResultGather(2, 4, "cpu")
    g = sharding.ResultGather(1, 4, "cpu")                 # one rank needs no group
    th, sc = g.result(g.submit(torch.ones(4, 1, 3, 3), torch.ones(4)))
    assert tuple(th.shape) == (4, 1, 3, 3) and float(sc.sum()) == 4.0


def test_bench_main_eight_ranks_gloo_is_config_4_s_shape():
    """BASELINE config 4's shape - 8 ranks x 16 frames = 128 frames per step, theta + consistency gathered - through bench.py's
    own main() on the CPU (gloo, a stand-in model): the whole-job line, every rank's rows in the gather, all 8 process groups
    torn down.  (What the driver's N = 8 run exercises around the kernels; the kernels themselves need the GPUs.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 177
    procs = [ctx.Process(target=_bench_main_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    got = [q.get(timeout=400) for _ in range(8)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert not any(g[2] for g in got)
    lines = [g[1] for g in got if g[1] is not None]
    assert len(lines) == 1                                   # only rank 0 prints
    line = lines[0]
    assert line["n_gpus"] == 8 and line["config"]["global_batch"] == 128 and line["scaling"] == "weak"
    gc = line["gather_check"]
    assert gc["rows_per_rank"] == 128 and gc["own_rows_equal_own_theta_on_every_rank"] and not gc["forced_on_one_rank"]
    assert [r["rank"] for r in line["per_rank"]["ranks"]] == list(range(8))
    assert abs(line["value"] - 8 * 16 * 3 / (line["ms_per_step"] * 3e-3)) < 0.02 * line["value"]
