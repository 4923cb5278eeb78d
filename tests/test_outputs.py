"""Output formats: PNG-in-pickle mask stream, *_court.json, mask formatting."""
import io
import json
import os

import numpy as np
import pytest
import torch

from sfh_amd import outputs
from sfh_amd import outputs as O
from sfh_amd import synth
from oracle import post_ref, torch_ref


def _pil():
    return pytest.importorskip("PIL.Image")


@pytest.mark.parametrize("shape", [(7, 13), (36, 64), (5, 9, 3), (1, 1)])
def test_png_roundtrip_and_pil(shape):
    Image = _pil()
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 4 if len(shape) == 2 else 256, size=shape, dtype=np.uint8)
    buf = outputs.encode_png(img)
    assert buf.dtype == np.uint8 and buf.ndim == 1
    assert np.array_equal(outputs.decode_png(buf), img)
    pil = np.array(Image.open(io.BytesIO(buf.tobytes())))
    want = img if img.ndim == 2 else img[:, :, ::-1]  # file holds RGB, memory holds BGR (cv2 convention)
    assert np.array_equal(pil, want)


@pytest.mark.parametrize("mode", ["L", "RGB", "RGBA"])
def test_png_decode_all_filters(mode):
    """PIL's encoder uses the adaptive scanline filters (sub/up/average/paeth)."""
    Image = _pil()
    rng = np.random.default_rng(3)
    nch = {"L": 1, "RGB": 3, "RGBA": 4}[mode]
    a = (np.add.outer(np.arange(24) * 5, np.arange(31) * 3)[..., None] + rng.integers(0, 9, (24, 31, nch))) % 256
    a = a.astype(np.uint8)
    bio = io.BytesIO()
    Image.fromarray(a[:, :, 0] if nch == 1 else a, mode).save(bio, format="PNG")
    got = outputs.decode_png(np.frombuffer(bio.getvalue(), np.uint8))
    want = a[:, :, 0] if nch == 1 else (a[:, :, ::-1] if nch == 3 else a[:, :, [2, 1, 0, 3]])
    assert np.array_equal(got, want)


def test_png_rejects_bad_input():
    with pytest.raises(ValueError):
        outputs.encode_png(np.zeros((4, 4), np.float32))
    with pytest.raises(ValueError):
        outputs.encode_png(np.full((2, 2), 300, np.int32))
    with pytest.raises(ValueError):
        outputs.decode_png(np.zeros(16, np.uint8))
    buf = outputs.encode_png(np.zeros((4, 4), np.uint8)).copy()
    buf[-20] ^= 1
    with pytest.raises(ValueError):
        outputs.decode_png(buf)


def test_mask_pickle_stream(tmp_path):
    rng = np.random.default_rng(0)
    masks = {f"frame_{i:04d}": rng.integers(0, 4, (18, 32), dtype=np.uint8) for i in range(5)}
    with outputs.MaskPickleWriter(str(tmp_path), postfix="court/warp_mask") as w:
        for n, m in masks.items():
            w.write(n, m)
    path = os.path.join(str(tmp_path), "court/warp_mask", "data.pkl")
    rd = outputs.MaskReader(path)
    assert [n for n, _ in rd.get()] == list(masks)
    for n, m in rd.get(decode=True):
        assert np.array_equal(m, masks[n])
    for _, buf in rd.get():  # records hold the raw PNG buffer, like cv2.imencode's second result
        assert isinstance(buf, np.ndarray) and buf.dtype == np.uint8


def test_court_json_schema(tmp_path):
    th = np.arange(18, dtype=np.float32).reshape(2, 1, 3, 3) + np.eye(3, dtype=np.float32)
    poi = np.linspace(0, 1, 2 * 33 * 2, dtype=np.float32).reshape(2, 33, 2)
    preds = {"theta": th, "poi": poi, "consist_score": np.array([0.123456789, 2.5], np.float32)}
    w = outputs.CourtJsonWriter(str(tmp_path), "game7", "run_42")
    w.add_batch(["clip/000001", "000002"], preds)
    path = w.close()
    assert os.path.basename(path) == "game7_court.json" and not os.path.exists(w.tmp_path)
    raw = json.load(open(path))
    assert list(raw) == ["000001", "000002", "model"] and raw["model"] == "run_42"
    assert raw["000001"]["score"] == 0.123457 and raw["000002"]["score"] == 2.5
    assert np.array(raw["000001"]["theta"]).shape == (1, 3, 3)
    assert np.allclose(np.array(raw["000002"]["poi"]), poi[1])
    assert open(path).read().startswith('{\n  "000001": {\n    "score"')  # indent=2
    frames, model = outputs.load_court_mapping(path)
    assert model == "run_42"
    f2c, c2f, score = frames["000001"]
    assert np.allclose(f2c, th[0, 0]) and np.allclose(f2c @ c2f, np.eye(3), atol=1e-5) and score == 0.123457


def test_palette_matches_reference_table():
    ids = np.arange(8, dtype=np.uint8).reshape(1, 2, 4)
    for nc in (4, 7, 8):
        want = post_ref.onehot_to_image(ids % nc, nc)[0].reshape(-1, 3)
        pal = outputs._palette_bytes(nc)
        assert np.array_equal(pal[(ids % nc).reshape(-1)], want)
    with pytest.raises(NotImplementedError):
        outputs._palette_bytes(5)


# ---------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("mask_type", ["gray", "bin", "rgb"])
@pytest.mark.parametrize("out_size", [None, (1280, 720), (100, 37), (642, 361)])
def test_format_masks_vs_oracle(mask_type, out_size):
    g = torch.Generator().manual_seed(5)
    B, nc, H, W = 3, 4, 90, 160
    logits = torch.randn(B, nc, H, W, generator=g)
    logits[:, :, :4, :4] = 0.25  # exact ties -> first maximum
    ids = torch_ref.preds_to_masks(logits).numpy().astype(np.uint8)
    want = post_ref.format_masks(ids, mask_type, nc, out_size or (W, H))
    for src in (logits.cuda(), torch.from_numpy(ids.astype(np.int32)).cuda(), torch.from_numpy(ids).cuda()):
        got = outputs.format_masks(src, mask_type, nc, out_size).cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
def test_transfer_gpu_to_cpu_keys():
    g = torch.Generator().manual_seed(6)
    preds = {"logits": torch.randn(2, 4, 36, 64, generator=g).cuda(),
             "warp_mask": torch.randint(0, 4, (2, 36, 64), generator=g, dtype=torch.int32).cuda(),
             "theta": torch.randn(2, 1, 3, 3, generator=g).cuda(),
             "consist_score": torch.rand(2, generator=g).cuda(), "name": ["a", "b"]}
    out = outputs.transfer_gpu_to_cpu(preds, ["segm_mask", "theta"], 4)
    assert sorted(out) == ["consist_score", "name", "segm_mask", "theta"]
    assert out["segm_mask"].dtype == np.uint8
    assert np.array_equal(out["segm_mask"], preds["logits"].cpu().argmax(1).numpy().astype(np.uint8))
    out = outputs.transfer_gpu_to_cpu(preds, ["warp_mask", "poi"], 4)
    assert sorted(out) == ["consist_score", "name", "warp_mask"]
    assert np.array_equal(out["warp_mask"], preds["warp_mask"].cpu().numpy().astype(np.uint8))
    with pytest.raises(NotImplementedError):
        outputs.preds_to_masks(preds["logits"][:, :1].contiguous(), 1)


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1, 3])
def test_frame_pipeline_gives_predict_s_outputs(scale):
    """sfh_amd.pipeline.FramePipeline (host uint8 frames in, host arrays out, two batches in flight) returns, for every
    batch, exactly what frames_u8_to_input -> predict() -> transfer_gpu_to_cpu returns for it - also when the decoded
    frames are 3x the network size (INTER_AREA downscale on the GPU)."""
    from sfh_amd import engine as E
    from sfh_amd.pipeline import FramePipeline
    from sfh_amd.reconstructor import Reconstructor
    w, h, B = 112, 90, 2
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :h, :w].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(w, h), unet_size=(w, h), warp_size=(w, h), warp_with_nearest=True)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 19))
    net.cuda().eval()
    batches = [torch.from_numpy(synth.synth_frames_u8(B, h * scale, w * scale, seed=40 + k)).pin_memory() for k in range(5)]
    req = ("theta", "warp_mask", "segm_mask", "poi")
    pipe = FramePipeline(net, B, (h * scale, w * scale), req_outputs=req, consistency=True)
    with torch.no_grad():
        got = list(pipe.run(iter(batches)))
        assert len(got) == len(batches)
        for fr, res in zip(batches, got):
            x = E.frames_u8_to_input(fr.cuda(), (w, h) if scale != 1 else None)
            want = O.transfer_gpu_to_cpu(net.predict(x, consistency=True, project_poi=True), set(req), 4)
            assert set(res) == {"theta", "warp_mask", "segm_mask", "poi", "consist_score"}
            for k in res:
                assert res[k].dtype == want[k].dtype and np.array_equal(res[k], want[k]), k
            assert res["segm_mask"].dtype == np.uint8 and res["warp_mask"].dtype == np.uint8


@pytest.mark.gpu
def test_frame_pipeline_tickets_enforce_the_slot_order():
    """Round 5 (review finding): a batch's ticket is its own object (slot, generation, own events).  The orders that would
    silently hand out another batch's data raise instead: fetching a ticket whose host buffers were given to the batch two
    submissions later, collecting into buffers whose batch was never fetched, reusing a slot that was not collected; and
    wait_uploaded() says when the caller's pinned frame buffer may be refilled."""
    from sfh_amd.pipeline import FramePipeline
    from sfh_amd.reconstructor import Reconstructor
    w, h, B = 112, 90, 2
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :h, :w].contiguous()
    poi = synth.load_court_poi("pitch", B)
    net = Reconstructor(court.cuda(), poi.cuda(), target_size=(w, h), unet_size=(w, h), warp_size=(w, h), warp_with_nearest=True)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 19))
    net.cuda().eval()
    fr = [torch.from_numpy(synth.synth_frames_u8(B, h, w, seed=60 + k)).pin_memory() for k in range(4)]
    pipe = FramePipeline(net, B, (h, w), req_outputs=("theta",))
    with torch.no_grad():
        t0 = pipe.submit(fr[0])
        t1 = pipe.submit(fr[1])
        assert t0 is not t1 and t0.slot is not t1.slot and (t0.gen, t1.gen) == (1, 1)
        with pytest.raises(RuntimeError, match="collect"):
            pipe.submit(fr[2])                       # slot 0 still holds batch 0
        with pytest.raises(RuntimeError, match="collect"):
            pipe.get(t0)                             # not collected yet
        pipe.collect(t0)
        with pytest.raises(RuntimeError, match="pending"):
            pipe.collect(t0)                         # twice
        t2 = pipe.submit(fr[2])                      # slot 0 again: a NEW ticket, generation 2
        assert t2 is not t0 and t2.slot is t0.slot and t2.gen == 2
        pipe.wait_uploaded(t2)
        buf = fr[2].clone()
        fr[2].zero_()                                # refilling the host buffer after wait_uploaded() changes nothing
        th0 = np.array(pipe.get(t0)["theta"])        # run()'s own order: fetch batch k after submitting batch k + 2
        pipe.collect(t1)
        with pytest.raises(RuntimeError, match="never fetched"):
            t3 = pipe.submit(fr[3])
            pipe.collect(t2)                         # fine: batch 0 was fetched
            pipe.collect(t3)                         # slot 1's buffers still hold batch 1, never fetched
        th1 = np.array(pipe.get(t1)["theta"])
        pipe.collect(t3)
        with pytest.raises(RuntimeError, match="generation"):
            pipe.get(t1)                             # its buffers now belong to batch 3
        th2, th3 = np.array(pipe.get(t2)["theta"]), np.array(pipe.get(t3)["theta"])
        from sfh_amd import engine as E
        for f, th in zip((fr[0], fr[1], buf, fr[3]), (th0, th1, th2, th3)):
            want = net.predict(E.frames_u8_to_input(f.cuda()), consistency=False)["theta"].cpu().numpy()
            assert np.array_equal(th, want)
