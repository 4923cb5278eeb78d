"""Frames in, results out: the GPU side of the reference's inference loop as one double-buffered pipeline.

The reference's ``predict.py`` runs three processes (predict.py:45-122,250-255): a DataLoader that decodes and
preprocesses frames on the host (utils/dataset.py:145-161,310-330), ``Workers.predict`` (frames ``.to(device)``,
``net.predict``) and ``Workers.transfer_gpu_to_cpu`` (``preds_to_masks`` + ``.cpu().numpy()`` of every requested
output).  ``FramePipeline`` is the device part of that loop in ONE process with HIP streams instead of processes:

    host uint8 HWC frames (pinned) --copy stream--> GPU uint8 --HIP--> /255, HWC->CHW (+ INTER_AREA downscale)
        --> Reconstructor.predict_async (UNet on the caller's stream, ResNet-STN / warp / CE on its side stream)
        --> uint8 arg-max mask, uint8 warp mask, theta, consistency score, POI --copy stream--> pinned host arrays

Two slots alternate: while batch k computes, batch k + 1 uploads and batch k - 1 downloads.  Everything a batch needs
on the host arrives in its slot's pinned buffers; ``get()`` waits for that batch's download event only.

Every batch has its own TICKET (slot, generation, its own upload / download events).  The order a caller must keep is
checked, not assumed: ``submit`` refuses a slot whose previous batch was not collected; ``collect`` refuses to download
into host buffers whose previous batch was never fetched with ``get``; ``get`` refuses a ticket whose host buffers have
meanwhile been given to a later batch.  ``wait_uploaded(ticket)`` (or ``ticket.uploaded``) tells when the caller's host
frame buffer has been read and may be refilled.

Outputs and dtypes are those of predict.py:92-118 (``outputs.transfer_gpu_to_cpu`` is the synchronous version):
``segm_mask`` uint8 (B,H,W), ``warp_mask`` uint8 (B,H,W), ``theta`` float32 (B,1,3,3), ``consist_score`` float32 (B,),
``poi`` float32 (B,N,2).
"""
import numpy as np
import torch

from . import engine as E
from . import outputs as O


# ONE copy stream per device, shared by every FramePipeline of the process, for uploads AND downloads.  HIP multiplexes streams
# onto a handful of hardware queues (four by default) in creation order: with an upload stream and a download stream per pipeline
# object, every second pipeline a process created got streams that share a queue with the model's compute / side streams, and its
# copies then waited behind kernels (measured: 13.2 ms per batch from the first pipeline, 14.2 from the second, 13.3 from the third
# ...; 13.7 / 16.0 with 1920x1080 frames; GPU_MAX_HW_QUEUES=8 removes the alternation: profiles/micro/e2e_parity_probe.py).  The
# two directions need no stream of their own: an upload is 0.25-1.9 ms, a download 0.14 ms, per 13 ms batch.
_COPY_STREAMS = {}


def _copy_stream(dev):
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _COPY_STREAMS.get(key)
    if st is None:
        st = _COPY_STREAMS[key] = torch.cuda.Stream(dev)
    return st


class Ticket:
    """one submitted batch: which slot it uses, its generation on that slot, its own events"""
    __slots__ = ("slot", "gen", "uploaded", "downloaded", "handle", "dev", "fetched")

    def __init__(self, slot, gen):
        self.slot, self.gen = slot, gen
        self.uploaded = self.downloaded = self.handle = self.dev = None
        self.fetched = False


class FramePipeline:
    def __init__(self, net, batch, frame_hw, req_outputs=("theta", "warp_mask"), consistency=False, channels=3):
        """net: a Reconstructor on the GPU in eval mode; frame_hw = (H, W) of the DECODED frames (net.unet_size, or any
        larger size: cv2.INTER_AREA's downscale runs on the GPU, engine.frames_u8_to_input); req_outputs as predict.py's --req_outputs."""
        self.net, self.B = net, int(batch)
        self.req = set(req_outputs)
        self.consistency = bool(consistency) or "consistency" in self.req
        self.poi = "poi" in self.req
        p = next(net.parameters())
        if p.device.type != "cuda":
            raise RuntimeError("FramePipeline needs the model on the GPU (no CPU fallback)")
        self.dev = dev = p.device
        H, W = int(frame_hw[0]), int(frame_hw[1])
        tw, th = net.unet_size
        self.target = None if (W, H) == (tw, th) else (tw, th)
        wh, ww = net._warp_hw
        nc = net.mask_classes
        self.h2d = self.d2h = _copy_stream(dev)
        pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory()
        self.slots = []
        for _ in range(2):
            # pending: the ticket submitted on this slot and not yet collected; collected: the ticket whose results the host
            # buffers hold (or are receiving)
            s = {"u8": torch.empty((self.B, H, W, channels), dtype=torch.uint8, device=dev),
                 "consumed": None, "host": {}, "pending": None, "collected": None, "gen": 0}
            if "segm_mask" in self.req:
                s["host"]["segm_mask"] = pin((self.B, net.target_size[1], net.target_size[0]), torch.uint8)
            if "warp_mask" in self.req and net.warper:
                s["host"]["warp_mask"] = pin((self.B, wh, ww), torch.uint8)
            if "theta" in self.req:
                s["host"]["theta"] = pin((self.B, 1, 3, 3), torch.float32)
            if self.consistency and net.warper and net.use_unet:
                s["host"]["consist_score"] = pin((self.B,), torch.float32)
            if self.poi:
                s["host"]["poi"] = pin((self.B,) + tuple(net.court_poi.shape[1:]), torch.float32)
            self.slots.append(s)
        self.k = 0
        self._nc = nc

    def submit(self, frames_u8_host):
        """frames_u8_host: uint8 (B,H,W,C) host tensor (pinned for a truly asynchronous upload).  Enqueues upload,
        preprocessing and the forward pass of this batch and returns its Ticket; nothing here waits for the GPU.  The host
        buffer is read asynchronously: refill it only after wait_uploaded(ticket)."""
        s = self.slots[self.k % 2]
        if s["pending"] is not None:
            raise RuntimeError("FramePipeline: collect() the batch submitted two calls ago before reusing its slot")
        self.k += 1
        s["gen"] += 1
        t = Ticket(s, s["gen"])
        cur = torch.cuda.current_stream(self.dev)
        with torch.cuda.stream(self.h2d):
            if s["consumed"] is not None:
                self.h2d.wait_event(s["consumed"])       # the preprocessing kernel that read this buffer last is done
            s["u8"].copy_(frames_u8_host, non_blocking=True)
            t.uploaded = torch.cuda.Event()
            t.uploaded.record(self.h2d)
        cur.wait_event(t.uploaded)
        x = E.frames_u8_to_input(s["u8"], self.target)
        s["consumed"] = torch.cuda.Event()
        s["consumed"].record(cur)
        t.handle = self.net.predict_async(x, consistency=self.consistency, project_poi=self.poi)
        s["pending"] = t
        return t

    def wait_uploaded(self, ticket):
        """block until the host frame buffer given to submit() has been read: it may be refilled afterwards"""
        ticket.uploaded.synchronize()

    def collect(self, ticket):
        """Enqueue post-processing and the download of a submitted batch (call it after submitting the NEXT batch, so
        that this batch's ResNet-STN / warp ran under that one's UNet).  Returns at once; get() waits."""
        t, s = ticket, ticket.slot
        if s["pending"] is not t:
            raise RuntimeError("FramePipeline.collect: this ticket is not the batch pending on its slot (collected already, "
                               "or from another pipeline)")
        prev = s["collected"]
        if prev is not None and not prev.fetched:
            raise RuntimeError("FramePipeline.collect: the slot's host buffers still hold a batch that was never fetched with "
                               "get(); this download would overwrite it")
        out = t.handle.result()                          # orders the current stream behind the batch
        t.handle = None
        cur = torch.cuda.current_stream(self.dev)
        devout = {}
        if "segm_mask" in s["host"]:
            devout["segm_mask"] = O.format_masks(out["logits"], "gray", self._nc)          # uint8 arg-max (postprocess.py:7-18)
        if "warp_mask" in s["host"]:
            devout["warp_mask"] = O.format_masks(out["warp_mask"].contiguous(), "gray", self._nc)   # int32 -> uint8 on the GPU
        for k in ("theta", "consist_score", "poi"):
            if k in s["host"]:
                devout[k] = out[k]
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(self.d2h):
            self.d2h.wait_event(ready)
            for k, v in devout.items():
                s["host"][k].copy_(v, non_blocking=True)
                v.record_stream(self.d2h)
            t.downloaded = torch.cuda.Event()
            t.downloaded.record(self.d2h)
        t.dev = devout
        s["pending"], s["collected"] = None, t
        return t

    def get(self, ticket):
        """-> {name: numpy array} of a collected batch (views of the slot's pinned buffers: valid until the NEXT batch of this
        slot is collected, copy what must live longer)."""
        t, s = ticket, ticket.slot
        if t.downloaded is None:
            raise RuntimeError("FramePipeline.get: collect() this batch first")
        if s["collected"] is not t:
            raise RuntimeError(f"FramePipeline.get: the slot's host buffers now hold generation {s['collected'].gen}, this ticket "
                               f"is generation {t.gen} - fetch a batch before the batch two submissions later is collected")
        t.downloaded.synchronize()
        t.dev = None
        t.fetched = True
        return {k: v.numpy() for k, v in s["host"].items()}

    def run(self, batches):
        """Generator over host uint8 batches -> result dicts (copies), two batches in flight.  The producer may hand over the
        SAME pinned buffer every time: the next item is pulled from `batches` only after the upload of the batch just
        submitted has read its buffer (wait_uploaded) - with the one copy stream per device that upload queues behind the
        download of the batch two submissions back, so without the wait a producer that refills one buffer would overwrite
        frames still waiting to be copied."""
        prev = None
        done = None
        it = iter(batches)
        fr = next(it, None)
        while fr is not None:
            t = self.submit(fr)
            if done is not None:
                yield {k: np.array(v) for k, v in self.get(done).items()}
                done = None
            if prev is not None:
                done = self.collect(prev)
            prev = t
            self.wait_uploaded(t)
            fr = next(it, None)
        if done is not None:
            yield {k: np.array(v) for k, v in self.get(done).items()}
        if prev is not None:
            yield {k: np.array(v) for k, v in self.get(self.collect(prev)).items()}
