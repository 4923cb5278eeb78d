"""Output formats on the far side of the hot path.

Mirrors what the reference's ``predict.py`` does with the dict returned by
``Reconstructor.predict``:

* ``preds_to_masks``            - utils/postprocess.py:7-18 (argmax -> uint8 class ids), on the GPU
* ``format_masks``              - predict.py:286-315: mask_type gray / bin / rgb + nearest resize to
                                  ``out_size``, one HIP pass producing the uint8 image that is written
* ``transfer_gpu_to_cpu``       - predict.py:79-121 (which keys survive, dtypes on the host)
* ``MaskPickleWriter`` / ``MaskReader`` - the PNG-in-pickle stream ``[name, png_bytes]``
                                  (predict.py:26-37, read back by viz_preds.py:52-75)
* ``CourtJsonWriter``           - ``{game}_court.json`` = ``{frame: {score, theta, poi}, ..., model}``
                                  (predict.py:343-357,399-407; consumed by utils/court.py:33-45)

The reference encodes PNGs with OpenCV, which is absent here; ``encode_png``/``decode_png`` are a
small zlib PNG codec for 8-bit gray / 3-channel images.  3-channel arrays are treated as BGR like
``cv2.imencode``/``cv2.imdecode`` do (stored RGB in the file), so streams are interchangeable.
"""
import json
import os
import pickle
import struct
import zlib

import numpy as np
import torch

from . import _lib
from .engine import _ptr, _stream

# class id -> colour, utils/postprocess.py:29-51 (tuples are written into the array as given)
_PALETTES = {
    4: {1: (0, 255, 0), 2: (255, 0, 0), 3: (0, 0, 255)},
    7: {1: (0, 255, 0), 2: (255, 0, 0), 3: (0, 0, 255), 4: (255, 255, 255), 5: (255, 0, 255), 6: (0, 255, 255)},
    8: {1: (0, 255, 0), 2: (255, 0, 0), 3: (0, 0, 255), 4: (255, 255, 255), 5: (255, 0, 255), 6: (0, 255, 255),
        7: (255, 255, 0)},
}
_MODES = {"gray": 0, "bin": 1, "rgb": 2}


def _palette_bytes(n_classes):
    if n_classes not in _PALETTES:
        raise NotImplementedError(f"no colour table for mask_classes={n_classes} (reference: 4, 7, 8)")
    pal = np.zeros((8, 3), dtype=np.uint8)
    for k, c in _PALETTES[n_classes].items():
        pal[k] = c
    return pal


def format_masks(src, mask_type="gray", n_classes=4, out_size=None):
    """uint8 output masks on the GPU.

    src: logits (B,nc,H,W) float32, a warp_mask (B,H,W) int32, or an id mask (B,H,W) uint8, on the
    GPU.  out_size = (W,H) like ``args.out_size`` (default: source size).  Returns a uint8 tensor
    (B,Hout,Wout) for gray/bin, (B,Hout,Wout,3) for rgb."""
    lib = _lib.load()
    if mask_type not in _MODES:
        raise NotImplementedError(f"mask_type={mask_type!r}")
    if not src.is_cuda or not src.is_contiguous():
        raise ValueError("format_masks needs a contiguous GPU tensor")
    if src.dim() == 4 and src.dtype == torch.float32:
        kind, nc = 2, src.shape[1]
        if nc < 2:
            raise NotImplementedError("single-channel (sigmoid) logits have no class-id mask")
        B, hs, ws = src.shape[0], src.shape[2], src.shape[3]
    elif src.dim() == 3 and src.dtype in (torch.int32, torch.uint8):
        kind, nc = (0 if src.dtype == torch.int32 else 1), n_classes
        B, hs, ws = src.shape
    else:
        raise ValueError(f"format_masks: unsupported source {tuple(src.shape)} {src.dtype}")
    wd, hd = (ws, hs) if out_size is None else (int(out_size[0]), int(out_size[1]))
    mode = _MODES[mask_type]
    pal = _palette_bytes(n_classes) if mode == 2 else None
    shape = (B, hd, wd, 3) if mode == 2 else (B, hd, wd)
    out = torch.empty(shape, dtype=torch.uint8, device=src.device)
    _lib.check(lib.sfh_mask_format_fwd(_ptr(src), kind, nc, B, hs, ws, hd, wd, mode,
                                       pal.ctypes.data if pal is not None else None, _ptr(out), _stream()),
               "mask_format")
    return out


def preds_to_masks(preds, n_classes=1, to_ndaray=True):
    """utils/postprocess.py:7-18 for n_classes > 1: argmax over the class axis (softmax is
    monotonic, so it is skipped), uint8.  The argument keeps the reference's spelling."""
    if n_classes <= 1:
        raise NotImplementedError("n_classes == 1 (sigmoid map) is not a class-id mask")
    m = format_masks(preds, "gray", n_classes)
    return m.cpu().numpy() if to_ndaray else m


def transfer_gpu_to_cpu(preds, req_outputs, mask_classes=4):
    """predict.py:92-118: post-process one predict() dict into host arrays, keeping only the
    requested outputs.  Masks are cast to uint8 on the GPU (4x less PCIe traffic than int32)."""
    out = {k: v for k, v in preds.items() if k in ("name", "orig_img")}
    if "segm_mask" in req_outputs and "logits" in preds:
        out["segm_mask"] = preds_to_masks(preds["logits"], mask_classes)
    if "warp_mask" in req_outputs and "warp_mask" in preds:
        wm = preds["warp_mask"]
        wm = wm if wm.dtype == torch.int32 else wm.to(torch.int32)
        out["warp_mask"] = format_masks(wm.contiguous(), "gray", mask_classes).cpu().numpy()
    if "theta" in req_outputs and "theta" in preds:
        out["theta"] = preds["theta"].cpu().numpy()
    if "consist_score" in preds:
        out["consist_score"] = preds["consist_score"].cpu().numpy()
    if "poi" in req_outputs and "poi" in preds:
        out["poi"] = preds["poi"].cpu().numpy()
    return out


# ------------------------------------------------------------------------------- PNG codec
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_png(img, level=1):
    """8-bit gray (H,W) or BGR (H,W,3) array -> PNG file bytes as a 1-D uint8 array (what
    ``cv2.imencode('.png', img)[1]`` holds)."""
    a = np.ascontiguousarray(img)
    if a.dtype == np.bool_ or a.dtype.kind in "iu":
        if a.size and (a.min() < 0 or a.max() > 255):
            raise ValueError("encode_png: values outside 0..255")
        a = a.astype(np.uint8)
    else:
        raise ValueError(f"encode_png: dtype {a.dtype}")
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    if a.ndim == 2:
        ctype = 0
    elif a.ndim == 3 and a.shape[2] == 3:
        ctype, a = 2, a[:, :, ::-1]  # BGR in memory -> RGB in the file
    else:
        raise ValueError(f"encode_png: shape {a.shape}")
    h, w = a.shape[:2]
    rows = np.ascontiguousarray(a).reshape(h, -1)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()  # filter 0 per scanline
    png = (_PNG_SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
           + _chunk(b"IDAT", zlib.compress(raw, level)) + _chunk(b"IEND", b""))
    return np.frombuffer(png, dtype=np.uint8)


def decode_png(buf):
    """Inverse of encode_png for 8-bit gray / RGB / RGBA non-interlaced PNGs (all five scanline
    filters).  3/4-channel images come back BGR(A) like ``cv2.imdecode(buf, IMREAD_UNCHANGED)``."""
    data = bytes(np.asarray(buf, dtype=np.uint8).reshape(-1))
    if data[:8] != _PNG_SIG:
        raise ValueError("decode_png: not a PNG")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if zlib.crc32(tag + body) & 0xFFFFFFFF != struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]:
            raise ValueError("decode_png: CRC mismatch")
        pos += 12 + n
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    nch = {0: 1, 2: 3, 6: 4}.get(ctype)
    if depth != 8 or nch is None or interlace:
        raise NotImplementedError(f"decode_png: depth {depth} colour type {ctype} interlace {interlace}")
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * nch)
    ftype, rows = raw[:, 0], raw[:, 1:]
    if not ftype.any():
        out = rows.copy()
    else:
        out = np.zeros((h, w * nch), dtype=np.uint8)
        prev = np.zeros(w * nch, dtype=np.int32)
        for y in range(h):
            cur = rows[y].astype(np.int32)
            f = int(ftype[y])
            if f == 2:
                cur = (cur + prev) & 255
            elif f != 0:
                for i in range(w * nch):
                    a = cur[i - nch] if i >= nch else 0
                    b = prev[i]
                    c = prev[i - nch] if i >= nch else 0
                    if f == 1:
                        p = a
                    elif f == 3:
                        p = (a + b) >> 1
                    else:
                        pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                        p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                    cur[i] = (cur[i] + p) & 255
            out[y] = cur
            prev = cur
    out = out.reshape(h, w, nch)
    if nch == 1:
        return out[:, :, 0]
    if nch == 3:
        return np.ascontiguousarray(out[:, :, ::-1])
    return np.ascontiguousarray(out[:, :, [2, 1, 0, 3]])


# --------------------------------------------------------------------- mask streams on disk
def save_mask_as_png(mask, dst_dir, name, postfix="mask"):
    """predict.py:20-25."""
    sub = os.path.join(dst_dir, postfix)
    os.makedirs(sub, exist_ok=True)
    with open(os.path.join(sub, name + ".png"), "wb") as f:
        f.write(encode_png(mask).tobytes())


class MaskPickleWriter:
    """``{dst_dir}/{postfix}/data.pkl``: a sequence of ``pickle.dump([name, png_buffer])`` records
    (predict.py:26-37)."""

    def __init__(self, dst_dir, postfix="mask"):
        sub = os.path.join(dst_dir, postfix)
        os.makedirs(sub, exist_ok=True)
        self.path = os.path.join(sub, "data.pkl")
        self._f = open(self.path, "wb+")

    def write(self, name, mask):
        pickle.dump([name, encode_png(mask)], self._f)

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class MaskReader:
    """viz_preds.py:52-75."""

    def __init__(self, path):
        self.entries = []
        with open(path, "rb") as f:
            while True:
                try:
                    self.entries.append(pickle.load(f))
                except EOFError:
                    break

    def get(self, decode=False):
        for name, buf in self.entries:
            yield name, (decode_png(buf) if decode else buf)

    decode = staticmethod(decode_png)


# ------------------------------------------------------------------------------ court json
def _json_default(obj):
    """arrays and numpy scalars -> plain lists / numbers (the court json holds theta and poi as nested lists)"""
    if hasattr(obj, "tolist"):
        return obj.tolist()
    raise TypeError(f"{type(obj).__name__} is not JSON serialisable")


def format_score(score):
    """predict.py:349: ``float('{:5f}'.format(score))`` (6 decimals)."""
    return float("{:5f}".format(float(score)))


class CourtJsonWriter:
    """Streams one JSON line per frame to ``{game}_court_processing.json`` and, on close,
    rewrites it as ``{game}_court.json`` = ``{frame: {...}, ..., "model": name}`` with indent 2
    (predict.py:343-357,399-407)."""

    def __init__(self, dst_dir, game_name, model_name):
        os.makedirs(dst_dir, exist_ok=True)
        self.tmp_path = os.path.join(dst_dir, f"{game_name}_court_processing.json")
        self.path = os.path.join(dst_dir, f"{game_name}_court.json")
        self.model_name = model_name
        self._f = open(self.tmp_path, "w+")

    def add(self, name, score=None, theta=None, poi=None):
        rec = {}
        if score is not None:
            rec["score"] = format_score(score)
        if theta is not None:
            rec["theta"] = np.asarray(theta)  # (1,3,3)
        if poi is not None:
            rec["poi"] = np.asarray(poi)
        json.dump({name: rec}, self._f, default=_json_default)
        self._f.write("\n")

    def add_batch(self, names, preds):
        """preds: host dict from transfer_gpu_to_cpu."""
        for i, n in enumerate(names):
            t = n.split("/")  # predict.py:318-323: "subdir/name" -> name
            self.add(t[1] if len(t) == 2 else t[0],
                     preds["consist_score"][i] if "consist_score" in preds else None,
                     preds["theta"][i] if "theta" in preds else None,
                     preds["poi"][i] if "poi" in preds else None)

    def close(self):
        if self._f is None:
            return self.path
        self._f.close()
        self._f = None
        with open(self.tmp_path, "r") as f:
            output = {k: v for line in f for k, v in json.loads(line).items()}
        output["model"] = self.model_name
        with open(self.path, "w") as f:
            json.dump(output, f, default=_json_default, indent=2)
        os.remove(self.tmp_path)
        return self.path

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def load_court_mapping(path):
    """utils/court.py:33-45: {frame: (theta_f2c, theta_c2f, score)} from a ``*_court.json``."""
    with open(path, "r") as f:
        raw = json.load(f)
    model = raw.pop("model", None)
    frames = {}
    for fid, d in raw.items():
        t = np.array(d["theta"])[0]
        frames[fid] = (t, np.linalg.inv(t), float(d["score"]))
    return frames, model
