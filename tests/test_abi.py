"""CPU-side checks of the C-ABI boundary: the library loads without a GPU and exports
exactly the entry points declared in include/sfh_amd.h; argument validation fails loudly."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "sfh_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sfh_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound():
    from sfh_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import sfh_amd.build as b
        b.build(verbose=False)
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sfh_amd.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table and header disagree"
    assert lib.sfh_version() >= 100


def test_conv_desc_layout_matches_header():
    """Field order of the ctypes mirror == field order of struct sfh_conv_desc."""
    from sfh_amd import _lib
    txt = open(os.path.join(ROOT, "include", "sfh_amd.h")).read()
    body = txt[txt.index("typedef struct sfh_conv_desc {"):txt.index("} sfh_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S).replace("typedef struct sfh_conv_desc {", "")
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        m = re.match(r"(const float\*|float\*|double\*|uint32_t\*|int32_t|int64_t)\s+(.*)", decl, flags=re.S)
        if m:
            fields += [f.strip() for f in m.group(2).split(",")]
    assert fields == [f[0] for f in _lib.ConvDesc._fields_]


def test_argument_validation_without_gpu():
    from sfh_amd import _lib
    lib = _lib.load()
    assert lib.sfh_packed_weight_floats(3, 64, 0, 64) == 4 * 9 * 1024
    assert lib.sfh_packed_weight_floats(3, 64, 64, 128) == 2 * 8 * 9 * 1024
    assert lib.sfh_packed_weight_floats(5, 64, 0, 64) == -1       # unsupported kernel size
    assert lib.sfh_packed_weight_floats(3, 64, 0, 48) == -1       # cout not a multiple of 64
    # split-operand formats: 3 bf16 planes (S3) / 2 fp16 planes (H2) of [cout/64][cin/32][tap][plane][4 KB]
    assert lib.sfh_packed_s3_weight_bytes(3, 64, 0, 64) == 2 * 9 * 3 * 4096
    assert lib.sfh_packed_h2_weight_bytes(3, 64, 0, 64) == 2 * 9 * 2 * 4096
    assert lib.sfh_packed_h2_weight_bytes(3, 48, 0, 64) == -1     # cin not a multiple of 32
    d = _lib.ConvDesc()
    rc = lib.sfh_conv_fwd(ctypes.byref(d), None)                  # all-null descriptor
    assert rc == -1 and b"null" in lib.sfh_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "conv_fwd")
    rc = lib.sfh_homography_warp_fwd(None, None, 0, 360, 640, 16, 360, 640, 0, 4.0, None, None, None)
    assert rc == -1
    # round 6 entries: argument checks fire before anything touches a device
    assert lib.sfh_vec_op(1, None, None, 16, 0, 1.0, None, None) == -1
    assert lib.sfh_multi_absminmax(None, 3, None, None) == -1
    assert lib.sfh_stn_input_assemble(None, 4, None, 3, None, 0, 2, 8, 8, 4, None, None) == -1      # 7 channels into 4
    assert b"7" in lib.sfh_last_error() or b"channels" in lib.sfh_last_error()
    assert lib.sfh_uv_loss(None, None, None, 5, 2, 2, 8, 8, 1.0, 1, None, None, None) == -1
    assert lib.sfh_adam_step(None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.1, 1.0, 1, None) == -1
    assert lib.sfh_grad_scale(None, 1, 1, 0, None, None, None) == -1


def test_model_refuses_cpu_execution():
    import torch
    from sfh_amd import synth
    from sfh_amd.reconstructor import Reconstructor
    net = Reconstructor(synth.load_court_template(batch_size=1), synth.load_court_poi(batch_size=1)).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net.predict(torch.zeros(1, 3, 360, 640))


def test_missing_library_fails_loudly(tmp_path):
    """No HIP extension -> an explicit error, never a silent CPU / PyTorch path."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['SFH_AMD_LIB'] = %r\n"
            "from sfh_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.SfhError as e:\n    print('SfhError:', e); sys.exit(7)\n"
            "sys.exit(0)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "nope.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 7, r.stdout + r.stderr
    assert "not found" in r.stdout and "no CPU/PyTorch fallback" in r.stdout
