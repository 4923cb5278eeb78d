"""Builds libsfh_amd.so (hipcc, gfx950 only) in-tree next to this file.

Plain ``hipcc -c`` per source + one link; no torch C++ ABI, no cmake.  Called by
``__graft_entry__.build()`` and usable stand-alone: ``python -m sfh_amd.build``.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libsfh_amd.so")
SOURCES = ["capi.hip", "conv_mfma.hip", "conv_s3.hip", "conv_c4h2.hip", "pointwise.hip", "warp.hip", "train.hip", "stem.hip",
           "wgrad_s3.hip", "probe.hip", "conv_small.hip", "conv_upfused.hip", "hostprep.hip"]
# warp.hip's coordinate arithmetic must not be contracted into FMAs (bit-exact nearest
# sampling against oracle/warp_ref.py); the flag is harmless elsewhere.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]
# The SLP vectoriser packs scalar fp32 arithmetic into v_pk_*_f32 instructions, which issue at half rate on gfx950:
# warp.hip is VALU-issue-bound (measured: profiles/micro/warp_variants.hip), and the conv epilogues (scale, shift,
# ReLU, plane split) run beside the co-resident workgroup's MFMAs, which already take half of the SIMD's issue slots
# (conv_s3.hip with / without the flag: 9.84-9.92 / 9.47-9.61 ms per batch for the DoubleConv launches).
_NO_SLP = ["-fno-slp-vectorize"]
EXTRA_FLAGS = {"warp.hip": _NO_SLP, "conv_s3.hip": _NO_SLP, "conv_mfma.hip": _NO_SLP, "stem.hip": _NO_SLP,
               "conv_c4h2.hip": _NO_SLP, "conv_small.hip": _NO_SLP, "conv_upfused.hip": _NO_SLP}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_epilogue.h"),
               os.path.join(os.path.dirname(_HERE), "include", "sfh_amd.h")]
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_diag(verbose=True, clock_only=False):
    """Diagnostic variant with in-kernel s_memtime stamps (profiles/diag_stamps.py); never loaded
    by the product path.  clock_only: no phase stamps (they fence the schedule), only the in-kernel clock probe
    of conv_s3_kernel (libsfh_amd_clock.so)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.join(_HERE, "libsfh_amd_clock.so" if clock_only else "libsfh_amd_diag.so")
    # -fgpu-rdc: the stamp accumulator (conv_mfma.hip) is referenced from other translation units
    cmd = [hipcc] + FLAGS + _NO_SLP + ["-DSFH_DIAG_STAMPS"] + (["-DSFH_DIAG_CLOCK_ONLY"] if clock_only else []) + [
        "-fgpu-rdc", "-shared", "-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--diag" in sys.argv:
        build_diag()
    elif "--clock" in sys.argv:
        build_diag(clock_only=True)
    else:
        build(force="--force" in sys.argv)
