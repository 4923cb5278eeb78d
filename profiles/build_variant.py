"""Build a VARIANT of the library for an A/B on one device (profiles/ab_bench.sh):
    python profiles/build_variant.py <name> -DFLAG[=v] ...   ->  sports-field-homography_amd/libsfh_amd_<name>.so
The product never loads it (SFH_AMD_LIB selects it for a bench run)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sfh_amd.build as b  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(b._HERE, f"libsfh_amd_{name}.so")
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + b.FLAGS + b._NO_SLP + flags + ["-shared", "-o", out] + [
    os.path.join(b.CSRC, s) for s in b.SOURCES]
print(" ".join(cmd), flush=True)
subprocess.check_call(cmd)
print(out)
