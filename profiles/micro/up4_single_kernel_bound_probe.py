"""Review item 6b (the fused Up block at u4 as ONE kernel), the measurement that bounds it.  Today: the composed 2x2 conv over the
low-resolution tensor (u4.fuse: 4 taps x 128 channels = K 512, writes an fp32 partial) + the skip-half 3x3 conv (u4.skip: K 576,
starts from the partial): two launches, 2 x 0.94 GB of partial traffic.  A single kernel would accumulate both halves in registers:
K = 1088 per output, no partial.  Its BEST case is a plain 3x3 launch of this kernel family with that much K per output and 64
couts at 360x640 - measured here as 128 -> 64 (K = 1152), scaled by 1088 / 1152 - before the costs the analysis in DESIGN.md
section 4 lists (parity-grouped pixel groups, four weight sets per wave, two halo geometries).
usage (GPU box): python profiles/micro/up4_single_kernel_bound_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import conv_rate_probe as P  # noqa: E402

print("plain 3x3 launches at 360x640 x 16 (conv_s3_kernel, launcher's choice):")
P.run(128, 64, 640, (360,))
P.run(64, 64, 640, (360,))
P.run(128, 64, 640, (360,))
P.run(64, 64, 640, (360,))
print("single-kernel Up block at u4, best case = the 128 -> 64 time x 1088 / 1152; today's pair: u4.fuse + u4.skip in profiles/r05_layer_table.txt")
