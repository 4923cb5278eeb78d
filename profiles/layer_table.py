"""Per-layer table from a rocprofv3 --kernel-trace CSV of bench.py (last step of the run).
usage: python profiles/layer_table.py <kernel_trace.csv> [batch]"""
import csv, os, re, sys
# several traces may be given (gpurun merges every call's files into the local directory): the newest one counts
paths = [a for a in sys.argv[1:] if a.endswith(".csv")]
rows = list(csv.DictReader(open(max(paths, key=os.path.getmtime))))
B = int(sys.argv[-1]) if len(sys.argv) > 2 and sys.argv[-1].isdigit() else 16
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name'] or 'frame_to_h2' in r['Kernel_Name']]
step = rows[idx[-1]:]
def gm(cin, cout, h, w, k=9): return cin * cout * h * w * k / 1e9
L = [("inc.0", gm(3, 64, 360, 640)), ("inc.3", gm(64, 64, 360, 640)),
     ("d1.0", gm(64, 128, 180, 320)), ("d1.3", gm(128, 128, 180, 320)),
     ("d2.0", gm(128, 256, 90, 160)), ("d2.3", gm(256, 256, 90, 160)),
     ("d3.0", gm(256, 512, 45, 80)), ("d3.3", gm(512, 512, 45, 80)),
     ("d4.0", gm(512, 1024, 22, 40)), ("d4.3", gm(1024, 1024, 22, 40)),
     ("u1.up", gm(1024, 512, 22, 40, 4)), ("u1.0", gm(1024, 512, 45, 80)), ("u1.3", gm(512, 512, 45, 80)),
     ("u2.up", gm(512, 256, 45, 80, 4)), ("u2.0", gm(512, 256, 90, 160)), ("u2.3", gm(256, 256, 90, 160)),
     ("u3.up", gm(256, 128, 90, 160, 4)), ("u3.0", gm(256, 128, 180, 320)), ("u3.3", gm(128, 128, 180, 320)),
     ("u4.up", gm(128, 64, 180, 320, 4)), ("u4.0", gm(128, 64, 360, 640)), ("u4.3", gm(64, 64, 360, 640))]
if any('S3Cfg<2,' in r['Kernel_Name'] for r in step):
    # fused Up blocks (no F.pad at that level): skip-half 3x3 conv + composed 2x2 quadrant conv over the
    # low-resolution tensor, credited with the u-half of the reference's 3x3 conv; no ConvTranspose launch
    # (u1 / u2: .a / .b in launch order - cfg<2, …> is the composed 2x2 launch, cfg<3, …> the skip half)
    L = L[:10] + [("u1.a", gm(512, 512, 45, 80)), ("u1.b", gm(512, 512, 45, 80)), ("u1.3", gm(512, 512, 45, 80)),
                  ("u2.a", gm(256, 256, 90, 160)), ("u2.b", gm(256, 256, 90, 160)), ("u2.3", gm(256, 256, 90, 160)),
                  # u3 / u4: the composed 2x2 conv runs first, the skip-half 3x3 conv finishes (engine.UNetEngine.up_swap)
                  ("u3.fuse", gm(128, 128, 180, 320)), ("u3.skip", gm(128, 128, 180, 320)), ("u3.3", gm(128, 128, 180, 320)),
                  ("u4.fuse", gm(64, 64, 360, 640)), ("u4.skip", gm(64, 64, 360, 640)), ("u4.3", gm(64, 64, 360, 640))]
if any('conv_upfused' in r['Kernel_Name'] for r in step):
    # round 5: u3 / u4's first conv is ONE kernel (csrc/conv_upfused.hip), credited with the whole 3x3 conv it stands for
    names = [n for n, _ in L]
    for lv, g in (("u3", gm(256, 128, 180, 320)), ("u4", gm(128, 64, 360, 640))):
        i = names.index(lv + ".fuse")
        L[i:i + 2] = [(lv + ".one", g)]
        names = [n for n, _ in L]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
is_conv = lambda n: 'conv_mfma' in n or 'conv_s3' in n or 'conv_upfused' in n or 'conv_small' in n or 'conv3x3_c4' in n or 'stem7x7' in n   # (c4: also conv3x3_c4h2)
convs = [r for r in step if is_conv(r['Kernel_Name'])]
tot = 0
for (nm, g), r in zip(L, convs[:len(L)]):
    d = dur(r)
    m_ = re.search(r'Cfg<([^>]*)>', r['Kernel_Name'])
    cfg = m_.group(1) if m_ else ('single-kernel Up' if 'upfused' in r['Kernel_Name'] else 'c4 tap-packed')
    print(f"{nm:6s} cfg<{cfg:22s}> {d:7.3f} ms {2*g*B/d:7.1f} TFLOP/s grid={r.get('Grid_Size_X')} vgpr={r.get('VGPR_Count')} lds={r.get('LDS_Block_Size')}")
    tot += d
print(f"UNet conv launches: {tot:.3f} ms")
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
print(f"step span {(t1-t0)/1e6:.3f} ms over {len(step)} kernels; busy {sum(dur(r) for r in step):.3f} ms")
rn = convs[len(L):]
print(f"ResNet conv launches: {sum(dur(r) for r in rn):.3f} ms ({len(rn)} launches)")
for r in step:
    if not is_conv(r['Kernel_Name']):
        print(f"  {r['Kernel_Name'][:70]:70s} {dur(r)*1e3:9.1f} us")
