"""Per-kernel times of the ResNet-STN part of the last step in a rocprofv3 kernel trace."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name'] or 'frame_to_h2' in r['Kernel_Name']]
step = rows[idx[-1]:]
k = [i for i, r in enumerate(step) if 'outconv' in r['Kernel_Name'] or 'stem7x7' in r['Kernel_Name']][0]
tot = 0
for r in step[k:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Kernel_Name']
    m = re.search(r'(S3Cfg|ConvCfg)<([^>]*)>(, (true|false))?', n)
    short = (m.group(1) + '<' + m.group(2) + '>' + (m.group(3) or '')) if m else n[:50]
    print(f"{short:50s} {d:8.1f} us blocks={int(r['Grid_Size_X'])//256}")
    tot += d
print(f"total {tot:.1f} us")
