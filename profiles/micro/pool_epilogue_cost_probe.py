import sys
sys.path.insert(0, "/root/repo/profiles/micro"); sys.path.insert(0, "/root/repo")
import conv_rate_probe as P
for cin, w, h in ((64, 640, 360), (128, 320, 180), (256, 160, 90), (512, 80, 45)):
    for pool in (False, True, False, True):
        print("pool" if pool else "plain", end=" ")
        P.run(cin, cin, w, (h,), pool=pool)
