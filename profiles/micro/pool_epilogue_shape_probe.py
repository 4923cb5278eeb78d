import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "profiles", "micro"))
import conv_rate_probe as P
for rep in range(2):
    for (cin, cout, w, h) in ((512, 512, 80, 45), (256, 256, 160, 90), (128, 128, 320, 180), (64, 64, 640, 360)):
        P.run(cin, cout, w, (h,), pool=False)
        P.run(cin, cout, w, (h,), pool=True)
