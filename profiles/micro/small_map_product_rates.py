import sys
sys.path.insert(0, "/root/repo/profiles/micro"); sys.path.insert(0, "/root/repo")
import conv_rate_probe as P
print("product kernel (conv_s3_kernel, the launcher's choice), same layers:")
P.run(512, 512, 20, (12,)); P.run(256, 256, 40, (23,)); P.run(128, 128, 80, (45,)); P.run(64, 64, 160, (90,)); P.run(1024, 1024, 40, (22,)); P.run(512, 512, 80, (45,))
