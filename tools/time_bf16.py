import sys, os
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import ablate
for _ in range(3):
    print(ablate.time_one("bf16"))
