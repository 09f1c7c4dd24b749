"""Time one fine launch (131072 rays x 128 samples) of a field kernel: python tools/time_field.py [bf16|fp32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ablate
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sizes = [(int(a.split("x")[0]), int(a.split("x")[1])) for a in sys.argv[2:]] or [(131072, 128)]
for N, S in sizes:
    for _ in range(2):
        r = ablate.time_one(prec, N=N, S=S)
        print(N, S, r, "ns/sample %.2f" % (r["ms"] * 1e6 / (N * S)))
