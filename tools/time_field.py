"""Time one fine launch (131072 rays x 128 samples) of a field kernel: python tools/time_field.py [bf16|fp32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ablate
for _ in range(3):
    print(ablate.time_one(sys.argv[1] if len(sys.argv) > 1 else "bf16"))
