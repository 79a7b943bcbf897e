"""Micro-benchmark of the dominant kernel (encoder LSTM cell) for A/B experiments and PMC passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libreasr_amd import synth
from libreasr_amd.engine import Engine
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = synth.model_cfg(name); sd = synth.synth_state_dict(cfg)
dtype = os.environ.get("LASR_DTYPE", "f32")
eng = Engine(sd, cfg, max_streams=B, dtype=dtype)
H = cfg["hidden"]
layers = [int(x) for x in os.environ.get("LASR_BENCH_LAYERS", "0,1").split(",")]
for layer in layers:
    I = cfg["feat"] if layer == 0 else H
    us = eng.bench_cell(layer, iters)
    fl = 2.0 * B * 4 * H * (I + H)
    eb = 4 if dtype == "f32" else 2
    print(f"dtype={dtype} layer {layer}: {us:.2f} us/launch  {fl/us/1e6:.1f} TFLOP/s  ({fl/us/1e6/157.3*100:.1f}% of f32 MFMA peak)  weights {eb*4*H*(I+H)/us/1e6:.2f} TB/s")
