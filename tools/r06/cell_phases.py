"""per-phase microseconds of the f32 encoder-cell launch (k_gemm<OpsF32, EpiLSTM<enc>>), IN THE JOB (pipelined
protocol, decode loop running on the second stream) and isolated (lasr_bench_cell), from the in-kernel stamps of LASR_DBG_TIMING
(lasr_gemm.hip.h: dbg[0] entry, [1] operands addressed + epilogue operands in flight, [2] K loop done, [3] partial tiles in LDS +
barrier, [4] epilogue done; [5] / [6] wall clock at entry / exit; [8..] per-wave end of the K loop).  The stamps are those of the
LAST cell launch before each dump; 12 dumps in the job.  (LASR_DBG_TIMING turns the hipGraph replays and the pump thread off: the
decode groups are launched from the API calls, as in round 3 -- the two streams still run side by side.)"""
import os, sys, json
os.environ["LASR_DBG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
from libreasr_amd import synth
from libreasr_amd.engine import Engine

cfg = synth.model_cfg("cfg2"); sd = synth.synth_state_dict(cfg); B = 64
eng = Engine(sd, cfg, max_streams=B)
N = 320
pcm = torch.as_tensor(np.stack([synth.synth_pcm(1, N * 1280, seed=1234 + s)[0] for s in range(B)]).reshape(B, N, 1280).transpose(1, 0, 2).copy()).cuda()
slots = [eng.open() for _ in range(B)]

def cell_phases():
    buf = np.zeros(5 * 4096 * 16, dtype=np.uint64)
    eng._chk(eng.lib.lasr_debug_timing(eng.ctx, buf.ctypes.data_as(C.c_void_p)))
    b = buf.reshape(5, 4096, 16).astype(np.float64)[0, :256]
    b = b[b[:, 0] > 0]
    if len(b) < 200:
        return None
    wall = (b[:, 6] - b[:, 5]) / 100.0                       # us per workgroup (100 MHz wall clock)
    cyc = b[:, 4] - b[:, 0]
    tpu = float(np.median(cyc / np.maximum(wall, 1e-3)))      # s_memtime ticks per us
    ph = np.stack([b[:, 1] - b[:, 0], b[:, 2] - b[:, 1], b[:, 3] - b[:, 2], b[:, 4] - b[:, 3]], 1) / tpu
    span = (b[:, 6].max() - b[:, 5].min()) / 100.0
    skew = (b[:, 5] - b[:, 5].min()) / 100.0
    tail = (b[:, 6].max() - b[:, 6]) / 100.0
    return {"span_us": span, "wg_wall_us": float(wall.mean()), "start_skew_mean_us": float(skew.mean()), "start_skew_max_us": float(skew.max()),
            "exit_tail_mean_us": float(tail.mean()),
            "setup_us": float(ph[:, 0].mean()), "k_loop_us": float(ph[:, 1].mean()), "k_loop_max_us": float(ph[:, 1].max()),
            "reduce_us": float(ph[:, 2].mean()), "epilogue_us": float(ph[:, 3].mean())}

def mean(ds):
    ds = [d for d in ds if d]
    return {k: round(float(np.mean([d[k] for d in ds])), 3) for k in ds[0]} if ds else None

depth, injob = 18, []
for k in range(N):
    eng.push_submit(slots, pcm[k], device_stable=True)
    while eng.pending() >= depth:
        eng.wait(); eng.fetch_many(slots, 64)
    if k >= 80 and k % 20 == 1:
        injob.append(cell_phases())
while eng.pending():
    eng.wait(); eng.fetch_many(slots, 64)
iso = []
for layer in (0, 1, 2, 3):
    us = eng.bench_cell(layer, 50)
    d = cell_phases()
    if d: d["bench_cell_us"] = us; iso.append(d)
out = {"in_job": mean(injob), "in_job_samples": len([d for d in injob if d]), "isolated": mean(iso),
       "note": "microseconds; span = max exit - min entry over the launch's 256 workgroups (what a kernel trace reports); phases are means over the workgroups"}
print(json.dumps(out, indent=1))
eng.close()
