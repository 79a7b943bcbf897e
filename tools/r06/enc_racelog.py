"""Race detector for the two-stream protocols: the encoder does not depend on the decode stream, so the per-step checksum log
(LASR_DBG_ENCLOG, lasr_debug_enclog) of a pipelined run must equal a synchronous run's word for word.  Prints, per pipelined
run, the first (step, entry) whose checksums differ and the rows.
usage: enc_racelog.py <model> <beam> <streams> <chunks> <runs> [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("LASR_DBG_ENCLOG", "64")
import numpy as np
from libreasr_amd import synth
from libreasr_amd.engine import Engine
name, W, B, n_chunks, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dtype = sys.argv[6] if len(sys.argv) > 6 else "bf16"
DEPTH = int(os.environ.get("DEPTH", "6"))
cfg = synth.model_cfg(name); sd = synth.synth_state_dict(cfg, seed=0)
L, T = cfg["enc_layers"], int(cfg.get("n_buffer", 2))
pcm = np.stack([synth.synth_pcm(1, n_chunks * 1280, seed=1234 + s)[0] for s in range(B)])
DEV = os.environ.get("DEV") == "1"
if DEV:
    import torch
    pcmd = [torch.as_tensor(np.ascontiguousarray(pcm[:, k * 1280:(k + 1) * 1280])).cuda() for k in range(n_chunks)]
eng = Engine(sd, cfg, max_streams=B, dtype=dtype, beam=W)
slots = [eng.open() for _ in range(B)]
cap = 8192 if W > 1 else 64
def run(mode):
    for s in slots: eng.reset(s, 15)
    eng.debug_enclog()
    for k in range(n_chunks):
        if mode == "sync":
            eng.push(slots, pcm[:, k * 1280:(k + 1) * 1280])
            if eng.step(slots): eng.fetch_many(slots, cap)
            continue
        eng.push_submit(slots, pcmd[k] if DEV else pcm[:, k * 1280:(k + 1) * 1280])
        while eng.pending() >= DEPTH:
            if eng.wait(): eng.fetch_many(slots, cap)
    while eng.pending():
        if eng.wait(): eng.fetch_many(slots, cap)
    log = eng.debug_enclog()
    if os.environ.get("LASR_DBG_PENDLOG") == "1":
        return log, np.stack([eng.debug_read("pendlog", i) for i in range(len(log))])
    return log, None
def entry_name(e, T):
    if e < T: return f"x0[t={e}]"
    if e < T + L: return f"c[l={e - T}]"
    if e < T + 2 * L: return f"h[l={e - T - L}]"
    if e < 2 * T + 2 * L: return f"y[t={e - T - 2 * L}]"
    return "pend" if e == 2 * T + 2 * L else "pcm_ring"
ref, ref_pend = run("sync")
print("steps logged:", ref.shape[0], "entries:", 2 * T + 2 * L, flush=True)
again, _ = run("sync")
print("sync vs sync identical:", bool((again == ref).all()), flush=True)
bad = 0
for r in range(N):
    cur, cur_pend = run("pipe")
    n = min(len(ref), len(cur))
    d = (ref[:n, :2 * T + 2 * L + 2, :B] != cur[:n, :2 * T + 2 * L + 2, :B])
    if d.any():
        bad += 1
        st = int(np.nonzero(d.any(axis=(1, 2)))[0][0])
        ents = [int(e) for e in np.nonzero(d[st].any(axis=1))[0]]
        rows = {entry_name(e, T): [int(x) for x in np.nonzero(d[st, e])[0]][:12] for e in ents[:3] + ents[-2:]}
        total = int(d.any(axis=1).sum())
        if cur_pend is not None:
            a, b = ref_pend[st][:B].reshape(B, -1, 128), cur_pend[st][:B].reshape(B, -1, 128)
            for row in sorted(set(int(x) for x in np.nonzero((a != b).any(axis=(1, 2)))[0]))[:4]:
                fr = [int(x) for x in np.nonzero((a[row] != b[row]).any(axis=1))[0]]
                f0 = fr[0]
                mels = np.nonzero(a[row, f0] != b[row, f0])[0]
                print(f"   row {row}: frames that differ {fr}; frame {f0}: {len(mels)} mel bins differ, ({[int(m) for m in mels]}) max |diff| {np.abs(a[row, f0] - b[row, f0]).max():.4g}, "
                      f"sync {a[row, f0, mels[:4]]} pipe {b[row, f0, mels[:4]]}", flush=True)
                # where else do the wrong values occur?  (another row / frame of this or the previous steps: stale data)
                for m in mels[:10]:
                    hits = []
                    for st2 in range(max(0, st - 2), st + 1):
                        for nm, arr in (("sync", ref_pend), ("pipe", cur_pend)):
                            A = arr[st2][:B].reshape(B, -1, 128)
                            rr, ff, mm = np.nonzero(A == b[row, f0, m])
                            for x, y, z in list(zip(rr, ff, mm))[:3]:
                                if not (nm == "pipe" and st2 == st and x == row and y == f0):
                                    hits.append((nm, st2, int(x), int(y), int(z)))
                    print(f"      bin {int(m)}: pipe value {b[row, f0, m]:.6f} also at (run, step, row, frame, bin) {hits[:4]}", flush=True)
        print(f"run {r}: first difference at step {st} of {n}: {rows}; (step, row) pairs that differ anywhere: {total}", flush=True)
print("runs that differ:", bad, "of", N)
if os.environ.get("LASR_X_CANARY"):
    os.environ["LASR_X_CANARY_DUMP"] = "1"
    n_chunks = 8
    run("sync")
eng.close()
