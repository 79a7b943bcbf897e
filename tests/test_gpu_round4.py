"""Round-4 GPU tests:

  * encoder cell tiling D (12 units x 64 rows per workgroup, `EpiLSTMe`, what configs[4]'s 1536-unit / 128-stream shape selects):
    against tiling C on the same weights -- bf16 bit-identical (same K split, same reduction order), f32 against the oracle --
    and streaming tokens equal
  * the pump thread (decode groups launched by a library thread) against LASR_PUMP=0 on the pipelined protocol: same tokens per
    model step
  * lasr_overlap_probe: the two engine streams run concurrently in a plain process (ratio ~1)"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from libreasr_amd import synth
from oracle import rnnt_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG12 = dict(feat=1280, embed=32, vocab=64, hidden=96, joint=64, enc_layers=3, pred_layers=2, pred_cell="NBRC",
             blank_bias=10.8, out_scale=8.0)


def make(cfg, **kw):
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    graft.build()
    sd = synth.synth_state_dict(cfg, seed=0)
    return Engine(sd, cfg, **kw), sd


def stream_tokens(eng, pcm, n_chunks):
    slots = [eng.open() for _ in range(pcm.shape[0])]
    got = [[] for _ in slots]
    try:
        for k in range(n_chunks):
            eng.push(slots, torch.as_tensor(np.ascontiguousarray(pcm[:, k * 1280:(k + 1) * 1280])).cuda())
            if eng.step(slots):
                for i, t in enumerate(eng.fetch_many(slots, 64)):
                    got[i] += t
    finally:
        for s in slots:
            eng.close_slot(s)
    return got


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_encoder_cell_tiling_d_equals_tiling_c(dtype, monkeypatch):
    B, n = 70, 20                                      # 70 rows: two 64-row m-groups, the second partly filled
    pcm = np.stack([synth.synth_pcm(1, n * 1280, seed=300 + s)[0] for s in range(B)])
    feats = np.stack([O.features_offline(p[:16000]) for p in pcm[:B]]).astype(np.float32)
    outs, toks = {}, {}
    for u12 in (0, 1):
        monkeypatch.setenv("LASR_ENC_U12", str(u12))
        eng, sd = make(CFG12, max_streams=128, dtype=dtype)
        try:
            out, h, c = eng.encoder(torch.as_tensor(feats).cuda(), return_state=True)
            outs[u12] = (out.cpu().numpy(), h.cpu().numpy(), c.cpu().numpy())
            toks[u12] = stream_tokens(eng, pcm, n)
        finally:
            eng.close()
    if dtype == "bf16":                                # same K split over 8 waves, same reduction order: bit-identical
        for a, b in zip(outs[0], outs[1]):
            assert np.array_equal(a, b)
        assert toks[0] == toks[1]
    else:
        m = O.OracleTransducer(sd, CFG12)
        ref, st = m.encoder(feats[:8])
        assert np.abs(outs[1][0][:8] - ref).max() < 5e-4
        assert np.abs(outs[1][0] - outs[0][0]).max() < 5e-4
        n_diff = sum(a != b for a, b in zip(toks[0], toks[1]))
        assert n_diff == 0, f"{n_diff} of {B} streams differ between the tilings"
    assert sum(len(t) for t in toks[1]) > 50


def test_pump_thread_equals_api_launched_groups():
    """The same pipelined run (4 streams out of phase, depth 4, push_submit) in two processes: LASR_PUMP=1 (default) and 0."""
    code = r'''
import json, sys, numpy as np, torch
sys.path.insert(0, %r)
from libreasr_amd import synth
from libreasr_amd.engine import Engine
cfg = synth.model_cfg("tiny"); sd = synth.synth_state_dict(cfg, seed=0)
eng = Engine(sd, cfg, max_streams=16)
B, n = 4, 40
pcm = np.stack([synth.synth_pcm(1, n * 1280, seed=900 + s)[0] for s in range(B)])
slots = [eng.open() for _ in range(B)]
steps = [[] for _ in range(B)]
def collect():
    if eng.wait():
        for i, t in enumerate(eng.fetch_many(slots, 64)):
            steps[i].append(t)
for k in range(n + B):
    act = [s for s in range(B) if 0 <= k - s < n]
    if not act: continue
    eng.push_submit([slots[s] for s in act], torch.as_tensor(np.stack([pcm[s, (k - s) * 1280:(k - s + 1) * 1280] for s in act])).cuda())
    while eng.pending() >= 4: collect()
while eng.pending(): collect()
print("RESULT" + json.dumps(steps))
''' % ROOT
    res = {}
    for pump in ("1", "0"):
        env = dict(os.environ, LASR_PUMP=pump)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert line, r.stdout[-2000:] + r.stderr[-2000:]
        res[pump] = line[0]
    assert res["1"] == res["0"]
    assert res["1"].count(",") > 50


def test_overlap_probe_sees_two_concurrent_streams():
    eng, _ = make(synth.model_cfg("tiny"), max_streams=16)
    try:
        r = eng.overlap_probe(5000)
        assert 0.9 < r < 1.4, r
    finally:
        eng.close()


def test_peek_and_reset_of_a_slot_whose_submitted_steps_are_decoded():
    """lasr_peek_slot / lasr_peek_many: the tokens seen early are the tokens lasr_step_wait hands out later; a slot may be reset
    (model state) while its decoded steps are still uncollected, and the stream then continues exactly as the oracle does after
    reset() between those two model steps (the servicer's reset rule applied without draining the pipeline); the neighbour
    stream is untouched."""
    import time
    from libreasr_amd._native import LasrError
    cfg = synth.model_cfg("tiny")
    eng, sd = make(cfg, max_streams=16)
    try:
        m = O.OracleTransducer(sd, cfg)
        n = 44
        pcm = np.stack([synth.synth_pcm(1, n * 1280, seed=4100 + s)[0] for s in range(2)])
        RESET_AFTER = 6                                    # model steps of stream 0 before its reset
        want = []
        for s in range(2):
            fe, dec = O.StreamFrontend(), m.stream_decoder()
            steps = []
            for k in range(n):
                o = fe.push(pcm[s, k * 1280:(k + 1) * 1280])
                if o is not None:
                    steps.append(dec.step(o))
                    if s == 0 and len(steps) == RESET_AFTER:
                        dec.reset()
            want.append(steps)
        slots = [eng.open(), eng.open()]
        got = [[], []]
        peeked = {}
        n_sub = 0

        def collect():
            if eng.wait():
                for i, t in enumerate(eng.fetch_many(slots, 64)):
                    got[i].append(t)

        for k in range(n):
            before = eng.pending()
            eng.push_submit(slots, torch.as_tensor(np.ascontiguousarray(pcm[:, k * 1280:(k + 1) * 1280])).cuda())
            if eng.pending() > before:
                n_sub += 1
                if n_sub == RESET_AFTER:
                    # steps in flight: wait (without collecting) until the decode loop has finished them for slot 0, reset it
                    t0 = time.time()
                    while True:
                        steps, n_in = eng.peek(slots[0])
                        if len(steps) == n_in:
                            break
                        assert time.time() - t0 < 10
                    assert n_in >= 1
                    many, nd, nf = eng.peek_many(slots, [0, 0])
                    assert many[0] == steps and int(nf[0]) == n_in
                    peeked[0] = steps
                    with pytest.raises(LasrError):          # without the flag a slot with uncollected steps is refused, decoded or not
                        eng.reset(slots[0], 7)
                    with pytest.raises(LasrError):          # front-end state cannot be reset under steps in flight
                        eng.reset(slots[0], 8, if_decoded=True)
                    eng.reset(slots[0], 7, if_decoded=True)
            while eng.pending() >= 3:
                collect()
        while eng.pending():
            collect()
        assert got[0] == want[0] and got[1] == want[1]
        n_col = len(peeked[0])
        assert peeked[0] == want[0][RESET_AFTER - n_col:RESET_AFTER]      # what peek showed == what wait handed out later
        assert sum(len(t) for t in got[0]) > 10
    finally:
        eng.close()


@pytest.mark.parametrize("int8", [False, True])
def test_lm_fusion_with_lookahead_equals_one_frame_per_iteration(int8, monkeypatch):
    """An attached LM no longer turns the decode lookahead off (k_select re-picks the token at the first non-blank frame of its
    window with that frame's fused scores; blank frames change neither predictor nor LM state, models.py:475-520): same tokens
    as LASR_LM_LOOKAHEAD=0 on 12 s of 8 streams, fp32 LM and int8-served LM (whose h images are now quantised
    by the cell kernel)."""
    cfg = synth.model_cfg("tiny_soft")
    lsd = synth.synth_lm_state_dict("tiny_lm")
    pcm = synth.synth_pcm(8, 16000 * 12, seed=4321)
    out = []
    for la in ("0", "1"):
        monkeypatch.setenv("LASR_LM_LOOKAHEAD", la)
        eng, _ = make(cfg, max_streams=8, dtype="f32")
        eng.attach_lm(lsd, int8=int8)
        out.append(stream_tokens(eng, pcm, pcm.shape[1] // 1280))
        del eng
    assert out[0] == out[1]
    assert sum(len(s) for s in out[0]) > 50


def test_lm_register_slots_and_lm_stream_are_bit_identical_switches():
    """k_lm_post / k_beam_fuse with 8 register slots per thread (V <= 2048) against LASR_KEEP16=1 (round 3's kernels), and the LM
    branch on its own stream (LASR_LM_SIDE=1) against the LM step in line: the same tokens AND the same -log p bits, greedy (offline + pipelined) and
    beam 4 with the LM inside the beam."""
    code = r'''
import json, sys, numpy as np, torch
sys.path.insert(0, %r)
from libreasr_amd import synth
from libreasr_amd.engine import Engine
cfg = synth.model_cfg("tiny_soft"); sd = synth.synth_state_dict(cfg, seed=0); lsd = synth.synth_lm_state_dict("tiny_lm")
out = {}
for beam in (1, 4):
    eng = Engine(sd, cfg, max_streams=8, beam=beam)
    eng.attach_lm(lsd)
    pcm = synth.synth_pcm(4, 16000 * 4, seed=55 + beam)
    slots = [eng.open() for _ in range(4)]
    eng.transcribe_pcm(slots, [pcm[i] for i in range(4)])
    off = [eng.fetch(s) for s in slots]
    out["off%%d" %% beam] = [[t, float(lp).hex()] for t, lp, _ in off]
    for s in slots: eng.reset(s, 15)
    got = [[] for _ in slots]
    def collect():
        if eng.wait():
            for i, t in enumerate(eng.fetch_many(slots, 256)): got[i].append(t)
    for k in range(pcm.shape[1] // 1280):
        eng.push_submit(slots, torch.as_tensor(np.ascontiguousarray(pcm[:, k * 1280:(k + 1) * 1280])).cuda())
        while eng.pending() >= 3: collect()
    while eng.pending(): collect()
    out["pipe%%d" %% beam] = got
    eng.close()
print("RESULT" + json.dumps(out))
''' % ROOT
    res = {}
    for tag, env in (("default", {}), ("keep16", {"LASR_KEEP16": "1"}), ("branch", {"LASR_LM_SIDE": "1"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert line, r.stdout[-2000:] + r.stderr[-2000:]
        res[tag] = line[0]
    assert res["default"] == res["keep16"]
    assert res["default"] == res["branch"]
    assert res["default"].count(",") > 100


def test_push_submit_rows_and_reset_many_equal_the_one_array_one_slot_forms():
    """lasr_push_submit_rows (the chunk of slots[i] at its own host address) against lasr_push_submit on one array, and
    lasr_stream_reset_many against a loop of lasr_stream_reset: same tokens per model step, resets in the middle included."""
    cfg = synth.model_cfg("tiny")
    B, n = 6, 48
    pcm = synth.synth_pcm(B, n * 1280, seed=777)
    res = []
    for form in ("array", "rows"):
        eng, _ = make(cfg, max_streams=16)
        slots = [eng.open() for _ in range(B)]
        steps = [[] for _ in range(B)]

        def collect():
            if eng.wait():
                for i, t in enumerate(eng.fetch_many(slots, 64)):
                    steps[i].append(t)
        scattered = [np.ascontiguousarray(pcm[i].reshape(n, 1280)) for i in range(B)]      # one array per stream, as a server has them
        for k in range(n):
            if form == "array":
                eng.push_submit(slots, np.stack([scattered[i][k] for i in range(B)]))
            else:
                addrs = np.array([scattered[i].ctypes.data + k * 1280 * 4 for i in range(B)], np.uint64)
                eng.push_submit_rows(slots, addrs)
            while eng.pending() >= 3:
                collect()
            if k in (15, 31):                 # everything collected, then streams 1, 2, 4 start over (encoder + predictor state)
                while eng.pending():
                    collect()
                if form == "array":
                    for i in (1, 2, 4):
                        eng.reset(slots[i], 7)
                else:
                    eng.reset_many([slots[i] for i in (1, 2, 4)], 7)
        while eng.pending():
            collect()
        res.append([list(x) for x in steps])
        with pytest.raises(Exception):
            eng.reset_many([slots[0], slots[0]], 7)
        if form == "rows":                    # argument errors: nothing is pushed, the engine stays usable
            from libreasr_amd._native import LasrError
            good = np.array([scattered[i].ctypes.data for i in range(B)], np.uint64)
            bad = good.copy(); bad[2] = 0
            with pytest.raises(LasrError):
                eng.push_submit_rows(slots, bad)                                  # a null row
            dev = torch.zeros(B, 1280, device="cuda")
            with pytest.raises(LasrError):
                eng.push_submit_rows(slots, np.array([dev.data_ptr() + i * 5120 for i in range(B)], np.uint64))   # device memory
            with pytest.raises(LasrError):
                eng.push_submit_rows([slots[0], 15], good[:2])                    # a slot that is not open
            eng.push_submit_rows([], np.zeros(0, np.uint64))                      # empty batch: a no-op
            eng.push_submit_rows(slots, good)                                     # and the engine still takes a push
            while eng.pending():
                collect()
        eng.close()
    assert res[0] == res[1]
    assert sum(len(t) for s in res[0] for t in s) > 50


def test_stored_bos_state_equals_the_predictor_pass_on_bos():
    """A reset stores the predictor state after the BOS step and its joint half, captured once at lasr_create, instead of running
    the predictor on BOS (LASR_BOS_CACHE=0): same tokens with resets in the middle, NBRC and LSTM predictors, f32 and bf16."""
    code = r'''
import json, sys, numpy as np, torch
sys.path.insert(0, %r)
from libreasr_amd import synth
from libreasr_amd.engine import Engine
out = {}
for name in ("tiny", "tiny_lstm"):
    for dtype in ("f32", "bf16"):
        cfg = synth.model_cfg(name); sd = synth.synth_state_dict(cfg, seed=0)
        eng = Engine(sd, cfg, max_streams=8, dtype=dtype)
        pcm = synth.synth_pcm(4, 16000 * 4, seed=31)
        slots = [eng.open() for _ in range(4)]
        got = [[] for _ in slots]
        for k in range(pcm.shape[1] // 1280):
            eng.push(slots, torch.as_tensor(np.ascontiguousarray(pcm[:, k * 1280:(k + 1) * 1280])).cuda())
            if eng.step(slots):
                for i, t in enumerate(eng.fetch_many(slots, 64)): got[i].append(t)
            if k %% 13 == 12:
                eng.reset(slots[k %% 4], 7)
        eng.transcribe_pcm(slots, [pcm[i] for i in range(4)])
        out[name + dtype] = [got, [[t, float(lp).hex()] for t, lp, _ in (eng.fetch(s) for s in slots)]]
        eng.close()
print("RESULT" + json.dumps(out))
''' % ROOT
    res = {}
    for tag, env in (("stored", {}), ("pass", {"LASR_BOS_CACHE": "0"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert line, r.stdout[-2000:] + r.stderr[-2000:]
        res[tag] = line[0]
    assert res["stored"] == res["pass"]
    assert res["stored"].count(",") > 100


def test_lm_pair_launches_equal_the_lm_step_in_line():
    """configs[1] with the reference's 4 x 768 LM on the pipelined protocol: LM layer l and stage l of the predictor / joint chain
    in ONE launch (k_gemm2, the default for this shape) against LASR_LM_PAIR=0 -- same tokens per model step, f32 and bf16 --
    and against the synchronous protocol (which never pairs and is pinned to the reference's goldens in tests/test_gpu_lm.py)."""
    code = r'''
import json, sys, numpy as np, torch
sys.path.insert(0, %r)
from libreasr_amd import synth
from libreasr_amd.engine import Engine
cfg = synth.model_cfg("cfg2"); sd = synth.synth_state_dict(cfg, seed=0); lsd = synth.synth_lm_state_dict("lm768")
out = {}
for dtype in ("f32", "bf16"):
    eng = Engine(sd, cfg, max_streams=8, dtype=dtype)
    eng.attach_lm(lsd)
    pcm = synth.synth_pcm(5, 16000 * 3, seed=91)
    slots = [eng.open() for _ in range(5)]
    got = [[] for _ in slots]
    def collect():
        if eng.wait():
            for i, t in enumerate(eng.fetch_many(slots, 256)): got[i].append(t)
    for k in range(pcm.shape[1] // 1280):
        eng.push_submit(slots, torch.as_tensor(np.ascontiguousarray(pcm[:, k * 1280:(k + 1) * 1280])).cuda())
        while eng.pending() >= 4: collect()
    while eng.pending(): collect()
    for s in slots: eng.reset(s, 15)
    sync = [[] for _ in slots]
    for k in range(pcm.shape[1] // 1280):
        eng.push(slots, torch.as_tensor(np.ascontiguousarray(pcm[:, k * 1280:(k + 1) * 1280])).cuda())
        if eng.step(slots):
            for i, t in enumerate(eng.fetch_many(slots, 256)): sync[i].append(t)
    assert got == sync, dtype
    out[dtype] = got
    eng.close()
print("RESULT" + json.dumps(out))
''' % ROOT
    res = {}
    for tag, env in (("pair", {}), ("inline", {"LASR_LM_PAIR": "0"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert line, r.stdout[-2000:] + r.stderr[-2000:]
        res[tag] = line[0]
    assert res["pair"] == res["inline"]
    assert res["pair"].count(",") > 100


def test_bench_neighbour_runs_and_reports():
    """lasr_bench_neighbour (experiment hook): a neighbour of every kind runs on its own stream for a few milliseconds and reports a
    positive rate; a second start before the first was collected is refused; the engine works afterwards."""
    from libreasr_amd._native import LasrError
    eng, _ = make(synth.model_cfg("tiny"), max_streams=16)
    try:
        for kind in (1, 2, 3, 4):
            eng.bench_neighbour(kind, 8, 3)
            with pytest.raises(LasrError):
                eng.bench_neighbour(kind, 8, 3)
            assert eng.bench_neighbour(0) > 0.0
        assert eng.bench_neighbour(0) == 0.0          # nothing running
        pcm = synth.synth_pcm(2, 16 * 1280, seed=5)
        assert sum(len(t) for t in stream_tokens(eng, pcm, 16)) >= 0
    finally:
        eng.close()
