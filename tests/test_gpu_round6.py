"""Round 6 (VERDICT r5 / ADVICE r5).

* native front: stream ids carry a generation -- a reader that still holds the id of a closed stream cannot write into the
  slot's next stream or end it; close releases a producer blocked on a full ring;
* native front: the reset rule judges a step by its TEXT (api-server.py:124 `y_one != ""`), through the tokenizer's set of
  ids that decode to "" (VERDICT r5 item 6), with one RESET flag per judged step even when reset_steps <= depth;
* real speech: the demo utterance of the reference (demo/3729-6852-0035.flac, configs[0]) through the GPU path against goldens
  the reference's own pipeline produced;
* bf16 operands against the fp32 reference: token error rate on the long goldens, recorded."""
import json
import os
import threading
import time

import numpy as np
import pytest
import torch

from libreasr_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def make_engine(name, max_streams=16, dtype="f32", beam=1):
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    graft.build()
    cfg = synth.model_cfg(name)
    sd = synth.synth_state_dict(cfg, seed=0)
    return Engine(sd, cfg, max_streams=max_streams, dtype=dtype, beam=beam), sd, cfg


def sync_stream(eng, chunks, reset_steps=0, is_empty=None):
    """One stream alone through the synchronous protocol, the servicer's rule applied between two steps (api-server.py:131-134)."""
    slot = eng.open()
    y, per_step, steps, nres = [], [], 0, 0
    for c in chunks:
        eng.push([slot], c[None])
        if not eng.step([slot]):
            continue
        ys = eng.fetch(slot)[0]
        steps += 1
        y += ys
        per_step.append(ys)
        empty = (not ys) if is_empty is None else is_empty(ys)
        if reset_steps and empty and steps >= reset_steps:
            eng.reset(slot, 1 | 2 | 4)
            steps, nres = 0, nres + 1
    eng.close_slot(slot)
    return y, per_step, nres


def test_native_front_stale_stream_id_names_nothing(golden_dir):
    from libreasr_amd import _native as N
    from libreasr_amd.front import RES_EOF, NativeFront
    eng, sd, cfg = make_engine("tiny")
    try:
        front = NativeFront(eng, depth=4, reset_steps=0)
        old = front.open()
        out = {}

        def producer():
            chunk = np.full(1280, 0.25, np.float32)     # (loud garbage: it would change the next stream's tokens)
            try:
                for _ in range(400):
                    front.push(old, chunk)
                out["producer"] = "returned"
            except N.LasrError as e:
                out["producer"] = e.code

        pause = front.paused()
        pause.__enter__()                              # nothing drains the ring: the producer blocks at the 65th chunk
        tp = threading.Thread(target=producer)
        tp.start()
        time.sleep(0.2)
        assert tp.is_alive()
        tc = threading.Thread(target=lambda: front.close(old))       # (needs the engine: completes after resume)
        tc.start()
        tp.join(timeout=5)                             # close released the blocked producer before it took the engine
        assert not tp.is_alive() and out["producer"] == N.LASR_ESTATE
        pause.__exit__(None, None, None)
        tc.join(timeout=10)
        assert not tc.is_alive()
        new = front.open()                             # the same slot, a new generation
        assert (new & 0xffff) == (old & 0xffff) and new != old
        for call in (lambda: front.push(old, np.zeros(1280, np.float32)), lambda: front.eof(old), lambda: front.next(old, 10),
                     lambda: front.close(old)):
            with pytest.raises(N.LasrError) as e:
                call()
            assert e.value.code == N.LASR_ESTATE
        # the new stream is untouched: the reference's golden tokens, and no end-of-stream before ITS eof
        g = np.load(os.path.join(golden_dir, "model_tiny.npz"))
        pcm = synth.synth_pcm(3, 16000 * 3, seed=1234)[0]
        chunks = synth.stream_chunks(pcm, 1280, lead=1, tail=10)
        front.push(new, np.concatenate(chunks[:20]))
        got = []
        deadline = time.time() + 20
        while len(got) < 4 and time.time() < deadline:                # some steps arrive, none of them the end of the stream
            r = front.next(new, 200)
            if r is None:
                continue
            assert not (r[1] & RES_EOF)
            got.append(r[0])
        front.push(new, np.concatenate(chunks[20:]))
        front.eof(new)
        while True:
            toks, flags = front.next(new)
            if flags & RES_EOF:
                break
            got.append(toks)
        assert [t for s in got for t in s] == list(g["st_tokens_0"])
        front.close(new)
        front.destroy()
    finally:
        eng.close()


@pytest.mark.parametrize("reset_steps,depth", [(25, 8), (3, 8)])
def test_native_front_reset_rule_judges_text(reset_steps, depth):
    """A tokenizer in which some ids decode to "": past the threshold a step that emitted only such ids resets the stream, as
    the reference's `y_one != ""` does.  Expected = the synchronous protocol with the text rule applied by this test; the run
    with the count rule (no empty ids) must differ, or the case proves nothing.  reset_steps = 3 <= depth: several steps of a
    stream are judged early before the first is collected -- every reset is reported on the step that caused it."""
    from libreasr_amd.front import RES_EOF, RES_RESET, NativeFront
    eng, sd, cfg = make_engine("tiny")
    try:
        specs = [(700 + i, sp) for i, sp in enumerate([6.0, 7.5, [("speech", 2.0), ("silence", 5.0), ("speech", 2.0)], 5.0,
                                                      [("silence", 1.0), ("speech", 6.0)], 8.0])]
        pcm = [synth.servicer_pcm(s_, sp) for s_, sp in specs]
        chunks = [synth.stream_chunks(p, 1280, lead=1, tail=10) for p in pcm]
        plain = [sync_stream(eng, ch) for ch in chunks]
        freq = {}
        for y, _, _ in plain:
            for t in y:
                freq[t] = freq.get(t, 0) + 1
        empty_ids = sorted(sorted(freq, key=lambda t: -freq[t])[:max(3, len(freq) // 3)])     # the most frequent third "decode to ''"
        is_empty = lambda ys: all(t in empty_ids for t in ys)
        want = [sync_stream(eng, ch, reset_steps, is_empty) for ch in chunks]
        count_rule = [sync_stream(eng, ch, reset_steps) for ch in chunks]
        assert [w[0] for w in want] != [w[0] for w in count_rule] or [w[2] for w in want] != [w[2] for w in count_rule]
        assert sum(w[2] for w in want) > 2
        assert any(ys and is_empty(ys) for w in want for ys in w[1])       # a non-empty token list with empty text occurs
        front = NativeFront(eng, depth=depth, reset_steps=reset_steps, empty_tokens=empty_ids)
        try:
            B = len(chunks)
            got, flags_seen = [[] for _ in range(B)], [[] for _ in range(B)]

            def run(i):
                sid = front.open()
                front.push(sid, np.concatenate(chunks[i]))
                front.eof(sid)
                while True:
                    toks, flags = front.next(sid)
                    if flags & RES_EOF:
                        break
                    got[i].append(toks)
                    flags_seen[i].append(bool(flags & RES_RESET))
                front.close(sid)

            ths = [threading.Thread(target=run, args=(i,)) for i in range(B)]
            [t.start() for t in ths]
            [t.join(timeout=120) for t in ths]
            assert not any(t.is_alive() for t in ths)
            for i in range(B):
                assert got[i] == want[i][1], f"stream {i} {specs[i]}"
                # the RESET flag sits on exactly the steps after which the rule fired
                steps, exp = 0, []
                for ys in want[i][1]:
                    steps += 1
                    fire = is_empty(ys) and steps >= reset_steps
                    exp.append(fire)
                    if fire:
                        steps = 0
                assert flags_seen[i] == exp, f"stream {i}: reset flags"
            assert front.stats()["resets"] == sum(w[2] for w in want)
        finally:
            front.destroy()
    finally:
        eng.close()


# ------------------------------------------------------------------------------- real speech (configs[0]'s utterance)
def speech(golden_dir):
    g = np.load(os.path.join(golden_dir, "speech_demo.npz"))
    return g, (g["pcm_i16"].astype(np.float32) / 32768.0)


def test_real_speech_logmel_and_features_all_frames(golden_dir):
    """The reference's demo utterance (20.65 s of speech, incl. its low-energy stretches): k_logmel on all 2 066 frames against the
    reference's TransformTime, the stacked features against the reference's pipeline; the same bounds as on synthetic PCM."""
    from oracle import rnnt_oracle as O
    g, pcm = speech(golden_dir)
    eng, _, _ = make_engine("tiny")
    try:
        lm = eng.logmel(dev(pcm[None]))
        out = lm[0].cpu().numpy()
        ref = g["logmel"]
        assert out.shape == ref.shape == (2066, 128)
        err = np.abs(out.astype(np.float64) - ref.astype(np.float64))
        assert float(err.max()) < 3e-4, f"log-mel vs the reference: max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"
        quiet = ref < np.percentile(ref, 5)                    # the quietest 5 % of the bins hold to the same bound
        assert quiet.sum() > 10000 and float(err[quiet].max()) < 3e-4
        st = eng.stack(lm)[0].cpu().numpy()
        assert st.shape == (int(g["n_frames"]), 1280) == (258, 1280)
        assert np.array_equal(st, O.stack_downsample(out))     # pure data movement
        assert float(np.abs(st[:4] - g["feats_first"]).max()) < 3e-4 and float(np.abs(st[-2:] - g["feats_last"]).max()) < 3e-4
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["cfg2", "ref6"])
def test_real_speech_tokens_equal_the_reference_on_every_protocol(name, golden_dir):
    g, pcm = speech(golden_dir)
    eng, _, _ = make_engine(name)
    try:
        slot = eng.open()
        eng.transcribe_pcm([slot], [dev(pcm)])
        toks, neg_logp, align = eng.fetch(slot, cap=4096)
        assert toks == list(g[f"{name}_off_tokens"])
        assert abs(neg_logp - float(g[f"{name}_off_neglogp"])) < 5e-2 and abs(align - float(g[f"{name}_off_align"])) < 1e-9
        chunks = synth.stream_chunks(pcm, 1280, lead=1, tail=10)
        want, want_counts = list(g[f"{name}_st_tokens"]), list(g[f"{name}_st_counts"])
        # synchronous protocol
        eng.reset(slot, 15)
        got, counts = [], []
        for c in chunks:
            eng.push([slot], dev(c[None]))
            if eng.step([slot]):
                ys = eng.fetch(slot)[0]
                got += ys
                counts.append(len(ys))
        assert got == want and counts == want_counts
        # pipelined protocol, 8 steps in flight
        eng.reset(slot, 15)
        got, counts = [], []

        def collect():
            if eng.wait():
                ys = eng.fetch_many([slot], 64)[0]
                got.extend(ys)
                counts.append(len(ys))

        for c in chunks:
            eng.push_submit([slot], c[None])
            while eng.pending() >= 8:
                collect()
        while eng.pending():
            collect()
        assert got == want and counts == want_counts
        eng.close_slot(slot)
    finally:
        eng.close()


@pytest.mark.parametrize("front", ["python", "native"])
def test_real_speech_through_the_grpc_servers(front, golden_dir):
    """The reference's own ASRServicer (api-server.py:82-135, 64-80) on the demo utterance, cfg2 weights: message for message
    through libreasr_amd.server on both fronts (api-client.py:32-59: one zero frame, 80 ms frames, ten zero frames), beside three
    synthetic streams that keep the batch busy."""
    import grpc
    import __graft_entry__ as graft
    graft.build()
    from libreasr_amd import server as srv
    from libreasr_amd.interfaces import libreasr_pb2 as ap
    from libreasr_amd.interfaces import libreasr_pb2_grpc as apg
    g, pcm = speech(golden_dir)
    n_msgs = int(g["cfg2_n_msgs"])
    want = [str(v) for v in g["cfg2_msgs"][:n_msgs]]
    assert len(g["cfg2_resets"]) >= 3                           # the golden run does contain resets
    others = [synth.servicer_pcm(seed, spec) for seed, spec in synth.SERVICER_STREAMS[:3]]
    server, sched, port = srv.serve("en", port="127.0.0.1:0", block=False, config_path="/nonexistent.yaml",
                                    synthetic="cfg2", max_streams=16, front=front)
    try:
        got = {}

        def client(i, p):
            with grpc.insecure_channel(f"127.0.0.1:{port}") as ch:
                stub = apg.ASRStub(ch)
                reqs = (ap.Audio(data=c.tobytes(), sr=16000) for c in synth.stream_chunks(p, 1280, lead=1, tail=10))
                got[i] = [t.data for t in stub.TranscribeStream(reqs)]

        ths = [threading.Thread(target=client, args=(i, p)) for i, p in enumerate([pcm] + others)]
        [t.start() for t in ths]
        [t.join(timeout=300) for t in ths]
        assert not any(t.is_alive() for t in ths)
        assert got[0] == want
        with grpc.insecure_channel(f"127.0.0.1:{port}") as ch:
            assert apg.ASRStub(ch).Transcribe(ap.Audio(data=pcm.tobytes(), sr=16000)).data == str(g["cfg2_unary"])
    finally:
        server.stop(0)
        sched.shutdown()


# ------------------------------------------------------------------------------- bf16 operands against the fp32 REFERENCE
@pytest.mark.parametrize("name,n_streams", [("cfg2", 3), ("ref6", 1), ("cfg5", 1)])
def test_bf16_token_error_rate_against_the_fp32_reference_long_goldens(name, n_streams, golden_dir):
    """What bf16 operands cost in tokens: the reference's own fp32 decode of 20.65 s (cfg2) / 10 s utterances
    (tests/golden/model_*_long.npz) against the engine with bf16 operands, greedy, offline and streaming -- token error rate =
    edit distance / reference length, recorded in gpurun_out/parity_counts.json (committed under profiles/).  "Token-for-token"
    is the fp32 contract and only fp32 meets it; the bf16 number is stated, not bounded (see the assertion's comment).  Beam 4
    (cfg2) is recorded against the same greedy reference for information (a wider search is a different decode)."""
    from oracle import parity as PR
    record_parity = PR.record
    g = np.load(os.path.join(golden_dir, f"model_{name}_long.npz"))
    pcm = synth.synth_pcm(n_streams, int(g["n_samples"]), seed=1234)
    eng, _, _ = make_engine(name, dtype="bf16")
    rec = {}
    try:
        slots = [eng.open() for _ in range(n_streams)]
        eng.transcribe_pcm(slots, [dev(p) for p in pcm])
        off = [eng.fetch(s, cap=4096)[0] for s in slots]
        for s in slots:
            eng.reset(s, 15)
        chunks = [synth.stream_chunks(p, 1280, lead=1, tail=10) for p in pcm]
        st = [[] for _ in slots]
        for k in range(len(chunks[0])):
            eng.push(slots, dev(np.stack([c[k] for c in chunks])))
            if eng.step(slots):
                for i, t in enumerate(eng.fetch_many(slots, 64)):
                    st[i] += t
        for kind, got in (("offline", off), ("streaming", st)):
            key = "off_tokens_" if kind == "offline" else "st_tokens_"
            ref = [list(g[f"{key}{s}"]) for s in range(n_streams)]
            d = sum(PR.edit_distance(got[s], ref[s]) for s in range(n_streams))
            n = sum(len(r) for r in ref)
            first = [next((q for q in range(min(len(a), len(b))) if a[q] != b[q]), min(len(a), len(b))) for a, b in zip(got, ref)]
            rec[kind] = dict(ter=round(d / max(1, n), 5), edits=int(d), reference_tokens=int(n), streams_identical=int(sum(a == b for a, b in zip(got, ref))),
                             first_difference_at_token=[int(f) for f in first])
            print(f"{name} bf16 greedy {kind} vs the fp32 reference: TER {d}/{n} = {d / max(1, n):.4f}; identical streams "
                  f"{rec[kind]['streams_identical']}/{n_streams}; first differences at token {first}")
            # A RECORD, not a bound: with these synthetic (random) weights the decode is chaotic -- the first decision a bf16 rounding
            # flips changes the predictor state, and the rest of the utterance follows another path (round 6: TER 0.4-0.6 on 20 s,
            # first differences at tokens 3..85).  What is asserted is that the paths START together.
            assert max(first) >= 3 or n < 8, rec[kind]
    finally:
        eng.close()
    if name == "cfg2":
        engb, _, _ = make_engine(name, dtype="bf16", beam=4)
        try:
            slots = [engb.open() for _ in range(n_streams)]
            engb.transcribe_pcm(slots, [dev(p) for p in pcm])
            offb = [engb.fetch(s, cap=4096)[0] for s in slots]
            ref = [list(g[f"off_tokens_{s}"]) for s in range(n_streams)]
            d = sum(PR.edit_distance(offb[s], ref[s]) for s in range(n_streams))
            rec["beam4_offline_vs_fp32_greedy"] = dict(distance=int(d), reference_tokens=int(sum(len(r) for r in ref)),
                                                      note="a wider search, not an error rate")
        finally:
            engb.close()
    record_parity(f"{name}_bf16_vs_fp32_reference_long", **rec)


@pytest.mark.parametrize("name,n_streams", [("cfg2", 3), ("ref6", 1)])
def test_bf16_decision_agreement_under_teacher_forcing(name, n_streams, golden_dir):
    """The free-running token error rate above is dominated by what happens AFTER the first flipped decision (another predictor
    state, another path).  The numeric cost of bf16 operands per decision: every joint evaluation on the fp32 reference's own path
    (its tokens: tests/golden/model_*_long.npz, 20.65 s / 10 s utterances) repeated with bf16 operands from the SAME history --
    encoder output of the bf16 engine, predictor state of the bf16 engine after the reference's token prefix -- and the argmax
    compared with the fp32 decision.  Recorded (gpurun_out/parity_counts.json); asserted: the fp32 engine re-walks the reference's
    path exactly, bf16 agrees on >= 97 % of the decisions and every disagreement sits at an fp32 margin below 0.25 (logit scale ~10)."""
    from oracle import parity as PR
    g = np.load(os.path.join(golden_dir, f"model_{name}_long.npz"))
    pcm = synth.synth_pcm(n_streams, int(g["n_samples"]), seed=1234)
    e32, _, cfg = make_engine(name, dtype="f32")
    ebf, _, _ = make_engine(name, dtype="bf16")
    blank, bos, cap = int(cfg.get("blank", 0)), int(cfg.get("bos", 2)), 3          # decode_greedy: max_iters = 3 (models.py:369)
    n_dec = n_agree = 0
    margins, worst = [], 0.0
    try:
        for s in range(n_streams):
            y = [int(t) for t in g[f"off_tokens_{s}"]]
            feats = e32.stack(e32.logmel(dev(pcm[s][None])))                      # the front-end is f32 in both engines
            enc32, encbf = e32.encoder(feats)[0], ebf.encoder(feats)[0]
            hp32 = [e32.predictor([[bos] + y[:u]])[0] for u in range(len(y) + 1)]
            hpbf = [ebf.predictor([[bos] + y[:u]])[0] for u in range(len(y) + 1)]
            u = 0
            for t in range(enc32.shape[0]):
                for it in range(cap):
                    l32, _, a32 = e32.joint(hp32[u][None], enc32[t][None])
                    lbf, _, abf = ebf.joint(hpbf[u][None], encbf[t][None])
                    a32, abf = int(a32[0]), int(abf[0])
                    n_dec += 1
                    n_agree += a32 == abf
                    worst = max(worst, float((l32 - lbf).abs().max()))
                    if a32 != abf:
                        top = torch.topk(l32[0], 2).values
                        margins.append(float(top[0] - top[1]))
                    if a32 == blank:
                        break
                    assert u < len(y) and a32 == y[u], f"the fp32 engine left the reference's path at frame {t}, token {u}"
                    u += 1
            assert u == len(y), (u, len(y))
    finally:
        e32.close(); ebf.close()
    rate = n_agree / max(1, n_dec)
    print(f"{name}: {n_dec} joint evaluations on the fp32 reference's path, bf16 operands agree on {n_agree} ({100 * rate:.2f} %); fp32 margins at "
          f"the disagreements {sorted(margins)}; largest |logit difference| {worst:.3f}")
    PR.record(f"{name}_bf16_teacher_forced_decisions", decisions=n_dec, agree=n_agree, rate=round(rate, 5),
              fp32_margins_at_disagreements=sorted(round(m, 4) for m in margins), max_abs_logit_difference=round(worst, 4))
    assert rate >= 0.97 and (not margins or max(margins) < 0.25), (rate, margins)


def test_roctx_ranges_switch():
    """LASR_ROCTX=1 (tracing): the marker library is found, the ranges are pushed around the protocol's host phases
    and the results do not change.  The switch is read at the first lasr_create of a process: a child process per setting."""
    import subprocess
    import sys
    prog = r'''
import sys, zlib
sys.path.insert(0, %r)
import numpy as np
from libreasr_amd import synth
from libreasr_amd.engine import Engine
cfg = synth.model_cfg("tiny"); sd = synth.synth_state_dict(cfg, seed=0)
eng = Engine(sd, cfg, max_streams=4)
slots = [eng.open() for _ in range(4)]
pcm = np.stack([synth.synth_pcm(1, 24 * 1280, seed=7 + s)[0] for s in range(4)])
out = []
for k in range(24):
    eng.push_submit(slots, pcm[:, k * 1280:(k + 1) * 1280])
    while eng.pending() >= 4:
        if eng.wait(): out.append(eng.fetch_many(slots, 64))
while eng.pending():
    if eng.wait(): out.append(eng.fetch_many(slots, 64))
eng.transcribe_pcm(slots[:1], [pcm[0]])
print("RESULT", eng.config("roctx"), zlib.crc32(repr(out).encode()), len(out))
''' % ROOT
    res = {}
    for on in ("0", "1"):
        env = dict(os.environ, LASR_ROCTX=on)
        p = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
        res[on] = (int(line[1]), line[2], int(line[3]))
    print(res)
    assert res["0"][0] == 0 and res["1"][0] == 1          # the image ships the profiler's marker library
    assert res["0"][1:] == res["1"][1:] and res["0"][2] > 8
