"""Race detectors of the two-stream (pipelined) protocol -- SURVEY.md section 5 "race detection", round 6.

The encoder side (front-end, LayerNorm, LSTM cells) depends on nothing the decode stream computes, so everything it produces
in a pipelined run must equal a synchronous run of the same input bit for bit.  lasr_debug_enclog stores exact per-row checksums
of the encoder's inputs and state behind every model step; lasr_debug_fe_race runs the streaming log-mel kernel back to back
beside one decode kernel at a time.  Round 6 found configs[4]'s shape (8 x 1536 bf16, beam 8, 128 streams) NOT reproducible run
to run: a few waves of the log-mel kernel per thousand returned wrong spectra whenever its workgroups shared a CU with the wide
decode tilings (profiles/r06/r06_experiments.txt R).  These tests are the regression gate of the fix (lasr_ctx::fe_lds_pad)."""
import os

import numpy as np
import pytest

from libreasr_amd import synth

pytestmark = pytest.mark.gpu


def _engine(name, beam, streams, dtype, env=None):
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    graft.build()
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        cfg = synth.model_cfg(name)
        sd = synth.synth_state_dict(cfg, seed=0)
        return Engine(sd, cfg, max_streams=streams, dtype=dtype, beam=beam), cfg
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _run(eng, slots, pcm, mode, depth, cap):
    for s in slots:
        eng.reset(s, 15)
    eng.debug_enclog()
    n_chunks = pcm.shape[1] // 1280
    out = []
    for k in range(n_chunks):
        x = pcm[:, k * 1280:(k + 1) * 1280]
        if mode == "sync":
            eng.push(slots, x)
            if eng.step(slots):
                out.append(eng.fetch_many(slots, cap))
            continue
        eng.push_submit(slots, x)
        while eng.pending() >= depth:
            if eng.wait():
                out.append(eng.fetch_many(slots, cap))
    while eng.pending():
        if eng.wait():
            out.append(eng.fetch_many(slots, cap))
    return eng.debug_enclog(), out


# (model, beam, streams, dtype, steps in flight, chunks, pipelined runs)
CASES = [
    ("cfg5", 8, 128, "bf16", 6, 32, 6),      # configs[4] per GPU: the shape that failed (12 of 12 runs before the fix)
    ("cfg5", 4, 128, "bf16", 6, 32, 4),      # 512 hypothesis rows: the same wide tilings
    ("cfg2", 8, 128, "bf16", 6, 32, 4),
    ("cfg2", 4, 64, "bf16", 6, 32, 3),       # configs[2]
    ("cfg2", 1, 64, "f32", 18, 64, 3),       # configs[1]
    ("cfg5", 1, 128, "bf16", 6, 32, 3),
]


@pytest.mark.parametrize("name,W,B,dtype,depth,n_chunks,runs", CASES)
def test_encoder_side_of_a_pipelined_run_equals_the_synchronous_run(name, W, B, dtype, depth, n_chunks, runs):
    eng, cfg = _engine(name, W, B, dtype, {"LASR_DBG_ENCLOG": "64"})
    try:
        slots = [eng.open() for _ in range(B)]
        pcm = np.stack([synth.synth_pcm(1, n_chunks * 1280, seed=1234 + s)[0] for s in range(B)])
        cap = 8192 if W > 1 else 64
        ref, ref_out = _run(eng, slots, pcm, "sync", depth, cap)
        assert ref.shape[0] >= n_chunks // 2 - 2 and ref.any()
        bad = []
        for r in range(runs):
            cur, cur_out = _run(eng, slots, pcm, "pipe", depth, cap)
            assert cur.shape == ref.shape
            d = ref[:, :, :B] != cur[:, :, :B]
            if d.any():
                st = int(np.nonzero(d.any(axis=(1, 2)))[0][0])
                bad.append((r, st, [int(x) for x in np.nonzero(d[st].any(axis=0))[0]][:8]))
        print(f"{name} beam {W} x {B} {dtype}: {ref.shape[0]} steps logged, {runs} pipelined runs, differing (run, first step, rows): {bad}")
        assert not bad
    finally:
        eng.close()


def test_log_mel_kernel_beside_each_decode_kernel():
    eng, cfg = _engine("cfg5", 8, 128, "bf16")
    try:
        B = 128
        assert eng.config("fe_lds_pad") > 0            # the wide decode tilings can run in this context
        slots = [eng.open() for _ in range(B)]
        pcm = np.stack([synth.synth_pcm(1, 8 * 1280, seed=1234 + s)[0] for s in range(B)])
        for k in range(8):
            eng.push(slots, pcm[:, k * 1280:(k + 1) * 1280])
            if eng.step(slots):
                eng.fetch_many(slots, 8192)
        for agg, nm in ((0, "nothing"), (1, "vocabulary GEMM"), (2, "predictor pass"), (3, "joint half")):
            bl, br = eng.debug_fe_race(300, agg, 4)
            print(f"beside {nm}: {bl} of 300 launches differ ({br} rows)")
            assert (bl, br) == (0, 0)
    finally:
        eng.close()


def test_wide_tilings_do_not_disturb_even_without_the_lds_pad():
    """First line of defence: the wide decode tilings run on v_mfma_f32_16x16x16_bf16 (OpsBF16k16), not on the 16x16x32 form that
    disturbed its CU neighbours -- the probe is clean with lds_pad = 0 as well (27-37 of 1 000 launches wrong before)."""
    eng, cfg = _engine("cfg5", 8, 128, "bf16")
    try:
        B = 128
        slots = [eng.open() for _ in range(B)]
        pcm = np.stack([synth.synth_pcm(1, 8 * 1280, seed=1234 + s)[0] for s in range(B)])
        for k in range(8):
            eng.push(slots, pcm[:, k * 1280:(k + 1) * 1280])
            if eng.step(slots):
                eng.fetch_many(slots, 8192)
        for agg, nm in ((1, "vocabulary GEMM"), (2, "predictor pass")):
            bl, br = eng.debug_fe_race(500, agg, 4, lds_pad=0)
            print(f"no pad, beside {nm}: {bl} of 500 launches differ ({br} rows)")
            assert (bl, br) == (0, 0)
    finally:
        eng.close()


def test_configs1_context_keeps_the_plain_launch():
    eng, cfg = _engine("cfg2", 1, 64, "f32")
    try:
        assert eng.config("fe_lds_pad") == 0           # no wide tiling can run: the headline's front-end launch is unchanged
    finally:
        eng.close()


def test_beam_results_reproducible_run_to_run():
    name, W, B, n_chunks = "cfg5", 8, 128, 32
    eng, cfg = _engine(name, W, B, "bf16")
    try:
        slots = [eng.open() for _ in range(B)]
        pcm = np.stack([synth.synth_pcm(1, n_chunks * 1280, seed=1234 + s)[0] for s in range(B)])
        runs = []
        for r in range(4):
            for s in slots:
                eng.reset(s, 15)
            hist = [[] for _ in range(B)]
            score = [0.0] * B
            for k in range(n_chunks):
                eng.push_submit(slots, pcm[:, k * 1280:(k + 1) * 1280])
                while eng.pending() >= 6:
                    if eng.wait():
                        for i in range(B):
                            t, nl, _ = eng.fetch(slots[i])
                            hist[i].append(tuple(t)); score[i] = -nl
            while eng.pending():
                if eng.wait():
                    for i in range(B):
                        t, nl, _ = eng.fetch(slots[i])
                        hist[i].append(tuple(t)); score[i] = -nl
            runs.append((hist, score))
        for r in range(1, len(runs)):
            dh = [i for i in range(B) if runs[r][0][i] != runs[0][0][i]]
            ds = [i for i in range(B) if runs[r][1][i] != runs[0][1][i]]
            print(f"run {r}: streams with another history {dh[:8]}, another final score {ds[:8]}")
            assert not dh and not ds
    finally:
        eng.close()
