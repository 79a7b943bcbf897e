"""Repeat tests: the scenarios whose single runs were green on one box and red on another.

Round 3's driver run failed test_streaming_matches_reference[tiny-3.0-3] on one token of one stream.  tests/soak.py
reproduced it at 3e-3 per iteration once the tests that precede it had run in the same process, and its state dump showed
the DEVICE had the right token and the right state: the host had read the step's token block before the copy of it had
landed (the synchronous protocol sent "payload copy, then flag copy" with hipMemcpyAsync and spun on the flag).  The results
are now stored and published by a kernel (k_publish: payload, system-scope fence, flag).  These tests repeat the scenario
often enough to see a 1e-2 per-iteration fault with near certainty and keep the state dump wired (lasr_debug_read)."""
import os
import sys

import numpy as np
import pytest

from libreasr_amd import synth

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))      # tests/soak.py

pytestmark = pytest.mark.gpu


def test_soak_sync_three_streams_against_the_reference_goldens(golden_dir):
    """Body of test_streaming_matches_reference[tiny-3.0-3] 300 times on the shared engine (the driver's failing case)."""
    import soak
    import test_gpu_parity as T
    bad, n = soak.scenario_sync(T, "tiny", 3.0, 3, 300, True, golden_dir, 0)
    assert (bad, n) == (0, 300)


def test_soak_pipelined_four_streams_out_of_phase(golden_dir):
    """4 streams started 0..3 chunks apart, 3 steps in flight, push + submit and push_submit: 100 times each."""
    import soak
    import test_gpu_parity as T
    for fused in (False, True):
        bad, n = soak.scenario_pipe(T, "tiny", 3.0, 4, 100, golden_dir, 3, 0, fused)
        assert (bad, n) == (0, 100), f"fused={fused}"


def test_debug_read_matches_the_oracle_state_after_a_model_step(golden_dir):
    """lasr_debug_read (the soak tool's state dump): after each model step of one stream the resident PCM ring, pending
    log-mel frames, LayerNorm'ed features, encoder output / state and the joint's encoder half equal the oracle's."""
    import soak
    import test_gpu_parity as T
    eng, m, cfg = T.engine("tiny")
    pcm = synth.synth_pcm(1, 16000, seed=77)[0]
    chunks = synth.stream_chunks(pcm, 1280, lead=1, tail=2)
    tr = soak.OracleTrace(m, chunks)
    slot = eng.open()
    try:
        j = 0
        for ch in chunks:
            eng.push([slot], T.dev(ch[None]))
            if eng.step([slot]):
                st = soak.dump_state(eng, slot, tr, j)
                assert st["ring"] == 0.0
                assert max(st["pend"]) < 2e-4 and max(st["x0"]) < 2e-4
                assert max(st["enc_out"] + st["enc_h"] + st["enc_c"] + st["pe"]) < 5e-4, st
                assert st["ints"]["ring_pos"] == st["ints"]["h_ring_pos"]
                j += 1
            eng.fetch(slot)
        assert j == len(tr.build()) and j >= 5
    finally:
        eng.close_slot(slot)


def test_soak_beam_on_the_pipelined_protocol():
    """The pipelined beam protocol (selection rounds replayed into the host trees by the pump thread, non-extended slots carried by
    k_beam_carry) against the oracle after every model step, 30 times per predictor cell."""
    import test_gpu_beam as B
    for name in ("tiny", "tiny_lstm"):
        for _ in range(30):
            B.test_beam_on_the_pipelined_protocol_equals_the_oracle_per_model_step(name, 4)
