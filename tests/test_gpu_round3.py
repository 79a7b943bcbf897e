"""Round-3 GPU tests (VERDICT r2 items 2, 5, 6 / ADVICE r2):

  * configs[1] at full size through the protocol, call and depth bench.py runs: lasr_push_submit, 12 model steps in flight,
    64 rows x 72 chunks, every row against the reference's torch-CPU path (oracle/torch_cpu.py, pinned to the reference's
    goldens in tests/test_oracle.py) and 8 rows against the numpy oracle
  * lasr_push_submit == lasr_push_pcm + lasr_step_submit, with streams out of phase (chunks that complete a model step and
    chunks that do not in the same call)
  * host-memory lifetime rules of include/lasr.h: default pushes COPY (a pinned buffer may be overwritten as soon as the
    call returns); LASR_PUSH_PINNED_NOCOPY + lasr_push_consumed(ticket) for the zero-copy form
  * the split front-end (k_fe_mel + k_stack_ln) against the per-chunk kernels: same tokens"""
import os

import numpy as np
import pytest
import torch

from libreasr_amd import synth
from oracle import rnnt_oracle as O

pytestmark = pytest.mark.gpu


def make(name, **kw):
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    graft.build()
    cfg = synth.model_cfg(name)
    sd = synth.synth_state_dict(cfg, seed=0)
    return Engine(sd, cfg, **kw), sd, cfg


def oracle_stream(m, pcm_row, n_chunks, n_buffer=2):
    fe, dec = O.StreamFrontend(n_buffer=n_buffer), m.stream_decoder()
    for k in range(n_chunks):
        o = fe.push(pcm_row[k * 1280:(k + 1) * 1280])
        if o is not None:
            dec.step(o)
    return dec.y


@pytest.mark.parametrize("depth", [12, 20, 25])        # round 3-4's bench depth, round 5's default, the engine's limit
def test_config1_64_rows_depth_12_push_submit_against_the_reference_path(depth):
    from oracle import torch_cpu as TC
    eng, sd, cfg = make("cfg2", max_streams=64)
    try:
        B, n = 64, 72
        pcm = np.stack([synth.synth_pcm(1, n * 1280, seed=1234 + s)[0] for s in range(B)])
        dev = torch.as_tensor(pcm.reshape(B, n, 1280).transpose(1, 0, 2).copy()).cuda()
        slots = [eng.open() for _ in range(B)]
        got = [[] for _ in range(B)]
        deepest = 0

        def collect():
            if eng.wait():
                for i, t in enumerate(eng.fetch_many(slots, 64)):
                    got[i] += t

        for k in range(n):
            eng.push_submit(slots, dev[k])
            deepest = max(deepest, eng.pending())
            if eng.pending() >= depth:
                collect()
        while eng.pending():
            collect()
        assert deepest == depth
        _, ref = TC.time_stream_path_batched(sd, cfg, list(pcm), n, threads=8)
        n_tok = sum(len(r) for r in ref)
        bad = [i for i in range(B) if got[i] != ref[i]]
        assert not bad, f"streams {bad} differ from the reference path"
        m = O.OracleTransducer(sd, cfg)
        for i in range(0, B, 8):
            assert got[i] == oracle_stream(m, pcm[i], n), f"stream {i} (numpy oracle)"
        print(f"configs[1], lasr_push_submit, depth {depth}: 64 streams x {n} chunks, {n_tok} tokens, all equal")
        assert n_tok > 600
    finally:
        eng.close()


def test_push_submit_equals_push_then_submit_with_streams_out_of_phase():
    eng, sd, cfg = make("tiny", max_streams=16)
    try:
        m = O.OracleTransducer(sd, cfg)
        n = 44
        pcm = synth.synth_pcm(4, n * 1280, seed=91)
        start = [0, 1, 1, 3]                        # calls where only some of the listed rows complete a model step
        ref = [oracle_stream(m, pcm[i], n) for i in range(4)]
        assert sum(len(r) for r in ref) > 20
        for fused in (True, False):
            slots = [eng.open() for _ in range(4)]
            got = [[] for _ in range(4)]

            def collect():
                if eng.wait():
                    for i, t in enumerate(eng.fetch_many(slots, 64)):
                        got[i] += t

            for g in range(n + max(start)):
                act = [i for i in range(4) if 0 <= g - start[i] < n]
                sl = [slots[i] for i in act]
                chunk = np.stack([pcm[i][(g - start[i]) * 1280:(g - start[i] + 1) * 1280] for i in act])
                if fused:
                    eng.push_submit(sl, torch.as_tensor(chunk).cuda() if g % 3 else chunk)      # device and host sources
                else:
                    eng.push(sl, chunk)
                    eng.submit(sl)
                if eng.pending() >= 5:
                    collect()
            while eng.pending():
                collect()
            for i in range(4):
                assert got[i] == ref[i], (fused, i)
            for s in slots:
                eng.close_slot(s)
    finally:
        eng.close()


def test_host_buffer_lifetime_rules():
    """Default: host memory (pinned included) is copied before the call returns -> scribbling over the buffer right after
    push / push_submit must not change a token.  LASR_PUSH_PINNED_NOCOPY: the buffer is read later; it may be reused once
    lasr_push_consumed(ticket) says so."""
    from libreasr_amd._native import LASR_EINVAL, LasrError
    eng, sd, cfg = make("tiny", max_streams=16)
    try:
        m = O.OracleTransducer(sd, cfg)
        n = 40
        pcm = synth.synth_pcm(2, n * 1280, seed=17)
        ref = [oracle_stream(m, pcm[i], n) for i in range(2)]
        assert sum(len(r) for r in ref) > 10
        for mode in ("copy_pinned", "copy_pageable", "nocopy"):
            slots = [eng.open() for _ in range(2)]
            got = [[], []]
            bufs = [torch.empty(2, 1280).pin_memory() for _ in range(3)]
            tickets = [None, None, None]
            for k in range(n):
                chunk = np.stack([p[k * 1280:(k + 1) * 1280] for p in pcm])
                if mode == "copy_pageable":
                    buf = chunk.copy()
                    assert eng.push_submit(slots, buf) >= 0
                    buf[:] = 7.0                                # free on return
                elif mode == "copy_pinned":
                    b = bufs[0]
                    b.copy_(torch.as_tensor(chunk))
                    t = eng.push(slots, b) if k % 2 else eng.push_submit(slots, b)
                    b.fill_(7.0)                                # free on return, although pinned
                    assert t >= 0
                    if k % 2:
                        eng.submit(slots)
                else:
                    j = k % 3
                    if tickets[j] is not None:                  # the buffer's previous push must have been read
                        spins = 0
                        while not eng.push_consumed(tickets[j]):
                            spins += 1
                            assert spins < 10_000_000
                        bufs[j].fill_(7.0)
                    bufs[j].copy_(torch.as_tensor(chunk))
                    tickets[j] = eng.push_submit(slots, bufs[j], pinned_nocopy=True)
                    assert tickets[j] >= 0
                if eng.pending() >= 4 and eng.wait():
                    for i, t in enumerate(eng.fetch_many(slots, 64)):
                        got[i] += t
            while eng.pending():
                if eng.wait():
                    for i, t in enumerate(eng.fetch_many(slots, 64)):
                        got[i] += t
            for i in range(2):
                assert got[i] == ref[i], (mode, i)
            for s in slots:
                eng.close_slot(s)
        s = eng.open()
        with pytest.raises(LasrError) as e:         # pageable memory cannot be read in place
            eng.push([s], np.zeros((1, 1280), np.float32), pinned_nocopy=True)
        assert e.value.code == LASR_EINVAL
        assert eng.push([s], torch.zeros(1, 1280).cuda()) == -1      # device memory: stream-ordered, no ticket
    finally:
        eng.close()


@pytest.mark.parametrize("n_buffer", [2, 3])
def test_split_frontend_equals_per_chunk_kernels(n_buffer):
    n, n_chunks = 3, 36
    pcm = synth.synth_pcm(n, n_chunks * 1280, seed=33)
    out = {}
    for name, env in (("split", {}), ("per_chunk", {"LASR_FE_LEGACY": "1"})):
        os.environ.update(env)
        try:
            eng, sd, cfg = make("tiny", max_streams=16, n_buffer=n_buffer)
        finally:
            for k in env:
                os.environ.pop(k, None)
        try:
            slots = [eng.open() for _ in range(n)]
            got = [[] for _ in range(n)]
            for k in range(n_chunks + 1):
                rows = [i for i in range(n) if 0 <= k - (i == 1) < n_chunks]         # stream 1 starts one call late
                chunk = np.stack([pcm[i][(k - (i == 1)) * 1280:(k - (i == 1) + 1) * 1280] for i in rows])
                eng.push_submit([slots[i] for i in rows], chunk)
                if eng.pending() >= 3 and eng.wait():
                    for i, t in enumerate(eng.fetch_many(slots, 64)):
                        got[i] += t
            while eng.pending():
                if eng.wait():
                    for i, t in enumerate(eng.fetch_many(slots, 64)):
                        got[i] += t
            out[name] = got
        finally:
            eng.close()
    m = O.OracleTransducer(sd, cfg)
    ref = [oracle_stream(m, pcm[i], n_chunks, n_buffer) for i in range(n)]
    assert sum(len(r) for r in ref) > 0
    for name, got in out.items():
        assert got == ref, name
