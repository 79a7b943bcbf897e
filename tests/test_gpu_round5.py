"""Round 5: the long path and the servicer pinned to the reference itself.

* engine == goldens the reference's own decode produced at SURVEY 8d's workload length (cfg2: 3 streams x 330 400 samples =
  20.65 s, T' = 258; ref6 / cfg5: 10 s), offline and on both streaming protocols;
* libreasr_amd.server's TranscribeStream message sequence == what the reference's OWN ASRServicer.TranscribeStream
  (api-server.py:82-135, imported by oracle/ref_fixture.py:ref_servicer) emitted for the same streams, resets included."""
import os
import threading

import numpy as np
import pytest
import torch

from libreasr_amd import synth

pytestmark = pytest.mark.gpu

_ENGINES = {}


def engine(name, max_streams=16):
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    if name not in _ENGINES:
        graft.build()
        cfg = synth.model_cfg(name)
        _ENGINES[name] = Engine(synth.synth_state_dict(cfg, seed=0), cfg, max_streams=max_streams)
    return _ENGINES[name]


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


LONG = [("cfg2", 3), ("ref6", 1), ("cfg5", 1)]


@pytest.mark.parametrize("name,n_streams", LONG)
def test_long_offline_matches_reference(name, n_streams, golden_dir):
    eng = engine(name)
    g = np.load(os.path.join(golden_dir, f"model_{name}_long.npz"))
    pcm = synth.synth_pcm(n_streams, int(g["n_samples"]), seed=1234)
    slots = [eng.open() for _ in range(n_streams)]
    try:
        eng.transcribe_pcm(slots, [dev(p) for p in pcm])
        for s, slot in enumerate(slots):
            toks, neg_logp, align = eng.fetch(slot, cap=4096)
            assert toks == list(g[f"off_tokens_{s}"]), f"stream {s}: tokens differ from the reference"
            assert abs(neg_logp - float(g[f"off_neglogp_{s}"])) < 5e-2
            assert abs(align - float(g[f"off_align_{s}"])) < 1e-9
    finally:
        for slot in slots:
            eng.close_slot(slot)


@pytest.mark.parametrize("name,n_streams", LONG)
@pytest.mark.parametrize("protocol", ["sync", "pipelined"])
def test_long_streaming_matches_reference(name, n_streams, protocol, golden_dir):
    eng = engine(name)
    g = np.load(os.path.join(golden_dir, f"model_{name}_long.npz"))
    pcm = synth.synth_pcm(n_streams, int(g["n_samples"]), seed=1234)
    slots = [eng.open() for _ in range(n_streams)]
    try:
        chunks = [synth.stream_chunks(p, 1280, lead=1, tail=10) for p in pcm]
        got = [[] for _ in slots]
        counts = [[] for _ in slots]

        def collect():
            if eng.wait():
                for s, t in enumerate(eng.fetch_many(slots, 64)):
                    got[s] += t
                    counts[s].append(len(t))

        for k in range(len(chunks[0])):
            batch = dev(np.stack([c[k] for c in chunks]))
            if protocol == "sync":
                eng.push(slots, batch)
                ran = eng.step(slots)
                for s, slot in enumerate(slots):
                    t, _, _ = eng.fetch(slot)
                    got[s] += t
                    if ran:
                        counts[s].append(len(t))
            else:
                eng.push_submit(slots, batch)
                if eng.pending() >= 12:
                    collect()
        while protocol != "sync" and eng.pending():
            collect()
        for s in range(n_streams):
            assert got[s] == list(g[f"st_tokens_{s}"]), f"stream {s}: streaming tokens differ from the reference"
            assert counts[s] == list(g[f"st_counts_{s}"])
    finally:
        for slot in slots:
            eng.close_slot(slot)


def servicer_golden(golden_dir, name="tiny"):
    g = np.load(os.path.join(golden_dir, f"servicer_{name}.npz"))
    out = []
    for i in range(int(g["n"])):
        n = int(g[f"n_msgs_{i}"])
        out.append(([str(v) for v in g[f"msgs_{i}"][:n]], [int(v) for v in g[f"resets_{i}"]], str(g[f"unary_{i}"])))
    return out


def test_grpc_server_equals_the_reference_servicer(golden_dir):
    """Seven concurrent TranscribeStream clients (api-client.py:32-47 chunking) + the unary RPC per stream through
    libreasr_amd.server on the batched scheduler: message for message what the reference's own servicer emitted (goldens from
    api-server.py:82-135 itself -- not from a restatement of it), incl. resets inside a > 4 s silence and before any token."""
    import grpc
    import __graft_entry__ as graft
    graft.build()
    from libreasr_amd import server as srv
    from libreasr_amd.interfaces import libreasr_pb2 as ap
    from libreasr_amd.interfaces import libreasr_pb2_grpc as apg

    gold = servicer_golden(golden_dir)
    pcm = [synth.servicer_pcm(seed, spec) for seed, spec in synth.SERVICER_STREAMS]
    n = len(pcm)
    assert n == len(gold)
    server, sched, port = srv.serve("en", port="127.0.0.1:0", block=False, config_path="/nonexistent.yaml",
                                    synthetic="tiny", max_streams=16)
    try:
        got = [None] * n
        barrier = threading.Barrier(n)

        def client(i):
            with grpc.insecure_channel(f"127.0.0.1:{port}") as ch:
                stub = apg.ASRStub(ch)

                def reqs():
                    barrier.wait()
                    for c in synth.stream_chunks(pcm[i], 1280, lead=1, tail=10):
                        yield ap.Audio(data=c.tobytes(), sr=16000)

                got[i] = [t.data for t in stub.TranscribeStream(reqs())]

        ths = [threading.Thread(target=client, args=(i,)) for i in range(n)]
        [t.start() for t in ths]
        [t.join(timeout=180) for t in ths]
        for i in range(n):
            assert got[i] == gold[i][0], f"stream {i} {synth.SERVICER_STREAMS[i]}"
        assert sum(len(gold[i][1]) for i in range(n)) >= 6          # the goldens do contain resets
        with grpc.insecure_channel(f"127.0.0.1:{port}") as ch:
            stub = apg.ASRStub(ch)
            for i in range(n):
                assert stub.Transcribe(ap.Audio(data=pcm[i].tobytes(), sr=16000)).data == gold[i][2]
    finally:
        server.stop(0)
        sched.shutdown()


@pytest.mark.parametrize("name,n_sec,n_streams", [("tiny", 3.0, 3), ("cfg2", 4.0, 2)])
def test_x_side_gemm_mode_matches_reference(name, n_sec, n_streams, golden_dir, monkeypatch):
    """LASR_ENC_XG=1 (round 5, measured slower and left off: profiles/r05/r05_experiments.txt C): the x side of a layer's frames as
    ONE GEMM per model step + recurrent cells with K = H.  Different summation order ((x sum + bias) + h sum), same contract:
    tokens == the reference's goldens offline and on the pipelined protocol, encoder output within the fused cell's tolerance."""
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    graft.build()
    monkeypatch.setenv("LASR_ENC_XG", "1")
    cfg = synth.model_cfg(name)
    eng = Engine(synth.synth_state_dict(cfg, seed=0), cfg, max_streams=16)
    try:
        assert eng.config("enc_xg") == 1
        g = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
        pcm = synth.synth_pcm(n_streams, int(16000 * n_sec), seed=1234)
        slots = [eng.open() for _ in range(n_streams)]
        eng.transcribe_pcm(slots, [dev(p) for p in pcm])
        for s, slot in enumerate(slots):
            toks, neg_logp, _ = eng.fetch(slot)
            assert toks == list(g[f"off_tokens_{s}"])
            assert abs(neg_logp - float(g[f"off_neglogp_{s}"])) < 2e-2
        from oracle import rnnt_oracle as O
        feats = O.features_offline(pcm[0])
        enc = eng.encoder(dev(feats[None].astype(np.float32)))[0].cpu().numpy()
        assert float(np.abs(enc - g["enc_out_0"]).max()) < 5e-4
        for slot in slots:
            eng.reset(slot, 15)
        chunks = [synth.stream_chunks(p, 1280, lead=1, tail=10) for p in pcm]
        got = [[] for _ in slots]
        counts = [[] for _ in slots]

        def collect():
            if eng.wait():
                for s, t in enumerate(eng.fetch_many(slots, 64)):
                    got[s] += t
                    counts[s].append(len(t))

        for k in range(len(chunks[0])):
            eng.push_submit(slots, dev(np.stack([c[k] for c in chunks])))
            if eng.pending() >= 6:
                collect()
        while eng.pending():
            collect()
        for s in range(n_streams):
            assert got[s] == list(g[f"st_tokens_{s}"])
            assert counts[s] == list(g[f"st_counts_{s}"])
    finally:
        eng.close()
