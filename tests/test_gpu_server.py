"""GPU test of the gRPC surface (libreasr.proto wire format, api-server.py semantics) with the
batching scheduler: concurrent TranscribeStream clients + a unary Transcribe, against transcripts
derived from the oracle with the servicer's own diff / reset rules."""
import itertools as it
import threading

import numpy as np
import pytest

from libreasr_amd import synth
from oracle import rnnt_oracle as O

pytestmark = pytest.mark.gpu


def expected_stream_transcripts(m, pcm, lang, sr=16000, chunk=1280):
    """api-server.py:118-135 applied to the oracle's per-call outputs."""
    fe, dec = O.StreamFrontend(sr=sr), m.stream_decoder()
    out, y, last, last_diff, steps = [], [], "", "", 0
    for c in synth.stream_chunks(pcm, chunk, lead=1, tail=10):
        o = fe.push(c)
        if o is None:
            continue
        y_seq = dec.step(o)
        steps += 1
        y = y + y_seq
        if lang.denumericalize(y_seq) != "":
            now = lang.denumericalize(y)
            diff = "".join(b for a, b in it.zip_longest(last, now) if a != b)
            last = now
            if diff == last_diff:
                continue
            last_diff = diff
            out.append(diff)
        elif O.should_reset(steps):
            dec.reset()
            steps = 0
    return out


def test_grpc_server_batched_streams_and_unary():
    import grpc
    import __graft_entry__ as graft
    graft.build()
    from libreasr_amd import server as srv
    from libreasr_amd.interfaces import libreasr_pb2 as ap
    from libreasr_amd.interfaces import libreasr_pb2_grpc as apg
    from libreasr_amd.lib.language import IdLanguage

    server, sched, port = srv.serve("en", port="127.0.0.1:0", block=False, config_path="/nonexistent.yaml",
                                    synthetic="tiny", max_streams=16)
    try:
        cfg = synth.model_cfg("tiny")
        m = O.OracleTransducer(synth.synth_state_dict(cfg, seed=0), cfg)
        lang = IdLanguage()
        n = 5
        pcm = synth.synth_pcm(n, 16000 * 3, seed=1234)
        got = [None] * n
        barrier = threading.Barrier(n)

        def client(i):
            with grpc.insecure_channel(f"127.0.0.1:{port}") as ch:
                stub = apg.ASRStub(ch)

                def reqs():                                            # api-client.py:32-47
                    barrier.wait()
                    for c in synth.stream_chunks(pcm[i], 1280, lead=1, tail=10):
                        yield ap.Audio(data=c.tobytes(), sr=16000)

                got[i] = [t.data for t in stub.TranscribeStream(reqs())]

        ths = [threading.Thread(target=client, args=(i,)) for i in range(n)]
        [t.start() for t in ths]
        [t.join(timeout=120) for t in ths]
        for i in range(n):
            assert got[i] == expected_stream_transcripts(m, pcm[i], lang), f"stream {i}"
        assert max(sched.batches) > 1, "concurrent streams were never stepped as one batch"
        with grpc.insecure_channel(f"127.0.0.1:{port}") as ch:
            stub = apg.ASRStub(ch)
            text = stub.Transcribe(ap.Audio(data=pcm[0].tobytes(), sr=16000)).data
            assert text == lang.denumericalize(m.decode_greedy(O.features_offline(pcm[0]))[0])
            # a 48 kHz unary request is resampled on the GPU first (Resample.encodes, transforms.py:135-144)
            pcm48 = synth.synth_pcm(1, 48000 * 2, seed=9, sr=48000)[0]
            text48 = stub.Transcribe(ap.Audio(data=pcm48.tobytes(), sr=48000)).data
            assert text48 == lang.denumericalize(m.decode_greedy(O.features_offline(O.resample(pcm48, 48000)))[0])
            # a 48 kHz streaming client (the browser's rate, apps/web/src/App.js:68), 80 ms = 3840-sample frames: every
            # 3-frame window is resampled and transformed as a whole, as the servicer does (api-server.py:83-115)
            got48 = [t.data for t in stub.TranscribeStream(
                ap.Audio(data=c.tobytes(), sr=48000) for c in synth.stream_chunks(pcm48, 3840, lead=1, tail=10))]
            assert got48 == expected_stream_transcripts(m, pcm48, lang, sr=48000, chunk=3840)
            assert len(got48) > 0
            # 16 kHz frames of another length (100 ms) go the same way
            got100 = [t.data for t in stub.TranscribeStream(
                ap.Audio(data=c.tobytes(), sr=16000) for c in synth.stream_chunks(pcm[1], 1600, lead=1, tail=8))]
            exp100 = []
            fe, dec = O.StreamFrontend(), m.stream_decoder()
            y, last, last_diff, steps = [], "", "", 0
            for c in synth.stream_chunks(pcm[1], 1600, lead=1, tail=8):
                o = fe.push(c)
                if o is None:
                    continue
                y_seq = dec.step(o)
                steps += 1
                y = y + y_seq
                if lang.denumericalize(y_seq) != "":
                    now = lang.denumericalize(y)
                    diff = "".join(b for a, b in it.zip_longest(last, now) if a != b)
                    last = now
                    if diff == last_diff:
                        continue
                    last_diff = diff
                    exp100.append(diff)
                elif O.should_reset(steps):
                    dec.reset()
                    steps = 0
            assert got100 == exp100
            with pytest.raises(grpc.RpcError):                         # windows too short for 10 frames are refused, not mis-decoded
                list(stub.TranscribeStream(ap.Audio(data=np.zeros(100, np.float32).tobytes(), sr=16000) for _ in range(3)))
    finally:
        server.stop(0)
        sched.shutdown()
