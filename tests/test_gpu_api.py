"""GPU tests of the drop-in Python surface: load_stuff / x_tfm / x_tfm_stream / Transducer.transcribe /
transcribe_stream (the calls api-server.py makes) and the LibreASR facade, against the goldens the
reference produced through the same calls."""
import os

import numpy as np
import pytest
import torch

from libreasr_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stuff():
    import __graft_entry__ as graft
    graft.build()
    from libreasr_amd.lib.inference import load_stuff
    return load_stuff("en", config_path="/nonexistent.yaml", synthetic="tiny", max_streams=16)


def test_servicer_call_sequence_offline(stuff, golden_dir):
    """ASRServicer.Transcribe (api-server.py:64-80): tensorize -> AudioTensor -> x_tfm -> model.transcribe."""
    from libreasr_amd.lib.transforms import AudioTensor
    from libreasr_amd.lib.utils import tensorize
    conf, lang, model, x_tfm, x_tfm_stream = stuff
    g = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    pcm = synth.synth_pcm(3, 48000, seed=1234)
    for s in range(3):
        aud = tensorize(pcm[s].tobytes())
        feats = x_tfm(AudioTensor(aud, 16000))[0]
        assert tuple(feats.shape) == (37, 1280, 1)
        text, metrics = model.transcribe(feats)
        assert text == " ".join(str(t) for t in g[f"off_tokens_{s}"])
        assert abs(metrics["alignment_score"] - float(g[f"off_align_{s}"])) < 1e-9


def test_servicer_call_sequence_stream(stuff, golden_dir):
    """ASRServicer.TranscribeStream (api-server.py:82-134): 3-chunk window -> x_tfm_stream -> transcribe_stream."""
    from libreasr_amd.lib.transforms import AudioTensor
    from libreasr_amd.lib.utils import tensorize
    conf, lang, model, x_tfm, x_tfm_stream = stuff
    g = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    pcm = synth.synth_pcm(3, 48000, seed=1234)
    for s in range(2):
        x_tfm_stream.saved.clear()

        def stream():
            frames = []
            for c in synth.stream_chunks(pcm[s], 1280, lead=1, tail=10):
                frames.append(tensorize(c.tobytes()))
                if len(frames) != 3:
                    continue
                aud = torch.cat(frames, dim=1)
                del frames[0]
                yield x_tfm_stream(AudioTensor(aud, 16000))

        y_all, counts, resets = [], [], 0
        for y, y_one, reset_fn in model.transcribe_stream(stream(), lang.denumericalize):
            counts.append(len(y) - len(y_all))
            y_all = list(y)
            assert callable(reset_fn)
        assert y_all == list(g[f"st_tokens_{s}"])
        assert counts == list(g[f"st_counts_{s}"])


def test_facade(golden_dir):
    from libreasr_amd.api import LibreASR
    asr = LibreASR.load("en", config_path="/nonexistent.yaml", synthetic="tiny", max_streams=16)
    g = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    pcm = synth.synth_pcm(3, 48000, seed=1234)
    ids = asr.transcribe([pcm[0], pcm[1].tobytes(), torch.as_tensor(pcm[2]).cuda()], return_ids=True)
    for s in range(3):
        assert ids[s] == list(g[f"off_tokens_{s}"])
    last = None
    for hyp in asr.stream(synth.stream_chunks(pcm[0], 1280, lead=1, tail=10), return_ids=True):
        last = hyp
    assert last == list(g["st_tokens_0"])
