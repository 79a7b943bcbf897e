"""On-device resampling for non-16 kHz clients (Resample.encodes, transforms.py:135-144).
torchaudio 0.6.0 is not in the reference tree: PARITY UNPINNED against the reference; the contract is the
oracle's restatement of kaldi LinearResample (oracle/rnnt_oracle.py:resample) + signal properties."""
import numpy as np
import pytest
import torch

from libreasr_amd import synth
from oracle import rnnt_oracle as O

pytestmark = pytest.mark.gpu


def _engine():
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    graft.build()
    cfg = synth.model_cfg("tiny")
    return Engine(synth.synth_state_dict(cfg, seed=0), cfg, max_streams=4), cfg


@pytest.mark.parametrize("sr", [48000, 44100, 8000, 22050, 32000])
def test_resample_matches_oracle_and_keeps_a_tone(sr):
    eng, _ = _engine()
    rng = np.random.default_rng(sr)
    n = sr + 123
    t = np.arange(n) / sr
    x = np.stack([0.5 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.standard_normal(n),
                  0.3 * np.sin(2 * np.pi * 1500 * t + 1.0)]).astype(np.float32)
    y = eng.resample(torch.as_tensor(x).cuda(), sr).cpu().numpy()
    for b in range(2):
        ref = O.resample(x[b], sr)
        assert y.shape[1] == len(ref) == O.resample_num_out(n, sr, 16000)
        assert np.abs(y[b] - ref).max() < 2e-5
    tone = 0.3 * np.sin(2 * np.pi * 1500 * np.arange(y.shape[1]) / 16000 + 1.0)
    assert np.abs(y[1][300:-300] - tone[300:-300]).max() < 2e-3      # a band-limited tone survives the rate change
    eng.close()


def test_non_16k_client_through_the_mirror():
    """x_tfm / x_tfm_stream with AudioTensor.sr = 48000: resample -> log-mel -> stack, as the reference pipeline
    orders them (transforms.py order 2 before TransformTime), checked against the oracle chain."""
    from libreasr_amd.lib.transforms import AudioTensor, OfflinePipeline, StreamPipeline
    eng, cfg = _engine()
    pcm48 = synth.synth_pcm(1, 48000 * 2, seed=5, sr=48000)[0]
    feats = OfflinePipeline(eng)(AudioTensor(torch.as_tensor(pcm48[None]), 48000))[0, :, :, 0].cpu().numpy()
    ref = O.features_offline(O.resample(pcm48, 48000))
    assert feats.shape == ref.shape
    assert np.abs(feats - ref).max() < 2e-3
    # streaming: 3-chunk windows of 3 x 3840 samples at 48 kHz, each window resampled on its own
    sp = StreamPipeline(eng)
    chunks = pcm48[: 3840 * 8].reshape(8, 3840)
    outs = []
    for k in range(2, 8):
        o = sp(AudioTensor(torch.as_tensor(np.concatenate(chunks[k - 2:k + 1])[None]), 48000))
        if o is not None:
            outs.append(o[:, :, 0].cpu().numpy())
    assert len(outs) == 3
    win = O.resample(np.concatenate(chunks[0:3]), 48000)
    spec = O.stream_postprocess(O.logmel(win))
    first = O.stack_downsample(spec)
    assert np.abs(outs[0][0] - first[0]).max() < 2e-3
    eng.close()
