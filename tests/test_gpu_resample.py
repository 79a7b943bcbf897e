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


def _oracle_stream(m, chunks, sr=16000, eng=None):
    """The servicer's per-call sequence on the oracle.  With `eng`, the Resample of each window is the engine's kernel
    (checked against the oracle's to 2e-5 in test_resample_matches_oracle_and_keeps_a_tone): a 1e-5 difference in the PCM
    can flip a near-tie argmax of the tiny random model hundreds of tokens into a stream, which says nothing about the
    window pipeline this test is about."""
    fe, dec = O.StreamFrontend(sr=16000 if eng is not None else sr), m.stream_decoder()
    per_call, frames = [], []
    for c in chunks:
        if eng is not None and sr != 16000:          # StreamFrontend's own windowing, with the resampled window injected
            frames.append(c)
            if len(frames) != 3:
                continue
            win = np.concatenate(frames)
            del frames[0]
            aud = eng.resample(torch.as_tensor(win[None]).cuda(), sr)[0].cpu().numpy()
            spec = O.stream_postprocess(O.logmel(aud), fe.n_stack)
            fe.saved.append(O.stack_downsample(spec, fe.n_stack, fe.downsample))
            o = None
            if len(fe.saved) == fe.n_buffer:
                o = np.concatenate(fe.saved, axis=0)
                fe.saved = []
        else:
            o = fe.push(c)
        if o is not None:
            per_call.append(dec.step(o))
    return per_call, dec


@pytest.mark.parametrize("sr,chunk", [(16000, 1280), (48000, 3840), (44100, 4410), (16000, 2000), (8000, 800)])
def test_step_window_generic_clients_match_the_oracle(sr, chunk):
    """lasr_step_window: the servicer's per-call sequence (3-frame window -> Resample of the window -> log-mel -> frames
    T//3+1.. -> stack -> Buffer -> model) for any client rate / frame length, two slots out of phase in one batch."""
    eng, cfg = _engine()
    m = O.OracleTransducer(synth.synth_state_dict(cfg, seed=0), cfg)
    pcm = synth.synth_pcm(2, sr * 2, seed=77, sr=sr)
    chunks = [synth.stream_chunks(pcm[i], chunk, lead=1, tail=6) for i in range(2)]
    ref, decs = zip(*[_oracle_stream(m, chunks[i], sr, eng) for i in range(2)])
    full = [_oracle_stream(m, chunks[i], sr)[0] for i in range(2)]        # resampling by the oracle too
    slots = [eng.open(), eng.open()]
    frames = [[], []]
    got = [[], []]
    n = len(chunks[0])
    for k in range(n + 1):                       # slot 1 starts one frame late: its model steps fall on other calls
        batch, wins = [], []
        for i in range(2):
            kk = k - i
            if not 0 <= kk < n:
                continue
            frames[i].append(chunks[i][kk])
            if len(frames[i]) == 3:
                batch.append(i)
                wins.append(np.concatenate(frames[i]))
                del frames[i][0]
        if not batch:
            continue
        before = [len(got[i]) for i in batch]
        eng.step_window([slots[i] for i in batch], np.stack(wins), sr)
        for i in batch:
            t = eng.fetch(slots[i])[0]
            got[i].append(t)
    for i in range(2):
        calls = [t for t in got[i]]
        # the engine returns [] for calls that did not run the model; the oracle lists model calls only
        flat_got = [tok for t in calls for tok in t]
        flat_ref = [tok for t in ref[i] for tok in t]
        if flat_got != flat_ref:
            # joint logits agree with the reference path to 1e-3 (north star): a decision whose top-1 / top-2 logits are closer
            # than that may legitimately go the other way; everything before the first such decision must be identical
            p = next((q for q in range(min(len(flat_got), len(flat_ref))) if flat_got[q] != flat_ref[q]), min(len(flat_got), len(flat_ref)))
            margins = [mg for n_before, mg in decs[i].decisions if n_before == p]
            assert margins and min(margins) < 2e-3, (sr, chunk, i, p, margins[:4])
            print(f"sr {sr} chunk {chunk} slot {i}: identical up to token {p} of {len(flat_ref)}, then a margin-tie ({min(margins):.2e})")
        assert len(flat_ref) > 0
        flat_full = [tok for t in full[i] for tok in t]
        assert flat_got[:50] == flat_full[:50]                               # (identical up to the first near-tie)
    eng.close()


def test_step_window_rejects_short_windows_and_mixed_forms():
    from libreasr_amd._native import LASR_EINVAL, LASR_ESTATE, LasrError
    eng, _ = _engine()
    s = eng.open()
    with pytest.raises(LasrError) as e:
        eng.step_window([s], np.zeros((1, 300), np.float32), 16000)
    assert e.value.code == LASR_EINVAL
    for _ in range(3):                                   # three fused-path chunks leave one frame pending in the ring
        eng.push([s], np.zeros((1, 1280), np.float32))
        eng.step([s])
    with pytest.raises(LasrError) as e:
        eng.step_window([s], np.zeros((1, 3840), np.float32), 16000)
    assert e.value.code == LASR_ESTATE
    eng.close()
