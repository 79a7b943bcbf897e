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


def _vocab64_yttm_model(path):
    """A 64-token YouTokenToMe model file for the tiny shape: pad/unk/bos/eos 0..3, '▁' + 'a'..'z' + "'" as 4..31, 32 merges."""
    chars = {0x2581: 4}
    for k, ch in enumerate("abcdefghijklmnopqrstuvwxyz'"):
        chars[ord(ch)] = 5 + k
    rules, nxt = [], 32
    for k in range(32):                                   # ▁a, ▁b, .. then ab, bc, ..: any valid chain of merges will do
        x, y = (4, 5 + k) if k < 20 else (5 + (k - 20), 6 + (k - 20))
        rules.append((x, y, nxt))
        nxt += 1
    with open(path, "w", encoding="utf-8") as f:
        f.write(f"{len(chars)} {len(rules)}\n")
        for cp, i in chars.items():
            f.write(f"{cp} {i}\n")
        for x, y, z in rules:
            f.write(f"{x} {y} {z}\n")
        f.write("1 0 2 3\n")


def test_model_archive_to_engine_end_to_end(tmp_path, monkeypatch, golden_dir):
    """a release archive as the reference ships it (model_utils.py:20-58: <lang>/model.pth in fastai's
    learn.save format {"model": state_dict, "opt": ...} + <lang>/tokenizer.yttm-model in libreasr-model-<lang>.tar.gz, and
    <lang>/lm.pth, lm.py:93) found in the working directory -> load_stuff -> engine on the GPU -> text.  Tokens must equal
    the goldens the reference produced with its own LM class attached; the text is those ids through the YTTM model."""
    import tarfile
    import __graft_entry__ as graft
    graft.build()
    from libreasr_amd.api import LibreASR
    from libreasr_amd.lib import yttm
    from libreasr_amd.lib.language import TokenizedLanguage
    cfg = synth.model_cfg("tiny_soft")
    sd = synth.synth_state_dict(cfg, seed=0)
    lsd = synth.synth_lm_state_dict("tiny_lm")
    src = tmp_path / "build" / "en"
    src.mkdir(parents=True)
    wrapped = {"model": {k: torch.as_tensor(v) for k, v in sd.items()},
               "opt": {"state": [], "hypers": [{"lr": 1e-3}]}}                       # fastai learn.save(with_opt=True)
    wrapped["model"]["encoder.rnn_stack.bns.0.num_batches_tracked"] = torch.tensor(7)  # present in real checkpoints
    torch.save(wrapped, str(src / "model.pth"))
    torch.save({k: torch.as_tensor(v) for k, v in lsd.items()}, str(src / "lm.pth"))
    _vocab64_yttm_model(str(src / "tokenizer.yttm-model"))
    work = tmp_path / "work"
    work.mkdir()
    with tarfile.open(str(work / "libreasr-model-en.tar.gz"), "w:gz") as tar:
        for name in ("model.pth", "tokenizer.yttm-model", "lm.pth"):
            tar.add(str(src / name), arcname=f"en/{name}")
    monkeypatch.chdir(work)
    asr = LibreASR.load("en", config_path="/nonexistent.yaml", max_streams=8)
    try:
        assert isinstance(asr.lang, TokenizedLanguage) and len(asr.lang) == 64
        assert asr.engine.lm_cfg is not None                                        # ./tmp/en/lm.pth was found and attached
        assert os.path.exists(work / "tmp" / "en" / "model.pth")
        g = np.load(os.path.join(golden_dir, "model_tiny_soft__tiny_lm.npz"))
        pcm = synth.synth_pcm(3, 48000, seed=1234)
        bpe = yttm.BPE(model=str(src / "tokenizer.yttm-model"))
        ids = asr.transcribe([pcm[0], pcm[1]], return_ids=True)
        for i in range(2):
            assert ids[i] == list(g[f"off_tokens_{i}"])
        texts = asr.transcribe([pcm[0], pcm[1]])
        for i in range(2):
            assert texts[i] == bpe.decode([list(g[f"off_tokens_{i}"])], ignore_ids=[0])[0]
            assert len(texts[i]) > 0 and not texts[i][0].isspace()
        last = None
        for hyp in asr.stream(synth.stream_chunks(pcm[0], 1280, lead=1, tail=10), return_ids=True):
            last = hyp
        assert last == list(g["st_tokens_0"])
    finally:
        asr.engine.close()


def test_load_stuff_serves_the_fp32_lm_when_the_int8_form_does_not_take_it():
    """maybe_quantize swallows a failed quantisation and serves the unquantised LM (utils.py:197-210).  An LM wider
    than the int8 path's exact-accumulation bound (hidden > 1024) must therefore attach as fp32, not raise."""
    import __graft_entry__ as graft
    graft.build()
    from libreasr_amd.lib.inference import load_stuff
    wide = dict(vocab=64, embed=32, hidden=1040, layers=1, emb_scale=4.0)
    conf, lang, model, x_tfm, x_tfm_stream = load_stuff("en", config_path="/nonexistent.yaml", synthetic="tiny", max_streams=16,
                                                        synthetic_lm=wide, lm_int8=True)
    try:
        assert model.engine.lm_cfg is not None and model.engine.lm_cfg["hidden"] == 1040
        pcm = synth.synth_pcm(1, 16000 * 2, seed=4)[0]
        slot = model.engine.open()
        model.engine.transcribe_pcm([slot], [pcm])
        toks, _, _ = model.engine.fetch(slot)
        assert isinstance(toks, list)
    finally:
        model.engine.close()
