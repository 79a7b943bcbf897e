"""
TEST INFRASTRUCTURE ONLY -- never imported by the product path (libreasr_amd/).

Imports the *real* reference implementation read-only from /root/reference so that
golden vectors can be generated with the reference's own code (oracle/make_golden.py)
and so that the numpy restatement (oracle/rnnt_oracle.py) can be pinned against it.

/root/reference does not exist on the GPU box; everything here is only usable in the
authoring container.  `available()` says whether the reference tree is present.

The reference depends on packages that are not installed here (IPython, fastai2,
fastcore, fastai2_audio, torchaudio).  They are replaced by minimal stubs:

  * fastai2.torch_core.Module  -- nn.Module whose metaclass runs nn.Module.__init__
    before the subclass __init__ (fastai's PrePostInitMeta; models.py:28,68,116,143,190
    rely on it).
  * fastcore Transform / Pipeline -- `__call__` -> `encodes`; Pipeline applies the
    transforms sorted by `.order` (transforms.py:123,136,270,328,430,445,457).
  * torchaudio.transforms.MelSpectrogram -- restated for torchaudio==0.6.0
    (docker/requirements.inference.txt:5): torch.stft(center, reflect, periodic hann
    zero-padded to n_fft, onesided) -> |.|^2 -> HTK mel filterbank (no norm).
    This one numeric dependency is NOT in the reference tree: "parity unpinned" for it
    beyond the published torchaudio 0.6.0 algorithm.
  * torchaudio.transforms.Resample -- identity for equal rates (only 16 kHz is in scope).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "libreasr", "lib"))


def _mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        sys.modules[name] = m
    m.__dict__.update(attrs)
    return m


class _PrePostInit(type):
    def __call__(cls, *a, **k):
        o = cls.__new__(cls)
        nn.Module.__init__(o)
        o.__init__(*a, **k)
        return o


class Module(nn.Module, metaclass=_PrePostInit):
    def __init__(self):
        pass


class _TfmMeta(type):
    pass


class Transform(metaclass=_TfmMeta):
    order = 0

    def __init__(self, *a, **k):
        pass

    def __call__(self, x, **k):
        return self.encodes(x)


class Pipeline:
    def __init__(self, tfms):
        self.fs = sorted(tfms, key=lambda t: getattr(t, "order", 0))

    def __call__(self, x):
        for f in self.fs:
            x = f(x)
            if x is None:
                return None
        return x


class AudioTensor(torch.Tensor):
    @staticmethod
    def __new__(cls, x, sr=16000):
        t = torch.as_tensor(x).as_subclass(cls)
        t.sr = sr
        return t

    def __init__(self, x, sr=16000):
        self.sr = sr

    @property
    def data(self):  # TransformTime reads `ai.data`
        return self.as_subclass(torch.Tensor)


def htk_fb(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio 0.6.0 functional.create_fb_matrix (HTK, no norm)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * np.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * np.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    zero = torch.zeros(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(zero, torch.min(down, up))


class MelSpectrogram(nn.Module):
    def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None,
                 f_min=0.0, f_max=None, pad=0, n_mels=128, power=2.0, **kw):
        super().__init__()
        self.n_fft = n_fft
        self.wl = win_length if win_length is not None else n_fft
        self.hl = hop_length if hop_length is not None else self.wl // 2
        self.power = power
        self.register_buffer("window", torch.hann_window(self.wl))
        f_max = float(sample_rate // 2) if f_max is None else f_max
        self.register_buffer("fb", htk_fb(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate))

    def forward(self, x):
        S = torch.stft(x, self.n_fft, self.hl, self.wl, self.window, center=True,
                       pad_mode="reflect", normalized=False, onesided=True,
                       return_complex=True)
        P = S.abs().pow(self.power)  # [C, F, T]
        return torch.matmul(P.transpose(1, 2), self.fb).transpose(1, 2)  # [C, n_mels, T]


class _Resample(nn.Module):
    def __init__(self, orig_freq=16000, new_freq=16000):
        super().__init__()
        assert orig_freq == new_freq, "only 16 kHz -> 16 kHz (identity) is in scope"

    def forward(self, x):
        return x


class _ComputeDeltas(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


_installed = False


def install_stubs():
    global _installed
    if _installed:
        return
    _installed = True
    _mod("IPython"); _mod("IPython.core")
    _mod("IPython.core.debugger", set_trace=lambda *a, **k: None)
    for name in ["fastai2", "fastai2.vision", "fastai2.vision.models", "fastai2.data",
                 "fastai2.data.all", "fastai2.optimizer", "fastai2.metrics", "fastai2.text",
                 "fastai2.text.core", "fastai2.text.data", "fastai2.text.models",
                 "fastai2.text.models.core", "fastai2.text.models.awdlstm",
                 "fastai2.text.learner", "fastai2.callback", "fastai2.callback.rnn",
                 "fastai2.callback.all", "fastai2.vision.learner", "fastcore",
                 "fastai2_audio", "fastai2_audio.core", "fastai2_audio.augment"]:
        _mod(name)
    _mod("fastai2.vision.models.xresnet", xresnet18=None)
    _mod("fastai2.layers", Debugger=None, ResBlock=None)
    _mod("fastai2.torch_core", Module=Module)
    _mod("fastai2.learner", CancelBatchException=Exception)
    import random

    class Text(str):
        pass

    _mod("fastai2.torch_basics", Transform=Transform, AudioTensor=AudioTensor, Text=Text,
         Tensor=torch.Tensor, FloatTensor=torch.FloatTensor, torch=torch, random=random,
         set_trace=lambda *a, **k: None, math=__import__("math"), np=np)
    _mod("fastcore.transform", _TfmMeta=_TfmMeta, Pipeline=Pipeline, Transform=Transform)
    _mod("fastai2_audio.core.signal", AudioTensor=AudioTensor)
    _mod("fastai2_audio.core.all", AudioTensor=AudioTensor)
    _mod("fastai2_audio.augment.signal", SignalShifter=None, AddNoise=None)
    ta = _mod("torchaudio", set_audio_backend=lambda *a, **k: None)
    ta.transforms = _mod("torchaudio.transforms", MelSpectrogram=MelSpectrogram,
                         ComputeDeltas=_ComputeDeltas, Resample=_Resample)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


class IdLang:
    """Stand-in for TokenizedLanguage: parity is on token ids (no tokenizer model ships)."""

    def denumericalize(self, ids):
        return " ".join(str(int(i)) for i in ids if int(i) != 0)


def ref_lm_int8(cfg, state_dict):
    """The LM as load_lm serves it (lm.py:86-100): the reference's own maybe_quantize (utils.py:197-210 =
    torch.quantization.quantize_dynamic({LSTM, Linear}, qint8)) applied to its LM class, on the installed torch."""
    import warnings
    lm = ref_lm(cfg, state_dict)
    from libreasr.lib.utils import maybe_quantize
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        q = maybe_quantize(lm, debug=False)
    assert "dynamic" in type(q.rnn).__module__, "quantize_dynamic did not convert the LSTM"
    return q.eval()


def ref_lm(cfg, state_dict):
    """The reference's LM class (lm.py:20-40), fp32, eval.  NOT quantised: load_lm (lm.py:86-100) runs
    torch.quantization.quantize_dynamic(int8) on it, whose numerics are un-vendored (fbgemm) -- the
    fusion arithmetic (LMFuser, lm.py:43-83) is what the goldens pin."""
    install_stubs()
    from libreasr.lib.lm import LM
    lm = LM(cfg["vocab"], cfg["embed"], cfg["hidden"], cfg["layers"], p=0.2)
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()}
    lm.load_state_dict(sd, strict=True)
    return lm.eval()


def ref_transducer(cfg, state_dict=None):
    """Instantiate the reference Transducer for an oracle `cfg` dict (see synth.model_cfg)."""
    install_stubs()
    from libreasr.lib.models import Transducer  # /root/reference/libreasr/lib/models.py:190

    enc = dict(rnn_type="LSTM", num_layers=cfg["enc_layers"], dropout=0.05, layer_norm=False,
               use_tmp_state_pcent=0.99)
    pred = dict(rnn_type=cfg["pred_cell"], num_layers=cfg["pred_layers"], dropout=0.05,
                layer_norm=False, use_tmp_state_pcent=0.99)
    m = Transducer(cfg["feat"], cfg["embed"], cfg["vocab"], cfg["hidden"], cfg["hidden"],
                   cfg["joint"], IdLang(), joint_method="concat",
                   encoder_kwargs=enc, predictor_kwargs=pred)
    if state_dict is not None:
        sd = {k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()}
        missing, unexpected = m.load_state_dict(sd, strict=False)
        # only BatchNorm's num_batches_tracked may be absent from a synthetic dict
        assert all(k.endswith("num_batches_tracked") for k in missing), missing
        assert not unexpected, unexpected
    return m.eval()


def ref_transforms(n_stack=10, downsample=8, n_buffer=2):
    """The two reference inference Pipelines (config/testing.yaml:339-374)."""
    install_stubs()
    import libreasr.lib.transforms as T  # /root/reference/libreasr/lib/transforms.py

    kw = dict(channels=1, target_sr=16000, sr=16000, win_length=0.025, hop_length=0.01,
              deltas=0, delta_win_length=2, mfcc_args={}, melkwargs=dict(n_fft=1024, n_mels=128),
              use_extra_features=False, random=False)
    x = [T.Resample(**kw), T.ChannelCut(**kw), T.TransformTime(**kw),
         T.StackDownsample(n_stack=n_stack, downsample=downsample), T.FixDimensions()]
    s = [T.Resample(**kw), T.ChannelCut(**kw), T.TransformTime(**kw),
         T.StreamPostprocess(n_stack=n_stack),
         T.StackDownsample(n_stack=n_stack, downsample=downsample), T.FixDimensions(),
         T.Buffer(n_buffer=n_buffer)]
    return Pipeline(x), Pipeline(s), AudioTensor


class _Msg:
    """Stand-in for the generated protobuf messages (libreasr.proto:9-17: Audio{data, sr}, Transcript{data}) when the reference's
    pre-3.20 generated module cannot be imported by the installed protobuf: transport, not servicer logic."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def ref_servicer(model, s_tfm, x_tfm=None, lang=None, downsample=8, n_buffer=2):
    """The reference's OWN gRPC servicer class (/root/reference/api-server.py:53-135) around an already built reference
    Transducer and the reference's stream Pipeline -- `load_stuff` (weights / tokenizer downloads, config parsing) is bypassed by
    constructing the object without __init__; TranscribeStream / Transcribe / should_reset are the reference's code, unmodified.
    Returns (servicer, module, resets): `resets` collects the value of `steps` every time the module's should_reset says True."""
    install_stubs()
    import importlib.util
    try:
        import interfaces.libreasr_pb2 as ap            # the reference's generated module (old protoc output)
        import interfaces.libreasr_pb2_grpc  # noqa: F401
    except Exception:
        for k in [k for k in sys.modules if k == "interfaces" or k.startswith("interfaces.")]:
            del sys.modules[k]
        ap = _mod("interfaces.libreasr_pb2", Audio=_Msg, Transcript=_Msg)
        _mod("interfaces", libreasr_pb2=ap)
        _mod("interfaces.libreasr_pb2_grpc", ASRServicer=object, add_ASRServicer_to_server=lambda *a, **k: None)
    # `from libreasr.lib.inference import *` (api-server.py:11) only has to provide torch and AudioTensor to the servicer's
    # methods; the real module pulls in the training stack's imports (config.py -> data.py -> fastai2)
    _mod("libreasr.lib.inference", torch=torch, AudioTensor=AudioTensor, load_stuff=None, __all__=["torch", "AudioTensor", "load_stuff"])
    _mod("matplotlib"); _mod("matplotlib.pyplot")
    spec = importlib.util.spec_from_file_location("ref_api_server", os.path.join(REF_ROOT, "api-server.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    resets = []
    orig = mod.should_reset

    def should_reset(steps, ds, nb):       # the reference's rule, observed
        r = orig(steps, ds, nb)
        if r:
            resets.append(int(steps))
        return r

    mod.should_reset = should_reset
    mod.log_print = lambda *a, **k: None
    sv = object.__new__(mod.ASRServicer)
    sv.lang_name = "en"; sv.conf = None; sv.downsample = downsample; sv.n_buffer = n_buffer
    sv.lang = lang if lang is not None else IdLang(); sv.model = model; sv.x_tfm = x_tfm; sv.x_tfm_stream = s_tfm
    return sv, mod, resets
