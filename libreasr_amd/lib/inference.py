"""Mirror of libreasr/lib/inference.py: load_stuff(lang) -> (conf, lang, model, x_tfm, x_tfm_stream)
(inference.py:18-51; used by ASRServicer.__init__, api-server.py:54-62).

Weights: a plain `state_dict` with the reference's key names from `conf["model"]["path"]`
(torch.load; a fastai {"model": ...} wrapper is unwrapped) or, when `synthetic=` names a shape in
libreasr_amd.synth.CONFIGS, seeded synthetic weights (no pretrained model ships with the reference)."""
import os

import torch

from .. import synth
from ..engine import Engine
from ..weights import infer_cfg
from .config import apply_overrides, engine_settings, model_cfg_from_conf, open_config, stream_settings
from .language import get_language
from .model_utils import extract_tars, load_lm_state_dict as _load_lm_sd, load_model_state_dict
from .models import Transducer
from .transforms import AudioTensor, OfflinePipeline, StreamPipeline  # noqa: F401


def load_state_dict(path):
    return load_model_state_dict(path)      # plain or fastai learn.save format (model_utils.py:79-85)


def load_lm_state_dict(conf, synthetic_lm=None):
    """config.py:140-147 + lm.py:86-100: the LM is loaded when `lm.enable` and `lm.path` resolve; a
    failure to load is not fatal in the reference ("[LM] Failed to load.").  fp32 (no int8 quantisation)."""
    if synthetic_lm is not None:
        return synth.synth_lm_state_dict(synthetic_lm)
    lm = (conf.get("lm", {}) or {}) if conf else {}
    if not lm.get("enable") or not lm.get("path") or not os.path.exists(lm["path"]):
        return None
    if not lm.get("path", "").endswith(".pth"):
        return None
    try:
        return _load_lm_sd(lm["path"])
    except Exception:
        print("[LM] Failed to load.")
        return None


def load_stuff(lang, config_path="./config/testing.yaml", synthetic=None, max_streams=None, device=None,
               synthetic_lm=None, dtype=None, beam=None, lm_int8=None):
    """max_streams / device / dtype / beam / lm_int8: None = the YAML's `engine:` section, else its defaults (16, 0, "f32", 1, True:
    config.engine_settings)."""
    torch.set_num_threads(2)                # inference.py:21
    conf, cfg, sd = {}, None, None
    if os.path.exists(config_path):
        conf = apply_overrides(open_config(config_path), inference=True, lang=lang)
    es = engine_settings(conf, max_streams=max_streams, device=device, dtype=dtype, beam=beam, lm_int8=lm_int8)
    max_streams, device, dtype, beam, lm_int8 = es["max_streams"], es["device"], es["dtype"], es["beam"], es["lm_int8"]
    if synthetic is not None:
        cfg = synth.model_cfg(synthetic)
        sd = synth.synth_state_dict(cfg, seed=0)
    else:
        path = ((conf.get("model", {}) or {}).get("path")) or f"./tmp/{lang}/model.pth"
        if not os.path.exists(path):
            extract_tars()                  # libreasr-model-*.tar.gz in the working directory (model_utils.py:50-58)
        sd = load_state_dict(path)
        cfg = model_cfg_from_conf(conf) if "model" in conf else infer_cfg(sd)
    n_stack, downsample, n_buffer = stream_settings(conf) if conf else (10, 8, 2)
    eng = Engine(sd, cfg, max_streams=max_streams, device=device, n_stack=n_stack, stride=downsample,
                 n_buffer=n_buffer, dtype=dtype, beam=beam)
    lm_sd = load_lm_state_dict(conf, synthetic_lm)
    if lm_sd is None and synthetic is None and os.path.exists(f"./tmp/{lang}/lm.pth") and not conf:
        try:                                 # no config file: the archive layout decides (testing.yaml:310-311 -> ./tmp/<lang>/lm.pth)
            lm_sd = _load_lm_sd(f"./tmp/{lang}/lm.pth")
        except Exception:
            print("[LM] Failed to load.")
    if lm_sd is not None:
        if beam > 1:
            lm_int8 = False                 # the beam takes the fp32 / bf16 LM (per-hypothesis LM state); int8 form: greedy only
        # lm_int8: as load_lm serves it (maybe_quantize, lm.py:97); False: fp32 / bf16 LM.  maybe_quantize swallows a failed
        # quantisation and serves the fp32 LM (utils.py:197-210): an LM shape the int8 path does not take does the same here
        from .._native import LASR_EINVAL, LasrError
        try:
            eng.attach_lm(lm_sd, int8=lm_int8)
        except LasrError as e:
            if not (lm_int8 and e.code == LASR_EINVAL):
                raise
            print(f"[quantization] failed ({e}); serving the unquantised LM")
            eng.attach_lm(lm_sd, int8=False)
        print("[LM] loaded.")
    # tokenizer: the configured file (testing.yaml:153-154, per-language override :320-321), else where the model archive
    # puts it (model_utils.py:31-47: <lang>/tokenizer.yttm-model under ./tmp)
    tok = ((conf.get("tokenizer", {}) or {}).get("model_file")) if conf else None
    if not (tok and os.path.exists(tok)):
        tok = f"./tmp/{lang}/tokenizer.yttm-model"
    language = get_language(tok if os.path.exists(tok) else None)
    model = Transducer(eng, language)
    sr = conf.get("sr", 16000) if conf else 16000
    ch = conf.get("channels", 1) if conf else 1
    x_tfm = OfflinePipeline(eng, channels=ch, target_sr=sr)
    x_tfm_stream = StreamPipeline(eng, n_stack=n_stack, n_buffer=n_buffer, channels=ch, target_sr=sr)
    print("[Inference] Model and Pipeline set up.")
    return conf, language, model, x_tfm, x_tfm_stream
