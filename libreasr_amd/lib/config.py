"""Mirror of libreasr/lib/config.py (inference subset): YAML + recursive `overrides` merge.

open_config / update follow config.py:23-42; parse_and_apply_config(inference=True, lang=...) applies
`overrides.inference` and then `overrides.<lang>` (config.py:102-110).  Training wiring is out of scope."""
import collections.abc
import copy

import yaml


def update(d, u):
    """Recursive dict merge (config.py:23-30)."""
    for k, v in u.items():
        if isinstance(v, collections.abc.Mapping):
            d[k] = update(d.get(k, {}) or {}, v)
        else:
            d[k] = v
    return d


def open_config(path="./config/testing.yaml"):
    with open(path, "r") as f:
        return yaml.safe_load(f)


def apply_overrides(conf, inference=True, lang=None):
    conf = copy.deepcopy(conf)
    ov = conf.get("overrides", {}) or {}
    if inference and "inference" in ov:
        update(conf, ov["inference"])
    if lang is not None and lang in (ov.get("languages", {}) or {}):
        update(conf, ov["languages"][lang])
    elif lang is not None and lang in ov:
        update(conf, ov[lang])
    return conf


def stream_settings(conf):
    """(n_stack, downsample, n_buffer) from the `stream` transform list (api-server.py:33-41)."""
    n_stack, downsample, n_buffer = 10, 8, 2
    for tfm in (conf.get("transforms", {}) or {}).get("stream", []) or []:
        args = tfm.get("args", {}) or {}
        if tfm.get("name") == "StackDownsample":
            n_stack, downsample = args.get("n_stack", n_stack), args.get("downsample", downsample)
        if tfm.get("name") == "Buffer":
            n_buffer = args.get("n_buffer", n_buffer)
    return n_stack, downsample, n_buffer


ENGINE_DEFAULTS = dict(max_streams=16, device=0, dtype="f32", beam=1, lm_int8=True, depth=12, front="python")


def engine_settings(conf, **explicit):
    """The `engine:` section of the YAML (no counterpart in the reference, whose model runs wherever torch puts it: SURVEY §5,
    new-build stance): how this build serves the model -- max_streams (resident stream slots = rows of every GEMM), device,
    dtype ("f32" | "bf16" operands), beam (1 = the reference's greedy decode), lm_int8 (serve the LM as the reference does:
    dynamically quantised), depth (model steps the serving front keeps in flight) and front ("python" | "native").  Per-language
    `overrides` apply as everywhere else.  Arguments passed explicitly (not None) win over the file, the file over the defaults."""
    out = dict(ENGINE_DEFAULTS)
    sec = (conf.get("engine", {}) or {}) if conf else {}
    unknown = set(sec) - set(ENGINE_DEFAULTS)
    if unknown:
        raise ValueError(f"engine: unknown keys {sorted(unknown)} (known: {sorted(ENGINE_DEFAULTS)})")
    out.update(sec)
    out.update({k: v for k, v in explicit.items() if v is not None})
    if out["dtype"] not in ("f32", "bf16"):
        raise ValueError(f"engine.dtype must be f32 or bf16, not {out['dtype']!r}")
    if out["front"] not in ("python", "native"):
        raise ValueError(f"engine.front must be python or native, not {out['front']!r}")
    for k in ("max_streams", "beam", "depth"):
        if int(out[k]) < 1:
            raise ValueError(f"engine.{k} must be >= 1")
    return out


def model_cfg_from_conf(conf):
    m = conf["model"]
    return dict(feat=m["feature_sz"], embed=m["embed_sz"], vocab=m["vocab_sz"], hidden=m["hidden_sz"],
                joint=m["joint_sz"], enc_layers=m["encoder"]["num_layers"],
                pred_layers=m["predictor"]["num_layers"], pred_cell=m["predictor"]["rnn_type"])
