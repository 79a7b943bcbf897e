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


def model_cfg_from_conf(conf):
    m = conf["model"]
    return dict(feat=m["feature_sz"], embed=m["embed_sz"], vocab=m["vocab_sz"], hidden=m["hidden_sz"],
                joint=m["joint_sz"], enc_layers=m["encoder"]["num_layers"],
                pred_layers=m["predictor"]["num_layers"], pred_cell=m["predictor"]["rnn_type"])
