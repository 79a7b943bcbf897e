"""
state_dict -> flat float32 blob in the order lasr_create expects (include/lasr.h, lasr_weight_count).

The input contract is the reference's `Transducer.state_dict()` key layout (SURVEY.md §8a row W1;
libreasr/lib/models.py:190-234, libreasr/lib/layers/custom_rnn.py:113-126,265-269).  The
MFMA-fragment packing itself happens inside liblasr_hip.so (it is a kernel implementation detail).
"""
import numpy as np


def _np(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def infer_cfg(sd):
    """Model shape from a reference state_dict."""
    keys = list(sd.keys())
    enc_layers = 1 + max(int(k.split(".")[3]) for k in keys if k.startswith("encoder.rnn_stack.rnns."))
    pred_layers = 1 + max(int(k.split(".")[3]) for k in keys if k.startswith("predictor.rnn_stack.rnns."))
    pred_cell = "LSTM" if "predictor.rnn_stack.rnns.0.weight_ih_l0" in sd else "NBRC"
    V, E = _np(sd["predictor.embed.weight"]).shape
    H = _np(sd["encoder.rnn_stack.rnns.0.weight_hh_l0"]).shape[1]
    F = _np(sd["encoder.input_norm.weight"]).shape[0]
    J = _np(sd["joint.joint.0.weight"]).shape[0]
    return dict(feat=F, embed=E, vocab=V, hidden=H, joint=J, enc_layers=enc_layers,
                pred_layers=pred_layers, pred_cell=pred_cell)


def flatten_state_dict(sd, cfg):
    """Concatenate the tensors in blob order; validates every shape against `cfg`."""
    F, E, V, H, J = cfg["feat"], cfg["embed"], cfg["vocab"], cfg["hidden"], cfg["joint"]
    parts = []

    def take(key, shape):
        if key not in sd:
            raise KeyError(f"state_dict is missing '{key}'")
        a = _np(sd[key]).reshape(-1) if shape is None else _np(sd[key])
        if shape is not None and tuple(a.shape) != tuple(shape):
            raise ValueError(f"'{key}' has shape {a.shape}, expected {shape}")
        parts.append(a.reshape(-1))

    def bn(prefix):
        for n in ("weight", "bias", "running_mean", "running_var"):
            take(f"{prefix}.{n}", (H,))

    def lstm(prefix, I):
        take(f"{prefix}.weight_ih_l0", (4 * H, I))
        take(f"{prefix}.weight_hh_l0", (4 * H, H))
        take(f"{prefix}.bias_ih_l0", (4 * H,))
        take(f"{prefix}.bias_hh_l0", (4 * H,))

    take("encoder.input_norm.weight", (F,))
    take("encoder.input_norm.bias", (F,))
    for i in range(cfg["enc_layers"]):
        take(f"encoder.rnn_stack.hs.{i}", (2, 1, 1, H))
        bn(f"encoder.rnn_stack.bns.{i}")
        lstm(f"encoder.rnn_stack.rnns.{i}", F if i == 0 else H)
    take("predictor.embed.weight", (V, E))
    if E != H:
        take("predictor.ffn.weight", (H, E))
        take("predictor.ffn.bias", (H,))
    is_lstm = cfg["pred_cell"] == "LSTM"
    for i in range(cfg["pred_layers"]):
        take(f"predictor.rnn_stack.hs.{i}", (2 if is_lstm else 1, 1, 1, H))
        bn(f"predictor.rnn_stack.bns.{i}")
        p = f"predictor.rnn_stack.rnns.{i}"
        if is_lstm:
            lstm(p, H)
        else:
            take(f"{p}.kernel", (H, 3 * H))
            take(f"{p}.recurrent_kernel", (H, 3 * H))
            take(f"{p}.bias", (3 * H,))
            take(f"{p}.recurrent_bias", (3 * H,))
    take("joint.joint.0.weight", (J, 2 * H))
    take("joint.joint.0.bias", (J,))
    take("joint.joint.2.weight", (V, J))
    take("joint.joint.2.bias", (V,))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


def flatten_lm_state_dict(sd):
    """LM (lm.py:20-40) state_dict -> (cfg, float32 blob) in the order lasr_attach_lm expects.  Accepts
    torch tensors or arrays; `linear.weight` may be absent when it is tied to `embed.weight`."""
    emb = _np(sd["embed.weight"])
    V, E = emb.shape
    parts = [emb.reshape(-1)]
    l = 0
    H = None
    while f"rnn.weight_ih_l{l}" in sd:
        w_ih, w_hh = _np(sd[f"rnn.weight_ih_l{l}"]), _np(sd[f"rnn.weight_hh_l{l}"])
        H = w_hh.shape[1]
        assert w_ih.shape == (4 * H, E if l == 0 else H) and w_hh.shape == (4 * H, H)
        parts += [w_ih.reshape(-1), w_hh.reshape(-1), _np(sd[f"rnn.bias_ih_l{l}"]).reshape(-1),
                  _np(sd[f"rnn.bias_hh_l{l}"]).reshape(-1)]
        l += 1
    assert l >= 1, "no rnn.weight_ih_l0 in the LM state_dict"
    w = _np(sd["linear.weight"]) if "linear.weight" in sd else emb
    assert w.shape == (V, H)
    parts += [w.reshape(-1), _np(sd["linear.bias"]).reshape(-1)]
    cfg = dict(vocab=int(V), embed=int(E), hidden=int(H), layers=l)
    return cfg, np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)
