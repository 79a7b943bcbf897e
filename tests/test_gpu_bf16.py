"""dtype=bf16 (BASELINE configs[2]/[4] arithmetic): bf16 MFMA operands, f32 accumulate / state / logits.

The reference has no bf16 mode (fp32 only), so the contract is stated by the oracle's operand="bf16"
emulation (oracle/rnnt_oracle.py: which tensors are rounded, nothing else changes) and checked with
tolerances; the fp32 path keeps the bit-exact token bar.  Tolerances here:
  * op level (one rounding step deep): 2e-3 abs on O(1) activations, 2e-2 on logits of scale ~8
  * end to end: the engine's tokens vs the emulation's tokens may differ where a rounding tie falls
    the other way (summation order differs inside f32 accumulation); >= 90 % sequence similarity and
    a hard bound against the fp32 oracle (>= 70 %) so a broken kernel cannot pass."""
import difflib

import numpy as np
import pytest
import torch

from libreasr_amd import synth
from oracle import rnnt_oracle as O

pytestmark = pytest.mark.gpu

_ENGINES = {}


def engine(name, max_streams=16):
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    key = (name, max_streams)
    if key not in _ENGINES:
        graft.build()
        cfg = synth.model_cfg(name)
        sd = synth.synth_state_dict(cfg, seed=0)
        _ENGINES[key] = (Engine(sd, cfg, max_streams=max_streams, dtype="bf16"),
                         O.OracleTransducer(sd, cfg, operand="bf16"), O.OracleTransducer(sd, cfg), cfg)
    return _ENGINES[key]


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def sim(a, b):
    if not a and not b:
        return 1.0
    return difflib.SequenceMatcher(None, list(a), list(b), autojunk=False).ratio()


def test_bf16_round_matches_torch():
    x = np.random.default_rng(0).standard_normal(4096).astype(np.float32) * 37.0
    ref = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(O.bf16_round(x), ref)


@pytest.mark.parametrize("name", ["tiny", "tiny_lstm", "cfg2"])
def test_bf16_ops(name):
    eng, mb, mf, cfg = engine(name)
    H = cfg["hidden"]
    # encoder: 0.5 s of audio (errors compound over time steps; this checks the cell, not chaos)
    pcm = synth.synth_pcm(2, 8000, seed=1234)
    feats = np.stack([O.features_offline(p) for p in pcm])
    out, h, c = eng.encoder(dev(feats), return_state=True)
    ref, st = mb.encoder(feats)
    e = np.abs(out.cpu().numpy() - ref).max()
    assert e < 0.08, f"encoder out vs bf16 emulation: {e}"          # BN scale ~5 on h errors of a few bf16 ulps
    e = np.abs(h.cpu().numpy() - np.stack([a[0] for a in st])).max()
    assert e < 0.02, f"h: {e}"
    # and it is a bf16-level approximation of the fp32 model, not something else
    ref32, _ = mf.encoder(feats)
    assert np.abs(out.cpu().numpy() - ref32).mean() < 0.05
    # predictor
    toks = np.array([[2, 5, 7], [2, 9, 9], [3, 1, 4]], dtype=np.int32)
    hp = eng.predictor(toks).cpu().numpy()
    for r in range(3):
        stp = None
        for t in toks[r]:
            x, stp = mb.predictor([t], stp)
        e = np.abs(hp[r] - x[0]).max()
        assert e < 0.06, f"predictor row {r}: {e}"
    # joint on given inputs: one GEMM -> tanh -> GEMM deep
    rng = np.random.default_rng(0)
    a = rng.standard_normal((5, H)).astype(np.float32)
    b = rng.standard_normal((5, H)).astype(np.float32)
    logits, lp, am = eng.joint(dev(a), dev(b))
    ref_lp, ref_z = mb.joint_logp(a, b)
    e = np.abs(logits.cpu().numpy() - ref_z).max()
    assert e < 5e-2, f"joint logits vs bf16 emulation: {e}"
    z32 = mf.joint_logp(a, b)[1]
    assert np.abs(logits.cpu().numpy() - z32).max() < 0.5           # bf16-level distance to the fp32 model


@pytest.mark.parametrize("name", ["tiny", "cfg2"])
def test_bf16_streaming_tokens(name):
    eng, mb, mf, cfg = engine(name)
    n = 3
    pcm = synth.synth_pcm(n, 16000 * 2, seed=1234)
    slots = [eng.open() for _ in range(n)]
    got = [[] for _ in range(n)]
    chunks = [synth.stream_chunks(pcm[i], 1280, lead=1, tail=4) for i in range(n)]
    for k in range(len(chunks[0])):
        eng.push(slots, np.stack([chunks[i][k] for i in range(n)]))
        if eng.step(slots):
            for i, t in enumerate(eng.fetch_many(slots)):
                got[i] += list(t)
    for s in slots:
        eng.close_slot(s)
    for i in range(n):
        fe, db, df = O.StreamFrontend(), mb.stream_decoder(), mf.stream_decoder()
        for ch in chunks[i]:
            o = fe.push(ch)
            if o is not None:
                db.step(o)
                df.step(o)
        sb, sf = sim(got[i], db.y), sim(got[i], df.y)
        print(f"{name} stream {i}: {len(got[i])} tokens, similarity vs bf16 emulation {sb:.3f}, vs fp32 oracle {sf:.3f}")
        assert len(got[i]) > 0
        assert sb >= 0.9, (got[i], db.y)
        assert sf >= 0.7, (got[i], df.y)


def test_bf16_rejects_odd_dims():
    from libreasr_amd.engine import Engine
    cfg = synth.model_cfg("tiny")
    cfg["hidden"] = 48
    cfg["joint"] = 48
    sd = synth.synth_state_dict(cfg, seed=0)
    Engine(sd, cfg, max_streams=4).close()                           # fine in f32 (multiples of 16)
    with pytest.raises(ValueError):
        Engine(sd, cfg, max_streams=4, dtype="bf16")
