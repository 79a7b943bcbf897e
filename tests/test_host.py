"""CPU tests of the host-side logic: FLAC decoder (config 1 input), YAML config + overrides,
tensorize, language stand-in, synthetic-weight recipe, the multi-GPU sharding/aggregation of bench.py
over gloo (world_size 2)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from libreasr_amd import synth
from libreasr_amd.lib import config as cfgmod
from libreasr_amd.lib.language import IdLanguage
from libreasr_amd.lib.utils import tensorize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FLAC = "/root/reference/demo/3729-6852-0035.flac"
REF_YAML = "/root/reference/config/testing.yaml"


@pytest.mark.skipif(not os.path.exists(REF_FLAC), reason="reference tree not mounted (GPU box)")
def test_flac_decoder_matches_streaminfo_md5(golden_dir):
    from libreasr_amd import flac
    pcm, sr, md5_ok = flac.decode(REF_FLAC)
    assert md5_ok and sr == 16000 and len(pcm) == 330400          # 20.65 s, MD5 93b7bac1...
    g = np.load(os.path.join(golden_dir, "demo_flac.npz"))
    assert np.array_equal(pcm[:4096], g["pcm_head"])
    assert abs(float(pcm.astype(np.float64).sum()) - float(g["pcm_sum"])) < 1e-9
    # the reference's x_tfm on the decoded demo gives 258 stacked frames; the oracle agrees
    from oracle import rnnt_oracle as O
    feats = O.features_offline(pcm)
    assert feats.shape == (int(g["n_frames"]), 1280) == (258, 1280)
    np.testing.assert_allclose(feats[:4], g["feats_first"], atol=3e-4)
    np.testing.assert_allclose(feats[-2:], g["feats_last"], atol=3e-4)


@pytest.mark.skipif(not os.path.exists(REF_YAML), reason="reference tree not mounted (GPU box)")
def test_reference_yaml_and_overrides():
    conf = cfgmod.open_config(REF_YAML)
    inf = cfgmod.apply_overrides(conf, inference=True, lang=None)
    assert inf["cuda"]["enable"] is False and inf["bs"] == 1            # testing.yaml:378-383
    assert cfgmod.stream_settings(inf) == (10, 8, 2)                     # testing.yaml:362-374
    m = cfgmod.model_cfg_from_conf(inf)
    assert m == dict(feat=1280, embed=512, vocab=2048, hidden=1024, joint=1024, enc_layers=6,
                     pred_layers=2, pred_cell="NBRC")                    # testing.yaml:202-229


def test_update_is_recursive():
    d = {"a": {"b": 1, "c": 2}, "x": 1}
    cfgmod.update(d, {"a": {"b": 5}, "y": 2})
    assert d == {"a": {"b": 5, "c": 2}, "x": 1, "y": 2}


def test_tensorize_and_language():
    x = np.arange(5, dtype=np.float32)
    t = tensorize(x.tobytes())
    assert tuple(t.shape) == (1, 5) and np.array_equal(t.numpy()[0], x)
    assert IdLanguage().denumericalize([0, 5, 0, 7]) == "5 7"


def test_synth_recipe_is_deterministic_and_shaped():
    cfg = synth.model_cfg("tiny")
    a, b = synth.synth_state_dict(cfg, seed=0), synth.synth_state_dict(cfg, seed=0)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert np.all(a["predictor.embed.weight"][0] == 0)                   # padding_idx row
    assert np.all(a["joint.joint.2.weight"][0] == 0)                     # constant blank logit
    p = synth.synth_pcm(2, 5000, seed=3)
    assert p.shape == (2, 5000) and p.dtype == np.float32 and np.abs(p).max() <= 1.0
    ch = synth.stream_chunks(p[0], 1280, lead=1, tail=10)
    assert len(ch) == 1 + 5000 // 1280 + 10 and not ch[0].any() and not ch[-1].any()   # api-client.py:32-47


def test_bench_multi_gpu_path_over_gloo():
    """world_size 2 on CPU: contiguous stream sharding, max-over-ranks time, sum-over-ranks units."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--selftest-dist"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert line, out.stdout + out.stderr
    d = json.loads(line[-1])
    assert d["world"] == 2 and d["elapsed_max"] == 1.5 and d["units_total"] == 128.0 and d["n_local"] == 64
    import bench
    assert bench.shard_streams(512, 8, 3) == list(range(192, 256))


def test_model_archive_round_trip(tmp_path):
    """libreasr-model-*.tar.gz (model_utils.py:31-58): <lang>/model.pth in fastai learn.save form + tokenizer;
    extraction, unwrapping and flattening give the blob lasr_create expects; path traversal is refused."""
    import io
    import tarfile
    import torch
    from libreasr_amd import synth
    from libreasr_amd.lib import model_utils as mu
    from libreasr_amd.weights import flatten_lm_state_dict, flatten_state_dict, infer_cfg
    cfg = synth.model_cfg("tiny")
    sd = {k: torch.as_tensor(v) for k, v in synth.synth_state_dict(cfg, seed=0).items()}
    lm = {k: torch.as_tensor(v) for k, v in synth.synth_lm_state_dict("tiny_lm").items()}
    src = tmp_path / "src" / "en"
    src.mkdir(parents=True)
    torch.save({"model": sd, "opt": {"state": []}}, src / "model.pth")
    torch.save(lm, src / "lm.pth")
    (src / "tokenizer.yttm-model").write_bytes(b"stub")
    arc = tmp_path / "libreasr-model-en.tar.gz"
    with tarfile.open(arc, "w:gz") as tar:
        for f in ("model.pth", "lm.pth", "tokenizer.yttm-model"):
            tar.add(src / f, arcname=f"en/{f}")
    dest = tmp_path / "tmp"
    names = mu.extract_tars([str(arc)], dest)
    assert "en/model.pth" in names
    p = mu.model_paths("en", dest)
    got = mu.load_model_state_dict(p["model"])
    assert set(got) == set(sd) and infer_cfg(got) == {k: cfg[k] for k in infer_cfg(got)}
    blob = flatten_state_dict(got, cfg)
    assert blob.dtype == np.float32 and blob.size == flatten_state_dict(synth.synth_state_dict(cfg, seed=0), cfg).size
    lcfg, lblob = flatten_lm_state_dict(mu.load_lm_state_dict(p["lm"]))
    assert lcfg == dict(vocab=64, embed=32, hidden=32, layers=2) and lblob.size == 64 * 32 + 2 * (4 * 32 * 32 * 2 + 8 * 32) + 64 * 32 + 64
    evil = tmp_path / "evil.tar.gz"
    with tarfile.open(evil, "w:gz") as tar:
        info = tarfile.TarInfo("../escape.txt")
        info.size = 1
        tar.addfile(info, io.BytesIO(b"x"))
    with pytest.raises(ValueError):
        mu.extract_tars([str(evil)], dest)
