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


def test_engine_section_of_the_yaml(tmp_path):
    """`engine:` (new-build stance): defaults < file < per-language override < explicit argument; unknown keys and
    bad values are refused before anything is loaded."""
    import yaml
    conf = {"engine": {"max_streams": 64, "dtype": "bf16", "depth": 18},
            "overrides": {"inference": {"engine": {"front": "native"}}, "languages": {"de": {"engine": {"beam": 4, "max_streams": 32}}}}}
    p = tmp_path / "c.yaml"
    p.write_text(yaml.safe_dump(conf))
    en = cfgmod.engine_settings(cfgmod.apply_overrides(cfgmod.open_config(str(p)), inference=True, lang="en"))
    assert en == dict(max_streams=64, device=0, dtype="bf16", beam=1, lm_int8=True, depth=18, front="native")
    de = cfgmod.engine_settings(cfgmod.apply_overrides(cfgmod.open_config(str(p)), inference=True, lang="de"), dtype="f32", beam=None)
    assert (de["max_streams"], de["beam"], de["dtype"], de["front"]) == (32, 4, "f32", "native")
    assert cfgmod.engine_settings({}) == cfgmod.ENGINE_DEFAULTS and cfgmod.engine_settings(None, beam=8)["beam"] == 8
    for bad in ({"engine": {"streams": 4}}, {"engine": {"dtype": "fp16"}}, {"engine": {"front": "rust"}}, {"engine": {"beam": 0}}):
        with pytest.raises(ValueError):
            cfgmod.engine_settings(bad)


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
    """world_size 2 on CPU: contiguous stream sharding, max-over-ranks time, sum-over-ranks units -- once under an
    external torch.distributed.run (how the driver launches N > 1) and once from plain `python bench.py --gpus 2`
    (bench.py spawns its own ranks)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmds = [[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
             "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
             "--gpus", "2", "--selftest-dist"],
            [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-dist"]]
    for cmd in cmds:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert line, out.stdout + out.stderr
        d = json.loads(line[-1])
        assert d["world"] == 2 and d["elapsed_max"] == 1.5 and d["units_total"] == 128.0 and d["n_local"] == 64
    import bench
    assert bench.shard_streams(512, 8, 3) == list(range(192, 256))


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` on a box with fewer than 2 GPUs must fail (rc != 0), not report n_gpus: 1."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LASR_BENCH_SAME_GPU"):
        env.pop(k, None)
    env["HIP_VISIBLE_DEVICES"] = ""            # no device visible, whatever the box has
    env["CUDA_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode != 0 and "GPU(s) visible" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_cpu_legs_and_labels():
    """The CPU-baseline legs own their PCM (the round-1 crash: slicing past the GPU run's buffer) and the
    workload label follows the actual model / dtype / beam / streams."""
    import argparse
    import bench
    from libreasr_amd import synth
    cfg = synth.model_cfg("tiny")
    sd = synth.synth_state_dict(cfg, seed=0)
    r = bench.cpu_reference_path(cfg, sd, 2, 30)
    assert r["value"] > 0 and r["cores"] == 2 and r["kind"] == "port" and r["tokens"] > 0
    n = bench.cpu_numpy_port(cfg, sd, 2, 30)
    assert n["value"] > 0
    ns = argparse.Namespace
    assert bench.workload_name(ns(model="cfg2", dtype="f32", beam=1), synth.model_cfg("cfg2"), 64).startswith("configs[1]: 64 ")
    w = bench.workload_name(ns(model="cfg5", dtype="bf16", beam=8), synth.model_cfg("cfg5"), 128)
    assert w.startswith("configs[4]: 128 ") and "8x1536" in w and "2xLSTM" in w and "beam width 8" in w
    assert "variant" in bench.workload_name(ns(model="cfg5", dtype="bf16", beam=1), synth.model_cfg("cfg5"), 128)
    assert abs(bench.flop_per_frame(synth.model_cfg("cfg2"), 0.3) - 85.25e6) < 0.01e6


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


def _tiny_yttm_model(path):
    """A hand-made YouTokenToMe model file (format of BPEState::dump): pad 0, unk 1, bos 2, eos 3, then the characters
    "▁ a b c ä" as ids 4..8, then merges in priority order."""
    chars = {0x2581: 4, ord("a"): 5, ord("b"): 6, ord("c"): 7, ord("ä"): 8}
    rules = [(5, 6, 9),      # a b   -> ab
             (4, 9, 10),     # ▁ ab  -> ▁ab
             (7, 7, 11),     # c c   -> cc
             (4, 7, 12),     # ▁ c   -> ▁c
             (10, 11, 13)]   # ▁ab cc -> ▁abcc
    with open(path, "w", encoding="utf-8") as f:
        f.write(f"{len(chars)} {len(rules)}\n")
        for cp, i in chars.items():
            f.write(f"{cp} {i}\n")
        for x, y, z in rules:
            f.write(f"{x} {y} {z}\n")
        f.write("1 0 2 3\n")


def test_yttm_model_reader_decodes_like_the_reference_call(tmp_path):
    """language.py:135-142: tokenizer.decode([ids], ignore_ids=[0])[0] on a YTTM model (restated reader, lib/yttm.py)."""
    from libreasr_amd.lib import yttm
    from libreasr_amd.lib.language import TokenizedLanguage, get_language
    mf = str(tmp_path / "tokenizer.yttm-model")
    _tiny_yttm_model(mf)
    bpe = yttm.BPE(model=mf)
    assert bpe.vocab_size() == 14
    assert bpe.vocab()[:4] == ["<PAD>", "<UNK>", "<BOS>", "<EOS>"] and bpe.vocab()[13] == "▁abcc" and bpe.vocab()[8] == "ä"
    # word-start marker -> space, leading space of the sentence dropped, blank (0) ignored, specials printed by name
    assert bpe.decode([[13, 0, 12, 0, 0, 10]], ignore_ids=[0]) == ["abcc c ab"]
    assert bpe.decode([10, 11, 8]) == ["abccä"]
    assert bpe.decode([[5, 4, 6]]) == ["a b"]                      # a bare ▁ token is a space
    assert bpe.decode([[2, 10, 3, 1]], ignore_ids=[0]) == ["<BOS> ab<EOS><UNK>"]
    assert bpe.decode([[0, 0]], ignore_ids=[0]) == [""]
    with pytest.raises(ValueError):
        bpe.decode([[99]])
    # dropout-free BPE encode: lowest rule rank first; unknown characters -> unk
    assert bpe.encode(["abcc c ab"]) == [[13, 12, 10]]
    assert bpe.encode("cab x", output_type=yttm.OutputType.SUBWORD) == ["▁c", "ab", "▁", "<UNK>"]
    assert bpe.subword_to_id("▁ab") == 10 and bpe.subword_to_id("zz") == 1
    lang = get_language(mf)
    assert isinstance(lang, TokenizedLanguage) and len(lang) == 14
    assert lang.denumericalize([13, 0, 12]) == "abcc c"
    assert lang.numericalize(" ABCC c </s>") == [13, 12]
    assert lang.denumericalize(lang.numericalize("ab cc ab")) == "ab cc ab"
    with pytest.raises(ValueError):
        yttm.BPE(model=__file__.replace("test_host.py", "conftest.py"))


class _FakeFastcoreL(list):                      # pickled under the name fastai's optimizer state uses: fastcore.foundation.L
    def __reduce__(self):
        return (_FakeFastcoreL, (list(self),))


class _Evil:
    def __reduce__(self):
        import os
        return (os.system, ("echo pwned > lasr_pwned_marker",))


def test_fastai_checkpoint_with_optimizer_state_loads_and_nothing_from_it_runs(tmp_path, monkeypatch):
    """the reference saves with_opt=True (libreasr/lib/patches.py:93) and fastai's Optimizer.state_dict() holds
    fastcore.foundation.L objects, which torch.load(weights_only=True) rejects.  The loader must still return the weights,
    without importing fastcore and without executing anything the file names."""
    import sys
    import types
    import torch
    from libreasr_amd import synth
    from libreasr_amd.lib import model_utils as mu
    cfg = synth.model_cfg("tiny")
    sd = {k: torch.as_tensor(v) for k, v in synth.synth_state_dict(cfg, seed=0).items()}
    fc, fcf = types.ModuleType("fastcore"), types.ModuleType("fastcore.foundation")
    fcf.L = _FakeFastcoreL
    keep = (_FakeFastcoreL.__module__, _FakeFastcoreL.__qualname__, _FakeFastcoreL.__name__)
    sys.modules["fastcore"], sys.modules["fastcore.foundation"] = fc, fcf
    _FakeFastcoreL.__module__, _FakeFastcoreL.__qualname__, _FakeFastcoreL.__name__ = "fastcore.foundation", "L", "L"
    p = tmp_path / "model.pth"
    try:
        torch.save({"model": sd, "opt": {"hypers": _FakeFastcoreL([{"lr": 1e-3, "mom": 0.9}]),
                                         "state": [{"grad_avg": torch.zeros(3)}]}}, p)
    finally:
        _FakeFastcoreL.__module__, _FakeFastcoreL.__qualname__, _FakeFastcoreL.__name__ = keep
        del sys.modules["fastcore"], sys.modules["fastcore.foundation"]
    got = mu.load_model_state_dict(p)
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    # a hostile "opt" entry is not executed
    monkeypatch.chdir(tmp_path)
    torch.save({"model": sd, "opt": _Evil()}, p)
    got = mu.load_model_state_dict(p)
    assert set(got) == set(sd) and not (tmp_path / "lasr_pwned_marker").exists()
    # not a checkpoint at all: a clear error
    torch.save({"opt": 1}, p)
    with pytest.raises(ValueError):
        mu.load_model_state_dict(p)


def test_wav_reader_formats(tmp_path):
    """libreasr_amd/wav.py: integer PCM 8 / 16 / 24 / 32 bit, float32 / float64, WAVE_FORMAT_EXTENSIBLE, odd chunk sizes, extra
    chunks; first channel only, scaled as torchaudio.load does (2^(bits-1))."""
    import struct
    import wave
    from libreasr_amd import wav

    rng = np.random.default_rng(0)
    n = 777

    def riff(fmt_body, data, extra=b""):
        body = b"WAVE" + extra + b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body + b"data" + struct.pack("<I", len(data)) + data
        if len(data) & 1:
            body += b"\0"
        return b"RIFF" + struct.pack("<I", len(body)) + body

    # 16-bit stereo written by the stdlib
    x16 = rng.integers(-32768, 32767, (n, 2)).astype(np.int16)
    p = str(tmp_path / "a.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(48000)
        w.writeframes(x16.tobytes())
    y, sr, bits = wav.decode(p)
    assert (sr, bits) == (48000, 16) and np.array_equal(y, x16[:, 0].astype(np.float32) / 32768.0)
    # 8-bit unsigned mono with an odd data size, and a LIST chunk in front of fmt
    x8 = rng.integers(0, 255, n).astype(np.uint8)
    p = str(tmp_path / "b.wav")
    open(p, "wb").write(riff(struct.pack("<HHIIHH", 1, 1, 8000, 8000, 1, 8), x8.tobytes(), extra=b"LIST" + struct.pack("<I", 3) + b"abc\0"))
    y, sr, bits = wav.decode(p)
    assert (sr, bits) == (8000, 8) and np.array_equal(y, (x8.astype(np.float32) - 128.0) / 128.0)
    # 24-bit mono
    v24 = rng.integers(-(1 << 23), (1 << 23) - 1, n)
    raw = b"".join(int(v).to_bytes(3, "little", signed=True) for v in v24)
    p = str(tmp_path / "c.wav")
    open(p, "wb").write(riff(struct.pack("<HHIIHH", 1, 1, 44100, 44100 * 3, 3, 24), raw))
    y, sr, bits = wav.decode(p)
    assert (sr, bits) == (44100, 24) and np.array_equal(y, (v24.astype(np.float64) / float(1 << 23)).astype(np.float32))
    # 32-bit integer, 3 channels, WAVE_FORMAT_EXTENSIBLE
    v32 = rng.integers(-(1 << 31), (1 << 31) - 1, (n, 3)).astype(np.int32)
    ext = struct.pack("<HHIIHH", 0xFFFE, 3, 16000, 16000 * 12, 12, 32) + struct.pack("<HHI", 22, 32, 7) + struct.pack("<H", 1) + b"\0" * 14
    p = str(tmp_path / "d.wav")
    open(p, "wb").write(riff(ext, v32.tobytes()))
    y, sr, bits = wav.decode(p)
    assert (sr, bits) == (16000, 32) and np.array_equal(y, (v32[:, 0].astype(np.float64) / float(1 << 31)).astype(np.float32))
    # float32 and float64
    f = rng.standard_normal((n, 2)).astype(np.float32)
    p = str(tmp_path / "e.wav")
    open(p, "wb").write(riff(struct.pack("<HHIIHH", 3, 2, 16000, 16000 * 8, 8, 32), f.tobytes()))
    y, sr, bits = wav.decode(p)
    assert (sr, bits) == (16000, 32) and np.array_equal(y, f[:, 0])
    p = str(tmp_path / "f.wav")
    open(p, "wb").write(riff(struct.pack("<HHIIHH", 3, 1, 22050, 22050 * 8, 8, 64), f[:, 1].astype(np.float64).tobytes()))
    y, sr, bits = wav.decode(p)
    assert (sr, bits) == (22050, 64) and np.array_equal(y, f[:, 1])
    # refusals
    for bad in (b"RIFX" + b"\0" * 40, riff(struct.pack("<HHIIHH", 7, 1, 8000, 8000, 1, 8), b"\0" * 8), riff(struct.pack("<HHIIHH", 1, 2, 8000, 8000, 3, 16), b"\0" * 8)):
        p = str(tmp_path / "bad.wav")
        open(p, "wb").write(bad)
        with pytest.raises(ValueError):
            wav.decode(p)


def test_wav_reader_header_edge_cases(tmp_path):
    """zero / odd bit widths are refused with ValueError (not ZeroDivisionError or a numpy reshape error), a streamed
    file (data size 0 or 0xFFFFFFFF, as ffmpeg / sox write to a pipe) is read to the end, an empty file is an error."""
    import struct
    from libreasr_amd import wav

    def riff(fmt, data, size=None):
        body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(data) if size is None else size) + data
        return b"RIFF" + struct.pack("<I", len(body)) + body

    p = str(tmp_path / "x.wav")
    for fmt in (struct.pack("<HHIIHH", 1, 1, 16000, 0, 0, 0), struct.pack("<HHIIHH", 1, 1, 16000, 16000, 1, 12),
                struct.pack("<HHIIHH", 1, 1, 16000, 32000, 0, 16)):
        open(p, "wb").write(riff(fmt, b"\0" * 16))
        with pytest.raises(ValueError):
            wav.decode(p)
    v = np.arange(-50, 50, dtype=np.int16)
    for size, riff_size in ((0xFFFFFFFF, None), (0, 0), (0, 0xFFFFFFFF)):
        raw = riff(struct.pack("<HHIIHH", 1, 1, 16000, 32000, 2, 16), v.tobytes(), size=size)
        if riff_size is not None:
            raw = raw[:4] + struct.pack("<I", riff_size) + raw[8:]
        open(p, "wb").write(raw)
        y, sr, bits = wav.decode(p)
        assert (sr, bits) == (16000, 16) and np.array_equal(y, v.astype(np.float32) / 32768.0)
    # an EMPTY data chunk (size 0, real RIFF size) followed by a LIST chunk is empty -- the trailing chunk is not PCM
    body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + struct.pack("<HHIIHH", 1, 1, 16000, 32000, 2, 16) + b"data" + struct.pack("<I", 0) \
        + b"LIST" + struct.pack("<I", 8) + b"INFOabcd"
    open(p, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
    with pytest.raises(ValueError):
        wav.decode(p)
    open(p, "wb").write(riff(struct.pack("<HHIIHH", 1, 1, 16000, 32000, 2, 16), b""))
    with pytest.raises(ValueError):
        wav.decode(p)


def test_every_library_switch_is_documented_with_an_owner():
    """docs/SWITCHES.md lists exactly the LASR_* variables the library reads (grep getenv in csrc/), each with an owner test
    or tool that exists in the tree (no experiment toggles without an owner)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    for p in glob.glob(os.path.join(root, "libreasr_amd", "csrc", "*.h*")):
        read |= set(re.findall(r'getenv\("(LASR_[A-Z0-9_]+)"\)', open(p).read()))
    doc = open(os.path.join(root, "docs", "SWITCHES.md")).read()
    rows = [ln for ln in doc.splitlines() if ln.startswith("| `LASR_")]
    documented = set()
    for ln in rows:
        cells = [c.strip() for c in ln.strip("|").split("|")]
        names = re.findall(r"`(LASR_[A-Z0-9_]+)`", cells[0])
        documented |= set(names)
        owner = cells[-1]
        files = re.findall(r"`?((?:tests|tools)/[\w/]+\.py|test_\w+\.py)", owner)
        assert files, f"no owner for {names}"
        for f in files:
            path = f if "/" in f else os.path.join("tests", f)
            assert os.path.exists(os.path.join(root, path)), (names, path)
    assert read == documented, (read - documented, documented - read)
    assert len(read) <= 30


def test_native_front_wrapper_frees_the_handle_only_when_no_call_is_inside():
    """libreasr_amd.front.NativeFront on a stand-in library (no GPU): stop() lets no new call in and calls lasr_front_stop;
    destroy() waits until the call that was blocked inside has returned before lasr_front_destroy runs; a second destroy is a no-op."""
    import threading
    import time
    import types
    from libreasr_amd import _native as N
    from libreasr_amd.front import NativeFront

    log = []
    release = threading.Event()

    class Lib:
        def lasr_front_create(self, ctx, depth, reset_steps, out):
            log.append("create")
            return 0

        def lasr_front_next(self, h, stream, buf, cap, n, fl, timeout):
            log.append("next-in")
            release.wait(5)                       # (the native call blocks until lasr_front_stop releases it)
            log.append("next-out")
            return N.LASR_ESTATE

        def lasr_front_stop(self, h):
            log.append("stop")
            release.set()
            return 0

        def lasr_front_destroy(self, h):
            log.append("destroy")

        def lasr_front_open(self, h, s):
            log.append("open")
            return 0

        def lasr_front_error(self, h):
            return b"stopped"

        def lasr_last_error(self, ctx):
            return b""

    eng = types.SimpleNamespace(lib=Lib(), ctx=None, desc=types.SimpleNamespace(chunk=1280, n_buffer=2, max_iters_stream=10),
                                _chk=lambda rc: rc)
    front = NativeFront(eng, depth=4)
    assert front in eng._fronts
    err = []

    def consumer():
        try:
            front.next(0)
        except N.LasrError as e:
            err.append(e.code)

    t = threading.Thread(target=consumer)
    t.start()
    while "next-in" not in log:
        time.sleep(0.001)
    front.destroy()                               # stop -> the consumer leaves -> destroy
    t.join(5)
    assert log == ["create", "next-in", "stop", "next-out", "destroy"]
    assert err == [N.LASR_ESTATE] and front.h is None and front not in eng._fronts
    with pytest.raises(N.LasrError):
        front.open()
    front.destroy()
    assert log[-1] == "destroy" and log.count("destroy") == 1


def test_native_front_on_a_stand_in_engine_under_asan_ubsan(tmp_path):
    """SURVEY §5's host-side sanitizer target: libreasr_amd/csrc/lasr_front.hip.h -- the part of the library
    that is threads, rings and lifetimes -- compiled with g++ against a stand-in of the dozen engine calls it makes
    (tests/c/front_standin.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer: producers / consumers per stream, the reset
    rule on text with reset_steps <= depth, a close under a blocked producer, stale stream ids, stop with threads inside.  (GPU
    ASan is not available on this pool; gcc 11's ThreadSanitizer lacks the pthread_cond_clockwait interceptor and reports every
    condition_variable::wait_for as a double lock, so TSan is not a gate here.)"""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = shutil.which("g++")
    assert cxx, "g++ is part of the image"
    exe = str(tmp_path / "front_standin")
    subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-fno-omit-frame-pointer", "-pthread", os.path.join(root, "tests", "c", "front_standin.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "front stand-in: ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


def test_abi_consumer_under_asan_ubsan(tmp_path):
    """tests/c/abi_consumer.c (plain C against include/lasr.h) built with -fsanitize=address,undefined against the real library:
    description checks, weight counts and the loud failure of lasr_create without a GPU run clean (leak detection off: the HIP
    runtime keeps its own allocations until exit)."""
    import shutil
    import subprocess
    import __graft_entry__ as graft
    graft.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "libreasr_amd", "csrc")
    exe = str(tmp_path / "abi_consumer_asan")
    subprocess.run([shutil.which("gcc"), "-std=c99", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "abi_consumer.c"), "-o", exe,
                    "-L", csrc, "-llasr_hip", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    import torch
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == (0 if torch.cuda.is_available() else 10), (r.returncode, r.stdout, r.stderr[-2000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
