"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/lasr.h declares.  No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import __graft_entry__ as graft
from libreasr_amd import _native as N
from libreasr_amd import synth
from libreasr_amd.weights import flatten_state_dict, infer_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    graft.build()
    return N.lib()


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(lasr_[a-z_0-9]+)\s*\(", hdr))


def test_every_header_symbol_is_exported_and_bound(lib):
    """include/lasr.h = the drop-in surface (every call maps to a reference interface in INTEGRATION.md), include/lasr_debug.h =
    bench / debug / trace / experiment hooks.  Each header's declarations == the Python binding's list of the same name, nothing
    is declared twice, and the library exports exactly the union."""
    import subprocess
    api, dbg = _declared("lasr.h"), _declared("lasr_debug.h")
    assert api == {n for n, _, _ in N.SYMBOLS}, (api ^ {n for n, _, _ in N.SYMBOLS})
    assert dbg == {n for n, _, _ in N.DEBUG_SYMBOLS}, (dbg ^ {n for n, _, _ in N.DEBUG_SYMBOLS})
    assert not (api & dbg)
    # no measurement / experiment hook leaks into the drop-in header
    assert not [n for n in api if n.startswith(("lasr_bench", "lasr_debug", "lasr_trace", "lasr_cell_prof", "lasr_overlap"))]
    for name in api | dbg:
        assert hasattr(lib, name)
    out = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("lasr_")}
    assert exported == api | dbg, (exported ^ (api | dbg))


def test_default_desc_and_weight_count(lib):
    d = N.ModelDesc()
    lib.lasr_default_desc(C.byref(d))
    assert (d.feat, d.hidden, d.enc_layers, d.pred_layers, d.vocab, d.blank, d.bos) == (1280, 1024, 4, 2, 2048, 0, 2)
    assert (d.n_fft, d.win, d.hop, d.n_mels, d.n_stack, d.stride, d.n_buffer, d.n_window, d.chunk) == \
        (1024, 400, 160, 128, 10, 8, 2, 3, 1280)
    assert (d.max_iters_offline, d.max_iters_stream) == (3, 10)     # models.py:369,458
    for name in ("tiny", "tiny_lstm", "cfg2", "cfg2_lstm"):
        cfg = synth.model_cfg(name)
        if name.startswith("cfg2"):
            # count only (do not materialise 53 M floats twice): formula check
            F, E, V, H, J = cfg["feat"], cfg["embed"], cfg["vocab"], cfg["hidden"], cfg["joint"]
        sd = synth.synth_state_dict(cfg) if name.startswith("tiny") else None
        d.feat, d.hidden, d.embed, d.joint, d.vocab = cfg["feat"], cfg["hidden"], cfg["embed"], cfg["joint"], cfg["vocab"]
        d.enc_layers, d.pred_layers = cfg["enc_layers"], cfg["pred_layers"]
        d.pred_cell = 1 if cfg["pred_cell"] == "LSTM" else 0
        n = lib.lasr_weight_count(C.byref(d))
        assert n > 0
        if sd is not None:
            blob = flatten_state_dict(sd, cfg)
            assert blob.size == n
            assert infer_cfg(sd) == {k: cfg[k] for k in infer_cfg(sd)}
    # reference default 6-2-1024: 69.80 M parameters ("70M", docs/docs.md:131-137)
    cfg = synth.model_cfg("ref6")
    d.feat, d.hidden, d.embed, d.joint, d.vocab = cfg["feat"], cfg["hidden"], cfg["embed"], cfg["joint"], cfg["vocab"]
    d.enc_layers, d.pred_layers, d.pred_cell = 6, 2, 0
    n = lib.lasr_weight_count(C.byref(d))
    # blob = parameters + BN running stats (2H per layer); hs are parameters
    n_params = n - 2 * 1024 * (6 + 2)
    assert abs(n_params - 69.80e6) < 0.05e6


def test_invalid_desc_is_rejected(lib):
    d = N.ModelDesc()
    lib.lasr_default_desc(C.byref(d))
    d.hidden = 1000          # not a multiple of 16
    assert lib.lasr_weight_count(C.byref(d)) == 0
    ctx = C.c_void_p()
    rc = lib.lasr_create(0, C.byref(d), None, 0, None, C.byref(ctx))
    assert rc == N.LASR_EINVAL
    assert b"invalid" in lib.lasr_last_error(ctx)
    lib.lasr_destroy(ctx)


def test_create_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = synth.model_cfg("tiny")
    blob = flatten_state_dict(synth.synth_state_dict(cfg), cfg)
    d = N.ModelDesc()
    lib.lasr_default_desc(C.byref(d))
    d.feat, d.hidden, d.embed, d.joint, d.vocab = cfg["feat"], cfg["hidden"], cfg["embed"], cfg["joint"], cfg["vocab"]
    d.enc_layers, d.pred_layers, d.pred_cell, d.max_streams = 2, 2, 0, 16
    ctx = C.c_void_p()
    rc = lib.lasr_create(0, C.byref(d), blob.ctypes.data_as(C.c_void_p), blob.size, None, C.byref(ctx))
    assert rc == N.LASR_EHIP                         # no device: an error code, never a CPU fallback
    lib.lasr_destroy(ctx)
    from libreasr_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine(synth.synth_state_dict(cfg), cfg, max_streams=16)


def test_flatten_rejects_bad_shapes():
    cfg = synth.model_cfg("tiny")
    sd = synth.synth_state_dict(cfg)
    sd["joint.joint.2.weight"] = sd["joint.joint.2.weight"][:, :-1]
    with pytest.raises(ValueError):
        flatten_state_dict(sd, cfg)
    del sd["joint.joint.2.weight"]
    with pytest.raises(KeyError):
        flatten_state_dict(sd, cfg)


def test_plain_c_consumer_links_and_fails_loudly_without_gpu(tmp_path):
    """include/lasr.h is C99; a C program linked against liblasr_hip.so sizes the weight blob of the reference
    shape (53.03 M parameters + BatchNorm statistics) and gets LASR_EHIP from lasr_create on a box without an MI355X."""
    import shutil
    import subprocess
    import __graft_entry__ as graft
    graft.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "libreasr_amd", "csrc")
    exe = str(tmp_path / "abi_consumer")
    cc = shutil.which("gcc")
    assert cc, "gcc is part of the image"
    subprocess.run([cc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c", "abi_consumer.c"), "-o", exe, "-L", csrc, "-llasr_hip",
                    "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    import torch
    has_gpu = torch.cuda.is_available()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert "weights 53039616" in r.stdout, r.stdout + r.stderr
    assert "lm weights" in r.stdout
    assert r.returncode == (0 if has_gpu else 10), (r.returncode, r.stdout, r.stderr)
