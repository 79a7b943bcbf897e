"""Mirror of the inference half of libreasr/lib/model_utils.py: model archives.

`libreasr-model-*.tar.gz` (model_utils.py:31-58) holds `<lang>/model.pth` -- a fastai `learn.save` file,
i.e. {"model": state_dict, "opt": ...} (fastai2 load_model, model_utils.py:79-85) -- and
`<lang>/tokenizer.yttm-model`; `lm.pth` is a plain LM state_dict (lm.py:93).  extract_tars() unpacks
them under ./tmp exactly like the reference, load_model_state_dict() / load_lm_state_dict() hand the
tensors to libreasr_amd.weights (nothing is quantised here: model_utils.py:90-93 discards the quantised
copy of the ASR model, and the LM runs in fp32 / bf16)."""
import glob
import os
import tarfile
from pathlib import Path

import torch

_PATH_DEST = Path("./tmp")


def extract_tars(paths_archive=None, path_dest=_PATH_DEST):
    """model_utils.py:50-58.  Members that would land outside `path_dest` are refused."""
    if paths_archive is None:
        paths_archive = glob.glob("./libreasr-model-*.tar.gz")
    dest = os.path.realpath(str(path_dest))
    out = []
    for arc in paths_archive:
        with tarfile.open(arc) as tar:
            for m in tar.getmembers():
                target = os.path.realpath(os.path.join(dest, m.name))
                if not (target == dest or target.startswith(dest + os.sep)) or not (m.isreg() or m.isdir()):
                    raise ValueError(f"unsafe member {m.name!r} in {arc}")        # links, devices, FIFOs, escapes
            try:
                tar.extractall(path=dest, filter="data")
            except TypeError:                                                  # Python without extraction filters
                tar.extractall(path=dest)
            out += [m.name for m in tar.getmembers()]
    return out


def _safe_load(path):
    """Checkpoints come out of an archive found in the working directory: tensors-only unpickling.  The fastai
    wrapper's "opt" entry is plain containers + tensors and loads under weights_only as well."""
    return torch.load(str(path), map_location="cpu", weights_only=True)


def load_model_state_dict(path):
    """`model.pth`: plain state_dict or the fastai {"model": ..., "opt": ...} wrapper -> {key: tensor}."""
    sd = _safe_load(path)
    if isinstance(sd, dict) and "model" in sd and not any(str(k).startswith("encoder.") for k in sd):
        sd = sd["model"]
    return {k: v for k, v in sd.items() if torch.is_tensor(v)}


def load_lm_state_dict(path):
    sd = _safe_load(path)
    return {k: v for k, v in sd.items() if torch.is_tensor(v)}


def model_paths(lang, path_dest=_PATH_DEST):
    base = Path(path_dest) / lang
    return dict(model=base / "model.pth", tokenizer=base / "tokenizer.yttm-model", lm=base / "lm.pth")
