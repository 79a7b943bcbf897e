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


class _Opaque(list):
    """Stand-in for every class a checkpoint names that is not a tensor building block: fastai's `learn.save(with_opt=True)`
    (libreasr/lib/patches.py:93) pickles the optimizer state next to the weights, and `Optimizer.state_dict()` holds
    `fastcore.foundation.L` objects (fastcore is not a dependency here, and arbitrary classes must not be imported from an
    archive found in the working directory anyway).  Only sd["model"] is read; whatever lands in a stand-in is dropped."""

    def __init__(self, *a, **k):
        super().__init__()

    def __setstate__(self, state):
        pass

    def __setitem__(self, k, v):
        pass

    def __call__(self, *a, **k):
        return _Opaque()


def _restricted_pickle():
    """A pickle module whose Unpickler resolves ONLY what tensors are made of; any other global becomes _Opaque."""
    import collections
    import pickle
    import types

    allowed_torch = {"_rebuild_tensor_v2", "_rebuild_tensor", "_rebuild_parameter", "_rebuild_parameter_with_state",
                     "_rebuild_qtensor", "Size", "device", "dtype"}

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module == "collections" and name == "OrderedDict":
                return collections.OrderedDict
            if module in ("torch._utils", "torch") and name in allowed_torch:
                return getattr(torch._utils if module == "torch._utils" else torch, name)
            if module == "torch" and (name.endswith("Storage") or name in ("float32", "float16", "bfloat16", "float64", "int64", "int32", "int8", "uint8", "bool")):
                return getattr(torch, name)
            if module == "torch.serialization" and name == "_get_layout":
                return torch.serialization._get_layout
            return _Opaque

    mod = types.ModuleType("libreasr_amd_restricted_pickle")
    mod.Unpickler = Unpickler
    mod.load = lambda f, **kw: Unpickler(f, **kw).load()
    mod.__name__ = "pickle"            # torch.load special-cases the module's name in one code path
    for k in ("HIGHEST_PROTOCOL", "DEFAULT_PROTOCOL", "PickleError", "UnpicklingError", "dumps", "loads", "dump", "Pickler"):
        setattr(mod, k, getattr(pickle, k))
    return mod


def _safe_load(path):
    """Checkpoints come out of an archive found in the working directory.  First tensors-only unpickling (torch's
    weights_only); a fastai `model.pth` written with the optimizer state names classes that loader rejects
    (fastcore.foundation.L): it is then read with an Unpickler that builds tensors and plain containers only and turns
    every other class into an inert stand-in -- nothing from the file is imported or executed."""
    import pickle
    try:
        return torch.load(str(path), map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:
        return torch.load(str(path), map_location="cpu", weights_only=False, pickle_module=_restricted_pickle())


def load_model_state_dict(path):
    """`model.pth`: plain state_dict or the fastai {"model": ..., "opt": ...} wrapper -> {key: tensor}."""
    sd = _safe_load(path)
    if isinstance(sd, dict) and "model" in sd and not any(str(k).startswith("encoder.") for k in sd):
        sd = sd["model"]
    if not isinstance(sd, dict) or not any(torch.is_tensor(v) for v in sd.values()):
        raise ValueError(f"{path}: no state_dict found (expected tensors under the reference's key names, or a fastai "
                         "{'model': state_dict, 'opt': ...} wrapper)")
    return {k: v for k, v in sd.items() if torch.is_tensor(v)}


def load_lm_state_dict(path):
    sd = _safe_load(path)
    return {k: v for k, v in sd.items() if torch.is_tensor(v)}


def model_paths(lang, path_dest=_PATH_DEST):
    base = Path(path_dest) / lang
    return dict(model=base / "model.pth", tokenizer=base / "tokenizer.yttm-model", lm=base / "lm.pth")
