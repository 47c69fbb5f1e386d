"""Checkpoint files in the reference's (SB3) zip layout (reference: core/common/save_util.py:294-466,
core/common/base_class.py:666-888): `data` (JSON), `<name>.pth` per state dict (policy, actor.optimizer, ...),
`pytorch_variables.pth`, `_stable_baselines3_version`, `system_info.txt`.

Safe by construction: tensors are read with `torch.load(weights_only=True)`, `data` with `json`; entries the
reference serialised with cloudpickle (`":serialized:"`) are never unpickled -- they are skipped (gymnasium spaces,
schedules and policy classes are rebuilt from the constructor arguments instead)."""
import io
import json
import os
import zipfile
from typing import Any, Optional, Tuple

import numpy as np
import torch as th


def _jsonable(v: Any):
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    if isinstance(v, (np.integer, np.floating)):
        return v.item()
    if isinstance(v, (list, tuple)):
        out = [_jsonable(x) for x in v]
        return None if any(x is _SKIP for x in out) else out
    if isinstance(v, dict) and all(isinstance(k, str) for k in v):
        out = {k: _jsonable(x) for k, x in v.items()}
        return _SKIP if any(x is _SKIP for x in out.values()) else out
    return _SKIP


_SKIP = object()


def data_to_json(data: dict) -> str:
    out = {}
    for k, v in data.items():
        j = _jsonable(v)
        if j is not _SKIP:
            out[k] = j
    return json.dumps(out, indent=4)


def json_to_data(text: str) -> dict:
    raw = json.loads(text)
    return {k: v for k, v in raw.items() if not (isinstance(v, dict) and ":serialized:" in v)}


def save_to_zip_file(path, data: Optional[dict], params: Optional[dict], pytorch_variables: Optional[dict], version: str) -> None:
    if isinstance(path, (str, os.PathLike)):
        path = str(path)
        if not path.endswith(".zip"):
            path += ".zip"
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with zipfile.ZipFile(path, mode="w") as archive:
        if data is not None:
            archive.writestr("data", data_to_json(data))
        if pytorch_variables is not None:
            with archive.open("pytorch_variables.pth", mode="w", force_zip64=True) as fh:
                th.save(pytorch_variables, fh)
        for name, sd in (params or {}).items():
            with archive.open(name + ".pth", mode="w", force_zip64=True) as fh:
                th.save(sd, fh)
        archive.writestr("_stable_baselines3_version", version)
        archive.writestr("system_info.txt", f"torch {th.__version__}; numpy {np.__version__}; MI355X-native build\n")


def load_from_zip_file(path, device="cpu") -> Tuple[dict, dict, dict]:
    if isinstance(path, (str, os.PathLike)):
        path = str(path)
        if not os.path.exists(path) and os.path.exists(path + ".zip"):
            path += ".zip"
    data, params, variables = {}, {}, {}
    with zipfile.ZipFile(path) as archive:
        names = archive.namelist()
        if "data" in names:
            data = json_to_data(archive.read("data").decode())
        for name in names:
            if not name.endswith(".pth"):
                continue
            obj = th.load(io.BytesIO(archive.read(name)), map_location=device, weights_only=True)
            if name == "pytorch_variables.pth":
                variables = obj
            else:
                params[name[:-4]] = obj
    return data, params, variables


def _open_path(path, mode: str, suffix: str):
    """reference: save_util.py:205-292 (open_path): str / pathlib paths get the suffix when they have none; parent
    directories are created when writing; file objects pass through."""
    import pathlib

    if isinstance(path, (io.BufferedIOBase, io.RawIOBase)):
        return path, False
    path = pathlib.Path(path)
    if path.suffix == "" and suffix:
        path = path.with_suffix("." + suffix)
    if "w" in mode:
        path.parent.mkdir(parents=True, exist_ok=True)
    return open(path, mode), True


def save_to_pkl(path, obj: Any, verbose: int = 0) -> None:
    """reference: save_util.py:349-367"""
    import pickle

    f, close = _open_path(path, "wb", "pkl")
    try:
        pickle.dump(obj, f, protocol=pickle.HIGHEST_PROTOCOL)
    finally:
        if close:
            f.close()


def load_from_pkl(path, verbose: int = 0) -> Any:
    """reference: save_util.py:370-384. Unpickling executes code from the file: load your own files only."""
    import pickle

    f, close = _open_path(path, "rb", "pkl")
    try:
        return pickle.load(f)
    finally:
        if close:
            f.close()
