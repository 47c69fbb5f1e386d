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
