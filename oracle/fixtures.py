"""TEST INFRASTRUCTURE ONLY.  (De)serialise oracle layer specs + graph inputs to flat ``.npz``
archives so golden vectors generated from the real reference (tests/golden/make_golden.py) can be
committed as small fixtures and replayed on the GPU box, where /root/reference does not exist."""
import json
from typing import Dict, List

import numpy as np
import torch


def _to_np(v):
    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


def pack_specs(layers: List[Dict]) -> Dict[str, np.ndarray]:
    """Tied layers (same dict object repeated) are stored once and referenced by index."""
    uniq, order, arrays, meta = [], [], {}, []
    for spec in layers:
        for i, u in enumerate(uniq):
            if u is spec:
                order.append(i)
                break
        else:
            order.append(len(uniq))
            uniq.append(spec)
    for i, spec in enumerate(uniq):
        m = {}
        for k, v in spec.items():
            if isinstance(v, torch.Tensor):
                arrays[f"L{i}.{k}"] = _to_np(v)
                m[k] = "tensor"
            elif isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
                for j, t in enumerate(v):
                    arrays[f"L{i}.{k}.{j}"] = _to_np(t)
                m[k] = ["tensorlist", len(v)]
            elif isinstance(v, list) and v and isinstance(v[0], list):
                for j, tl in enumerate(v):
                    for q, t in enumerate(tl):
                        arrays[f"L{i}.{k}.{j}.{q}"] = _to_np(t)
                m[k] = ["tensorlistlist", [len(tl) for tl in v]]
            else:
                m[k] = ["py", v]
        meta.append(m)
    arrays["__layers__"] = np.frombuffer(
        json.dumps({"order": order, "meta": meta}).encode(), dtype=np.uint8)
    return arrays


def unpack_specs(npz) -> List[Dict]:
    info = json.loads(bytes(npz["__layers__"]).decode())
    uniq = []
    for i, m in enumerate(info["meta"]):
        spec = {}
        for k, tag in m.items():
            if tag == "tensor":
                spec[k] = torch.from_numpy(np.array(npz[f"L{i}.{k}"]))
            elif tag[0] == "tensorlist":
                spec[k] = [torch.from_numpy(np.array(npz[f"L{i}.{k}.{j}"])) for j in range(tag[1])]
            elif tag[0] == "tensorlistlist":
                spec[k] = [[torch.from_numpy(np.array(npz[f"L{i}.{k}.{j}.{q}"])) for q in range(n)]
                           for j, n in enumerate(tag[1])]
            else:
                spec[k] = tag[1]
        uniq.append(spec)
    return [uniq[i] for i in info["order"]]


def pack_adj(adj) -> Dict[str, np.ndarray]:
    out = {"__num_edge_types__": np.asarray(len(adj))}
    for t, (s, d) in enumerate(adj):
        out[f"adj.{t}.src"] = _to_np(s).astype(np.int64)
        out[f"adj.{t}.dst"] = _to_np(d).astype(np.int64)
    return out


def unpack_adj(npz):
    T = int(npz["__num_edge_types__"])
    return [(torch.from_numpy(np.array(npz[f"adj.{t}.src"])),
             torch.from_numpy(np.array(npz[f"adj.{t}.dst"]))) for t in range(T)]
