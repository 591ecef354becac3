"""Disjoint-union minibatch assembly with the index arithmetic on the MI355X.

Mirrors the graph part of the reference's batching interface
(ptgnn/neuralmodels/gnn/graphneuralnetwork.py): `initialize_minibatch` (:372-384),
`extend_minibatch_with` (:386-438) and `finalize_minibatch` (:445-493) -- same inputs (the per-graph
`TensorizedGraphData`: int32 `(src, dst)` pairs per edge type, `reference_nodes`, `num_nodes`), same
stopping rule, same output dict keys and int64 layouts, bit for bit.

What differs is where the work happens.  The reference adds the running node offset to every edge array
on the host (`adj + nodes_in_mb_so_far`, one numpy op per graph and edge type), concatenates, converts
to int64 and uploads 2T + 1 + 2R tensors one by one; `node_to_graph_idx` comes out of a Python generator
(:440-443).  Here `extend` only records the raw arrays; `finalize` copies them into ONE pinned int32
staging buffer, uploads it with a single async H2D copy together with a small segment table, and one
HIP launch (`ptgnn_amd_batch_offsets_i64`, csrc/batching.hip) writes every int64 index tensor of the
minibatch -- offsets applied, widened, `node_to_graph_idx` and `reference_node_graph_idx` filled.  The
returned tensors are views into one device buffer.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ptgnn_amd import _lib


class PackedMinibatch:
    """Host-side packing of a minibatch (pure numpy; `finalize` ships it to the device).

    staging  int32 [n_in]   : every raw per-graph array, block after block
    seg_start int64 [S + 1], seg_add int64 [S] : the segment table of ptgnn_amd_batch_offsets_i64
    layout   : name -> (begin, end) element range of each output tensor in the int64 result
    """

    def __init__(self, staging, seg_start, seg_add, n_out, layout, num_graphs):
        self.staging, self.seg_start, self.seg_add = staging, seg_start, seg_add
        self.n_out, self.layout, self.num_graphs = n_out, layout, num_graphs

    def evaluate_on_host(self) -> np.ndarray:
        """The kernel's formula in numpy (used by the CPU tests to pin the host logic)."""
        n_in = self.staging.shape[0]
        vals = np.zeros(self.n_out, dtype=np.int64)
        vals[:n_in] = self.staging
        seg = np.searchsorted(self.seg_start, np.arange(self.n_out), side="right") - 1
        return vals + self.seg_add[seg] if self.n_out else vals


class MinibatchBuilder:
    """Accumulates tensorized graphs; `finalize(device)` returns the reference's minibatch dict entries
    `adjacency_lists`, `node_to_graph_idx`, `reference_node_graph_idx`, `reference_node_ids`,
    `num_graphs` (graphneuralnetwork.py:457-493)."""

    def __init__(self, num_edge_types: int, stop_extending_minibatch_after_num_nodes: int = 10000):
        self.num_edge_types = int(num_edge_types)
        self.stop_extending_minibatch_after_num_nodes = int(stop_extending_minibatch_after_num_nodes)
        self._graphs: List[Tuple[List[Tuple[np.ndarray, np.ndarray]], Dict[str, np.ndarray], int]] = []
        self.num_nodes_in_mb = 0

    def __len__(self) -> int:
        return len(self._graphs)

    def extend(self, adjacency_lists: Sequence[Tuple[np.ndarray, np.ndarray]], num_nodes: int,
               reference_nodes: Optional[Dict[str, np.ndarray]] = None) -> bool:
        """extend_minibatch_with (:386-438): returns whether the minibatch may keep growing."""
        if len(adjacency_lists) != self.num_edge_types:
            raise ValueError(f"expected {self.num_edge_types} adjacency lists, got {len(adjacency_lists)}")
        adj = []
        for s, d in adjacency_lists:
            s = np.ascontiguousarray(s, dtype=np.int32).reshape(-1)
            d = np.ascontiguousarray(d, dtype=np.int32).reshape(-1)
            if s.shape != d.shape:
                raise ValueError("source and target arrays of an edge type differ in length")
            adj.append((s, d))
        refs = {k: np.ascontiguousarray(v, dtype=np.int32).reshape(-1)
                for k, v in (reference_nodes or {}).items()}
        self._graphs.append((adj, refs, int(num_nodes)))
        self.num_nodes_in_mb += int(num_nodes)
        return self.num_nodes_in_mb < self.stop_extending_minibatch_after_num_nodes

    # -- host packing ---------------------------------------------------------------------------
    def pack(self) -> PackedMinibatch:
        G, T = len(self._graphs), self.num_edge_types
        node_off = np.zeros(G + 1, dtype=np.int64)
        for g, (_, _, n) in enumerate(self._graphs):
            node_off[g + 1] = node_off[g] + n
        ref_names: List[str] = []
        for _, refs, _ in self._graphs:           # dict insertion order of first appearance (:431-434)
            for k in refs:
                if k not in ref_names:
                    ref_names.append(k)

        pieces: List[np.ndarray] = []             # raw int32 arrays, in staging order
        seg_len: List[int] = []
        seg_add: List[int] = []
        layout: Dict[str, Tuple[int, int]] = {}
        pos = 0

        def block(name: str, arrays: List[Optional[np.ndarray]], adds, lengths=None):
            nonlocal pos
            begin = pos
            for g, arr in enumerate(arrays):
                n = int(arr.shape[0]) if arr is not None else int(lengths[g])
                if arr is not None and n:
                    pieces.append(arr)
                seg_len.append(n)
                seg_add.append(int(adds[g]))
                pos += n
            layout[name] = (begin, pos)

        for t in range(T):
            block(f"adj.{t}.src", [gr[0][t][0] for gr in self._graphs], node_off[:G])
            block(f"adj.{t}.dst", [gr[0][t][1] for gr in self._graphs], node_off[:G])
        empty = np.zeros(0, dtype=np.int32)
        for k in ref_names:
            block(f"ref_ids.{k}", [gr[1].get(k, empty) for gr in self._graphs], node_off[:G])
        n_in = pos
        # fill segments: value = graph index
        block("node_to_graph_idx", [None] * G, np.arange(G), lengths=[gr[2] for gr in self._graphs])
        for k in ref_names:
            block(f"ref_gidx.{k}", [None] * G, np.arange(G),
                  lengths=[gr[1].get(k, empty).shape[0] for gr in self._graphs])
        staging = np.concatenate(pieces) if pieces else np.zeros(0, dtype=np.int32)
        assert staging.shape[0] == n_in
        seg_start = np.zeros(len(seg_len) + 1, dtype=np.int64)
        np.cumsum(np.asarray(seg_len, dtype=np.int64), out=seg_start[1:])
        return PackedMinibatch(staging, seg_start, np.asarray(seg_add, dtype=np.int64), pos, layout, G)

    # -- device ---------------------------------------------------------------------------------
    def finalize(self, device) -> Dict:
        """finalize_minibatch (:445-493) with the index arithmetic on `device` (an MI355X)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.PtgnnAmdError("MinibatchBuilder.finalize assembles the minibatch on the MI355X; "
                                     f"got device {device} (use the reference batcher for CPU runs)")
        lib = _lib.load()
        pk = self.pack()
        n_in, S = int(pk.staging.shape[0]), int(pk.seg_add.shape[0])
        # one pinned buffer: [seg_start | seg_add] int64 then the int32 staging (8-byte aligned start)
        meta = 2 * S + 1
        host = torch.empty(meta * 8 + n_in * 4, dtype=torch.uint8, pin_memory=True)
        hv = host.numpy()
        hv[: (S + 1) * 8].view(np.int64)[:] = pk.seg_start
        hv[(S + 1) * 8: meta * 8].view(np.int64)[:] = pk.seg_add
        hv[meta * 8:].view(np.int32)[:] = pk.staging
        dev = host.to(device, non_blocking=True)
        seg_start = dev[: (S + 1) * 8].view(torch.int64)
        seg_add = dev[(S + 1) * 8: meta * 8].view(torch.int64)
        raw = dev[meta * 8:].view(torch.int32)
        out = torch.empty(pk.n_out, dtype=torch.int64, device=device)
        rc = lib.ptgnn_amd_batch_offsets_i64(raw.data_ptr() if n_in else None, n_in, seg_start.data_ptr(),
                                             seg_add.data_ptr(), S, pk.n_out, out.data_ptr(),
                                             torch.cuda.current_stream(device).cuda_stream)
        _lib.check(rc, "ptgnn_amd_batch_offsets_i64")
        # `host` must outlive the async copy: torch's caching host allocator defers its reuse until the
        # copy's stream event has completed, so dropping the reference here is safe

        def view(name):
            b, e = pk.layout[name]
            return out[b:e]
        ref_names = [k[len("ref_ids."):] for k in pk.layout if k.startswith("ref_ids.")]
        return {
            "adjacency_lists": [(view(f"adj.{t}.src"), view(f"adj.{t}.dst"))
                                for t in range(self.num_edge_types)],
            "node_to_graph_idx": view("node_to_graph_idx"),
            "reference_node_graph_idx": {k: view(f"ref_gidx.{k}") for k in ref_names},
            "reference_node_ids": {k: view(f"ref_ids.{k}") for k in ref_names},
            "num_graphs": pk.num_graphs,
        }
