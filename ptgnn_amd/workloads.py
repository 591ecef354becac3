"""Synthetic stand-ins for the five BASELINE.json configs.  Pure generators:
seeded numpy/torch CPU tensors in the exact integer layout ptgnn's
`GraphNeuralNetworkModel.finalize_minibatch` produces (int64 src/dst per edge type,
`node_to_graph_idx`, reference ids).  Real datasets are not available offline.
"""
from typing import Dict, List, Tuple

import numpy as np
import torch

Adj = List[Tuple[torch.Tensor, torch.Tensor]]


def random_graph(num_nodes: int, num_edges: int, seed: int = 1234) -> Adj:
    """Config 2: one edge type, uniform endpoints (`randint(0, N, (E,))`, generator seed 1234)."""
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, num_nodes, (num_edges,), generator=g, dtype=torch.int64)
    dst = torch.randint(0, num_nodes, (num_edges,), generator=g, dtype=torch.int64)
    return [(src, dst)]


def node_states(num_nodes: int, dim: int, seed: int = 1234) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(num_nodes, dim, generator=g, dtype=torch.float32)


def batched_graphs(num_graphs: int, nodes_per_graph: int, num_raw_types: int,
                   raw_edges_per_node: float, refs_per_graph: int = 20, seed: int = 1234,
                   jitter: float = 0.2) -> Dict:
    """Configs 1/3/4: a disjoint-union batch of `num_graphs` random graphs.  Node counts are
    ~U[(1-jitter), (1+jitter)] * nodes_per_graph; raw edges = raw_edges_per_node * n per graph,
    split over the raw types by a Zipf law, endpoints uniform within the graph; node ids are
    offset exactly as extend_minibatch_with does (graphneuralnetwork.py:418-423)."""
    rng = np.random.RandomState(seed)
    zipf = 1.0 / np.arange(1, num_raw_types + 1)
    zipf /= zipf.sum()
    src: List[List[np.ndarray]] = [[] for _ in range(num_raw_types)]
    dst: List[List[np.ndarray]] = [[] for _ in range(num_raw_types)]
    n2g, ref_ids, ref_g = [], [], []
    off = 0
    for g in range(num_graphs):
        n = int(nodes_per_graph * rng.uniform(1 - jitter, 1 + jitter))
        e_total = int(raw_edges_per_node * n)
        counts = rng.multinomial(e_total, zipf)
        for t in range(num_raw_types):
            src[t].append(rng.randint(0, n, size=counts[t]).astype(np.int64) + off)
            dst[t].append(rng.randint(0, n, size=counts[t]).astype(np.int64) + off)
        n2g.append(np.full(n, g, dtype=np.int64))
        r = rng.choice(n, size=min(refs_per_graph, n), replace=False).astype(np.int64) + off
        ref_ids.append(r)
        ref_g.append(np.full(len(r), g, dtype=np.int64))
        off += n
    return {
        "adjacency_lists": [(torch.from_numpy(np.concatenate(s)), torch.from_numpy(np.concatenate(d)))
                            for s, d in zip(src, dst)],
        "node_to_graph_idx": torch.from_numpy(np.concatenate(n2g)),
        "reference_node_ids": {"supernodes": torch.from_numpy(np.concatenate(ref_ids))},
        "reference_node_graph_idx": {"supernodes": torch.from_numpy(np.concatenate(ref_g))},
        "num_graphs": num_graphs,
        "num_nodes": off,
    }


def graph_list(num_graphs: int, nodes_lo: int, nodes_hi: int, num_raw_types: int, raw_edges_per_node: float,
               refs_per_graph: int = 0, seed: int = 1234) -> List[Dict]:
    """Config 1 "as the reference batches it" (SURVEY.md 8d): SEPARATE tensorized graphs -- the per-graph records
    `GraphNeuralNetworkModel.tensorize` hands to `extend_minibatch_with` (graphneuralnetwork.py:325-367,386-438): int32
    `(src, dst)` pairs per raw edge type with graph-local node ids, `num_nodes`, `reference_nodes` -- so that the batcher
    (ptgnn_amd.batching.MinibatchBuilder here, oracle.mp_oracle.batch_graphs on the checker side) forms the
    minibatches under its own node cap.  Node counts ~U[nodes_lo, nodes_hi], raw edges = raw_edges_per_node * n with
    uniform endpoints, split over the raw types by the Zipf law of `batched_graphs`."""
    rng = np.random.RandomState(seed)
    zipf = 1.0 / np.arange(1, num_raw_types + 1)
    zipf /= zipf.sum()
    graphs = []
    for _ in range(num_graphs):
        n = int(rng.randint(nodes_lo, nodes_hi + 1))
        counts = rng.multinomial(int(raw_edges_per_node * n), zipf)
        adj = [(rng.randint(0, n, size=c).astype(np.int32), rng.randint(0, n, size=c).astype(np.int32)) for c in counts]
        refs = {}
        if refs_per_graph:
            refs["supernodes"] = rng.choice(n, size=min(refs_per_graph, n), replace=False).astype(np.int32)
        graphs.append({"num_nodes": n, "adjacency_lists": adj, "reference_nodes": refs})
    return graphs


def power_law_graph(num_nodes: int, num_edges: int, alpha: float = 0.8, seed: int = 1234,
                    chunk: int = 10_000_000) -> Adj:
    """Config 5: dst ~ power law (w_i = (i+1)^-alpha through a fixed random node permutation),
    src uniform.  Generated in chunks to bound host memory."""
    rng = np.random.RandomState(seed)
    w = np.power(np.arange(1, num_nodes + 1, dtype=np.float64), -alpha)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    node_perm = rng.permutation(num_nodes).astype(np.int64)
    srcs, dsts = [], []
    for lo in range(0, num_edges, chunk):
        m = min(chunk, num_edges - lo)
        d = np.searchsorted(cdf, rng.random_sample(m), side="left").clip(0, num_nodes - 1)
        dsts.append(node_perm[d])
        srcs.append(rng.randint(0, num_nodes, size=m).astype(np.int64))
    return [(torch.from_numpy(np.concatenate(srcs)), torch.from_numpy(np.concatenate(dsts)))]


CONFIGS = {
    # name: description used in bench.py's `config.workload`
    "cfg1_ppi_ggnn": "PPI-like, 24 graphs ~2.4k nodes, 14 raw edges/node, T=3, 1 GGNN layer H=64 (CPU reference case)",
    "cfg2_random_mlp": "synthetic random graph N=200k E=1.1M, T=1, 1 MLP-MP layer H=M=128, sum",
    "cfg3_graph2class_ggnn": "Graph2Class-style batch: 48 graphs ~2.5k nodes (N~120k), T0=8 -> T=17, "
                             "8 GGNN layers H=128 (Typilus GGNN arch), max",
    "cfg4_varmisuse_mlp": "VarMisuse-style batch: 40 graphs ~2k nodes (N~80k), T0=10 -> T=21, 8 MLP-MP layers",
    "cfg5_powerlaw": "power-law graph N=10M E=100M H=256 (8 GPUs; scaled per GPU)",
}
