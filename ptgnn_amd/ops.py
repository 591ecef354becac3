"""Python host wrappers over the C ABI (include/ptgnn_amd.h).

torch is plumbing here: it owns device memory (outputs/workspaces are torch tensors so the caching
allocator and stream semantics are preserved) and supplies the current HIP stream.  All compute
happens in libptgnn_amd.so; there is no eager fallback.
"""
import ctypes
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch

from ptgnn_amd import _lib

REDUCE_IDS = {"sum": 0, "add": 0, "mean": 1, "max": 2, "min": 3}
EPI_NONE, EPI_GELU, EPI_LAYERNORM, EPI_GELU_LAYERNORM = 0, 1, 2, 3
ACT_IDS = {None: 0, "none": 0, "tanh": 1, "relu": 2}


GEMM_MODES = {"tile": 0, "stream": 1}


def set_gemm_mode(mode) -> int:
    """Kernel family of the dense blocks (ptgnn_amd_set_gemm_mode): "tile" = round-1 128x128 tile kernels, "stream" =
    weight-stationary streaming kernels; both exact fp32 MFMA with the same bits.  Returns the previous mode id.
    (The opt-in 3 x bf16 split arithmetic of rounds 2-4 was removed in round 5.)"""
    lib = _lib.load()
    prev = lib.ptgnn_amd_get_gemm_mode()
    if isinstance(mode, str) and mode not in GEMM_MODES:
        raise _lib.PtgnnAmdError(f"unknown GEMM mode {mode!r} (have {sorted(GEMM_MODES)}; the split mode was removed)")
    _lib.check(lib.ptgnn_amd_set_gemm_mode(int(GEMM_MODES.get(mode, mode))), "ptgnn_amd_set_gemm_mode")
    return prev


def get_gemm_mode() -> int:
    return _lib.load().ptgnn_amd_get_gemm_mode()


def launch_counts() -> dict:
    """{kernel family: launches made by this process} (ptgnn_amd_launch_count): tests take differences around a call
    to assert which kernel a shape / size / mode was dispatched to."""
    lib = _lib.load()
    out, i = {}, 0
    while True:
        name = lib.ptgnn_amd_launch_name(i)
        if name is None:
            return out
        out[name.decode()] = int(lib.ptgnn_amd_launch_count(i))
        i += 1


def launches_since(before: dict) -> dict:
    """Kernel families launched since `before = launch_counts()` -> {name: count}, zero entries dropped."""
    now = launch_counts()
    return {k: v - before.get(k, 0) for k, v in now.items() if v != before.get(k, 0)}


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


class KernelTimer:
    """Optional HIP-event bracket around every C-ABI launch (bench.py's live roofline numbers).
    Events are recorded on the stream the kernel is launched on; nothing synchronises until
    `summary()` is called."""

    def __init__(self):
        self.records = []   # (name, start_event, end_event, work dict)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, s, e, work in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "bytes": 0.0, "flops": 0.0})
            d["calls"] += 1
            d["ms"] += s.elapsed_time(e)
            b, f = work.get("bytes", 0.0), work.get("flops", 0.0)
            d["bytes"] += b() if callable(b) else b
            d["flops"] += f() if callable(f) else f
        return out


_TIMER: Optional[KernelTimer] = None


def set_kernel_timer(timer: Optional[KernelTimer]):
    global _TIMER
    _TIMER = timer


class _timed:
    __slots__ = ("name", "work", "s")

    def __init__(self, name, **work):
        self.name, self.work = name, work

    def __enter__(self):
        if _TIMER is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if _TIMER is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            _TIMER.records.append((self.name, self.s, e, self.work))
        return False


def _require_cuda_f32(name: str, t: torch.Tensor, dims: int = 2):
    if not t.is_cuda:
        raise _lib.PtgnnAmdError(
            f"{name} must live on the GPU: the C-ABI wrappers have no CPU path (got device {t.device}; CPU tensors are served "
            "one level up, by the layers' and the facade's plain-torch route)")
    if t.dtype != torch.float32:
        raise _lib.PtgnnAmdError(f"{name} must be float32 (got {t.dtype})")
    if t.dim() != dims:
        raise _lib.PtgnnAmdError(f"{name} must be {dims}-D (got shape {tuple(t.shape)})")


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    """Accept row-major 2-D views with unit inner stride (e.g. column slices); copy otherwise."""
    if t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t
    if t.shape[1] == 1 and t.stride(0) >= 1:
        return t
    return t.contiguous()


def _ptr_or(t: torch.Tensor, stand_in: torch.Tensor) -> int:
    """Device pointer of `t`; a tensor WITHOUT elements (the message table of a minibatch whose edge types are all empty)
    has none, so a stand-in that is never dereferenced (no CSR slot refers to a row of it) is passed instead."""
    return t.data_ptr() if t.numel() else stand_in.data_ptr()


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# ------------------------------------------------------------------------------------------------
# graph plan
# ------------------------------------------------------------------------------------------------
class GraphPlan:
    """Destination-sorted CSR of all edge types of one minibatch (see ptgnn_amd_csr_build).

    rowptr int32 [N+1]; col int32 [E] = (src << type_bits) | type; perm int32 [E] = position of
    the CSR slot's edge in the type-major concatenation of the adjacency lists.
    """

    __slots__ = ("rowptr", "col", "perm", "type_bits", "num_nodes", "num_edges", "num_types",
                 "num_src_rows", "_backward", "_adj", "_adj_refs", "_inv_perm", "_ready", "_waited",
                 "_hub_tickets", "hub_entries", "hub_count", "_slot_rows", "_ident", "_transposed", "_uniq", "_hub_posted",
                 "_has_hubs", "__weakref__")

    def __init__(self, rowptr, col, perm, type_bits, num_nodes, num_edges, num_types):
        self.rowptr, self.col, self.perm = rowptr, col, perm
        self.type_bits, self.num_nodes = type_bits, num_nodes
        self.num_edges, self.num_types = num_edges, num_types
        self.num_src_rows = num_nodes
        self._backward = None
        self._adj = None       # the adjacency tensors the plan was built from (for the backward plan)
        self._adj_refs = None
        self._inv_perm = None
        self._ready = None     # event recorded on the plan stream after the build (None = same stream)
        self._waited = set()
        self._hub_tickets = {}
        self.hub_entries = self.hub_count = None   # (chunk, row) pairs of rows > HUB_THRESHOLD
        self._slot_rows = self._ident = self._transposed = None
        self._uniq = None      # UniqueMessages | pending read-back | False (not worth it / not applicable)
        self._hub_posted = False   # the hub count's asynchronous read-back has been posted (ops.gather_update_supported)
        self._has_hubs = None      # ... and has arrived: True / False (None = not known on the host)

    def may_have_hubs(self) -> bool:
        """Only plans with more edges than the threshold can contain a hub row (whether they do is
        known on the device: `hub_count`)."""
        return self.hub_entries is not None

    def hub_tickets(self, msg_dim: int) -> torch.Tensor:
        """Arrival counters of the hub chunks: zeroed once, left zero by every launch -- and owned by the launches of ONE
        stream at a time (include/ptgnn_amd.h), so they are kept per (size, current stream): two streams that
        aggregate over the same plan concurrently must not count each other's chunks."""
        n = _lib.load().ptgnn_amd_hub_ticket_count(self.num_edges, msg_dim)
        dev = self.rowptr.device
        key = (n, torch.cuda.current_stream(dev).cuda_stream)
        t = self._hub_tickets.get(key)
        if t is None:
            t = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
            self._hub_tickets[key] = t
        return t

    def wait(self) -> None:
        """Make the current stream wait for a plan that was built on the side stream (once per stream).
        Consumers call this right before their first launch that reads rowptr/col/perm."""
        if self._ready is not None:
            cur = torch.cuda.current_stream(self.rowptr.device)
            if cur.cuda_stream not in self._waited:
                cur.wait_event(self._ready)
                self._waited.add(cur.cuda_stream)

    def unique_messages(self) -> Optional["UniqueMessages"]:
        """Message rows of the layers whose message depends on (edge type, source) only (GGNN without edge features /
        per-edge dropout): one row per pair that occurs instead of one per edge -- see ptgnn_amd_unique_sources.  Built
        on the first call behind the plan build and shared by all layers of the minibatch; everything the de-duplicated
        launches need stays on the device (no host synchronisation).  None when it does not apply (small batches, more
        than 64 edge types) or when recent minibatches saved fewer than UNIQUE_MIN_SAVING of their rows."""
        if self._uniq is None:
            self._uniq = _launch_unique_sources(self) or False
        return self._uniq or None

    def backward_plan(self) -> "GraphPlan":
        """Plan of the transposed problem, rows = src * T + type, col = dst: row r of the [N*T, M] view
        of the message-table gradient sums the output gradients of its out-edges.  Built lazily on the
        first backward of a minibatch and shared by all layers."""
        if self._backward is None:
            if self._adj is None:
                raise _lib.PtgnnAmdError("this plan was built without keeping its adjacency lists")
            self.wait()
            self._backward = build_plan(self._adj, self.num_src_rows * self.num_types, mode=2)
        return self._backward

    def forward_slot_of_backward_slot(self) -> torch.Tensor:
        """int32 [E]: for slot i of the backward plan, the forward-plan slot of the same edge."""
        bp = self.backward_plan()
        if bp._inv_perm is None:   # reuse the field on the backward plan as the cache
            inv = self.inverse_perm()
            bp._inv_perm = inv[bp.perm[: self.num_edges].to(torch.int64)].to(torch.int32).contiguous()
        return bp._inv_perm

    def slot_rows(self) -> torch.Tensor:
        """int32 [E]: destination row of every CSR slot (rowptr expanded); built on the first backward
        of a minibatch, shared by all layers."""
        if self._slot_rows is None:
            self.wait()
            E = self.num_edges
            deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
            rows = torch.arange(self.num_nodes, device=self.rowptr.device, dtype=torch.int32)
            self._slot_rows = torch.repeat_interleave(rows, deg, output_size=E) if E > 0 else rows[:0]
        return self._slot_rows

    def identity_index(self) -> List[torch.Tensor]:
        """Per-type views of arange(E) (int64): the "source index" that makes the grouped edge GEMM
        read its A rows in message order (backward of the message Linear)."""
        if self._ident is None:
            if self._adj is None:
                raise _lib.PtgnnAmdError("this plan was built without keeping its adjacency lists")
            ar = torch.arange(max(self.num_edges, 1), device=self.rowptr.device, dtype=torch.int64)
            out, off = [], 0
            for s_, _ in self._adj:
                n = int(s_.shape[0])
                out.append(ar[off: off + n])
                off += n
            self._ident = out
        return self._ident

    def transposed_plan(self) -> "GraphPlan":
        """rows = source node, col/perm over the same message order: segment-sums per-edge input
        gradients back onto the source rows."""
        if self._transposed is None:
            if self._adj is None:
                raise _lib.PtgnnAmdError("this plan was built without keeping its adjacency lists")
            self.wait()
            self._transposed = build_plan(self._adj, self.num_src_rows, mode=1)
        return self._transposed

    def inverse_perm(self) -> torch.Tensor:
        """original edge position -> CSR slot (int64)."""
        if self._inv_perm is None:
            self.wait()
            inv = torch.empty(max(self.num_edges, 1), dtype=torch.int64, device=self.perm.device)
            inv[self.perm[: self.num_edges].to(torch.int64)] = torch.arange(
                self.num_edges, device=self.perm.device)
            self._inv_perm = inv
        return self._inv_perm


class UniqueMessages:
    """slot_row int32 [E]: message row of every CSR slot; unique_src int64: source node of every message row
    (type-major); edge_table: the device-resident launch table of ptgnn_amd_edge_linear_shared_f32; capacity: rows the
    message table must hold; counts: device int64 [T + 1] rows per type and in all."""
    __slots__ = ("slot_row", "unique_src", "edge_table", "capacity", "counts", "num_edges", "num_types", "_host",
                 "_event", "_counts")

    def __init__(self, slot_row, unique_src, edge_table, capacity, counts, num_edges, num_types):
        self.slot_row, self.unique_src, self.edge_table = slot_row, unique_src, edge_table
        self.capacity, self.counts, self.num_edges, self.num_types = capacity, counts, num_edges, num_types
        self._host = self._event = self._counts = None

    def host_counts(self, wait: bool = False) -> Optional[List[int]]:
        """Rows per edge type + the total, once the asynchronous read-back has arrived (None before).  `wait` blocks
        on the read-back's own event -- not on the stream: work enqueued after the bookkeeping keeps running."""
        if self._counts is None and self._event is not None:
            if wait:
                self._event.synchronize()
            if self._event.query():
                self._counts = [int(c) for c in self._host.tolist()]
                _PINNED_FREE.setdefault(int(self._host.numel()), []).append(self._host)
                self._host = self._event = None
        elif self._counts is None and wait:     # built under graph capture: no read-back was posted
            self._counts = [int(c) for c in self.counts.tolist()]
        return self._counts

    def rows(self, wait: bool = False) -> Optional[int]:
        """Rows of the message table (None while the read-back is in flight, unless `wait`)."""
        c = self.host_counts(wait)
        return None if c is None else c[self.num_types]

    def adjacency(self):
        """Per edge type (unique source ids, same): the adjacency input of the host-sized launches (weight gradient,
        tests).  Waits for the row counts."""
        adj, off = [], 0
        for c in self.host_counts(wait=True)[:-1]:
            adj.append((self.unique_src[off: off + c], self.unique_src[off: off + c]))
            off += c
        return adj


# Sharing message rows costs ~6 small launches per minibatch.  Whether it pays is only known afterwards (the row counts
# come back asynchronously): when the minibatches seen so far saved fewer than UNIQUE_MIN_SAVING of their rows, the
# next UNIQUE_BACKOFF plans keep the per-edge form, then one is probed again.
UNIQUE_MIN_SAVING = float(os.environ.get("PTGNN_AMD_UNIQUE_MIN_SAVING", "0.05"))
UNIQUE_MIN_EDGES = int(os.environ.get("PTGNN_AMD_UNIQUE_MIN_EDGES", "65536"))
UNIQUE_BACKOFF = 16
_UNIQ_PENDING: List["UniqueMessages"] = []
_UNIQ_SKIP = [0]
_PINNED_FREE = {}    # words -> pinned int64 buffers not in flight (pinning host memory costs far more than the kernels)


def _pinned_words(n: int) -> torch.Tensor:
    free = _PINNED_FREE.setdefault(n, [])
    return free.pop() if free else torch.empty(n, dtype=torch.int64).pin_memory()


def _poll_unique_stats() -> None:
    for u in list(_UNIQ_PENDING):
        rows = u.rows()
        if rows is not None:
            _UNIQ_PENDING.remove(u)
            if u.num_edges > 0 and rows > (1.0 - UNIQUE_MIN_SAVING) * u.num_edges:
                _UNIQ_SKIP[0] = UNIQUE_BACKOFF
    del _UNIQ_PENDING[:-8]


def _launch_unique_sources(plan: "GraphPlan") -> Optional[UniqueMessages]:
    lib = _lib.load()
    E, T, ns = plan.num_edges, plan.num_types, plan.num_src_rows
    if plan._adj is None or E < UNIQUE_MIN_EDGES or T > 64:
        return None
    capturing = torch.cuda.is_current_stream_capturing()
    if not capturing:
        _poll_unique_stats()
        if _UNIQ_SKIP[0] > 0:
            _UNIQ_SKIP[0] -= 1
            return None
    plan.wait()
    dev = plan.col.device
    cap = max(1, min(E, ns * T))
    slot_row = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    unique_src = torch.empty(cap, dtype=torch.int64, device=dev)
    counts = torch.empty(T + 1, dtype=torch.int64, device=dev)
    table = torch.empty(int(lib.ptgnn_amd_edge_table_bytes()), dtype=torch.uint8, device=dev)
    ws_bytes = int(lib.ptgnn_amd_unique_sources_workspace_bytes(ns, T))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    with _timed("unique_sources", bytes=E * 12.0 + ns * T / 4.0):
        rc = lib.ptgnn_amd_unique_sources(plan.col.data_ptr(), E, plan.type_bits, T, ns, slot_row.data_ptr(),
                                          unique_src.data_ptr(), cap, counts.data_ptr(), table.data_ptr(),
                                          ws.data_ptr(), ws_bytes, _stream(slot_row))
    _lib.check(rc, "ptgnn_amd_unique_sources")
    u = UniqueMessages(slot_row, unique_src, table, cap, counts, E, T)
    if not capturing:
        u._host = _pinned_words(T + 1)
        with torch.cuda.device(dev):
            u._host.copy_(counts, non_blocking=True)
            u._event = torch.cuda.Event()
            u._event.record(torch.cuda.current_stream(dev))
        _UNIQ_PENDING.append(u)
    return u


def edge_linear_shared_supported(state_dim: int, msg_dim: int, num_types: int) -> bool:
    return bool(_lib.load().ptgnn_amd_edge_linear_shared_supported(state_dim, msg_dim, num_types))


def edge_linear_shared(x: torch.Tensor, uniq: UniqueMessages, weights: Sequence[torch.Tensor],
                       act: Optional[str] = None) -> torch.Tensor:
    """msg[r] = act(W_t x[unique_src[r]]) over the message rows of `uniq` (GraphPlan.unique_messages): the grouped
    per-edge GEMM of `edge_linear` with one row per distinct (edge type, source) pair; rows beyond the table's count
    are not written.  The launch geometry comes from the device-resident table: no host synchronisation."""
    lib = _lib.load()
    _require_cuda_f32("x", x)
    x = _rowmajor(x)
    T, H, M = uniq.num_types, x.shape[1], weights[0].shape[0]
    if len(weights) != T:
        raise _lib.PtgnnAmdError(f"edge_linear_shared: {len(weights)} weights for {T} edge types")
    ws = [w.detach().contiguous() for w in weights]
    for w in ws:
        if tuple(w.shape) != (M, H) or not w.is_cuda or w.dtype != torch.float32:
            raise _lib.PtgnnAmdError(f"edge_linear_shared: weight shape {tuple(w.shape)} does not match [{M}, {H}]")
    msg = torch.empty(uniq.capacity, M, dtype=torch.float32, device=x.device)
    wp = (ctypes.c_void_p * T)(*[w.data_ptr() for w in ws])

    def rows():                       # resolved when the timer is summarised: the count is back by then
        r = uniq.rows(wait=True)
        return float(r if r is not None else uniq.num_edges)
    with _timed("edge_linear_shared", flops=lambda: 2.0 * rows() * H * M,
                bytes=lambda: 4.0 * (rows() * H + rows() * M + T * M * H) + 8.0 * rows()):
        rc = lib.ptgnn_amd_edge_linear_shared_f32(x.data_ptr(), _ld(x), x.shape[0], H, uniq.edge_table.data_ptr(),
                                                  ctypes.cast(wp, ctypes.c_void_p), T, M, ACT_IDS[act],
                                                  msg.data_ptr(), M, _stream(msg))
    _lib.check(rc, "ptgnn_amd_edge_linear_shared_f32")
    return msg


# ------------------------------------------------------------------------------------------------
# index range guard
# ------------------------------------------------------------------------------------------------
# The reference device-asserts on an out-of-range node id (F.embedding, gatedmessagepassing.py:54-56).
# Here the plan build clamps such ids to row 0 (nothing is ever read or written out of bounds) and counts
# them in a per-device accumulator; the count travels back through a pinned host word WITHOUT a sync and
# is looked at on later plan builds (or on demand: `check_indices(sync=True)`).  A non-zero count raises
# PtgnnAmdError: the results of the offending minibatch are garbage, like the reference's would be.
_BAD = {}    # device index -> {"dev": int32[1] accumulator, "host": pinned int32[1], "event": Event | None}
VALIDATE_INDICES = os.environ.get("PTGNN_AMD_VALIDATE", "async")   # "async" | "sync" | "off"


def _bad_state(device):
    key = torch.device(device).index or 0
    st = _BAD.get(key)
    if st is None:
        st = {"dev": torch.zeros(1, dtype=torch.int32, device=device),
              "host": torch.zeros(1, dtype=torch.int32).pin_memory(), "event": None}
        _BAD[key] = st
    return st


def _raise_bad(st, count: int):
    st["dev"].zero_()
    st["host"].zero_()
    st["event"] = None
    raise _lib.PtgnnAmdError(
        f"{count} node id(s) outside [0, num_nodes) reached the graph plan build (adjacency lists / scatter "
        "index / dim_size too small). They were clamped to row 0 so no memory was touched out of bounds, but "
        "the outputs of that minibatch are wrong.")


def check_indices(device=None, sync: bool = False) -> None:
    """Raise if a plan build on `device` saw an out-of-range node id.  sync=False only looks at read-backs
    that have already completed (no host-device synchronisation)."""
    for key, st in list(_BAD.items()):
        if device is not None and (torch.device(device).index or 0) != key:
            continue
        if sync:
            n = int(st["dev"].item())
            if n:
                _raise_bad(st, n)
            continue
        ev = st["event"]
        if ev is not None and ev.query():
            st["event"] = None
            n = int(st["host"][0])
            if n:
                _raise_bad(st, n)


def _post_plan_readback(st) -> None:
    if st["event"] is None:   # one read-back in flight at a time
        dev = st["dev"].device
        with torch.cuda.device(dev):     # the copy and its event belong to the stream of the plan's device
            st["host"].copy_(st["dev"], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        st["event"] = ev


_PLAN_CONTROL = {}   # (device index, stream handle) -> zero-at-rest control block of the plan build


def _plan_control(dev: torch.device) -> torch.Tensor:
    """The control block ptgnn_amd_csr_build wants (include/ptgnn_amd.h): zero-filled once, then owned by the
    builds of ONE stream -- stream order serialises them and every build leaves it zero-filled.  Under graph
    capture a fresh zero-filled block is captured with the build instead (a cached one could be shared with
    eager builds that run while the graph replays)."""
    nbytes = int(_lib.load().ptgnn_amd_csr_control_bytes())
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ctl = _PLAN_CONTROL.get(key)
    if ctl is None:
        ctl = _PLAN_CONTROL[key] = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    return ctl


def build_plan(adjacency_lists: Sequence[Tuple[torch.Tensor, torch.Tensor]], num_nodes: int,
               transposed: bool = False, want_perm: bool = True,
               num_src_rows: Optional[int] = None, mode: Optional[int] = None) -> GraphPlan:
    """One stable sort per minibatch; reused by every layer of the forward.  mode: 0 forward plan
    (rows = dst), 1 transposed (rows = src), 2 backward plan (rows = src * T + type, col = dst;
    `num_nodes` must then be source rows * T)."""
    if mode is None:
        mode = 1 if transposed else 0
    lib = _lib.load()
    T = len(adjacency_lists)
    if T == 0:
        raise _lib.PtgnnAmdError("build_plan: at least one edge type is required")
    dev = None
    srcs, dsts, counts = [], [], []
    for t, (s, d) in enumerate(adjacency_lists):
        if not (s.is_cuda and d.is_cuda):
            raise _lib.PtgnnAmdError("build_plan: adjacency lists must be CUDA tensors (no CPU path)")
        if s.dtype != torch.int64 or d.dtype != torch.int64:
            raise _lib.PtgnnAmdError("build_plan: adjacency lists must be int64 "
                                     "(GraphNeuralNetworkModel.finalize_minibatch layout)")
        if s.dim() != 1 or s.shape != d.shape:
            raise _lib.PtgnnAmdError(f"build_plan: edge type {t}: src/dst must be equal-length 1-D")
        s, d = s.contiguous(), d.contiguous()
        dev = s.device if dev is None else dev
        srcs.append(s)
        dsts.append(d)
        counts.append(int(s.shape[0]))
    E = sum(counts)
    type_bits = lib.ptgnn_amd_type_bits(T)
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    perm = torch.empty(max(E, 1), dtype=torch.int32, device=dev) if want_perm else None
    ws_bytes = lib.ptgnn_amd_csr_workspace_bytes(E, num_nodes)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    control = _plan_control(dev)
    hub_entries = hub_count = None
    if HUB_THRESHOLD > 0 and E > HUB_THRESHOLD:
        hub_entries = torch.empty(2 * ((E + 1023) // 1024), 2, dtype=torch.int32, device=dev)
        hub_count = torch.empty(1, dtype=torch.int32, device=dev)
    PtrArr, CntArr = ctypes.c_void_p * T, ctypes.c_int64 * T
    src_ptrs = PtrArr(*[s.data_ptr() if s.numel() else None for s in srcs])
    dst_ptrs = PtrArr(*[d.data_ptr() if d.numel() else None for d in dsts])
    cnts = CntArr(*counts)
    bad = None
    capturing = torch.cuda.is_current_stream_capturing()
    if VALIDATE_INDICES != "off":
        bad = _bad_state(dev)
        if not capturing:
            check_indices(dev)      # surfaces an earlier minibatch's bad ids (never blocks)
    # algorithmic bytes: read 16 B/edge (int64 src+dst), write 4 B/edge col (+4 perm) + rowptr
    with _timed("csr_build", bytes=E * (16 + 4 + (4 if want_perm else 0)) + 4.0 * (num_nodes + 1)):
        rc = lib.ptgnn_amd_csr_build(ctypes.cast(src_ptrs, ctypes.c_void_p),
                                     ctypes.cast(dst_ptrs, ctypes.c_void_p),
                                     ctypes.cast(cnts, ctypes.c_void_p), T, num_nodes,
                                     int(num_src_rows or 0),
                                     mode, rowptr.data_ptr(), col.data_ptr(),
                                     perm.data_ptr() if perm is not None else None, None,
                                     HUB_THRESHOLD if hub_entries is not None else 0,
                                     hub_entries.data_ptr() if hub_entries is not None else None,
                                     hub_count.data_ptr() if hub_count is not None else None,
                                     bad["dev"].data_ptr() if bad is not None else None,
                                     control.data_ptr(),
                                     ws.data_ptr(), ws_bytes, _stream(rowptr))
    if rc != 0:
        _PLAN_CONTROL.clear()       # a build that stopped half-way may have left its control block non-zero
    _lib.check(rc, "ptgnn_amd_csr_build")
    if bad is not None:
        if capturing:
            pass                    # a captured plan build keeps counting; read-backs resume outside the graph
        elif VALIDATE_INDICES == "sync":
            check_indices(dev, sync=True)
        else:
            _post_plan_readback(bad)
    # `ws`, `srcs`, `dsts` are stream-ordered: torch's caching allocator only hands their memory to
    # later work on the same stream, so dropping the references here is safe.
    # col/perm keep >= 1 element so their base pointer is never null (E == 0 batches are legal)
    plan = GraphPlan(rowptr, col, perm, 0 if mode == 2 else type_bits, num_nodes, E, T)
    plan.hub_entries, plan.hub_count = hub_entries, hub_count
    if mode == 0:
        plan.num_src_rows = int(num_src_rows or num_nodes)
        plan._adj = list(zip(srcs, dsts))
    return plan


def shard_index(adjacency_lists: Sequence[Tuple[torch.Tensor, torch.Tensor]], lo: int, hi: int,
                bounds: torch.Tensor, total_nodes: int):
    """ptgnn_amd_shard_index: (local_src [E], local_dst [E], need_ids buffer, stats) of a dst-range shard whose
    edges are given in global ids; see include/ptgnn_amd.h.  No host synchronisation."""
    lib = _lib.load()
    T = len(adjacency_lists)
    srcs = [a[0].contiguous() for a in adjacency_lists]
    dsts = [a[1].contiguous() for a in adjacency_lists]
    for s_, d_ in zip(srcs, dsts):
        if not (s_.is_cuda and d_.is_cuda and s_.dtype == torch.int64 and d_.dtype == torch.int64 and s_.shape == d_.shape):
            raise _lib.PtgnnAmdError("shard_index: adjacency lists must be equal-length CUDA int64 tensors")
    dev = srcs[0].device
    counts = [int(s_.shape[0]) for s_ in srcs]
    E = sum(counts)
    world = int(bounds.shape[0]) - 1
    local_src = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
    local_dst = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
    cap = max(0, min(E, total_nodes - (hi - lo)))
    need = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
    stats = torch.empty(world + 2 + T, dtype=torch.int64, device=dev)
    ws_bytes = int(lib.ptgnn_amd_shard_index_workspace_bytes(total_nodes))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    PtrArr, CntArr = ctypes.c_void_p * T, ctypes.c_int64 * T
    sp = PtrArr(*[s_.data_ptr() if s_.numel() else None for s_ in srcs])
    dp = PtrArr(*[d_.data_ptr() if d_.numel() else None for d_ in dsts])
    cn = CntArr(*counts)
    # global source ids outside [0, total_nodes) are counted like the plan build's bad ids (after the remap they are
    # ordinary own / halo rows, which that guard can no longer see)
    bad = _bad_state(dev) if VALIDATE_INDICES != "off" else None
    with _timed("shard_index", bytes=E * 32.0 + total_nodes / 4.0):
        rc = lib.ptgnn_amd_shard_index(ctypes.cast(sp, ctypes.c_void_p), ctypes.cast(dp, ctypes.c_void_p),
                                       ctypes.cast(cn, ctypes.c_void_p), T, int(lo), int(hi), bounds.data_ptr(), world,
                                       int(total_nodes), local_src.data_ptr(), local_dst.data_ptr(), need.data_ptr(),
                                       cap, stats.data_ptr(), bad["dev"].data_ptr() if bad is not None else None,
                                       ws.data_ptr(), ws_bytes, _stream(local_src))
    _lib.check(rc, "ptgnn_amd_shard_index")
    if bad is not None and not torch.cuda.is_current_stream_capturing():
        if VALIDATE_INDICES == "sync":
            check_indices(dev, sync=True)
        else:
            _post_plan_readback(bad)
    return local_src[:E], local_dst[:E], counts, need, stats


def plan_from_sorted_index(index: torch.Tensor, num_segments: int) -> GraphPlan:
    """Plan of a segment reduce whose index is already sorted (node_to_graph_idx of a disjoint-union
    batch): no sort -- rowptr is a searchsorted over the index, col/perm are the identity."""
    if not index.is_cuda or index.dtype != torch.int64 or index.dim() != 1:
        raise _lib.PtgnnAmdError("plan_from_sorted_index: expected a 1-D CUDA int64 index")
    n = int(index.shape[0])
    bounds = torch.arange(num_segments + 1, device=index.device, dtype=torch.int64)
    rowptr = torch.searchsorted(index, bounds).to(torch.int32)
    ident = torch.arange(max(n, 1), device=index.device, dtype=torch.int32)
    plan = GraphPlan(rowptr, ident, ident, 0, num_segments, n, 1)
    if HUB_THRESHOLD > 0 and n > HUB_THRESHOLD:   # graphs are long segments: reuse the hub machinery
        lib = _lib.load()
        plan.hub_entries = torch.empty(2 * ((n + 1023) // 1024), 2, dtype=torch.int32, device=index.device)
        plan.hub_count = torch.zeros(1, dtype=torch.int32, device=index.device)
        rc = lib.ptgnn_amd_hub_list(rowptr.data_ptr(), num_segments, HUB_THRESHOLD,
                                    plan.hub_entries.data_ptr(), plan.hub_count.data_ptr(), _stream(rowptr))
        _lib.check(rc, "ptgnn_amd_hub_list")
    return plan


_PLAN_CACHE: List[GraphPlan] = []
_PLAN_CACHE_SIZE = 4

# Rows longer than this are reduced chunk-parallel by the hub kernels (see gather_reduce.hip).
HUB_THRESHOLD = 2048

# The plan (sort) is latency-bound integer work and the first dense block of a layer (pre-transform /
# per-edge GEMM) does not read it, so the build CAN run on a side HIP stream under that GEMM, the
# aggregation kernel waiting on the plan's event (PTGNN_AMD_OVERLAP_PLAN=1).
# Off: measured again in round 2 with the streaming GEMMs (one 8-wave workgroup per CU, so the plan kernels do fit
# beside them): cfg2 0.411 -> 0.456 ms per step, cfg3 4.34 -> 4.39 ms -- the co-resident plan workgroups take LDS
# bandwidth and issue slots from the MFMA kernel for longer than the plan build lasts on an idle chip.
OVERLAP_PLAN_BUILD = os.environ.get("PTGNN_AMD_OVERLAP_PLAN", "0") not in ("", "0")
_SIDE_STREAMS = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE_STREAMS[key] = st
    return st


def _build_plan_overlapped(adjacency_lists, num_nodes: int) -> GraphPlan:
    dev = adjacency_lists[0][0].device
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    side.wait_stream(main)                      # the adjacency tensors' producers ran on `main`
    with torch.cuda.stream(side):
        plan = build_plan(adjacency_lists, num_nodes)
        plan._ready = torch.cuda.Event()
        plan._ready.record(side)
    for s, d in adjacency_lists:                # read by side-stream kernels: defer allocator reuse
        s.record_stream(side)
        d.record_stream(side)
    for t in (plan.rowptr, plan.col, plan.perm, plan.hub_entries, plan.hub_count):  # alloc on `side`, used on `main`
        if t is not None:
            t.record_stream(main)
    return plan


def plan_for(adjacency_lists: Sequence[Tuple[torch.Tensor, torch.Tensor]], num_nodes: int) -> GraphPlan:
    """Plan lookup keyed on the *identity and version* of the adjacency tensors, so the L layers of
    one forward (which all receive the same tensors, graphneuralnetwork.py:122-131) share one
    sort.  Weak references guarantee a freed-and-reallocated tensor can never alias a stale plan."""
    for plan in _PLAN_CACHE:
        refs = plan._adj_refs
        if plan.num_nodes != num_nodes or len(refs) != len(adjacency_lists):
            continue
        ok = True
        for (rs, vs, rd, vd), (s, d) in zip(refs, adjacency_lists):
            if rs() is not s or rd() is not d or s._version != vs or d._version != vd:
                ok = False
                break
        if ok:
            return plan
    plan = _build_plan_overlapped(adjacency_lists, num_nodes) if OVERLAP_PLAN_BUILD else \
        build_plan(adjacency_lists, num_nodes)
    plan._adj_refs = [(weakref.ref(s), s._version, weakref.ref(d), d._version)
                      for s, d in adjacency_lists]
    _PLAN_CACHE.insert(0, plan)
    del _PLAN_CACHE[_PLAN_CACHE_SIZE:]
    return plan


def clear_plan_cache():
    del _PLAN_CACHE[:]


# ------------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------------
def _hub_workspace(plan: GraphPlan, msg_dim: int, with_arg: bool, device):
    """Chunk-partial buffer for the hub kernels, or (None, 0) when the plan is known hub-free."""
    if not plan.may_have_hubs():
        return None, 0
    nbytes = _lib.load().ptgnn_amd_hub_workspace_bytes(plan.num_edges, msg_dim, 1 if with_arg else 0)
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def gather_reduce(ysrc: torch.Tensor, plan: GraphPlan, msg_dim: int, reduce: str,
                  ydst: Optional[torch.Tensor] = None, epilogue: int = EPI_NONE,
                  ln_weight: Optional[torch.Tensor] = None, ln_bias: Optional[torch.Tensor] = None,
                  ln_eps: float = 1e-5, return_arg: bool = False, type_bits: Optional[int] = None,
                  col: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                  rows: Optional[Tuple[int, int]] = None):
    """out[v] = EPI(reduce_{slots of v} ysrc[src, t*M:(t+1)*M] (+ ydst[v, t*M:(t+1)*M])).
    `out`: optional caller-owned [num_nodes, msg_dim] fp32 destination (e.g. one half of a stacked buffer).
    `rows` = (lo, hi): only the destination rows [lo, hi) are computed and written (needs `out`, no arg)."""
    lib = _lib.load()
    _require_cuda_f32("ysrc", ysrc)
    ysrc = _rowmajor(ysrc)
    ld_y = _ld(ysrc)
    ld_yd = ld_y
    if ydst is not None:
        _require_cuda_f32("ydst", ydst)
        ydst = _rowmajor(ydst)
        ld_yd = _ld(ydst)
    if reduce not in REDUCE_IDS:
        raise ValueError(f"unknown aggregation function {reduce!r}")
    N = plan.num_nodes
    caller_out = out
    if out is None:
        out = torch.empty(N, msg_dim, dtype=torch.float32, device=ysrc.device)
    elif tuple(out.shape) != (N, msg_dim) or out.dtype != torch.float32 or not out.is_contiguous():
        raise _lib.PtgnnAmdError(f"gather_reduce: `out` must be a contiguous float32 [{N}, {msg_dim}] tensor")
    arg = None
    if return_arg:
        arg = torch.empty(N, msg_dim, dtype=torch.int32, device=ysrc.device)
    if epilogue & EPI_LAYERNORM:
        ln_weight, ln_bias = ln_weight.contiguous(), ln_bias.contiguous()
    plan.wait()
    tb = plan.type_bits if type_bits is None else type_bits
    colt = plan.col if col is None else col
    # algorithmic bytes (SURVEY.md 8d "(L)"): per edge one message row + its col entry; per node the
    # output row, the rowptr entry and (MLP-MP) the destination-term row; gathers get no cache credit
    nbytes = (plan.num_edges * (4.0 * msg_dim + 4) + N * (4.0 * msg_dim + 4)
              + (N * 4.0 * msg_dim if ydst is not None else 0.0)
              + (N * 4.0 * msg_dim if arg is not None else 0.0))
    lo, hi = (0, N) if rows is None else (int(rows[0]), int(rows[1]))
    if rows is not None:
        if caller_out is None or return_arg:
            raise _lib.PtgnnAmdError("gather_reduce: a row range needs a caller-owned `out` (the other rows are not "
                                     "written) and returns no arg")
        nbytes *= (hi - lo) / max(N, 1)
    hub_ws, hub_bytes = _hub_workspace(plan, msg_dim, arg is not None, ysrc.device)
    with _timed("gather_reduce", bytes=nbytes):
        rc = lib.ptgnn_amd_gather_reduce_rows_f32(
            _ptr_or(ysrc, plan.rowptr), ld_y, ydst.data_ptr() if ydst is not None else None, ld_yd,
            plan.rowptr.data_ptr(), colt.data_ptr(), tb, N, msg_dim,
            REDUCE_IDS[reduce], epilogue,
            ln_weight.data_ptr() if ln_weight is not None else None,
            ln_bias.data_ptr() if ln_bias is not None else None, float(ln_eps),
            out.data_ptr(), msg_dim, arg.data_ptr() if arg is not None else None,
            plan.num_edges, HUB_THRESHOLD if hub_ws is not None else 0,
            plan.hub_entries.data_ptr() if hub_ws is not None else None,
            plan.hub_count.data_ptr() if hub_ws is not None else None,
            hub_ws.data_ptr() if hub_ws is not None else None, hub_bytes,
            plan.hub_tickets(msg_dim).data_ptr() if hub_ws is not None else None, lo, hi, _stream(out))
    _lib.check(rc, "ptgnn_amd_gather_reduce_rows_f32")
    return (out, arg) if return_arg else out


# The fused aggregation + node update serves minibatch-sized plans (every row folds serially: no hub launches)
GATHER_UPDATE = os.environ.get("PTGNN_AMD_GATHER_UPDATE", "1") not in ("", "0")
GATHER_UPDATE_MAX_EDGES = 1 << 21


# ... and plans without hub rows: the fused kernel folds every row serially, so a row of > HUB_THRESHOLD in-edges (which the
# unfused aggregation splits over chunk workgroups) would set its duration.  Whether a plan has such rows is known on the
# device only (`plan.hub_count`); it is read back asynchronously once per plan, and a non-zero count sends the next
# GATHER_UPDATE_BACKOFF calls to the unfused pair (both forms are exact: this is a speed decision, never a correctness one).
GATHER_UPDATE_BACKOFF = 64
_HUB_PENDING: List[Tuple["torch.cuda.Event", torch.Tensor, "weakref.ref"]] = []
_HUB_SKIP = [0]


def _poll_hub_counts() -> None:
    for item in list(_HUB_PENDING):
        ev, host, plan_ref = item
        if ev.query():
            _HUB_PENDING.remove(item)
            hubs = int(host[0]) > 0
            plan = plan_ref()
            if plan is not None:
                plan._has_hubs = hubs        # a fact of THIS plan: decides every later call over it
            if hubs:
                _HUB_SKIP[0] = GATHER_UPDATE_BACKOFF
            _PINNED_FREE.setdefault(1, []).append(host)
    del _HUB_PENDING[:-16]


def gather_update_supported(msg_dim: int, out_dim: int, plan: "GraphPlan") -> bool:
    """Whether the fused aggregation + update launch serves this call.  Callers evaluate their cheaper conditions
    (gradients needed, dropout) FIRST: a True here may consume one step of the back-off below."""
    if not (GATHER_UPDATE and plan.num_edges < GATHER_UPDATE_MAX_EDGES
            and bool(_lib.load().ptgnn_amd_gather_update_supported(int(msg_dim), int(out_dim)))):
        return False
    if plan.hub_count is None:              # too few edges for a hub row to exist
        return True
    if plan._has_hubs is not None:          # the plan's own count has come back (a cached plan of full-graph inference)
        return not plan._has_hubs
    if torch.cuda.is_current_stream_capturing():
        return _HUB_SKIP[0] == 0
    _poll_hub_counts()
    if plan._has_hubs is not None:
        return not plan._has_hubs
    if not plan._hub_posted:
        plan._hub_posted = True
        plan.wait()
        dev = plan.hub_count.device
        host = _pinned_words(1)
        with torch.cuda.device(dev):
            host.copy_(plan.hub_count.to(torch.int64), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        _HUB_PENDING.append((ev, host, weakref.ref(plan)))
    # this plan's count is still in flight: go by what the recent plans reported
    if _HUB_SKIP[0] > 0:
        _HUB_SKIP[0] -= 1
        return False
    return True


def gather_update(msgs: torch.Tensor, plan: GraphPlan, reduce: str, col: torch.Tensor, type_bits: int, epilogue: int,
                  ln_weight: Optional[torch.Tensor], ln_bias: Optional[torch.Tensor], ln_eps: float,
                  weight: torch.Tensor, bias: Optional[torch.Tensor], act: Optional[str],
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(W . EPI(reduce_{slots of v} msgs[col >> type_bits]) + b) in ONE launch (ptgnn_amd_gather_update_f32): the
    aggregation of `gather_reduce` and the Linear of `linear`, same bits, without the [N, M] aggregate in memory."""
    lib = _lib.load()
    _require_cuda_f32("msgs", msgs)
    _require_cuda_f32("weight", weight)
    msgs, weight = _rowmajor(msgs), weight.contiguous()
    N, M, out_dim = plan.num_nodes, weight.shape[1], weight.shape[0]      # msgs: [E, M] rows or a [rows, T * M] table
    if msgs.shape[1] % M != 0 or reduce not in REDUCE_IDS:
        raise _lib.PtgnnAmdError(f"gather_update: weight {tuple(weight.shape)} / reduce {reduce!r} do not fit messages of "
                                 f"width {msgs.shape[1]}")
    if out is None:
        out = torch.empty(N, out_dim, dtype=torch.float32, device=msgs.device)
    elif tuple(out.shape) != (N, out_dim) or out.dtype != torch.float32 or out.stride(1) != 1 or not out.is_cuda:
        raise _lib.PtgnnAmdError(f"gather_update: `out` must be a float32 CUDA [{N}, {out_dim}] view with unit inner stride")
    if epilogue & EPI_LAYERNORM:
        ln_weight, ln_bias = ln_weight.contiguous(), ln_bias.contiguous()
    if bias is not None:
        bias = bias.contiguous()
    plan.wait()
    E = plan.num_edges
    with _timed("gather_update", bytes=E * (4.0 * M + 4) + N * (4.0 * out_dim + 4) + 4.0 * out_dim * M,
                flops=2.0 * N * M * out_dim):
        rc = lib.ptgnn_amd_gather_update_f32(
            _ptr_or(msgs, plan.rowptr), _ld(msgs), plan.rowptr.data_ptr(), col.data_ptr(), int(type_bits), N, M, REDUCE_IDS[reduce],
            int(epilogue), ln_weight.data_ptr() if epilogue & EPI_LAYERNORM else None,
            ln_bias.data_ptr() if epilogue & EPI_LAYERNORM else None, float(ln_eps), weight.data_ptr(),
            bias.data_ptr() if bias is not None else None, out_dim, ACT_IDS[act], out.data_ptr(), _ld(out), _stream(out))
    _lib.check(rc, "ptgnn_amd_gather_update_f32")
    return out


def gather_reduce_masked(grad: torch.Tensor, arg: torch.Tensor, bplan: GraphPlan,
                         slot_of: torch.Tensor, msg_dim: int) -> torch.Tensor:
    """out[r] = sum_{i in row r} [arg[col_i] == slot_of[i]] * grad[col_i] over a backward plan."""
    lib = _lib.load()
    _require_cuda_f32("grad", grad)
    grad = _rowmajor(grad)
    bplan.wait()
    out = torch.empty(bplan.num_nodes, msg_dim, dtype=torch.float32, device=grad.device)
    hub_ws, hub_bytes = _hub_workspace(bplan, msg_dim, False, grad.device)
    with _timed("gather_reduce_masked", bytes=bplan.num_edges * (8.0 * msg_dim + 8) + bplan.num_nodes * (4.0 * msg_dim + 4)):
        rc = lib.ptgnn_amd_gather_reduce_masked_f32(_ptr_or(grad, bplan.rowptr), _ld(grad), _ptr_or(arg, bplan.rowptr),
                                                    bplan.rowptr.data_ptr(), bplan.col.data_ptr(),
                                                    slot_of.data_ptr(), bplan.num_nodes, msg_dim,
                                                    out.data_ptr(), msg_dim, bplan.num_edges,
                                                    HUB_THRESHOLD if hub_ws is not None else 0,
                                                    bplan.hub_entries.data_ptr() if hub_ws is not None else None,
                                                    bplan.hub_count.data_ptr() if hub_ws is not None else None,
                                                    hub_ws.data_ptr() if hub_ws is not None else None,
                                                    hub_bytes,
                                                    bplan.hub_tickets(msg_dim).data_ptr() if hub_ws is not None else None,
                                                    _stream(out))
    _lib.check(rc, "ptgnn_amd_gather_reduce_masked_f32")
    return out


def segment_reduce(messages: torch.Tensor, plan: GraphPlan, reduce: str, return_arg: bool = False):
    """The torch_scatter seam over a plan: messages [E, D] are in the type-major concatenation
    order of the adjacency lists (abstractmessagepassing.py:38-50)."""
    _require_cuda_f32("messages", messages)
    if messages.shape[0] != plan.num_edges:
        raise _lib.PtgnnAmdError("segment_reduce: messages rows != number of edges in the plan")
    if plan.perm is None:
        raise _lib.PtgnnAmdError("segment_reduce: plan was built without perm")
    return gather_reduce(messages, plan, messages.shape[1], reduce, return_arg=return_arg,
                         type_bits=0, col=plan.perm)


def segment_mul(messages: torch.Tensor, plan: GraphPlan) -> torch.Tensor:
    """reduce="mul" of the torch_scatter seam over a plan (ptgnn_amd_segment_mul_f32): per destination row the product
    of its messages in edge order; rows without in-edges are 1, like torch_scatter's scatter_mul."""
    lib = _lib.load()
    _require_cuda_f32("messages", messages)
    if messages.shape[0] != plan.num_edges:
        raise _lib.PtgnnAmdError("segment_mul: messages rows != number of edges in the plan")
    if plan.perm is None:
        raise _lib.PtgnnAmdError("segment_mul: plan was built without perm")
    plan.wait()
    msg = _rowmajor(messages)
    n, d, E = plan.num_nodes, messages.shape[1], plan.num_edges
    out = torch.empty(n, d, dtype=torch.float32, device=messages.device)
    with _timed("segment_mul", bytes=E * (4.0 * d + 4) + n * 4.0 * d):
        rc = lib.ptgnn_amd_segment_mul_f32(msg.data_ptr() if E > 0 else None, _ld(msg) if E > 0 else d,
                                           plan.rowptr.data_ptr(), plan.perm.data_ptr(), n, E, d, out.data_ptr(), d,
                                           _stream(out))
    _lib.check(rc, "ptgnn_amd_segment_mul_f32")
    return out


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           act: Optional[str] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = act(x W^T + b) on fp32 MFMA.  weight is nn.Linear layout [n_out, k]."""
    lib = _lib.load()
    _require_cuda_f32("x", x)
    _require_cuda_f32("weight", weight)
    x, weight = _rowmajor(x), weight.contiguous()
    rows, k = x.shape
    n_out = weight.shape[0]
    if weight.shape[1] != k:
        raise _lib.PtgnnAmdError(f"linear: x has {k} columns but weight expects {weight.shape[1]}")
    if out is None:
        out = torch.empty(rows, n_out, dtype=torch.float32, device=x.device)
    if bias is not None:
        bias = bias.contiguous()
    with _timed("linear", flops=2.0 * rows * k * n_out, bytes=4.0 * (rows * k + n_out * k + rows * n_out)):
        rc = lib.ptgnn_amd_linear_f32(x.data_ptr(), rows, k, _ld(x), weight.data_ptr(), n_out,
                                      bias.data_ptr() if bias is not None else None, ACT_IDS[act],
                                      out.data_ptr(), _ld(out), _stream(out))
    _lib.check(rc, "ptgnn_amd_linear_f32")
    return out


def linear_add(x: torch.Tensor, weight: torch.Tensor, addend: torch.Tensor, bias: Optional[torch.Tensor] = None,
               act: Optional[str] = None) -> torch.Tensor:
    """act(x W^T + b) + addend.  One launch where the streaming GEMM takes the shape (the add rides its store epilogue:
    ptgnn_amd_linear_add_f32), else the GEMM followed by torch's add -- the same sum either way."""
    lib = _lib.load()
    _require_cuda_f32("x", x)
    _require_cuda_f32("weight", weight)
    _require_cuda_f32("addend", addend)
    x, weight, addend = _rowmajor(x), weight.contiguous(), _rowmajor(addend)
    rows, k = x.shape
    n_out = weight.shape[0]
    if weight.shape[1] != k or tuple(addend.shape) != (rows, n_out):
        raise _lib.PtgnnAmdError(f"linear_add: shapes x {tuple(x.shape)}, weight {tuple(weight.shape)}, addend "
                                 f"{tuple(addend.shape)} do not agree")
    out = torch.empty(rows, n_out, dtype=torch.float32, device=x.device)
    if bias is not None:
        bias = bias.contiguous()
    with _timed("linear", flops=2.0 * rows * k * n_out, bytes=4.0 * (rows * k + n_out * k + 2 * rows * n_out)):
        rc = lib.ptgnn_amd_linear_add_f32(x.data_ptr(), rows, k, _ld(x), weight.data_ptr(), n_out,
                                          bias.data_ptr() if bias is not None else None, ACT_IDS[act],
                                          addend.data_ptr(), _ld(addend), out.data_ptr(), _ld(out), _stream(out))
    if rc == _lib.EUNSUPPORTED:
        return linear(x, weight, bias, act=act, out=out).add_(addend)
    _lib.check(rc, "ptgnn_amd_linear_add_f32")
    return out


def dropout_bitmask(rows: int, width: int, p: float, seed: int, device) -> Optional[torch.Tensor]:
    """Keep mask of the per-edge dropout as one bit per element: int32 [rows, width / 32], bit b of word c = column
    32 c + b (ptgnn_amd_dropout_bitmask; the same hash of (seed, row, column) the seed-taking entry points evaluate).
    None when `width` is not a multiple of 32 (callers then stay on the hash form)."""
    if width % 32 != 0 or rows <= 0 or p <= 0.0:
        return None
    lib = _lib.load()
    bits = torch.empty(rows, width // 32, dtype=torch.int32, device=device)
    with _timed("dropout_bitmask", bytes=rows * width / 8.0):
        rc = lib.ptgnn_amd_dropout_bitmask(rows, width, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, bits.data_ptr(),
                                           _stream(bits))
    _lib.check(rc, "ptgnn_amd_dropout_bitmask")
    return bits


def edge_linear_masked_supported(state_dim: int, msg_dim: int, mode: int) -> bool:
    return bool(_lib.load().ptgnn_amd_edge_linear_masked_supported(int(state_dim), int(msg_dim), int(mode)))


def edge_weight_grad_masked_supported(state_dim: int, msg_dim: int) -> bool:
    return bool(_lib.load().ptgnn_amd_edge_weight_grad_masked_supported(int(state_dim), int(msg_dim)))


def edge_linear(x: torch.Tensor, adjacency_lists, weights: Sequence[torch.Tensor], use_dst: bool,
                act: Optional[str] = None, dropout: Optional[Tuple[int, float, int]] = None,
                mask_bits: Optional[torch.Tensor] = None,
                edge_feats: Optional[Sequence[Optional[torch.Tensor]]] = None) -> torch.Tensor:
    """msg[off_t + e] = act([x[src_t[e]] ; x[dst_t[e]] (if use_dst)] W_t^T) for every edge type in one
    launch; rows in type-major message order.  weights[t] is the type's nn.Linear weight.
    `edge_feats[t]` = [E_t, F] per-edge feature rows appended to the message input (weights [M, H (+H) + F];
    gatedmessagepassing.py:57-61, mlpmessagepassing.py:96-98) -- see `_edge_linear_feat`.
    dropout = (mode, p, seed): nn.Dropout(p) on the gathered input rows (mode 1) or on the output rows
    (mode 2, the input-gradient form) with the hash mask of ptgnn_amd_edge_linear_dropout_f32; with
    `mask_bits` (`dropout_bitmask` of the same p and seed) the streaming kernel applies the mask from its bits."""
    if edge_feats is not None and any(f is not None and f.shape[-1] > 0 for f in edge_feats):
        if dropout is not None and dropout[0] != 0 and dropout[1] > 0.0:
            raise _lib.PtgnnAmdError("edge_linear: the dropout forms take no edge features")
        return _edge_linear_feat(x, adjacency_lists, weights, use_dst, act, edge_feats)
    lib = _lib.load()
    _require_cuda_f32("x", x)
    x = _rowmajor(x)
    T = len(adjacency_lists)
    H = x.shape[1]
    M = weights[0].shape[0]
    counts = [int(a[0].shape[0]) for a in adjacency_lists]
    E = sum(counts)
    msg = torch.empty(max(E, 1), M, dtype=torch.float32, device=x.device)
    ws = [w.detach().contiguous() for w in weights]
    for w in ws:
        if tuple(w.shape) != (M, H * (2 if use_dst else 1)) or not w.is_cuda or w.dtype != torch.float32:
            raise _lib.PtgnnAmdError(f"edge_linear: weight shape {tuple(w.shape)} does not match "
                                     f"[{M}, {H * (2 if use_dst else 1)}]")
    srcs = [a[0].contiguous() for a in adjacency_lists]
    dsts = [a[1].contiguous() for a in adjacency_lists]
    PtrArr, CntArr = ctypes.c_void_p * T, ctypes.c_int64 * T
    sp = PtrArr(*[s.data_ptr() if s.numel() else None for s in srcs])
    dp = PtrArr(*[d.data_ptr() if d.numel() else None for d in dsts])
    wp = PtrArr(*[w.data_ptr() for w in ws])
    cn = CntArr(*counts)
    K = H * (2 if use_dst else 1)
    if dropout is not None and dropout[0] != 0 and dropout[1] > 0.0:
        if use_dst or act is not None:
            raise _lib.PtgnnAmdError("edge_linear: dropout supports the GGNN form only (no target half, no act)")
        if mask_bits is not None and E > 0 and edge_linear_masked_supported(H, M, dropout[0]):
            words = (M if dropout[0] == 2 else H) // 32
            if tuple(mask_bits.shape) != (E, words) or mask_bits.dtype != torch.int32 or not mask_bits.is_cuda:
                raise _lib.PtgnnAmdError(f"edge_linear: mask_bits must be int32 [{E}, {words}] on the device")
            mask_bits = mask_bits.contiguous()
            with _timed("edge_linear", flops=2.0 * E * K * M, bytes=4.0 * (E * K + E * M + T * M * K) + 8.0 * E):
                rc = lib.ptgnn_amd_edge_linear_masked_f32(
                    x.data_ptr(), _ld(x), x.shape[0], H, ctypes.cast(sp, ctypes.c_void_p),
                    ctypes.cast(cn, ctypes.c_void_p), ctypes.cast(wp, ctypes.c_void_p), T, M, msg.data_ptr(), M,
                    int(dropout[0]), float(dropout[1]), mask_bits.data_ptr(), _stream(msg))
            if rc != _lib.EUNSUPPORTED:
                _lib.check(rc, "ptgnn_amd_edge_linear_masked_f32")
                return msg[:E]
            # the streaming kernel declined at run time (first use inside a graph capture, an operand that is not
            # 16-byte aligned, PTGNN_AMD_* developer switches): the seed form below evaluates the SAME hash inside the
            # tile kernels -- bit-identical mask -- and rewrites every output row
        with _timed("edge_linear", flops=2.0 * E * K * M, bytes=4.0 * (E * K + E * M + T * M * K) + 8.0 * E):
            rc = lib.ptgnn_amd_edge_linear_dropout_f32(
                x.data_ptr(), _ld(x), x.shape[0], H, ctypes.cast(sp, ctypes.c_void_p),
                ctypes.cast(cn, ctypes.c_void_p),
                ctypes.cast(wp, ctypes.c_void_p), T, M, msg.data_ptr(), M, int(dropout[0]),
                float(dropout[1]), int(dropout[2]) & 0xFFFFFFFFFFFFFFFF, _stream(msg))
        _lib.check(rc, "ptgnn_amd_edge_linear_dropout_f32")
        return msg[:E] if E > 0 else msg[:0]
    with _timed("edge_linear", flops=2.0 * E * K * M, bytes=4.0 * (E * K + E * M + T * M * K) + 8.0 * E):
        rc = lib.ptgnn_amd_edge_linear_f32(x.data_ptr(), _ld(x), x.shape[0], H, ctypes.cast(sp, ctypes.c_void_p),
                                           ctypes.cast(dp, ctypes.c_void_p) if use_dst else None,
                                           ctypes.cast(cn, ctypes.c_void_p),
                                           ctypes.cast(wp, ctypes.c_void_p), T, M, ACT_IDS[act],
                                           msg.data_ptr(), M, _stream(msg))
    _lib.check(rc, "ptgnn_amd_edge_linear_f32")
    return msg[:E] if E > 0 else msg[:0]


def _edge_linear_feat(x, adjacency_lists, weights, use_dst: bool, act, edge_feats) -> torch.Tensor:
    """The grouped per-edge GEMM with feature rows as a third K range of its A operand: nothing of the reference's
    [E, H (+H) + F] message input is materialised.  Feature widths that are not a multiple of 4 are zero-padded (the
    [E_t, F] block and the weight's feature columns: exact zeros in the products), so only the small block is copied."""
    lib = _lib.load()
    _require_cuda_f32("x", x)
    x = _rowmajor(x)
    T = len(adjacency_lists)
    H, M = x.shape[1], weights[0].shape[0]
    Hs = H * (2 if use_dst else 1)
    counts = [int(a[0].shape[0]) for a in adjacency_lists]
    E = sum(counts)
    if len(edge_feats) != T:
        raise _lib.PtgnnAmdError(f"edge_linear: {len(edge_feats)} feature blocks for {T} edge types")
    F = max(int(f.shape[-1]) for f in edge_feats if f is not None)
    F4 = (F + 3) // 4 * 4
    feats, ws = [], []
    for t, (f, w, n) in enumerate(zip(edge_feats, weights, counts)):
        w = w.detach()
        if tuple(w.shape) != (M, Hs + F) or not w.is_cuda or w.dtype != torch.float32:
            raise _lib.PtgnnAmdError(f"edge_linear: weight shape {tuple(w.shape)} does not match [{M}, {Hs + F}]")
        if f is None or tuple(f.shape) != (n, F) or not f.is_cuda:
            raise _lib.PtgnnAmdError(f"edge_linear: features of type {t} must be a device tensor [{n}, {F}]")
        f = f.detach().float()
        if F4 != F:
            f = torch.nn.functional.pad(f, (0, F4 - F))
            w = torch.nn.functional.pad(w, (0, F4 - F))
        feats.append(f.contiguous())
        ws.append(w.contiguous())
    msg = torch.empty(max(E, 1), M, dtype=torch.float32, device=x.device)
    srcs = [a[0].contiguous() for a in adjacency_lists]
    dsts = [a[1].contiguous() for a in adjacency_lists]
    PtrArr, CntArr = ctypes.c_void_p * T, ctypes.c_int64 * T
    sp = PtrArr(*[s.data_ptr() if s.numel() else None for s in srcs])
    dp = PtrArr(*[d.data_ptr() if d.numel() else None for d in dsts])
    fp = PtrArr(*[f.data_ptr() if f.numel() else None for f in feats])
    wp = PtrArr(*[w.data_ptr() for w in ws])
    cn = CntArr(*counts)
    K = Hs + F
    with _timed("edge_linear_feat", flops=2.0 * E * K * M, bytes=4.0 * (E * K + E * M + T * M * K) + 8.0 * E):
        rc = lib.ptgnn_amd_edge_linear_feat_f32(
            x.data_ptr(), _ld(x), x.shape[0], H, ctypes.cast(sp, ctypes.c_void_p),
            ctypes.cast(dp, ctypes.c_void_p) if use_dst else None, ctypes.cast(fp, ctypes.c_void_p), F4, F4,
            ctypes.cast(cn, ctypes.c_void_p), ctypes.cast(wp, ctypes.c_void_p), T, M, ACT_IDS[act], msg.data_ptr(), M,
            _stream(msg))
    _lib.check(rc, "ptgnn_amd_edge_linear_feat_f32")
    return msg[:E] if E > 0 else msg[:0]


def edge_weight_grad(x: torch.Tensor, adjacency_lists, grad_msg: torch.Tensor, use_dst: bool,
                     dropout_p: float = 0.0, dropout_seed: int = 0,
                     mask_bits: Optional[torch.Tensor] = None) -> torch.Tensor:
    """grad_w[t] = grad_msg_t^T . [x[src_t] ; x[dst_t] (if use_dst)]  for all types -> [T, M, K].
    `mask_bits`: the dropout keep mask as bits (`dropout_bitmask` of the same p and seed) instead of the hash."""
    lib = _lib.load()
    _require_cuda_f32("x", x)
    _require_cuda_f32("grad_msg", grad_msg)
    x, grad_msg = _rowmajor(x), _rowmajor(grad_msg)
    T, H, M = len(adjacency_lists), x.shape[1], grad_msg.shape[1]
    K = H * (2 if use_dst else 1)
    counts = [int(a[0].shape[0]) for a in adjacency_lists]
    E = sum(counts)
    if grad_msg.shape[0] != E:
        raise _lib.PtgnnAmdError("edge_weight_grad: grad_msg rows != number of edges")
    srcs = [a[0].contiguous() for a in adjacency_lists]
    dsts = [a[1].contiguous() for a in adjacency_lists]
    PtrArr, CntArr = ctypes.c_void_p * T, ctypes.c_int64 * T
    sp = PtrArr(*[s.data_ptr() if s.numel() else None for s in srcs])
    dp = PtrArr(*[d.data_ptr() if d.numel() else None for d in dsts])
    cn = CntArr(*counts)
    grad_w = torch.empty(T, M, K, dtype=torch.float32, device=x.device)
    ws_bytes = lib.ptgnn_amd_edge_wgrad_workspace_bytes(E, T, M, K)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=x.device)
    gm_ptr = grad_msg.data_ptr() if E > 0 else x.data_ptr()
    if (mask_bits is not None and dropout_p > 0.0 and E > 0 and not use_dst
            and edge_weight_grad_masked_supported(H, M)):
        if tuple(mask_bits.shape) != (E, H // 32) or mask_bits.dtype != torch.int32 or not mask_bits.is_cuda:
            raise _lib.PtgnnAmdError(f"edge_weight_grad: mask_bits must be int32 [{E}, {H // 32}] on the device")
        mask_bits = mask_bits.contiguous()
        with _timed("edge_weight_grad", flops=2.0 * E * K * M, bytes=4.0 * (E * K + E * M + T * M * K) + 8.0 * E):
            rc = lib.ptgnn_amd_edge_weight_grad_masked_f32(
                x.data_ptr(), _ld(x), x.shape[0], H, ctypes.cast(sp, ctypes.c_void_p),
                ctypes.cast(cn, ctypes.c_void_p), gm_ptr, _ld(grad_msg), T, M, float(dropout_p),
                mask_bits.data_ptr(), grad_w.data_ptr(), ws.data_ptr(), ws_bytes, _stream(grad_w))
        if rc != _lib.EUNSUPPORTED:
            _lib.check(rc, "ptgnn_amd_edge_weight_grad_masked_f32")
            return grad_w
        # declined at run time (see edge_linear): the seed form computes the same mask from its hash
    with _timed("edge_weight_grad", flops=2.0 * E * K * M, bytes=4.0 * (E * K + E * M + T * M * K) + 8.0 * E):
        rc = lib.ptgnn_amd_edge_weight_grad_f32(
            x.data_ptr(), _ld(x), x.shape[0], H, ctypes.cast(sp, ctypes.c_void_p),
            ctypes.cast(dp, ctypes.c_void_p) if use_dst else None, ctypes.cast(cn, ctypes.c_void_p),
            gm_ptr, _ld(grad_msg) if E > 0 else M, T, M, float(dropout_p),
            int(dropout_seed) & 0xFFFFFFFFFFFFFFFF, grad_w.data_ptr(), ws.data_ptr(), ws_bytes, _stream(grad_w))
    _lib.check(rc, "ptgnn_amd_edge_weight_grad_f32")
    return grad_w


def linear_weight_grad(x: torch.Tensor, grad_y: torch.Tensor, want_bias: bool = False):
    """grad_w [n_out, k] = grad_y^T . x  (weight gradient of y = x W^T), deterministic split-row MFMA GEMM;
    with `want_bias` also grad_b [n_out] = column sums of grad_y from the same pass -> (grad_w, grad_b)."""
    lib = _lib.load()
    _require_cuda_f32("x", x)
    _require_cuda_f32("grad_y", grad_y)
    x, grad_y = _rowmajor(x), _rowmajor(grad_y)
    rows, k = x.shape
    n_out = grad_y.shape[1]
    if grad_y.shape[0] != rows:
        raise _lib.PtgnnAmdError("linear_weight_grad: x and grad_y row counts differ")
    if rows == 0:
        gw = torch.zeros(n_out, k, dtype=torch.float32, device=x.device)
        return (gw, torch.zeros(n_out, dtype=torch.float32, device=x.device)) if want_bias else gw
    grad_w = torch.empty(n_out, k, dtype=torch.float32, device=x.device)
    grad_b = torch.empty(n_out, dtype=torch.float32, device=x.device) if want_bias else None
    ws_bytes = lib.ptgnn_amd_edge_wgrad_workspace_bytes(rows, 1, n_out, k)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=x.device)
    with _timed("linear_weight_grad", flops=2.0 * rows * k * n_out, bytes=4.0 * (rows * k + rows * n_out + n_out * k)):
        rc = lib.ptgnn_amd_linear_weight_grad_f32(x.data_ptr(), _ld(x), k, grad_y.data_ptr(), _ld(grad_y),
                                                  rows, n_out, grad_w.data_ptr(),
                                                  grad_b.data_ptr() if want_bias else None,
                                                  ws.data_ptr(), ws_bytes, _stream(grad_w))
    _lib.check(rc, "ptgnn_amd_linear_weight_grad_f32")
    return (grad_w, grad_b) if want_bias else grad_w


def segment_spread(grad: torch.Tensor, arg: Optional[torch.Tensor], plan: GraphPlan) -> torch.Tensor:
    """Backward of `segment_reduce` w.r.t. the messages: [N, D] row gradients -> [E, D] in message order
    (max/min: only the recorded winner slot of each (row, column) receives the gradient)."""
    lib = _lib.load()
    _require_cuda_f32("grad", grad)
    grad = _rowmajor(grad)
    E, D = plan.num_edges, grad.shape[1]
    out = torch.empty(E, D, dtype=torch.float32, device=grad.device)
    if E == 0:
        return out
    if arg is not None:
        arg = arg.contiguous()
    plan.wait()
    with _timed("segment_spread", bytes=E * (4.0 * D + 8) + plan.num_nodes * 4.0 * D * (2 if arg is not None else 1)):
        rc = lib.ptgnn_amd_segment_spread_f32(grad.data_ptr(), _ld(grad),
                                              arg.data_ptr() if arg is not None else None,
                                              plan.slot_rows().data_ptr(), plan.perm.data_ptr(), E, D,
                                              out.data_ptr(), D, _stream(out))
    _lib.check(rc, "ptgnn_amd_segment_spread_f32")
    return out


def row_epilogue(x: torch.Tensor, flags: int, ln_weight: Optional[torch.Tensor] = None,
                 ln_bias: Optional[torch.Tensor] = None, ln_eps: float = 1e-5) -> torch.Tensor:
    """y = LayerNorm(GELU(x)) over the rows of x (`flags`: EPI_GELU | EPI_LAYERNORM) -- the training-time twin of
    the fused aggregation epilogue (mlpmessagepassing.py:114-116)."""
    lib = _lib.load()
    _require_cuda_f32("x", x)
    x = _rowmajor(x)
    n, d = x.shape
    y = torch.empty(n, d, dtype=torch.float32, device=x.device)
    if flags & EPI_LAYERNORM:
        ln_weight, ln_bias = ln_weight.contiguous(), ln_bias.contiguous()
    with _timed("row_epilogue", bytes=8.0 * n * d):
        rc = lib.ptgnn_amd_row_epilogue_f32(x.data_ptr(), _ld(x), n, d, flags,
                                            ln_weight.data_ptr() if flags & EPI_LAYERNORM else None,
                                            ln_bias.data_ptr() if flags & EPI_LAYERNORM else None, float(ln_eps),
                                            y.data_ptr(), d, _stream(y))
    _lib.check(rc, "ptgnn_amd_row_epilogue_f32")
    return y


def row_epilogue_backward(x: torch.Tensor, grad_y: torch.Tensor, flags: int, ln_weight: Optional[torch.Tensor] = None,
                          ln_eps: float = 1e-5):
    """(grad_x, grad_gamma, grad_beta) of `row_epilogue`; the last two are None without LayerNorm."""
    lib = _lib.load()
    _require_cuda_f32("x", x)
    _require_cuda_f32("grad_y", grad_y)
    x, grad_y = _rowmajor(x), _rowmajor(grad_y)
    n, d = x.shape
    gx = torch.empty(n, d, dtype=torch.float32, device=x.device)
    ln = bool(flags & EPI_LAYERNORM)
    gg = gb = ws = None
    ws_bytes = 0
    if ln:
        ln_weight = ln_weight.contiguous()
        gg = torch.empty(d, dtype=torch.float32, device=x.device)
        gb = torch.empty(d, dtype=torch.float32, device=x.device)
        ws_bytes = int(lib.ptgnn_amd_row_epilogue_workspace_bytes(n, d))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=x.device)
    with _timed("row_epilogue_backward", bytes=12.0 * n * d):
        rc = lib.ptgnn_amd_row_epilogue_backward_f32(x.data_ptr(), _ld(x), grad_y.data_ptr(), _ld(grad_y), n, d, flags,
                                                     ln_weight.data_ptr() if ln else None, float(ln_eps),
                                                     gx.data_ptr(), d, gg.data_ptr() if ln else None,
                                                     gb.data_ptr() if ln else None,
                                                     ws.data_ptr() if ln else None, ws_bytes, _stream(gx))
    _lib.check(rc, "ptgnn_amd_row_epilogue_backward_f32")
    return gx, gg, gb


def act_dropout_backward(grad: torch.Tensor, y: torch.Tensor, keep: Optional[torch.Tensor], scale: float,
                         act: Optional[str]) -> torch.Tensor:
    """grad * (keep ? scale : 0) * act'(y) in one pass (ptgnn_amd_act_dropout_backward_f32); `y` is the activation's
    output, `keep` the dropout's bool mask or None."""
    lib = _lib.load()
    _require_cuda_f32("grad", grad)
    grad, y = grad.contiguous(), y.contiguous()
    out = torch.empty_like(grad)
    n = grad.numel()
    if keep is not None:
        keep = keep.contiguous()
        if keep.dtype != torch.bool or keep.numel() != n:
            raise _lib.PtgnnAmdError("act_dropout_backward: keep must be a bool mask of grad's shape")
    with _timed("act_dropout_backward", bytes=(12.0 + (1.0 if keep is not None else 0.0)) * n):
        rc = lib.ptgnn_amd_act_dropout_backward_f32(grad.data_ptr(), y.data_ptr(),
                                                    keep.data_ptr() if keep is not None else None, float(scale),
                                                    ACT_IDS[act], n, out.data_ptr(), _stream(out))
    _lib.check(rc, "ptgnn_amd_act_dropout_backward_f32")
    return out


def gru_cell(a: torch.Tensor, h: torch.Tensor, w_ih, w_hh, b_ih, b_hh, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`out`: optional caller-owned [n, hd] fp32 destination with unit inner stride (e.g. the right half of the
    buffer a concat residual returns)."""
    lib = _lib.load()
    _require_cuda_f32("a", a)
    _require_cuda_f32("h", h)
    a, h = _rowmajor(a), _rowmajor(h)
    n, m = a.shape
    hd = h.shape[1]
    if out is None:
        out = torch.empty(n, hd, dtype=torch.float32, device=a.device)
    elif tuple(out.shape) != (n, hd) or out.dtype != torch.float32 or out.stride(1) != 1 or not out.is_cuda:
        raise _lib.PtgnnAmdError(f"gru_cell: `out` must be a float32 CUDA [{n}, {hd}] view with unit inner stride")
    with _timed("gru_cell", flops=2.0 * n * 3 * hd * (m + hd), bytes=4.0 * (n * (m + 2 * hd) + 3 * hd * (m + hd))):
        rc = lib.ptgnn_amd_gru_cell_f32(a.data_ptr(), _ld(a), h.data_ptr(), _ld(h),
                                        w_ih.contiguous().data_ptr(), w_hh.contiguous().data_ptr(),
                                        b_ih.contiguous().data_ptr(), b_hh.contiguous().data_ptr(),
                                        n, m, hd, out.data_ptr(), _ld(out), _stream(out))
    _lib.check(rc, "ptgnn_amd_gru_cell_f32")
    return out


# Aggregation -> GRU of one GGNN layer, pipelined over destination-row ranges (round 5; OFF by default -- it measured
# slower).  The aggregation is latency / HBM-bound (cfg3: 73-76 us, most wave cycles waiting on memory), the fused GRU cell
# MFMA-bound (191-240 us), and after the aggregation the layer is row-wise (gatedmessagepassing.py:63-69): row range i
# of the GRU only needs row range i of the aggregate.  So the ranges' aggregations can run back to back on a side stream
# while the main stream runs the GRU of the ranges that are done.  Same kernels, same per-row arithmetic: bit-identical to
# the unsplit pair (tests/test_gpu_pipeline.py).  Measured on the cfg3 headline step (profiles/r05_notes.md 1): 3.93 ms
# unsplit, 4.19 ms with 2 ranges, 4.25 with 3, 4.36 with 4 -- the persistent GRU workgroups hold every CU's LDS and most
# of its wave slots, so the co-resident aggregation crawls (99 us per HALF against 76 us for the whole matrix alone) and
# each extra GRU launch pays its weight-slab fill again (2 x 136 us against 240 us).  PTGNN_AMD_AGG_PIPELINE = number of row
# ranges (default 1 = the unsplit pair).
AGG_PIPELINE = int(os.environ.get("PTGNN_AMD_AGG_PIPELINE", "1"))
AGG_PIPELINE_MIN_ROWS = int(os.environ.get("PTGNN_AMD_AGG_PIPELINE_MIN_ROWS", "65536"))


def aggregate_gru(ysrc: torch.Tensor, plan: GraphPlan, msg_dim: int, reduce: str, h: torch.Tensor, w_ih, w_hh, b_ih,
                  b_hh, type_bits: Optional[int] = None, col: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GRUCell(aggregate(ysrc over plan), h) -- `gather_reduce` followed by `gru_cell`, pipelined over row ranges on
    large minibatches (see above)."""
    N = plan.num_nodes
    pieces = AGG_PIPELINE
    if (pieces < 2 or N < AGG_PIPELINE_MIN_ROWS or N < 64 * pieces or torch.cuda.is_current_stream_capturing()):
        agg = gather_reduce(ysrc, plan, msg_dim, reduce, type_bits=type_bits, col=col)
        return gru_cell(agg, h, w_ih, w_hh, b_ih, b_hh, out=out)
    dev = h.device
    hd = h.shape[1]
    agg = torch.empty(N, msg_dim, dtype=torch.float32, device=dev)
    if out is None:
        out = torch.empty(N, hd, dtype=torch.float32, device=dev)
    bounds = [min(N, (N * i // pieces + 31) // 32 * 32) for i in range(pieces)] + [N]
    main, side = torch.cuda.current_stream(dev), _side_stream(dev)
    plan.wait()
    gather_reduce(ysrc, plan, msg_dim, reduce, type_bits=type_bits, col=col, out=agg, rows=(bounds[0], bounds[1]))
    first_done = torch.cuda.Event()
    first_done.record(main)
    ready = []
    with torch.cuda.stream(side):
        side.wait_event(first_done)          # the ranges' aggregations never overlap each other (shared hub tickets)
        for i in range(1, pieces):
            gather_reduce(ysrc, plan, msg_dim, reduce, type_bits=type_bits, col=col, out=agg,
                          rows=(bounds[i], bounds[i + 1]))
            ev = torch.cuda.Event()
            ev.record(side)
            ready.append(ev)
    for i in range(pieces):
        if i > 0:
            main.wait_event(ready[i - 1])
        lo, hi = bounds[i], bounds[i + 1]
        if hi > lo:
            gru_cell(agg[lo:hi], h[lo:hi], w_ih, w_hh, b_ih, b_hh, out=out[lo:hi])
    return out


def gru_cell_train(a: torch.Tensor, h: torch.Tensor, w_ih, w_hh, b_ih, b_hh):
    """Fused GRU cell that also returns the gates the backward needs: (h', gates [n, 4*hd] = r|z|n|gh_n)."""
    lib = _lib.load()
    _require_cuda_f32("a", a)
    _require_cuda_f32("h", h)
    a, h = _rowmajor(a), _rowmajor(h)
    n, m = a.shape
    hd = h.shape[1]
    out = torch.empty(n, hd, dtype=torch.float32, device=a.device)
    gates = torch.empty(n, 4 * hd, dtype=torch.float32, device=a.device)
    with _timed("gru_cell", flops=2.0 * n * 3 * hd * (m + hd),
                bytes=4.0 * (n * (m + 2 * hd + 4 * hd) + 3 * hd * (m + hd))):
        rc = lib.ptgnn_amd_gru_cell_train_f32(a.data_ptr(), _ld(a), h.data_ptr(), _ld(h),
                                              w_ih.contiguous().data_ptr(), w_hh.contiguous().data_ptr(),
                                              b_ih.contiguous().data_ptr(), b_hh.contiguous().data_ptr(),
                                              n, m, hd, out.data_ptr(), hd, gates.data_ptr(), _stream(out))
    _lib.check(rc, "ptgnn_amd_gru_cell_train_f32")
    return out, gates


def gru_gates_backward(grad_out: torch.Tensor, gates: torch.Tensor, h: torch.Tensor):
    """Backward of the GRU gate math: (d_gi [n, 3hd], d_gh [n, 3hd], d_h_direct [n, hd])."""
    lib = _lib.load()
    _require_cuda_f32("grad_out", grad_out)
    _require_cuda_f32("h", h)
    grad_out, h = _rowmajor(grad_out), _rowmajor(h)
    n, hd = h.shape
    d_gi = torch.empty(n, 3 * hd, dtype=torch.float32, device=h.device)
    d_gh = torch.empty(n, 3 * hd, dtype=torch.float32, device=h.device)
    d_h = torch.empty(n, hd, dtype=torch.float32, device=h.device)
    with _timed("gru_gates_backward", bytes=4.0 * n * hd * 13):
        rc = lib.ptgnn_amd_gru_cell_backward_gates_f32(grad_out.data_ptr(), _ld(grad_out), gates.data_ptr(),
                                                       h.data_ptr(), _ld(h), n, hd, d_gi.data_ptr(),
                                                       d_gh.data_ptr(), d_h.data_ptr(), _stream(d_h))
    _lib.check(rc, "ptgnn_amd_gru_cell_backward_gates_f32")
    return d_gi, d_gh, d_h


def gather_rows(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _require_cuda_f32("x", x)
    x = _rowmajor(x)
    if idx.dtype != torch.int64 or not idx.is_cuda:
        raise _lib.PtgnnAmdError("gather_rows: idx must be a CUDA int64 tensor")
    idx = idx.contiguous()
    out = torch.empty(idx.shape[0], x.shape[1], dtype=torch.float32, device=x.device)
    rc = lib.ptgnn_amd_gather_rows_f32(x.data_ptr(), _ld(x), idx.data_ptr(), idx.shape[0],
                                       x.shape[1], out.data_ptr(), x.shape[1], _stream(out))
    _lib.check(rc, "ptgnn_amd_gather_rows_f32")
    return out


def weighted_pool(x: torch.Tensor, w: torch.Tensor, plan: GraphPlan) -> torch.Tensor:
    """out[g] = sum_{i in segment g} sigmoid(x_i . w) x_i over the plan of an element -> sample map
    (ptgnn_amd_weighted_pool_f32: WeightedSumVarSizedElementReduce, varsizedsummary.py:68-81, in one pass over x)."""
    lib = _lib.load()
    _require_cuda_f32("x", x)
    x = _rowmajor(x)
    n, d = x.shape
    if n != plan.num_edges or w.numel() != d or plan.perm is None:
        raise _lib.PtgnnAmdError(f"weighted_pool: x {tuple(x.shape)}, w {tuple(w.shape)} do not fit a plan over "
                                 f"{plan.num_edges} elements")
    w = w.detach().reshape(-1).contiguous()
    G = plan.num_nodes
    out = torch.empty(G, d, dtype=torch.float32, device=x.device)
    ws_bytes = int(lib.ptgnn_amd_weighted_pool_workspace_bytes(G, n, d))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=x.device)
    plan.wait()
    with _timed("weighted_pool", bytes=4.0 * (n * d + G * d + d) + 4.0 * n):
        rc = lib.ptgnn_amd_weighted_pool_f32(x.data_ptr() if n else None, _ld(x) if n else d, w.data_ptr(),
                                             plan.rowptr.data_ptr(), plan.perm.data_ptr(), G, n, d, out.data_ptr(), d,
                                             ws.data_ptr(), ws_bytes, _stream(out))
    _lib.check(rc, "ptgnn_amd_weighted_pool_f32")
    return out


def weighted_pool_backward(x: torch.Tensor, w: torch.Tensor, index: torch.Tensor, grad_out: torch.Tensor):
    """(grad_x [n, d], grad_w [d]) of `weighted_pool` from grad_out [G, d] and the int64 element -> sample map."""
    lib = _lib.load()
    _require_cuda_f32("x", x)
    _require_cuda_f32("grad_out", grad_out)
    x, grad_out = _rowmajor(x), _rowmajor(grad_out)
    n, d = x.shape
    if index.dtype != torch.int64 or not index.is_cuda or index.shape[0] != n:
        raise _lib.PtgnnAmdError("weighted_pool_backward: the map must be a CUDA int64 tensor with one entry per element")
    index = index.contiguous()
    w = w.detach().reshape(-1).contiguous()
    gx = torch.empty(n, d, dtype=torch.float32, device=x.device)
    gw = torch.empty(d, dtype=torch.float32, device=x.device)
    ws_bytes = int(lib.ptgnn_amd_weighted_pool_backward_workspace_bytes(n, d))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=x.device)
    with _timed("weighted_pool_backward", bytes=4.0 * (3 * n * d) + 8.0 * n):
        rc = lib.ptgnn_amd_weighted_pool_backward_f32(x.data_ptr() if n else None, _ld(x) if n else d, w.data_ptr(),
                                                      index.data_ptr() if n else None,
                                                      grad_out.data_ptr() if grad_out.numel() else None,
                                                      _ld(grad_out) if grad_out.numel() else d, n, d,
                                                      gx.data_ptr() if n else None, d, gw.data_ptr(), ws.data_ptr(),
                                                      ws_bytes, _stream(gx))
    _lib.check(rc, "ptgnn_amd_weighted_pool_backward_f32")
    return gx, gw
