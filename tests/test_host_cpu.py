"""CPU-side checks (no GPU): the C-ABI library builds, loads and exports every symbol the header
declares; the host mirror of the reference interface behaves (constructors, state_dict keys,
properties, loud failure on CPU tensors)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ptgnn_amd import build
    path = build.build()
    assert os.path.exists(path)
    from ptgnn_amd import _lib
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptgnn_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptgnn_amd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    from ptgnn_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 10
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/ptgnn_amd.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in ptgnn_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms


def test_version_and_pure_host_entry_points(lib):
    assert lib.ptgnn_amd_version() == 102 >= __import__('ptgnn_amd')._lib.MIN_VERSION
    assert [lib.ptgnn_amd_type_bits(t) for t in (1, 2, 3, 4, 5, 17, 32, 33)] == [0, 1, 2, 2, 3, 5, 5, 6]
    assert lib.ptgnn_amd_last_error() is not None


def test_argument_validation_without_a_gpu(lib):
    # bad arguments are rejected before any HIP call is made
    from ptgnn_amd import PtgnnAmdError, _lib
    rc = lib.ptgnn_amd_linear_f32(None, 4, 0, 4, None, 4, None, 0, None, 4, None)
    assert rc == -1 and b"linear" in lib.ptgnn_amd_last_error()
    rc = lib.ptgnn_amd_gather_reduce_f32(None, 4, None, 4, None, None, 0, 4, 4, 9, 0, None, None, 1e-5,
                                         None, 4, None, 0, 0, None, None, None, 0, None, None)
    assert rc == -1 and b"reduce" in lib.ptgnn_amd_last_error()
    with pytest.raises(PtgnnAmdError):
        _lib.check(rc, "gather_reduce")
    rc = lib.ptgnn_amd_gru_cell_f32(None, 1, None, 1, None, None, None, None, 3, 4, 4, None, 4, None)
    assert rc == -1


def test_layers_mirror_reference_interface_and_dispatch_on_the_device():
    from ptgnn_amd import PtgnnAmdError, layers as L
    g = L.GatedMessagePassingLayer(state_dimension=8, message_dimension=12, num_edge_types=3,
                                   message_aggregation_function="max", dropout_rate=0.1,
                                   edge_feature_dimension=0)
    assert g.input_state_dimension == 8 and g.output_state_dimension == 8
    assert list(g.state_dict()) == [
        "_GatedMessagePassingLayer__edge_message_transformation_layers.0.weight",
        "_GatedMessagePassingLayer__edge_message_transformation_layers.1.weight",
        "_GatedMessagePassingLayer__edge_message_transformation_layers.2.weight",
        "_GatedMessagePassingLayer__state_update.weight_ih",
        "_GatedMessagePassingLayer__state_update.weight_hh",
        "_GatedMessagePassingLayer__state_update.bias_ih",
        "_GatedMessagePassingLayer__state_update.bias_hh"]
    m = L.MlpMessagePassingLayer(input_state_dimension=8, output_state_dimension=6, message_dimension=10,
                                 num_edge_types=2, message_aggregation_function="sum")
    assert m.input_state_dimension == 8 and m.output_state_dimension == 6
    assert list(m.state_dict()) == [
        "_MlpMessagePassingLayer__edge_message_transformation_layers.0._MLP__mlp_modules.1.weight",
        "_MlpMessagePassingLayer__edge_message_transformation_layers.1._MLP__mlp_modules.1.weight",
        "_MlpMessagePassingLayer__state_update.0.weight", "_MlpMessagePassingLayer__state_update.0.bias",
        "_MlpMessagePassingLayer__state_update.1.weight", "_MlpMessagePassingLayer__state_update.1.bias"]
    assert tuple(m.state_dict()["_MlpMessagePassingLayer__edge_message_transformation_layers.0."
                                "_MLP__mlp_modules.1.weight"].shape) == (10, 16)
    adj = [(torch.tensor([0]), torch.tensor([1]))] * 3
    # CPU tensors: device dispatch to the plain-torch route (ptgnn_amd/torch_route.py; predict.py runs on "cpu") ...
    y = g.eval()(torch.randn(4, 8), adj, None, {}, {}, [torch.empty(1, 0)] * 3)
    assert tuple(y.shape) == (4, 8) and not y.is_cuda
    # ... while every C-ABI wrapper keeps refusing them (the HIP path has no eager substitute) and the sharded
    # forms exist on the GPU only
    from ptgnn_amd import ops
    with pytest.raises(PtgnnAmdError):
        ops.linear(torch.randn(4, 8), torch.randn(3, 8))
    with pytest.raises(PtgnnAmdError):
        ops.build_plan(adj, 4)
    with pytest.raises(PtgnnAmdError):
        g.forward_sharded(torch.randn(4, 8), None)
    with pytest.raises(AssertionError):          # wrong number of edge types: the reference's assert
        g(torch.randn(4, 8), adj[:2], None, {}, {}, [torch.empty(1, 0)] * 2)
    r = L.ConcatResidualLayer(8)
    o = r.pass_through_dummy_layer()
    x = torch.randn(4, 8)
    assert o(x, adj, None, {}, {}, []) is x
    assert tuple(r(2 * x, adj, None, {}, {}, []).shape) == (4, 16) and r.output_state_dimension == 16
    with pytest.raises(AssertionError):          # residual used without its origin layer
        r(x, adj, None, {}, {}, [])
    mean = L.MeanResidualLayer(8)
    mean.pass_through_dummy_layer()(x, adj, None, {}, {}, [])
    torch.testing.assert_close(mean(3 * x, adj, None, {}, {}, []),
                               torch.stack((x, 3 * x), dim=-1).mean(dim=-1), rtol=0, atol=0)


def test_golden_fixture_weights_load_into_layers():
    from conftest import load_golden
    from helpers import layer_from_spec
    from oracle.fixtures import unpack_specs
    for name in ("ggnn_layer_max", "mlp_layer_sum_target", "mlp_layer_sum_hidden1", "mlp_layer_max_noln_nodense"):
        (spec,) = unpack_specs(load_golden(name))
        layer = layer_from_spec(spec)
        back = layer.export_weights()
        assert back["kind"] == spec["kind"] and back["agg"] == spec["agg"]
        if spec["kind"] == "ggnn":
            for a, b in zip(back["edge_w"], spec["edge_w"]):
                assert torch.equal(a, b)
        else:
            for la, lb in zip(back["edge_mlp"], spec["edge_mlp"]):
                for a, b in zip(la, lb):
                    assert torch.equal(a, b)


def test_scatter_facade_rejects_unsupported_layouts():
    from ptgnn_amd import PtgnnAmdError
    from ptgnn_amd.scatter import scatter
    from ptgnn_amd.scatter import _prepare
    with pytest.raises(PtgnnAmdError, match="supports src"):     # the HIP side's layout rule (argument check only)
        _prepare(torch.randn(4, 3, 2), torch.tensor([0, 1, 0, 1]), 0, None, None)
    assert tuple(scatter(torch.randn(4, 3, 2), torch.tensor([0, 1, 0, 1]), dim=0).shape) == (2, 3, 2)   # host route
    with pytest.raises(PtgnnAmdError):
        scatter(torch.randn(4, 3), torch.tensor([0, 1, 0, 1]), dim=0, out=torch.zeros(2, 3))
    with pytest.raises(ValueError):
        from ptgnn_amd.scatter import segment_reduce
        segment_reduce(torch.randn(2, 2), None, "median")      # not one of torch_scatter's reduce names


def test_product_package_never_imports_the_oracle():
    import pathlib
    for f in pathlib.Path(ROOT, "ptgnn_amd").rglob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f


def test_minibatch_builder_host_packing_matches_reference_golden():
    """The host half of the device batcher (segment table + staging order), evaluated with the kernel's
    formula in numpy, reproduces the reference's finalize_minibatch layout bit for bit."""
    import numpy as np
    from conftest import load_golden
    from helpers import golden_batcher_graphs, replay_minibatches
    from ptgnn_amd.batching import MinibatchBuilder
    g = load_golden("batcher")
    T0, graphs = golden_batcher_graphs(g)
    packs = replay_minibatches(MinibatchBuilder, T0, graphs, int(g["stop_after"]), lambda b: b.pack())
    assert len(packs) == int(g["num_minibatches"])
    for bi, pk in enumerate(packs):
        vals = pk.evaluate_on_host()
        assert vals.dtype == np.int64 and pk.num_graphs == int(g[f"mb{bi}.num_graphs"])

        def view(name):
            b, e = pk.layout[name]
            return vals[b:e]
        np.testing.assert_array_equal(view("node_to_graph_idx"), g[f"mb{bi}.node_to_graph_idx"])
        for t in range(T0):
            np.testing.assert_array_equal(view(f"adj.{t}.src"), g[f"mb{bi}.adj.{t}.src"])
            np.testing.assert_array_equal(view(f"adj.{t}.dst"), g[f"mb{bi}.adj.{t}.dst"])
        for k in ("supernodes", "slot"):
            np.testing.assert_array_equal(view(f"ref_ids.{k}"), g[f"mb{bi}.ref_ids.{k}"])
            np.testing.assert_array_equal(view(f"ref_gidx.{k}"), g[f"mb{bi}.ref_gidx.{k}"])


def test_minibatch_builder_refuses_cpu():
    import numpy as np
    import pytest
    from ptgnn_amd import _lib
    from ptgnn_amd.batching import MinibatchBuilder
    b = MinibatchBuilder(1)
    b.extend([(np.zeros(2, np.int32), np.ones(2, np.int32))], 3, {})
    with pytest.raises(_lib.PtgnnAmdError):
        b.finalize("cpu")


def test_forward_scope_shares_derived_tensors_only_inside_a_scope():
    """Stacked tied weights are built once per forward scope (so autograd accumulates one stacked gradient
    per use) and never cached outside one or with grad disabled."""
    import torch
    from ptgnn_amd import layers as L
    calls = []

    def make():
        calls.append(1)
        return torch.zeros(2, requires_grad=True)
    owner = object()
    a, b = L._scoped(owner, "k", make), L._scoped(owner, "k", make)
    assert a is not b and len(calls) == 2                      # no scope: nothing is kept
    with L.forward_scope():
        c, d = L._scoped(owner, "k", make), L._scoped(owner, "k", make)
        assert c is d and len(calls) == 3
        assert L._scoped(object(), "k", make) is not c         # keyed on the owning module object
        with L.forward_scope():                                # nested scopes share the outer cache
            assert L._scoped(owner, "k", make) is c
        with torch.no_grad():
            assert L._scoped(owner, "k", make) is not c        # inference never caches autograd tensors
    assert getattr(L._SCOPE, "cache", None) is None
    with L.forward_scope():
        assert L._scoped(owner, "k", make) is not c            # a new forward builds its own


def test_forward_scope_is_per_thread():
    """Two threads (e.g. nn.DataParallel replicas) interleaving their forward scopes never see each other's
    cache and never leave one behind."""
    import threading
    import torch
    from ptgnn_amd import layers as L
    owner, seen = object(), {}
    enter_a, enter_b, exit_a = threading.Event(), threading.Event(), threading.Event()

    def worker_a():
        with L.forward_scope():
            seen["a"] = L._scoped(owner, "k", lambda: torch.zeros(1, requires_grad=True))
            enter_a.set(); enter_b.wait(5)
        exit_a.set()

    def worker_b():
        enter_a.wait(5)
        with L.forward_scope():
            enter_b.set(); exit_a.wait(5)
            seen["b"] = L._scoped(owner, "k", lambda: torch.ones(1, requires_grad=True))
        seen["b_after"] = getattr(L._SCOPE, "cache", None)

    ta, tb = threading.Thread(target=worker_a), threading.Thread(target=worker_b)
    ta.start(); tb.start(); ta.join(10); tb.join(10)
    assert seen["a"] is not seen["b"] and float(seen["b"]) == 1.0 and seen["b_after"] is None
    assert getattr(L._SCOPE, "cache", None) is None


def test_training_path_selection_rules():
    import torch
    from ptgnn_amd import dense, layers as L
    assert L._edge_training_ok(128, 128) and not L._edge_training_ok(128, 48) and not L._edge_training_ok(100, 64)
    assert L._prefer_edge_path(625_130, 115_772, 17, 128, 128)          # Graph2Class batch: edge form
    assert not L._prefer_edge_path(1_100_000, 200_000, 1, 128, 128)     # config 2: per-node table
    x, w = torch.zeros(4, 8), torch.zeros(12, 8)
    assert not dense._kernel_dims_ok(x, w)                              # CPU tensors never take the HIP nodes


def test_shared_message_rows_entry_points_validate_and_back_off(lib, monkeypatch):
    """Host side of the shared-message-row form (ptgnn_amd_unique_sources / ptgnn_amd_edge_linear_shared_f32): argument
    checks happen before any HIP call; the launch table has the size the Python wrapper allocates; the bookkeeping
    backs off for UNIQUE_BACKOFF plans after a minibatch whose (type, source) pairs were (nearly) all distinct."""
    from ptgnn_amd import ops
    assert lib.ptgnn_amd_edge_table_bytes() == 3 * 64 * 8 + 65 * 8 + 2 * 65 * 4 + 8       # StreamEdgeTable, 8-aligned
    assert lib.ptgnn_amd_unique_sources_workspace_bytes(1000, 3) > 3000                   # >= a flag byte per pair
    assert lib.ptgnn_amd_unique_sources(None, 10, 2, 3, 100, None, None, 10, None, None, None, 0, None) == -1
    assert b"unique_sources" in lib.ptgnn_amd_last_error()
    assert lib.ptgnn_amd_unique_sources(None, 0, 1, 3, 100, None, None, 0, None, None, None, 0, None) == -1   # T > 2^bits
    assert lib.ptgnn_amd_edge_linear_shared_f32(None, 4, 10, 64, None, None, 3, 64, 0, None, 64, None) == -1
    assert b"edge_linear_shared" in lib.ptgnn_amd_last_error()
    assert lib.ptgnn_amd_edge_linear_shared_supported(128, 128, 17) == 1
    assert lib.ptgnn_amd_edge_linear_shared_supported(128, 128, 65) == 0       # one table holds 64 edge types
    assert lib.ptgnn_amd_edge_linear_shared_supported(100, 128, 17) == 0       # not a shape of the streaming edge GEMM

    class Fake:
        def __init__(self, edges, rows):
            self.num_edges, self._r = edges, rows

        def rows(self):
            return self._r
    monkeypatch.setattr(ops, "_UNIQ_PENDING", [Fake(1000, None), Fake(1000, 800)])
    monkeypatch.setattr(ops, "_UNIQ_SKIP", [0])
    ops._poll_unique_stats()
    assert ops._UNIQ_SKIP[0] == 0 and len(ops._UNIQ_PENDING) == 1          # 20 % saved: keep going; one still in flight
    ops._UNIQ_PENDING.append(Fake(1000, 990))
    ops._poll_unique_stats()
    assert ops._UNIQ_SKIP[0] == ops.UNIQUE_BACKOFF                          # 1 % saved: skip the next plans


def test_committed_bench_line_keeps_the_driver_contract():
    """The newest committed N = 1 bench line (profiles/r*_bench_n1.json) carries every field the driver and the
    judge read, with self-consistent numbers: value = E / (ms_per_step / layers), roofline.frac = achieved / peak
    <= 1, a cpu_baseline of kind "port", parity inside its tolerance."""
    import glob
    import json
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")))
    if not lines:
        pytest.skip("no committed bench line")
    d = json.load(open(lines[-1]))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    cfg = d["config"]
    per_layer_s = d["ms_per_step"] * 1e-3 / cfg["mp_layers_per_step"]
    edges = cfg["edges_per_gpu"]
    if "edges_per_minibatch" in cfg:      # round 5: the timed steps rotate over several minibatches (step i: minibatch i mod R)
        epm = cfg["edges_per_minibatch"]
        edges = sum(epm[(d["warmup"] + i) % len(epm)] for i in range(d["steps"])) / d["steps"]
    assert abs(d["value"] - edges / per_layer_s) <= 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] <= 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["parity"]["max_abs"] <= d["parity"]["tol"] == 1e-5


def test_only_the_c_abi_is_exported(lib):
    """-fvisibility=hidden + csrc/exports.map: the dynamic symbol table holds the header's ptgnn_amd_* functions and
    nothing else (no mangled ptgnn_amd::stream_*, set_error, libstdc++ instantiations, __hip_cuid_*)."""
    import subprocess
    from ptgnn_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert names == declared_symbols(), sorted(set(names) ^ set(declared_symbols()))


def test_library_carries_no_vendor_sort_or_scan(lib):
    """The plan build is hand-written end to end: no rocPRIM symbol or kernel name in the product .so."""
    import subprocess
    from ptgnn_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rocprim" not in out.lower()
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"rocprim" not in blob, "a rocPRIM kernel or symbol is linked into libptgnn_amd.so"
    src = open(os.path.join(os.path.dirname(_lib.LIB_PATH), "csr_build.hip")).read()
    assert "#include <rocprim" not in src and "hipcub" not in src


def test_plan_build_refuses_sizes_beyond_the_int32_plan_format(lib):
    """`E < 2^31`, `num_src_rows * 2^type_bits < 2^31` per GPU.  The check is an argument check: it answers
    with PTGNN_AMD_EUNSUPPORTED (and a message naming the numbers) before anything touches the device, so it runs
    here without a GPU; the pointers are never dereferenced."""
    import ctypes
    fake = ctypes.c_void_p(0x1000)

    def build(counts, num_nodes, num_src_rows=0):
        T = len(counts)
        ptrs = (ctypes.c_void_p * T)(*[0x1000] * T)
        cnts = (ctypes.c_int64 * T)(*counts)
        return lib.ptgnn_amd_csr_build(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(ptrs, ctypes.c_void_p),
                                       ctypes.cast(cnts, ctypes.c_void_p), T, num_nodes, num_src_rows, 0, fake, fake,
                                       fake, None, 0, None, None, None, None, fake, 1 << 20, None)

    assert build([1 << 30, 1 << 30], 1000) == -2                      # E = 2^31
    assert b"exceed the int32 plan format" in lib.ptgnn_amd_last_error()
    assert build([10] * 32, 1 << 26) == -2                             # rows * 2^type_bits = 2^26 * 2^5 = 2^31
    assert build([10], 1 << 31) == -2                                  # rows = 2^31
    assert build([10] * 17, 1000, num_src_rows=1 << 27) == -2          # halo table: source rows * 2^5 >= 2^31


def test_parity_rollup_names_the_configs_on_the_relaxed_bar():
    """the top-level `parity` of the bench line says which configs pass on the float64-attributed bar
    only (`configs_on_relaxed_bar`), ANDs `strict_1e-5` over every config checked and lists failures."""
    from benchmarks.common import attributed_parity, parity_rollup
    strict = {"max_abs": 1.2e-6, "tol": 1e-5, "strict_1e-5": True}
    got = torch.tensor([[1.0, 2.0]])
    relaxed = attributed_parity(got + 5e-5, got, (got + 4e-5).double())       # ours 1e-5 from exact, the oracle 4e-5
    assert relaxed["ok"] and not relaxed["strict_1e-5"]
    failed = attributed_parity(got + 5e-4, got, got.double())
    assert not failed["ok"]
    roll = parity_rollup(strict, {"config2": {"max_abs": 2e-6, "tol": 1e-5}, "config4": relaxed,
                                  "config5_shard": {"ok": True, "ggnn": {"strict_1e-5": False, "ok": True},
                                                    "mlp_mp": {"strict_1e-5": True, "ok": True}},
                                  "config1": {"ggnn64": strict, "ppi_arch_mlp256": strict}})
    assert roll["max_abs"] == 1.2e-6 and roll["strict_1e-5"] is False and roll["configs_failed"] == []
    assert roll["configs_on_relaxed_bar"] == ["config4", "config5_shard.ggnn"]
    assert set(roll["configs_checked"]) == {"primary", "config2", "config4", "config5_shard.ggnn", "config5_shard.mlp_mp",
                                            "config1.ggnn64", "config1.ppi_arch_mlp256"}
    assert parity_rollup(strict, {"config2": strict})["strict_1e-5"] is True
    assert parity_rollup(strict, {"config4": failed})["configs_failed"] == ["config4"]


def test_bench_self_launch_refuses_more_ranks_than_gpus_with_a_clear_message():
    """`python bench.py --gpus N` without WORLD_SIZE starts its own ranks (round 6); on a node with fewer GPUs it says so
    instead of dying inside a rendezvous.  (The launch itself is exercised on the GPU box: tests/test_gpu_two_rank.py.)"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PTGNN_AMD_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 64: this node shows" in (r.stderr + r.stdout), r.stderr[-2000:]
    # under a launcher with a mismatching world size the message names both ways of starting it
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and ("WORLD_SIZE=3" in r.stderr or "needs an MI355X" in r.stderr), r.stderr[-2000:]


def test_bench_stdout_carries_only_the_json_line():
    """benchmarks.common.hold_stdout / emit_line: whatever C stdio (the collective backend's banner) or Python prints while
    the run is in progress lands on stderr; stdout gets the one JSON line."""
    import subprocess
    import sys
    code = ("import ctypes, sys\nsys.path.insert(0, %r)\nfrom benchmarks import common as C\nC.hold_stdout()\n"
            "ctypes.CDLL(None).printf(b'RCCL version : banner through C stdio\\n')\nprint('python noise')\n"
            "C.emit_line('{\"ok\": 1}')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout == '{"ok": 1}\n', (r.stdout, r.stderr[-500:])
    assert "banner through C stdio" in r.stderr and "python noise" in r.stderr
