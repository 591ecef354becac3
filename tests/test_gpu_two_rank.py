"""N > 1 with REAL processes on the one GPU a test box has: `torch.distributed.run` starts W ranks of
tests/two_rank_gpu_check.py (all on cuda:0, process group gloo -- RCCL refuses two ranks per device), which
check the sharded layers / stack / training step against the unsharded ones.  See that file's docstring."""
import os
import re
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_paths_over_real_process_group(world):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "two_rank_gpu_check.py")]
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = (r.stdout + "\n" + r.stderr)[-6000:]
    assert r.returncode == 0, tail
    # 8 layer cases + 1 stack + 2 training cases + 2 graph-boundary cases + 1 planner case + 1 fuzz case per rank (ranks may
    # interleave their lines)
    assert len(re.findall(r"rank \d+ ok ", r.stdout)) == 15 * world, tail


@pytest.mark.parametrize("world", [4, 8])
def test_powerlaw_shard_with_hub_rows_at_4_and_8_ranks(world):
    """SURVEY.md 8e at the rank counts BASELINE configs 4 and 5 name (no multi-GPU box behind `gpurun`: 4 and 8 REAL
    processes share cuda:0 over gloo): a cfg5-shaped power-law graph whose hub rows have sources on every rank, through
    the HIP index pass, the halo all-to-all(v) and the hub / long-row aggregation -- see `case_powerlaw_hubs`."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "two_rank_gpu_check.py")]
    # processes that share one GPU turn every cross-stream event wait into a scheduling quantum: one stream per process
    env = dict(os.environ, OMP_NUM_THREADS="2", TWO_RANK_CASES="powerlaw", PTGNN_AMD_HUB_STREAM="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + "\n" + r.stderr)[-6000:]
    assert r.returncode == 0, tail
    assert len(re.findall(r"rank \d+ ok powerlaw_", r.stdout)) == 3 * world, tail
    assert sum(int(m) for m in re.findall(r"powerlaw_ggnn_sum hubs_here=(\d+)", r.stdout)) > 0, tail


def test_bench_watchdog_emits_the_primary_line_when_a_rank_stalls():
    """bench.py --gpus 2 (gloo, both ranks on cuda:0 -- the validation hook for 1-GPU boxes) with rank 1 stalled right
    before the sharded cut-edge variants: rank 0 blocks in their first collective, the per-rank watchdog ends every
    rank cleanly and rank 0 still prints the ONE JSON line with the primary measurement (distributedtrainer.py:250-297
    is the reference's multi-GPU entry; a hung collective must not cost the driver its SCALE line)."""
    import json
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-secondary"]
    env = dict(os.environ, OMP_NUM_THREADS="4", PTGNN_AMD_BENCH_BACKEND="gloo", PTGNN_AMD_BENCH_SHARE_GPU="1",
               PTGNN_AMD_BENCH_FAULT="hang:1", PTGNN_AMD_BENCH_VARIANT_DEADLINE="15")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = (r.stdout + "\n" + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, tail
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and "timed out" in res["cut_edges_variant"]["error"], tail


def test_bench_two_ranks_runs_the_sharded_variants_end_to_end():
    """`python bench.py --gpus 2` started PLAINLY -- no launcher, no WORLD_SIZE: bench.py starts its own ranks under
    torch.distributed.run on 127.0.0.1 (round 6; the reference's multi-GPU entry spawns its ranks from one process too,
    distributedtrainer.py:250-265), here over gloo with both ranks on cuda:0 (RCCL refuses two ranks per device): the
    primary weak-scaling line plus BOTH dst-range-sharded workloads with real cut edges -- the cfg5 shard (HIP index pass,
    halo all-to-all, two-block mode) and the cfg4 stack in its graph-boundary and its through-graphs partition -- must
    complete and explain themselves in the one JSON line.  (The launcher form is covered by the watchdog test above.)"""
    import json
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-secondary"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="4", PTGNN_AMD_BENCH_BACKEND="gloo", PTGNN_AMD_BENCH_SHARE_GPU="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + "\n" + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, tail
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["value"] > 0
    assert [p[0] for p in res["rccl_ranks_seen"]] == [0, 1]
    v = res["cut_edges_variant"]
    assert "error" not in v, tail
    c5 = v["cfg5_shard"]
    assert "error" not in c5 and "error" not in v["cfg4_stack"], tail
    # (in this validation mode the cfg5 variant runs on a tenth of the shard: 125 k nodes per rank)
    assert c5["ms_per_step"] > 0 and c5["halo_rows_all_ranks"] > 100_000 and not c5["no_cut"] and c5["all_to_all_ms"] > 0
    c4 = v["cfg4_stack"]
    assert c4["graph_boundaries"]["ms_per_forward"] > 0
    tg = c4["through_graphs"]
    assert tg["ms_per_forward"] > 0 and not tg["no_cut"] and tg["halo_rows_all_ranks"] > 0
    # round 4: the split's key figures also sit in `config`, where a SCALE record of the driver reads them
    lift = res["config"]["dst_range_split"]
    assert lift["cfg5_shard"]["ms_per_step"] == c5["ms_per_step"] and lift["cfg5_shard"]["all_to_all_ms"] > 0
    assert lift["cfg4_stack_through_graphs"]["ms_per_forward"] == tg["ms_per_forward"]
    assert lift["cfg4_stack_graph_boundaries"]["ms_per_forward"] > 0 and lift["collective_backend"] == "gloo"
    # round 6: the north-star split's own edges/s is a first-class key with its one-GPU counterpart beside it
    top = res["sharded_cfg5"]
    assert top["n_gpus"] == 2 and top["edges_per_sec_per_layer"] == c5["edges_per_sec_per_layer"]
    assert top["one_gpu_edges_per_sec_per_layer"] > 0 and 0 < top["vs_n_times_one_gpu"] < 2 and top["cut_fraction"] == 0.5
