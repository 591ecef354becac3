"""Shared plumbing of the bench legs: logging, the process group, the contract's timed region, the per-kernel
HIP-event pass and the roofline table.  `bench.py` at the repo root is the driver entry; the legs live beside this file."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_T0 = time.perf_counter()


def _log(msg):
    """Progress on stderr (stdout carries exactly one JSON line)."""
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}\n")
        sys.stderr.flush()


HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)


PARITY_TOL = 1e-5   # BASELINE.json north_star: fp32 node states within 1e-5 of the reference CPU path


_GROUP = {"up": False}


def have_group() -> bool:
    return _GROUP["up"]


def _init_group(rank, world, local):
    import torch.distributed as dist
    backend = os.environ.get("PTGNN_AMD_BENCH_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:     # not under a launcher (world = 1): any free port, never a fixed one
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    _GROUP["up"] = True


def dist_setup(args):
    """RANK / LOCAL_RANK / WORLD_SIZE from the launcher's environment; LOCAL_RANK -> device; the process group (RCCL)
    when there is more than one rank or the run itself is a sharded one (`--force-sharded`)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Validation hook for 1-GPU boxes: PTGNN_AMD_BENCH_BACKEND=gloo PTGNN_AMD_BENCH_SHARE_GPU=1 runs the N > 1 code
    # (sharding, all-to-all, rank reductions) with every rank on cuda:0 -- RCCL refuses two ranks per device.  The
    # numbers of such a run mean nothing; the default (one GPU per rank over RCCL) is what the driver launches.
    if os.environ.get("PTGNN_AMD_BENCH_SHARE_GPU", "0") not in ("", "0"):
        local = 0
        # two PROCESSES time-slicing one GPU turn every cross-stream event wait into a scheduling quantum (measured:
        # 42 -> 345 ms per cfg5 step with the aggregation's side streams engaged): keep the library on one stream here
        os.environ.setdefault("PTGNN_AMD_HUB_STREAM", "0")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` "
                         "(self-launching) or under torch.distributed.run with --nproc-per-node equal to --gpus")
    if world > 1 or args.force_sharded or args.sharded_variants:
        _init_group(rank, world, local)
    else:
        torch.cuda.set_device(0)
    return rank, world, torch.device("cuda", local if world > 1 else 0)


def drop_group() -> None:
    if _GROUP["up"]:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        _GROUP["up"] = False


def late_group(dev) -> None:
    """World-1 process group for the sharded leg of a plain N = 1 run, created right before that leg (inside its
    try/except): a collective backend that fails to come up then costs that leg, never the primary line."""
    if not _GROUP["up"]:
        _init_group(0, 1, dev.index or 0)


def barrier_sync(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def sum_over_ranks(value, world, dev):
    if world == 1:
        return value
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def max_over_ranks(seconds, world, dev):
    if world == 1:
        return seconds
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())



# ------------------------------------------------------------------------------------------------
def timed_region(step_fn, steps, warmup, world, dev):
    """The contract's timed region: W untimed warm-up steps, then EXACTLY K steps bracketed by
    barrier + synchronize on both sides, max over ranks.  No per-kernel instrumentation runs here:
    a HIP event pair around every launch costs ~0.1 ms of queue serialisation per kernel on this
    stack and would be charged to `value`.  The per-kernel HIP-event pass runs right after, over the
    same K steps of the same inputs (`kernel_pass`)."""
    for _ in range(warmup):
        step_fn()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    barrier_sync(world)
    dt = time.perf_counter() - t0
    return max_over_ranks(dt, world, dev), kernel_pass(step_fn, steps, world)


def kernel_pass(step_fn, steps, world):
    """K more steps with a HIP-event bracket around every C-ABI launch (events are recorded on the
    stream the kernels are launched on); feeds `roofline` and `kernels`."""
    from ptgnn_amd import ops
    timer = ops.KernelTimer()
    barrier_sync(world)
    ops.set_kernel_timer(timer)
    for _ in range(steps):
        step_fn()
    barrier_sync(world)
    ops.set_kernel_timer(None)
    return timer.summary()


def kernel_table(summary):
    table = {}
    for name, d in summary.items():
        ms = d["ms"] / d["calls"]
        row = {"calls": d["calls"], "avg_ms": round(ms, 5)}
        if name in ("linear", "gru_cell", "edge_linear", "edge_linear_shared", "edge_weight_grad", "linear_weight_grad"):
            tf = d["flops"] / d["calls"] / (ms * 1e-3) / 1e12
            row.update(bound="mfma", achieved=round(tf, 2), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                       frac=round(tf / MFMA_F32_PEAK_TFLOPS, 4))
        else:
            gbs = d["bytes"] / d["calls"] / (ms * 1e-3) / 1e9
            row.update(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                       frac=round(gbs / HBM_PEAK_GBS, 4))
        row["algorithmic_bytes_per_launch"] = round(d["bytes"] / d["calls"])
        row["total_ms"] = round(d["ms"], 4)
        table[name] = row
    return table


# bench kernel bracket -> device kernel names it may resolve to (streaming core first, round-1 tile kernels second)
PMC_KERNEL = {"linear": ("k_stream_linear", "k_linear_tlp"), "gather_reduce": ("k_gather_reduce",),
              "gru_cell": ("k_stream_gru", "k_gru"), "edge_linear": ("k_stream_edge", "k_edge_linear")}


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of THIS command
    (profiles/r0N_<workload>_traffic.json, written by scripts/gpu_profile.sh + summarize_prof.py:
    FETCH_SIZE x2 wide-read correction + WRITE_SIZE, separate --pmc runs; newest round first).  PMC counters
    cannot be read from inside a plain bench run, so the figure is the profiled one and names its source; null
    when no profile holds the kernel."""
    root = os.path.join(ROOT, "profiles")
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(root, f"{rnd}_{workload}_traffic.json")
        try:
            with open(path) as f:
                prof = json.load(f)
        except (OSError, ValueError):
            continue
        for name in PMC_KERNEL.get(kernel, ()):
            row = prof.get("kernels", {}).get(name)
            if row:
                return {"traffic": row["hbm_bytes_per_launch"], "traffic_unit": "bytes/launch", "traffic_kernel": name,
                        "traffic_source": f"profiles/{rnd}_{workload}_traffic.json ({prof['source']})"}
    return {"traffic": None}



def repeat_stats(step_fn, steps, blocks=5):
    """min / median ms per step over repeated K-step blocks (robust to DVFS and first-touch effects); the
    contract's `ms_per_step` stays the single timed region."""
    per = []
    for _ in range(blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) / steps * 1e3)
    per.sort()
    return {"blocks": blocks, "steps_per_block": steps, "ms_per_step_min": round(per[0], 4),
            "ms_per_step_median": round(per[len(per) // 2], 4), "ms_per_step_max": round(per[-1], 4)}



def attributed_parity(got, want, exact=None, tol=PARITY_TOL):
    """The parity record every config of the line carries: `max_abs` against the fp32 oracle, `strict_1e-5` = the
    north star's literal bar, and -- where fp32 itself cannot hold that bar (stacked LayerNorms, 1e3-term sums) --
    the attribution against a float64 evaluation `exact`: the HIP path may sit no further from exact arithmetic than
    2 x the reference's own fp32 arithmetic does.  `ok` = strict, or attributed."""
    err = float((got - want).abs().max())
    rec = {"max_abs": err, "tol": tol, "strict_1e-5": bool(err <= tol)}
    if exact is not None:
        ours = float((got.double() - exact).abs().max())
        ref = float((want.double() - exact).abs().max())
        rec.update(ours_vs_fp64=ours, oracle_vs_fp64=ref)
        rec["ok"] = bool(err <= tol or ours <= max(tol, 2.0 * ref))
    else:
        rec["ok"] = bool(err <= tol)
    return rec


def parity_rollup(primary, by_config):
    """Top-level `parity` of the line: the primary workload's record plus, over EVERY config the run checked,
    `strict_1e-5` (AND) and the names of the configs that pass on the float64-attributed bar only."""
    out = dict(primary)
    checked, relaxed, failed = [], [], []

    def visit(name, rec):
        if not isinstance(rec, dict):
            return
        if "strict_1e-5" in rec or "max_abs" in rec:
            strict = rec.get("strict_1e-5", rec.get("max_abs", 1.0) <= rec.get("tol", PARITY_TOL))
            ok = rec.get("ok", strict)
            checked.append(name)
            if not ok:
                failed.append(name)
            elif not strict:
                relaxed.append(name)
            return
        for k, v in rec.items():          # nested records (cfg5: ggnn / mlp_mp; cfg1: per architecture)
            if isinstance(v, dict):
                visit(f"{name}.{k}", v)
    visit("primary", primary)
    for name, rec in by_config.items():
        visit(name, rec)
    out["configs_checked"] = checked
    out["strict_1e-5"] = not relaxed and not failed
    out["configs_on_relaxed_bar"] = relaxed
    out["configs_failed"] = failed
    return out


# stdout carries exactly ONE line, the JSON.  The collective backend prints a banner through C stdio (RCCL: "RCCL version :
# ...", five lines, block-buffered on a pipe until some later flush), and since round 6 a process group exists in every run,
# N = 1 included.  So file descriptor 1 points at stderr for the whole run -- C stdio and Python's sys.stdout alike -- and is
# handed back for the one line that belongs there.
_STDOUT = {"fd": None}


def hold_stdout() -> None:
    if _STDOUT["fd"] is None:
        sys.stdout.flush()
        _STDOUT["fd"] = os.dup(1)
        os.dup2(2, 1)


def emit_line(line: str) -> None:
    """Drain whatever C stdio still buffers (into stderr), restore stdout, print the line."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    if _STDOUT["fd"] is not None:
        os.dup2(_STDOUT["fd"], 1)
        os.close(_STDOUT["fd"])
        _STDOUT["fd"] = None
    print(line, flush=True)
