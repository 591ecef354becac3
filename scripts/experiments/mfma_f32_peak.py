"""Run ON THE GPU BOX: python scripts/experiments/mfma_f32_peak.py  (builds the probe with hipcc first)."""
import ctypes
import os
import subprocess
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libmfma_f32_peak.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC",
                       os.path.join(HERE, "mfma_f32_peak.hip"), "-o", so])
lib = ctypes.CDLL(so)
out = torch.empty(512 * 4096, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for data in ("random", "zero"):
    src = (torch.randn(4096, device="cuda") if data == "random" else torch.zeros(4096, device="cuda"))
    for kind, name, flop_per in ((0, "32x32x2", 4096), (1, "16x16x4", 2048)):
        for threads, blocks in ((256, 256), (512, 256), (256, 512), (256, 1024)):
            iters = 4000
            nacc = 4 if kind == 0 else 8
            for _ in range(2):
                lib.mfma_peak(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()), iters, threads, blocks,
                              kind, ctypes.c_void_p(st))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                lib.mfma_peak(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()), iters, threads, blocks,
                              kind, ctypes.c_void_p(st))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            waves = threads // 64 * blocks
            flops = waves * iters * 8 * nacc * flop_per
            print(f"{data:6s} {name} threads={threads} blocks={blocks}: {dt * 1e3:.2f} ms  {flops / dt / 1e12:.1f} TFLOP/s",
                  flush=True)
