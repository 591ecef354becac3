"""Run ON THE GPU BOX: python scripts/experiments/vmem_issue_probe.py  (see vmem_issue_probe.hip)"""
import ctypes
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libvmem_issue_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC",
                       os.path.join(HERE, "vmem_issue_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
table_rows, out_rows = 115772, 1 << 20                    # 59 MB table of 512-B rows; 512 MB store target
table = torch.randn(table_rows, 128, device="cuda")
out = torch.empty(out_rows, 128, device="cuda")
sink = torch.empty(256 * 512, device="cuda")
st = torch.cuda.current_stream().cuda_stream
names = {0: "bare fp32 MFMA loop", 1: "+1 store / 16 MFMA (resident)", 2: "+2 stores (resident)", 3: "+4 stores (resident)",
         4: "+1 store (streaming)", 5: "+2 stores (streaming)", 6: "+4 stores (streaming)", 7: "+1 gather load",
         8: "+2 gather loads", 9: "+4 gather loads", 10: "+1 store +1 gather (edge GEMM mix)", 11: "+2 stores +2 gathers",
         12: "+1 nt store (streaming)", 13: "+2 nt stores (streaming)", 14: "+4 nt stores (streaming)",
         15: "+1 buffer store aux 0 (streaming)", 16: "+1 buffer store sc1", 17: "+1 buffer store sc0 sc1",
         18: "+1 buffer store sc0", 19: "+1 buffer store sc0 nt", 20: "+2 buffer stores sc1", 21: "+2 buffer stores sc0 sc1",
         22: "+1 global_load, used 1 iter later", 23: "+1 buffer_load, used 1 iter later", 24: "+2 global_loads (later)",
         25: "+2 buffer_loads (later)", 26: "+4 global_loads (later)", 27: "+4 buffer_loads (later)"}
out_bytes = (1 << 20) * 128 * 4
assert out_bytes < (1 << 31) * 1                      # the buffer descriptor of the probe addresses 2 GB
iters, blocks, base = 2000, 256, None
for variant, name in names.items():
    def call():
        rc = lib.vmem_probe(ctypes.c_void_p(table.data_ptr()), ctypes.c_int64(table_rows), ctypes.c_void_p(out.data_ptr()),
                            ctypes.c_int64(out_rows), iters, blocks, variant, ctypes.c_void_p(sink.data_ptr()),
                            ctypes.c_void_p(st))
        assert rc == 0
    for _ in range(2):
        call()
    evs = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); call(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[2]
    flops = blocks * 8 * iters * 16 * 4096
    per_iter_cycles = ms * 1e-3 / iters * 2.18e9            # wall cycles per 16-MFMA group of ONE wave (2 waves share a SIMD)
    if base is None:
        base = per_iter_cycles
    n_mem = {1: 1, 2: 2, 3: 4, 4: 1, 5: 2, 6: 4, 7: 1, 8: 2, 9: 4, 10: 2, 11: 4, 12: 1, 13: 2, 14: 4, 15: 1, 16: 1, 17: 1, 18: 1, 19: 1, 20: 2, 21: 2, 22: 1, 23: 1, 24: 2, 25: 2, 26: 4, 27: 4}.get(variant, 0)
    extra = (per_iter_cycles - base) / n_mem if n_mem else 0.0
    print(f"{name:36s}: {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s   {per_iter_cycles:7.0f} cycles per 16-MFMA group per wave"
          f"   (+{extra:5.0f} per memory instruction)", flush=True)
