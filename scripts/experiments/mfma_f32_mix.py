"""Run ON THE GPU BOX: python scripts/experiments/mfma_f32_mix.py"""
import ctypes, os, subprocess, time
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libmfma_f32_mix.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC",
                       os.path.join(HERE, "mfma_f32_mix.hip"), "-o", so])
lib = ctypes.CDLL(so)
out = torch.empty(512 * 256, device="cuda")
src = torch.randn(4096, device="cuda")
st = torch.cuda.current_stream().cuda_stream
names = {0: "bare fp32", 1: "+16 v_fma / 16 mfma", 2: "+64 v_fma / 16 mfma", 3: "+4 ds_read_b128", 4: "+16 ds_read_b128",
         5: "+2 global_load_x4", 6: "+8 global_load_x4", 7: "+16 v_fma +4 ds +2 gl", 8: "bare bf16 32x32x16",
         9: "bf16 +64 v_fma", 10: "+256 v_fma / 16 mfma"}
for variant, name in names.items():
    iters = 4000
    call = lambda: lib.mfma_mix(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()), iters, 256, variant,
                                ctypes.c_void_p(st))
    for _ in range(2): call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    per = 32 * 32 * 16 * 2 if variant in (8, 9) else 4096
    flops = 8 * 256 * iters * 16 * per
    print(f"{name:26s}: {dt * 1e3:7.2f} ms  {flops / dt / 1e12:7.1f} TFLOP/s   ({dt / iters / 16 * 2.4e9 / 2:.1f} clk per MFMA per wave-pair slot @2.4GHz)", flush=True)
