"""tools/probe/mfma_power.hip: sustained TFLOP/s of a bare MFMA stream (8 waves on every CU) for 16x16x32 and 32x32x16 fp16, random
normal / all-zero operands, ~30 ms per measurement after a warm-up of the same length.  Build first:
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/probe/mfma_power.hip -o tools/probe/libmfma_power.so"""
import ctypes, os, sys
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libmfma_power.so"))
lib.mfma_power_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
cus = torch.cuda.get_device_properties(0).multi_processor_count
blocks, threads = cus, 512
sink = torch.zeros(blocks * threads, device=dev)
data = {"random": torch.randn(blocks * threads * 8 * 8, device=dev).half().view(torch.int32), "zeros": torch.zeros(blocks * threads * 8 * 4, dtype=torch.int32, device=dev)}
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
flops_per_iter_wave = 16 * 2 * 16 * 16 * 32      # 16 MFMAs of 16x16x32 (= 8 of 32x32x16)
for name, seed in data.items():
    for shape in (16, 32):
        st = torch.cuda.current_stream().cuda_stream
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.mfma_power_launch(shape, seed.data_ptr(), sink.data_ptr(), blocks, iters, st)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0
        ms = e0.elapsed_time(e1)
        tf = flops_per_iter_wave * iters * blocks * 8 / (ms * 1e-3) / 1e12
        print(f"{name:7s} {'16x16x32' if shape == 16 else '32x32x16'}: {ms:7.2f} ms  {tf:7.1f} TFLOP/s", flush=True)
