"""Rate of the copies the end-to-end bench leg depends on: 44 MB device -> page-locked host, 11 MB host -> device."""
import time
import torch

dev = torch.device("cuda:0")
d = torch.empty(16 * 720 * 1280 * 3, dtype=torch.uint8, device=dev)
h = torch.empty_like(d, device="cpu").pin_memory()
hl = torch.empty(16 * 180 * 320 * 3, dtype=torch.float32).pin_memory()
dl = torch.empty_like(hl, device=dev)
s = torch.cuda.Stream()
for name, src, dst in (("D2H 44 MB", d, h), ("H2D 11 MB", hl, dl)):
    with torch.cuda.stream(s):
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            dst.copy_(src, non_blocking=True)
        s.synchronize()
        dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {dt * 1e3:.3f} ms = {src.numel() * src.element_size() / dt / 1e9:.1f} GB/s")
