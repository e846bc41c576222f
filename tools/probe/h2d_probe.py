"""How fast does a 31 MB float32 array (the HR-sized inputs of a 40-sample training batch) reach the device from pageable numpy
memory, from page-locked memory, and from pageable memory through page-locked staging filled by several threads?"""
import threading
import time

import numpy as np
import torch

n = 40 * 256 * 256 * 3
a = np.random.default_rng(0).random(n, dtype=np.float32)
d = torch.empty(n, dtype=torch.float32, device="cuda")
t = torch.from_numpy(a)
pin = torch.empty(n, dtype=torch.float32).pin_memory()
torch.cuda.synchronize()


def timeit(f, reps=10):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"pageable -> device: {timeit(lambda: d.copy_(t)):.2f} ms")
print(f"pinned   -> device: {timeit(lambda: d.copy_(pin, non_blocking=True)):.2f} ms")
print(f"memcpy pageable -> pinned, 1 thread: {timeit(lambda: pin.copy_(t)):.2f} ms")
pn = pin.numpy()
for nt in (2, 4, 8):
    cuts = [n * i // nt for i in range(nt + 1)]

    def par():
        th = [threading.Thread(target=np.copyto, args=(pn[cuts[i]:cuts[i + 1]], a[cuts[i]:cuts[i + 1]])) for i in range(nt)]
        for x in th:
            x.start()
        for x in th:
            x.join()

    print(f"memcpy pageable -> pinned, {nt} threads: {timeit(par):.2f} ms")
