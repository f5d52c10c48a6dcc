import torch, time
for mb in (8, 32, 64, 128, 256):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        reps = max(4, 2048 // mb)
        t = time.perf_counter()
        for _ in range(reps): d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"H2D {mb} MB chunks: {n * reps / dt / 1e9:.1f} GB/s")
        t = time.perf_counter()
        for _ in range(reps): h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"D2H {mb} MB chunks: {n * reps / dt / 1e9:.1f} GB/s")
