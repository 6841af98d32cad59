import time, torch
torch.cuda.init()
for rep in range(3):
    t0 = time.perf_counter()
    bufs = [torch.empty(8 * 400, 3, 96, 96, dtype=torch.uint8, pin_memory=True) for _ in range(7)]
    t1 = time.perf_counter()
    a = [torch.empty(8 * 256000, dtype=torch.float32, pin_memory=True) for _ in range(7)]
    t2 = time.perf_counter()
    print(f"pinned alloc: 7 x 88 MB u8 {t1 - t0:.3f} s, 7 x 8 MB f32 {t2 - t1:.3f} s")
    src = torch.randint(0, 256, (400, 3, 96, 96), dtype=torch.uint8)
    t0 = time.perf_counter()
    for i in range(8):
        bufs[0][i * 400:(i + 1) * 400].copy_(src)
    t1 = time.perf_counter()
    d = bufs[0].to("cuda", non_blocking=True); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"collate 8 clips into pinned: {1e3 * (t1 - t0):.1f} ms; H2D 88 MB: {1e3 * (t2 - t1):.1f} ms")
    del bufs, a
