"""raw PCIe rates of the box: pinned H2D, D2H, both at once; large (256 MB) and frame-sized (2 MB) copies"""
import time, torch
def rate(fn, nbytes, reps):
    fn(); torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - a) / 1e9
for mb, reps in ((256, 10), (2, 400)):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory(); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda"); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def up():
        with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    def down():
        with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
    def both():
        up(); down()
    print(f"{mb} MB copies: H2D {rate(up, n, reps):.1f} GB/s, D2H {rate(down, n, reps):.1f} GB/s, both {rate(both, 2 * n, reps):.1f} GB/s (sum)")
