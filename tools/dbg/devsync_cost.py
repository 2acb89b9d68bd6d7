# developer: what an idle torch.cuda.synchronize() costs as a function of the number of HIP streams that exist
import time, torch
x = torch.zeros(1, device="cuda")
def cost():
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[100] * 1e6
print("streams 0 extra: %.1f us" % cost())
keep = []
for k in (1, 2, 4, 8, 16):
    while len(keep) < k:
        s = torch.cuda.Stream(); keep.append(s)
        with torch.cuda.stream(s): x.add_(1)
    torch.cuda.synchronize()
    print("streams %d extra: %.1f us" % (k, cost()))
