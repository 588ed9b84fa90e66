"""Which of the loader's per-epoch device operations block the host?  Each is issued on a side stream while ~40 ms of work is
queued on the current stream; an operation that returns in microseconds does not synchronise."""
import time, torch
dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev)
index = torch.arange(50000, device=dev)
side = torch.cuda.Stream(device=dev)


def busy():
    for _ in range(12):
        a @ a


def probe(name, fn, warm=True):
    if warm:
        with torch.cuda.stream(side):
            fn()
        torch.cuda.synchronize()
    busy()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        out = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-44s returned after %8.1f us (queued work took %.1f ms)" % (name, (t1 - t0) * 1e6, (t2 - t0) * 1e3))
    return out


g = torch.Generator(device=dev)
probe("torch.Generator(device) + manual_seed", lambda: torch.Generator(device=dev).manual_seed(5))
probe("torch.randperm(50000, generator, device)", lambda: torch.randperm(50000, generator=g, device=dev))
out = torch.empty(50000, dtype=torch.int64, device=dev)
probe("torch.randperm(..., out=)", lambda: torch.randperm(50000, generator=g, device=dev, out=out))
probe("torch.rand(50000).argsort()", lambda: torch.rand(50000, generator=g, device=dev).argsort())
probe("torch.rand(50000).sort()", lambda: torch.rand(50000, generator=g, device=dev).sort())
probe("index[order]", lambda: index[out])
probe("torch.index_select(index, 0, order)", lambda: torch.index_select(index, 0, out))
probe("torch.cuda.Event().record()", lambda: torch.cuda.Event().record(side))
