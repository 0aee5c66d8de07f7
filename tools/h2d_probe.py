"""Developer probe: pinned host-to-device copy rate with one-behind event waits, and what a dependent kernel on
another stream / a small device-to-host copy do to it."""
import time
import torch

mb = 64
h = [torch.empty(mb * 1024 * 1024, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
d = [torch.empty_like(h[0], device="cuda") for _ in range(2)]
ho = [torch.empty(512 * 1024, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
n = 24


def rate(dt):
    return round(n * mb * 1.048576e-3 / dt, 2)


s_in, s_out, s_c = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
ev = [torch.cuda.Event(), torch.cuda.Event()]
ev_done = [torch.cuda.Event(), torch.cuda.Event()]
for mode in ("warm", "copy only", "copy + kernel on default stream", "copy + kernel on a side stream",
             "copy + kernel (default) + D2H on third stream", "copy + kernel (default) + D2H same stream"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        b = k & 1
        with torch.cuda.stream(s_in):
            d[b].copy_(h[b], non_blocking=True)
            ev[b].record(s_in)
        last = ev[b]
        if "kernel" in mode:
            cs = s_c if "side" in mode else torch.cuda.default_stream()
            cs.wait_event(ev[b])
            with torch.cuda.stream(cs):
                r = d[b][: 512 * 1024] + 1
                if "same stream" in mode:
                    ho[b].copy_(r, non_blocking=True)
                ev_done[b].record(cs)
            last = ev_done[b]
            if "third" in mode:
                s_out.wait_event(ev_done[b])
                with torch.cuda.stream(s_out):
                    ho[b].copy_(r, non_blocking=True)
                    ev_done[b].record(s_out)
        if k > 0:
            (ev_done if "kernel" in mode else ev)[(k - 1) & 1].synchronize()
    torch.cuda.synchronize()
    print(mode, rate(time.perf_counter() - t0), "GB/s")
