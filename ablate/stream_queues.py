"""Which torch streams overlap with which?  Two spin kernels (torch.cuda._sleep) on two streams take the time of one
when the streams sit on different hardware queues and the time of two when they share one."""
import time
import torch
torch.cuda.set_device(0)
N = 4_000_000
streams = [torch.cuda.Stream() for _ in range(8)]


def both(a, b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(N)
    if b is not None:
        with torch.cuda.stream(b):
            torch.cuda._sleep(N)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


both(streams[0], streams[1])
one = min(both(s, None) for s in streams)
print('one spin kernel: %.2f ms' % one)
for i, a in enumerate(streams):
    print(i, ' '.join('%4.1f' % (both(a, b) / one) if b is not a else '   -' for b in streams))
print('with the default stream:', ' '.join('%4.1f' % (both(torch.cuda.default_stream(), b) / one) for b in streams))
