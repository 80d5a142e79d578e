"""streams_apart against the overlap matrix: burn a few torch streams first, then pick."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.set_device(0)
from ts_trajgen_amd.search import streams_apart
burn = [torch.cuda.Stream() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3)]
for s in burn:
    with torch.cuda.stream(s):
        torch.zeros(8, device='cuda')
got = streams_apart('cuda:0', 2)
print('picked', [hex(s.cuda_stream) for s in got])
import time
def both(a, b):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(a): torch.cuda._sleep(4_000_000)
    with torch.cuda.stream(b): torch.cuda._sleep(4_000_000)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
print('pair %.2f ms, with default %.2f / %.2f ms' % (both(got[0], got[1]), both(torch.cuda.default_stream(), got[0]), both(torch.cuda.default_stream(), got[1])))
