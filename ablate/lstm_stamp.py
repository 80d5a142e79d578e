import ctypes as C, os, sys
os.environ['TSG_LIB_OVERRIDE'] = os.path.abspath('ablate/libtsg_lstamp.so')
sys.path.insert(0, os.getcwd())
sys.argv = ['bench.py', '--warmup', '0', '--cpu-seconds', '0', '--profile-every', '0'] + sys.argv[1:]
import runpy
try:
    runpy.run_path('bench.py', run_name='__main__')
finally:
    L = C.CDLL(os.environ['TSG_LIB_OVERRIDE'])
    out = (C.c_ulonglong * 1024)()
    L.tsg_debug_read_lstm_stamps(out, 0)
    import numpy as np
    v = np.array(list(out), dtype=np.float64).reshape(2, 64, 8).sum(axis=1).reshape(-1)
    for l in (0, 1):
        n = max(1, v[8 * l + 7])
        print('layer', l, 'waves', n, {k: round(v[8 * l + i] / n * 10 / 1e3, 2) for i, k in
              enumerate(['rows', 'prologue', 'kloop', 'acc-wait', 'epilogue'])}, 'us per wave')
