"""-DTSG_STAMP=1 -DTSG_STAMP_AT=3000 build (tools/build_variant.sh stamp "..."): per-lane time stamps of the frontier kernel
in one superstep (one record per search slot, no atomics): where a lane spends its time, how long waves live."""
import ctypes as C, os, sys
import numpy as np
os.environ['TSG_LIB_OVERRIDE'] = os.path.abspath('ablate/libtsg_stamp.so')
sys.path.insert(0, os.getcwd())
sys.argv = ['bench.py', '--ods', '65536', '--warmup', '0', '--cpu-seconds', '0', '--profile-every', '0', '--two-stage-ods', '0']
import runpy
try:
    runpy.run_path('bench.py', run_name='__main__')
finally:
    L = C.CDLL(os.environ['TSG_LIB_OVERRIDE'])
    out = (C.c_ulonglong * (32768 * 32))()
    L.tsg_debug_read_stamps(out, 0)
    v = np.frombuffer(out, dtype=np.uint64).reshape(32768, 32).astype(np.int64)
    v = v[v[:, 14] == 1]
    names = {0: 'first round trip', 1: 'relax', 2: 'loop top', 3: 'pop', 4: 'node record', 5: 'successors', 6: 'table',
             7: 'expansion stores', 16: 'relax: until the table entries arrived', 17: 'relax: pushes', 8: 'emit (incl. waiting for the wave)', 9: 'lane total'}
    print('lanes', len(v), 'relaxes per lane %.2f' % v[:, 12].mean(), 'pops per lane %.2f' % v[:, 13].mean(), 'pushes per lane %.2f' % v[:, 18].mean())
    print('levels climbed per push %.2f, heap size at push (mean) %.0f' % (v[:, 19].sum() / max(1, v[:, 18].sum()), v[:, 20].sum() / max(1, v[:, 18].sum())))
    for k, nm in names.items():
        print('%-36s mean %8.0f  p50 %8.0f  p99 %8.0f  max %8.0f cycles' % (nm, v[:, k].mean(), np.percentile(v[:, k], 50), np.percentile(v[:, k], 99), v[:, k].max()))
    t0 = v[:, 15]; t1 = t0 + v[:, 9]
    print('launch span (first start to last end) %d cycles; starts spread over %d cycles' % (t1.max() - t0.min(), t0.max() - t0.min()))
    w = v[:, 11]
    print('wall clock (100 MHz) span of ends: %d ticks' % (w.max() - w.min()))
    own = v[:, [0, 1, 2, 3, 4, 5, 6, 7]].sum(axis=1)
    print('own work per lane: mean %.0f p99 %.0f max %.0f' % (own.mean(), np.percentile(own, 99), own.max()))
    for npops in range(0, 5):
        m = v[:, 13] == npops
        if m.sum():
            print('lanes with %d pops: %6d  own work mean %.0f  pop time mean %.0f' % (npops, m.sum(), own[m].mean(), v[m, 3].mean()))
