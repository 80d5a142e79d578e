import ctypes as C, os, sys, json, subprocess
os.environ['TSG_LIB_OVERRIDE'] = os.path.abspath('ablate/libtsg_stamp.so')
sys.path.insert(0, os.getcwd())
sys.argv = ['bench.py', '--ods', '16384', '--warmup', '0', '--cpu-seconds', '0', '--profile-every', '0']
import runpy
try:
    runpy.run_path('bench.py', run_name='__main__')
finally:
    from ts_trajgen_amd import _lib
    L = C.CDLL(os.environ['TSG_LIB_OVERRIDE'])
    out = (C.c_ulonglong * 16)()
    L.tsg_debug_read_stamps(out, 0)
    v = list(out)
    print('threads', v[7], 'pops', v[8])
    n=max(1,v[7])
    names={0:'relax',6:'loop-top',1:'pop',3:'postpop',4:'futile-check(not futile)',9:'futile-path',5:'emit',10:'search-over-pre',11:'finish+refill',2:'tail'}
    print('threads',v[7],'pops',v[8]); print({names[k]: round(v[k]/n/1e3,1) for k in names})
