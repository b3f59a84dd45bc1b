"""Shader-clock stamps inside k_cfm_attn_bwd (build with -DBWD_TIMING): third window of (head 0, group 0), all ten waves, on wave 0's
clock (cycles from wave 0's top of the window): 0 top, 1 DMA wait done, 2 barrier B1 passed, 3 row DMAs issued, 4 / 5 first / second key
tile done, 6 row wait done, 7 barrier B2 passed, 8 query phase / conversion done, 9 top of the next window.
usage: python scripts/r04_ks_timing.py build/libcffm_kst.so"""
import ctypes as C, os, sys, runpy
so = os.path.abspath(sys.argv[1])
sys.argv = [sys.argv[0], so]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'r04_attn_bwd_bench.py'), run_name='__main__')
raw = C.CDLL(so)
buf = (C.c_longlong * 256)()
assert raw.cffm_debug_bwd_stamps(buf) == 0
t0 = buf[0]
for k in range(12):
    print('wave %d' % k, ' '.join('%6d' % (buf[k * 16 + i] - t0) for i in range(10)))
