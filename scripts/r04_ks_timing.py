"""Shader-clock stamps inside k_cfm_attn_bwd_ks (build with -DBWD_TIMING): third window of (head 0, group 0) wave 0 / wave 5 and of
(head 0, group 13) wave 9: cycles from the top of the window to: 1 DMA wait done, 2 barrier B1 passed, 3 DMAs issued, 4 / 5 first /
second key tile done, 6 row wait done, 7 barrier B2 passed, 8 query phase / conversion done, 9 top of the next window.
usage: python scripts/r04_ks_timing.py build/libcffm_kst.so"""
import ctypes as C, os, sys, runpy
so = os.path.abspath(sys.argv[1])
sys.argv = [sys.argv[0], so]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'r04_attn_bwd_bench.py'), run_name='__main__')
raw = C.CDLL(so)
buf = (C.c_longlong * 48)()
assert raw.cffm_debug_bwd_stamps(buf) == 0
for k, name in enumerate(('g0 wave0', 'g0 wave5', 'g13 wave9')):
    r = [buf[k * 16 + i] for i in range(10)]
    print('%-10s' % name, ' '.join('%6d' % (v - r[0]) for v in r))
