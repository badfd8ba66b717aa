"""Host micro-benchmarks of the library (no GPU needed): Keccak-f, point arithmetic, fixed-base multiplication, scalar inversion, transcript prefix, the AVX-512 chains."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
from rofl_project_code_amd import api
L=api.lib()
ns=ctypes.c_double()
names={0:"keccak",1:"dbl",2:"add",3:"encode+add",4:"fixed mul",5:"sc inverse",6:"prefix/commit"}
for what,it in ((0,100000),(1,200000),(2,200000),(3,20000),(4,5000),(5,5000),(6,8192)):
    r=[]
    for _ in range(7):
        L.rofl_dbg_host_bench(what,it,ctypes.byref(ns)); r.append(ns.value)
    print("%-14s min %.1f ns" % (names[what], min(r)))
a=ctypes.c_double(); b=ctypes.c_double()
for W,c,l in ((16,16,8),(26,10,8)):
    r=[];q=[]
    for _ in range(7):
        L.rofl_dbg_host_horner8_selftest(W,c,l,ctypes.byref(a),ctypes.byref(b)); r.append(a.value); q.append(b.value)
    print("horner8 W=%d c=%d: simd %.1f us scalar %.1f us" % (W,c,min(r),min(q)))
r=[];q=[]
for _ in range(7):
    L.rofl_dbg_host_encode8_selftest(16,ctypes.byref(a),ctypes.byref(b)); r.append(a.value); q.append(b.value)
print("encode8 x16: simd %.1f us scalar %.1f us" % (min(r),min(q)))
