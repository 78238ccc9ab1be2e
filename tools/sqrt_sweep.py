import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiosignalprocess_amd.ns import load_library
lib = load_library()
lib.AspNs_debug_compare.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_float, C.c_int]
n = C.c_uint32(); bad = (C.c_uint32 * 64)()
last_bad = -1; first_bad = None; tot = 0
step = 1 << 22
for start in range(0, 0x7f800001, step):
    cnt = min(step, 0x7f800001 - start)
    lib.AspNs_debug_compare(10, 9, start, cnt, C.byref(n), bad, 1.0, 0)
    if n.value:
        tot += n.value
        if first_bad is None: first_bad = start
        last_bad = start + cnt
        if start >= 0x01000000: print(hex(start), n.value, [hex(b) for b in list(bad)[:3]])
print("total", tot, "first", hex(first_bad), "last chunk end", hex(last_bad))
