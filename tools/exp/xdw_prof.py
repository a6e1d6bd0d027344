"""debug: per-phase wave-cycles of k_xdrop_wave (librsk_xdwprof.so built with -DXDW_PROF) on the 256 x 30,000 -db search"""
import ctypes, os, subprocess, sys
os.environ["RSK_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "reseek_amd", "librsk_xdwprof.so")
os.environ["RSK_MKF_OVERLAP"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.argv = ["bench_search.py", "qdb", "256", "30000", "sensitive"]
import runpy
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench_search.py"), run_name="__main__")
except SystemExit:
    pass
import reseek_amd.capi as capi
out = (ctypes.c_ulonglong * 12)()
print("rc", capi.lib().rsk_debug_xdw_prof(out))
v = list(out)
names = ["body cycles", "tail cycles", "traceback cycles", "-", "rows", "chunks", "tail cells", "tb steps", "extensions with a path"]
for n, x in zip(names, v): print("%-24s %d" % (n, x))
if v[4]:
    print("per row: body %.0f cyc, tail %.0f cyc; chunks/row %.2f, tail cells/row %.2f; traceback %.0f cyc/step, %.0f steps/ext" % (
        v[0] / v[4], v[1] / v[4], v[5] / v[4], v[6] / v[4], v[2] / max(v[7], 1), v[7] / max(v[8], 1)))
