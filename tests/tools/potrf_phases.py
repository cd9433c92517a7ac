"""Phase cycle counters of the Cholesky diagonal-block kernel.  Needs a library built with the timers compiled in:
   touch channel-pruning_amd/csrc/refit.hip; make -C channel-pruning_amd/csrc FLAGS_refit=-DCP_POTRF_TIMERS=1   (off by default)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, cpmi355, cp_oracle
ctx = cpmi355.Context(0)
X, W2, Y, B2 = cp_oracle.synth_layer(40, 5000, 256, 256, 3)
prob = cpmi355.LayerProblem(ctx, X, W2, Y)
mask = np.zeros(256, bool); mask[:131] = True
prob.refit(mask); prob.refit(mask)
out = (ctypes.c_ulonglong * 8)()
ctx.lib.cp_debug_potrf_cycles.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
ctx.lib.cp_debug_potrf_cycles(ctx.h, out)
names = ["load", "step1 diag16 (wave0)", "step2 subst", "step3 mfma update", "T blocks", "V assembly", "outputs"]
tot = sum(out[i] for i in range(7))
for i in range(7): print("%-24s %8d cycles %5.1f%%" % (names[i], out[i], 100.0 * out[i] / tot))
print("total", tot, "cycles =", tot / 2.4e3, "us @2.4GHz")
