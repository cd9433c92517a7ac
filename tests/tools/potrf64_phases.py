"""Phase counters of the task-queue Cholesky (k_potrf64).  Needs the timers compiled in:
   touch channel-pruning_amd/csrc/refit.hip; make -C channel-pruning_amd/csrc FLAGS_refit=-DCP_POTRF_TIMERS=1"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, cpmi355, cp_oracle
ctx = cpmi355.Context(0)
ctx.lib.cp_debug_potrf_cycles.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
ctx.lib.cp_debug_potrf_reset.argtypes = [ctypes.c_void_p]
for (c, kept) in ((256, 131), (512, 460)):
    X, W2, Y, B2 = cp_oracle.synth_layer(40, 5000, c, 64, 3)
    prob = cpmi355.LayerProblem(ctx, X, W2, Y)
    mask = np.zeros(c, bool); mask[:kept] = True
    prob.refit(mask)
    ctx.enable_stage_timing(1)
    ctx.lib.cp_debug_potrf_reset(ctx.h)
    prob.refit(mask)
    st = dict(ctx.last_stage_times())
    out = (ctypes.c_ulonglong * 8)()
    ctx.lib.cp_debug_potrf_cycles(ctx.h, out)
    nd, npan = max(1, out[6]), max(1, out[7])
    print("p = %d: refit_cholesky %.3f ms, %d diagonal + %d panel tasks" % (kept * 9, st["refit_cholesky"], out[6], out[7]))
    for i, nm in enumerate(["diag: wait + accumulate", "diag: factor", "diag: invert + write + flag"]):
        print("   %-28s %9.0f ticks / task" % (nm, out[i] / nd))
    for i, nm in zip((3, 4, 5), ["panel: wait + accumulate", "panel: wait for diagonal", "panel: multiply + store + flag"]):
        print("   %-28s %9.0f ticks / task" % (nm, out[i] / npan))
    ctx.enable_stage_timing(0)
    prob.free()
