"""benchkit.common -- what every leg of bench.py shares: paths, constants, the synthetic operands of the conv3_x block,
the golden check, the algorithmic flop / byte counts of SURVEY.md section 8d, the torch.distributed environment."""
import os
import sys

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before any HIP initialisation; why 16: cpmi355/capi.py::load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "channel-pruning_amd") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))

N_SAMPLES, KSIZE = 5000, 3
# 0 = sklearn's own operation order in the CD step (the drop-in's default, lib/cfgs.py); 3 = CP_CD_RECIPROCAL | CP_CD_DELTA
# (rounding-level differences, same masks on every golden, ~4 % faster job)
CD_FLAGS = int(os.environ.get("CP_BENCH_CD_FLAGS", "0"))
F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X public FP64 matrix figure (the guide lists no f64 row); the
                              # measured v_mfma_f64_16x16x4_f64 issue rate is reported next to it
MIN_TIMED_SECONDS = 2.0

# ---- workload tables ------------------------------------------------------------------------------------------
BLOCK_LAYERS = [  # (layer_id, c, n, rank)  -- ids match tests/golden/L0[123]_*.npz
    (31, 128, 256, 64),
    (32, 256, 256, 128),
    (33, 256, 256, 128),
]
BLOCK_GOLDEN = {31: "L01_conv2_2_conv3_1", 32: "L02_conv3_1_conv3_2", 33: "L03_conv3_2_conv3_3"}

from cpmi355 import jobs as cpjobs   # noqa: E402  (workload tables + the synthetic generator; no device code)


def synth(layer_id, c, n):
    """SURVEY.md section 8d generator for the conv3_x block layers (k = 3, N = 5000, ReLU'd X)"""
    return cpjobs.synth(dict(layer_id=layer_id, N=N_SAMPLES, c=c, n=n, k=KSIZE))


def sketch_matrix(p):
    """the seeded test matrix of the sketched weight goldens (oracle/cp_oracle.py::sketch_matrix, restated)"""
    return np.random.RandomState(777).randn(int(p), 32)


def golden_check(name, idxs, newW2):
    """-> (mask identical, weight rel. Frobenius error [estimated from the sketch when the golden holds no full tensor])"""
    gpath = os.path.join(ROOT, "tests", "golden", name + ".npz")
    if not os.path.isfile(gpath):
        return None, None
    g = np.load(gpath)
    same = bool(np.array_equal(idxs, g["idxs"]))
    if not same:
        return False, None
    if newW2 is None:             # the weights of this layer did not travel to this rank (exchange "masks"): the mask alone
        return True, None
    wm = newW2.reshape(newW2.shape[0], -1)
    if "newW2_sketch" in g.files:
        sk = wm @ sketch_matrix(wm.shape[1])
        return True, float(np.linalg.norm(sk - g["newW2_sketch"]) / np.linalg.norm(g["newW2_sketch"]))
    return True, float(np.linalg.norm(newW2 - g["newW2"]) / np.linalg.norm(g["newW2"]))


def layer_flops(c, n, pp, N=N_SAMPLES, kk=KSIZE * KSIZE):
    """(SURVEY.md 8d algorithmic flops of one dictionary() call [full-matrix counts], flops the launches execute
    [symmetric halves, 128-padded tiles])"""
    S = min(400, N // 20)
    alg = (2.0 * c * S * kk * n + 2.0 * S * n * c * c + 2.0 * S * n * c + 2.0 * N * pp * pp + 2.0 * N * pp * n
           + pp ** 3 / 3.0 + 2.0 * pp * pp * n)
    pad = lambda v, a: (v + a - 1) // a * a   # noqa: E731
    ck, P, n_pad, Np = pad(c * kk, 128), pad(pp, 128), pad(n, 128), pad(N, 16)
    tri = lambda m: m // 128 * (m // 128 + 1) // 2 * 128.0 * 128.0   # noqa: E731
    exe = (tri(ck) * 2.0 * (pad(S, 16) + n) + 2.0 * S * n * ck + tri(P) * 2.0 * Np + 2.0 * P * n_pad * Np
           + P ** 3 / 3.0 + 2.0 * P * P * n_pad)
    return alg, exe


def algorithmic_bytes(c, n, pp, N=N_SAMPLES, kk=KSIZE * KSIZE):
    """inputs once at f32 (+ the f64 Y the caller hands over) and the outputs at f64"""
    return 4.0 * (N * c * kk + N * n + n * c * kk) + 4.0 * N * n + 8.0 * (n * pp + n) + c


def host_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([i.get("num_threads", 1) for i in threadpool_info()] + [1]))
    except Exception:
        return os.cpu_count() or 1


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ==================================================================================================================
# distributed plumbing
# ==================================================================================================================
class Env:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = os.environ.get("CP_BENCH_DIST_BACKEND", "nccl")   # "gloo": several ranks on ONE GPU (flow test)
        self.dist = None
        self.torch = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                self.local_rank = self.local_rank % max(1, torch.cuda.device_count())
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group(self.backend)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if self.dist is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def bcast_int(self, value):
        if self.dist is None:
            return int(value)
        t = self.torch.tensor([int(value)], dtype=self.torch.int64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.broadcast(t, src=0)
        return int(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def git_head():
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None



# ==================================================================================================================
# workload: vgg16 (the north_star job)
# ==================================================================================================================
JOB_TEXT = {     # (config.workload -- at most 230 characters, it travels on the line --, metric)
    "vgg16": ("vgg16: ONE instance of the whole-network job = the 12 conv->conv pairs of VGG-16, kept channels d_c = int(c/1.15) "
              "(the reference's 3C-4x table), N=5000, k=3; 1 job = 12 dictionary() calls; 1 step = jobs_per_step jobs",
              "conv layers pruned/sec (VGG-16 4x, 5k samples)"),
    "resnet50": ("resnet50: ONE instance of the ResNet-50 2x job = the 40 selections of the released model (16 samplers c=64..2048 "
                 "1x1, 8 branch2a->2b 3x3, 16 branch2b->2c 1x1 residual-aware), N=5000; 1 step = jobs_per_step jobs",
                 "conv layers pruned/sec (ResNet-50 2x, 5k samples)"),
    "vgg16_5x": ("vgg16_5x: ONE instance of the VGG-16 5x job = the 10 pruned conv->conv pairs of the released model (kept "
                 "24,22,41,51,108,89,111,184,276,228), N=20000, k=3; 1 step = jobs_per_step jobs",
                 "conv layers pruned/sec (VGG-16 5x, 20k samples)"),
}


# measured ms of one layer alone by channel count (profiles/r02_*): the LPT costs of the vgg16 job
VGG16_COST_MS = {64: 1.5, 128: 3.0, 256: 6.8, 512: 15.5}
PROFILE_TAG = "r06"      # profiles/<tag>_pmc_{fetch,write}_size_kb.md feed roofline.traffic

