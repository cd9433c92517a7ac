"""bench.py -- conv layers pruned per second (BASELINE.json metric) on MI355X.

Workloads, all built from SURVEY.md section 8d's synthetic generator (float32 X and W2 -- exactly what the
reference's float64 arrays hold -- float64 Y), operands RESIDENT in HBM before the timed region, results (mask,
weights, bias) back on the host inside it:

--workload vgg16 (default; north_star's job, BASELINE.json configs[2] on one node)
    ONE instance of the whole-network job: the 12 conv -> conv pairs of VGG-16 with the reference's 3C-4x kept-channel
    count d_c = max(int(c / 1.15), rank) (/root/reference/lib/net.py:1309-1327, 1346-1349), N = 5000 samples per
    layer.  One job = 12 dictionary() problems (c = 64, 64, 128, 128, 256 x3, 512 x5).  With --gpus N the layers are
    sharded over the ranks (cpmi355.shard.prune_sharded: LPT assignment, ONE uint8 all_gather of the channel masks and
    ONE all_gather of the owners' packed (W, b) on RCCL) -- STRONG scaling: the same 12 layers whatever N is.  On a rank
    all its layers are in flight together (cpmi355.shard.ResidentLayerSet: a HIP stream + host thread per chunk of
    equal-width layers, the alpha searches of a chunk as the workgroups of one launch).
    A "step" is `jobs_per_step` back-to-back jobs (chosen during warm-up so that the timed region is >= 2 s);
    value = 12 * jobs / elapsed = layers/s of a single job instance, job_ms = its wall-clock.

--workload resnet50 (BASELINE.json configs[3]): the 40 selections of the released ResNet-50 2x model
    (temp/resnet-50-cp.prototxt: channel samplers in front of branch2a with c up to 2048, 3x3 and residual-aware 1x1
    consumers; cpmi355/jobs.py), N = 5000; same machinery, same JSON line.
--workload vgg16_5x (BASELINE.json configs[4]): the 10 pruned pairs of the released VGG-16 5x model
    (temp/channel_pruning.prototxt) at N = 20000 samples per layer.

--gpus N > 1 without a torch.distributed environment: bench.py launches itself under torch.distributed.run with N
ranks (one per GPU, backend "nccl" = RCCL), so `python bench.py --gpus 8` alone is a valid command.

--workload block (BASELINE.json configs[1]: the VGG-16 conv3_x block, rank = c/2)
    single_instance: the three layers of ONE block instance side by side, nothing else on the chip;
    value: replica throughput -- many independent copies of the block in flight (--inflight groups x --batch copies),
    the regime of a job with hundreds of equal layers; the line says how many really ran.

Prints ONE JSON line (rank 0), shorter than 4 KB: the contract keys + the round's headline figures (benchkit/line.py).
Everything else a leg measures (per-layer tables, stage brackets, chunk reports, notes) goes to bench_detail.json next to this
file (--detail PATH), never to stdout.  The default vgg16 run at N = 1 also carries, as short legs after the headline job,
the resnet50 and vgg16_5x jobs (`other_workloads`) and one R3 pass (`r3`) -- --no-extras skips them.
`roofline` = the MFMA kernel with the largest sum of launch time per job (k_chol_step, the factorisation chain; the refit
Gram GEMM under `gram`): algorithmic flops / the HIP-event time of its launches recorded inside libcpmi355 on their launch
streams during the timed jobs.  `cpu_baseline` = the CPU port of the reference path (oracle/cp_oracle.py driving scikit-learn's
own Lasso / LinearRegression: the arithmetic the reference runs) on a bounded sample of the same layers (the four with
c <= 128 and one with c = 256: about 20 s) on this box's host cores, at the BLAS thread count that is fastest on the
box (swept over 1 / 8 / 32 / all on the three smallest layers first; the CD itself is single-threaded); --cpu-full:
every layer of the job.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchkit import common as _common      # noqa: E402  (sets GPU_MAX_HW_QUEUES before any HIP initialisation, sys.path)
from benchkit.common import Env, cpjobs    # noqa: E402,F401  (tests use bench.Env / bench.cpjobs)
from benchkit.line import render           # noqa: E402


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` with no torch.distributed environment: run this same command under
    torch.distributed.run, one rank per GPU on this node (rendezvous on 127.0.0.1)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("vgg16", "resnet50", "vgg16_5x", "block", "r3"),
                    default=os.environ.get("CP_BENCH_WORKLOAD", "vgg16"))
    ap.add_argument("--scaling", choices=("weak", "strong"), default=os.environ.get("CP_BENCH_SCALING", "strong"),
                    help="N > 1: strong (default) = ONE job instance, its layers sharded over the GPUs (BASELINE configs[2]); the line "
                         "also carries replica_throughput (one instance per GPU) and strong_scaling_bound; weak = replicas as `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="time the CPU port on every layer of the job (the default for vgg16: ~45 s)")
    ap.add_argument("--cpu-sample", action="store_true", help="vgg16: CPU port on the five cheapest layers only (~5 s)")
    ap.add_argument("--sequential-alpha", action="store_true",
                    help="vgg16 on one GPU, the reference's own order: layer after layer, every search starting from the alpha "
                         "the previous layer ended with (cfgs.alpha carry, /root/reference/lib/decompose.py:491, 626-627)")
    ap.add_argument("--no-block", action="store_true", help="vgg16: skip the conv3_x single-instance figures")
    ap.add_argument("--no-extras", action="store_true",
                    help="vgg16, N = 1: skip the short resnet50 / vgg16_5x / R3 legs (other_workloads, r3)")
    ap.add_argument("--precompute-heaviest", type=int, default=None,
                    help="layers whose full normal equations are computed under their alpha search (default: the library's 2)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-jobs-in-flight leg (N = 1)")
    ap.add_argument("--no-form-ab", action="store_true",
                    help="skip the A/B of the factorisation's two forms (launch per step / persistent) on the resident job (N = 1)")
    ap.add_argument("--no-row-assist", action="store_true",
                    help="N > 1, strong: never split a layer's refit rows over its owner and a helper rank (shard.plan_assists)")
    ap.add_argument("--no-exchange-rounds", action="store_true",
                    help="N > 1, strong: ONE exchange after all layers instead of the light layers' results travelling early")
    ap.add_argument("--exchange", choices=("gather", "allgather", "masks"), default=os.environ.get("CP_BENCH_EXCHANGE", "gather"),
                    help="N > 1, strong: what travels after the masks' all_gather -- gather (default, north_star's split): every "
                         "owner's packed (W, b) to rank 0 only; allgather: to every rank; masks: nothing (the weights stay with "
                         "their owner)")
    ap.add_argument("--no-gather", action="store_true", help="skip the sampled-point im2col (extract_XY) measurement")
    ap.add_argument("--no-pcie-f64", action="store_true", help="skip the float64-X variant of the PCIe-inclusive pass")
    ap.add_argument("--profile-mode", action="store_true",
                    help="only whole jobs (1 + warmup + steps x jobs_per_step of them), nothing else on the GPU: for rocprofv3")
    ap.add_argument("--per-stream", type=int, default=int(os.environ.get("CP_BENCH_PER_STREAM", "0")),
                    help="equal-width layers per stream / cp_prune_layers call (0: 1, resnet50: 2)")
    ap.add_argument("--jobs-per-step", type=int, default=0, help="fixed jobs per step (0 = fill >= 2 s)")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("CP_BENCH_INFLIGHT", "6")),
                    help="block: worker groups (3 HIP streams each) running passes over the block concurrently")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("CP_BENCH_BATCH", "8")),
                    help="block: block copies a worker group prunes per cp_prune_layers call")
    ap.add_argument("--detail", default=os.environ.get("CP_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")),
                    help="where everything that is not on the line goes (JSON; '' = nowhere)")
    return ap.parse_args()


def vgg16_extras(args, env, out):
    """the legs that ride along with the default line at N = 1 (outside the timed region of `value`)"""
    if args.workload == "vgg16" and not args.no_block:
        from benchkit.block import block_single_instance, close_workers
        single, group = block_single_instance(env.local_rank)
        close_workers(group)
        out["conv3_block_single_instance"] = single
        # north_star's headline shape (BASELINE.json configs[1]: the conv3_x block, rank = c / 2, 5000 samples, one
        # MI355X) as a first-class number next to `value` (configs[2] on one GPU): layers/s of ONE instance of the
        # block, its three layers side by side, masks checked against the reference goldens L01..L03
        out["value_conv3_block"] = {"value": single["layers_per_s"], "unit": "layers/s", "ms_per_pass": single["ms_per_pass"],
                                    "mask_parity_vs_reference_golden": single["mask_parity_vs_reference_golden"],
                                    "workload": "conv3_x block (conv2_2->conv3_1 c=128, conv3_1->conv3_2 and "
                                                "conv3_2->conv3_3 c=256; n=256, k=3, rank=c/2), N=5000, one instance"}
    if not args.no_gather:
        from benchkit.gather import bench_patch_gather
        out["patch_gather"] = bench_patch_gather(env.local_rank)
    if args.workload == "vgg16" and not args.no_extras:
        # (Behind the other legs these three used to run 20-45 % slower than in a process of their own: the process had had more
        #  than ~16 hardware queues alive, and the runtime never gives one back.  GPU_MAX_HW_QUEUES is 16 now -- cpmi355/capi.py.)
        from benchkit.extras import short_job
        from benchkit.r3 import r3_short_pass
        out["other_workloads"] = {job: short_job(env.local_rank, job) for job in ("resnet50", "vgg16_5x")}
        out["r3"] = r3_short_pass(env.local_rank)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    env = Env()
    if env.world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the torch.distributed environment has WORLD_SIZE=%d" % (args.gpus, env.world))
    if args.sequential_alpha:
        if env.world != 1:
            raise SystemExit("bench.py: --sequential-alpha is a one-GPU mode (every layer needs the previous layer's alpha)")
        from benchkit.sequential import bench_sequential_alpha
        out = bench_sequential_alpha(args, env)
    elif args.workload == "r3":
        if env.world != 1:
            raise SystemExit("bench.py: --workload r3 is a one-GPU workload (R3 is a sequential loop over the convs)")
        from benchkit.r3 import bench_r3
        out = bench_r3(args, env)
    elif args.workload != "block":
        from benchkit.job import bench_job
        out = bench_job(args, env, args.workload)
        if out is not None and not args.profile_mode and env.world == 1:
            vgg16_extras(args, env, out)
    else:
        from benchkit.block import bench_block
        out = bench_block(args, env)
    if env.rank == 0 and out is not None:
        if args.detail:
            out["detail_file"] = os.path.relpath(args.detail, ROOT) if args.detail.startswith(ROOT) else args.detail
            try:
                with open(args.detail, "w") as fh:
                    json.dump(out, fh, indent=1)
            except OSError as e:
                print("bench.py: could not write %s: %s" % (args.detail, e), file=sys.stderr)
        print(render(out), flush=True)
    env.close()
    sys.stdout.flush()
    sys.stderr.flush()


if __name__ == "__main__":
    main()
