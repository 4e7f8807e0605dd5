#!/usr/bin/env python3
"""bench.py -- images/sec of the per-image forward hot path (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this process starts N rank processes itself (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in their environment) BEFORE it touches HIP -- the parent never initialises the GPU and never exec()s -- relays
rank 0's single JSON line and exits non-zero if any rank fails.  Under torchrun
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) the ranks are already there and nothing is spawned.
`HPE_FORCE_DIST=1 python bench.py --gpus 1` takes the same spawned path with a world of one (rehearses the RCCL calls).

One "step" = one pass of the whole path over one batch of 256 synthetic 224x224x3 images per GPU, already
resident in HBM: ResNet-50 v1 encoder (fp32 MFMA) -> 3 IEF regressor stages -> SMPL (LBS, joint regress,
orthographic reprojection) at ALL three stages (what Trainer.val_step evaluates; nothing is skipped) ->
for N > 1 one RCCL all-gather of the final theta [256,85] per rank over xGMI.  Images shard by batch
(independent units, no data-path collective besides that gather): weak scaling.  --config5 adds the two
reprojection losses of every stage (one library call) and, for N > 1, ONE all-reduce of the [3,4] block of
(kp numerator, kp count, -, mesh sum).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel family, the 53 convolution layers (implicit-GEMM
conv_gemm_f32_dma_kernel; the 3x3 layers run as fp32 Winograd F(2x2,3x3), which does 2.25x fewer multiplies for
the same layer -- the algorithmic FLOPs priced here are the direct convolution's): achieved =
7.7119 GFLOP/img * 256 img / (encoder span of the last timed step, HIP events recorded on the launch stream; the
batch-chunk streams overlap their conv launches, so the span -- not a sum of overlapping durations -- is the family's
time), peak = 157.3 TFLOP/s fp32 MFMA.  `roofline.serial` is the same quantity with the chunk streams off and events
around each launch (one extra step after the timed region); it is the number the rocprofv3 kernel stats in profiles/
add up to.  `sustained` repeats the step untimed for >= 10 s and reports its rate with sampled board power / clock.
`cpu_baseline` is the CPU oracle (a NumPy / torch-CPU restatement of the reference path -- TensorFlow is not
installable here) timed on this host's cores with BASELINE.md §3's protocol (B = 1 and B = 64,
2 warm-ups, median of 5), rank 0, N == 1 only.  The run FAILS (exit 3) if the parity block exceeds the 1e-4 bar.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENCODER_GFLOP_PER_IMG = 7.711850496  # 2 * 3,855,925,248 MAC (resnet_spec.encoder_macs_per_image)
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (same guide)
PEAK_HBM_GBS = 8000.0
PARITY_BAR = 1e-4


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (BASELINE metric: 256)")
    ap.add_argument("--cpu-sample", type=int, default=64, help="batch of the cpu_baseline's large case (BASELINE.md §3: 64; 0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--sustain", type=float, default=10.0, help="seconds of the untimed steady-state loop after the timed steps (0 = skip)")
    ap.add_argument("--encoder-dtype", default="fp32", choices=["fp32", "bf16"], help="bf16 = BASELINE configs[3] (bf16 encoder, fp32 SMPL)")
    ap.add_argument("--config5", action="store_true", help="also evaluate kp + mesh reprojection losses of every stage (BASELINE configs[4])")
    ap.add_argument("--no-pipeline", action="store_true", help="serial steps: the regressor + SMPL tail of batch k does NOT overlap the encoder of batch k+1")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------- rank launcher
def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """Start n fresh rank processes of this script (the parent has made no HIP call and makes none), relay rank 0's JSON
    line, exit with the first failing rank's code."""
    import subprocess

    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    live = set(range(n))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print("bench.py: rank %d exited with code %d; stopping the other ranks" % (r, code), file=sys.stderr)
                for q in live:
                    procs[q].terminate()  # exact PIDs this process started
        time.sleep(0.05)
    reader.join(timeout=10)
    text = (out0[0] if out0 else b"").decode(errors="replace")
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    if rc == 0 and len(lines) != 1:
        print("bench.py: expected one JSON line from rank 0, got %d" % len(lines), file=sys.stderr)
        rc = 1
    for ln in lines[:1]:
        print(ln, flush=True)
    sys.exit(rc)


# ------------------------------------------------------------------------------------------------- power / clock sampling
class PowerSampler(object):
    """Samples board power (W) and shader clock (MHz) of the busiest GPU from sysfs hwmon in a side thread
    (rocm-smi as the fallback); no GPU call is made from the thread."""

    def __init__(self, period=0.5):
        import glob
        import threading

        self.period = period
        self.samples = []
        self._stop = False
        self._hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self._t = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _sysfs(self):
        best = None
        for h in self._hw:
            p = self._read(os.path.join(h, "power1_average"))
            if p is None:
                p = self._read(os.path.join(h, "power1_input"))
            if p is None:
                continue
            f = self._read(os.path.join(h, "freq1_input"))
            cand = (p / 1e6, None if f is None else f / 1e6)
            if best is None or cand[0] > best[0]:
                best = cand
        return best

    @staticmethod
    def _smi():
        import re
        import subprocess

        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        except Exception:
            return None
        pw = [float(m.group(1)) for m in re.finditer(r"Package Power[^:]*:\s*([0-9.]+)", o)]
        ck = [float(m.group(1)) for m in re.finditer(r"sclk clock level:[^(]*\(([0-9.]+)Mhz\)", o)]
        if not pw:
            return None
        i = max(range(len(pw)), key=lambda k: pw[k])
        return (pw[i], ck[i] if i < len(ck) else None)

    def _run(self):
        use_smi = self._sysfs() is None
        while not self._stop:
            s = self._smi() if use_smi else self._sysfs()
            if s is not None:
                self.samples.append((time.perf_counter(),) + s)
            time.sleep(self.period if not use_smi else max(self.period, 1.0))

    def start(self):
        self._t.start()

    def stop(self):
        self._stop = True
        self._t.join(timeout=15)

    def summary(self, skip_s=2.0):
        if not self.samples:
            return None, None, 0
        t0 = self.samples[0][0]
        use = [s for s in self.samples if s[0] - t0 >= skip_s] or self.samples
        pw = [s[1] for s in use]
        ck = [s[2] for s in use if s[2] is not None]
        return (round(sum(pw) / len(pw), 1), round(sum(ck) / len(ck), 1) if ck else None, len(use))


# ------------------------------------------------------------------------------------------------- the benchmark (one rank)
def main():
    args = parse_args()
    force_dist = bool(os.environ.get("HPE_FORCE_DIST"))
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist):
        launch_ranks(args.gpus)  # does not return

    # The encoder runs its batch chunks on 2 HIP streams, the regressor + SMPL tail on a third and RCCL adds its own; the HIP runtime
    # multiplexes all streams of a process onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and once two busy streams share a
    # queue their overlap is lost (round 1: 20.1 -> 18.4 ms/step with a process group alive).  Must be set before HIP initialises.
    try:
        if int(os.environ.get("GPU_MAX_HW_QUEUES", "0")) < 8:
            os.environ["GPU_MAX_HW_QUEUES"] = "8"
    except ValueError:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"

    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)  # see the print at the end

    import numpy as np
    import torch

    import hpe_amd
    from hpe_amd import distributed as D
    from hpe_amd import synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or force_dist  # HPE_FORCE_DIST: rehearse the RCCL calls at world 1
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    B = args.batch
    smpl = synthetic.make_smpl_model()
    enc = synthetic.make_encoder_params()
    reg = synthetic.make_regressor_params()
    mean_vals = synthetic.make_mean_params()

    class Cfg(object):
        img_size, num_stage, batch_size, data_format = 224, 3, B, "NHWC"
        checkpoint_dir = smpl_model_path = None
        encoder_dtype = args.encoder_dtype

    pred = hpe_amd.Predictor(Cfg(), smpl_model=smpl, mean_params=mean_vals, encoder_params=enc, regressor_params=reg,
                             device=local_rank)
    eng = pred.engine
    images = torch.from_numpy(synthetic.make_images(B, seed=1000 + rank)).cuda()
    want = eng.DEFAULT_OUTPUTS + (("verts2d",) if args.config5 else ())
    # two output sets used alternately: the all-gather of step k reads theta_k while step k+1 already writes theta_{k+1}
    # Steady-state serving is software-pipelined across batches: the encoder of batch k+1 (caller's stream) overlaps the
    # latency-bound regressor + SMPL tail of batch k (the ctx's tail stream).  Every consumer of a batch's outputs -- the loss
    # kernels, the collectives -- is enqueued on the tail stream, so nothing of a step is skipped or left outside the timed
    # region: the fence at its end waits for the last tail and the last gather.
    pipe = not args.no_pipeline
    plans = [eng.make_forward_plan(B, all_stages=True, want=want, pipelined=pipe) for _ in range(2)]
    run, outs = plans[0]
    tail = eng.tail_stream() if pipe else None
    pipe_on = [pipe]  # cleared for the per-launch-timed step (hpe_forward_pipelined then runs serially on the caller's stream)

    import contextlib

    def on_tail():
        return torch.cuda.stream(tail) if (tail is not None and pipe_on[0]) else contextlib.nullcontext()

    losses = {}
    if args.config5:
        seg_np, kp_np = synthetic.make_lsp_targets(B, seed=2000 + rank)
        seg_gts = torch.from_numpy(seg_np[..., 0].copy()).cuda()
        kp_gts = torch.from_numpy(kp_np).cuda()
        loss_out = [torch.zeros((3, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    theta_all = [torch.empty((world * B, 85), dtype=torch.float32, device="cuda") for _ in range(2)] if use_dist else None
    pending = [None, None]
    step_no = [0]

    def step():
        k = step_no[0] & 1
        step_no[0] += 1
        if use_dist and pending[k] is not None:
            with on_tail():
                pending[k].wait()  # the gather that last used this output set (two steps ago), before the tail overwrites it
            pending[k] = None
        o = plans[k][0](images)
        with on_tail():
            if args.config5:
                # one library call for the 2 x 3 losses, then (N > 1) ONE all-reduce of the [3,4] block
                packed = eng.val_losses(kp_gts, [st["kp2d"] for st in o], seg_gts, [st["verts2d"] for st in o], out=loss_out[k])
                if use_dist:
                    packed = D.reduce_losses(packed)
                losses["packed"] = packed
            if use_dist:
                # the ONE data-path collective: all-gather of the predicted theta over RCCL, asynchronous so that it overlaps
                # the next batch's encoder (it is waited for before its buffers are reused and before the timed region ends)
                pending[k] = dist.all_gather_into_tensor(theta_all[k], o[-1]["theta"], async_op=True)
        return o

    # one-time initialisation that is not a benchmark step: code-object load / first-launch setup of every kernel and the RCCL
    # communicator (both are lazy); the W warm-up steps and the K timed steps follow
    eng.forward(images[:2], all_stages=True)
    if use_dist:
        dist.all_gather_into_tensor(theta_all[0], outs[-1]["theta"])
    torch.cuda.synchronize()

    def fence():
        for k in range(2):
            if use_dist and pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_roofline:
        eng.enable_timing(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    roofline = None
    loss_roofline = None
    phase = None
    if not args.no_roofline:
        # (1) live, over the timed region: HIP events on the launch stream around the encoder (fork/join of the
        #     batch-chunk streams).  The conv launches of the chunks overlap, so the kernel family's time is the
        #     encoder span (it also contains pad / max-pool / avg-pool, ~2 % -> the fraction is conservative).
        tm = eng.timings()  # events of the last timed step
        span_ms = tm["encoder_ms"]
        achieved = ENCODER_GFLOP_PER_IMG * B / span_ms  # GFLOP/ms == TFLOP/s
        if args.config5:
            lt = eng.loss_timings()
            n_sil = int((seg_gts > 0).sum().item())
            pairs = float(n_sil) * 6890.0 * 3.0  # pixel -> vertex candidate pairs of the 3 stages of one step
            # v_mfma_f32_32x32x2_f32 evaluates 1024 (vertex, pixel) pairs in 64 cycles on one of the 1024 SIMDs at 2.4 GHz
            peak_pairs = 1024.0 / 64.0 * 1024 * 2.4e9
            a2b_ms = lt["a2b_search_ms"]
            loss_roofline = {
                "kernel": "pixel -> nearest vertex x 3 stages: nn_a2b_grid_kernel (cell grid over the mesh, candidates as one K=2 fp32 MFMA "
                          "per 32 vertices x 32 pixels) + nn_a2b_mfma_kernel (full search) for images whose mesh is concentrated in < 40 cells",
                "bound": "mfma", "achieved": round(pairs / a2b_ms / 1e9, 3), "peak": round(peak_pairs / 1e12, 3), "unit": "Tpair/s",
                "frac": round(pairs / a2b_ms / 1e9 / (peak_pairs / 1e12), 4), "launch_ms": round(a2b_ms / 3.0, 4),
                "pairs_per_launch": pairs / 3.0, "val_losses_ms_per_step": round(lt["val_losses_ms"], 4),
                "b2a_rows_bytes_per_launch": B * (224 * 4 * 8 * ((6890 + 255) // 256) + 6890 * 8),
                "note": "achieved = (pixel, vertex) pairs of the FULL search / time: pairs the grid search never evaluates count, so this "
                        "is an equivalent rate and can exceed the matrix-core peak; HPE_MESH_A2B=mfma times the full search alone",
            }
        # (2) serial cross-check, extra steps after the timed region: chunk streams off, events around each of the
        #     53 launches; the sum matches the rocprofv3 --kernel-trace --stats durations (profiles/).
        eng.enable_timing(2)
        fence()
        pipe_on[0] = False
        step()
        fence()
        pipe_on[0] = pipe
        ts = eng.timings()
        per_conv = eng.conv_timings()
        serial_tf = ENCODER_GFLOP_PER_IMG * B / ts["conv_ms"]
        PEAK = PEAK_FP32_MFMA_TFLOPS if args.encoder_dtype == "fp32" else PEAK_BF16_MFMA_TFLOPS
        roofline = {
            "bound": "mfma",
            "kernel": eng.encoder_kernel_description() + "; batch chunks on %s concurrent streams" % os.environ.get("HPE_STREAMS", "2"),
            "achieved": round(achieved, 3),
            "peak": PEAK,
            "unit": "TFLOP/s",
            "frac": round(achieved / PEAK, 4),
            "traffic": None,
            "launch_ms": round(span_ms, 4),
            "flop_per_launch": ENCODER_GFLOP_PER_IMG * B * 1e9,
            "serial": {"sum_of_53_launch_ms": round(ts["conv_ms"], 4), "achieved": round(serial_tf, 3),
                       "frac": round(serial_tf / PEAK, 4)},
        }
        if args.encoder_dtype == "bf16":
            # at 16x the fp32 matrix rate the bf16 encoder is HBM bound: price it in algorithmic bytes
            nbytes = hpe_amd.resnet_spec.encoder_min_bytes_per_image(2) * B
            gbs = nbytes / span_ms / 1e6
            roofline.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                             "bytes_per_launch": nbytes, "mfma_tflops": round(achieved, 1)})
            roofline.pop("flop_per_launch")
            roofline["serial"] = {"sum_of_53_launch_ms": round(ts["conv_ms"], 4)}
        # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot run
        # inside the timed region); scaled by batch, null if no measurement of this build + dtype is committed.
        import glob

        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*conv_hbm_traffic_%s.json" % args.encoder_dtype)))
        if not cands and args.encoder_dtype == "fp32":
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*final_conv_hbm_traffic.json")))  # round-1 naming
        if cands:
            tj = json.load(open(cands[-1]))
            roofline["traffic"] = round(tj["total_bytes"] * B / tj["batch"])
            roofline["traffic_source"] = os.path.relpath(cands[-1], ROOT)
        phase = {"encoder_ms": round(tm["encoder_ms"], 3), "regress_smpl_ms": round(tm["regress_smpl_ms"], 3),
                 "step_ms_events": round(tm["total_ms"], 3)}
        eng.enable_timing(0)
        if rank == 0 and os.environ.get("HPE_BENCH_LAYERS"):
            specs = list(hpe_amd.resnet_spec.CONV_SPECS)
            flops = {s.name: 2.0 * s.kh * s.kw * s.cin * s.cout * s.hout * s.hout * B for s in specs}
            for s, ms in zip(specs, per_conv):
                fl, note = flops[s.name], ""
                # conv_block: the projection shortcut runs inside branch2c's launch (dual-source GEMM) and has no launch of its own
                short = s.name.replace("branch2c", "branch1")
                dual = os.environ.get("HPE_DUAL", "1") != "0" and s.name.endswith("a_branch2c") and short in flops
                if dual:
                    fl, note = fl + flops[short], "  (+ %s in the same launch)" % short
                if s.name.endswith("branch1") and os.environ.get("HPE_DUAL", "1") != "0":
                    print("%-18s %8.3f ms        -     (inside %s)" % (s.name, ms, s.name.replace("branch1", "branch2c")), file=sys.stderr)
                    continue
                print("%-18s %8.3f ms %7.1f TF%s" % (s.name, ms, fl / ms / 1e9, note), file=sys.stderr)

    # ---- steady state: the same step, untimed by the driver, for >= args.sustain seconds with power / clock sampled
    sustained = None
    if args.sustain > 0:
        sampler = PowerSampler() if rank == 0 else None
        fence()
        if sampler:
            sampler.start()
        n_sus = 0
        t_s = time.perf_counter()
        while True:
            for _ in range(20):
                step()
            n_sus += 20
            torch.cuda.synchronize()
            flag = torch.tensor([1.0 if time.perf_counter() - t_s >= args.sustain else 0.0], device="cuda")
            if use_dist:
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)  # all ranks leave the loop together
            if float(flag.item()) > 0:
                break
        fence()
        dt_s = time.perf_counter() - t_s
        if sampler:
            sampler.stop()
            pw, ck, ns = sampler.summary()
            sustained = {"ms_per_step": round(dt_s / n_sus * 1e3, 4), "steps": n_sus, "seconds": round(dt_s, 2),
                         "images_per_sec": round(world * B * n_sus / dt_s, 2), "avg_power_w": pw, "sclk_mhz": ck, "samples": ns}
            trace = os.environ.get("HPE_POWER_TRACE")
            if trace:
                with open(trace, "w") as f:
                    f.write("t_s,power_w,sclk_mhz\n")
                    for ts_, p_, c_ in sampler.samples:
                        f.write("%.2f,%.1f,%s\n" % (ts_ - sampler.samples[0][0], p_, "" if c_ is None else "%.0f" % c_))

    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import hmr_oracle as O  # the checker, timed as the CPU baseline

        n = min(args.cpu_sample, B)
        img_np = images[:n].cpu().numpy()
        osmpl = O.SMPL(smpl)
        mean = O.load_mean_param(mean_vals)

        def timed(batch, reps=5, warm=2):
            x = img_np[:batch]
            for _ in range(warm):
                r = O.predict(x, enc, reg, osmpl, mean)
            ts_ = []
            for _ in range(reps):
                tc = time.perf_counter()
                r = O.predict(x, enc, reg, osmpl, mean)
                ts_.append(time.perf_counter() - tc)
            ts_.sort()
            return r, ts_[len(ts_) // 2]

        _, t1 = timed(1)
        ref, tn = timed(n)
        cpu_baseline = {
            "value": round(n / tn, 3),
            "unit": "images/sec",
            "cores": int(torch.get_num_threads()),
            "kind": "port",
            "sample": "BASELINE.md §3 protocol: batch 1 and batch %d of the bench images, full path (ResNet-50 + 3 IEF stages + SMPL x3), "
                      "2 warm-ups then the median of 5 runs each; value = the batch-%d rate.  CPU restatement of the reference "
                      "(NumPy + torch-CPU conv2d), not TensorFlow" % (n, n),
            "batch1": {"images_per_sec": round(1.0 / t1, 3), "median_ms": round(t1 * 1e3, 2)},
            "batch%d" % n: {"images_per_sec": round(n / tn, 3), "median_ms": round(tn * 1e3, 2)},
        }
        last_outs = plans[(step_no[0] - 1) & 1][1]

        def _rel(a, b):  # global-max normalisation (the north star's "1e-4 relative")
            return float(np.abs(a - b).max() / np.abs(b).max())

        def _rel_rms(a, b):  # the tensor's own scale: max error over its RMS (small-magnitude outputs are not hidden)
            return float(np.abs(a - b).max() / (np.sqrt(np.mean(np.square(b.astype(np.float64)))) + 1e-30))

        j = last_outs[-1]["joints"][:n].cpu().numpy()
        v = last_outs[-1]["verts"][:n].cpu().numpy()
        parity = {
            # "MPJPE vs ref" of BASELINE.json: mean Euclidean distance to the oracle's joints, same inputs and weights, on the
            # images of the CPU sample taken out of the full-size batch (so the Winograd / chunked paths are what is checked)
            "mpjpe_vs_oracle": float(np.linalg.norm(j - ref["generated_joints"], axis=-1).mean()),
            "verts_rel_err": _rel(v, ref["generated_verts"]),
            "joints_rel_err": _rel(j, ref["generated_joints"]),
            "bar": "1e-4 relative fp32",
        }
        for key, rk in (("J_transformed", "J_transformed"), ("theta", "theta"), ("kp2d", "generated_kp2d"), ("cams", "generated_cams")):
            if key in last_outs[-1]:
                a = last_outs[-1][key][:n].cpu().numpy()
                parity["%s_rel_err" % key] = _rel(a, ref[rk])
                parity["%s_rel_rms" % key] = _rel_rms(a, ref[rk])
        if "J_transformed" in last_outs[-1]:
            j24 = last_outs[-1]["J_transformed"][:n].cpu().numpy()
            parity["mpjpe24_vs_oracle"] = float(np.linalg.norm(j24 - ref["J_transformed"], axis=-1).mean())
        parity["note"] = ("*_rel_err = max|d| / max|ref| (the north star's bar, gated); *_rel_rms = max|d| / RMS(ref), the tensor's own "
                          "scale (reported; kp2d = s(x+t) is ill-conditioned where the synthetic camera scale s has cancelled to ~0.03)")
        gated = [k for k in parity if k.endswith("_rel_err")] + ["cams_rel_rms", "theta_rel_rms"]
        parity["worst_gated"] = max(parity[k] for k in gated)
        parity["pass"] = bool(args.encoder_dtype != "fp32" or parity["worst_gated"] <= PARITY_BAR)
        if args.encoder_dtype != "fp32":
            parity["bar"] = "reported, not gated (bf16 encoder, BASELINE configs[3])"

    if rank == 0:
        value = world * B * args.steps / dt
        line = {
            "metric": "images/sec (224x224, batch 256/GPU)",
            "value": round(value, 2),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.encoder_dtype == "fp32" else "bf16 (encoder; fp32 accumulate, fp32 regressor+SMPL)",
            "data": "synthetic",
            "config": {
                "workload": "batch=%d/GPU 224x224x3 synthetic images, %s ResNet-50 v1 + 3-iter regressor + SMPL LBS at all 3 "
                            "stages%s (BASELINE metric config: batch 256/GPU; %s)" % (
                                B, args.encoder_dtype, ", RCCL all-gather of theta" if use_dist else "",
                                ("configs[3]" if args.encoder_dtype == "bf16" else "configs[2]") if world > 1 else
                                ("configs[3] on one GPU" if args.encoder_dtype == "bf16" else "configs[1] at the metric batch")),
                "global_batch": world * B,
                "parallelism": "dp%d (batch shard, replicated weights)" % world,
                "pipeline": ("regressor+SMPL tail of batch k overlaps the encoder of batch k+1 (tail stream); all work of the K steps "
                             "completes inside the timed region") if pipe else "serial steps",
            },
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        if args.config5:
            line["config"]["workload"] += " + kp/mesh reprojection losses of all 3 stages, one all-reduce of the [3,4] loss block (configs[4])"
            pk = losses["packed"].cpu().numpy()
            line["losses_last_step"] = {"kpr": [float(60.0 * x) for x in pk[:, 2]], "mr": [float(0.001 * x) for x in pk[:, 3]]}
            line["loss_roofline"] = loss_roofline
        if phase:
            line["phase_ms"] = phase
        if sustained:
            line["sustained"] = sustained
        if parity:
            line["parity"] = parity
        # stdout carries exactly one line: native libraries (RCCL prints a version banner on fd 1 when its communicator is
        # created) wrote to stderr for the whole run, the JSON goes to the real stdout
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if use_dist:
        if rank == 0 and world == 1:
            last = (step_no[0] - 1) & 1
            assert torch.equal(theta_all[last], plans[last][1][-1]["theta"]), "all-gather at world 1 must return the local theta"
        dist.destroy_process_group()
    if parity is not None and not parity["pass"]:
        print("bench.py: PARITY FAILED: worst gated relative error %.3g > %.1e" % (parity["worst_gated"], PARITY_BAR), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
