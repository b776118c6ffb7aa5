#!/usr/bin/env python3
"""bench.py — predicted frames/s of the convlstm-shi hot path (BASELINE.json metric) on N MI355X GPUs of one node.

Headline workload (config.workload): the default `convlstm-shi` model (EF-ConvLSTM, vp_suite/models/
precipitation_nowcasting/ef_conv_lstm.py:31-65) on MovingMNIST-shaped synthetic frames [B, 10+10, 1, 64, 64], 10 context ->
10 predicted frames (BASELINE.json configs[1]). One "step" = one pass of the hot path over one batch: mode=infer is
VPModel.forward under no_grad; mode=train is the model's training iteration (forward + MSE + backward, PredRNN:
forward and time-reversed forward) + gradient all-reduce (RCCL) + Adam. Batch-sharded data parallel, per-GPU batch
fixed (weak scaling), no data-path collective in infer mode.

Launch forms:
  python bench.py --gpus 1 ...                   one process
  python bench.py --gpus N ...   (N > 1)         spawns `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
                                                 CHILD process before anything touches the GPU, relays its output and exit code
  python -m torch.distributed.run ... bench.py   one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment)

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     — live HIP-event timing of the dominant kernel (fused ConvLSTM cell), algorithmic FLOPs / duration
  cpu_baseline — the oracle's PyTorch-CPU restatement of the same forward, timed on the host cores (rank 0, N=1 only)
  extras       — further configurations (small batch, training step, exact fp32, predrnn-pp, 128x128x3 10->20), each timed
                 for >= --extras-seconds with its own roofline (N=1: 8 entries; N>1: the training / C4-shaped entries that
                 carry the gradient all-reduce)
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: f32 matrix 157.3 TF, bf16 dense ~2.5 PF. The split-bf16 mode ("bf16x3") spends three bf16 MFMAs
# per algorithmic product, so the dense peak of THAT arithmetic is 2500 / 3 algorithmic TFLOP/s.
BF16_DENSE_TFLOPS = 2500.0
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": BF16_DENSE_TFLOPS / 3.0, "bf16": BF16_DENSE_TFLOPS}
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prewarm", type=float, default=0.5,
                    help="seconds of untimed steps before the W warmup steps (brings an idle GPU out of its low-power state); 0 = none")
    ap.add_argument("--batch", type=int, default=128,
                    help="per-GPU batch. 128 fills the chip on every block of the model; the reference's default training "
                         "batch_size is 32 (defaults.py:37-64); both, and batch 4, are reported in `extras`")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer")
    ap.add_argument("--model", choices=["convlstm-shi", "predrnn-pp"], default="convlstm-shi",
                    help="convlstm-shi = BASELINE configs[1] (the bench line); predrnn-pp = configs[2] (secondary workload)")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "bf16"], default="bf16x3",
                    help="operand mode of the fused cell kernels: f32 = exact fp32 MFMA; bf16x3 = split-bf16 operands, "
                         "3 bf16 MFMAs per product, fp32 accumulate (fp32-level accuracy, parity-tested at 1e-4)")
    ap.add_argument("--img", type=int, default=64)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--context", type=int, default=10)
    ap.add_argument("--pred", type=int, default=10)
    ap.add_argument("--layers", type=int, default=None, help="predrnn-pp: num_layers (default 3; BASELINE configs[4] 'deep' = 4)")
    ap.add_argument("--cell", type=str, default=None,
                    help="Cin,Ch,H,W: time ONE ConvLSTM block alone (kernel micro-bench, SURVEY.md §8d) instead of the model")
    ap.add_argument("--name", type=str, default=None, help="name of the configuration (selects the committed PMC summary for roofline.traffic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--extras-seconds", type=float, default=3.0, help="minimum timed region of each `extras` entry")
    ap.add_argument("--stub", action="store_true",
                    help="(tests/test_host_logic.py) a CPU stand-in for the model on the gloo backend: the launcher, the rank plumbing, the "
                         "barrier + MAX-over-ranks timing brackets and the one JSON line of rank 0 run without a GPU; measures nothing")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run as a child process.
    Nothing in this process has touched the GPU yet (no torch.cuda call, not even `import torch`), and the child is a
    fresh process — never an exec of a GPU-initialised one."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required for RCCL on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class Spec:
    """One measured configuration."""

    def __init__(self, name, model="convlstm-shi", mode="infer", batch=128, precision="bf16x3", img=64, channels=1,
                 context=10, pred=10, layers=None, cell=None):
        self.name, self.model, self.mode, self.batch, self.precision = name, model, mode, batch, precision
        self.img, self.channels, self.context, self.pred, self.layers = img, channels, context, pred, layers
        self.cell = cell   # (Cin, Ch, H, W): ONE ConvLSTM block alone (kernel micro-bench), T = context steps

    def workload(self):
        if self.cell:
            cin, ch, h, w = self.cell
            return (f"one ConvLSTM block (conv_lstm_hzzone.py:38-70) alone: Cin={cin}, Ch={ch}, {h}x{w} map, 3x3, peepholes, "
                    f"T={self.context} steps, zero initial state, random input sequence and weights")
        deep = f", num_layers={self.layers}" if self.layers else ""
        return (f"{self.model} (default hyper-parameters{deep}) on MovingMNIST-shaped synthetic frames "
                f"{self.channels}x{self.img}x{self.img}, {self.context}->{self.pred}, random-init weights")


class Runner:
    def __init__(self, spec, dev, rank, world, use_dist):
        import torch
        from vp_suite_amd.models import MODEL_CLASSES
        self.spec, self.dev, self.rank, self.world, self.use_dist = spec, dev, rank, world, use_dist
        torch.manual_seed(0)  # identical random-init weights on every rank
        if spec.cell:
            from vp_suite_amd import ops
            cin, ch, h, w = spec.cell
            self.model = None
            self.cell_w = torch.randn(4 * ch, cin + ch, 3, 3, device=dev) / math.sqrt(9.0 * (cin + ch))
            self.cell_b = torch.randn(4 * ch, device=dev) * 0.1
            self.cell_peep = [torch.randn(1, ch, h, w, device=dev) * 0.1 for _ in range(3)]
            torch.manual_seed(42 + rank)
            self.x = ops.to_channels_last(torch.rand(spec.batch, spec.context, cin, h, w, device=dev))
            self.semantics = "ops.convlstm_seq (= model_blocks.ConvLSTM.forward) under no_grad: T fused cell-step launches per call"
            return
        kw = dict(img_shape=(spec.channels, spec.img, spec.img), action_size=0, tensor_value_range=[0.0, 1.0],
                  cell_precision=spec.precision)
        if spec.layers:
            kw["num_layers"] = spec.layers
        self.model = MODEL_CLASSES[spec.model](str(dev), **kw).to(dev)
        if os.environ.get("VPX_BENCH_EXPERIMENT"):   # A/B runs (tools/): bits of vpx_set_option(VPX_OPT_EXPERIMENT), include/vpx.h
            from vp_suite_amd import _lib as vlib
            vlib.lib().vpx_set_option(vlib.OPT_EXPERIMENT, int(os.environ["VPX_BENCH_EXPERIMENT"]))
        if os.environ.get("VPX_BENCH_FUSE_REVERSED") in ("0", "1") and hasattr(self.model, "fuse_reversed_pass"):   # A/B runs (tools/)
            self.model.fuse_reversed_pass = os.environ["VPX_BENCH_FUSE_REVERSED"] == "1"
        if os.environ.get("VPX_BENCH_DECOUPLE_SLAB_LIMIT") and hasattr(self.model, "DECOUPLE_SLAB_LIMIT"):   # A/B runs (tools/)
            self.model.DECOUPLE_SLAB_LIMIT = int(os.environ["VPX_BENCH_DECOUPLE_SLAB_LIMIT"])
        if os.environ.get("VPX_BENCH_DEFER_WGRAD") in ("0", "1") and hasattr(self.model, "defer_weight_gradients"):   # A/B runs (tools/)
            self.model.defer_weight_gradients = os.environ["VPX_BENCH_DEFER_WGRAD"] == "1"
        with torch.no_grad():
            for n, p in self.model.named_parameters():
                if n.split(".")[-1] in ("Wci", "Wcf", "Wco"):
                    p.normal_(0.0, 0.1)  # exercise the peephole path (reference init is zeros)
        torch.manual_seed(42 + rank)
        frames = torch.rand(spec.batch, spec.context + spec.pred, spec.channels, spec.img, spec.img, device=dev)
        complete = self.model.NEEDS_COMPLETE_INPUT
        self.x, self.target = (frames if complete else frames[:, :spec.context]), frames[:, spec.context:]
        self.semantics = "VPModel.forward under no_grad"
        if spec.mode == "train":
            from vp_suite_amd.train import DataParallelTrainer
            self.trainer = DataParallelTrainer(self.model, lr=1e-4, world_size=world, force_collectives=use_dist)
            self.semantics = ("model.training_loss (PredRNN_V2.train_iter semantics: train-time sampling mask, forward + "
                              "time-reversed forward averaged) + backward + flat-bucket all-reduce + fused Adam"
                              if spec.model == "predrnn-pp" else
                              "forward + MSE + backward (BPTT) + flat-bucket all-reduce + fused Adam")

    def step(self):
        import torch
        if self.spec.cell:
            from vp_suite_amd import ops
            with torch.no_grad():
                ops.convlstm_seq(self.x, None, None, self.cell_w, self.cell_b, *self.cell_peep, seq_len=self.spec.context,
                                 in_channels=self.spec.cell[0], precision=self.spec.precision)
            return
        if self.spec.mode == "train":
            self.trainer.step(self.x, self.target, self.spec.pred)
        else:
            with torch.no_grad():
                self.model(self.x, pred_frames=self.spec.pred)

    def barrier(self):
        import torch
        import torch.distributed as dist
        if self.use_dist:
            dist.barrier()
        if self.dev.type == "cuda":
            torch.cuda.synchronize()

    def timed(self, steps, warmup):
        """W untimed steps, then EXACTLY `steps` steps between barrier + synchronize brackets; MAX over ranks."""
        import torch
        import torch.distributed as dist
        from vp_suite_amd import ops
        for _ in range(warmup):
            self.step()
        prof = ops.KernelProfile()
        ops.PROFILE = prof
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.barrier()
        elapsed = time.perf_counter() - t0
        ops.PROFILE = None
        if self.use_dist:
            t = torch.tensor([elapsed], device=self.dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, prof.summary()

    def prewarm(self, seconds):
        """Untimed steps for about `seconds` of wall time BEFORE the W warmup steps: an idle MI355X sits in a low-power state
        (rocm-smi: sclk ~100 MHz) and a short run — B=4: 23 steps of 2 ms — can end inside the clock ramp (measured: the first
        process on an idle box 4.2 ms per step, the next two 2.12). Same step count on every rank. Returns the steps run."""
        import torch
        import torch.distributed as dist
        if seconds <= 0:
            return 0
        self.step()
        self.barrier()
        t0 = time.perf_counter()
        self.step()
        self.barrier()
        n = min(2000, int(math.ceil(seconds / max(time.perf_counter() - t0, 1e-6))))
        if self.use_dist:
            t = torch.tensor([n], device=self.dev, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n = int(t.item())
        for _ in range(n):
            self.step()
        self.barrier()
        return n + 2

    def calibrated_steps(self, seconds):
        """Step count for a timed region of at least `seconds` (same count on every rank)."""
        import torch
        import torch.distributed as dist
        el, _ = self.timed(2, 2)
        n = max(3, int(math.ceil(seconds / max(el / 2, 1e-6))))
        if self.use_dist:
            t = torch.tensor([n], device=self.dev, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n = int(t.item())
        return n


def loaded_lib_sha16():
    """sha256(vp-suite_amd/libvpx_hip.so)[:16] of the library this process runs on."""
    import hashlib
    try:
        with open(os.path.join(ROOT, "vp-suite_amd", "libvpx_hip.so"), "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        return None


def traffic_from_file(path, lib_sha):
    """(bytes per launch, source) of one committed PMC summary — (None, reason) when the file was collected on ANOTHER build of the library
    (its `lib_sha16` differs from the loaded library's, or it has none: pre-round-6 files): a kernel change without a fresh PMC pass must
    not keep reporting the old bytes."""
    rel = "profiles/" + os.path.basename(path)
    with open(path) as fh:
        doc = json.load(fh)
    t = doc.get("hbm_traffic_bytes_per_launch")
    if t is None:
        return None, None
    if doc.get("lib_sha16") is None or doc.get("lib_sha16") != lib_sha:
        return None, f"stale: {rel} was collected on library {doc.get('lib_sha16')}, this run loaded {lib_sha}"
    return round(t["total"]), rel


class StubRunner(Runner):
    """--stub: what a rank does around the model, without the model (CPU tensors, gloo). A step sleeps VPX_BENCH_STUB_MS[rank] milliseconds
    (comma-separated per rank; default 2) and, in train mode, all-reduces a small bucket — so a test can see that the slowest rank sets
    the reported time."""

    def __init__(self, spec, dev, rank, world, use_dist):
        import torch
        self.spec, self.dev, self.rank, self.world, self.use_dist = spec, dev, rank, world, use_dist
        self.model = None
        ms = [float(v) for v in os.environ.get("VPX_BENCH_STUB_MS", "2").split(",")]
        self.sleep_s = ms[min(rank, len(ms) - 1)] * 1e-3
        self.bucket = torch.ones(1024)
        self.semantics = "stub (no model): sleep + gloo all-reduce in train mode"

    def step(self):
        import torch.distributed as dist
        time.sleep(self.sleep_s)
        if self.spec.mode == "train" and self.use_dist:
            dist.all_reduce(self.bucket)


def measured_traffic(spec):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/, collected with
    tools/collect_profiles.sh on this exact workload: separate --pmc FETCH_SIZE / WRITE_SIZE passes, read = 2 x FETCH_SIZE
    per the gfx950 correction); None when no committed pass matches the configuration OR the library that was measured."""
    lib_sha = loaded_lib_sha16()
    # a PMC summary committed under this configuration's name (tools/prof_extra.sh + tools/summarize_extra.py) ...
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(f"_pmc_{spec.name}.json"))
    if cands:
        return traffic_from_file(os.path.join(ROOT, "profiles", cands[-1]), lib_sha)
    if spec.cell:
        return None, None
    # ... or the headline collection's (tools/collect_profiles.sh) for the 64x64 10->10 convlstm-shi configurations
    if not (spec.img == 64 and spec.channels == 1 and spec.context == 10 and spec.pred == 10):
        return None, None
    tag = {"convlstm-shi": ""}.get(spec.model)
    if tag is None:
        return None, None
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles"))
                   if f.endswith(f"_pmc_bench_{spec.mode}_b{spec.batch}_{spec.precision}.json"))
    if not cands:
        return None, None
    return traffic_from_file(os.path.join(ROOT, "profiles", cands[-1]), lib_sha)


def roofline(spec, ps):
    ach_tflops = ps["flops"] / (ps["ms"] * 1e-3) / 1e12 if ps["ms"] > 0 else 0.0
    ach_gbs = ps["bytes"] / (ps["ms"] * 1e-3) / 1e9 if ps["ms"] > 0 else 0.0
    peak = PEAK_TFLOPS[spec.precision]
    launches = max(ps["launches"], 1)
    traffic, traffic_source = measured_traffic(spec)   # NOT measured in this run: the committed PMC pass of the same configuration
    return {
        "bound": "mfma", "achieved": round(ach_tflops, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
        "frac": round(ach_tflops / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
        "algorithmic_bytes_per_launch": round(ps["bytes"] / launches),
        "wide_read_bytes_per_launch": round(ps.get("wide_read_bytes", 0.0) / launches),
        "kernel": ((f"cell2_kernel_q<Cell2Epi, true, 4> (second-generation fused ConvLSTM cell step: pre-split bf16x3 operands, "
                    f"LDS-DMA staging, v_mfma_f32_16x16x32_bf16, 16x16-pixel tiles at two workgroups per CU; cell3_kernel — 8-channel "
                    f"slices, hoisted input projection — on grids below 256 workgroups), forward" if spec.precision == "bf16x3" else
                    ("cell2_kernel_q<Cell2Epi, true, 4, true> (the same fused step on the hi parts only: one v_mfma_f32_16x16x32_bf16 per "
                     "product, no lo planes staged; 16x16 maps / small grids: conv_gemm_kernel<EpiConvLSTM, bf16>), forward"
                     if spec.precision == "bf16" else
                     f"conv_gemm_kernel<EpiConvLSTM, {spec.precision}> (fused ConvLSTM cell step, forward)"))
                   if spec.model == "convlstm-shi" else
                   ("c5_kernel<8> + c5_kernel<2|4> + c1_kernel (ST-LSTM cell step, forward: both gate groups as the jobs of one launch on 16x16-pixel "
                    "tiles with 8-channel stages, fused gate math; conv_o + output gate; conv_last as a streaming 1x1; K-split job forms + "
                    "pointwise stages on grids below 48 pixel tiles)" if spec.precision == "bf16x3" else
                    f"conv_gemm_kernel<EpiSTGate/EpiSTOut/EpiPlain, {spec.precision}> (ST-LSTM cell step, forward)")),
        "note": ("achieved = algorithmic fp32 FLOPs / kernel time. bf16x3 issues 3 bf16 MFMAs per algorithmic "
                 "product: peak = 2500 TF dense bf16 / 3, i.e. frac is the share of the bf16 MFMA pipe's dense "
                 "peak the kernel keeps busy; frac_of_bf16_dense_peak prices the same time against the plain 2500 TF"
                 if spec.precision == "bf16x3" else
                 ("plain bf16 operands (hi parts only, one MFMA per product: cell2_kernel_q<.., 4, true>): outside the 1e-4 parity bar, "
                  "reported as an extra" if spec.precision == "bf16"
                  else "exact fp32 MFMA (v_mfma_f32_32x32x2_f32)")),
        "frac_of_bf16_dense_peak": (round(ach_tflops / BF16_DENSE_TFLOPS, 4) if spec.precision != "f32" else None),
        "vs_fp32_matrix_peak": round(ach_tflops / PEAK_TFLOPS["f32"], 4),
        "launches": ps["launches"], "avg_launch_us": round(ps["ms"] * 1e3 / launches, 2),
        "algorithmic_gflop_per_launch": round(ps["flops"] / launches / 1e9, 3),
        "hbm_view": {"achieved_GBps": round(ach_gbs, 1), "peak_GBps": HBM_PEAK_GBS, "frac": round(ach_gbs / HBM_PEAK_GBS, 4),
                     "note": "north-star 'fraction of HBM roofline' of the fused cell: algorithmic bytes / kernel time; "
                             "the cell is MFMA-bound (>=339 FLOP/B), so this stays far below 0.5 at 1e-4 parity (SURVEY.md §8d)"},
    }


def run_extra(spec, dev, rank, world, use_dist, seconds):
    import torch
    r = Runner(spec, dev, rank, world, use_dist)
    steps = r.calibrated_steps(seconds)
    elapsed, ps = r.timed(steps, 1)
    units = spec.context if spec.cell else spec.pred   # a cell entry counts cell steps of one sample, a model entry predicted frames
    out = {"name": spec.name, "workload": spec.workload(), "mode": spec.mode, "semantics": r.semantics,
           "dtype": spec.precision, "per_gpu_batch": spec.batch, "global_batch": spec.batch * world, "n_gpus": world,
           "steps": steps, "timed_region_s": round(elapsed, 3), "ms_per_step": round(elapsed / steps * 1e3, 4),
           "value": round(world * spec.batch * units * steps / elapsed, 2),
           "unit": "cell steps x samples/s" if spec.cell else "frames/s",
           "roofline": roofline(spec, ps)}
    for k in ("note", "vs_fp32_matrix_peak") + (() if spec.cell else ("hbm_view",)):
        out["roofline"].pop(k, None)
    if spec.cell:
        out["roofline"]["hbm_view"]["note"] = (
            "north-star 'fraction of HBM roofline' on the fused cell: algorithmic bytes (SURVEY.md §8d) / kernel time / 8 TB/s. "
            "Structural ceiling at 1e-4 parity: the cell is MFMA-bound (2.416 GFLOP vs 5.24 MB per unit at 64x64x64ch), bf16x3 "
            "spends 3 MFMAs per product, so frac <= 5.24 MB / (2.416 GFLOP / 833 TF) / 8 TB/s = 0.23 at roofline.frac = 1")
    del r
    torch.cuda.empty_cache()
    return out


def host_topology():
    """Host cores of this box as `lscpu` / the cgroup / the affinity mask state them: north_star asks for the CPU figure 'on the same
    box's host cores (core count stated)'. usable = what a process here may actually occupy (min of physical cores, affinity, CPU quota)."""
    import subprocess
    topo = {"logical": os.cpu_count()}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        want = {"Thread(s) per core": "threads_per_core", "Core(s) per socket": "cores_per_socket", "Socket(s)": "sockets",
                "NUMA node(s)": "numa_nodes", "Model name": "model"}
        for ln in out.splitlines():
            k, _, v = ln.partition(":")
            if k.strip() in want:
                v = v.strip()
                topo[want[k.strip()]] = int(v) if v.isdigit() else v
    except Exception:   # noqa: BLE001 (no lscpu: the counts below still hold)
        pass
    try:
        topo["affinity"] = len(os.sched_getaffinity(0))
    except Exception:   # noqa: BLE001
        topo["affinity"] = topo["logical"]
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
        topo["cgroup_cpus"] = None if q == "max" else round(int(q) / int(per), 2)
    except Exception:   # noqa: BLE001
        topo["cgroup_cpus"] = None
    phys = topo.get("cores_per_socket", 0) * topo.get("sockets", 0) or max(1, (topo["logical"] or 2) // 2)
    usable = min(phys, topo["affinity"] or phys)
    if topo["cgroup_cpus"]:
        usable = max(1, min(usable, int(topo["cgroup_cpus"])))
    topo["physical_cores"], topo["usable_cores"] = phys, usable
    return topo


def cpu_baseline(model, spec, seconds):
    """Times the oracle's plain-PyTorch CPU restatement (oracle/torch_ref.py) of the same forward on the host cores. Two bounded samples
    of the same 10->10 workload: (1) batch 4 (BASELINE configs[0]) at the best point of a short thread scan — the small problem
    over-subscribes a big socket; (2) batch 32, one warm-up + timed iterations on ALL usable physical cores (host_topology) — the point
    north_star words ('host cores, core count stated'). `value` = the better of the two, `cores` = the threads it ran on."""
    import torch
    from oracle import torch_ref
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    all_cores = torch.get_num_threads()
    topo = host_topology()
    b = 4
    x = torch.rand(b, spec.context, spec.channels, spec.img, spec.img)
    best, scan = None, {}
    for threads in sorted({all_cores, min(all_cores, 32), min(all_cores, 16)}, reverse=True):
        torch.set_num_threads(threads)
        with torch.no_grad():
            torch_ref.ef_convlstm_forward(sd, x, spec.pred)  # warm-up
            n, t0 = 0, time.perf_counter()
            while True:
                torch_ref.ef_convlstm_forward(sd, x, spec.pred)
                n += 1
                el = time.perf_counter() - t0
                if el > seconds / 6 or n >= 50:
                    break
        fps = n * b * spec.pred / el
        scan[str(threads)] = round(fps, 2)
        if best is None or fps > best[0]:
            best = (fps, threads, n, el)
    # (2) a batch that can use the socket: all usable physical cores, bounded to ~seconds / 2
    bb, tb = 32, max(1, min(topo["usable_cores"], all_cores))
    torch.set_num_threads(tb)
    xb = torch.rand(bb, spec.context, spec.channels, spec.img, spec.img)
    with torch.no_grad():
        t0 = time.perf_counter()
        torch_ref.ef_convlstm_forward(sd, xb, spec.pred)   # warm-up (also sizes the timed part)
        warm = time.perf_counter() - t0
        nb, t0 = 0, time.perf_counter()
        while True:
            torch_ref.ef_convlstm_forward(sd, xb, spec.pred)
            nb += 1
            elb = time.perf_counter() - t0
            if elb + warm > seconds / 2 or nb >= 8:
                break
    fps_b = nb * bb * spec.pred / elb
    torch.set_num_threads(all_cores)
    fps, threads, n, el = best
    points = [{"batch": b, "threads": threads, "value": round(fps, 2), "iterations": n, "seconds": round(el, 2)},
              {"batch": bb, "threads": tb, "value": round(fps_b, 2), "iterations": nb, "seconds": round(elb, 2)}]
    top = max(points, key=lambda q: q["value"])
    return {"value": top["value"], "unit": "predicted frames/s", "cores": top["threads"], "kind": "port",
            "points": points, "host": topo, "all_cores": {"cores": all_cores, "value": scan[str(all_cores)]}, "thread_scan": scan,
            "sample": f"oracle/torch_ref.ef_convlstm_forward (PyTorch-CPU restatement of the reference path), {spec.context}->{spec.pred}, "
                      f"{spec.channels}x{spec.img}x{spec.img}: batch {b} on {threads} threads (best of a scan {{all,32,16}}) {fps:.1f} f/s; "
                      f"batch {bb} on {tb} threads (= usable physical cores of {topo.get('sockets', '?')} socket(s) x "
                      f"{topo.get('cores_per_socket', '?')} cores, {topo.get('numa_nodes', '?')} NUMA node(s)) {fps_b:.1f} f/s; value = the better"}


# The driver keeps a bounded tail of stdout and parses the LAST line: r02's 10.5 KB line parsed, r03's 27 KB line did not
# (BENCH_r03.json: "parsed": null). The final stdout line is therefore the compact form below (< LINE_BUDGET bytes); the full
# record (long kernel / workload / semantics / note strings of every entry) goes to bench_extras.json.
LINE_BUDGET = 6144
ROOFLINE_KEEP = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch", "wide_read_bytes_per_launch",
                 "launches", "avg_launch_us", "frac_of_bf16_dense_peak")


def compact_roofline(rf):
    out = {k: rf[k] for k in ROOFLINE_KEEP if k in rf}
    hv = rf.get("hbm_view")
    if hv:
        out["hbm_view"] = {"achieved_GBps": hv["achieved_GBps"], "peak_GBps": hv["peak_GBps"], "frac": hv["frac"]}
    return out


def compact_extra(e):
    if "error" in e:
        return {"name": e["name"], "error": e["error"][:120]}
    rf = e.get("roofline", {})
    out = {"name": e["name"], "dtype": e["dtype"], "ms_per_step": e["ms_per_step"], "value": e["value"],
           "unit": "cellsteps/s" if e["unit"].startswith("cell") else e["unit"], "frac": rf.get("frac"),
           "traffic": rf.get("traffic")}
    if rf.get("hbm_view"):
        out["hbm_frac"] = rf["hbm_view"]["frac"]
    return out


def compact_line(full):
    """The ONE stdout line: headline + roofline + cpu_baseline + per-extra {name, dtype, ms_per_step, value, unit, frac,
    hbm_frac, traffic}; everything long lives in the sidecar file named by `extras_file`."""
    out = {k: v for k, v in full.items() if k not in ("roofline", "cpu_baseline", "extras", "config")}
    cfg = dict(full["config"])
    cfg.pop("semantics", None)
    if isinstance(cfg.get("prewarm"), dict):
        cfg["prewarm"] = {k: cfg["prewarm"][k] for k in ("seconds", "steps") if k in cfg["prewarm"]}
    out["config"] = cfg
    out["roofline"] = compact_roofline(full["roofline"])
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "points", "all_cores", "sample") if k in cb}
        if "host" in cb:
            out["cpu_baseline"]["host"] = {k: cb["host"].get(k) for k in ("logical", "physical_cores", "usable_cores", "sockets", "numa_nodes", "cgroup_cpus")}
        out["cpu_baseline"]["sample"] = cb["sample"][:360]
    if "extras" in full:
        out["extras"] = [compact_extra(e) for e in full["extras"]]
        out["extras_file"] = "bench_extras.json"
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= LINE_BUDGET:   # never let the extras cost the headline: drop them from the line, they stay in the sidecar
        out["extras"] = [{"name": e["name"], "value": e.get("value"), "frac": e.get("frac")} for e in out.get("extras", [])]
        line = json.dumps(out, separators=(",", ":"))
    if len(line) >= LINE_BUDGET:
        out.pop("extras", None)
        line = json.dumps(out, separators=(",", ":"))
    return line


def extras_for(world):
    if world == 1:
        return [
            Spec("infer_b128"),   # the headline configuration timed for >= --extras-seconds (the same name in the N > 1 list)
            Spec("infer_b4", batch=4),
            Spec("infer_b32", batch=32),
            Spec("train_b32", mode="train", batch=32),
            Spec("train_b128", mode="train", batch=128),
            Spec("infer_b128_f32", precision="f32"),
            # BASELINE configs[1]'s literal dtype: plain bf16 operands (hi parts only) — OUTSIDE the 1e-4 parity bar (max|d|/max|ref|
            # ~2e-3 per block, 3e-2 held at model level), reported next to the bf16x3 headline, never mixed into it
            Spec("infer_b128_bf16", precision="bf16"),
            Spec("predrnn_infer_b128", model="predrnn-pp"),
            Spec("c4_infer_b4_128x128x3_10to20", batch=4, img=128, channels=3, pred=20),
            Spec("c4_train_b4_128x128x3_10to20", mode="train", batch=4, img=128, channels=3, pred=20),
            # the cell the north-star states its target on (K1 at (Cin,Ch,H,W) = (64,64,64,64), B in {4,32,128}) ...
            Spec("cell_64x64x64_b128", cell=(64, 64, 64, 64), batch=128),
            Spec("cell_64x64x64_b32", cell=(64, 64, 64, 64), batch=32),
            Spec("cell_64x64x64_b4", cell=(64, 64, 64, 64), batch=4),
            Spec("cell_64x64x64_b128_bf16", cell=(64, 64, 64, 64), batch=128, precision="bf16"),
            # ... and the six block shapes of convlstm-shi at the default batch
            Spec("cell_enc1_16x64x64_b128", cell=(16, 64, 64, 64), batch=128),
            Spec("cell_enc2_64x96x32_b128", cell=(64, 96, 32, 32), batch=128),
            Spec("cell_enc3_96x96x16_b128", cell=(96, 96, 16, 16), batch=128),
            Spec("cell_fore2_96x96x32_b128", cell=(96, 96, 32, 32), batch=128),
            Spec("cell_fore1_96x64x64_b128", cell=(96, 64, 64, 64), batch=128),
            # PredRNN-V2 (BASELINE configs[2]) training iteration, and the deep long-horizon shape of configs[4]
            Spec("predrnn_train_b128", model="predrnn-pp", mode="train", batch=128),
            Spec("predrnn_train_b32", model="predrnn-pp", mode="train", batch=32),
            Spec("c5_infer_b4_128x128x3_10to30_L4", model="predrnn-pp", batch=4, img=128, channels=3, pred=30, layers=4),
            Spec("c5_train_b2_128x128x3_10to30_L4", model="predrnn-pp", mode="train", batch=2, img=128, channels=3, pred=30, layers=4),
        ]
    # N > 1: the entries in which ranks exchange gradients (the north-star's DP-scaling figure), at the default batch and
    # at BASELINE configs[3]'s 4 samples per GPU
    return [
        Spec("infer_b128"),   # the headline again, sustained: shared with the N = 1 line (no collective in this mode)
        Spec("train_b128", mode="train", batch=128),
        Spec("c4_train_b4_128x128x3_10to20", mode="train", batch=4, img=128, channels=3, pred=20),
        Spec("c4_infer_b4_128x128x3_10to20", batch=4, img=128, channels=3, pred=20),
    ]


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL on this host driver); before the HIP runtime starts
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" in os.environ and world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} ranks; reporting n_gpus={world}", file=sys.stderr)
    if args.stub:
        dev = torch.device("cpu")
        args.no_extras = args.no_cpu_baseline = True
        args.prewarm = 0.0
    else:
        ndev = torch.cuda.device_count()
        if local_rank >= ndev:
            print(f"[bench] rank {rank}: local rank {local_rank} has no GPU ({ndev} visible); one process per GPU is required",
                  file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ  # under torch.distributed.run always go through RCCL (also at N=1)
    os.environ.setdefault("NCCL_DEBUG", "WARN")  # keep RCCL's version banner off stdout: rank 0 prints exactly one JSON line
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        world = dist.get_world_size()  # the size RCCL reports

    import vp_suite_amd  # noqa: F401

    spec = Spec(args.name or "headline", model=args.model, mode=args.mode, batch=args.batch, precision=args.precision, img=args.img,
                channels=args.channels, context=args.context, pred=args.pred, layers=args.layers,
                cell=tuple(int(v) for v in args.cell.split(",")) if args.cell else None)
    runner = (StubRunner if args.stub else Runner)(spec, dev, rank, world, use_dist)
    prewarm_steps = runner.prewarm(args.prewarm)
    elapsed, ps = runner.timed(args.steps, args.warmup)

    out = None
    if rank == 0:
        frames_total = world * spec.batch * (spec.context if spec.cell else spec.pred) * args.steps
        out = {
            "metric": "predicted frames/sec (whole node), MovingMNIST 64x64 10->10",
            "value": round(frames_total / elapsed, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": spec.precision,
            "data": "synthetic",
            "config": {"workload": spec.workload(), "mode": spec.mode, "semantics": runner.semantics,
                       "per_gpu_batch": spec.batch, "global_batch": spec.batch * world, "parallelism": f"dp{world}",
                       "ranks": world, "backend": (("gloo (stub)" if args.stub else "nccl (RCCL)") if use_dist else "single process"),
                       "prewarm": {"seconds": args.prewarm, "steps": prewarm_steps,
                                   "what": "untimed steps before the W warmup steps: clock ramp out of the idle power state"}},
            "roofline": roofline(spec, ps),
        }
    cpu_model = runner.model if (world == 1 and not args.no_cpu_baseline and spec.model == "convlstm-shi") else None
    if cpu_model is None:
        del runner
        if not args.stub:
            torch.cuda.empty_cache()

    extras = []
    if not args.no_extras:
        for es in extras_for(world):
            try:
                extras.append(run_extra(es, dev, rank, world, use_dist, args.extras_seconds))
            except Exception as exc:  # an extra must never take the headline down; say what happened instead
                if use_dist:
                    raise
                extras.append({"name": es.name, "error": f"{type(exc).__name__}: {exc}"})
    if rank == 0:
        if extras:
            out["extras"] = extras
        if cpu_model is not None:
            out["cpu_baseline"] = cpu_baseline(cpu_model, spec, args.cpu_seconds)
        full = json.dumps(out)
        try:   # the full record: sidecar file next to bench.py (gpurun_out/ too when it exists)
            for d in [ROOT] + ([os.path.join(ROOT, "gpurun_out")] if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else []):
                with open(os.path.join(d, "bench_extras.json"), "w") as fh:
                    fh.write(full + "\n")
        except OSError as exc:
            print(f"[bench] could not write bench_extras.json: {exc}", file=sys.stderr)
        # nothing long on either stream: the driver's capture of this run is bounded (head or tail, unknown)
        print(compact_line(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
