#!/usr/bin/env python
"""bench.py -- Gibbs samples/s of the CD-1 hot path (BASELINE.json metric) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): ClassificationRBM 784 visible x 512 hidden, C = 10, CD-1, minibatch 256
PER GPU (weak scaling; with N ranks the global batch is 256*N and one RCCL all-reduce of the flat
gradient buffer precedes the update, BASELINE.json configs[3]).  A "step" = one full CD-1 update:
h0 ~ p(h|v0,y0); v1 ~ p(v|h0); y1 ~ p(y|h0); p(h|v1,y1); two-phase gradient; parameter update.
Inputs are synthetic (Philox Bernoulli(0.5) visibles, uniform labels, reference init) and resident in
HBM before the timed region.  value = B * k * steps * N / time  (Gibbs samples/s, one sample = one
full block-Gibbs sweep of one chain); visible-unit updates/s = 784 * value is printed alongside.

The JSON line also carries `roofline` (dominant kernel, HIP-event timed inside this process) and
`cpu_baseline` (a torch-CPU restatement of the same step in the reference's idiom on the host cores, with the
bit-exact C/OpenMP oracle's timing as a secondary field; rank 0, N = 1 only, bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V, H, NC, B, K_CD = 784, 512, 10, 256, 1
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0          # ibid., HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--graph", type=int, default=8, help="steps captured per hipGraph (0 = eager launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dp-graph", type=int, default=0, help="with several ranks: capture the data-parallel step (RCCL all-reduce included) in hipGraphs")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel code path (gradient all-reduce + separate update) even with one rank")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def main():
    args = parse()
    # stdout carries exactly ONE JSON line: libraries that print banners there (RCCL prints its version on
    # stdout at init) are diverted to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import gc

    import torch
    import torch.distributed as dist

    # no cycle collection in this short process: the collector may otherwise run inside a stream capture (destroying a device
    # object there is not a legal call) or put a pause into the timed region
    gc.disable()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    dp = world > 1 or args.force_dp
    if dp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:      # --force-dp without torchrun: a one-rank group
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    from qbm_image_classification_amd import _lib
    from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps, ParamSet
    from qbm_image_classification_amd._lib import QBM_U8, CdWorkspace, check, ptr

    ops = HipRbmOps(str(dev))
    lib = ops.lib

    # ---- parameters: reference init (src/ClassificationRBM.py:26-30), torch seed 42 ----------------
    torch.manual_seed(42)
    P = ParamSet((torch.randn(V, H) * 0.1).to(dev), (torch.ones(V) * 0.5).to(dev), torch.zeros(H, device=dev),
                 torch.zeros(NC, H, device=dev), torch.zeros(NC, device=dev))
    G = GradSet(V, H, NC, dev)
    lr, factor, sparse = 0.05, 1.0, 0.0

    # steps per captured graph: the requested size, or -- when it does not divide the timed step count -- the largest
    # divisor of it up to 16, so that a short timed region (the driver runs a few tens of steps) is whole graph replays
    # rather than a tail of host-bound eager launches (measured for 20 steps: graphs of 10 -> 30.7 us per step, of 20 ->
    # 31.0, of 5 -> 31.3, of 2 -> 32.8; 2000 steps in graphs of 8 -> 29.8: a region this short also pays for the idle
    # gap the synchronize in front of it leaves on the device)
    if args.graph > 0 and args.steps % args.graph != 0:
        divs = [d for d in range(2, 17) if args.steps % d == 0]
        if divs:
            args.graph = divs[-1]
    # ---- synthetic minibatches, resident in HBM ----------------------------------------------------
    NB = max(8, args.graph if args.graph > 0 else 8)
    row_base = rank * B
    data_v, data_y = [], []
    for i in range(NB):
        u = ops.uniform(1234, i, row_base, B, V)
        data_v.append((u < 0.5).to(torch.uint8).contiguous())   # binary visibles held as uint8 {0,1}
        ul = ops.uniform(4321, i, row_base, B, 1).view(-1)
        lab = torch.clamp((ul * NC).long(), max=NC - 1)
        data_y.append(torch.nn.functional.one_hot(lab, NC).float().contiguous())
    ws = ops.cd_workspace(P, B)
    cws = CdWorkspace(*(ptr(t) for t in ws))
    ps, gs = P.struct(), G.struct()
    gs_dp = G.struct(with_count=True)     # data parallel: the gradient launch stores its row count behind the gradient blocks
    global_B = float(B * world)
    # every per-step argument is built ONCE (device pointers of the resident minibatches, the two structs): a step on the
    # host is then one ctypes call -- and, data parallel, one collective
    vp = [ptr(t) for t in data_v]
    yp = [ptr(t) for t in data_y]
    pps, pgs, pgd, pcw = C.byref(ps), C.byref(gs), C.byref(gs_dp), C.byref(cws)
    cd_step, dp_step, upd_dev = lib.qbm_rbm_cd_step, lib.qbm_rbm_dp_step, lib.qbm_rbm_update_dev
    SUM = dist.ReduceOp.SUM

    def step(i, stream, first=False):
        """one CD-1 update on resident minibatch i % NB.  Data parallel: ONE C-ABI call (the update of the step before from the
        all-reduced buffer, its global row count read on the device, then this minibatch's chain and gradient:
        qbm_rbm_dp_step) and ONE collective (the flat gradient buffer with the row count, RCCL over xGMI); `first`: no update
        is pending.  The last update of a run is applied by flush()."""
        j = i % NB
        if not dp:
            check(cd_step(pps, pgs, vp[j], QBM_U8, yp[j], B, K_CD, 42, 4 * i, row_base, None, None, pcw, 1, lr, factor, global_B,
                          sparse, stream), "qbm_rbm_cd_step")
        else:
            check(dp_step(pps, pgd, vp[j], QBM_U8, yp[j], B, K_CD, 42, 4 * i, row_base, None, None, pcw, 0 if first else 1, lr,
                          factor, sparse, stream), "qbm_rbm_dp_step")
            dist.all_reduce(G.flat, op=SUM)

    def flush(stream):
        if dp:
            check(upd_dev(pps, pgd, lr, factor, 0.0, ptr(G.count), sparse, None, stream), "qbm_rbm_update_dev")

    def cur():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    graphs = None
    graph_note = None
    if dp and world > 1 and not args.dp_graph:
        # Several ranks: eager launches by default.  The captured data-parallel step has only ever run on ONE rank (no
        # multi-GPU box is available to the build); a collective captured inconsistently across ranks would not raise but
        # hang, and with an all-reduce in every step the host keeps up with the device anyway.  --dp-graph 1 captures.
        args.graph = 0
        graph_note = "data-parallel step on several ranks: eager launches (--dp-graph 1 captures RCCL in the graph)"
    if args.graph > 0 and dp:
        # data-parallel step (gradient materialised, RCCL all-reduce, update) captured as well: RCCL collectives are
        # capturable; if this build of torch / RCCL refuses, the bench falls back to eager launches and says so
        S = args.graph
        try:
            for i in range(S):
                step(i, cur(), first=(i == 0))
            flush(cur())
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(S):
                    step(i, cur(), first=(i == 0))
                flush(cur())                      # S chains, S collectives, S updates per replay
            graphs = (g, S)
        except Exception as e:   # noqa: BLE001
            graph_note = f"graph capture of the data-parallel step failed ({type(e).__name__}: {str(e)[:120]}): eager launches"
            torch.cuda.synchronize()
    if args.graph > 0 and not dp:
        # S consecutive steps (S resident minibatches) per hipGraph: amortises host launch cost
        S = args.graph
        for i in range(S):                                           # eager warm-up of every code path
            step(i, cur())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(S):
                step(i, cur())
        graphs = (g, S)

    def run(nsteps):
        """exactly nsteps CD-1 updates (data parallel: nsteps chains, nsteps collectives, nsteps updates)"""
        if graphs is None:
            st = cur()
            for i in range(nsteps):
                step(i, st, first=(i == 0))
            if nsteps:
                flush(st)
        else:
            g, S = graphs
            for _ in range(nsteps // S):
                g.replay()
            st = cur()
            for i in range(nsteps % S):
                step(i, st, first=(i == 0))
            if nsteps % S:
                flush(st)

    def barrier():
        if dp:
            dist.barrier()
        # poll an event before the blocking synchronize: the host then sees the end of the work without the wake-up
        # latency of a sleeping wait (~15 us: 2 % of a 20-step timed region); the synchronize that follows returns at once
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():
            pass
        torch.cuda.synchronize()

    # ---- roofline, part 1: per-kernel launch duration with HIP events on the launching stream.  Runs BEFORE the timed
    # region, on every rank: its ~3000 launches also bring the GPU to its steady clock state, so that a short timed region
    # (the driver times 20 steps = 0.6 ms) measures the same state a long one does; the JSON line says how many launches
    # that was (config.pre_timed_kernel_launches) -------------------------------------------------------------------
    def time_kernel(fn, reps=1000, per_graph=50, prologue=None):
        """average duration of one launch, back to back: `per_graph` launches captured in a hipGraph (the host's launch
        rate -- ~7 us per ctypes call -- would otherwise be what is measured), replayed reps / per_graph times between
        two HIP events on the launching stream.  fn(stream, i): launch i of a graph; prologue(): captured once in front"""
        st = torch.cuda.current_stream()
        sp = cur()
        if prologue:
            prologue()
        for i in range(5):
            check(fn(sp, i), "per-kernel timing launch")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            sp_g = cur()
            if prologue:
                prologue()
            for i in range(per_graph):
                fn(sp_g, i)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps // per_graph):
            g.replay()
        e1.record(st)
        e1.synchronize()
        return e0.elapsed_time(e1) / (reps // per_graph * per_graph) * 1e-3

    if True:   # every rank (the same untimed work everywhere: no rank waits for another in the first collective)
        v0, y0 = data_v[0], data_y[0]
        ph0, h, vk, yk, phk = ws[:5]
        ng = _lib.RbmGrads(None, ptr(G.gb), ptr(G.gd), ptr(G.gU), ptr(G.gc))
        scratchW = P.W.clone()
        # the tail of the step -- last hidden phase + gradient + update -- is ONE launch (k_hgrad) where the library takes it
        # (qbm_rbm_cd_tail refuses otherwise: then the step is k_hidden x2, k_visible, k_grad)
        tws = ops.cd_workspace(P, B)
        tcws = CdWorkspace(*(ptr(t) for t in tws))
        tP = ParamSet(scratchW, P.b.clone(), P.d.clone(), P.U.clone(), P.c.clone())
        tps = tP.struct()
        tws[0].copy_(ph0)
        tail = lambda s, i: lib.qbm_rbm_cd_tail(C.byref(tps), C.byref(ng), ptr(v0), QBM_U8, ptr(y0), B, ptr(vk), ptr(yk), C.byref(tcws), 1,
                                                lr, factor, global_B, sparse, i + 1, s)
        fused_tail = lib.qbm_rbm_cd_tail(C.byref(tps), C.byref(ng), ptr(v0), QBM_U8, ptr(y0), B, ptr(vk), ptr(yk), C.byref(tcws), 1,
                                         lr, factor, global_B, sparse, 1, cur()) == 0
        torch.cuda.synchronize()
        hid = lambda s, i: lib.qbm_rbm_sample_hidden_ex(C.byref(ps), ptr(v0), QBM_U8, ptr(y0), B, ptr(ph0), None, ptr(h), 42, 0, 0, s)
        vis = lambda s, i: lib.qbm_rbm_sample_visible_class_ex(C.byref(ps), ptr(h), QBM_U8, B, None, None, ptr(vk), None, ptr(yk), 42, 1, 2, 0, s)
        grd = lambda s, i: lib.qbm_rbm_two_phase_grad(V, H, NC, B, ptr(v0), QBM_U8, ptr(y0), ptr(ph0), ptr(vk), QBM_U8, ptr(yk), ptr(phk),
                                                      C.byref(ng), ptr(scratchW), None, lr, factor, global_B, sparse, s)
        f_hid, f_vis, f_grd = 2.0 * B * V * H + 2.0 * B * NC * H, 2.0 * B * V * H, 4.0 * B * V * H + 4.0 * B * NC * H
        # name: (launcher, algorithmic FLOPs per launch, launches per CD-1 step, prologue of the timing graph)
        if fused_tail:
            kernels = {
                "k_hidden (sample_hidden)": (hid, f_hid, 1, None),
                "k_visible (sample_visible + class conditional)": (vis, f_vis, 1, None),
                # rounds 1..50 of a timing graph on flag words re-zeroed at its start (include/qbm_hip.h: qbm_rbm_cd_tail)
                "k_hgrad (last sample_hidden + two-phase gradient + update, one launch)": (tail, f_hid + f_grd, 1, lambda: tws[5][576:1600].zero_()),
            }
        else:
            kernels = {
                "k_hidden (sample_hidden, x2 per step)": (hid, f_hid, 2, None),
                "k_visible (sample_visible + class conditional)": (vis, f_vis, 1, None),
                "k_grad (two-phase gradient + fused W update)": (grd, f_grd, 1, None),
            }
        timed = {k: time_kernel(f, prologue=pro) for k, (f, _, _, pro) in kernels.items()}

    # warm-up: at least W steps, through the launch path that is timed (whole graph replays when steps are captured)
    warm_steps = args.warmup if graphs is None else -(-args.warmup // graphs[1]) * graphs[1]
    # (nothing host-side between the per-kernel timing above, the warm-up and the timed region: a 30 ms pause there -- a
    # gc.collect() was tried -- lets the device clock down and costs a 20-step region 1.1 us per step)
    run(warm_steps)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    per_rank_ms = None
    if dp:
        # every rank's own clock around the same region (a future SCALE record shows which rank lags), then the MAX over ranks
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = dt
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(x / args.steps * 1e3, 5) for x in t.tolist()]
        dt = float(t.max().item())
    assert torch.isfinite(P.W).all(), "parameters diverged"
    assert int(ws[5][512].item()) == 0, "a launch that synchronises workgroups inside an XCD gave up during the timed run"

    ms_per_step = dt / args.steps * 1e3
    value = B * K_CD * args.steps * world / dt

    roofline = None
    if rank == 0:
        dom = max(timed, key=lambda k: timed[k] * kernels[k][2])   # most time per CD-1 step
        # HBM bytes per launch of that kernel from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate
        # rocprofv3 --pmc passes: tools/gpu_profiles.sh -> profiles/rNN_pmc_traffic.json); null if absent
        traffic, traffic_source = None, None
        try:
            import glob

            newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]   # latest round's counters
            with open(newest) as f:
                traffic = json.load(f)["kernels"][dom.split(" ")[0]]["hbm_bytes_per_launch"]
            # NOT measured in this run: PMC counters need rocprofv3 around the process (tools/gpu_profiles.sh)
            traffic_source = "imported from " + os.path.relpath(newest, ROOT) + " (rocprofv3 --pmc passes of this bench)"
        except (OSError, KeyError, ValueError, IndexError):
            pass
        fl = kernels[dom][1]
        ach = fl / timed[dom] / 1e12
        alg_bytes = 4.0 * B * V + 8.0 * (V * H + V + H + NC * H + NC) + 4.0 * B    # SURVEY.md 8.D, per CD-1 step
        roofline = {
            "bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "avg_launch_us": round(timed[dom] * 1e6, 3),
            "all_kernels_us": {k: round(t * 1e6, 3) for k, t in timed.items()},
            "step_flops": (4 * K_CD + 6) * B * V * H + (4 * K_CD + 6) * B * NC * H,
            "step_achieved_tflops": round(((4 * K_CD + 6) * B * (V + NC) * H) / (dt / args.steps) / 1e12, 3),
            "step_algorithmic_hbm_bytes": alg_bytes,
            "step_hbm_frac_of_8TBs": round(alg_bytes / (dt / args.steps) / 1e9 / PEAK_HBM_GBS, 5),
        }

    # ---- CPU baseline: the oracle (C/OpenMP restatement) on the host cores, bounded sample ----------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds)

    if rank == 0:
        out = {
            "metric": "gibbs_samples_per_sec", "value": round(value, 1), "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "per_rank_ms_per_step": per_rank_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ClassificationRBM 784v x 512h, C=10, CD-1, batch 256 per GPU (BASELINE.json configs[3] shape; "
                                   "global batch 256*N, RCCL all-reduce of the flat gradient per step when N>1)",
                       "batch_per_gpu": B, "cd_k": K_CD, "warmup_steps_run": warm_steps,
                       "pre_timed_kernel_launches": (1000 + 55 + 5) * len(timed),
                       "launches_per_step": "k_hidden, k_visible, k_hgrad" if any("k_hgrad" in k for k in timed) else "k_hidden, k_visible, k_hidden, k_grad (+ all-reduce, k_update when data-parallel)",
                       "host_calls_per_step": "1 C-ABI call (qbm_rbm_dp_step) + 1 collective" if dp else "1 C-ABI call (qbm_rbm_cd_step)",
                       "draws": "a replayed graph repeats the draw ids and minibatches it captured (same work per step)" if graphs else "fresh draw ids per step", "launch": (f"hipGraph x{graphs[1]} steps" if graphs else "eager") + (f" ({graph_note})" if graph_note else ""),
                       "parallelism": f"dp{world}" + (" (forced data-parallel path)" if args.force_dp and world == 1 else "")},
            "visible_unit_updates_per_sec": round(value * V, 1),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dp:
        dist.destroy_process_group()


def usable_cores() -> int:
    """CPU cores this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(budget_s: float):
    """The same CD-1 step (same shapes, same synthetic inputs) on the host cores, bounded sample.
    Primary: a torch-CPU restatement in the reference's idiom (oracle/rbm_torch_cpu.py: torch.matmul + elementwise ops,
    torch.rand uniforms), torch.set_num_threads(usable cores) -- SURVEY.md 8.D / BASELINE.md section 3: "the repo's own
    CPU trainer".  Secondary (`bit_exact_oracle`): the C/OpenMP oracle that the parity tests use; it pays for a fixed
    summation order and a software Philox, so it is the slower of the two."""
    import numpy as np
    import torch

    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)      # before libgomp starts: one thread per usable core
    os.environ.setdefault("OMP_WAIT_POLICY", "ACTIVE")
    from oracle import qbm_oracle as orc
    from oracle.rbm_torch_cpu import TorchCpuRbm

    orc.build()
    rs = np.random.RandomState(42)
    W = (rs.standard_normal((V, H)) * 0.1).astype(np.float32)
    b = np.full(V, 0.5, np.float32)
    d = np.zeros(H, np.float32)
    U = np.zeros((NC, H), np.float32)
    c = np.zeros(NC, np.float32)
    v0 = (orc.philox_uniform(1234, 0, 0, B, V) < 0.5).astype(np.float32)
    lab = np.minimum((orc.philox_uniform(4321, 0, 0, B, 1).reshape(-1) * NC).astype(np.int64), NC - 1)
    y0 = np.eye(NC, dtype=np.float32)[lab]

    def timed(one, budget):
        one(0)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            one(n + 1)
            n += 1
        return n, time.perf_counter() - t0

    torch.manual_seed(0)
    rbm = TorchCpuRbm(W, b, d, U, c, lr=0.05, threads=cores)
    tv0, ty0 = torch.from_numpy(v0), torch.from_numpy(y0)
    n_t, dt_t = timed(lambda i: rbm.cd1_step(tv0, ty0), 0.6 * budget_s)
    assert torch.isfinite(rbm.W).all()

    def one(i):
        ph0, vk, yk, phk = orc.rbm_cd_chain(v0, y0, W, b, d, U, c, K_CD, 42, 4 * i, 0)
        g = orc.rbm_two_phase_grad(v0, y0, ph0, vk, yk, phk)
        orc.rbm_update(W, b, d, U, c, *g, 0.05, 1.0, float(B), 0.0)

    n_o, dt_o = timed(one, 0.4 * budget_s)
    return {"value": round(B * K_CD * n_t / dt_t, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "what": "torch-CPU restatement of the CD-1 step in the reference's idiom (oracle/rbm_torch_cpu.py), "
                    f"torch.set_num_threads({cores})",
            "sample": f"{n_t} CD-1 steps of the same 784x512, B=256, C=10 workload in {dt_t:.1f} s",
            "ms_per_step": round(dt_t / n_t * 1e3, 3),
            "bit_exact_oracle": {"value": round(B * K_CD * n_o / dt_o, 1), "unit": "samples/s", "cores": cores,
                                 "sample": f"{n_o} CD-1 steps in {dt_o:.1f} s (oracle/qbm_oracle.c, OpenMP; the parity checker)",
                                 "ms_per_step": round(dt_o / n_o * 1e3, 3)}}


if __name__ == "__main__":
    main()
