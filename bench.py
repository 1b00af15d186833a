#!/usr/bin/env python
"""Throughput bench of the per-frame stylization path (BASELINE.json metric: stylized frames/sec at
512x512, 1 style; 1/2/4/8 MI355X + CPU ref).

    python bench.py --gpus N --steps K --warmup W [--size 256|512|1024] [--multistyle S]

`value` is HOST -> HOST, as SURVEY.md §8(d) defines the metric and as the reference's transfer() works
(test/framework.py:109 `.to(device)` ... :40 `.cpu()`): a step = one `rrv_transfer_batch` call over `--batch`
synthetic SxS frames per GPU (reflect-padded to P = roundup64(S+128) as generate_real_video.py:61-83 does), uint8
frames in (page-locked) host memory -> float32 BGR frames in (page-locked) host memory; the sub-batches of 16 are
pipelined inside the call (H2D / kernels / D2H on two HIP streams).  Extras, never `value`: the HBM -> HBM rate of
the same frames (`device_resident_frames_per_s`), the rate with pageable caller arrays, and the unpadded-in /
cropped-out entry.

--multistyle S (BASELINE config 5, default size 1024): the "Multi-style Interpolation" flow — encoder features of
every frame cached in HBM (test.py:87-101), statistics from every 16th + the last (:72-85), then per frame the DECODER
ONLY with the S-style blended state and the reference's weight ramp (:127-131); a step = `--batch` frames, feature in
HBM -> float32 frame in host memory.

N > 1: one process per GPU.  Started plainly (`python bench.py --gpus N`) the script launches its N ranks itself;
under torch.distributed.run it uses the ranks it is given.  Rank 0 runs prepare_style + add + compute and broadcasts
the 70 KB state blob per style (RCCL); frames are sharded, no per-frame communication ("weak" scaling: every rank
stylizes steps x batch frames).  Prints ONE JSON line (rank 0).
"""
import argparse
import importlib
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0
EPI_BITS = {"E_RELU": 1, "E_LRELU": 2, "E_NORM1": 4, "E_RES": 8, "E_RES_UPS": 16, "E_NORM2": 32, "E_POOL": 64}
HBM_KERNELS = ("conv_first", "conv_last")      # the two thin ends of the path are HBM-bound, everything else MFMA-bound


def rocprof_name(kernel):
    """bench kernel label -> the demangled name rocprofv3 reports, e.g.
    'conv_wino<E_RELU | E_POOL>' -> 'void conv_wino_split_k<65, 0>(ConvP)'."""
    if "<" not in kernel:
        return kernel + "_k"
    base, args = kernel.split("<", 1)
    args = args.rstrip(">")
    def epi(txt):
        txt = txt.strip()
        return str(sum(EPI_BITS[t.strip()] for t in txt.split("|"))) if txt[:2] == "E_" else txt
    if base == "conv_wino":      # the row-split 8-wave kernel, <EPI, ABL>
        return "void conv_wino_split_k<%s, 0>(ConvP)" % epi(args)
    if base == "conv_f43":       # F(4x4,3x3), <EPI[, LAY]>: LAY = tensor layouts (conv_f43.h; 0 = NHWC in and out)
        e, _, lay = args.partition(",")
        return "void conv_f43_k<%s, %s>(ConvP)" % (epi(e), lay.strip() or "0")
    if base in ("conv_upw", "conv_upw_sc"):       # conv_wino_k<EPI, ABL, waves, UPS, SC>: rerevst_hip.hip UPW_NW
        return "void conv_wino_k<%s, 0, 4, 1, %d, 0>(ConvP)" % (epi(args), 1 if base.endswith("_sc") else 0)      # last: PERIMG
    if base == "conv_mfma":
        bn, taps, e = args.split(",", 2)
        return "void conv_mfma_k<%s, %s, %s, 0, 0, 1>(ConvP)" % (bn.strip(), taps.strip(), epi(e))
    return kernel


def traffic_file(size, multistyle=0):
    """The committed PMC pass of this configuration (tools/profile_round.sh -> profiles/hbm_traffic_<cfg>.json)."""
    cfg = ("ms%d" % multistyle) if multistyle else str(size)
    return os.path.join(ROOT, "profiles", "hbm_traffic_%s.json" % cfg)


def measured_traffic(kernel, size=512, multistyle=0):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same
    configuration (PMC counters cannot be collected inside the timed run), or None when no pass is committed."""
    try:
        with open(traffic_file(size, multistyle)) as f:
            t = json.load(f)
        name = rocprof_name(kernel)
        # the per-image-state instantiation (last template argument 1) is the same kernel for this purpose: config 5's
        # grouped launches run it, everything else the shared-state one
        for cand in (name, name.replace(", 0>(ConvP)", ", 1>(ConvP)")):
            for k, v in t["kernels"].items():
                if k == cand or k.startswith(cand + "("):
                    return v["bytes_per_launch"]
        return None
    except Exception:
        return None


def pin_to_gpu_numa_node(local):
    """Pin this rank's host threads to the NUMA node of its GPU (PCI bus id -> /sys/bus/pci/devices/*/numa_node): the
    staging copies and the launch thread stay next to the GPU's root complex.  Returns a short description."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read().strip())
        if node < 0:
            return "gpu %s: no NUMA node reported" % bdf
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            txt = f.read().strip()
        cpus = set()
        for part in txt.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return "gpu %s: node %d has no allowed cpus" % (bdf, node)
        os.sched_setaffinity(0, cpus)
        return "gpu %s -> NUMA node %d (%d cpus)" % (bdf, node, len(cpus))
    except Exception as e:       # containers without sysfs PCI topology: stay unpinned
        return "unpinned (%s)" % (type(e).__name__,)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) and wait."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live and not rc:          # a rank that dies must not leave the others waiting in a collective
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                rc = rc or r
    if rc:
        for p in live:
            p.kill()
        for p in live:
            p.wait()
    raise SystemExit(rc)


def roofline_and_kernels(rows, nprof, frames_per_step, size, multistyle=0):
    """Per-kernel aggregation of the live HIP-event rows of the profiled steps and the roofline of the dominant one.
    `frac` = EXECUTED FLOPs / event time / fp32-MFMA peak (what the matrix pipe actually issues); the reference's
    direct-convolution FLOPs over the same time is `algorithmic_tflops` (the transform-domain kernels need 2.25x /
    4x fewer multiplies, so it may exceed the peak) and their ratio `algorithmic_speedup`."""
    agg, layers = {}, {}
    for full, ms, fl, by, fx in rows:
        name = full.split("@")[0]
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += fl; a[3] += by; a[4] += fx
        if "@" in full:
            l = layers.setdefault(full, [0, 0.0, 0.0, 0.0])
            l[0] += 1; l[1] += ms; l[2] += fl; l[3] += fx
    if not agg:
        return None, [], {}
    kern = []
    tot_ms = sum(a[1] for a in agg.values())
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        hbm = name.startswith(HBM_KERNELS) or a[2] == 0      # no matrix work (statistics, pointwise, partial sums): bandwidth kernels
        k = {"kernel": name, "launches_per_step": a[0] / nprof, "ms_per_frame": round(a[1] / nprof / frames_per_step, 4),
             "bound": "hbm" if hbm else "mfma"}
        if a[1] > 0 and a[3] > 0:
            k["gbs"] = round(a[3] / a[1] / 1e6, 1)
            k["algorithmic_bytes_per_launch"] = round(a[3] / a[0])      # input + output (+ residual, weights), each once (tests/test_abi_and_host.py checks the PMC read bytes against it)
        if a[1] > 0 and a[2] > 0 and not hbm:
            k["tflops_executed"] = round(a[4] / a[1] / 1e9, 2)
            k["frac_of_mfma_peak"] = round(a[4] / a[1] / 1e9 / PEAK_F32_MFMA_TFLOPS, 4)
            k["tflops_algorithmic"] = round(a[2] / a[1] / 1e9, 2)
        if a[1] > 0 and hbm and a[3] > 0:
            k["frac_of_hbm_peak"] = round(a[3] / a[1] / 1e6 / PEAK_HBM_GBS, 4)
        kern.append(k)
    dom = max(agg.items(), key=lambda kv: kv[1][1])
    n, t_ms, fl, by, fx = dom[1]
    executed = fx / t_ms / 1e9
    mf = [a for nm, a in agg.items() if nm.startswith(("conv_mfma", "conv_wino", "conv_upw", "conv_f43"))]
    roof = {"bound": "mfma", "kernel": dom[0], "achieved": round(executed, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(executed / PEAK_F32_MFMA_TFLOPS, 4), "frac_basis": "executed",
            "frac_executed": round(executed / PEAK_F32_MFMA_TFLOPS, 4),
            "frac_algorithmic": round(fl / t_ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4),
            "traffic": measured_traffic(dom[0], size, multistyle),
            "traffic_source": "profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this configuration; "
                              "bytes per launch = (F*FETCH_SIZE + WRITE_SIZE)*1024; FETCH_SIZE counts 64 B per L2 request whatever its "
                              "width, so F is calibrated per access pattern (profiles/r05_fetch_calib.txt): 2 for >= 128-byte contiguous "
                              "reads (conv_first_k, conv_last_k, streaming kernels), 1 for the 64-byte LDS-DMA rows of the 16-channel-chunk "
                              "kernels (conv_wino_k, conv_wino_split_k, conv_mfma_k), 1 for conv_f43_k (its chunk-major 32-byte pieces "
                              "calibrate to 0.64-0.93 by channel count: an upper bound); `fetch_factor` in every entry of the file)"
                              % os.path.basename(traffic_file(size, multistyle)),
            "algorithmic_bytes_per_launch": round(by / n),
            "algorithmic_tflops": round(fl / t_ms / 1e9, 2), "algorithmic_speedup": round(fl / fx, 3),
            "note": "achieved / frac / frac_executed = FLOPs the kernel EXECUTES / HIP-event time / fp32-MFMA peak (what the matrix "
                    "pipe issues); frac_algorithmic / algorithmic_tflops = FLOPs of the reference's direct 3x3 convolution "
                    "(SURVEY 8(d)) over the same time — above 1 because the transform-domain kernels multiply 2.25x-4x less, not "
                    "because work is skipped",
            "avg_launch_ms": round(t_ms / n, 5), "share_of_gpu_time": round(t_ms / tot_ms, 3),
            "all_matrix_kernels_executed_tflops": round(sum(a[4] for a in mf) / sum(a[1] for a in mf) / 1e9, 2),
            "all_matrix_kernels_executed_frac": round(sum(a[4] for a in mf) / sum(a[1] for a in mf) / 1e9 / PEAK_F32_MFMA_TFLOPS, 4),
            "all_matrix_kernels_algorithmic_tflops": round(sum(a[2] for a in mf) / sum(a[1] for a in mf) / 1e9, 2)}
    return roof, kern, layers


def c_abi_broadcast_check(model, dist, rank, world, n_styles, timeout_s=90.0):
    """rrv_comm_unique_id (rank 0) -> the 128 bytes over the process group -> rrv_comm_init_rank -> rrv_broadcast_state
    per style -> every rank compares its state before / after bit for bit.  Runs in a watchdog thread: a communicator
    that never forms must not hang the bench."""
    import threading
    res = {}

    def work():
        try:
            before = [model.get_state(k).copy() for k in range(n_styles)]
            ids = [model.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            comm = model.comm_init_rank(ids[0], world, rank)
            for k in range(n_styles):
                model.broadcast_state(comm, 0, rank, k)
            same = all(np.array_equal(model.get_state(k), before[k]) for k in range(n_styles))
            model.comm_destroy(comm)
            res["v"] = "ok: %d style blob(s) over ncclBroadcast, %d rank(s), state %s" % (n_styles, world, "bit-identical" if same else "DIFFERS")
        except Exception as e:
            res["v"] = "error: %s" % (str(e)[:200],)

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    return res.get("v", "timeout after %.0f s" % timeout_s)


def cpu_baseline(setup, unit, what, budget_s=25.0):
    """The CPU port (oracle/rerevst_oracle.py with its convolutions on torch's CPU conv2d = the reference's own
    primitive), timed on this box's host cores on a bounded sample: median of 5 frames with all cores after one
    warm-up, then a 1-core run (3 frames, or 2 when they are slow).  setup(O) -> (make_input(k), run(x))."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rerevst_oracle as O      # checker / timed CPU port only
    O.set_conv_backend("torch")
    make_input, run = setup(O)
    ncores = os.cpu_count() or 1
    res = {}
    for label, nthreads, nmin in (("all", min(ncores, 64), 5), ("one", 1, 3)):
        torch.set_num_threads(nthreads)
        ctx = None
        try:
            from threadpoolctl import threadpool_limits
            ctx = threadpool_limits(limits=nthreads)
        except Exception:
            pass
        if label == "all":
            run(make_input(0))                        # warm-up
        ts, t_tot, k = [], 0.0, 1
        while len(ts) < nmin:
            x = make_input(k)
            t0 = time.perf_counter()
            run(x)
            ts.append(time.perf_counter() - t0)
            t_tot += ts[-1]
            k += 1
            if label == "one" and t_tot > budget_s and len(ts) >= 2:
                break
        if ctx is not None:
            ctx.restore_original_limits()
        res[label] = (statistics.median(ts), len(ts), nthreads)
    torch.set_num_threads(min(ncores, 64))
    O.set_conv_backend("numpy")
    med, n, thr = res["all"]
    med1, n1, _ = res["one"]
    return {"value": round(1.0 / med, 4), "unit": unit, "cores": thr, "kind": "port",
            "sample": "median of %d %s through oracle/rerevst_oracle.py (fp32, convolutions on torch CPU conv2d = the reference's "
                      "primitive; %d threads of %d host cores) after 1 warm-up" % (n, what, thr, ncores),
            "one_core_value": round(1.0 / med1, 4), "one_core_sample": "median of %d, 1 thread" % n1}


def sub_leg(extra_args, timeout_s=600):
    """One of the other BASELINE configurations as a short run of this script in a child process (after the timed headline
    region, same GPU): returns the few fields the headline line carries as extras, or {"error": ...}."""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--no-extras"] + extra_args
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "RRV_BENCH_FORCE_DIST")}
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        roof = d.get("roofline") or {}
        return {"frames_per_s": d["value"], "steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"],
                "workload": d["config"]["workload"], "frames_per_step": d["config"]["frames_per_step_per_gpu"],
                "dominant_kernel": roof.get("kernel"), "frac_executed": roof.get("frac_executed"), "frac_algorithmic": roof.get("frac_algorithmic"),
                "all_matrix_kernels_executed_frac": roof.get("all_matrix_kernels_executed_frac"),
                "end_to_end_frames_per_s": d.get("end_to_end_frames_per_s"), "feature_cache_frames_per_s": d.get("feature_cache_frames_per_s")}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (default 64 at 512x512, scaled with the frame area)")
    ap.add_argument("--size", type=int, default=0, help="frame side: 256 / 512 / 1024 (default 512; 1024 with --multistyle)")
    ap.add_argument("--frames", type=int, default=0, help="frames of the synthetic video (default: BASELINE configs, 100 / 300 / 300)")
    ap.add_argument("--multistyle", type=int, default=0, help="S > 0: BASELINE config 5, S-style interpolation, decoder only per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short 256x256 (config 2) and 4-style 1024x1024 (config 5) legs appended to the default run")
    ap.add_argument("--profile-steps", type=int, default=1)
    ap.add_argument("--pipeline", type=int, default=2, help="sub-batches in flight per GPU (1 or 2 HIP streams)")
    ap.add_argument("--pageable", action="store_true", help="time the host entry with pageable caller arrays instead of page-locked ones")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RRV_BENCH_BACKEND=gloo lets the N>1 control flow be exercised on a 1-GPU box (all ranks on GPU 0)
    backend = os.environ.get("RRV_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > max(1, torch.cuda.device_count()):
        raise SystemExit("--gpus %d but only %d GPU(s) visible (one process per GPU; RRV_BENCH_BACKEND=gloo shares GPUs for a control-flow check)"
                         % (world, torch.cuda.device_count()))
    local = local % max(1, torch.cuda.device_count())
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # RRV_BENCH_FORCE_DIST=1: run the process-group code path (init, barrier, broadcast, all_reduce, gather) at world size 1
    # too, so that the RCCL branch executes on a one-GPU box (RCCL works with a single rank)
    use_dist = world > 1 or os.environ.get("RRV_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    torch.cuda.set_device(local)
    affinity = pin_to_gpu_numa_node(local) if os.environ.get("RRV_BENCH_NO_PIN") != "1" else "unpinned (RRV_BENCH_NO_PIN)"
    dev = torch.device("cuda", local)
    cdev = dev if backend == "nccl" else torch.device("cpu")     # where collective payloads live

    pkg = importlib.import_module("rerevst-code_amd")
    video = importlib.import_module("rerevst-code_amd.video")
    NS = args.multistyle
    S = args.size or (1024 if NS else 512)
    # BASELINE configs: 100 frames at 256x256, 300 at 512x512 / 1024x1024 on one GPU, 1200 frames at 512x512 over 8 GPUs
    # (150 per GPU; kept for every N > 1 so that the work per GPU — and rank 0's preparation per GPU — is the same)
    NF = args.frames or ({256: 100}.get(S, 300) if world == 1 else {256: 100}.get(S, 150) * world)
    P = video.padded_size(S)
    B = args.batch or max(8, min(128, (128 * 640 * 640) // (P * P) // 8 * 8))      # 128 frames per call at 512x512 (eight sub-batches of 16), 128 at 256x256, 32 at 1024x1024
    weights = pkg.synthetic_weights(0)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    if NS:
        model = pkg.MultiStyleStylization(weights, cuda=True, style_num=NS, device=local)
    else:
        model = pkg.Stylization(weights, cuda=True, device=local)
    model.set_pipeline(args.pipeline)
    MS_GROUP = int(os.environ.get("RRV_BENCH_MS_GROUP", "0")) or max(1, min(16, (16 * 640 * 640) // (P * P)))      # (experiment knob; default =) frames per launch sequence of rrv_transfer_features_batch (the library's rule, set explicitly: what config.sub_batch reports)
    if NS:
        model.set_multistyle_group(MS_GROUP)
    if os.environ.get("RRV_BENCH_GRID_SHARE"):       # experiment knob: every launch takes 1/n of the CUs (DESIGN 4 "One frame per call")
        model.set_grid_share(int(os.environ["RRV_BENCH_GRID_SHARE"]))

    # ---- once-per-video preparation on rank 0, state broadcast over RCCL ------------------
    lo, hi = video.shard_range(NF, rank, world)                # this rank's contiguous shard of the video
    first, nshard = lo, max(1, hi - lo)
    t0 = time.time()
    blob = torch.empty(max(1, NS) * 17536, dtype=torch.float32, device=cdev)
    feats = None
    if NS:
        # features of this rank's frames, cached in HBM (rank 0 also needs the sampled ones for the statistics)
        n_cached = min(nshard, args.steps * B + args.warmup * B)
        my_ids = [first + i % nshard for i in range(n_cached)]
        styles = [video.resize_bilinear(pkg.synth_style(512, 512, kind="noise", seed=7 + k), (384, 384)) for k in range(NS)]   # test.py:53
        if rank == 0:
            model.prepare_style(styles)
            model.clean()
            for i in video.sample_indices_multistyle(NF, 16):
                f = model.generate_content_features(video.reflect_pad(pkg.synth_frame(i, S, S, kind="noise"), P, P))   # padded BEFORE encoding (test.py:96)
                model.add_patch(f)
            model.compute_norm()
            model.release_features()
            blob.copy_(torch.from_numpy(np.concatenate([model.get_state(k) for k in range(NS)])))
        # the caching pass ("Multi-style Interpolation/test.py":87-101: every frame padded, encoded once, cached): padded frames in
        # page-locked host memory -> features in HBM through ONE rrv_generate_content_features_batch call (frame synthesis and padding
        # are not part of it; the reference reads its frames from disk)
        h_frames = pkg.pinned_empty((n_cached, P, P, 3), np.uint8)
        for k, i in enumerate(my_ids):
            h_frames[k] = video.reflect_pad(pkg.synth_frame(i, S, S, kind="noise"), P, P)
        model.release_features()
        model.generate_content_features_batch(h_frames[:min(8, n_cached)])       # warm-up: workspaces, staging buffers
        model.release_features()
        model.sync()
        t1 = time.time()
        feats = model.generate_content_features_batch(h_frames)
        model.sync()
        cache_s = time.time() - t1
        del h_frames
    else:
        if rank == 0:
            model.prepare_style(pkg.synth_style(512, 512, kind="noise", seed=7))
            model.clean()
            for i in video.sample_indices(NF):
                model.add(pkg.synth_frame(i, S, S, kind="noise"))      # unpadded (generate_real_video.py:139-143)
            model.compute()
            blob.copy_(torch.from_numpy(model.get_state()))
    prep_s = time.time() - t0
    if use_dist:
        dist.broadcast(blob, src=0)
        if rank != 0:
            b = blob.cpu().numpy()
            for k in range(max(1, NS)):
                model.set_state(b[k * 17536:(k + 1) * 17536], k)

    # ---- this rank's frames / outputs in page-locked host memory (or pageable with --pageable) -----------
    alloc = (lambda shp, dt: np.empty(shp, dt)) if args.pageable else pkg.pinned_empty
    h_out = alloc((2, B, P, P, 3), np.float32)
    if NS:
        nW = len(feats)
        def step(i):
            ks = [(i * B + j) % nW for j in range(B)]    # position in the shard; global frame index -> the reference's weight ramp
            model.transfer_many([feats[k] for k in ks], [video.ramp_weights(my_ids[k], NF, NS, blend="all") for k in ks], out=h_out[i & 1])
    else:
        n_batches = max(1, min(max(1, nshard // B), args.steps))      # distinct batches kept resident in host memory
        my_ids = [first + i % nshard for i in range(n_batches * B)]
        h_in = alloc((n_batches, B, P, P, 3), np.uint8)
        for k, i in enumerate(my_ids):
            h_in[k // B, k % B] = video.reflect_pad(pkg.synth_frame(i, S, S, kind="noise"), P, P)
        def step(i):
            model.transfer_batch(h_in[i % n_batches], out=h_out[i & 1])

    for i in range(args.warmup):
        step(i)
    model.sync()
    barrier()
    nprof = max(0, min(args.profile_steps, args.steps))
    t0 = time.perf_counter()
    for i in range(args.steps - nprof):
        step(i)
    rows = []
    if nprof:
        model.profile_begin()          # HIP events on the library's own stream, inside the timed region
        for i in range(args.steps - nprof, args.steps):
            step(i)
        rows = model.profile_end()
    model.sync()
    barrier()
    dt = time.perf_counter() - t0
    per_rank_s = [dt]
    if use_dist:
        mine = torch.tensor([dt], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                       # each rank's own wall time (a straggler shows up here)
        per_rank_s = [float(x.item()) for x in every]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the one collective of the path once more through the C ABI (rrv_broadcast_state: ncclBroadcast on an ncclComm_t
    # built from a unique id that travels over the process group), checked bit for bit against the state every rank holds
    c_abi_bcast = None
    if use_dist and backend == "nccl" and os.environ.get("RRV_BENCH_NO_CABI_BCAST") != "1":
        c_abi_bcast = c_abi_broadcast_check(model, dist, rank, world, max(1, NS))

    if rank == 0:
        roof, kern, layers = roofline_and_kernels(rows, nprof, B, S, NS)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            blobs = [model.get_state(k) for k in range(max(1, NS))]
            def frame_of(k):
                return video.reflect_pad(pkg.synth_frame(k, S, S, kind="noise"), P, P)
            if NS:
                def setup(O):
                    o = O.MultiStylization(weights, NS)
                    for k in range(NS):
                        o.per_style[k].set_state(blobs[k])
                    # the encoder pass is the caching step (outside the metric): it runs in make_input, the clock sees the decoder
                    return (lambda k: o.generate_content_features(frame_of(k))), (lambda f: o.transfer(f, video.ramp_weights(7, NF, NS, blend="all")))
                cpu = cpu_baseline(setup, "frames/s", "cached relu4_1 features of padded %dx%d frames, %d-style blended decoder" % (P, P, NS))
            else:
                def setup(O):
                    o = O.Stylization(weights)
                    o.set_state(blobs[0])
                    return frame_of, o.transfer
                cpu = cpu_baseline(setup, "frames/s", "padded %dx%d frames" % (P, P))
        what = ("%d-frame synthetic %dx%d video (padded %dx%d), %d-style interpolation (every frame blends all %d styles: smooth weight "
                "ramp; decoder only per frame, encoder features cached in HBM), frames sharded per GPU" % (NF, S, S, P, P, NS, NS)) if NS else \
               ("%d-frame synthetic %dx%d video (padded %dx%d), 1 style, frames sharded per GPU" % (NF, S, S, P, P))
        out = {"metric": "stylized frames/sec at %dx%d, %s" % (S, S, ("%d-style interpolation" % NS) if NS else "1 style"),
               "value": round(world * args.steps * B / dt, 3),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": what, "entry": ("rrv_transfer_features_batch: relu4_1 features in HBM -> float32 frames in %s host memory" if NS else
                                                      "rrv_transfer_batch: uint8 frames in %s host memory -> float32 frames in the same (H2D + kernels + D2H)")
                                               % ("pageable" if args.pageable else "page-locked"),
                          "frames_per_step_per_gpu": B, "sub_batch": MS_GROUP if NS else max(1, min(32, B, (16 * 640 * 640) // (P * P))), "batches_in_flight": args.pipeline,
                          "sampled_frames": len(video.sample_indices_multistyle(NF, 16) if NS else video.sample_indices(NF)),
                          "parallelism": "frame-shard x%d" % world},
               "ms_per_frame": round(1e3 * dt / args.steps / B, 4), "roofline": roof, "cpu_baseline": cpu,
               "prep_seconds_rank0": round(prep_s, 3), "kernels": kern}
        # diagnostics of a multi-rank run (the driver computes scaling efficiency itself from the per-N `value`s)
        rates = [args.steps * B / t for t in per_rank_s]
        out["per_rank"] = {"frames_per_s": [round(r, 1) for r in rates], "min": round(min(rates), 1), "max": round(max(rates), 1),
                           "slowest_rank": int(np.argmin(rates)), "process_group": (backend if use_dist else None),
                           "rank0_affinity": affinity, "c_abi_rccl_broadcast": c_abi_bcast}
        if NS:
            fc = len(feats) / cache_s
            out["feature_cache_frames_per_s"] = round(fc, 1)
            # a video end to end = the caching pass (encoder, once per frame) THEN the decoder pass: serial rates combine harmonically
            out["end_to_end_frames_per_s"] = round(1.0 / (1.0 / fc + 1.0 / out["value"]), 1)
        if world == 1 and not args.no_extras and not NS:
            # extras, NOT `value`.  (1) HBM -> HBM on the same frames, one sub-batch per launch, two batches in flight (round 1's headline)
            sbf = max(1, min(32, B, (16 * 640 * 640) // (P * P)))       # the library's sub-batch for this frame size
            d_in = torch.from_numpy(np.ascontiguousarray(h_in[0][:2 * sbf if B >= 2 * sbf else B])).to(dev)
            nb8 = max(1, d_in.shape[0] // sbf)
            bb = d_in.shape[0] // nb8
            d_in = d_in[:nb8 * bb].view(nb8, bb, P, P, 3)
            d_out = torch.empty((4, bb, P, P, 3), dtype=torch.float32, device=dev)
            nrep = max(8, min(60, args.steps * B // bb))
            torch.cuda.synchronize()
            for i in range(2):
                model.transfer_batch_device(d_in[i % nb8].data_ptr(), bb, P, P, d_out[i & 3].data_ptr())
            model.sync()
            t1 = time.perf_counter()
            for i in range(nrep):
                model.transfer_batch_device(d_in[i % nb8].data_ptr(), bb, P, P, d_out[i & 3].data_ptr())
            model.sync()
            out["device_resident_frames_per_s"] = round(nrep * bb / (time.perf_counter() - t1), 1)
            # (2) the driver-level entry: UNPADDED frames in, cropped frames out (pad / crop inside the first / last kernel,
            # full-resolution layers compute only the tiles the crop window needs): device-resident and host -> host
            raw = torch.from_numpy(np.stack([pkg.synth_frame(i, S, S, kind="noise") for i in my_ids[:2 * bb]])).to(dev).view(2, bb, S, S, 3)
            d_crop = torch.empty((4, bb, S, S, 3), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            for i in range(2):
                model.transfer_frames_device(raw[i % 2].data_ptr(), bb, S, S, d_crop[i & 3].data_ptr())
            model.sync()
            t1 = time.perf_counter()
            for i in range(nrep):
                model.transfer_frames_device(raw[i % 2].data_ptr(), bb, S, S, d_crop[i & 3].data_ptr())
            model.sync()
            out["cropped_entry_device_frames_per_s"] = round(nrep * bb / (time.perf_counter() - t1), 1)
            h_raw = pkg.pinned_empty((B, S, S, 3), np.uint8)
            for k in range(B):
                h_raw[k] = pkg.synth_frame(my_ids[k % len(my_ids)], S, S, kind="noise")
            h_crop = pkg.pinned_empty((B, S, S, 3), np.float32)
            model.transfer_frames(h_raw, out=h_crop)
            nr2 = max(2, min(6, args.steps))
            t1 = time.perf_counter()
            for i in range(nr2):
                model.transfer_frames(h_raw, out=h_crop)
            out["cropped_entry_host_frames_per_s"] = round(nr2 * B / (time.perf_counter() - t1), 1)
            # (3) the other kind of caller memory, and the reference's one-frame-per-call surface
            other = (lambda shp, dt: np.empty(shp, dt)) if not args.pageable else pkg.pinned_empty
            o_in = other((B, P, P, 3), np.uint8); o_in[...] = h_in[0]
            o_out = other((B, P, P, 3), np.float32)
            model.transfer_batch(o_in, out=o_out)
            t1 = time.perf_counter()
            for i in range(nr2):
                model.transfer_batch(o_in, out=o_out)
            out["%s_host_frames_per_s" % ("page_locked" if args.pageable else "pageable")] = round(nr2 * B / (time.perf_counter() - t1), 1)
            # the reference's own call surface: ONE frame per call (generate_real_video.py:164), pageable numpy arrays in
            # and out.  (a) plain transfer(frame) as the reference loop is written; (b) the same loop with the look-ahead
            # form transfer_async / result (submit frame i+1, then collect frame i)
            n1 = min(len(h_in[0]), 48)
            one = [np.array(h_in[0][k]) for k in range(n1)]
            for k in range(2):
                model.transfer(one[k])
            t1 = time.perf_counter()
            for k in range(n1):
                model.transfer(one[k])
            out["one_frame_per_call_frames_per_s"] = round(n1 / (time.perf_counter() - t1), 1)
            def lookahead(depth):
                q = []
                t1 = time.perf_counter()
                for k in range(n1):
                    q.append(model.transfer_async(one[k]))
                    if len(q) > depth:
                        model.result(q.pop(0))
                while q:
                    model.result(q.pop(0))
                return round(n1 / (time.perf_counter() - t1), 1)
            lookahead(3)
            out["one_frame_per_call_lookahead_frames_per_s"] = lookahead(3)       # three frames submitted ahead of the one collected
            out["one_frame_per_call_lookahead1_frames_per_s"] = lookahead(1)
            # (4) zero-copy host I/O (rrv_set_host_io(1)): the first kernel reads the page-locked frames over PCIe, the last one
            # writes the stylized frames there — no copy kernels, no copy-stream events.  Same frames, same entries.
            model.set_host_io(1)
            model.transfer_batch(h_in[0], out=h_out[0])
            t1 = time.perf_counter()
            for i in range(nr2):
                model.transfer_batch(h_in[i % n_batches], out=h_out[i & 1])
            out["zero_copy_host_frames_per_s"] = round(nr2 * B / (time.perf_counter() - t1), 1)
            t1 = time.perf_counter()
            for k in range(n1):
                model.transfer(one[k])
            out["zero_copy_one_frame_per_call_frames_per_s"] = round(n1 / (time.perf_counter() - t1), 1)
            for mode, key in ((2, "zero_copy_input_only_frames_per_s"), (3, "zero_copy_output_only_frames_per_s")):
                model.set_host_io(mode)
                model.transfer_batch(h_in[0], out=h_out[0])
                t1 = time.perf_counter()
                for i in range(nr2):
                    model.transfer_batch(h_in[i % n_batches], out=h_out[i & 1])
                out[key] = round(nr2 * B / (time.perf_counter() - t1), 1)
            model.set_host_io(0)
        # (5) SURVEY 8(f)4, the driver's file -> file rate: PNG frames on disk -> decode (worker threads) -> transfer_frames ->
        # encode / write (worker threads), the reference script's whole loop (generate_real_video.py:152-171) without its
        # one-off preparation.  Smooth synthetic frames (white noise is the codecs' worst case, not video).
        if world == 1 and not args.no_extras and not NS and os.environ.get("RRV_BENCH_NO_DRIVER") != "1":
            try:
                import shutil
                import tempfile
                drv = importlib.import_module("rerevst-code_amd.driver")
                base = tempfile.mkdtemp(prefix="rrv_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
                try:
                    nfr = 192 if S <= 512 else 48
                    os.makedirs(os.path.join(base, "in"))
                    paths = [os.path.join(base, "in", "f%04d.png" % i) for i in range(nfr)]
                    from concurrent.futures import ThreadPoolExecutor
                    with ThreadPoolExecutor(drv.default_io_threads()) as ex:
                        list(ex.map(lambda a: drv.write_image_bgr(a[0], pkg.synth_frame(a[1], S, S, kind="smooth")), zip(paths, range(nfr))))
                    drv.write_image_bgr(os.path.join(base, "style.png"), pkg.synth_style(512, 512, kind="smooth", seed=7))
                    st = {}
                    drv.stylize_files(model, os.path.join(base, "style.png"), paths, os.path.join(base, "out"), chunk=32, log=lambda *_: None, stats=st)
                    out["driver_png_to_png_frames_per_s"] = round(st["frames_per_s"], 1)
                    out["driver_png_to_png"] = {"frames": nfr, "io_threads": st["io_threads"], "host_cores": os.cpu_count(), "chunk": 32,
                                                "gpu_call_seconds": round(st["gpu_call_s"], 3), "stage_seconds": round(st["frames_s"], 3),
                                                "preparation_seconds": round(st["prep_s"], 3),
                                                "what": "%dx%d PNG files (tmpfs) -> stylized PNG files, decode / transfer_frames / encode overlapped" % (S, S)}
                finally:
                    shutil.rmtree(base, ignore_errors=True)
            except Exception as e:
                out["driver_png_to_png_frames_per_s"] = None
                out["driver_png_to_png"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        # BASELINE configs 2 and 5 as short legs of the SAME default invocation (the driver runs only this one), after the timed
        # headline region: extras, never `value`.  Each is this script in a child process with its own steps / warm-up.
        if world == 1 and not args.no_extras and not args.no_other_configs and not NS and S == 512 and not args.pageable:
            model.close()
            c2 = sub_leg(["--size", "256", "--steps", "12", "--warmup", "3"])
            c5 = sub_leg(["--multistyle", "4", "--steps", "16", "--warmup", "3"])
            out["config2"] = c2
            out["config5"] = c5
            out["config2_frames_per_s"] = c2.get("frames_per_s")
            out["config5_frames_per_s"] = c5.get("frames_per_s")
        if os.environ.get("RRV_BENCH_LAYERS"):
            out["layers"] = [{"layer": k, "ms_per_frame": round(v[1] / nprof / B, 4), "tflops": round(v[2] / v[1] / 1e9, 1),
                              "tflops_executed": round(v[3] / v[1] / 1e9, 1)}
                             for k, v in sorted(layers.items(), key=lambda kv: -kv[1][1])]
        print(json.dumps(out), flush=True)
    model.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
