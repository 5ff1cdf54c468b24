#!/usr/bin/env python3
"""bench.py -- images/sec of the ViT-B/16 forward pass on N MI355X GPUs.

A "step" is one pass of the hot path (patch-embed -> 12 encoder layers -> final
LayerNorm -> classifier -> softmax) over one batch of 512 synthetic 224x224x3
images per GPU, already resident in HBM when the timed region starts; with N > 1
each step ends with the RCCL gather of the class logits to rank 0 (the only
collective on the path).  Batches shard by image, weights are replicated, so
per-GPU work is fixed as N grows ("weak").

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fc1
GEMM: gemm_p3_kernel<8,256,EPI_GELU,OUT_P3> plus its 128x128-tile launch for the last
partial scheduling round, csrc/gemm_p3.hip): algorithmic FLOPs per launch divided by
its mean launch duration, measured with HIP events recorded on the launch stream
inside the timed region.  `cpu_baseline` is the reference's own ViT_seq.c
(oracle/_ref, built in the build container) or, failing that, the oracle port,
timed on this box's host cores on a bounded sample: first one image on one thread
(the reference's own execution model, ViT_seq.c:433), then one image per usable
core; the images are chosen to straddle the GPU path's tile and launch boundaries
and their logits give the parity of the GPU result in the same run.
`end_to_end` is the host-pointer entry vit_hip_forward (separately allocated host
images in, probabilities out, PCIe included) -- reported beside `value`, never as it.
At N = 1 and the default dtype the line also carries secondary legs, each beside and
never instead of `value`: the opt-in reduced-precision modes (`bf16_gemm_mode`,
`fp8_block_scaled_gemm_mode`, `fp32_fp16x2_emulation_mode`, each with its own
`roofline` block) and `class_token_rows_only_last_layer` (the opt-in that evaluates
the last layer's projection and MLP on the rows the classifier reads; logits checked
bit for bit against the full evaluation in the same run).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0          # same guide, "HBM3E peak BW" (spec)


def model_flops(cfg, tokens: int) -> dict[str, float]:
    """Algorithmic FLOPs (2*MAC) per image, per operator class."""
    E, F, H, L = cfg.embed_dim, cfg.mlp_hidden, cfg.num_heads, cfg.depth
    D, T, NP = E // H, tokens, tokens - 1
    K0 = cfg.in_chans * cfg.patch_size ** 2
    return {
        "patch_embed": 2.0 * NP * K0 * E,
        "qkv_gemm": L * 2.0 * T * E * 3 * E,
        "attention": L * 2.0 * 2 * H * T * T * D,
        "out_proj_gemm": L * 2.0 * T * E * E,
        "fc1_gemm": L * 2.0 * T * E * F,
        "fc2_gemm": L * 2.0 * T * F * E,
        "head_gemm": 2.0 * E * cfg.num_classes,
    }


# Images whose CPU logits are compared with the GPU's: first / last of the batch, both sides of a
# 256-row tile edge inside an image pair (rows 255|256 belong to image 1), and both sides of the
# hand-over from the 256x256-tile launch to the 128x128-tile launch (row 98 304 = image 499 at batch 512).
PARITY_IMAGES = [0, 1, 255, 498, 499, 511]


def host_cpu_info() -> dict:
    """What this process may use of the box: logical CPUs, affinity, cgroup quota, model name."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    if quota:
        usable = min(usable, quota)
    return {"cpu_model": model, "nproc": os.cpu_count() or 1, "usable_cores": usable}


def cpu_baseline(batch: int, n_procs: int):
    """Time the CPU path on this box.  Returns (dict for the JSON line, {image index: logits})."""
    harness = ROOT / "oracle" / "_ref" / "ref_harness"
    info = host_cpu_info()
    # one image per usable core, at most 32 at once: a bounded sample (~6 s of wall, < 12 GB of host memory) on any node
    n_procs = n_procs or min(info["usable_cores"], 32)
    # the parity images first, then as many more images as there are cores to fill
    want = [i for i in PARITY_IMAGES if i < batch]
    images = (want + [i for i in range(batch) if i not in want])[:max(n_procs, 1) + 1]   # 1 alone, then n_procs at once
    with tempfile.TemporaryDirectory() as td:
        if harness.exists() and os.access(harness, os.X_OK):
            from oracle.oracle import read_records

            def run(indices):
                t0 = time.perf_counter()
                procs = [subprocess.Popen([str(harness), "full", str(i), "1", "0", str(Path(td) / f"img{i}.bin")],
                                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i in indices]
                codes = [p.wait(timeout=900) for p in procs]      # reap every child before judging any
                return all(c == 0 for c in codes), time.perf_counter() - t0

            ok1, dt1 = run(images[:1])                 # one thread, one image: the reference's own model
            okn, dtn = run(images[1:]) if len(images) > 1 else (True, 0.0)
            if ok1 and okn:
                logits = {i: read_records(Path(td) / f"img{i}.bin")["logits"] for i in images}
                n_all = max(len(images) - 1, 1)
                return ({"value": (n_all / dtn) if len(images) > 1 else 1.0 / dt1, "unit": "images/sec",
                         "cores": n_all if len(images) > 1 else 1, "kind": "reference",
                         "single_thread": {"value": 1.0 / dt1, "unit": "images/sec", "cores": 1,
                                           "seconds_per_image": round(dt1, 2)},
                         **info,
                         "sample": f"the reference's own ViT_seq.c (oracle/_ref): 1 synthetic image on one thread "
                                   f"({dt1:.1f} s), then {n_all} images, one per process, at once ({dtn:.1f} s wall)"},
                        logits)
        # fall back to the oracle port (still only a baseline / checker)
        from oracle.oracle import Oracle
        orc = Oracle("vit_b_16")
        w = orc.synth_weights(0)
        t0 = time.perf_counter()
        logits0, _, _ = orc.forward(orc.synth_image(0), w)
        dt = time.perf_counter() - t0
        return ({"value": 1.0 / dt, "unit": "images/sec", "cores": 1, "kind": "port", **info,
                 "sample": f"1 synthetic image through oracle/vit_seq_port.c on one thread, {dt:.1f} s wall"},
                {0: logits0})


def committed_traffic(root: Path):
    """(bytes per fc1 launch, where from, stale?) from the newest profiles/rNN_pmc_traffic.json -- withheld (None, ..., True)
    when csrc/gemm_p3.hip is no longer the file the PMC passes were taken on (sha256 stored by tools/pmc_traffic.py)."""
    import hashlib
    pmcs = sorted((root / "profiles").glob("r*_pmc_traffic.json"))
    if not pmcs:
        return None, None, None
    rec = json.loads(pmcs[-1].read_text())
    sha_now = hashlib.sha256((root / "vit-with-opencl_amd" / "csrc" / "gemm_p3.hip").read_bytes()).hexdigest()
    stale = rec.get("kernel_source_sha256") != sha_now
    src = f"profiles/{pmcs[-1].name} (rocprofv3 --pmc passes, B=512; kernel source sha256 " \
          f"{str(rec.get('kernel_source_sha256'))[:16]}, git {str(rec.get('git_head'))[:12]})"
    return (None if stale else rec["traffic_bytes_per_launch"]), src, stale


FC1_KERNELS = ("gemm_p3_kernel<8, 256, 1, 1, 3>", "gemm_p3_kernel<4, 128, 1, 1, 3>")   # EPI_GELU, OUT_PLANES, three parts: fc1 only


def pmc_child(B: int) -> None:
    """Child-process leg run UNDER rocprofv3 by measure_fc1_traffic: two forwards of the default fp32 path at batch B (the
    counters do not depend on the pixel values: the image buffer is zero-filled).  No torch in this process."""
    import __graft_entry__ as graft
    pkg = graft.load_package()
    L = pkg.lib()
    cfg = pkg.preset("vit_b_16")
    model = pkg.ViTHip(cfg, pkg.synth_weights(cfg, 0), device=0, max_batch=B)
    d_images = pkg.DeviceBuffer(B * cfg.in_chans * cfg.img_size * cfg.img_size)
    pkg.binding.check(L.vh_memset(d_images.ptr, 0, B * cfg.in_chans * cfg.img_size * cfg.img_size * 4, None), "vh_memset")
    d_logits = pkg.DeviceBuffer(B * cfg.num_classes)
    for _ in range(2):
        model.forward_device(d_images.ptr, B, d_logits.ptr, None, model.stream)
    pkg.binding.check(L.vh_device_sync(), "sync")
    model.close()


def measure_fc1_traffic(B: int):
    """HBM-side bytes per fc1 launch, measured in THIS run: two separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE --
    they do not fit one pass, MI355X_MICROARCH.md) over a child process that runs the same forward, per-dispatch means
    of the fc1 kernels (the 256x256-tile launch + its 128x128-tile launch) added, corrected as the guide prescribes:
    counters are KiB, and on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads.  -> (bytes | None, note)"""
    import csv
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VIT_HIP_NO_TORCH="1", TMPDIR="/tmp")
    means = {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = Path(td) / counter
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", str(out), "--", sys.executable,
                   str(Path(__file__).resolve()), "--pmc-child", "--batch", str(B)]
            try:
                r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=150)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter}: timed out"
            files = list(out.rglob("*counter_collection.csv"))
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter}: exit {r.returncode}: {r.stderr[-200:]}"
            per_kernel = {}
            with open(files[0], newline="") as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] != counter:
                        continue
                    name = row["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
                    if name in FC1_KERNELS:
                        per_kernel.setdefault(name, []).append(float(row["Counter_Value"]))
            if FC1_KERNELS[0] not in per_kernel:
                return None, f"no fc1 dispatch in the {counter} pass"
            means[counter] = sum(sum(v) / len(v) for v in per_kernel.values())      # big launch + tail launch, mean per dispatch each
    traffic = (2.0 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024.0
    return traffic, (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes, child process, "
                     f"{time.perf_counter() - t0:.0f} s), per-dispatch means of {' + '.join(FC1_KERNELS)}; "
                     f"traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of wide reads; L2 memory side, "
                     f"Infinity-Cache hits included); FETCH_SIZE {means['FETCH_SIZE']:.0f} KiB, WRITE_SIZE {means['WRITE_SIZE']:.0f} KiB")


def spawn_ranks(n: int) -> int:
    """Start `n` ranks of this script under torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) as a child
    process and wait for it.  Called before anything in this process has initialised the GPU; nothing is re-exec'd."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def in_library_multi_child(n_dev: int, first_dev: int, B: int, steps: int) -> None:
    """Child-process leg: the library's own multi-GPU entry (vit_hip_create_multi + vit_hip_forward_device_multi: one
    replica per device inside ONE process, shards resident in HBM, the classifier gather over RCCL in C) on devices
    first_dev .. first_dev + n_dev - 1.  Prints one JSON object.  No torch in this process: the library binds the system
    HIP runtime and opens the system librccl itself."""
    import ctypes as C
    import __graft_entry__ as graft
    pkg = graft.load_package()
    L = pkg.lib()
    cfg = pkg.preset("vit_b_16")
    NC, per = cfg.num_classes, cfg.in_chans * cfg.img_size * cfg.img_size
    devices = list(range(first_dev, first_dev + n_dev))
    weights = pkg.synth_weights(cfg, 0)
    m = pkg.ViTHipMulti(cfg, weights, devices, max_batch_per_device=B)
    d_imgs = []
    for g, dev in enumerate(devices):          # shard g = global images g*B .. g*B+B-1, resident on device g
        pkg.binding.check(L.vh_set_device(dev), "vh_set_device")
        buf = pkg.DeviceBuffer(B * per)
        for lo in range(0, B, 64):
            k = min(64, B - lo)
            chunk = pkg.synth_images(cfg, g * B + lo, k)
            pkg.binding.check(L.vh_h2d(buf.ptr.value + lo * per * 4, chunk.ctypes.data, k * per * 4, None), "vh_h2d")
            pkg.binding.check(L.vh_device_sync(), "sync")
        d_imgs.append(buf)
    pkg.binding.check(L.vh_set_device(devices[0]), "vh_set_device")
    d_logits, d_probs = pkg.DeviceBuffer(n_dev * B * NC), pkg.DeviceBuffer(n_dev * B * NC)
    counts = [B] * n_dev
    ptrs = [b.ptr for b in d_imgs]
    for _ in range(2):
        m.forward_device(ptrs, counts, d_logits.ptr, d_probs.ptr)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.forward_device(ptrs, counts, d_logits.ptr, d_probs.ptr)     # synchronous on return
    dt = (time.perf_counter() - t0) / steps
    enqueue_ms = m.last_enqueue_ms()          # per device: host time its thread took to enqueue its shard (last step)
    pkg.binding.check(L.vh_set_device(devices[0]), "vh_set_device")
    got = d_logits.to_numpy((n_dev, B, NC))
    probs = d_probs.to_numpy((n_dev * B, NC))
    # every other shard's first and last 4 images recomputed by the root's replica on the same batch positions
    edge = sorted(set(list(range(min(4, B))) + list(range(max(B - 4, 0), B))))
    one = pkg.ViTHip(cfg, weights, device=devices[0], max_batch=B)
    d_chk = pkg.DeviceBuffer(B * NC)
    verified = 0
    for g in range(1, n_dev):
        for i in edge:
            img = pkg.synth_images(cfg, g * B + i, 1)
            pkg.binding.check(L.vh_h2d(d_imgs[0].ptr.value + i * per * 4, img.ctypes.data, per * 4, None), "vh_h2d")
        pkg.binding.check(L.vh_device_sync(), "sync")
        one.forward_device(d_imgs[0].ptr, B, d_chk.ptr, None, one.stream)
        pkg.binding.check(L.vh_device_sync(), "sync")
        if not np.array_equal(d_chk.to_numpy((B, NC))[edge], got[g][edge]):
            raise SystemExit(f"in-library multi-GPU leg: rows gathered from device {devices[g]} differ from the root's recomputation")
        verified += 1
    one.close()
    m.close()
    print(json.dumps({"what": "vit_hip_create_multi + vit_hip_forward_device_multi: one process, one replica, one stream and one "
                              "enqueuing host thread per device, image shards resident in HBM, logits gathered on the first "
                              "device by grouped ncclSend/ncclRecv in C (librccl opened at run time), softmax of all rows on "
                              "the root; synchronous per step", "devices": devices, "batch_per_device": B, "steps": steps,
                      "value": round(n_dev * B / dt, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 3),
                      "host_enqueue_ms": enqueue_ms,
                      "gathered_shards_verified_bitwise": verified,
                      "gather_verified": (verified == n_dev - 1) if n_dev > 1 else None,   # None: one device, nothing was exchanged
                      "logits_finite": bool(np.isfinite(got).all()),
                      "prob_sum_last_image": float(probs[-1].sum())}), flush=True)
    if n_dev > 1 and verified != n_dev - 1:
        raise SystemExit(f"in-library multi-GPU leg: {verified} of {n_dev - 1} gathered shards verified")


def run_in_library_multi(n_dev: int, first_dev: int, B: int) -> dict:
    """Run the leg above in a child process (bounded by a timeout; a failure there cannot take this run down)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                              "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
                                                              "TORCHELASTIC_RUN_ID", "VIT_DIST_FORCE", "VIT_DIST_BACKEND")}
    env["VIT_HIP_NO_TORCH"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, str(Path(__file__).resolve()), "--in-library-multi-child", str(n_dev), "--first-device", str(first_dev),
           "--batch", str(B), "--steps", "5"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return {"error": "timed out after 240 s"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"exit status {r.returncode}", "stderr_tail": r.stderr[-600:]}
    return json.loads(lines[-1])


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="images per GPU per step")
    ap.add_argument("--dtype", choices=["f32", "bf16", "fp8", "f32_fp16x2"], default="f32",
                    help="f32 = the parity path (default, what `value` is quoted on); bf16 = bf16-operand GEMMs "
                         "(BASELINE config 3; logits ~1e-2 from ViT_seq.c, so never the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-pointer (PCIe-inclusive) leg")
    ap.add_argument("--model", choices=["vit_b_16", "vit_l_16", "vit_h_14"], default="vit_b_16",
                    help="vit_b_16 is BASELINE.json's metric; the others are the parity-test shapes, timed for DESIGN.md")
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes for the CPU baseline (0 = auto)")
    ap.add_argument("--no-in-library-multi", action="store_true", help="skip the in-library multi-GPU leg (child process)")
    ap.add_argument("--sustain-s", type=float, default=30.0,
                    help="seconds of the same step behind the timed region for the `sustained` leg (0 = skip)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 PMC passes (child process)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--in-library-multi-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--first-device", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.in_library_multi_child:
        return in_library_multi_child(args.in_library_multi_child, args.first_device, args.batch, args.steps)
    if args.pmc_child:
        return pmc_child(args.batch)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as children of this process, which has
        # not touched the GPU and never will; it relays their output and exit status.
        raise SystemExit(spawn_ranks(args.gpus))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}: the launcher's --nproc-per-node and --gpus "
                         f"disagree (unset WORLD_SIZE to let bench.py start the ranks itself)")

    import __graft_entry__ as graft
    comm = None
    torch = None
    if world_env > 1 or os.environ.get("VIT_DIST_FORCE", "0") == "1":
        import torch  # before the library: one HIP runtime per process (binding.lib())
        pkg = graft.load_package()
        from vit_with_opencl_amd.host.dist import Comm
        comm = Comm()   # backend nccl (= RCCL); VIT_DIST_BACKEND=gloo only for single-GPU rehearsal
    pkg = graft.load_package()
    rank = comm.rank if comm else 0
    world = comm.world if comm else 1
    # one GPU per rank; VIT_BENCH_DEVICE pins every rank to one card for a gloo rehearsal
    device = int(os.environ.get("VIT_BENCH_DEVICE", comm.local_rank if comm else 0))

    cfg = pkg.preset(args.model)
    tokens = pkg.binding.tokens(cfg)
    label = {"vit_b_16": "ViT-B/16", "vit_l_16": "ViT-L/16", "vit_h_14": "ViT-H/14"}[args.model]
    B, NC = args.batch, cfg.num_classes

    weights = pkg.synth_weights(cfg, 0)
    model = pkg.ViTHip(cfg, weights, device=device, max_batch=B, precision=args.dtype)
    L = pkg.lib()

    # Synthetic batch, distinct images per rank: global image index = rank*B + i.  Generated
    # in slices and uploaded once; the timed region starts with everything in HBM.
    d_images = pkg.DeviceBuffer(B * cfg.in_chans * cfg.img_size * cfg.img_size)
    per = cfg.in_chans * cfg.img_size * cfg.img_size
    for lo in range(0, B, 64):
        n = min(64, B - lo)
        chunk = pkg.synth_images(cfg, rank * B + lo, n)
        pkg.binding.check(L.vh_h2d(d_images.ptr.value + lo * per * 4, chunk.ctypes.data, n * per * 4, None), "vh_h2d")
        pkg.binding.check(L.vh_device_sync(), "sync")

    d_probs = pkg.DeviceBuffer(B * NC)
    use_rccl = comm is not None and comm.backend == "nccl"
    if use_rccl:
        # logits land in a torch tensor so RCCL can gather them; launch on torch's stream
        # on a torch side stream (a non-zero handle: 0 would mean "the context's own stream" to the
        # library, which RCCL does not order against); the gather is enqueued on the same stream
        t_logits = torch.empty(B, NC, device="cuda", dtype=torch.float32)
        d_logits_ptr = t_logits.data_ptr()
        t_stream = torch.cuda.Stream()
        stream = t_stream.cuda_stream
        assert stream != 0
    else:
        d_logits = pkg.DeviceBuffer(B * NC)
        d_logits_ptr = d_logits.ptr
        stream = model.stream
    gathered = [None]

    def logits_host():
        """this rank's [B][NC] logits of the last step, on the host"""
        return t_logits.cpu().numpy() if use_rccl else d_logits.to_numpy((B, NC))

    def step():
        model.forward_device(d_images.ptr, B, d_logits_ptr, d_probs.ptr, stream)
        if use_rccl:
            with torch.cuda.stream(t_stream):
                gathered[0] = comm.gather_rows(t_logits)
        elif comm is not None:   # gloo rehearsal: through host memory
            gathered[0] = comm.gather_rows(torch.from_numpy(logits_host()))

    def fence():
        if comm is not None:
            comm.barrier()
        if use_rccl:
            torch.cuda.synchronize()
        pkg.binding.check(L.vh_device_sync(), "sync")

    for _ in range(args.warmup):
        step()
    fence()
    # Timed region: HIP events (on the launch stream) bracket only the kernel the roofline is quoted
    # on -- every recorded launch puts two event packets between kernels, ~2 % of a step when all
    # 88 launches are recorded.  The per-operator table comes from a separate, untimed pass below.
    model.profile_select(["fc1_gemm"])
    model.profile_enable(args.steps)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof_timed = model.profile_read()
    PROF_STEPS = 3
    model.profile_select(None)
    model.profile_enable(PROF_STEPS)
    for _ in range(PROF_STEPS):
        step()
    fence()
    prof = model.profile_read()
    model.profile_enable(0)

    if comm is not None:
        elapsed = comm.max_over_ranks(elapsed)

    # N > 1: rank 0 checks what the gather delivered.  Its own rows must be its own logits; for every other rank
    # it recomputes that rank's first and last images (global index r*B + i) ON THE SAME BATCH POSITIONS -- so the
    # same tiles and launches produce them -- and requires the gathered rows to equal them bit for bit.
    gathered_ok = None
    if comm is not None and rank == 0:
        rows_all = [np.ascontiguousarray(t.cpu().numpy()) for t in gathered[0]]
        assert len(rows_all) == world and all(r.shape == (B, NC) for r in rows_all), "gather returned the wrong shapes"
        own = logits_host()
        if not np.array_equal(rows_all[0], own):
            raise SystemExit("bench.py: rank 0's gathered rows differ from its own logits")
        edge = sorted(set(list(range(min(4, B))) + list(range(max(B - 4, 0), B))))
        d_chk = pkg.DeviceBuffer(B * NC)
        gathered_ok = 0
        for r in range(1, world):
            for i in edge:
                one = pkg.synth_images(cfg, r * B + i, 1)
                pkg.binding.check(L.vh_h2d(d_images.ptr.value + i * per * 4, one.ctypes.data, per * 4, None), "vh_h2d")
            pkg.binding.check(L.vh_device_sync(), "sync")
            model.forward_device(d_images.ptr, B, d_chk.ptr, None, model.stream)
            pkg.binding.check(L.vh_device_sync(), "sync")
            chk = d_chk.to_numpy((B, NC))
            if not np.array_equal(chk[edge], rows_all[r][edge]):
                bad = [i for i in edge if not np.array_equal(chk[i], rows_all[r][i])]
                raise SystemExit(f"bench.py: rows gathered from rank {r} differ from rank 0's recomputation of global "
                                 f"images {[r * B + i for i in bad]}")
            gathered_ok += 1
        for i in edge:     # rank 0's own images back in place
            one = pkg.synth_images(cfg, i, 1)
            pkg.binding.check(L.vh_h2d(d_images.ptr.value + i * per * 4, one.ctypes.data, per * 4, None), "vh_h2d")
        pkg.binding.check(L.vh_device_sync(), "sync")
        d_chk.free()

    # Secondary leg (N=1, default dtype only): the same step with bf16-operand GEMMs, reported
    # beside -- never instead of -- the fp32 `value`.
    bf16_leg = emu_leg = fp8_leg = None

    def end_to_end(m):
        """vit_hip_forward on 8*B separately allocated host images (PCIe, gather and scatter included)."""
        host = pkg.synth_images(cfg, 0, min(B, 512))
        n_e2e = 8 * B
        arr = (pkg.binding.ImageData * n_e2e)()
        for i in range(n_e2e):
            arr[i].n, arr[i].c, arr[i].h, arr[i].w = n_e2e, cfg.in_chans, cfg.img_size, cfg.img_size
            arr[i].data = pkg.binding.fptr(host[i % host.shape[0]])
        h_probs = np.empty((n_e2e, NC), dtype=np.float32)
        prow = (pkg.binding.f32p * n_e2e)(*[pkg.binding.fptr(h_probs[i]) for i in range(n_e2e)])
        pkg.binding.check(L.vit_hip_forward(m.ctx, arr, min(n_e2e, 2 * B), None, prow), "vit_hip_forward")   # warm-up
        t0e = time.perf_counter()
        pkg.binding.check(L.vit_hip_forward(m.ctx, arr, n_e2e, None, prow), "vit_hip_forward")
        dte = time.perf_counter() - t0e
        return {"value": round(n_e2e / dte, 1), "unit": "images/sec", "images": n_e2e, "chunk": B,
                "what": "vit_hip_forward: host images (separately allocated, pageable) -> pinned staging -> H2D -> forward -> "
                        "probabilities D2H -> caller's rows; double-buffered over chunks",
                "prob_sum_image0": float(h_probs[0].sum())}

    def secondary(precision, label, peak_tf, peak_note):
        m2 = pkg.ViTHip(cfg, weights, device=device, max_batch=B, precision=precision)
        folded2 = bool(L.vit_hip_ln_fold(m2.ctx))
        d_l2 = pkg.DeviceBuffer(B * NC)
        for _ in range(3):
            m2.forward_device(d_images.ptr, B, d_l2.ptr, d_probs.ptr, m2.stream)
        pkg.binding.check(L.vh_device_sync(), "sync")
        t2 = time.perf_counter()
        for _ in range(10):
            m2.forward_device(d_images.ptr, B, d_l2.ptr, d_probs.ptr, m2.stream)
        pkg.binding.check(L.vh_device_sync(), "sync")
        dt2 = (time.perf_counter() - t2) / 10
        m2.profile_enable(2)
        for _ in range(2):
            m2.forward_device(d_images.ptr, B, d_l2.ptr, d_probs.ptr, m2.stream)
        pkg.binding.check(L.vh_device_sync(), "sync")
        p2 = m2.profile_read()
        l2 = d_l2.to_numpy((B, NC))
        l32 = logits_host()
        # the fold's cancellation exposure: |mean| / std of the residual rows (of the first 64 images, after the last layer)
        exposure = None
        if folded2:
            xt = m2.read_tokens(min(B, 64)).astype(np.float64)
            exposure = round(float(np.abs(xt.mean(1) / xt.std(1)).max()), 4)
        e2e2 = end_to_end(m2) if (args.model == "vit_b_16" and not args.no_end_to_end) else None
        m2.close()
        fc1_ms2 = p2["fc1_gemm"][0] / max(p2["fc1_gemm"][1], 1)
        fc1_tf = 2.0 * B * tokens * cfg.embed_dim * cfg.mlp_hidden / (fc1_ms2 * 1e-3) / 1e12
        return ({"dtype": label, "value": round(B / dt2, 1), "unit": "images/sec", "ms_per_step": round(dt2 * 1e3, 3),
                 "max_abs_dlogit_vs_f32_path": float(np.abs(l2 - l32).max()),
                 "argmax_agreement_with_f32_path": float((l2.argmax(1) == l32.argmax(1)).mean()),
                 "roofline": {"bound": "mfma", "kernel": "fc1 GEMM of this mode", "achieved": round(fc1_tf, 1), "peak": peak_tf,
                              "unit": "TFLOP/s", "frac": round(fc1_tf / peak_tf, 4), "peak_basis": peak_note, "traffic": None},
                 "end_to_end": e2e2,
                 "layer_norms_folded": folded2, "residual_rows_max_abs_mean_over_std": exposure,
                 "launches_per_step": {k: cnt // 2 for k, (ms, cnt) in p2.items() if cnt},
                 "kernels_avg_ms": {k: round(ms / cnt, 4) for k, (ms, cnt) in p2.items() if cnt}}, l2)

    # Opt-in leg (never `value`): the last encoder layer's output projection and MLP evaluated for the class-token
    # rows only -- the rows the classifier reads; logits identical bit for bit (checked here).
    cls_leg = None
    if comm is None and args.dtype == "f32":
        l_full = logits_host()
        if model.set_last_layer_cls_only(True) is False:
            for _ in range(2):
                model.forward_device(d_images.ptr, B, d_logits_ptr, d_probs.ptr, stream)
            pkg.binding.check(L.vh_device_sync(), "sync")
            t2 = time.perf_counter()
            for _ in range(5):
                model.forward_device(d_images.ptr, B, d_logits_ptr, d_probs.ptr, stream)
            pkg.binding.check(L.vh_device_sync(), "sync")
            dt2 = (time.perf_counter() - t2) / 5
            cls_leg = {"what": "vit_hip_set_last_layer_cls_only(1): last layer's out-proj + LayerNorm + MLP on the "
                               "class-token rows only (the rows the classifier reads, ViT_seq.c:511); opt-in, default off",
                       "value": round(B / dt2, 1), "unit": "images/sec", "ms_per_step": round(dt2 * 1e3, 3),
                       "logits_bit_identical_to_full_evaluation": bool(np.array_equal(logits_host(), l_full))}
        model.set_last_layer_cls_only(False)
        model.forward_device(d_images.ptr, B, d_logits_ptr, d_probs.ptr, stream)
        pkg.binding.check(L.vh_device_sync(), "sync")

    emu_logits0 = None
    if comm is None and args.dtype == "f32":
        def leg(*a):
            """a secondary leg that fails is reported in its place (and on stderr); `value` and the line do not depend on it"""
            try:
                return secondary(*a)
            except Exception as e:   # noqa: BLE001 -- anything a leg's own context can raise
                import traceback
                traceback.print_exc()
                return {"error": f"{type(e).__name__}: {e}"[:400]}, None
        bf16_leg, _ = leg("bf16", "bf16 GEMM operands, fp32 accumulate/residual/attention", 2500.0, "dense bf16 MFMA")
        fp8_leg, _ = leg("fp8", "block-scaled e4m3 GEMM operands (OCP MX, 32-element e8m0 scales, no calibration), fp32 "
                         "accumulate/residual", 5000.0, "dense block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4)")
        emu_leg, emu_l = leg("f32_fp16x2", "fp32 operands emulated with two fp16 parts and three fp16 MFMAs per "
                             "product (22 of 24 significand bits; NOT exact), everything else as the fp32 path",
                             2500.0 / 3.0, "dense fp16 MFMA peak / 3: three fp16 MFMAs per emulated product block")
        emu_logits0 = emu_l[0] if emu_l is not None else None
        # restore the fp32 path's probabilities for the checks below
        model.forward_device(d_images.ptr, B, d_logits_ptr, d_probs.ptr, stream)
        pkg.binding.check(L.vh_device_sync(), "sync")

    e2e = None
    if comm is None and args.model == "vit_b_16" and not args.no_end_to_end:
        e2e = end_to_end(model)
        # restore the device-resident outputs the checks below read
        model.forward_device(d_images.ptr, B, d_logits_ptr, d_probs.ptr, stream)
        pkg.binding.check(L.vh_device_sync(), "sync")

    # What a Main.c user sees (Main.c:51-57 times the whole call): ViT_opencl() on 100 host images, context creation
    # (weight upload + repack + arenas) included, split into setup and forward by the library's own clock.
    drop_in = None
    if comm is None and args.dtype == "f32" and args.model == "vit_b_16" and not args.no_end_to_end:
        import ctypes as C
        n_di = 100
        host = pkg.synth_images(cfg, 0, n_di)
        arr = pkg.binding.image_array(host)
        h_probs = np.empty((n_di, NC), dtype=np.float32)
        prow = (pkg.binding.f32p * n_di)(*[pkg.binding.fptr(h_probs[i]) for i in range(n_di)])
        nets = pkg.binding.networks(weights)
        walls, setups, forwards = [], [], []
        sys.stdout.flush()
        saved = os.dup(1)                      # the drop-in prints its "setup time" lines like the reference: keep them
        devnull = os.open(os.devnull, os.O_WRONLY)   # out of this program's one-line stdout
        os.dup2(devnull, 1)
        try:
            for _ in range(3):
                t0d = time.perf_counter()
                L.ViT_opencl(arr, nets, prow)
                walls.append(time.perf_counter() - t0d)
                su, fw = C.c_double(), C.c_double()
                L.vit_hip_last_call_seconds(C.byref(su), C.byref(fw))
                setups.append(su.value)
                forwards.append(fw.value)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
            os.close(devnull)
        own = logits_host()
        drop_in = {"what": "ViT_opencl(images, networks, probabilities) on 100 separately allocated host images, as Main.c:54 "
                           "calls it: context creation (346 MB of weights H2D, planes repack, arenas, pinned staging) + "
                           "forward + teardown; best of 3 calls in this process (the first also pays code-object loading)",
                   "images": n_di, "wall_s": round(min(walls), 4), "setup_s": round(min(setups), 4),
                   "forward_s": round(min(forwards), 4), "first_call_wall_s": round(walls[0], 4),
                   "images_per_sec_whole_call": round(n_di / min(walls), 1),
                   "argmax_equal_to_device_resident_path": bool((h_probs.argmax(1) == own[:n_di].argmax(1)).all())
                   if B >= n_di else None}

    # Sustained load: the SAME step for >= 30 s of wall time behind the timed region (and behind the secondary legs, so
    # that those stay comparable with earlier rounds' lines).  `value` rests on a window of a second or two on a board
    # that regulates power over milliseconds and temperature over tens of seconds; this leg says what the rate is once
    # the board is warm.  Every rank runs the same number of steps (the count follows from
    # the max-over-ranks time above), synchronised about once a second; fc1's per-launch time (HIP events, as in the
    # timed region) is the clock proxy.  No reference counterpart (Main.c:51-57 times one shot).
    sustained = None
    if args.sustain_s > 0:
        step_s = elapsed / args.steps
        chunk = max(1, int(round(1.0 / step_s)))
        n_chunks = max(2, int(np.ceil(args.sustain_s / (chunk * step_s))))
        model.profile_select(["fc1_gemm"])
        model.profile_enable(chunk)          # every read below rewinds the event pool
        marks, fc1_marks = [], []
        fence()
        t0s = time.perf_counter()
        for _ in range(n_chunks):
            for _ in range(chunk):
                step()
            if use_rccl:
                torch.cuda.synchronize()
            pkg.binding.check(L.vh_device_sync(), "sync")
            marks.append(time.perf_counter() - t0s)
            ms_c, cnt_c = model.profile_read()["fc1_gemm"]
            fc1_marks.append(ms_c / max(cnt_c, 1))
        model.profile_enable(0)
        model.profile_select(None)
        total_s = marks[-1]

        def window(lo_s, hi_s):
            """chunks that END inside (lo_s, hi_s]: images/s per GPU and mean fc1 ms over them"""
            idx = [k for k, t in enumerate(marks) if lo_s < t <= hi_s] or [len(marks) - 1]
            t_begin = marks[idx[0] - 1] if idx[0] > 0 else 0.0
            rate = len(idx) * chunk * B / (marks[idx[-1]] - t_begin)
            return rate, float(np.mean([fc1_marks[k] for k in idx]))
        first_rate, first_fc1 = window(0.0, 5.0)
        last_rate, last_fc1 = window(total_s - 5.0, total_s)
        if comm is not None:     # slowest rank decides, as for `value`
            first_rate = B * chunk / comm.max_over_ranks(B * chunk / first_rate)
            last_rate = B * chunk / comm.max_over_ranks(B * chunk / last_rate)
        sustained = {"what": f"the same step repeated for {total_s:.1f} s behind the timed region ({chunk * n_chunks} steps, host "
                             f"sync every {chunk}); whole-job images/sec over the first and the last 5 s, fc1 ms per launch "
                             "(HIP events) as the clock proxy",
                     "seconds": round(total_s, 2), "steps": chunk * n_chunks,
                     "first_5s": {"value": round(world * first_rate, 1), "fc1_ms": round(first_fc1, 4)},
                     "last_5s": {"value": round(world * last_rate, 1), "fc1_ms": round(last_fc1, 4)},
                     "mean": round(world * B * chunk * n_chunks / total_s, 1), "unit": "images/sec"}

    failed = None
    if rank == 0:
        flops = model_flops(cfg, tokens)
        total_flops = sum(flops.values())
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed

        kernels = {}
        for name, (ms, cnt) in prof.items():
            if cnt == 0:
                continue
            avg_ms = ms / cnt
            entry = {"launches_per_step": cnt // PROF_STEPS, "avg_ms": round(avg_ms, 4),
                     "share_of_step": round(ms / PROF_STEPS / ms_per_step, 4)}
            if name in flops:
                per_launch = flops[name] * B / (cnt // PROF_STEPS)
                entry["tflops"] = round(per_launch / (avg_ms * 1e-3) / 1e12, 2)
                if name in ("out_proj_gemm", "fc2_gemm"):
                    # the two projections that update the fp32 residual stream in place are, in the reduced modes, memory
                    # kernels with a GEMM inside: report their algorithmic bytes per second next to the FLOP rate (operand in,
                    # residual rows in and out, and -- LayerNorms folded -- the next projection's operand out)
                    a_b = {"f32": 6.0, "f32_fp16x2": 4.0, "bf16": 2.0, "fp8": 1.03125}[args.dtype]
                    if os.environ.get("VIT_HIP_P3", "1") == "0" or os.environ.get("VIT_HIP_GEMM_FP32", "s").startswith("n"):
                        a_b = 4.0 if args.dtype == "f32" else a_b
                    rows_all = B * tokens
                    kdim = cfg.embed_dim if name == "out_proj_gemm" else cfg.mlp_hidden
                    nbytes = rows_all * (kdim * a_b + cfg.embed_dim * 8.0)
                    if bool(L.vit_hip_ln_fold(model.ctx)):
                        nbytes += rows_all * cfg.embed_dim * a_b
                    entry["hbm_gbs"] = round(nbytes / (avg_ms * 1e-3) / 1e9, 1)
                    entry["frac_hbm_peak"] = round(entry["hbm_gbs"] / PEAK_HBM_GBS, 4)
            elif name == "layer_norm":
                # algorithmic bytes: read one [rows][E] fp32 tensor, write it as fp32 or (pre-split path) as three
                # bf16 parts = 6 bytes per value (the final LN is tiny)
                p3_ln = args.dtype == "f32" and os.environ.get("VIT_HIP_P3", "1") != "0" and \
                    not os.environ.get("VIT_HIP_GEMM_FP32", "s").startswith("n")
                bytes_per = B * tokens * cfg.embed_dim * (4.0 + (6.0 if p3_ln else 4.0 if args.dtype in ("f32", "f32_fp16x2") else
                                                                 2.0 if args.dtype == "bf16" else 1.03125 if args.dtype == "fp8" else 1.0))
                if cnt // PROF_STEPS == 1:
                    # the reduced modes fold every LayerNorm but the final one into the projection behind it (csrc/norm_fold.h):
                    # what is left is the final LayerNorm on the B class-token rows, fp32 in and out -- a launch-latency kernel
                    bytes_per = B * cfg.embed_dim * 8.0
                    entry["note"] = "final LayerNorm on the class-token rows only; the other 24 are folded into the QKV / fc1 projections"
                entry["gbs"] = round(bytes_per / (avg_ms * 1e-3) / 1e9, 1)
                entry["frac_hbm_peak"] = round(entry["gbs"] / PEAK_HBM_GBS, 4)
            kernels[name] = entry

        fc1_ms, fc1_cnt = prof_timed["fc1_gemm"]      # measured inside the timed region
        fc1_flops_per_launch = 2.0 * B * tokens * cfg.embed_dim * cfg.mlp_hidden
        achieved = fc1_flops_per_launch / (fc1_ms / fc1_cnt * 1e-3) / 1e12
        # fp32 products are formed on the bf16 cores from an exact 3-way split (6 bf16 MFMAs per
        # product block, gemm_mfma.hip SPLIT3), unless VIT_HIP_GEMM_FP32=native selects the fp32 MFMA.
        native = os.environ.get("VIT_HIP_GEMM_FP32", "split3").startswith("n")
        if args.dtype == "f32_fp16x2":
            peak_tf, peak_note = 2500.0 / 3.0, "dense fp16 MFMA peak / 3: three fp16 MFMAs per emulated fp32 product block"
        elif args.dtype == "fp8":
            peak_tf, peak_note = 5000.0, "dense block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4: twice the bf16 rate)"
        elif args.dtype != "f32":
            peak_tf, peak_note = 2500.0, "dense bf16 MFMA"
        elif native:
            peak_tf, peak_note = PEAK_F32_MFMA_TFLOPS, "native fp32 MFMA (v_mfma_f32_32x32x2_f32)"
        else:
            peak_tf, peak_note = 2500.0 / 6.0, ("dense bf16 MFMA peak / 6: six bf16 MFMAs per fp32-equivalent "
                                                "product block; the native fp32 MFMA peak is 157.3")
        # MFMA utilisation of every projection GEMM against the same peak (the north star asks for it on QKV / MLP)
        for gname in ("qkv_gemm", "out_proj_gemm", "fc1_gemm", "fc2_gemm"):
            if gname in kernels and "tflops" in kernels[gname]:
                kernels[gname]["frac_mfma_peak"] = round(kernels[gname]["tflops"] / peak_tf, 4)
        # HBM-side bytes per launch of that kernel come from separate rocprofv3 --pmc passes
        # (FETCH_SIZE doubled per the gfx950 note, + WRITE_SIZE); bench.py cannot collect PMCs itself.
        p3 = args.dtype == "f32" and not native and os.environ.get("VIT_HIP_P3", "1") != "0"
        # The committed figure is tied to the kernel source it was measured on: tools/pmc_traffic.py stores the sha256
        # of csrc/gemm_p3.hip; when the file has changed since, the figure is withheld and marked stale.
        traffic, traffic_src, traffic_stale = None, None, None
        if B == 512 and args.model == "vit_b_16" and p3:
            if comm is None and not args.no_pmc and "VIT_HIP_P3" not in os.environ:
                traffic, traffic_src = measure_fc1_traffic(B)      # live: PMC passes over a child process of this very run
                traffic_stale = False if traffic is not None else None
            if traffic is None:     # N > 1, --no-pmc, or the profiler could not run: the committed passes, tied to the kernel source
                live_note = traffic_src
                traffic, traffic_src, traffic_stale = committed_traffic(ROOT)
                if live_note and traffic_src:
                    traffic_src += f" [live measurement unavailable: {live_note}]"
        rows = B * tokens
        if p3:     # operands and result as three bf16 parts: 6 bytes per value
            alg_bytes = (rows * (cfg.embed_dim + cfg.mlp_hidden) + cfg.mlp_hidden * cfg.embed_dim) * 6
            kname = ("gemm_p3_kernel<8,256,EPI_GELU,OUT_P3> (+ its gemm_p3_kernel<4,128,...> launch for the last partial "
                     "scheduling round), csrc/gemm_p3.hip")
            arith = ("both operands pre-split exactly into 3 bf16 parts by their producers (weights at context creation, "
                     "activations by LayerNorm), 6 x v_mfma_f32_16x16x32_bf16 per block, no split arithmetic in the K loop; "
                     "the GELU output is written pre-split for fc2")
        else:
            alg_bytes = (rows * (cfg.embed_dim + cfg.mlp_hidden) + cfg.mlp_hidden * cfg.embed_dim) * 4
            kname = ("gemm_f32_kernel<...EPI_GELU...>" if native else "gemm_mx_kernel<8,256,EPI_GELU,OUT_MX> (csrc/gemm_mx.hip)"
                     if args.dtype == "fp8" else "gemm_p3_kernel<8,256,EPI_GELU,OUT_PLANES,NPL=1> (csrc/gemm_p3.hip)"
                     if args.dtype == "bf16" else "gemm_mf16_kernel<...,EPI_GELU,...> (csrc/gemm_mfma.hip)")
            arith = ("native fp32 MFMA (v_mfma_f32_32x32x2_f32), Tile<256,256,4,4>" if native else
                     "bf16 operands on v_mfma_f32_16x16x32_bf16" if args.dtype == "bf16" else
                     "block-scaled e4m3 operands (MX, 32-element e8m0 scales) on v_mfma_scale_f32_16x16x128_f8f6f4" if args.dtype == "fp8" else
                     "fp32 operands as two fp16 parts, 3 x v_mfma_f32_16x16x32_f16 per block (not exact)"
                     if args.dtype == "f32_fp16x2" else
                     "exact 3-way bf16 split of fp32 operands inside the K loop (weights pre-split), 6 x v_mfma_f32_16x16x32_bf16 per block")
        roofline = {"bound": "mfma",
                    "kernel": "%s: fc1 GEMM (M=%d N=%d K=%d); %s" % (kname, rows, cfg.mlp_hidden, cfg.embed_dim, arith),
                    "peak_basis": peak_note,
                    "achieved": round(achieved, 2), "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": round(achieved / peak_tf, 4), "traffic": traffic,
                    "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                    "algorithmic_bytes": alg_bytes}

        # sanity of what was computed + parity against the CPU path in the same run
        if comm is not None:
            logits0 = gathered[0][0][0].cpu().numpy()
            assert len(gathered[0]) == world and all(t.shape == (B, NC) for t in gathered[0])
        else:
            logits0 = logits_host()[0]
        probs0 = d_probs.to_numpy((B, NC))[0]
        out = {
            "metric": f"images/sec {label} 224x224 bs{B}" if (args.model, B) != ("vit_b_16", 512) else "images/sec ViT-B/16 224x224 bs512", "value": round(value, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{label} 224x224 {args.dtype} forward (patch-embed..softmax), batch {B} per GPU, "
                                   f"device-resident inputs, random-init weights", "global_batch": world * B,
                       "parallelism": f"dp{world} (batch shards, replicated weights, RCCL gather of logits)"},
            "gemm_arithmetic": ("bf16 operands, fp32 accumulate" if args.dtype == "bf16" else
                                "block-scaled e4m3 operands (OCP MX: one e8m0 scale per 32 K elements), fp32 accumulate"
                                if args.dtype == "fp8" else
                                "fp32 operands as two fp16 parts (22 significant bits), 3 fp16 MFMAs per product, fp32 accumulate"
                                if args.dtype == "f32_fp16x2" else
                                "fp32 (native fp32 MFMA)" if native else
                                "fp32 operands split exactly into 3 bf16 parts, 6 bf16 MFMAs per product, fp32 accumulate"),
            "model_tflops": round(total_flops * B * args.steps * world / elapsed / 1e12, 2),
            "model_frac_of_f32_mfma_peak": round(total_flops * B * args.steps / elapsed / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "layer_norms_folded": bool(L.vit_hip_ln_fold(model.ctx)),
            "roofline": roofline, "kernels": kernels,
            "kernels_note": f"per-operator averages from a separate pass of {PROF_STEPS} steps with every launch bracketed "
                            "by HIP events; the roofline kernel is bracketed inside the timed region",
        }
        if sustained is not None:
            sustained["last_5s_vs_value"] = round(sustained["last_5s"]["value"] / value, 4)
            sustained["last_5s_more_than_3pct_below_value"] = bool(sustained["last_5s"]["value"] < 0.97 * value)
            out["sustained"] = sustained
        if bf16_leg is not None:
            out["bf16_gemm_mode"] = bf16_leg
            out["fp8_block_scaled_gemm_mode"] = fp8_leg
        if emu_leg is not None:
            out["fp32_fp16x2_emulation_mode"] = emu_leg
        if cls_leg is not None:
            out["class_token_rows_only_last_layer"] = cls_leg
        if e2e is not None:
            out["end_to_end"] = e2e
        if drop_in is not None:
            out["drop_in_100"] = drop_in
        if comm is not None:
            out["ranks"] = world
            out["gather"] = {"backend": "rccl" if use_rccl else comm.backend, "rccl_ranks": world if use_rccl else 0,
                             "gathered_rows_verified": gathered_ok,
                             "what": "rank 0 recomputed the first and last 4 images of every other rank's shard (global "
                                     "index rank*B+i) on the same batch positions and found the gathered rows bit-identical"}
        parity_failed = None
        if not args.no_cpu_baseline and args.model == "vit_b_16":
            # rank 0 only; at N > 1 the other ranks wait at the closing barrier meanwhile
            base, ref_logits = cpu_baseline(B, args.cpu_procs)
            out["cpu_baseline"] = base
            if ref_logits:
                gl = logits_host()
                imgs = sorted(ref_logits)
                dl = {i: float(np.abs(gl[i] - ref_logits[i]).max()) for i in imgs}
                out["parity"] = {"max_abs_dlogit_vs_ViT_seq": max(dl.values()),
                                 "argmax_equal": bool(all(int(gl[i].argmax()) == int(ref_logits[i].argmax()) for i in imgs)),
                                 "tolerance": 1e-4, "images": imgs,
                                 "per_image": {str(i): dl[i] for i in imgs if i in PARITY_IMAGES},
                                 "note": "images 0/1, 255, 498/499 and 511 sit on the GPU path's tile and launch boundaries "
                                         "(row 98 304, where the 256x256-tile launch hands over to the 128x128-tile launch, "
                                         "is in image 499)"}
                if emu_leg is not None and emu_logits0 is not None and 0 in ref_logits:
                    emu_leg["max_abs_dlogit_vs_ViT_seq"] = float(np.abs(emu_logits0 - ref_logits[0]).max())
                # the stated tolerance is enforced for the parity path (fp32 and its fp16-pair emulation); the reduced
                # modes state theirs in DESIGN 9/10 and tests/ (bf16: 4e-2)
                tol = {"f32": 1e-4, "f32_fp16x2": 1e-4, "bf16": 4e-2}.get(args.dtype)
                out["parity"]["tolerance"] = tol
                if tol is not None and not (out["parity"]["max_abs_dlogit_vs_ViT_seq"] <= tol and
                                            (out["parity"]["argmax_equal"] or args.dtype == "bf16")):
                    parity_failed = f"parity {out['parity']['max_abs_dlogit_vs_ViT_seq']:.3e} exceeds {tol} (or arg-max differs)"
        out["checks"] = {"logits_finite": bool(np.isfinite(logits0).all()),
                         "prob_sum_image0": float(probs0.sum())}
        if args.dtype == "f32" and args.model == "vit_b_16" and not args.no_in_library_multi:
            # the library's own multi-GPU entry (RCCL gather in C) on the same devices, in a child process, while the
            # other ranks wait on the host (Comm.rank0_says_done): beside `value`, never instead of it
            one_card = "VIT_BENCH_DEVICE" in os.environ
            out["in_library_multi_gpu"] = run_in_library_multi(1 if one_card else world, device if one_card else 0, B)
        print(json.dumps(out), flush=True)
        if parity_failed or not out["checks"]["logits_finite"]:
            failed = parity_failed or "non-finite logits"

    model.close()
    if comm is not None:
        comm.rank0_says_done()      # ranks > 0 wait here, on the host, while rank 0 ran its CPU baseline and checks
        comm.close()
    if failed:
        raise SystemExit("bench.py: " + failed)


if __name__ == "__main__":
    main()
