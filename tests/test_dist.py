"""World-size-2 (and 3) gloo tests of the N>1 path on CPU: shards cover the batch once,
rank 0 reassembles the logits in global image order, timing is the max over ranks."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,total", [(2, 8), (2, 7), (3, 10)])
def test_sharded_gather_over_gloo(tmp_path, world, total):
    port = _free_port()
    out = tmp_path / "rank0.json"
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "dist_worker.py"), str(total), "1000",
                                       str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = json.loads(out.read_text())
    assert res["ok"] and res["rows"] == total and res["world"] == world
    assert res["slowest"] == float(world)            # max over ranks of (1 + rank)
    assert sum(res["counts"]) == total


@pytest.mark.gpu
def test_bench_two_ranks_launch_themselves_verify_the_gather_and_report_the_cpu_baseline():
    """bench.py's own N > 1 body on the test box's single card: `python bench.py --gpus 2` with no launcher in the
    environment starts its two ranks itself (torch.distributed.run as a child of a process that never touches the
    GPU), both pinned to device 0 with the gather over gloo (RCCL refuses two ranks on one GPU; the RCCL group of one
    is covered by test_rccl_gather_is_ordered_after_the_forward).  The line must carry the rank count, the bitwise
    verification of the other rank's gathered rows, the CPU baseline and the parity block -- everything the N = 1 line has."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VIT_DIST_BACKEND="gloo", VIT_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "64", "--cpu-procs", "2", "--sustain-s", "3"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["steps"] == 2 and out["config"]["global_batch"] == 128
    assert out["gather"]["gathered_rows_verified"] == 1 and out["gather"]["backend"] == "gloo"
    assert out["value"] > 0 and out["scaling"] == "weak" and out["roofline"]["frac"] > 0
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] in ("reference", "port")
    assert out["parity"]["max_abs_dlogit_vs_ViT_seq"] <= 1e-4 and out["parity"]["argmax_equal"]
    assert out["checks"]["logits_finite"]
    lib_leg = out["in_library_multi_gpu"]          # the C-side RCCL entry, degenerate group of one on this box
    assert "error" not in lib_leg and lib_leg["devices"] == [0] and lib_leg["value"] > 0 and lib_leg["logits_finite"]
    assert len(lib_leg["host_enqueue_ms"]) == 1 and 0.0 < lib_leg["host_enqueue_ms"][0] < 1000.0 and lib_leg["gather_verified"] is None
    sus = out["sustained"]                          # the same step for a few seconds behind the timed region, both ranks in step
    assert sus["seconds"] >= 2.5 and sus["first_5s"]["value"] > 0 and sus["last_5s"]["value"] > 0 and sus["last_5s"]["fc1_ms"] > 0
    assert isinstance(sus["last_5s_more_than_3pct_below_value"], bool)


@pytest.mark.gpu
def test_bench_single_gpu_line_carries_every_leg_without_errors():
    """The N = 1 line the driver records: besides `value`, `roofline` and the parity block, the secondary legs (bf16 / MX fp8 /
    fp16-pair modes, class-token-rows-only last layer, end to end, drop-in) -- each leg is guarded in bench.py (a failure is
    reported in its place), so this test is what keeps a broken leg from passing silently."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "64", "--sustain-s", "0",
                        "--no-pmc", "--no-cpu-baseline", "--no-in-library-multi"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["dtype"] == "f32" and 0 < out["roofline"]["frac"] < 1
    for key in ("bf16_gemm_mode", "fp8_block_scaled_gemm_mode", "fp32_fp16x2_emulation_mode", "class_token_rows_only_last_layer",
                "end_to_end", "drop_in_100"):
        assert key in out and "error" not in out[key], (key, out.get(key))
    for key in ("bf16_gemm_mode", "fp8_block_scaled_gemm_mode"):
        assert out[key]["value"] > out["value"] and out[key]["layer_norms_folded"] is True
        assert out[key]["launches_per_step"]["layer_norm"] == 1 and 0 <= out[key]["residual_rows_max_abs_mean_over_std"] < 1
    assert out["class_token_rows_only_last_layer"]["logits_bit_identical_to_full_evaluation"] is True


@pytest.mark.gpu
def test_bench_rccl_branch_with_a_group_of_one():
    """bench.py's RCCL-specific body -- logits in a torch tensor, forward on a torch side stream, dist.gather over the
    nccl (= RCCL) backend enqueued on the same stream, barrier + device synchronise around the timed region -- with a
    process group of ONE rank on the box's single card ($VIT_DIST_FORCE=1): everything the N > 1 launch does except a
    second GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("VIT_DIST_BACKEND", "VIT_BENCH_DEVICE")}
    env.update(VIT_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64",
                        "--no-cpu-baseline", "--no-in-library-multi", "--sustain-s", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["gather"]["backend"] == "rccl" and out["gather"]["rccl_ranks"] == 1
    assert out["value"] > 0 and out["checks"]["logits_finite"] and abs(out["checks"]["prob_sum_image0"] - 1.0) < 1e-5
