"""CPU tests of the host side: ABI layout, exported symbols, config bookkeeping,
synthetic-data determinism, loaders, sharding.  No compute calls (no GPU here)."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/MulticoreMainProject")


def test_library_loads_and_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    for sym in pkg.binding.EXPORTS:
        assert hasattr(L, sym), f"libvit_hip.so does not export {sym}"


def test_headers_declare_exactly_the_exports(pkg):
    """Every function prototype in include/*.h is in EXPORTS (and vice versa)."""
    import re
    declared = set()
    for h in (ROOT / "include").glob("*.h"):
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        text = re.sub(r"#define VH_CHECK.*?while \(0\)", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(vh_[a-z0-9_]+|vit_[a-z0-9_]+|ViT_opencl|load_image_data|load_weights)\s*\(", text))
    declared -= {"vh_check_err_"}
    assert declared == set(pkg.binding.EXPORTS)


def test_struct_layout_is_the_reference_abi(pkg):
    b = pkg.binding
    assert C.sizeof(b.ImageData) == 24 and b.ImageData.data.offset == 16
    assert [getattr(b.ImageData, f).offset for f in "nchw"] == [0, 4, 8, 12]
    assert C.sizeof(b.Network) == 16 and b.Network.data.offset == 0 and b.Network.size.offset == 8


@pytest.mark.skipif(not REF.exists(), reason="reference tree not present")
def test_struct_layout_against_reference_header(tmp_path):
    """Compile a probe that includes the reference's own Network.h next to ours and
    static-asserts identical sizes/offsets (header only; nothing is copied)."""
    src = tmp_path / "probe.c"
    src.write_text(r'''
#include <stddef.h>
#include "%s/Network.h"
typedef ImageData RefImage; typedef Network RefNet;
#define ImageData OurImage
#define Network OurNet
#define load_image_data our_load_image_data
#define load_weights our_load_weights
#include "%s/include/Network.h"
_Static_assert(sizeof(RefImage) == sizeof(OurImage), "ImageData size");
_Static_assert(offsetof(RefImage, data) == offsetof(OurImage, data), "ImageData.data");
_Static_assert(offsetof(RefImage, w) == offsetof(OurImage, w), "ImageData.w");
_Static_assert(sizeof(RefNet) == sizeof(OurNet), "Network size");
_Static_assert(offsetof(RefNet, size) == offsetof(OurNet, size), "Network.size");
int main(void) { return 0; }
''' % (REF, ROOT))
    subprocess.run(["gcc", "-fcommon", "-c", str(src), "-o", str(tmp_path / "probe.o")], check=True)


def test_config_presets_and_tensor_sizes(pkg):
    L, b = pkg.lib(), pkg.binding
    cfg = pkg.preset("vit_b_16")
    assert (cfg.img_size, cfg.patch_size, cfg.embed_dim, cfg.depth, cfg.num_heads, cfg.mlp_hidden) == \
        (224, 16, 768, 12, 12, 3072)
    assert cfg.eps == 1e-6 and b.tokens(cfg) == 197
    assert L.vit_config_num_tensors(C.byref(cfg)) == 152
    size = lambda i: L.vit_config_tensor_size(C.byref(cfg), i)  # noqa: E731
    # SURVEY Appendix B
    assert [size(i) for i in (0, 1, 2, 3)] == [768, 589824, 768, 151296]
    assert [size(4 + k) for k in range(12)] == [768, 768, 1769472, 2304, 589824, 768, 768, 768,
                                                 2359296, 3072, 2359296, 768]
    assert [size(i) for i in (148, 149, 150, 151)] == [768, 768, 768000, 1000]
    assert size(152) == 0 and size(-1) == 0
    total = sum(size(i) for i in range(152))
    assert total == 86567656  # 86.57 M parameters
    with pytest.raises(ValueError):
        pkg.preset("vit_tiny")
    h14 = pkg.preset("vit_h_14")
    assert b.tokens(h14) == 257 and L.vit_config_num_tensors(C.byref(h14)) == 392


def _np_synth(count, seed, scale, offset):
    """numpy twin of vit_synth_fill (csrc/vit_config.c)."""
    def mix(x):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))
    with np.errstate(over="ignore"):
        base = mix(np.array([seed], dtype=np.uint64))
        u24 = (mix(base + np.arange(count, dtype=np.uint64)) >> np.uint64(40)).astype(np.int32)
    u = u24.astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)
    return np.float32(offset) + np.float32(scale) * u


def test_synthetic_data_is_deterministic_and_portable(pkg, oracle):
    L, b = pkg.lib(), pkg.binding
    a = np.empty(10007, dtype=np.float32)
    L.vit_synth_fill(b.fptr(a), a.size, 1234, 0.25, 0.3)
    assert np.array_equal(a, _np_synth(a.size, 1234, 0.25, 0.3))
    assert np.array_equal(a, oracle.synth_fill(a.size, 1234, 0.25, 0.3))  # oracle build sees the same bytes
    assert a.min() >= 0.05 and a.max() < 0.55 and abs(float(a.mean()) - 0.3) < 0.01
    cfg = pkg.preset("vit_b_16")
    img = pkg.synth_images(cfg, 5, 1)[0]
    assert np.array_equal(img, oracle.synth_image(5)) and img.min() >= -2 and img.max() < 2
    w = pkg.synth_weights(cfg, 0)
    assert len(w) == 152 and np.array_equal(w[150], oracle.synth_weights(0)[150])


def test_loaders_round_trip(pkg, tmp_path):
    """load_image_data / load_weights read the reference's on-disk formats
    (Network.c:41-71: 4 x int32 header; :111-132 index from file name; :208-211 rounding)."""
    L, b = pkg.lib(), pkg.binding
    imgs = np.arange(2 * 3 * 4 * 4, dtype=np.float32).reshape(2, 3, 4, 4) / 7
    path = tmp_path / "input-2.bin"
    assert L.vit_write_image_file(str(path).encode(), b.image_array(imgs), 2) == 0
    raw = path.read_bytes()
    assert np.array_equal(np.frombuffer(raw[:16], dtype=np.int32), [2, 3, 4, 4])
    loaded = L.load_image_data(str(path).encode())
    assert loaded[0].n == 2 and loaded[1].w == 4
    got = np.ctypeslib.as_array(loaded[1].data, shape=(48,))
    assert np.array_equal(got, imgs[1].ravel())
    assert not L.load_image_data(str(tmp_path / "missing.bin").encode())

    wdir = tmp_path / "Network"
    wdir.mkdir()
    w7 = np.array([0.1234564, -0.9999996, 1.5, 2.0000004], dtype=np.float32)
    assert L.vit_write_weight_file(str(wdir).encode(), 7, b"encoder_layers_x_in_proj_bias", b.fptr(w7), 4) == 0
    (wdir / "notes.txt").write_text("ignored")
    (wdir / "Weight_99_too_big.bin").write_bytes(b"\0" * 8)
    nets = (b.Network * 10)()
    L.load_weights(str(wdir).encode(), nets, 10)
    assert nets[7].size == 4 and not nets[0].data and nets[0].size == 0
    got = np.ctypeslib.as_array(nets[7].data, shape=(4,))
    t = (w7 * np.float32(1e6)).astype(np.float64)          # the fp32 product, exactly
    r = np.copysign(np.floor(np.abs(t) + 0.5), t)          # roundf: half away from zero
    expect = r.astype(np.float32) / np.float32(1e6)
    assert np.array_equal(got, expect)


def test_vit_hip_create_rejects_bad_tensors_before_touching_the_device(pkg):
    L, b = pkg.lib(), pkg.binding
    cfg = pkg.preset("vit_b_16")
    nets = (b.Network * 152)()
    ctx = C.c_void_p()
    assert L.vit_hip_create(C.byref(ctx), C.byref(cfg), nets, 151, 0, 1) == 2   # wrong count
    assert L.vit_hip_create(C.byref(ctx), C.byref(cfg), nets, 152, 0, 1) == 3   # NULL tensors
    assert not ctx.value


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    """In the build container there is no device: vh_init must fail with a message,
    and nothing computes on the CPU instead."""
    L = pkg.lib()
    if L.vh_device_count() > 0:
        pytest.skip("a GPU is present")
    assert L.vh_init(0) != 0
    assert b"no HIP device" in L.vh_last_error()
    cfg = pkg.preset("vit_b_16")
    with pytest.raises(pkg.VitHipError):
        pkg.ViTHip(cfg, pkg.synth_weights(cfg, 0), device=0, max_batch=1)


def test_product_path_never_touches_the_oracle():
    """Nothing under vit-with-opencl_amd/ or include/ may import, include, link or call
    anything under oracle/ (comments may mention it)."""
    import re
    pat = re.compile(r"import\s+oracle|from\s+oracle|oracle/|liboracle|oracle\.py|\bport_[a-z_0-9]+\s*\(|ref_harness")
    files = [p for p in list((ROOT / "vit-with-opencl_amd").rglob("*")) + list((ROOT / "include").rglob("*"))
             if p.is_file() and (p.suffix in {".py", ".c", ".h", ".hip"} or p.name == "Makefile")]
    assert len(files) >= 10
    for p in files:
        assert not pat.search(p.read_text(errors="ignore")), p


def test_shard_range_covers_batch_exactly_once(pkg):
    for total in (1, 7, 512, 513, 4096):
        for world in (1, 2, 3, 8):
            spans = [pkg.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= (total + world - 1) // world


def test_shard_runner_covers_the_batch_once_and_reports_failures(pkg):
    """The in-library multi-GPU driver without a GPU: vit_shard_range / vit_shard_run (what
    vit_hip_forward_multi is made of) with a stub "forward" that records its shard -- every image is
    visited exactly once, shards are contiguous and in device order, each non-empty shard runs on its own
    host thread, empty shards are skipped, and a failing shard's status comes back."""
    import threading
    L, b = pkg.lib(), pkg.binding
    for total in (0, 1, 5, 64, 100, 513):
        for shards in (1, 2, 3, 8):
            spans = []
            for s_ in range(shards):
                lo, hi = C.c_int(), C.c_int()
                L.vit_shard_range(total, s_, shards, C.byref(lo), C.byref(hi))
                spans.append((lo.value, hi.value))
                assert (lo.value, hi.value) == b.shard_range(total, s_, shards)     # same rule as the per-process form
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(shards - 1))
            seen = np.zeros(max(total, 1), dtype=np.int32)
            calls, lock = [], threading.Lock()

            def stub(arg, shard, lo, hi):
                with lock:
                    calls.append((shard, lo, hi, threading.get_ident()))
                seen[lo:hi] += 1
                return 0
            assert L.vit_shard_run(total, shards, b.SHARD_FN(stub), None) == 0
            assert (seen[:total] == 1).all()
            assert sorted(c[:3] for c in calls) == [(s_, lo, hi) for s_, (lo, hi) in enumerate(spans) if hi > lo]
            assert len({c[3] for c in calls}) == len(calls)                            # one host thread per shard
    fail = b.SHARD_FN(lambda arg, shard, lo, hi: 7 if shard == 2 else 0)
    assert L.vit_shard_run(100, 4, fail, None) == 7
    assert L.vit_shard_run(100, 0, fail, None) != 0 and L.vit_shard_run(-1, 2, fail, None) != 0


def test_shards_are_entered_concurrently_and_timed(pkg):
    """The enqueue stage of vit_hip_forward_device_multi (csrc/vit_gather_rccl.c) is vit_shard_run_timed with one shard
    per device.  A stub "enqueue" that waits at a rendezvous for ALL shards can only return if every shard was entered
    before any had finished -- a runner that entered them one after the other (round 3's loop) times out at the barrier.
    The per-shard host times come back, and a failing shard's status wins after every other shard has been joined."""
    import threading
    import time
    L, b = pkg.lib(), pkg.binding
    for n in (2, 4, 8):
        barrier = threading.Barrier(n, timeout=20.0)
        inside, lock, peak = [0], threading.Lock(), [0]

        def stub(arg, shard, lo, hi):
            assert hi - lo == 1 and lo == shard
            with lock:
                inside[0] += 1
                peak[0] = max(peak[0], inside[0])
            try:
                barrier.wait()
            except threading.BrokenBarrierError:
                return 9
            time.sleep(0.002 * (shard + 1))
            with lock:
                inside[0] -= 1
            return 0
        ms = (C.c_double * n)()
        assert L.vit_shard_run_timed(n, n, b.SHARD_FN(stub), None, ms) == 0
        assert peak[0] == n                                   # all shards were inside the stub at once
        assert all(ms[s_] >= 2.0 * (s_ + 1) * 0.9 for s_ in range(n)) and max(ms) < 20000.0
    # a failing shard: its status is returned, and only after the slow shards have finished (nothing left running)
    done = []

    def slow_or_fail(arg, shard, lo, hi):
        if shard == 1:
            return 5
        time.sleep(0.05)
        done.append(shard)
        return 0
    ms = (C.c_double * 4)()
    assert L.vit_shard_run_timed(4, 4, b.SHARD_FN(slow_or_fail), None, ms) == 5
    assert sorted(done) == [0, 2, 3]
    assert L.vit_shard_run_timed(4, 4, b.SHARD_FN(slow_or_fail), None, None) == 5      # the times are optional


def test_planes_file_header_is_checked_before_anything_is_allocated(pkg, tmp_path):
    """vit_hip_create_from_planes on files that must be refused -- without a GPU, i.e. before the library has touched a
    device or allocated a slab: a foreign file, another operand-layout version, a header asking for absurd dimensions
    (a corrupt or hostile file must not be able to request arbitrary host and HBM allocations), and a header whose slab
    sizes are not the ones this library derives from (shape, precision, fold flag)."""
    import struct
    L, b = pkg.lib(), pkg.binding
    L.vit_hip_create_from_planes.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int]
    FMT = "<8sIii8i4xd4QQiiQ"
    assert struct.calcsize(FMT) == 120

    def header(magic=b"VITPLN02", precision=0, n_tensors=152, ints=(224, 16, 3, 1000, 768, 12, 12, 3072), sizes=(1, 1, 1, 152), fold=(0, 0),
               version=2):
        return struct.pack(FMT, magic, 120, precision, n_tensors, *ints, 1e-6, *sizes, fold[0], fold[1], version, 0)

    def load(blob):
        path = tmp_path / "x.planes"
        path.write_bytes(blob + b"\0" * 4096)
        ctx = C.c_void_p()
        rc = L.vit_hip_create_from_planes(C.byref(ctx), str(path).encode(), 0, 4)
        assert not ctx.value
        return rc, L.vh_last_error().decode()
    rc, msg = load(header(magic=b"VITPLN01"))
    assert rc == 124 and "not a planes file" in msg
    rc, msg = load(header(version=1))
    assert rc == 124
    rc, msg = load(header(ints=(224, 16, 3, 1000, 768, 100000, 12, 3072), n_tensors=4 + 12 * 100000 + 4))
    assert rc != 0 and "shape or precision" in msg                      # depth 100 000: refused, nothing allocated
    rc, msg = load(header(ints=(224, 16, 3, 1000, 1 << 20, 12, 12, 3072)))
    assert rc != 0 and "shape or precision" in msg                      # embed_dim 2^20
    rc, msg = load(header(sizes=(1 << 50, 1 << 50, 0, 152)))
    assert rc == 125 and "slab sizes" in msg                             # a petabyte of weights: refused by the size check
    rc, msg = load(header(precision=7))
    assert rc != 0
    ctx = C.c_void_p()
    assert L.vit_hip_create_from_planes(C.byref(ctx), str(tmp_path / "missing.planes").encode(), 0, 4) == 123


def test_fp8_reference_round_trips_every_code():
    """tests/fp8_ref.py (the numpy statement of OCP e4m3 the GPU casts are checked against):
    every finite code survives dequantise -> quantise, ties go to the even code, overflow
    saturates at 448."""
    import fp8_ref
    codes = np.array([c for c in range(256) if (c & 0x7f) != 0x7f], dtype=np.uint8)
    vals = fp8_ref.dequantize(codes)
    back = fp8_ref.quantize(vals)
    keep = codes != 0x80                                   # -0.0 quantises to +0 or -0: both fine
    assert np.array_equal(back[keep] & 0x7f, codes[keep] & 0x7f) and np.array_equal(back[keep] >> 7, codes[keep] >> 7)
    assert fp8_ref.dequantize(fp8_ref.quantize(np.array([1e9, -1e9, 448.0, 464.0], np.float32))).tolist() == [448.0, -448.0, 448.0, 448.0]
    assert fp8_ref.quantize(np.array([1.0625, 1.1875, 2.0 ** -10], np.float32)).tolist() == [56, 58, 0]


def test_fp8_reference_equals_torch_float8_e4m3fn():
    """Pins tests/fp8_ref.py (builder-authored) to an independent implementation of OCP e4m3: PyTorch's
    float8_e4m3fn cast on a million values (all magnitudes from subnormal to beyond 448, exact ties
    included) and every code's value.  torch saturates nothing (values beyond 448 become NaN), so the
    comparison is on |x| <= 448; the saturation rule is checked in the test above."""
    torch = pytest.importorskip("torch")
    import fp8_ref
    rng = np.random.default_rng(7)
    x = np.concatenate([
        rng.standard_normal(400000).astype(np.float32) * np.float32(3.0),
        (rng.random(300000).astype(np.float32) * np.float32(2.0) - np.float32(1.0)) * np.float32(448.0),
        np.exp(rng.uniform(np.log(2.0 ** -12), np.log(448.0), 290000)).astype(np.float32) * rng.choice([-1.0, 1.0], 290000).astype(np.float32),
        (fp8_ref.TABLE[:-1] + fp8_ref.TABLE[1:]).astype(np.float32) / np.float32(2.0),          # exact midpoints: ties
        -((fp8_ref.TABLE[:-1] + fp8_ref.TABLE[1:]).astype(np.float32) / np.float32(2.0)),
        fp8_ref.TABLE.astype(np.float32),
        np.array([0.0, -0.0, 2.0 ** -10, 2.0 ** -9, 3 * 2.0 ** -11, 448.0, -448.0], np.float32)])
    x = x[np.abs(x) <= 448.0]
    want = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = fp8_ref.quantize(x)
    zero = (want & 0x7f) == 0                              # signed zeros: compare magnitudes only
    assert np.array_equal(got[~zero], want[~zero]) and np.array_equal(got[zero] & 0x7f, want[zero] & 0x7f)
    codes = np.array([c for c in range(256) if (c & 0x7f) != 0x7f], dtype=np.uint8)
    assert np.array_equal(fp8_ref.dequantize(codes), torch.from_numpy(codes).view(torch.float8_e4m3fn).to(torch.float32).numpy())
    assert x.size > 900000


def test_result_file_and_strict_comparison(pkg, tmp_path):
    """vit_write_result_file writes Main.c's line format with a per-image arg-max (class 0 does not
    leak between images as with Main.c:59's pred_idx); vit_compare_rows reports what comparator.c's
    0.01 check hides."""
    import ctypes as C
    L, b = pkg.lib(), pkg.binding
    rng = np.random.default_rng(5)
    want = rng.standard_normal((6, 1000)).astype(np.float32)
    want[2, 0] = 9.0                       # image 2 predicts class 0 after image 1 predicted class 7
    want[1, 7] = 9.0
    rows = (b.f32p * 6)(*[b.fptr(want[i]) for i in range(6)])
    path = tmp_path / "result.txt"
    assert L.vit_write_result_file(str(path).encode(), rows, 6, 1000) == 0
    lines = path.read_text().splitlines()
    assert lines[1] == "[1] label: 7 / prob: 9.000000" and lines[2] == "[2] label: 0 / prob: 9.000000"
    assert [int(l.split("label:")[1].split("/")[0]) for l in lines] == want.argmax(1).tolist()

    got = want + rng.uniform(-1e-3, 1e-3, want.shape).astype(np.float32)
    got[4, want[4].argmax()] -= 50.0       # a real top-1 error
    a, c = np.sort(want[5])[-1], np.sort(want[5])[-2]
    i1, i2 = want[5].argmax(), np.argsort(want[5])[-2]
    want[5, i2] = want[5, i1] - 1e-4       # a near tie in the reference ...
    got[5] = want[5]
    got[5, i2] += 3e-4                     # ... flipped by an error of its own size
    rep = b.CompareReport()
    assert L.vit_compare_rows(b.fptr(got), b.fptr(want), 6, 1000, 3e-4, C.byref(rep)) == 0
    assert rep.rows == 6 and rep.classes == 1000 and rep.nonfinite == 0
    assert abs(rep.max_abs_diff - 50.0) < 1e-2
    assert rep.top1_equal == 4 and rep.top1_equal_or_near_tie == 5
    assert 0.9 < rep.top5_overlap <= 1.0
    got[0, 3] = np.nan
    assert L.vit_compare_rows(b.fptr(got), b.fptr(want), 6, 1000, 3e-4, C.byref(rep)) == 0 and rep.nonfinite == 1


def test_public_headers_are_plain_c_and_cxx(tmp_path):
    """The boundary is a C ABI: every public header compiles on its own as C11 (the reference's
    Main.c is C) and as C++ (extern "C" guards), with warnings as errors."""
    import subprocess
    root = Path(__file__).resolve().parent.parent
    for hdr in ("Network.h", "ViT_opencl.h", "kernelHandler.h"):
        for lang, std, cc in (("c", "-std=c11", "gcc"), ("c++", "-std=c++17", "g++")):
            src = tmp_path / f"t.{'c' if lang == 'c' else 'cpp'}"
            src.write_text(f'#include "{hdr}"\nint main(void) {{ return 0; }}\n')
            r = subprocess.run([cc, std, "-Wall", "-Wextra", "-Werror", "-fsyntax-only", f"-I{root / 'include'}", str(src)],
                               capture_output=True, text=True)
            assert r.returncode == 0, f"{hdr} as {lang}: {r.stderr[-800:]}"


def test_mx_reference_scale_rule_layout_and_torch_cross_check():
    """tests/mx_ref.py, the numpy statement the GPU's block-scaled fp8 producers are compared with byte for byte:
    (i) the scale of a block is the SMALLEST power of two that brings its maximum inside e4m3's range (<= 448), so
    nothing saturates and no smaller scale would do; (ii) the elements are torch's float8_e4m3fn cast of x / scale;
    (iii) dequantise(quantise(x)) is within half an e4m3 step of the block maximum's binade; (iv) the storage order
    values[K/128][rows][128], scales[K/128][4][rows] with block b in lane-group slot 2 (b & 1) + (b >> 1)."""
    torch = pytest.importorskip("torch")
    import mx_ref
    rng = np.random.default_rng(11)
    rows, cols = 64, 512
    x = (rng.standard_normal((rows, cols)) * np.exp(rng.uniform(-12, 12, (rows, cols // 32)).repeat(32, axis=1))).astype(np.float32)
    x[3, 64:96] = 0.0                                              # an all-zero block
    x[5, 0] = np.float32(1.8 * 2.0 ** 9)                           # maximum with a significand above 1.75: exponent one up
    values, scales = mx_ref.quantize(x)
    assert values.shape == (cols // 128, rows, 128) and scales.shape == (cols // 128, 4, rows)
    blocks = x.reshape(rows, cols // 32, 32)
    amax = np.abs(blocks).max(axis=2).astype(np.float64)
    # undo the storage order
    sb = np.empty((rows, cols // 128, 4), dtype=np.int32)
    for b in range(4):
        sb[:, :, b] = scales[:, 2 * (b & 1) + (b >> 1), :].T.astype(np.int32) - 127
    e = sb.reshape(rows, cols // 32)
    live = amax > 2.0 ** -118
    assert (amax[live] / 2.0 ** e[live] <= 448.0).all()                       # nothing clips
    assert (amax[live] / 2.0 ** (e[live] - 1) > 448.0).all()                  # and no smaller scale would do
    assert (e[~live] == -126).all()
    q = values.transpose(1, 0, 2).reshape(rows, cols // 32, 32)
    scaled = (blocks * np.ldexp(np.float32(1.0), -e)[:, :, None].astype(np.float32)).astype(np.float32)
    want = torch.from_numpy(scaled).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    zero = (want & 0x7f) == 0
    assert np.array_equal(q[~zero], want[~zero]) and np.array_equal(q[zero] & 0x7f, want[zero] & 0x7f)
    back = mx_ref.dequantize(values, scales).reshape(rows, cols // 32, 32)
    assert (np.abs(back - blocks) <= 2.0 ** -4 * np.maximum(amax, 2.0 ** -126)[:, :, None] * 1.0000001).all()


def test_bench_traffic_figure_is_withheld_when_the_kernel_source_changed(tmp_path):
    """bench.py's roofline.traffic comes from committed PMC passes; the record carries the sha256 of csrc/gemm_p3.hip and
    the figure is reported only while that file is unchanged (the committed record must match the committed source)."""
    import hashlib
    import importlib
    import json
    import shutil
    bench = importlib.import_module("bench")
    traffic, src, stale = bench.committed_traffic(ROOT)
    assert "pmc_traffic.json" in src and (traffic is None) == bool(stale)
    if stale:      # not a failure of the code under test: bench.py will say `traffic_stale` until the passes are redone
        print("NOTE: profiles/rNN_pmc_traffic.json predates the current csrc/gemm_p3.hip: re-run tools/pmc_passes.sh + "
              "tools/pmc_traffic.py")
    else:
        assert traffic > 2.3e9          # at least the algorithmic bytes of the fc1 launch
    fake = tmp_path / "repo"
    (fake / "profiles").mkdir(parents=True)
    (fake / "vit-with-opencl_amd" / "csrc").mkdir(parents=True)
    newest = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))[-1]
    shutil.copy(newest, fake / "profiles" / newest.name)
    kernel = (ROOT / "vit-with-opencl_amd" / "csrc" / "gemm_p3.hip").read_text()
    (fake / "vit-with-opencl_amd" / "csrc" / "gemm_p3.hip").write_text(kernel)
    rec0 = json.loads(newest.read_text())
    rec0["kernel_source_sha256"] = hashlib.sha256(kernel.encode()).hexdigest()
    (fake / "profiles" / newest.name).write_text(json.dumps(rec0))
    assert bench.committed_traffic(fake)[2] is False and bench.committed_traffic(fake)[0] == rec0["traffic_bytes_per_launch"]
    (fake / "vit-with-opencl_amd" / "csrc" / "gemm_p3.hip").write_text(kernel + "\n/* edited */\n")
    assert bench.committed_traffic(fake) [0] is None and bench.committed_traffic(fake)[2] is True
    rec = json.loads(newest.read_text())
    assert len(rec["kernel_source_sha256"]) == 64 and (rec["kernel_source_sha256"] == hashlib.sha256(kernel.encode()).hexdigest()) == (not stale)


def test_bench_starts_its_own_ranks_for_gpus_n(monkeypatch):
    """`python bench.py --gpus N` without a launcher: torch.distributed.run as a CHILD process (nothing re-exec'd),
    rendezvous on 127.0.0.1, the caller's arguments passed through, dmabuf IPC kept in the environment."""
    import importlib
    import subprocess
    bench = importlib.import_module("bench")
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.spawn_ranks(4) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_drop_in_with_zero_images_returns_without_a_device(pkg, capfd):
    """image[0].n == 0: the reference's per-image loop does not run (ViT_opencl.c:926).  The drop-in returns at once --
    on this GPU-less container any device call would have exited the process (VH_CHECK)."""
    b, L = pkg.binding, pkg.lib()
    img = (b.ImageData * 1)()
    img[0].n, img[0].c, img[0].h, img[0].w = 0, 3, 224, 224
    nets = (b.Network * 152)()
    rows = (b.f32p * 1)()
    L.ViT_opencl(img, nets, rows)
    su, fw = C.c_double(-1), C.c_double(-1)
    L.vit_hip_last_call_seconds(C.byref(su), C.byref(fw))
    assert su.value == 0.0 and fw.value == 0.0
