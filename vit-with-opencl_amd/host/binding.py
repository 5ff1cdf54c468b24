"""ctypes binding of libvit_hip.so -- test / bench plumbing over the C ABI.

The product is the C library (include/*.h); this module only loads it, mirrors
its structs, and turns non-zero status codes into exceptions.  There is no
Python or CPU implementation of any operator here: if the library or a gfx950
device is missing, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent.parent
LIB_PATH = PKG_DIR / "libvit_hip.so"
CSRC = PKG_DIR / "csrc"

f32p = C.POINTER(C.c_float)
voidp = C.c_void_p

# Every symbol include/kernelHandler.h, include/ViT_opencl.h and include/Network.h declare.
EXPORTS = [
    "vh_device_count", "vh_init", "vh_last_error", "vh_device_name",
    "vh_stream_create", "vh_stream_destroy", "vh_stream_sync", "vh_device_sync",
    "vh_event_create", "vh_event_destroy", "vh_event_record", "vh_stream_wait_event", "vh_event_sync",
    "vh_event_elapsed_ms",
    "vh_malloc", "vh_free", "vh_host_alloc", "vh_host_free", "vh_memset", "vh_h2d", "vh_d2h", "vh_d2d",
    "vh_launch_patch_embed", "vh_launch_layer_norm", "vh_launch_linear", "vh_launch_attention",
    "vh_launch_softmax",
    "vit_config_preset", "vit_config_tokens", "vit_config_num_tensors", "vit_config_tensor_size",
    "ViT_opencl", "vit_hip_last_call_seconds", "vit_hip_create", "vit_hip_destroy", "vit_hip_forward", "vit_hip_forward_device",
    "vit_hip_config", "vit_hip_stream", "vit_hip_max_batch", "vit_hip_weight", "vit_hip_read_tokens",
    "vit_hip_profile_enable", "vit_hip_profile_read", "vit_hip_profile_select", "vit_hip_create_ex", "vit_hip_precision",
    "vit_hip_export_planes", "vit_hip_create_from_planes",
    "vh_patch_embed_workspace", "vh_launch_patch_embed_ws", "vh_patch_planes_k", "vh_launch_conv_weight_planes",
    "vh_launch_patch_embed_planes", "vh_launch_split3_planes", "vh_launch_linear_w3",
    "vh_launch_split2h_planes", "vh_launch_linear_h2", "vh_launch_attention_h2", "vh_launch_attention_f16",
    "vh_launch_absmax",
    "vh_launch_split3_rows", "vh_launch_merge3_rows", "vh_launch_layer_norm_p3", "vh_launch_attention_p3",
    "vh_launch_linear_p3", "vh_launch_split_rows", "vh_launch_merge_rows", "vh_launch_layer_norm_planes",
    "vh_launch_attention_planes_bf16", "vh_launch_linear_planes", "vh_launch_attention_planes",
    "vh_launch_quantize_mx_rows", "vh_launch_quantize_mx_act", "vh_mx_act_scale_bytes", "vh_launch_linear_mx", "vh_launch_layer_norm_mx",
    "vh_launch_attention_planes_f16", "vh_launch_linear_mx_planes_f16", "vh_launch_attention_planes_f16_mx",
    "vh_launch_attention_planes_f16_hd80", "vh_launch_attention_planes_f16_hd80_operand",
    "vh_launch_gather_rows", "vit_hip_set_last_layer_cls_only",
    "vh_launch_conv_weight_planes_parts", "vh_launch_patch_embed_planes3",
    "vh_launch_colsum_planes3", "vh_launch_patch_embed_planes3_norm", "vh_launch_linear_p3_norm", "vh_launch_linear_p3_resid_norm",
    "vh_launch_fold_gamma", "vh_launch_fold_bias", "vh_launch_colsum_operand", "vh_launch_patch_embed_planes_norm",
    "vh_launch_linear_planes_norm", "vh_launch_linear_planes_resid_norm", "vh_launch_linear_mx_norm",
    "vh_launch_linear_mx_resid_norm", "vit_hip_ln_fold",
    "vh_set_device", "vit_hip_create_multi", "vit_hip_forward_multi", "vit_hip_destroy_multi", "vit_hip_multi_devices",
    "vit_hip_multi_ctx", "vit_shard_range", "vit_shard_run", "vit_shard_run_timed", "vit_hip_multi_last_enqueue_ms",
    "vit_hip_forward_device_multi", "vit_hip_device", "vh_set_error",
    "vit_synth_fill", "vit_synth_tensor", "vit_synth_image",
    "load_image_data", "load_weights", "vit_write_image_file", "vit_write_weight_file",
    "vit_write_result_file", "vit_compare_rows",
]


class VitConfig(C.Structure):
    """`vit_config` (include/ViT_opencl.h)."""

    _fields_ = [
        ("img_size", C.c_int), ("patch_size", C.c_int), ("in_chans", C.c_int),
        ("num_classes", C.c_int), ("embed_dim", C.c_int), ("depth", C.c_int),
        ("num_heads", C.c_int), ("mlp_hidden", C.c_int), ("eps", C.c_double),
    ]


class CompareReport(C.Structure):
    """`vit_compare_report` (include/ViT_opencl.h)."""
    _fields_ = [("rows", C.c_int), ("classes", C.c_int), ("max_abs_diff", C.c_double), ("mean_abs_diff", C.c_double),
                ("top1_equal", C.c_int), ("top1_equal_or_near_tie", C.c_int), ("top5_overlap", C.c_double),
                ("nonfinite", C.c_int)]


class ImageData(C.Structure):
    """`ImageData` (include/Network.h; reference Network.h:7-14)."""

    _fields_ = [("n", C.c_int), ("c", C.c_int), ("h", C.c_int), ("w", C.c_int), ("data", f32p)]


class Network(C.Structure):
    """`Network` (include/Network.h; reference Network.h:19-23)."""

    _fields_ = [("data", f32p), ("size", C.c_size_t)]


# int fn(void *arg, int shard, int lo, int hi) -- the callback type of vit_shard_run
SHARD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int)


class VitHipError(RuntimeError):
    pass


def build_library(force: bool = False) -> Path:
    """Compile csrc/ into libvit_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.c")) + list(CSRC.glob("*.h")) + \
        list((PKG_DIR.parent / "include").glob("*.h"))
    stale = (not LIB_PATH.exists()) or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in srcs)
    if force or stale:
        r = subprocess.run(["make", "-C", str(CSRC), "-j8"], capture_output=True, text=True)
        if r.returncode != 0:
            raise VitHipError("building libvit_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load libvit_hip.so (no silent fallback: a missing library is an error)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise VitHipError(f"{LIB_PATH} is missing: run __graft_entry__.build() (make -C {CSRC})")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (same
    # SONAME as /opt/rocm's).  If torch is going to live in this process it must be
    # loaded FIRST so that libvit_hip.so binds to the runtime torch uses -- otherwise
    # torch-allocated HBM and our streams would belong to two different runtimes.
    if os.environ.get("VIT_HIP_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    # VIT_HIP_LIB: a differently built copy of the same library (kernel tuning experiments only)
    L = C.CDLL(os.environ.get("VIT_HIP_LIB") or str(LIB_PATH))
    i, sz = C.c_int, C.c_size_t
    L.vh_last_error.restype = C.c_char_p
    L.vh_device_name.restype = C.c_char_p
    L.vh_init.argtypes = [i]
    L.vh_stream_create.argtypes = [C.POINTER(voidp)]
    L.vh_stream_destroy.argtypes = [voidp]
    L.vh_stream_sync.argtypes = [voidp]
    L.vh_event_create.argtypes = [C.POINTER(voidp)]
    L.vh_event_destroy.argtypes = [voidp]
    L.vh_event_record.argtypes = [voidp, voidp]
    L.vh_stream_wait_event.argtypes = [voidp, voidp]
    L.vh_event_sync.argtypes = [voidp]
    L.vh_event_elapsed_ms.argtypes = [C.POINTER(C.c_float), voidp, voidp]
    L.vh_malloc.argtypes = [C.POINTER(voidp), sz]
    L.vh_free.argtypes = [voidp]
    L.vh_host_alloc.argtypes = [C.POINTER(voidp), sz]
    L.vh_host_free.argtypes = [voidp]
    L.vh_memset.argtypes = [voidp, i, sz, voidp]
    L.vh_h2d.argtypes = [voidp, voidp, sz, voidp]
    L.vh_d2h.argtypes = [voidp, voidp, sz, voidp]
    L.vh_d2d.argtypes = [voidp, voidp, sz, voidp]
    L.vh_launch_patch_embed.argtypes = [voidp] + [voidp] * 6 + [i] * 5
    L.vh_launch_patch_embed_ws.argtypes = [voidp] + [voidp] * 6 + [i] * 5 + [voidp, sz]
    L.vh_patch_embed_workspace.argtypes = [i] * 5
    L.vh_patch_planes_k.argtypes = [i, i]
    L.vh_launch_conv_weight_planes.argtypes = [voidp, voidp, voidp, i, i, i]
    L.vh_launch_patch_embed_planes.argtypes = [voidp] + [voidp] * 6 + [i] * 5 + [voidp, sz]
    L.vh_patch_embed_workspace.restype = sz
    L.vh_launch_layer_norm.argtypes = [voidp] + [voidp] * 4 + [i, i, C.c_long, C.c_long, C.c_double]
    L.vh_launch_linear.argtypes = [voidp] + [voidp] * 4 + [i, i, i, i, voidp]
    L.vh_launch_attention.argtypes = [voidp, voidp, voidp, i, i, i, i]
    L.vh_launch_attention_h2.argtypes = [voidp, voidp, voidp, i, i, i, i]
    L.vh_launch_attention_f16.argtypes = [voidp, voidp, voidp, i, i, i, i]
    L.vh_launch_split3_planes.argtypes = [voidp, voidp, voidp, i, i]
    L.vh_launch_linear_w3.argtypes = [voidp] + [voidp] * 4 + [i, i, i, i, voidp]
    L.vh_launch_split2h_planes.argtypes = [voidp, voidp, voidp, i, i, C.c_float]
    L.vh_launch_linear_h2.argtypes = [voidp, voidp, voidp, C.c_float, voidp, voidp, i, i, i, i, voidp]
    L.vh_launch_softmax.argtypes = [voidp, voidp, voidp, i, i]
    L.vh_launch_split3_rows.argtypes = [voidp, voidp, voidp, i, i]
    L.vh_launch_merge3_rows.argtypes = [voidp, voidp, voidp, i, i]
    L.vh_launch_layer_norm_p3.argtypes = [voidp] + [voidp] * 4 + [i, i, C.c_long, C.c_double]
    L.vh_launch_attention_p3.argtypes = [voidp, voidp, voidp, i, i, i, i]
    L.vh_launch_linear_p3.argtypes = [voidp, voidp, i, voidp, voidp, voidp, i, i, i, i, voidp]
    L.vh_launch_split_rows.argtypes = [voidp, voidp, voidp, i, i, i]
    L.vh_launch_merge_rows.argtypes = [voidp, voidp, voidp, i, i, i]
    L.vh_launch_layer_norm_planes.argtypes = [voidp] + [voidp] * 4 + [i, i, i, C.c_long, C.c_double]
    L.vh_launch_attention_planes_bf16.argtypes = [voidp, voidp, voidp, i, i, i, i]
    L.vh_launch_attention_planes.argtypes = [voidp, voidp, voidp, i, i, i, i]
    L.vh_launch_gather_rows.argtypes = [voidp, voidp, voidp, i, i, i, i, i]
    L.vit_hip_set_last_layer_cls_only.argtypes = [voidp, i]
    L.vh_launch_attention_planes_f16.argtypes = [voidp, voidp, voidp, i, i, i, i, i]
    L.vh_launch_attention_planes_f16_mx.argtypes = [voidp, voidp, voidp, voidp, i, i, i, i]
    L.vh_launch_attention_planes_f16_hd80.argtypes = [voidp, voidp, voidp, i, i, i, i]
    L.vh_launch_attention_planes_f16_hd80_operand.argtypes = [voidp, voidp, voidp, voidp, i, i, i, i, i]
    L.vh_launch_linear_mx_planes_f16.argtypes = [voidp, voidp, voidp, voidp, voidp, voidp, voidp, i, i, i]
    L.vh_launch_quantize_mx_rows.argtypes = [voidp, voidp, voidp, voidp, i, i]
    L.vh_launch_layer_norm_mx.argtypes = [voidp] + [voidp] * 5 + [i, i, C.c_long, C.c_double]
    L.vh_launch_linear_mx.argtypes = [voidp, voidp, voidp, voidp, voidp, voidp, voidp, voidp, i, i, i, i, voidp]
    L.vh_launch_linear_planes.argtypes = [voidp, voidp, i, voidp, voidp, i, voidp, i, i, i, i, voidp]
    L.vh_launch_conv_weight_planes_parts.argtypes = [voidp, voidp, voidp, i, i, i, i]
    L.vh_launch_patch_embed_planes3.argtypes = [voidp] + [voidp] * 6 + [i] * 5 + [voidp, sz]
    L.vh_launch_colsum_planes3.argtypes = [voidp, voidp, voidp, i, i]
    L.vh_launch_patch_embed_planes3_norm.argtypes = [voidp] + [voidp] * 6 + [i] * 5 + [voidp, sz, voidp, voidp]
    L.vh_launch_linear_p3_norm.argtypes = [voidp, voidp, i, voidp, voidp, voidp, voidp, voidp, C.c_double, i, i, i, i]
    L.vh_launch_linear_p3_resid_norm.argtypes = [voidp, voidp, voidp, voidp, voidp, voidp, i, i, i, voidp, voidp]
    L.vh_launch_quantize_mx_act.argtypes = [voidp, voidp, voidp, voidp, i, i]
    L.vh_mx_act_scale_bytes.argtypes = [i, i]
    L.vh_mx_act_scale_bytes.restype = sz
    L.vh_launch_fold_gamma.argtypes = [voidp, voidp, voidp, voidp, i, i]
    L.vh_launch_fold_bias.argtypes = [voidp, voidp, voidp, voidp, voidp, i, i]
    L.vh_launch_colsum_operand.argtypes = [voidp, voidp, voidp, voidp, i, i]
    L.vh_launch_patch_embed_planes_norm.argtypes = [voidp] + [voidp] * 6 + [i] * 5 + [voidp, sz, voidp, voidp, voidp]
    L.vh_launch_linear_planes_norm.argtypes = [voidp, voidp, i, voidp, voidp, voidp, voidp, voidp, C.c_double, i, i, i, i]
    L.vh_launch_linear_planes_resid_norm.argtypes = [voidp, voidp, voidp, voidp, voidp, voidp, i, i, i, voidp, voidp, voidp]
    L.vh_launch_linear_mx_norm.argtypes = [voidp, voidp, voidp, i, voidp, voidp, voidp, voidp, voidp, voidp, voidp, C.c_double, i, i, i, i]
    L.vh_launch_linear_mx_resid_norm.argtypes = [voidp, voidp, voidp, voidp, voidp, voidp, voidp, voidp, i, i, i, voidp, voidp, voidp]
    L.vit_hip_ln_fold.argtypes = [voidp]
    L.vit_config_preset.argtypes = [C.POINTER(VitConfig), C.c_char_p]
    L.vit_config_tokens.argtypes = [C.POINTER(VitConfig)]
    L.vit_config_num_tensors.argtypes = [C.POINTER(VitConfig)]
    L.vit_config_tensor_size.argtypes = [C.POINTER(VitConfig), i]
    L.vit_config_tensor_size.restype = sz
    L.ViT_opencl.argtypes = [C.POINTER(ImageData), C.POINTER(Network), C.POINTER(f32p)]
    L.ViT_opencl.restype = None
    L.vit_hip_last_call_seconds.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.vit_hip_last_call_seconds.restype = None
    L.vit_hip_create.argtypes = [C.POINTER(voidp), C.POINTER(VitConfig), C.POINTER(Network), i, i, i]
    L.vit_hip_create_ex.argtypes = [C.POINTER(voidp), C.POINTER(VitConfig), C.POINTER(Network), i, i, i, i]
    L.vit_hip_precision.argtypes = [voidp]
    L.vit_hip_export_planes.argtypes = [voidp, C.c_char_p]
    L.vit_hip_create_from_planes.argtypes = [C.POINTER(voidp), C.c_char_p, i, i]
    f = C.c_float
    L.vh_launch_absmax.argtypes = [voidp, voidp, sz, voidp]
    L.vit_hip_destroy.argtypes = [voidp]
    L.vit_hip_destroy.restype = None
    L.vh_set_device.argtypes = [i]
    L.vit_hip_create_multi.argtypes = [C.POINTER(voidp), C.POINTER(VitConfig), C.POINTER(Network), i, C.POINTER(i), i, i, i]
    L.vit_hip_forward_multi.argtypes = [voidp, C.POINTER(ImageData), i, f32p, C.POINTER(f32p)]
    L.vit_hip_destroy_multi.argtypes = [voidp]
    L.vit_hip_destroy_multi.restype = None
    L.vit_hip_multi_devices.argtypes = [voidp]
    L.vit_hip_multi_ctx.argtypes = [voidp, i]
    L.vit_hip_multi_ctx.restype = voidp
    L.vit_hip_device.argtypes = [voidp]
    L.vit_hip_forward_device_multi.argtypes = [voidp, C.POINTER(voidp), C.POINTER(i), voidp, voidp]
    L.vh_set_error.argtypes = [i, C.c_char_p]
    L.vit_shard_range.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i)]
    L.vit_shard_range.restype = None
    L.vit_shard_run.argtypes = [i, i, SHARD_FN, voidp]
    L.vit_shard_run_timed.argtypes = [i, i, SHARD_FN, voidp, C.POINTER(C.c_double)]
    L.vit_hip_multi_last_enqueue_ms.argtypes = [voidp, C.POINTER(C.c_double), i]
    L.vit_hip_forward.argtypes = [voidp, C.POINTER(ImageData), i, f32p, C.POINTER(f32p)]
    L.vit_hip_forward_device.argtypes = [voidp, voidp, i, voidp, voidp, voidp]
    L.vit_hip_stream.argtypes = [voidp]
    L.vit_hip_stream.restype = voidp
    L.vit_hip_max_batch.argtypes = [voidp]
    L.vit_hip_weight.argtypes = [voidp, i]
    L.vit_hip_weight.restype = voidp
    L.vit_hip_read_tokens.argtypes = [voidp, i, f32p]
    L.vit_hip_profile_enable.argtypes = [voidp, i]
    L.vit_hip_profile_select.argtypes = [voidp, C.c_uint]
    L.vit_hip_profile_read.argtypes = [voidp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.vit_synth_fill.argtypes = [f32p, sz, C.c_ulonglong, C.c_float, C.c_float]
    L.vit_synth_fill.restype = None
    L.vit_synth_tensor.argtypes = [C.POINTER(VitConfig), i, C.c_ulonglong, f32p]
    L.vit_synth_tensor.restype = None
    L.vit_synth_image.argtypes = [C.POINTER(VitConfig), i, f32p]
    L.vit_synth_image.restype = None
    L.load_image_data.argtypes = [C.c_char_p]
    L.load_image_data.restype = C.POINTER(ImageData)
    L.load_weights.argtypes = [C.c_char_p, C.POINTER(Network), i]
    L.load_weights.restype = None
    L.vit_write_image_file.argtypes = [C.c_char_p, C.POINTER(ImageData), i]
    L.vit_write_weight_file.argtypes = [C.c_char_p, i, C.c_char_p, f32p, sz]
    L.vit_write_result_file.argtypes = [C.c_char_p, C.POINTER(f32p), i, i]
    L.vit_compare_rows.argtypes = [f32p, f32p, i, i, C.c_double, C.POINTER(CompareReport)]
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise VitHipError(f"{what} failed with status {rc}: {lib().vh_last_error().decode()}")


def fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags.c_contiguous, "need contiguous float32"
    return a.ctypes.data_as(f32p)


def preset(name: str) -> VitConfig:
    cfg = VitConfig()
    if lib().vit_config_preset(C.byref(cfg), name.encode()) != 0:
        raise ValueError(f"unknown preset {name}")
    return cfg


def tokens(cfg: VitConfig) -> int:
    return lib().vit_config_tokens(C.byref(cfg))


def synth_weights(cfg: VitConfig, seed_base: int = 0) -> list[np.ndarray]:
    L = lib()
    out = []
    for idx in range(L.vit_config_num_tensors(C.byref(cfg))):
        a = np.empty(L.vit_config_tensor_size(C.byref(cfg), idx), dtype=np.float32)
        L.vit_synth_tensor(C.byref(cfg), idx, seed_base, fptr(a))
        out.append(a)
    return out


def synth_images(cfg: VitConfig, first: int, count: int) -> np.ndarray:
    L = lib()
    a = np.empty((count, cfg.in_chans, cfg.img_size, cfg.img_size), dtype=np.float32)
    for i in range(count):
        L.vit_synth_image(C.byref(cfg), first + i, fptr(a[i]))
    return a


def networks(weights: list[np.ndarray]):
    arr = (Network * len(weights))()
    for i, w in enumerate(weights):
        arr[i].data = fptr(w)
        arr[i].size = w.size
    return arr


def image_array(images: np.ndarray):
    """[n][C][H][W] float32 -> ImageData[n] borrowing the numpy rows (like Network.c:86-90)."""
    n, c, h, w = images.shape
    arr = (ImageData * n)()
    for i in range(n):
        arr[i].n, arr[i].c, arr[i].h, arr[i].w = n, c, h, w
        arr[i].data = fptr(images[i])
    return arr


class DeviceBuffer:
    """A vh_malloc'd float buffer with explicit copies (no hidden host mirror)."""

    def __init__(self, count: int):
        self.count = int(count)
        self.ptr = voidp()
        check(lib().vh_malloc(C.byref(self.ptr), max(self.count, 1) * 4), "vh_malloc")

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a, dtype=np.float32)
        d = cls(a.size)
        check(lib().vh_h2d(d.ptr, a.ctypes.data_as(voidp), a.size * 4, None), "vh_h2d")
        check(lib().vh_device_sync(), "vh_device_sync")
        return d

    def to_numpy(self, shape=None) -> np.ndarray:
        out = np.empty(self.count, dtype=np.float32)
        check(lib().vh_d2h(out.ctypes.data_as(voidp), self.ptr, self.count * 4, None), "vh_d2h")
        check(lib().vh_device_sync(), "vh_device_sync")
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr:
            lib().vh_free(self.ptr)
            self.ptr = voidp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ViTHip:
    """Resident-weights context (vit_hip_create / forward / destroy)."""

    PRECISIONS = {"f32": 0, "bf16": 1, "fp8": 2, "f32_fp16x2": 3}

    def __init__(self, cfg: VitConfig, weights: list[np.ndarray], device: int = 0, max_batch: int = 64,
                 precision: str = "f32"):
        self.cfg = cfg
        self.L = lib()
        self._weights = weights  # keep host arrays alive during create
        self.ctx = voidp()
        self.precision = precision
        rc = self.L.vit_hip_create_ex(C.byref(self.ctx), C.byref(cfg), networks(weights), len(weights),
                                      device, max_batch, self.PRECISIONS[precision])
        check(rc, "vit_hip_create_ex")
        self.max_batch = max_batch
        self.tokens = tokens(cfg)

    @classmethod
    def from_planes(cls, path, device: int = 0, max_batch: int = 64) -> "ViTHip":
        """A context from ONE repacked-weights file (vit_hip_export_planes / vit_hip_create_from_planes)."""
        self = cls.__new__(cls)
        self.L, self._weights, self.ctx = lib(), None, voidp()
        check(self.L.vit_hip_create_from_planes(C.byref(self.ctx), str(path).encode(), device, max_batch),
              "vit_hip_create_from_planes")
        self.L.vit_hip_config.restype = C.POINTER(VitConfig)
        self.L.vit_hip_config.argtypes = [voidp]
        self.cfg = VitConfig.from_buffer_copy(self.L.vit_hip_config(self.ctx).contents)
        self.precision = {v: k for k, v in cls.PRECISIONS.items()}[self.L.vit_hip_precision(self.ctx)]
        self.max_batch, self.tokens = max_batch, tokens(self.cfg)
        return self

    def export_planes(self, path) -> None:
        check(self.L.vit_hip_export_planes(self.ctx, str(path).encode()), "vit_hip_export_planes")

    @property
    def stream(self):
        return self.L.vit_hip_stream(self.ctx)

    def forward(self, images: np.ndarray):
        """Host-pointer path: [n][C][H][W] -> (logits[n][classes], probs[n][classes])."""
        images = np.ascontiguousarray(images, dtype=np.float32)
        n, nc = images.shape[0], self.cfg.num_classes
        logits = np.empty((n, nc), dtype=np.float32)
        probs = np.empty((n, nc), dtype=np.float32)
        rows = (f32p * n)(*[fptr(probs[i]) for i in range(n)])
        check(self.L.vit_hip_forward(self.ctx, image_array(images), n, fptr(logits), rows), "vit_hip_forward")
        return logits, probs

    def forward_device(self, d_images, n: int, d_logits=None, d_probs=None, stream=None):
        """Device-resident path; pointers are ints / c_void_p / DeviceBuffer.ptr."""
        check(self.L.vit_hip_forward_device(self.ctx, d_images, n, d_logits, d_probs, stream),
              "vit_hip_forward_device")

    def sync(self):
        check(self.L.vh_stream_sync(self.stream), "vh_stream_sync")

    def read_tokens(self, n: int) -> np.ndarray:
        out = np.empty((n * self.tokens, self.cfg.embed_dim), dtype=np.float32)
        check(self.L.vit_hip_read_tokens(self.ctx, n, fptr(out)), "vit_hip_read_tokens")
        return out

    OP_NAMES = ["patch_embed", "layer_norm", "qkv_gemm", "attention", "out_proj_gemm", "fc1_gemm",
                "fc2_gemm", "head_gemm", "softmax"]

    def set_last_layer_cls_only(self, on: bool) -> bool:
        """Opt-in: the last layer's output projection and MLP on the class-token rows only (identical logits).
        Returns the previous setting."""
        return bool(self.L.vit_hip_set_last_layer_cls_only(self.ctx, 1 if on else 0))

    def profile_enable(self, max_forwards: int):
        check(self.L.vit_hip_profile_enable(self.ctx, max_forwards), "vit_hip_profile_enable")

    def profile_select(self, names=None):
        """Record only these operators (names from OP_NAMES); None = all."""
        mask = 0 if not names else sum(1 << self.OP_NAMES.index(n) for n in names)
        check(self.L.vit_hip_profile_select(self.ctx, mask), "vit_hip_profile_select")

    def profile_read(self) -> dict[str, tuple[float, int]]:
        """-> {operator: (summed ms, launches)} since the last read."""
        k = len(self.OP_NAMES)
        ms, cnt = (C.c_double * k)(), (C.c_long * k)()
        check(self.L.vit_hip_profile_read(self.ctx, ms, cnt), "vit_hip_profile_read")
        return {name: (ms[j], cnt[j]) for j, name in enumerate(self.OP_NAMES)}

    def close(self):
        if self.ctx:
            self.L.vit_hip_destroy(self.ctx)
            self.ctx = voidp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ViTHipMulti:
    """One replica per device behind one call (vit_hip_create_multi / forward_multi / destroy_multi)."""

    def __init__(self, cfg: VitConfig, weights: list[np.ndarray], devices: list[int], max_batch_per_device: int = 64,
                 precision: str = "f32"):
        self.cfg, self.L = cfg, lib()
        self.handle = voidp()
        devs = (C.c_int * len(devices))(*devices)
        check(self.L.vit_hip_create_multi(C.byref(self.handle), C.byref(cfg), networks(weights), len(weights), devs,
                                          len(devices), max_batch_per_device,
                                          {"f32": 0, "bf16": 1, "fp8": 2, "f32_fp16x2": 3}[precision]), "vit_hip_create_multi")

    def forward(self, images: np.ndarray):
        images = np.ascontiguousarray(images, dtype=np.float32)
        n, nc = images.shape[0], self.cfg.num_classes
        logits = np.empty((n, nc), dtype=np.float32)
        probs = np.empty((n, nc), dtype=np.float32)
        rows = (f32p * n)(*[fptr(probs[i]) for i in range(n)])
        check(self.L.vit_hip_forward_multi(self.handle, image_array(images), n, fptr(logits), rows), "vit_hip_forward_multi")
        return logits, probs

    def forward_device(self, d_images: list, counts: list[int], d_logits_root, d_probs_root=None):
        """Device-resident shards (d_images[g] on device g), logits gathered onto device 0 over RCCL (C side)."""
        n = len(counts)
        ptrs = (voidp * n)(*[p if isinstance(p, voidp) else voidp(p) for p in d_images])
        check(self.L.vit_hip_forward_device_multi(self.handle, ptrs, (C.c_int * n)(*counts), d_logits_root, d_probs_root),
              "vit_hip_forward_device_multi")

    def last_enqueue_ms(self) -> list[float]:
        """Host milliseconds each device's thread spent enqueuing its shard in the last forward_device."""
        ms = (C.c_double * 64)()
        n = self.L.vit_hip_multi_last_enqueue_ms(self.handle, ms, 64)
        return [round(ms[d], 4) for d in range(max(n, 0))]

    def close(self):
        if self.handle:
            self.L.vit_hip_destroy_multi(self.handle)
            self.handle = voidp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous batch shard of rank `rank`: images [lo, hi).  Images never
    interact (the reference processes them strictly one at a time,
    ViT_opencl.c:926), so the batch dimension is the only partition."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total)
