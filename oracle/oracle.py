"""ctypes front end of the CPU oracle (oracle/liboracle_port.so).

TEST INFRASTRUCTURE ONLY -- may be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg, never by the product path.

Parity status: pinned for ViT-B/16 against golden vectors produced by the
reference's own ViT_seq.c (see oracle/vit_seq_port.c header and
oracle/make_golden.py); "parity unpinned" for ViT-L/16 and ViT-H/14, for which
the reference has no code.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
PORT_SO = HERE / "liboracle_port.so"
REF_DIR = HERE / "_ref"
REF_HARNESS = REF_DIR / "ref_harness"

f32p = C.POINTER(C.c_float)


class VitConfig(C.Structure):
    """Mirror of `vit_config` (include/ViT_opencl.h)."""

    _fields_ = [
        ("img_size", C.c_int), ("patch_size", C.c_int), ("in_chans", C.c_int),
        ("num_classes", C.c_int), ("embed_dim", C.c_int), ("depth", C.c_int),
        ("num_heads", C.c_int), ("mlp_hidden", C.c_int), ("eps", C.c_double),
    ]


class Network(C.Structure):
    """Mirror of `Network` (include/Network.h; reference Network.h:19-23)."""

    _fields_ = [("data", f32p), ("size", C.c_size_t)]


def build(force: bool = False) -> None:
    """Compile the port (and the reference library/harness where /root/reference exists)."""
    if force or not PORT_SO.exists() or PORT_SO.stat().st_mtime < (HERE / "vit_seq_port.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "port"], check=True, capture_output=True)
    subprocess.run(["make", "-C", str(HERE), "ref"], check=True, capture_output=True)


def _ptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(f32p)


class Oracle:
    def __init__(self, preset: str = "vit_b_16"):
        if not PORT_SO.exists():
            build()
        self.lib = C.CDLL(str(PORT_SO))
        L = self.lib
        L.vit_config_preset.argtypes = [C.POINTER(VitConfig), C.c_char_p]
        L.vit_config_tokens.argtypes = [C.POINTER(VitConfig)]
        L.vit_config_num_tensors.argtypes = [C.POINTER(VitConfig)]
        L.vit_config_tensor_size.argtypes = [C.POINTER(VitConfig), C.c_int]
        L.vit_config_tensor_size.restype = C.c_size_t
        L.vit_synth_tensor.argtypes = [C.POINTER(VitConfig), C.c_int, C.c_ulonglong, f32p]
        L.vit_synth_image.argtypes = [C.POINTER(VitConfig), C.c_int, f32p]
        L.vit_synth_fill.argtypes = [f32p, C.c_size_t, C.c_ulonglong, C.c_float, C.c_float]
        L.port_forward_image.argtypes = [C.POINTER(VitConfig), f32p, C.POINTER(Network), f32p, f32p, f32p, C.c_int]
        L.port_conv2d.argtypes = [C.POINTER(VitConfig), f32p, f32p, f32p, f32p]
        L.port_tokens.argtypes = [C.POINTER(VitConfig), f32p, f32p, f32p, f32p]
        L.port_layer_norm.argtypes = [C.POINTER(VitConfig), f32p, f32p, f32p, f32p, C.c_int]
        L.port_linear.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int, f32p, f32p]
        L.port_mha.argtypes = [C.POINTER(VitConfig), f32p, f32p, f32p, f32p, f32p, f32p, C.c_int]
        L.port_attention.argtypes = [C.POINTER(VitConfig), f32p, f32p, C.c_int]
        L.port_mlp.argtypes = [C.POINTER(VitConfig), f32p, f32p, f32p, f32p, f32p, f32p, C.c_int]
        L.port_encoder.argtypes = [C.POINTER(VitConfig), f32p, f32p, C.POINTER(Network), C.c_int]
        L.port_softmax.argtypes = [f32p, f32p, C.c_int]
        L.port_gelu.argtypes = [C.c_float]
        L.port_gelu.restype = C.c_float
        self.cfg = VitConfig()
        if L.vit_config_preset(C.byref(self.cfg), preset.encode()) != 0:
            raise ValueError(f"unknown preset {preset}")

    # ---- shapes -----------------------------------------------------------------
    @property
    def tokens(self) -> int:
        return self.lib.vit_config_tokens(C.byref(self.cfg))

    @property
    def num_tensors(self) -> int:
        return self.lib.vit_config_num_tensors(C.byref(self.cfg))

    def tensor_size(self, idx: int) -> int:
        return self.lib.vit_config_tensor_size(C.byref(self.cfg), idx)

    # ---- synthetic data ---------------------------------------------------------
    def synth_weights(self, seed_base: int = 0) -> list[np.ndarray]:
        out = []
        for i in range(self.num_tensors):
            a = np.empty(self.tensor_size(i), dtype=np.float32)
            self.lib.vit_synth_tensor(C.byref(self.cfg), i, seed_base, _ptr(a))
            out.append(a)
        return out

    def synth_image(self, index: int) -> np.ndarray:
        c = self.cfg
        a = np.empty((c.in_chans, c.img_size, c.img_size), dtype=np.float32)
        self.lib.vit_synth_image(C.byref(c), index, _ptr(a))
        return a

    def synth_fill(self, count: int, seed: int, scale: float, offset: float) -> np.ndarray:
        a = np.empty(count, dtype=np.float32)
        self.lib.vit_synth_fill(_ptr(a), count, seed, scale, offset)
        return a

    @staticmethod
    def networks(weights: list[np.ndarray]):
        arr = (Network * len(weights))()
        for i, w in enumerate(weights):
            arr[i].data = _ptr(w)
            arr[i].size = w.size
        return arr

    # ---- model ------------------------------------------------------------------
    def forward(self, image: np.ndarray, weights: list[np.ndarray], stop_after_layers: int = -1):
        """-> (logits[classes], probs[classes], tokens[T][E]) for one CHW image."""
        c = self.cfg
        image = np.ascontiguousarray(image, dtype=np.float32)
        logits = np.zeros(c.num_classes, dtype=np.float32)
        probs = np.zeros(c.num_classes, dtype=np.float32)
        toks = np.zeros((self.tokens, c.embed_dim), dtype=np.float32)
        self.lib.port_forward_image(C.byref(c), _ptr(image), self.networks(weights), _ptr(logits),
                                    _ptr(probs), _ptr(toks), stop_after_layers)
        return logits, probs, toks

    # ---- stages -----------------------------------------------------------------
    def conv2d(self, image, w, b):
        c = self.cfg
        g = c.img_size // c.patch_size
        out = np.empty((c.embed_dim, g, g), dtype=np.float32)
        self.lib.port_conv2d(C.byref(c), _ptr(np.ascontiguousarray(image)), _ptr(out), _ptr(w), _ptr(b))
        return out

    def tokens_from_conv(self, conv, cls, pos):
        out = np.empty((self.tokens, self.cfg.embed_dim), dtype=np.float32)
        self.lib.port_tokens(C.byref(self.cfg), _ptr(np.ascontiguousarray(conv)), _ptr(out), _ptr(cls), _ptr(pos))
        return out

    def layer_norm(self, x, w, b):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        self.lib.port_layer_norm(C.byref(self.cfg), _ptr(x), _ptr(out), _ptr(w), _ptr(b), x.shape[0])
        return out

    def linear(self, x, w, b, out_features):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty((x.shape[0], out_features), dtype=np.float32)
        self.lib.port_linear(_ptr(x), _ptr(out), x.shape[0], x.shape[1], out_features, _ptr(w), _ptr(b))
        return out

    def mha(self, x, in_w, in_b, out_w, out_b):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        self.lib.port_mha(C.byref(self.cfg), _ptr(x), _ptr(out), _ptr(in_w), _ptr(in_b), _ptr(out_w), _ptr(out_b), x.shape[0])
        return out

    def attention(self, qkv):
        qkv = np.ascontiguousarray(qkv, dtype=np.float32)
        out = np.empty((qkv.shape[0], self.cfg.embed_dim), dtype=np.float32)
        self.lib.port_attention(C.byref(self.cfg), _ptr(qkv), _ptr(out), qkv.shape[0])
        return out

    def mlp(self, x, w1, b1, w2, b2):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        self.lib.port_mlp(C.byref(self.cfg), _ptr(x), _ptr(out), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), x.shape[0])
        return out

    def encoder(self, x, layer_weights: list[np.ndarray]):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        self.lib.port_encoder(C.byref(self.cfg), _ptr(x), _ptr(out), self.networks(layer_weights), x.shape[0])
        return out

    def softmax(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        self.lib.port_softmax(_ptr(x), _ptr(out), x.size)
        return out

    def gelu(self, xs):
        return np.array([self.lib.port_gelu(float(v)) for v in xs], dtype=np.float32)


# ---- the real reference, where its build is present -----------------------------

def have_reference() -> bool:
    return REF_HARNESS.exists() and os.access(REF_HARNESS, os.X_OK)


def read_records(path) -> dict[str, np.ndarray]:
    """Parse a ref_harness output file: {name[32], uint64 count, float32 data}*."""
    out = {}
    raw = Path(path).read_bytes()
    off = 0
    while off < len(raw):
        name = raw[off:off + 32].split(b"\0", 1)[0].decode()
        count = int(np.frombuffer(raw, dtype=np.uint64, count=1, offset=off + 32)[0])
        out[name] = np.frombuffer(raw, dtype=np.float32, count=count, offset=off + 40).copy()
        off += 40 + 4 * count
    return out


def run_reference(mode: str, *args, out_path=None, timeout=3600) -> dict[str, np.ndarray] | str:
    """Run oracle/_ref/ref_harness (the reference's unmodified ViT_seq.c)."""
    cmd = [str(REF_HARNESS), mode, *map(str, args)]
    if out_path is not None:
        cmd.append(str(out_path))
    r = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=timeout)
    if out_path is not None:
        return read_records(out_path)
    return r.stderr
