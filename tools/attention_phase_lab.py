#!/usr/bin/env python3
"""Diagnostic build of the planes attention kernel with in-kernel s_memtime stamps: where a wave's cycles go
(Q.K^T, barrier, softmax, P.V, epilogue, barrier), per wave of workgroup 0.  Writes a stamped COPY of
csrc/attention_p3.hip and a small driver under tools/scratch/ (git-ignored) and compiles them; the library is not
touched and no stamp executes in it.  Never quote the stamped build's run time -- read its shares.
Usage (container): python tools/attention_phase_lab.py      then on the GPU box: tools/scratch/attn_lab [parts: 3 | 1]
`python tools/attention_phase_lab.py h16` does the same for csrc/attention_h16.hip (ViT-H/14: head_dim 80, T = 257, two
rounds of 16-query tiles per item; bf16-planes output): tools/scratch/attn_h16_lab."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "vit-with-opencl_amd" / "csrc"
OUT = ROOT / "tools" / "scratch"


def replace_once(s, old, new):
    assert s.count(old) >= 1, old[:60]
    return s.replace(old, new, 1)


def main_h16():
    OUT.mkdir(exist_ok=True)
    s = (SRC / "attention_h16.hip").read_text()
    s = replace_once(s, "                                                               int T, int E, int H, int n_items, float scale_log2e)\n{",
                     "                                                               int T, int E, int H, int n_items, float scale_log2e, unsigned long long *stamps)\n{\n"
                     "    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;\n"
                     "#define STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[k] += t_ - tlast; tlast = t_; } while (0)")
    s = replace_once(s, "    f32x4 carry[2] = {};", "    tlast = __builtin_amdgcn_s_memtime();\n    f32x4 carry[2] = {};")
    s = replace_once(s, "            if (rnd == 0) {\n                __syncthreads();                         /* A: V of this item has landed */",
                     "            STAMP(0);\n            if (rnd == 0) {\n                __syncthreads();                         /* A: V of this item has landed */")
    s = replace_once(s, "            if (active) {\n                /* row softmax per query", "            STAMP(1);\n            if (active) {\n                /* row softmax per query")
    s = replace_once(s, "                /* O^T = V^T P^T: rows = d of group dt", "                STAMP(2);\n                /* O^T = V^T P^T: rows = d of group dt")
    s = replace_once(s, "                /* O^T register r of d group dt", "                STAMP(3);\n                /* O^T register r of d group dt")
    s = replace_once(s, "        __syncthreads();      /* B: K of the next item has landed; every wave is done with V of this one */\n    }\n}",
                     "        STAMP(4);\n        __syncthreads();\n        STAMP(5);\n    }\n"
                     "    if (lane == 0 && blockIdx.x < 8)\n        for (int k = 0; k < 6; ++k)\n            stamps[(blockIdx.x * 16 + wave) * 8 + k] = tacc[k];\n}")
    s = s.replace("static_cast<unsigned char *>(out_scales), T, E, H, n_items, c);", "static_cast<unsigned char *>(out_scales), T, E, H, n_items, c, g_stamps);")
    s = replace_once(s, "namespace {\n", "unsigned long long *g_stamps = nullptr;\nnamespace {\n")
    (OUT / "attention_h16_stamped.hip").write_text(s)
    (OUT / "attn_h16_lab.hip").write_text(DRIVER_H16)
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-I", str(ROOT / "include"), "-I", str(SRC),
           "-I", str(OUT), str(OUT / "attn_h16_lab.hip"), str(SRC / "kernelHandler.hip"), "-o", str(OUT / "attn_h16_lab")]
    subprocess.run(cmd, check=True)
    print("built", OUT / "attn_h16_lab")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "h16":
        return main_h16()
    OUT.mkdir(exist_ok=True)
    s = (SRC / "attention_p3.hip").read_text()
    s = replace_once(s, "                                                               int T, int E, int H, int n_items, int RB)\n{",
                     "                                                               int T, int E, int H, int n_items, int RB, unsigned long long *stamps)\n{\n"
                     "    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;\n"
                     "#define STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[k] += t_ - tlast; tlast = t_; } while (0)")
    s = replace_once(s, "    for (; item < n_items; item += gridDim.x) {", "    tlast = __builtin_amdgcn_s_memtime();\n    for (; item < n_items; item += gridDim.x) {")
    s = replace_once(s, "        __syncthreads();                                 /* V of this item has landed; K buffer and Q registers are free */",
                     "        STAMP(0);\n        __syncthreads();\n        STAMP(1);")
    s = replace_once(s, "        /* O^T = V^T P^T: rows = d", "        STAMP(2);\n        /* O^T = V^T P^T: rows = d")
    s = replace_once(s, "        /* Output.  A lane holds d = 32dt", "        STAMP(3);\n        /* Output.  A lane holds d = 32dt")
    s = replace_once(s, "        __syncthreads();      /* K and Q of the next item have landed; every wave is done with V of this one */\n    }\n}",
                     "        STAMP(4);\n        __syncthreads();\n        STAMP(5);\n    }\n"
                     "    if (lane == 0 && blockIdx.x < 8)\n        for (int k = 0; k < 6; ++k)\n            stamps[(blockIdx.x * 8 + wave) * 8 + k] = tacc[k];\n}")
    s = s.replace("lds, st, qkv3, out3, out_scales, T, E, H, n_items, RB);", "lds, st, qkv3, out3, out_scales, T, E, H, n_items, RB, g_stamps);")
    s = replace_once(s, "namespace {\n", "unsigned long long *g_stamps = nullptr;\nnamespace {\n")
    (OUT / "attention_p3_stamped.hip").write_text(s)
    (OUT / "attn_lab.hip").write_text(DRIVER)
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-I", str(ROOT / "include"), "-I", str(SRC),
           "-I", str(OUT), str(OUT / "attn_lab.hip"), str(SRC / "kernelHandler.hip"), str(SRC / "gemm_p3.hip"), "-o", str(OUT / "attn_lab")]
    subprocess.run(cmd, check=True)
    print("built", OUT / "attn_lab")


DRIVER = r'''// diagnostic build: phase shares of attention_p3_kernel from in-kernel s_memtime stamps
#include "attention_p3_stamped.hip"
#include <cstdio>
#include <vector>
extern "C" int vh_launch_split_rows(vh_stream_t, const float *, void *, int, int, int);
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void fillr(float *x, size_t n, unsigned seed) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; x[i] = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
int main(int argc, char **argv)
{
    const int parts = argc > 1 ? atoi(argv[1]) : 3;
    const int n = 512, T = 197, E = 768, H = 12, rows = n * T;
    CK(hipSetDevice(0));
    float *qkv; char *q3, *o3;
    CK(hipMalloc(&qkv, (size_t)rows * 3 * E * 4)); CK(hipMalloc(&q3, (size_t)rows * 3 * E * 6)); CK(hipMalloc(&o3, (size_t)rows * E * 6));
    CK(hipMalloc(&g_stamps, 8 * 8 * 8 * 8)); CK(hipMemset(g_stamps, 0, 8 * 8 * 8 * 8));
    fillr<<<(unsigned)(((size_t)rows * 3 * E + 255) / 256), 256>>>(qkv, (size_t)rows * 3 * E, 7u);
    // parts == 1: bf16 bit patterns are read as fp16 -- timing only
    if (vh_launch_split_rows(nullptr, qkv, q3, rows, 3 * E, parts == 3 ? 3 : 1)) { printf("split failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < 10; ++k) {
            int rc = parts == 3 ? vh_launch_attention_planes(nullptr, q3, o3, n, T, E, H) : vh_launch_attention_planes_f16(nullptr, q3, o3, 1, n, T, E, H);
            if (rc) { printf("launch failed\n"); return 1; }
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("parts %d: %.3f ms per launch (stamped build: not the kernel's time)\n", parts, ms / 10);
    }
    std::vector<unsigned long long> h(8 * 8 * 8);
    CK(hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost));
    const char *names[6] = {"Q.K^T", "barrier A", "DMA issue + softmax", "P.V", "epilogue", "barrier B"};
    for (int w = 0; w < 7; ++w) {
        unsigned long long tot = 0; for (int k = 0; k < 6; ++k) tot += h[(0 * 8 + w) * 8 + k];
        printf("workgroup 0 wave %d (24 items, last launch): %llu cycles;", w, tot);
        for (int k = 0; k < 6; ++k) printf("  %s %.1f%%", names[k], 100.0 * h[(0 * 8 + w) * 8 + k] / tot);
        printf("\n");
    }
    return 0;
}
'''

DRIVER_H16 = r'''// diagnostic build: phase shares of attention_h16_kernel from in-kernel s_memtime stamps
#include "attention_h16_stamped.hip"
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void fillh(_Float16 *x, size_t n, unsigned seed) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; x[i] = (_Float16)(((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23))); }
int main()
{
    const int n = 256, T = 257, E = 1280, H = 16, rows = n * T;
    CK(hipSetDevice(0));
    _Float16 *qkv; char *o;
    CK(hipMalloc(&qkv, (size_t)rows * 3 * E * 2)); CK(hipMalloc(&o, (size_t)rows * E * 4));
    CK(hipMalloc(&g_stamps, 8 * 16 * 8 * 8)); CK(hipMemset(g_stamps, 0, 8 * 16 * 8 * 8));
    fillh<<<(unsigned)(((size_t)rows * 3 * E + 255) / 256), 256>>>(qkv, (size_t)rows * 3 * E, 7u);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < 10; ++k)
            if (vh_launch_attention_planes_f16_hd80_operand(nullptr, qkv, o, nullptr, 1, n, T, E, H)) { printf("launch failed\n"); return 1; }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("h16: %.3f ms per launch (stamped build: not the kernel's time)\n", ms / 10);
    }
    std::vector<unsigned long long> h(8 * 16 * 8);
    CK(hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost));
    const char *names[6] = {"Q.K^T", "barrier A / A' (+ DMA issue)", "softmax", "P.V", "output", "barrier B"};
    for (int w = 0; w < 9; ++w) {
        unsigned long long tot = 0; for (int k = 0; k < 6; ++k) tot += h[(0 * 16 + w) * 8 + k];
        printf("workgroup 0 wave %d (16 items, last launch): %llu cycles;", w, tot);
        for (int k = 0; k < 6; ++k) printf("  %s %.1f%%", names[k], 100.0 * h[(0 * 16 + w) * 8 + k] / tot);
        printf("\n");
    }
    return 0;
}
'''

if __name__ == "__main__":
    sys.exit(main())
