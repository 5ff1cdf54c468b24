// Probe: which hardware-ID fields distinguish the two workgroups co-resident on a CU?
// Diagnostic only (not part of the product).  hipcc --offload-arch=gfx950 tools/hwid_probe.hip -o /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin)
{
    extern __shared__ float smem[];
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, offset 0, size 32
        unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
        unsigned lds = __builtin_amdgcn_s_getreg((6) | (0 << 6) | (31 << 11));  // HW_REG_LDS_ALLOC
        unsigned long long t = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = lds;
        out[blockIdx.x * 4 + 3] = (unsigned)t;
    }
    smem[threadIdx.x] = threadIdx.x;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
    __syncthreads();
    if (smem[threadIdx.x] < 0) out[0] = 0;
}
int main()
{
    const int nb = 1536, lds = 73728;
    unsigned* d; hipMalloc(&d, nb * 16);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), lds, 0, d, 200);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4); hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    unsigned t0 = h[3];
    for (int b = 0; b < nb; ++b) if ((int)(h[b*4+3] - t0) < 0) t0 = h[b*4+3];
    std::map<unsigned, std::vector<int>> percu;
    for (int b = 0; b < nb; ++b) {
        unsigned hw = h[b*4], xcc = h[b*4+1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, tg = (hw >> 16) & 0xf, wave = hw & 0xf, simd = (hw >> 4) & 3;
        if (b < 48 || (b >= 512 && b < 530))
            printf("b=%4d hw=%08x xcc=%u se=%u sh=%u cu=%2u tg=%u simd=%u wave=%u lds=%08x t=%u\n", b, hw, xcc, se, sh, cu, tg, simd, wave, h[b*4+2], (h[b*4+3]-t0));
        if (b < 512) percu[(xcc << 12) | (se << 5) | (sh << 4) | cu].push_back(b);
    }
    printf("distinct CU keys among first 512 blocks: %zu\n", percu.size());
    int hist[8] = {0}; for (auto& kv : percu) hist[kv.second.size() < 7 ? kv.second.size() : 7]++;
    for (int i = 0; i < 8; ++i) printf("  CUs with %d first-round blocks: %d\n", i, hist[i]);
    int shown = 0;
    for (auto& kv : percu) { if (shown++ > 12) break; printf("  key %05x:", kv.first); for (int b : kv.second) printf(" b%d(tg=%u,lds=%x)", b, (h[b*4]>>16)&0xf, h[b*4+2]); printf("\n"); }
    return 0;
}
