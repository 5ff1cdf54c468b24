/* lab_common.h -- helpers of the ablation labs (tools/p3_lab.hip, tools/mx_lab.hip); not part of the library. */
#ifndef VIT_HIP_LAB_COMMON_H
#define VIT_HIP_LAB_COMMON_H

/* Hold back the SECOND workgroup that lands on a compute unit by `cycles`
 * shader clocks, so that the two co-resident workgroups of a CU run half a tile apart -- one's epilogue (VALU, stores)
 * under the other's K loop (matrix pipe) -- instead of in lockstep.  slots: zeroed [4096] counters, one per (XCC, SE, SH,
 * CU); without it the workgroups lo <= blockIdx.x < hi are the late ones. */
__device__ __forceinline__ void lab_stagger_start(int lo, int hi, int cycles, unsigned *slots)
{
    bool late = (int)blockIdx.x >= lo && (int)blockIdx.x < hi;
    if (slots) {
        __shared__ unsigned arrival;
        if (threadIdx.x == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));           /* HW_REG_HW_ID */
            const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xf;   /* HW_REG_XCC_ID */
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            arrival = atomicAdd(&slots[(xcc << 8) | (se << 5) | (sh << 4) | cu], 1u);
        }
        __syncthreads();
        late = arrival == 1;
    }
    if (late) {
        const long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < cycles)
            __builtin_amdgcn_s_sleep(16);
    }
}

#endif
