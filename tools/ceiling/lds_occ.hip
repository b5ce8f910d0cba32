// How many 512-thread workgroups share a CU for a given dynamic LDS size (what is the allocation granule of the 160 KiB?):
// hipOccupancyMaxActiveBlocksPerMultiprocessor for a kernel of few registers, LDS sizes around a third and a half of the LDS.
//   hipcc --offload-arch=gfx950 -O2 -o lds_occ lds_occ.hip && ./lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" __global__ void __launch_bounds__(512) probe(unsigned* out)
{
    extern __shared__ unsigned s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (out) out[threadIdx.x] = s[511 - threadIdx.x];
}
// ... and what the hardware does: 3 x 256 workgroups that each hold their LDS for ~100 us; one round if three share a CU, two if not
extern "C" __global__ void __launch_bounds__(512) hold(unsigned* out, long long ticks)
{
    extern __shared__ unsigned s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (out && s[threadIdx.x] == 0xdeadbeefu) out[0] = 1;
}
int main()
{
    {
        hipFuncSetAttribute(reinterpret_cast<const void*>(hold), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int sizes[] = {49408, 52 * 1024, 53248 + 256, 54272, 54613, 54784, 64 * 1024, 80 * 1024, 81920 + 256};
        for (int b : sizes) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(hold, dim3(768), dim3(512), (size_t)b, 0, nullptr, 10000LL);   // 100 MHz clock: 100 us
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("hold: lds %6d B, 768 workgroups x 100 us: %.3f ms\n", b, best);
        }
    }
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int sizes[] = {49408, 52 * 1024, 53248 + 256, 53248 + 512, 53248 + 1024, 54000, 54272, 54528, 54600, 54613, 54784, 55296, 64 * 1024, 80 * 1024, 81920 + 256, 160 * 1024};
    for (int b : sizes) {
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe, 512, (size_t)b);
        printf("lds %6d B: %d workgroups per CU (%s)\n", b, nb, hipGetErrorString(e));
    }
    return 0;
}
