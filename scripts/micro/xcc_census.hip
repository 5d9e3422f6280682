// xcc_census.hip -- which XCD does workgroup b run on?  (HW_REG_XCC_ID via s_getreg_b32)
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/xcc_census scripts/micro/xcc_census.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_census(unsigned *out)
{
    const unsigned raw = __builtin_amdgcn_s_getreg(20 | ((16 - 1) << 11));   // 16 bits of HW_REG_XCC_ID
    if (threadIdx.x == 0) out[blockIdx.x] = raw;
}

int main()
{
    for (int blocks : {16, 256, 512}) {
        unsigned *d;
        CK(hipMalloc(&d, blocks * 4));
        hipLaunchKernelGGL(k_census, dim3(blocks), dim3(1024), 65536, 0, d);
        std::vector<unsigned> h(blocks);
        CK(hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost));
        int hist[16] = {0};
        for (unsigned v : h) hist[v & 15]++;
        printf("%d blocks: first 16 raw values:", blocks);
        for (int i = 0; i < 16 && i < blocks; i++) printf(" %x", h[i]);
        printf("\n  histogram of (raw & 15):");
        for (int i = 0; i < 16; i++) printf(" %d", hist[i]);
        printf("\n");
        CK(hipFree(d));
    }
    return 0;
}
