#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    int* d; hipMalloc(&d, 64 * 4);
    k<<<64, 64>>>(d);
    int h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%d:%x ", i, h[i]);
    printf("\n");
    return 0;
}
