// What do v_permlane16_swap_b32 / v_permlane32_swap_b32 move on gfx950?  Prints, per 16-lane row of the two
// results, which operand and row it came from.  Build: hipcc -O3 --offload-arch=gfx950 permlane_probe.hip -o permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    const unsigned l = threadIdx.x;
    const unsigned a = 0x100 + l, b = 0x200 + l;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[128 + l] = q[0]; o[192 + l] = q[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane16_swap result[0]", "permlane16_swap result[1]", "permlane32_swap result[0]", "permlane32_swap result[1]"};
    for (int i = 0; i < 4; ++i) {
        printf("%s:", names[i]);
        for (int r = 0; r < 4; ++r) {
            unsigned v = h[i * 64 + r * 16];
            printf("  row%d <- %c.row%d", r, (v >> 8) == 1 ? 'a' : 'b', (v & 0xff) / 16);
            for (int j = 1; j < 16; ++j) if (h[i * 64 + r * 16 + j] != v + j) printf("(!lane %d)", j);
        }
        printf("\n");
    }
    return 0;
}
