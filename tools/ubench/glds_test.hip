// Probe: __builtin_amdgcn_global_load_lds (dwordx4) semantics on gfx950.
// Each lane gives its own global source; the LDS destination is wave-uniform base + lane*16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ float4 g_zero16 = {0.f, 0.f, 0.f, 0.f};

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

__global__ void k(const float *src, float *dst, int n4) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // 2 slots per thread: slot = j*256 + tid ; reversed source order to prove per-lane source
    for (int j = 0; j < 2; ++j) {
        const int slot = j * 256 + tid;
        const float *s = (slot % 7 == 3) ? reinterpret_cast<const float *>(&g_zero16)
                                         : src + 4 * (n4 - 1 - slot);
        float *wave_base = smem + 4 * (j * 256 + wave * 64);  // wave-uniform
        __builtin_amdgcn_global_load_lds((glb_ptr_t)s, (lds_ptr_t)wave_base, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int j = 0; j < 2; ++j) {
        const int slot = j * 256 + tid;
        // read a slot written by another wave
        const int other = (slot + 64) % 512;
        float4 v = *reinterpret_cast<float4 *>(smem + 4 * other);
        *reinterpret_cast<float4 *>(dst + 4 * other) = v;
    }
    (void)lane;
}

int main() {
    const int n4 = 512;
    std::vector<float> h(4 * n4), o(4 * n4, -1.f);
    for (int i = 0; i < 4 * n4; ++i) h[i] = i;
    float *d, *e;
    hipMalloc(&d, h.size() * 4); hipMalloc(&e, h.size() * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4 * n4 * 4, 0, d, e, n4);
    hipError_t err = hipDeviceSynchronize();
    hipMemcpy(o.data(), e, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int slot = 0; slot < n4; ++slot)
        for (int q = 0; q < 4; ++q) {
            float want = (slot % 7 == 3) ? 0.f : h[4 * (n4 - 1 - slot) + q];
            if (o[4 * slot + q] != want) { if (bad < 5) printf("slot %d q %d got %f want %f\n", slot, q, o[4 * slot + q], want); ++bad; }
        }
    printf("glds probe: err=%d bad=%d\n", (int)err, bad);
    return bad != 0;
}
