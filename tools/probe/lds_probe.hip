// Probe (GPU box): semantics of the LDS-direct load used by field_update's table windows on gfx950.
// global_load_lds_dwordx4 via __builtin_amdgcn_global_load_lds(src, lds_dst, 16, 0, 0): every lane reads 16 bytes at ITS
// global address (8-byte alignment is enough), the wave writes 64 x 16 contiguous bytes at the wave-uniform LDS address.
// Build and run: hipcc --offload-arch=gfx950 -O3 tools/probe/lds_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
// Expected output: "off N: 0 mismatches" for the three source offsets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double *g, double *out, int off) {
    extern __shared__ double lds[];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_void;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < 8; c += 4)
        __builtin_amdgcn_global_load_lds((glb_void *)(g + off + c * 128 + lane * 2), (lds_void *)(lds + c * 128), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[i];
}
int main() {
    std::vector<double> h(4096); for (int i = 0; i < 4096; ++i) h[i] = i;
    double *g, *o; hipMalloc(&g, 4096 * 8); hipMalloc(&o, 1024 * 8);
    hipMemcpy(g, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    for (int off : {0, 3, 17}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 1024 * 8, 0, g, o, off);
        std::vector<double> r(1024); hipMemcpy(r.data(), o, 1024 * 8, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 1024; ++i) if (r[i] != off + i) { if (bad < 5) printf("off %d: [%d] = %g\n", off, i, r[i]); ++bad; }
        printf("off %d: %d mismatches\n", off, bad);
    }
    return 0;
}
