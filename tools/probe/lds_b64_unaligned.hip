// Probe (GPU box): does ds_read_b64 at a 4-byte-aligned (not 8-byte-aligned) LDS address return the right two dwords on gfx950,
// and what does it cost next to the aligned read and to ds_read_b32?  (tile_step's windowed sweep would read the table entries of
// two adjacent sites per lane with one ds_read_b64; the entry pair is 8-byte aligned for one parity of x - p only.)
// Build here: hipcc --offload-arch=gfx950 -O3 tools/probe/lds_b64_unaligned.hip -o tools/probe/lds_b64_unaligned.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>   // 0: b32, 1: b64 aligned, 2: b64 at +4 bytes
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned long long *cyc, int iters) {
    __shared__ unsigned lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (unsigned)i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    typedef __attribute__((address_space(3))) const unsigned lds_u;
    typedef __attribute__((address_space(3))) const unsigned long long lds_u64;
    unsigned acc = 0;
    const unsigned base = (unsigned)(size_t)(lds_u *)lds;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned off = (unsigned)(it & 15) * 256u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {
                const unsigned a = base + off + (unsigned)lane * 4u + (unsigned)u * 1024u;
                acc += *reinterpret_cast<lds_u *>(a);
            } else {
                const unsigned a = base + off + (unsigned)lane * 8u + (unsigned)u * 2048u + (MODE == 2 ? 4u : 0u);
                unsigned long long v;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
                acc += (unsigned)v + (unsigned)(v >> 32) * 3u;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_pipe(unsigned *out, unsigned long long *cyc, int iters) {   // 8 reads in flight, one wait
    __shared__ unsigned lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (unsigned)i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    typedef __attribute__((address_space(3))) const unsigned lds_u;
    typedef __attribute__((address_space(3))) const unsigned long long lds_u64;
    unsigned acc = 0;
    const unsigned base = (unsigned)(size_t)(lds_u *)lds;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned off = (unsigned)(it & 15) * 256u;
        if (MODE == 0) {
            unsigned v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<lds_u *>(base + off + (unsigned)lane * 4u + (unsigned)u * 1024u);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        } else {
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<lds_u64 *>(base + off + (unsigned)lane * 8u + (unsigned)u * 2048u + (MODE == 2 ? 4u : 0u));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (unsigned)v[u] + (unsigned)(v[u] >> 32) * 3u;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned *o; unsigned long long *c;
    hipMalloc(&o, 1024 * 256 * 4); hipMalloc(&c, 1024 * 8);
    const int iters = 2000, nb = 768;                 // 3 workgroups per CU
    // correctness of the misaligned read
    {
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, o, c, 1);
        std::vector<unsigned> r(256); hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 256; ++t) {
            unsigned want = 0; const int lane = t & 63;
            for (int u = 0; u < 8; ++u) { const unsigned i = (lane * 8u + u * 2048u + 4u) / 4u; want += i * 2654435761u + (i + 1) * 2654435761u * 3u; }
            bad += r[t] != want;
        }
        printf("ds_read_b64 at 4-byte alignment: %d of 256 lanes wrong\n", bad);
    }
    auto run = [&](auto kern, const char *name, int bytes_per_lane) {
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, o, c, iters);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, o, c, iters);
        std::vector<unsigned long long> cy(nb); hipMemcpy(cy.data(), c, nb * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto x : cy) m += (double)x; m /= nb;
        printf("%-28s %8.1f cycles (s_memtime, 100 MHz ticks x 24?) per 8 wave-reads; %d B/lane\n", name, m / iters, bytes_per_lane);
    };
    run(k<0>, "b32 serial", 4); run(k<1>, "b64 aligned serial", 8); run(k<2>, "b64 +4 serial", 8);
    run(k_pipe<0>, "b32 8 in flight", 4); run(k_pipe<1>, "b64 aligned 8 in flight", 8); run(k_pipe<2>, "b64 +4 8 in flight", 8);
    return 0;
}
