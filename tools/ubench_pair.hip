// Microbenchmark of candidate inner loops for pair_propose (diagnostic tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/ubench tools/ubench_pair.hip && /tmp/ubench
// Every wave owns 64 sorted targets and sweeps NT source tiles (64 sources each) that lie inside the table's
// reach, like the FAST path.  Variants:
//   0  scalar loads of (site*8, sign) pairs, 16 sources per group          (what the library does)
//   1  one coalesced vector load per tile (lane = source), prefetched one tile ahead, v_readlane broadcast
//   2  like 1 but sign via a 64-bit ballot mask (one readlane per source)
//   3  the library's loop: vector-loaded tile (prefetched), spin-partitioned per-wave LDS ring, broadcast ds_read_b128
//   4  like 3 with TWO target tiles per wave (each lane owns two targets): half the ring traffic per pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int TILE = 64;
constexpr uint32_t SG_PLUS = 0x3FF00000u, SG_MINUS = 0xBFF00000u;

__device__ __forceinline__ uint32_t sad_vsv(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t sad3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ double lds_at(uint32_t addr) {
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    return *reinterpret_cast<lds_cdouble *>(addr);
}

template <int VAR>
__global__ __launch_bounds__(256) void k(const uint2 *__restrict__ spair, const double *__restrict__ table, int tlen,
                                         const uint32_t *__restrict__ tpos8, int nt_per_wave, int ntiles_total,
                                         double *out, unsigned long long *cycles) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i <= tlen; i += blockDim.x) lds[i] = table[i];
    __syncthreads();
    typedef __attribute__((address_space(3))) double lds_double;
    const uint32_t tbase = (uint32_t)(size_t)(lds_double *)lds;
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(gw);
    const uint32_t pi8 = tpos8[(size_t)gw * TILE + lane];
    // every wave sweeps nt_per_wave consecutive tiles starting at a wave-dependent offset
    const int first = (wave_u * 7) % (ntiles_total - nt_per_wave);
    double accW[4] = {0, 0, 0, 0}, accS[4] = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (VAR == 0) {
#pragma unroll 1
        for (int t = 0; t < nt_per_wave; ++t) {
            const uint2 *__restrict__ tile = spair + (size_t)(first + t) * TILE;
#pragma unroll 1
            for (int g = 0; g < TILE; g += 16) {
                uint32_t p8[16], sh[16];
                const uint4 *t4 = reinterpret_cast<const uint4 *>(tile + g);
#pragma unroll
                for (int q = 0; q < 8; ++q) { const uint4 v = t4[q]; p8[2*q] = v.x; sh[2*q] = v.y; p8[2*q+1] = v.z; sh[2*q+1] = v.w; }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const double wt = lds_at(sad_vsv(pi8, p8[q], tbase));
                    accW[q & 3] += wt;
                    accS[q & 3] = fma(wt, __hiloint2double((int)sh[q], 0), accS[q & 3]);
                }
            }
        }
    } else if (VAR <= 2) {
        const uint2 *__restrict__ base = spair + (size_t)first * TILE + lane;
        uint2 cur = base[0];
#pragma unroll 1
        for (int t = 0; t < nt_per_wave; ++t) {
            const uint2 nxt = base[(size_t)(t + 1 < nt_per_wave ? t + 1 : t) * TILE];   // prefetch (vmcnt)
            if (VAR == 1) {
#pragma unroll
                for (int q = 0; q < TILE; ++q) {
                    const uint32_t pj8 = __builtin_amdgcn_readlane(cur.x, q);
                    const uint32_t sh = __builtin_amdgcn_readlane(cur.y, q);
                    const double wt = lds_at(sad_vsv(pi8, pj8, tbase));
                    accW[q & 3] += wt;
                    accS[q & 3] = fma(wt, __hiloint2double((int)sh, 0), accS[q & 3]);
                }
            } else {
                const unsigned long long mask = __ballot(cur.y == SG_PLUS);
#pragma unroll
                for (int q = 0; q < TILE; ++q) {
                    const uint32_t pj8 = __builtin_amdgcn_readlane(cur.x, q);
                    const uint32_t sh = (mask >> q) & 1ull ? SG_PLUS : SG_MINUS;
                    const double wt = lds_at(sad_vsv(pi8, pj8, tbase));
                    accW[q & 3] += wt;
                    accS[q & 3] = fma(wt, __hiloint2double((int)sh, 0), accS[q & 3]);
                }
            }
            cur = nxt;
        }
    }
    if (VAR >= 3) {
        uint32_t *ring = reinterpret_cast<uint32_t *>(lds + ((tlen + 2) / 2 * 2)) + (threadIdx.x >> 6) * TILE;
        const uint4 *ring4 = reinterpret_cast<const uint4 *>(ring);
        const uint32_t pi8b = pi8 + 8u * 130u;                 // second target (VAR 4): a tile further right
        double accPb[4] = {0, 0, 0, 0}, accMb[4] = {0, 0, 0, 0};
        const uint32_t *__restrict__ base = reinterpret_cast<const uint32_t *>(spair) + ((size_t)first * TILE + lane) * 2;
        uint32_t nxt = base[0] | (base[1] == SG_PLUS ? 1u : 0u);
#pragma unroll 1
        for (int t = 0; t < nt_per_wave; ++t) {
            const uint32_t word = nxt;
            const size_t tn = (size_t)(t + 1 < nt_per_wave ? t + 1 : t) * TILE * 2;
            nxt = base[tn] | (base[tn + 1] == SG_PLUS ? 1u : 0u);
            const bool plus = word & 1u;
            const unsigned long long pm = __ballot(plus);
            const int nplus = __popcll(pm);
            const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
            ring[plus ? below : nplus + (lane - below)] = word & ~7u;
#pragma unroll 1
            for (int g = 0; g < TILE; g += 16) {
                uint32_t p8[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const uint4 v = ring4[g / 4 + q]; p8[4*q] = v.x; p8[4*q+1] = v.y; p8[4*q+2] = v.z; p8[4*q+3] = v.w; }
                double wa[16], wb[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) { wa[q] = lds_at(sad3(pi8, p8[q], tbase)); if (VAR == 4) wb[q] = lds_at(sad3(pi8b, p8[q], tbase)); }
                const int cut = nplus - g;
                if (cut >= 16) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) { accW[q & 3] += wa[q]; if (VAR == 4) accPb[q & 3] += wb[q]; }
                } else if (cut <= 0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) { accS[q & 3] += wa[q]; if (VAR == 4) accMb[q & 3] += wb[q]; }
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const double sp = q < cut ? 1.0 : 0.0;
                        accW[q & 3] = fma(wa[q], sp, accW[q & 3]); accS[q & 3] = fma(wa[q], 1.0 - sp, accS[q & 3]);
                        if (VAR == 4) { accPb[q & 3] = fma(wb[q], sp, accPb[q & 3]); accMb[q & 3] = fma(wb[q], 1.0 - sp, accMb[q & 3]); }
                    }
                }
            }
        }
        if (VAR == 4) { accW[0] += accPb[0] + accPb[1] + accPb[2] + accPb[3]; accS[0] += accMb[0] + accMb[1] + accMb[2] + accMb[3]; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)gw * TILE + lane] = (accW[0] + accW[1]) + (accW[2] + accW[3]) + ((accS[0] + accS[1]) + (accS[2] + accS[3]));
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { cycles[gw] = t1 - t0; if (gw == 0) { cycles[gridDim.x * 4] = r1 - r0; } }
}

int main(int argc, char **argv) {
    const int tlen = 4001, ntiles = 1564, N = ntiles * TILE, nt_per_wave = argc > 1 ? atoi(argv[1]) : 16;
    std::vector<double> tab(tlen + 1, 0.0);
    for (int t = 0; t < tlen; ++t) tab[t] = 1.0 / (1 + t);
    // sorted particles at density 0.5
    std::vector<uint32_t> pos(N);
    uint32_t p = 0;
    srand(1);
    for (int i = 0; i < N; ++i) { p += 1 + (rand() % 3); pos[i] = p; }
    std::vector<uint2> sp(N);
    for (int i = 0; i < N; ++i) sp[i] = make_uint2(pos[i] << 3, (rand() & 1) ? SG_PLUS : SG_MINUS);
    uint2 *d_sp; double *d_tab, *d_out; uint32_t *d_t; unsigned long long *d_cyc;
    hipMalloc(&d_sp, N * sizeof(uint2)); hipMalloc(&d_tab, tab.size() * 8);
    hipMemcpy(d_sp, sp.data(), N * sizeof(uint2), hipMemcpyHostToDevice);
    hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = (tlen + 2) * 8 + 4 * TILE * 4;
    for (int var : {0, 3, 4})
        for (int wgs_per_cu : {3, 4}) {
            const int nwg = 256 * wgs_per_cu, nwaves = nwg * 4;
            // targets of wave w: tile (w*7 % ...) shifted by ~2000 sites so that all pairs are inside the table
            std::vector<uint32_t> tp((size_t)nwaves * TILE);
            for (int w = 0; w < nwaves; ++w) {
                const int first = (w * 7) % (ntiles - nt_per_wave);
                const uint32_t lo = pos[(size_t)first * TILE], hi = pos[(size_t)(first + nt_per_wave) * TILE - 1];
                const uint32_t centre = (lo + hi) / 2;
                for (int l = 0; l < TILE; ++l) tp[(size_t)w * TILE + l] = (centre - 64 + 2 * l + (rand() & 1)) << 3;
                if ((hi - lo) / 2 + 80 + 130 + 64 >= (uint32_t)tlen) { printf("window too wide\n"); return 1; }
            }
            hipMalloc(&d_t, tp.size() * 4); hipMalloc(&d_out, tp.size() * 8); hipMalloc(&d_cyc, (nwaves + 1) * 8);
            hipMemcpy(d_t, tp.data(), tp.size() * 4, hipMemcpyHostToDevice);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (var == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), lds, 0, d_sp, d_tab, tlen, d_t, nt_per_wave, ntiles, d_out, d_cyc);
                if (var == 1) hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), lds, 0, d_sp, d_tab, tlen, d_t, nt_per_wave, ntiles, d_out, d_cyc);
                if (var == 2) hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(256), lds, 0, d_sp, d_tab, tlen, d_t, nt_per_wave, ntiles, d_out, d_cyc);
                if (var == 3) hipLaunchKernelGGL(k<3>, dim3(nwg), dim3(256), lds, 0, d_sp, d_tab, tlen, d_t, nt_per_wave, ntiles, d_out, d_cyc);
                if (var == 4) hipLaunchKernelGGL(k<4>, dim3(nwg), dim3(256), lds, 0, d_sp, d_tab, tlen, d_t, nt_per_wave, ntiles, d_out, d_cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            std::vector<unsigned long long> cyc(nwaves + 1);
            hipMemcpy(cyc.data(), d_cyc, (nwaves + 1) * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (int i = 0; i < nwaves; ++i) mean += cyc[i]; mean /= nwaves;
            printf("   [clock of wave 0: %.0f cycles / %.0f ticks(100MHz) = %.2f GHz] ", (double)cyc[0], (double)cyc[nwaves], cyc[0] / (double)cyc[nwaves] * 0.1);
            const double wave_sources = (double)nwaves * nt_per_wave * TILE * (var == 4 ? 2 : 1);
            printf("var %d  WG/CU %d  waves/SIMD %d : %.1f us  %.1f cycles/source/wave  chip: %.2f ns per wave-source => %.3g pairs/s\n",
                   var, wgs_per_cu, wgs_per_cu, best * 1e3, mean / (nt_per_wave * TILE), best * 1e6 / wave_sources,
                   wave_sources * 64 / (best * 1e-3));
            hipFree(d_t); hipFree(d_out); hipFree(d_cyc);
        }
    return 0;
}
