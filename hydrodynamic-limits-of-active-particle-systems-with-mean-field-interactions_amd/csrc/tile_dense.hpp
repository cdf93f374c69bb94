// tile_dense.hpp -- the step of a tiles handle whose field update is the convolution of ntt_conv.hpp (included by aps_hip.hip inside
// its anonymous namespace, after tile_step.hpp).
//
// Hot path replaced: step_gillespie (PARTICLE_solver_CLASS.py:254-448) in the synchronous scheme of DESIGN.md 3, on the
// site-centric state of tile_step.hpp.  With the convolution, {W, S} of every site are complete when the step starts and the
// step's field changes go into the coefficient signals: nothing of tile_step's sweep (deposit lists, table, accumulators, the
// reduction over the waves) is left, and what remains -- cells and field in, proposals, exclusion, cells and coefficients out --
// is this kernel: two barriers, registers for eight waves per SIMD.
//   0  the cells of the frame (TD_SITES sites + one either side) -> LDS; the occupied ones of the valid frame -> one particle list
//   1  a lane per particle: Philox, {W, S} of its site, occupancy of the three sites around it -> proposal byte
//   2  a lane per owned site: exclusion (tile_step's rules, same code), new cell(s), the event's deposits as atomic adds into
//      c_W, c_S at site + Rt and -- within the table's reach of a wall -- at the mirror site (ntt_conv.hpp)
// The frames are this kernel's own (the cells are indexed by site, not by tile): TD_OWN owned sites, two halo sites either side.
#pragma once

#ifndef APS_TD_SITES
#define APS_TD_SITES 1024                  /* tuning builds: sites per frame */
#endif
constexpr int TD_SITES = APS_TD_SITES, TD_OWN = TD_SITES - 4;
__host__ __device__ inline size_t td_lds_cells(int K) { return ((size_t)(TD_SITES + 2) * K * 4 + 7) / 8 * 8; }
__host__ __device__ inline size_t td_lds_bytes(int K) { return td_lds_cells(K) + (size_t)TD_SITES * K * 8 + ((size_t)TD_SITES * K + 15) / 16 * 16 + 16; }
inline int td_tiles(int L) { return (L + TD_OWN - 1) / TD_OWN; }

// F32: {W, S} are int2 in units of 2^-q (32-bit field) / double2 (binary64 field: the ratio S / W is all a proposal needs of them)
#ifdef APS_TD_WAVES                        /* tuning builds: pin the waves per SIMD the register allocator aims for */
#define TD_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(APS_TD_WAVES, APS_TD_WAVES)))
#else
#define TD_WAVES_ATTR
#endif
template <bool K1, bool F32>
__global__ TD_WAVES_ATTR __launch_bounds__(FU_THREADS) void tile_dense(const TileArgs a) {
    using WS = typename TsField<F32>::ws_t;
    extern __shared__ double lds[];
    const int L = a.L, K = K1 ? 1 : a.K;
    char *lds_c = reinterpret_cast<char *>(lds);
    uint32_t *cellL = reinterpret_cast<uint32_t *>(lds_c);                                   // [(TD_SITES + 2) K]: frame positions -1 .. TD_SITES
    uint2 *plist = reinterpret_cast<uint2 *>(lds_c + td_lds_cells(K));                       // [TD_SITES K] {pos | k << 16, cell}
    uint8_t *propL = reinterpret_cast<uint8_t *>(plist + (size_t)TD_SITES * K);              // [TD_SITES K]
    int *misc = reinterpret_cast<int *>(propL + ((size_t)TD_SITES * K + 15) / 16 * 16);      // 0: particles on the frame
    const int t = threadIdx.x, lane = t & 63, e = blockIdx.y, tile = (int)blockIdx.x;
    const int own0 = tile * TD_OWN, own_n = min(TD_OWN, L - own0), nfr = own_n + 4;          // owned sites, valid frame positions
    const int x0 = own0 - 2;                                                                 // site of frame position 0
    const uint32_t *__restrict__ cell_e = a.cell_in + (size_t)e * L * K;
    const WS *__restrict__ ws_e = reinterpret_cast<const WS *>(a.ws_in) + (size_t)e * L;
    const bool torus = a.periodic != 0;                        // (L is far longer than a frame then: ntt_setup)
    auto frame_site = [&](int i) -> int {                      // site of frame position i (-1 .. nfr), or -1: beyond a wall / the valid frame
        if (i < -1 || i > nfr) return -1;
        int s = x0 + i;
        if (torus) { s = s < 0 ? s + L : (s >= L ? s - L : s); return s; }
        return (s < 0 || s >= L) ? -1 : s;
    };
    if (t == 0) misc[0] = 0;
    for (int i = t; i < (TD_SITES * K + 3) / 4; i += FU_THREADS) reinterpret_cast<uint32_t *>(propL)[i] = 0u;   // EV_NONE everywhere
    const unsigned long long step = a.stepw[a.par];
    const double beta = a.beta[e];
    if (tile == 0 && e == 0 && t == 0) a.stepw[a.par ^ 1] = step + 1ull;                     // nobody reads that word during this step
    __syncthreads();
    // ---------------------------------------------------------------- 0  cells, particle list
    const int ncell = (TD_SITES + 2) * K;
    auto append = [&](const uint32_t v, const int pos, const int k) {   // an occupied cell of the valid frame -> the particle list
        const bool occ = v != CELL_EMPTY && pos >= 0 && pos < nfr;
        const unsigned long long mm = __ballot(occ);
        int base = 0;
        if (lane == 0 && mm) base = atomicAdd(&misc[0], __popcll(mm));
        base = __builtin_amdgcn_readfirstlane(base);
        if (occ) plist[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u))] = make_uint2((uint32_t)pos | ((uint32_t)k << 16), v);
    };
    if constexpr (K1) {                                        // every cell word of the frame requested before the first is looked at
        constexpr int NR = (TD_SITES + 2 + FU_THREADS - 1) / FU_THREADS;
        uint32_t creg[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int c = u * FU_THREADS + t, s = c < ncell ? frame_site(c - 1) : -1;
            creg[u] = cell_e[(unsigned)max(s, 0)];
            if (s < 0) creg[u] = CELL_EMPTY;
        }
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int c = u * FU_THREADS + t;
            if (c < ncell) cellL[c] = creg[u];
            append(creg[u], c - 1, 0);
        }
    } else {
        for (int c0 = 0; c0 < ncell; c0 += FU_THREADS) {
            const int c = c0 + t;
            const int pos = c / K - 1, k = c - (pos + 1) * K;
            const int s = c < ncell ? frame_site(pos) : -1;
            uint32_t v = cell_e[(unsigned)max(s, 0) * (unsigned)K + (unsigned)k];
            if (s < 0) v = CELL_EMPTY;
            if (c < ncell) cellL[c] = v;
            append(v, pos, k);
        }
    }
    __syncthreads();
    const int n_part = misc[0];
    auto occ_at = [&](int pos) -> int {                        // occupancy of frame position pos (-1 .. nfr)
        if (K1) return cellL[pos + 1] != CELL_EMPTY;
        int n = 0;
        for (int k = 0; k < K; ++k) n += cellL[(pos + 1) * K + k] != CELL_EMPTY;
        return n;
    };
    // ---------------------------------------------------------------- 1  proposals, a lane per particle
    {
        const Model M = *a.model;                              // uniform address: scalar loads
        for (int j = t; j < n_part; j += FU_THREADS) {         // (two particles a lane and round, both requests out first: slower -- 81 VGPRs, five waves)
            const uint2 pc = plist[j];
            const int pos = (int)(pc.x & 0xFFFFu), k = (int)(pc.x >> 16), s = frame_site(pos);
            const WS f = ws_e[(unsigned)s];
            uint32_t x[4];
            philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), pc.y & CELL_ID, (uint32_t)(a.ens_base + e), a.seed_lo, a.seed_hi, x);
            const bool anch = a.anchor ? a.anchor[s] != 0 : false;
            // S / W does not care about the unit (2^-q in the integer field; the binary64 quotient of two integers times one power of two is that of the integers)
            propL[pos * K + k] = decide_proposal(M, anch, s, (pc.y & CELL_PLUS) ? 1 : -1, (pc.y & CELL_BOUND) != 0, clip_field((double)f.y, (double)f.x), beta,
                                                 occ_at(pos), occ_at(pos - 1), occ_at(pos + 1), x);
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- 2  exclusion, new cells of the owned sites, coefficients
    auto rank_at = [&](int j, uint32_t id) -> int {            // proposers of frame site j with an id below `id` (they sit on j - 1 and j + 1)
        int n = 0;
        for (int k = 0; k < K; ++k) {
            const uint32_t cl_ = cellL[j * K + k], cr_ = cellL[(j + 2) * K + k];
            const int el = propL[(j - 1) * K + k] & 7, er = propL[(j + 1) * K + k] & 7;
            n += (cl_ != CELL_EMPTY && (el == EV_RIGHT || el == EV_FWD) && (cl_ & CELL_ID) < id);
            n += (cr_ != CELL_EMPTY && er == EV_LEFT && (cr_ & CELL_ID) < id);
        }
        return n;
    };
    auto cap_at = [&](int j) -> int { const int c = K - occ_at(j); return c < 1 ? 1 : (c > 32 ? 32 : c); };
    uint32_t *cell_o = a.cell_out + (size_t)e * L * K;
    int *const cw_sig = a.dense + ((size_t)e << (a.dense_m + 1)), *const cs_sig = cw_sig + ((size_t)1 << a.dense_m);
    const int Rt = a.dense_rt;
    // a field change at site s, with its image: the mirror site beyond a wall (-1 - s, 2 L - 1 - s), or the same site one period on (s + L, s - L)
    auto emit = [&](const int s, const int cw, const int cs) {
        const int at = s + Rt;
        if (cw) atomicAdd(cw_sig + at, cw);
        atomicAdd(cs_sig + at, cs);
        const int img = s < Rt ? (torus ? at + L : Rt - 1 - s) : (s >= L - Rt ? (torus ? at - L : 2 * L - 1 - s + Rt) : -1);
        if (img >= 0) { if (cw) atomicAdd(cw_sig + img, cw); atomicAdd(cs_sig + img, cs); }
    };
    // the event of the particle `c` on site s: returns whether it is still on the site afterwards (c updated);
    // hop_granted: the exclusion rule let its hop to frame position j through
    auto own_event = [&](uint32_t &c, const int ev, const int s, const int j, const bool hop_granted) -> bool {
        const int sgn = (c & CELL_PLUS) ? 1 : -1;
        bool stays = true;
        if (ev == EV_LEFT || ev == EV_RIGHT || ev == EV_FWD) {
            if (hop_granted) { stays = false; emit(s, -1, -sgn); emit(frame_site(j), 1, sgn); }
        } else if (ev == EV_BIND) c |= CELL_BOUND;
        else if (ev == EV_UNBIND) c &= ~CELL_BOUND;
        else if (ev == EV_FLIP) { c ^= CELL_PLUS; emit(s, 0, -2 * sgn); }
        else if (ev == EV_EXIT) {
            stays = false;
            const TileRare R = *a.rare;
            const unsigned kx = atomicAdd(&R.n_exit[e], 1u);
            if ((int)kx < R.exit_cap) {
                double *row = R.exit_log + ((size_t)e * R.exit_cap + kx) * 3;
                row[0] = (double)step; row[1] = (double)s; row[2] = (double)(c & CELL_ID);
            }
            R.src[(size_t)e * R.Npad + R.slot_of[(size_t)e * R.N + (c & CELL_ID)]] =
                (uint32_t)s | DEAD_BIT | ((c & CELL_PLUS) ? SPIN_BIT : 0u) | ((c & CELL_BOUND) ? BOUND_BIT : 0u);
            emit(s, -1, -sgn);
        }
        return stays;
    };
    for (int xi = 2 + t; xi < 2 + own_n; xi += FU_THREADS) {
        const int s = x0 + xi;
        uint32_t *out = cell_o + (unsigned)s * (unsigned)K;
        if constexpr (K1) {
            // one cell per site: the window of five sites of tile_step's K = 1 path (a hop is only proposed into a site that was empty)
            const uint32_t cm2 = cellL[xi - 1], cm1 = cellL[xi], c0 = cellL[xi + 1], cp1 = cellL[xi + 2], cp2 = cellL[xi + 3];
            const int pm2 = propL[xi - 2] & 7, pm1 = propL[xi - 1] & 7, p0 = propL[xi] & 7, pp1 = propL[xi + 1] & 7, pp2 = xi + 2 < TD_SITES ? propL[xi + 2] & 7 : 0;
            const bool from_l2 = cm2 != CELL_EMPTY && (pm2 == EV_RIGHT || pm2 == EV_FWD);
            const bool from_l1 = cm1 != CELL_EMPTY && (pm1 == EV_RIGHT || pm1 == EV_FWD);
            const bool from_r1 = cp1 != CELL_EMPTY && pp1 == EV_LEFT;
            const bool from_r2 = cp2 != CELL_EMPTY && pp2 == EV_LEFT;
            uint32_t newc = CELL_EMPTY;
            if (c0 != CELL_EMPTY) {
                uint32_t c = c0;
                const bool left = p0 == EV_LEFT;
                const bool granted = left ? !(from_l2 && (cm2 & CELL_ID) < (c & CELL_ID)) : !(from_r2 && (cp2 & CELL_ID) < (c & CELL_ID));
                if (own_event(c, p0, s, left ? xi - 1 : xi + 1, granted)) newc = c;
            }
            if (from_l1 && !(from_r1 && (cp1 & CELL_ID) < (cm1 & CELL_ID))) newc = cm1;
            if (from_r1 && !(from_l1 && (cm1 & CELL_ID) < (cp1 & CELL_ID))) newc = cp1;
            out[0] = newc;
            continue;
        }
        int n_out = 0;
        for (int k = 0; k < K; ++k) {                          // the particles on this site: stay (possibly changed) or leave
            uint32_t c = cellL[(xi + 1) * K + k];
            if (c == CELL_EMPTY) continue;
            const int ev = propL[xi * K + k] & 7;
            const bool hop = ev == EV_LEFT || ev == EV_RIGHT || ev == EV_FWD;
            const int j = ev == EV_LEFT ? xi - 1 : xi + 1;
            const bool granted = hop && rank_at(j, c & CELL_ID) < cap_at(j);
            if (own_event(c, ev, s, j, granted)) out[n_out++] = c;
        }
        const int cap = cap_at(xi);
        for (int k = 0; k < K; ++k) {                          // granted arrivals from the left and right neighbour
            const uint32_t cl_ = cellL[xi * K + k], cr_ = cellL[(xi + 2) * K + k];
            const int el = propL[(xi - 1) * K + k] & 7, er = propL[(xi + 1) * K + k] & 7;
            if (cl_ != CELL_EMPTY && (el == EV_RIGHT || el == EV_FWD) && rank_at(xi, cl_ & CELL_ID) < cap) out[n_out++] = cl_;
            if (cr_ != CELL_EMPTY && er == EV_LEFT && rank_at(xi, cr_ & CELL_ID) < cap) out[n_out++] = cr_;
        }
        for (int k = n_out; k < K; ++k) out[k] = CELL_EMPTY;
    }
}
