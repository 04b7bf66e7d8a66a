// xr_dial3.h — XR-Maze v1/v2, LDS form of the frontier router, round 3.  Included by xr_kernels.hip only
// (device code, gfx950 / wave64).  Same results as xr_dial.h / the oracle, bit for bit; a different machine underneath.
//
// What round 2's router spent its time on (profiles/r02_m_route_phase_cycles.txt, 158 k cycles per route): a mask scan per
// round (17 %), barrier skew between the four waves of a round (22 %), a two-hops-per-round-trip back-trace that re-derives
// every predecessor (18 %), a pocket flood per route (part of 14 % set-up).  None of that is the search itself — ~200-350 node
// expansions per route.  This form removes them:
//
//  * ONE wave searches (16 quads, one lane per direction).  The sim (tools/sim/dial3_sim.py) says 16 quads need 41 lock-step
//    hop iterations per route where 64 need 33: the frontier of an A* band is a few nodes wide.  With one wave there is no
//    workgroup barrier inside a search, no cross-wave atomics, and every queue counter is a wave-uniform scalar.  The other waves
//    build the field, then wait at the closing barrier (a parked wave costs nothing); which wave searches rotates with the env id so
//    that the workgroups of a CU do not pile onto one SIMD.
//  * Explicit queues instead of a mask scan per round.  Keys f = d + h live on a fixed grid of buckets of width 2^dshift:
//    `cur` = nodes of the bucket being expanded, `later` = the HOT part of the frontier as (node, bucket) entries (bucket <
//    current + XR3_HOTW), everything else (`cold`: leftovers of earlier searches, overflow, far keys) is a node bitmask with a
//    lower bound of its keys — h only grows from one search of a net to the next, so an old lower bound stays one.  A round =
//    min bucket of the hot list (one DPP reduction) + one partition pass; the cold mask is classified only when the frontier
//    reaches its lower bound.  Capacities never affect the result: what does not fit goes cold.
//  * The field word carries the predecessor direction:  dist << 5 | pdir << 2 | held << 1 | valid.  Lowering is a min() on the
//    WORD, so among candidates of equal distance the lowest pdir (E,S,W,N,U,D = 0..5, the spec's back-trace order,
//    reference baseline/build_3Dgrid.py:127) wins: when the search stops every node with d + h <= best holds its exact distance AND
//    its first tight predecessor (every tight predecessor u of such a node has d(u) + h(u) <= best too, so it was expanded with
//    its final distance and its candidate word took part in the min).  The back-trace is a pointer chase: one LDS read per node.
//    27 distance bits: the host only selects this form when (N + 1) * (longest edge + penalty) < 2^27 (else: xr_dial.h).
//  * Isolated pins (closed pockets) and the per-net constants (first pin, number of pins) are static: decided at load
//    (xr_batch_load_regions), one word per net — no flood, no O(nap^2) pin census per route.
//
// Why the results cannot differ: any label-correcting order reaches the same pruned fixpoint (xr_dial.h's argument: for every edge
// u->v, word(v) <= candidate(u, v) or u is queued / cold / deferred; the search stops when a lower bound of every pending key
// exceeds `best`).  Duplicate queue entries and stale bucket tags only cost a wasted expansion.
#pragma once

// capacities XR3_CAPC / XR3_CAPL / XR3_TMP and XR3_LDS_BYTES: xr_device.h (the host sizes the launch from them)
#ifndef XR3_HOTW
#define XR3_HOTW 3            // buckets ahead of the current one that are kept as list entries
#endif
#define XR3_UNREACHED 0xFFFFFFFDu      // | held << 1   (dist bits all ones, pdir 7)
#define XR3_DMAX 0x07FFFFFFu           // distance of an unreached word
#define XR3_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

#ifndef XR3_WAVE_MIN_SHFL
// wave-wide min by DPP (row_shr 1,2,4,8 -> lane 15 of every row; row_bcast:15, row_bcast:31 -> lane 63), ~12 VALU ops, no LDS
__device__ __forceinline__ uint32_t xr3_wave_min(uint32_t v) {
    const int id = -1;        // identity of min on u32
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x111, 0xF, 0xF, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x112, 0xF, 0xF, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x114, 0xF, 0xF, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x118, 0xF, 0xF, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x142, 0xA, 0xF, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x143, 0xC, 0xF, false); v = t < v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
#else
__device__ __forceinline__ uint32_t xr3_wave_min(uint32_t v) { return xr_wave_min_u32(v); }
#endif

__device__ __forceinline__ int xr3_mbcnt(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// V2: XR-Maze v2 knobs compiled in (guide cost, rip-up-and-reroute), as in xr_dial.h
template <bool V2>
__device__ __forceinline__ void xr_dial3_route_env(const XrBatchDev& b, const int e, const int a, char* smem) {
    __shared__ unsigned short s_ap_f[XR_MAX_AP_PER_NET];
    __shared__ short s_ap_pin[XR_MAX_AP_PER_NET];
    __shared__ unsigned char s_ap_conn[XR_MAX_AP_PER_NET];      // 0 target, 1 connected, 2 isolated (static, from the load)
    __shared__ uint32_t s_cold_lb;                              // lower bound of the keys of the cold set (XR_DIAL_INF: empty)
    __shared__ uint32_t s_best, s_bnew;                         // per round: smallest target distance, smallest hot bucket
    __shared__ int s_hb[6];                                     // bounding box of the unconnected targets: x, y (coordinates x32), z
    __shared__ int s_cnt;                                       // mask compaction: next free slot of s_tmp
    __shared__ int s_nLc[2];                                    // entries of the hot list (current / being rebuilt)
    __shared__ int s_nc[3], s_qh[3];                            // bucket queues: entries, next entry to hand out
    __shared__ int s_remaining, s_abort;
    __shared__ int s_gb[4], s_retry;                            // XR-Maze v2: guide box of the net (track indices), rip-up decision

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    if (!xr_step_prologue(b, e, a)) return;

    XR_T0();
    const XrRegionDev R = b.regions[b.env_region[e]];
    const int X = R.X, Y = R.Y, Z = R.Z, N = R.N;
    const int YZ = Y * Z;
    const uint32_t ldir = R.ldir_mask;
    const int mw = (N + 31) >> 5;
    const uint32_t umw = (uint32_t)mw, magic_mw = R.magic_mw;
    const uint32_t uYZ = (uint32_t)YZ, uZ = (uint32_t)Z;
    const int round_cap = b.round_cap > 0 ? b.round_cap : 1024 + N;
    const int16_t* __restrict__ node_net = b.rg_node_net + R.node_off;
    int16_t* __restrict__ owner = b.owner + (int64_t)e * b.n_max;

    // LDS carve:  field u32[n_max] | cold u32[mw_max] | defer u32[mw_max] | later u32[CAPL] | tab u32[x_max+2 + y_max+2] |
    //             cur u16[3][CAPC] | tmp u16[TMP]
    uint32_t* field = reinterpret_cast<uint32_t*>(smem);
    const int mw_max = (b.n_max >> 5) + 1;
    uint32_t* s_cold = field + b.n_max;
    uint32_t* s_defer = s_cold + mw_max;          // also: "to be classified when the next search starts" (new sources)
    uint32_t* s_later = s_defer + mw_max;
    // coordinate tables (x32, relative to the first track): tab[k] = 32*(xs[clamp(k-1)] - xs[0]), k = 0 .. X+1 (the coordinate of
    // track x is tab[x+1]; padded at both ends); the y table follows at XO
    uint32_t* s_tab = s_later + XR3_CAPL;
    const int XO = b.x_max + 2;
    unsigned short* s_cur = reinterpret_cast<unsigned short*>(s_tab + XO + b.y_max + 2);
    unsigned short* s_tmp = s_cur + 3 * XR3_CAPC;

    // loads that depend on (e, a) only: issued now, consumed after the grid build
    const int ap_lo = b.net_csr[R.net_off + a], ap_hi = b.net_csr[R.net_off + a + 1];
    const int nap = ap_hi - ap_lo;    // 1 <= nap <= XR_MAX_AP_PER_NET (checked at load)
    const int ninfo = b.net_info[R.net_off + a];               // first pin | pins << 14 | isolated pins << 22 | first pin isolated << 30
    int my_ap_f = 0, my_ap_pin = 0, my_ap_iso = 0;
    if (tid < nap) {
        my_ap_f = b.ap_node[R.ap_off + ap_lo + tid]; my_ap_pin = b.ap_pin[R.ap_off + ap_lo + tid];
        my_ap_iso = b.ap_flags[R.ap_off + ap_lo + tid];
    }
    uint32_t my_xc = 0, my_yc = 0;
    if (tid <= X + 1) my_xc = (uint32_t)(b.coords[R.xs_off + min(max(tid - 1, 0), X - 1)] - b.coords[R.xs_off]) << 5;
    if (tid <= Y + 1) my_yc = (uint32_t)(b.coords[R.ys_off + min(max(tid - 1, 0), Y - 1)] - b.coords[R.ys_off]) << 5;
    const int first_pin = ninfo & 0x3FFF, npins = (ninfo >> 14) & 0xFF;
    const int n_isolated = (ninfo >> 30) & 1 ? npins - 1 : (ninfo >> 22) & 0xFF;      // unreachable pins known up front

    // ---- grid build: field word of every node for THIS net (16-byte loads of node_net / owner, four chunks in flight) ----
    auto build_field = [&]() __attribute__((always_inline)) {
        const int nchunk = (N + 7) >> 3;
        for (int c0 = tid; c0 < nchunk; c0 += 4 * nthr) {
            int4 vn[4], vo[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ci = c0 + u * nthr;
                if (ci < nchunk) {
                    vn[u] = *reinterpret_cast<const int4*>(node_net + (ci << 3));
                    vo[u] = *reinterpret_cast<const int4*>(owner + (ci << 3));
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ci = c0 + u * nthr;
                if (ci >= nchunk) continue;
                const int f0 = ci << 3;
                const int pn[4] = {vn[u].x, vn[u].y, vn[u].z, vn[u].w}, po[4] = {vo[u].x, vo[u].y, vo[u].z, vo[u].w};
                uint32_t w[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int nn = (int)(short)((j & 1) ? (pn[j >> 1] >> 16) : (pn[j >> 1] & 0xFFFF));
                    const int ow = (int)(short)((j & 1) ? (po[j >> 1] >> 16) : (po[j >> 1] & 0xFFFF));
                    const uint32_t ww = XR3_UNREACHED | (((ow != 0 && ow != a) || (nn > 0 && nn != a)) ? 2u : 0u);
                    w[j] = (nn == -1 || f0 + j >= N) ? 0u : ww;
                }
                uint4* dst = reinterpret_cast<uint4*>(field + f0);
                dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
                dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
            }
        }
        for (int i = tid; i < mw; i += nthr) { s_cold[i] = 0; s_defer[i] = 0; }
    };
    build_field();
    if (tid <= X + 1) s_tab[tid] = my_xc;
    if (tid <= Y + 1) s_tab[XO + tid] = my_yc;
    for (int i = tid + nthr; i <= X + 1; i += nthr)
        s_tab[i] = (uint32_t)(b.coords[R.xs_off + min(i - 1, X - 1)] - b.coords[R.xs_off]) << 5;
    for (int i = tid + nthr; i <= Y + 1; i += nthr)
        s_tab[XO + i] = (uint32_t)(b.coords[R.ys_off + min(i - 1, Y - 1)] - b.coords[R.ys_off]) << 5;
    if (tid == 0) { s_gb[0] = 0x7FFFFFFF; s_gb[1] = -1; s_gb[2] = 0x7FFFFFFF; s_gb[3] = -1; s_cold_lb = XR_DIAL_INF; }
    if (V2 && b.guide_cost) __syncthreads();
    for (int i = tid; i < nap; i += nthr) {
        const int pin = i < nthr ? my_ap_pin : (int)b.ap_pin[R.ap_off + ap_lo + i];
        const int apf = i < nthr ? my_ap_f : b.ap_node[R.ap_off + ap_lo + i];
        const int iso = i < nthr ? my_ap_iso : (int)b.ap_flags[R.ap_off + ap_lo + i];
        s_ap_f[i] = (unsigned short)apf;
        s_ap_pin[i] = (short)pin;
        s_ap_conn[i] = (unsigned char)((iso & 1) ? 2 : (pin == first_pin ? 1 : 0));
        if (V2 && b.guide_cost) {                // XR-Maze v2: the net's guide = bounding box of all its access points (+ margin)
            const int gy = (apf / Z) % Y, gx = apf / YZ;
            atomicMin(&s_gb[0], gx); atomicMax(&s_gb[1], gx); atomicMin(&s_gb[2], gy); atomicMax(&s_gb[3], gy);
        }
    }
    __syncthreads();
    XR_LAP(0);

    const int lane = tid & 63, wv = tid >> 6;
    // the wave that selects targets and traces paths (everything else is done by the whole workgroup): rotates with the env
#if defined(XR_PHASE_TIMING)
    const int sw = XR_TIMING_TID >> 6;
#elif defined(XR3_SW0)
    const int sw = 0;
#else
    const int sw = (int)(((uint32_t)e * 0x9E3779B1u) >> 16) % (nthr >> 6);
#endif

    const uint32_t via5 = (uint32_t)b.via_cost << 5;
    uint32_t pen5 = (uint32_t)b.pen_cost << 5;              // (XR-Maze v2: doubled by every rip-up-and-reroute attempt)
    int dshift;                                             // bucket width 2^dshift ~ dial_mult x the smallest edge length
    {
        const uint32_t dl = max(R.w_min * (uint32_t)b.dial_mult, 1u);
        dshift = 31 - __clz((int)dl);
        if (dshift > 0 && ((dl >> (dshift - 1)) & 1u)) dshift += 1;          // (>= 1.5 x 2^k rounds up)
        dshift = min(max(dshift, 11), 20);                   // bucket tags are 16 bits: key < 2^27
    }
    const uint32_t guide5 = V2 ? (uint32_t)b.guide_cost << 5 : 0u;
    const int gx0 = s_gb[0] - b.guide_margin, gx1 = s_gb[1] + b.guide_margin, gy0 = s_gb[2] - b.guide_margin, gy1 = s_gb[3] + b.guide_margin;
    auto guide_of = [&](int x, int y) __attribute__((always_inline)) -> uint32_t {
        if (!V2) return 0u;
        return (guide5 != 0u && (x < gx0 || x > gx1 || y < gy0 || y > gy1)) ? guide5 : 0u;
    };
    const int16_t claim_val = (int16_t)(V2 && b.maze_end_iter > 1 ? -a : a);   // rip-up: claims are tentative (-a) until the attempt stands
    int32_t* __restrict__ path = b.path + (int64_t)e * b.path_cap;
    int attempt = 0;

    // per-lane constants of a quad: lane 4g + d relaxes direction d of quad g's node: 0 +planar, 1 -planar, 2 +z, 3 -z
    const int dir = lane & 3, qbase = lane & ~3;
    const int sgn = (dir & 1) ? -1 : 1;
    const bool planar = dir < 2;
    const int stepH = planar ? sgn * YZ : sgn, stepV = planar ? sgn * Z : sgn;           // flat-index step on a horizontal / vertical layer
    const int limH = planar ? X : Z, limV = planar ? Y : Z;
    // predecessor direction stored in the NEIGHBOUR's word (pointing back at the quad's node), E,S,W,N,U,D = 0..5
    const uint32_t pdH = (planar ? (sgn > 0 ? 2u : 0u) : (sgn > 0 ? 5u : 4u)) << 2, pdV = (planar ? (sgn > 0 ? 1u : 3u) : (sgn > 0 ? 5u : 4u)) << 2;

    auto node_xyz = [&](uint32_t f, int& x, int& y, int& z) __attribute__((always_inline)) {
        uint32_t ux, ur, uy, uz;
        xr_divmod(f, uYZ, R.magic_yz, ux, ur);
        xr_divmod(ur, uZ, R.magic_z, uy, uz);
        x = (int)ux; y = (int)uy; z = (int)uz;
    };
    auto mask_or = [&](uint32_t* mask, uint32_t f) __attribute__((always_inline)) {
        uint32_t q, r;
        xr_divmod(f, umw, magic_mw, q, r);
        atomicOr(&mask[r], 1u << q);
    };
    // append by the lanes with `pred` to a list whose length is the LDS counter `cnt`: one atomic per wave.  Returns the lane's
    // slot (valid where pred); the counter may run past the capacity — readers clamp it.
    auto block_append = [&](bool pred, int* cnt) __attribute__((always_inline)) -> int {
        const unsigned long long m = __ballot(pred);
        if (m == 0ULL) return 0;
        const int first = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == first) base = atomicAdd(cnt, (int)__popcll(m));
        base = __builtin_amdgcn_readlane(base, first);
        return base + xr3_mbcnt(m);
    };

    for (;;) {                                              // attempts (exactly one unless maze_end_iter > 1)
    // sources of the first search: the access points of the lowest pin (classified when the search starts)
    for (int i = tid; i < nap; i += nthr)
        if (s_ap_conn[i] == 1) { field[s_ap_f[i]] &= 3u; mask_or(s_defer, (uint32_t)s_ap_f[i]); }
    if (tid == 0) { s_remaining = npins - 1 - n_isolated; s_nLc[0] = 0; s_nLc[1] = 0; s_abort = 0; }
    // the tracing wave's bookkeeping (uniform over that wave; meaningless in the others)
    int d_vio = 0, d_wl = 0, d_via = 0, plen = 0, status = XR_ENV_OK, d_held = 0, nrounds = 0;
    uint64_t h = wv == sw ? b.hash[e] : 0ULL;
    int par = 0;                                            // which of s_nLc counts the hot list
    __syncthreads();

    while (s_remaining > 0) {                               // uniform: written before the barrier that precedes every test
        // ---- new search: heuristic box = bounding box of the access points of the unconnected (not isolated) pins -------------
        if (tid == 0) { s_hb[0] = 0x7FFFFFFF; s_hb[1] = -0x7FFFFFFF; s_hb[2] = 0x7FFFFFFF; s_hb[3] = -0x7FFFFFFF; s_hb[4] = 0x7FFFFFFF; s_hb[5] = -1; }
        __syncthreads();
        for (int i = tid; i < nap; i += nthr)
            if (!s_ap_conn[i]) {
                int ax, ay, az;
                node_xyz((uint32_t)s_ap_f[i], ax, ay, az);
                const int cx = (int)s_tab[ax + 1], cy = (int)s_tab[XO + ay + 1];
                atomicMin(&s_hb[0], cx); atomicMax(&s_hb[1], cx); atomicMin(&s_hb[2], cy); atomicMax(&s_hb[3], cy);
                atomicMin(&s_hb[4], az); atomicMax(&s_hb[5], az);
            }
        __syncthreads();
        const int hb0 = s_hb[0], hb1 = s_hb[1], hb2 = s_hb[2], hb3 = s_hb[3], hb4 = s_hb[4], hb5 = s_hb[5];
        // h(v): distance to that box — coordinate differences + one via cost per layer (a consistent lower bound), from
        // coordinates x32
        auto heur_c = [&](int xc, int yc, int z) __attribute__((always_inline)) -> uint32_t {
            const int hx = max(0, max(hb0 - xc, xc - hb1)), hy = max(0, max(hb2 - yc, yc - hb3));
            const int hz = max(0, max(hb4 - z, z - hb5));
            return ((uint32_t)(hx + hy) >> 5) + __umul24((uint32_t)hz, (uint32_t)b.via_cost);
        };
        // a node with key `key` (bucket kb) joins the frontier: hot list entry while it is near and there is room, else cold
        uint32_t hotlim = 0x10000u;                      // first bucket that is NOT kept as a list entry (search start: all are)
        auto push_later = [&](bool valid, uint32_t f, uint32_t key, uint32_t kb) __attribute__((always_inline)) {
            const bool hot = valid && kb < hotlim;
            const int pos = block_append(hot, &s_nLc[par]);
            const bool ok = hot && pos < XR3_CAPL;
            if (ok) s_later[pos] = f | (kb << 16);
            if (valid && !ok) { mask_or(s_cold, f); atomicMin(&s_cold_lb, key); }
        };
        // the set bits of a node mask -> classified into the hot list / the cold mask, by the whole workgroup.  Ends with a
        // barrier; `mask` is empty afterwards except for what classification put back (cold scan: what stays cold).
        auto scan_mask = [&](uint32_t* mask) __attribute__((always_inline)) {
            auto classify = [&](bool act, uint32_t f) __attribute__((always_inline)) {
                const uint32_t w = field[f];
                int x, y, z;
                node_xyz(f, x, y, z);
                const uint32_t key = (w >> 5) + heur_c((int)s_tab[x + 1], (int)s_tab[XO + y + 1], z);
                push_later(act, f, key, key >> dshift);
            };
            // compaction: a thread lists the nodes of its words (words tid, tid + nthr, ...) from a slot range it reserves with
            // one atomic, and clears the words; then the list is classified one node per thread per step
            if (tid == 0) s_cnt = 0;
            __syncthreads();
            int cnt = 0;
            for (int wi = tid; wi < mw; wi += nthr) cnt += __popc(mask[wi]);
            int pos = cnt ? atomicAdd(&s_cnt, cnt) : 0;
            __syncthreads();
            const int total = s_cnt;
            if (total <= XR3_TMP) {
                if (cnt)
                    for (int wi = tid; wi < mw; wi += nthr) {
                        uint32_t m = mask[wi];
                        if (m) mask[wi] = 0u;
                        while (m) {
                            const int q = __ffs((int)m) - 1;
                            m &= m - 1;
                            s_tmp[pos++] = (unsigned short)(q * mw + wi);        // (transposed bit order: node = bit * mw + word)
                        }
                    }
                __syncthreads();
                for (int i0 = 0; i0 < total; i0 += nthr) {
                    const bool act = i0 + tid < total;
                    classify(act, act ? (uint32_t)s_tmp[i0 + tid] : 0u);
                }
                __syncthreads();
                return;
            }
            // more nodes than the list holds (rare): XR3_TMP / 32 words at a time.  Every word is taken exactly once — a node
            // that classification puts BACK into this mask (cold scan) lands in a word already done (it stays) or still to
            // come (it is classified again: harmless) — so the scan ends whatever the capacities are.
            for (int w0 = 0; w0 < mw; w0 += XR3_TMP / 32) {
                __syncthreads();
                if (tid == 0) s_cnt = 0;
                __syncthreads();
                const int wi = w0 + tid;
                uint32_t m = (tid < XR3_TMP / 32 && wi < mw) ? mask[wi] : 0u;
                int p = m ? atomicAdd(&s_cnt, __popc(m)) : 0;
                if (m) mask[wi] = 0u;
                while (m) {
                    const int q = __ffs((int)m) - 1;
                    m &= m - 1;
                    s_tmp[p++] = (unsigned short)(q * mw + wi);
                }
                __syncthreads();
                const int tot = s_cnt;
                for (int i0 = 0; i0 < tot; i0 += nthr) {
                    const bool act = i0 + tid < tot;
                    classify(act, act ? (uint32_t)s_tmp[i0 + tid] : 0u);
                }
            }
            __syncthreads();
        };

        // search start: new sources + deferred nodes are classified (the cold set stays cold: its lower bound still holds)
        scan_mask(s_defer);
        XR_LAP(1);

        bool aborted = false;
        for (int nsr = 0;; nsr++) {
            // ---- bucket advance: bound, smallest hot bucket, the cold set's lower bound ---------------------------------------
            if (tid == 0) {
                s_best = XR3_DMAX; s_bnew = 0xFFFFu; s_nLc[par ^ 1] = 0;
                s_nc[0] = 0; s_nc[1] = 0; s_nc[2] = 0; s_qh[0] = 0; s_qh[1] = 0; s_qh[2] = 0;
            }
            __syncthreads();
            const int nL = min(s_nLc[par], XR3_CAPL);
            {
                uint32_t lb = XR3_DMAX, tmin = 0xFFFFu;     // smallest tentative distance of an unconnected target; smallest hot bucket
                for (int i = tid; i < nap; i += nthr)
                    if (!s_ap_conn[i]) { const uint32_t d = field[s_ap_f[i]] >> 5; lb = d < lb ? d : lb; }
                for (int i = tid; i < nL; i += nthr) { const uint32_t t = s_later[i] >> 16; tmin = t < tmin ? t : tmin; }
                if (wv * 64 < nap) { lb = xr3_wave_min(lb); if (lane == 0 && lb != XR3_DMAX) atomicMin(&s_best, lb); }
                if (wv * 64 < nL) { tmin = xr3_wave_min(tmin); if (lane == 0) atomicMin(&s_bnew, tmin); }
            }
            __syncthreads();
            const uint32_t best = s_best;                 // XR3_DMAX: no target reached yet
            const uint32_t bnew = s_bnew;                 // 0xFFFF: the hot list is empty
            const uint32_t cold_lb = s_cold_lb;
            const bool hot_any = nL > 0, cold_any = cold_lb != XR_DIAL_INF;
            const uint32_t hot_lb = hot_any ? (bnew << dshift) : XR_DIAL_INF;
            const uint32_t all_lb = hot_lb < cold_lb ? hot_lb : cold_lb;
            if (all_lb == XR_DIAL_INF || (best != XR3_DMAX && all_lb > best)) break;       // exhausted, or every pending key > best
            if (nsr >= round_cap || s_abort) { aborted = true; break; }
#ifdef XR3_NO_ROOM_RULE
            if (cold_any && (!hot_any || (cold_lb >> dshift) <= bnew)) {
#else
            if (cold_any && (!hot_any || ((cold_lb >> dshift) <= bnew && nL <= XR3_CAPL / 2))) {
#endif
                // the frontier reached the cold set's lower bound (and the hot list has room): classify the cold set; what
                // stays cold gets an exact bound.  No room: the hot bucket goes first (order never affects the result).
                const uint32_t cb = cold_lb >> dshift;
                hotlim = (hot_any && bnew < cb ? bnew : cb) + XR3_HOTW;
                __syncthreads();                                  // (every thread has read s_cold_lb)
                if (tid == 0) s_cold_lb = XR_DIAL_INF;
                scan_mask(s_cold);                                // (begins and ends with a barrier)
                continue;
            }
            const int bcur = (int)bnew;
            hotlim = bnew + XR3_HOTW;
            const uint32_t hi = (bnew + 1u) << dshift;
            if (wv == sw) nrounds++;
#ifdef XR_PHASE_TIMING
            if (tid == XR_TIMING_TID) _ph[7] += 1;
#endif
            // ---- partition: entries of this bucket -> queue 0, the rest rebuilt into the list (other counter) -----------------
            {
                constexpr int KP = (XR3_CAPL + 63) / 64;         // entries per thread when the workgroup is a single wave
                uint32_t ent[KP];
#pragma unroll
                for (int k = 0; k < KP; k++) { const int i = tid + k * nthr; ent[k] = i < nL ? s_later[i] : 0xFFFFFFFFu; }
                __syncthreads();                                  // all reads before the list is rewritten
#pragma unroll
                for (int k = 0; k < KP; k++) {
                    if (k * nthr >= nL) break;                    // uniform
                    const bool have = tid + k * nthr < nL;
                    const bool isc = have && (ent[k] >> 16) == bnew;
                    const int pc = block_append(isc, &s_nc[0]);
                    const bool toc = isc && pc < XR3_CAPC;
                    if (toc) s_cur[pc] = (unsigned short)(ent[k] & 0xFFFFu);
                    const bool keep = have && !toc;               // (a bucket larger than the queue: the rest stays in the list)
                    const int pk = block_append(keep, &s_nLc[par ^ 1]);
                    if (keep) s_later[pk] = ent[k];
                }
                par ^= 1;
                __syncthreads();
            }
            XR_LAP(6);
            // ---- the bucket: sub-rounds over three rotating queues (take from one, in-bucket nodes go to the next, the third is
            // reset); within a sub-round every wave's quads follow chains and take queue entries until the queue is drained ------
            for (int sub = 0;; sub++) {
                const int cbuf = sub % 3, nbuf = (sub + 1) % 3, rbuf = (sub + 2) % 3;
                const int ncur = min(s_nc[cbuf], XR3_CAPC);
                if (ncur == 0) break;                             // uniform
                if (sub >= round_cap) {                           // hang guard: what is queued goes cold, the search aborts
                    for (int i = tid; i < ncur; i += nthr) { mask_or(s_cold, (uint32_t)s_cur[cbuf * XR3_CAPC + i]); atomicMin(&s_cold_lb, 0u); }
                    if (tid == 0) s_abort = 1;
                    break;
                }
                if (tid == 0) { s_nc[rbuf] = 0; s_qh[rbuf] = 0; }
#ifdef XR3_COUNT
                if (tid == XR_TIMING_TID) { _ph[1] += 1; _ph[3] += ncur; }      // sub-rounds; queue entries
#endif
                const unsigned short* qcur = s_cur + cbuf * XR3_CAPC;
                unsigned short* qnxt = s_cur + nbuf * XR3_CAPC;
                int gf = -1, gx = 0, gy = 0, gz = 0;
                bool drained = false;
                for (int nhop = 0;; nhop++) {
                    if (nhop >= round_cap) {                      // hang guard (every hop lowers a field word: finite anyway)
                        if (gf >= 0 && dir == 0) { mask_or(s_cold, (uint32_t)gf); atomicMin(&s_cold_lb, 0u); s_abort = 1; }
                        break;
                    }
                    const unsigned long long idle_g = __ballot(gf < 0) & 0x1111111111111111ULL;       // one bit per idle quad
                    if (idle_g != 0ULL && !drained) {
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&s_qh[cbuf], (int)__popcll(idle_g));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base >= ncur) drained = true;
                        else if (gf < 0) {
                            const int idx = base + (int)__popcll(idle_g & ((1ULL << qbase) - 1ULL));
                            if (idx < ncur) { gf = (int)qcur[idx]; node_xyz((uint32_t)gf, gx, gy, gz); }
                        }
                    }
                    if (__ballot(gf >= 0) == 0ULL) break;                                            // this wave: queue drained, no chain alive
#ifdef XR3_COUNT
                    if (tid == XR_TIMING_TID) { _ph[4] += 1; _ph[5] += (long long)__popcll(__ballot(gf >= 0)) >> 2; }   // hop iterations; active quads (this wave)
#endif
                    // ---- one hop of every active quad: all LDS reads together (safe addresses), ALU, one atomic ------------
                    const bool act = gf >= 0;
                    const int gfs = act ? gf : 0;
                    const bool vert = (ldir >> gz) & 1u;
                    const int c = planar ? (vert ? gy : gx) : gz;
                    const int lim = vert ? limV : limH;
                    const bool inb = act && (unsigned)(c + sgn) < (unsigned)lim;
                    const int nf = inb ? gfs + (vert ? stepV : stepH) : gfs;
                    const uint32_t gw = field[gfs], wn = field[nf];
                    const uint32_t xq = s_tab[gx + 1], yq = s_tab[XO + gy + 1];
                    const uint32_t cb = s_tab[(vert ? XO : 0) + (planar ? c + 1 + sgn : 0)];           // neighbour's coordinate along the layer's axis
                    const uint32_t ca = vert ? yq : xq;
                    const uint32_t dlt = cb - ca, adl = (int)dlt < 0 ? 0u - dlt : dlt;
                    const uint32_t len5 = planar ? adl : via5;
                    const int nx = gx + ((planar && !vert) ? sgn : 0), ny = gy + ((planar && vert) ? sgn : 0), nz = gz + (planar ? 0 : sgn);
                    const uint32_t cand5 = (gw & ~31u) + len5 + ((wn & 2u) ? pen5 : 0u) + guide_of(nx, ny);
                    const uint32_t cw = cand5 | (vert ? pdV : pdH) | (wn & 3u);
                    const uint32_t key = (cand5 >> 5) + heur_c((int)((planar && !vert) ? cb : xq), (int)((planar && vert) ? cb : yq), nz);   // f = d + h
                    // blockage, or no improvement of the WORD (distance, then predecessor direction): nothing to do
                    const bool go = inb && wn != 0u && gw < 0xFFFFFFE0u && cand5 < 0xFFFFFFC0u && cw < wn;
                    const bool refused = go && best != XR3_DMAX && key > best;                  // bound pruning (on f)
                    bool improved = false;
                    if (go && !refused) {
                        const uint32_t old = atomicMin(&field[nf], cw);
                        improved = (cw >> 5) < (old >> 5);                                     // the DISTANCE went down (not only the direction)
                    }
                    const bool chain_cand = improved && key < hi;
                    const uint32_t c4 = (uint32_t)(__ballot(chain_cand) >> qbase) & 15u;
                    const int win = c4 ? __ffs((int)c4) - 1 : -1;                              // the chain goes on with the first lowered direction
                    const bool psh = improved && dir != win;
                    if (__ballot(psh) != 0ULL) {                                               // the others join the frontier
                        const uint32_t kb = key >> dshift;
                        const bool wantc = psh && kb == (uint32_t)bcur;
                        const int pc = block_append(wantc, &s_nc[nbuf]);
                        const bool toc = wantc && pc < XR3_CAPC;
                        if (toc) qnxt[pc] = (unsigned short)nf;
                        push_later(psh && !toc, (uint32_t)nf, key, kb);
                    }
                    const unsigned long long mR = __ballot(refused);
                    if (mR != 0ULL) {                                                          // an edge refused by the bound: the node is looked at again
                        const uint32_t r4 = (uint32_t)(mR >> qbase) & 15u;
                        if (r4 && dir == 0 && act) mask_or(s_defer, (uint32_t)gf);
                    }
                    if (act) {
                        if (win >= 0) {                                 // every lane of the quad moves to the winner's node
                            const int sw_ = (win & 1) ? -1 : 1;
                            const int wx = (win < 2 && !vert) ? sw_ : 0, wy = (win < 2 && vert) ? sw_ : 0, wz = win < 2 ? 0 : sw_;
                            gf += wx * YZ + wy * Z + wz;
                            gx += wx; gy += wy; gz += wz;
                        } else gf = -1;
                    }
                }
                __syncthreads();
            }
            XR_LAP(2);
        }
        // what is still in the hot list goes cold: its keys are >= the tag's bucket edge, and keys only grow from here on
        {
            const int nL = min(s_nLc[par], XR3_CAPL);
            for (int i = tid; i < nL; i += nthr) {
                const uint32_t ent = s_later[i];
                mask_or(s_cold, ent & 0xFFFFu);
                atomicMin(&s_cold_lb, (ent >> 16) << dshift);
            }
        }
        __syncthreads();
        if (tid == 0) { s_nLc[par] = 0; }
        XR_LAP(6);

        // ===================== the tracing wave: target, back-trace, new sources (the others wait at the barrier) ==========
        if (wv == sw) {
            int remaining = s_remaining;
            if (aborted) {                        // round cap: the remaining pins are charged as unreachable, nothing is traced
                d_vio += remaining; status |= XR_ENV_ROUTER_ABORT | XR_ENV_UNREACHABLE; remaining = 0;
            } else {
                // ---- nearest access point of an unconnected pin; ties -> lowest flat index --------------------------------
                uint32_t md = XR3_DMAX;
                for (int i = lane; i < nap; i += 64)
                    if (!s_ap_conn[i]) { const uint32_t d = field[s_ap_f[i]] >> 5; md = d < md ? d : md; }
                const uint32_t bd = xr3_wave_min(md);
                if (bd == XR3_DMAX) {             // every remaining pin unreachable
                    d_vio += remaining; status |= XR_ENV_UNREACHABLE; remaining = 0;
                } else {
                    uint32_t mf = 0xFFFFFFFFu;
                    for (int i = lane; i < nap; i += 64)
                        if (!s_ap_conn[i] && (field[s_ap_f[i]] >> 5) == bd) { const uint32_t f = s_ap_f[i]; mf = f < mf ? f : mf; }
                    const int tf = (int)xr3_wave_min(mf);
                    int tpin = 0;
                    for (int i0 = 0; i0 < nap; i0 += 64) {
                        const int i = i0 + lane;
                        const unsigned long long mm = __ballot(i < nap && (int)s_ap_f[i] == tf);
                        if (mm) { tpin = (int)s_ap_pin[i0 + __ffsll((long long)mm) - 1]; break; }        // (node ids are unique per net)
                    }
                    // ---- back-trace: a pointer chase along the predecessor directions (every lane runs the same chase: the
                    // reads are broadcasts; lane 0 records).  Path nodes are listed in s_tmp and become sources / get claimed
                    // in parallel afterwards.
                    int v = tf, np = 0;
                    uint32_t vw = field[v];
                    int x = 0, y = 0, z = 0;
                    if (V2) node_xyz((uint32_t)v, x, y, z);
                    auto flush = [&]() __attribute__((always_inline)) {        // the listed path nodes: sources of the next search, claimed if nobody holds them
                        XR3_WSYNC();
                        for (int i = lane; i < np; i += 64) {
                            const uint32_t f = s_tmp[i];
                            field[f] &= 3u;
                            mask_or(s_defer, f);
                            if (owner[f] == 0) owner[f] = claim_val;
                        }
                        XR3_WSYNC();
                        np = 0;
                    };
                    for (int nt = 0; (vw >> 5) != 0u; nt++) {
                        const uint32_t pd = (vw >> 2) & 7u;
                        if (nt > N || pd > 5u) { status |= 0x100; break; }          // (distances strictly decrease: cannot happen)
                        const int off = pd == 0u ? YZ : pd == 1u ? -Z : pd == 2u ? -YZ : pd == 3u ? Z : pd == 4u ? 1 : -1;
                        const int u = v + off;
                        const uint32_t uw = field[u];
                        const uint32_t step5 = (vw & ~31u) - (uw & ~31u) - ((vw & 2u) ? pen5 : 0u) - guide_of(x, y);   // the edge itself, x32
                        if (vw & 2u) { d_vio += 1; d_held += 1; }
                        if (pd >= 4u) d_via += 1; else d_wl += (int)(step5 >> 5);
                        if (lane == 0) { if (plen < b.path_cap) path[plen] = v; s_tmp[np] = (unsigned short)v; }
                        plen++; np++;
                        fnv_mix(h, (uint32_t)v);
                        if (V2) { x += (pd == 0u) - (pd == 2u); y += (pd == 3u) - (pd == 1u); }
                        v = u; vw = uw;
                        if (np == XR3_TMP) flush();
                    }
                    flush();
                    if (status & 0x100) remaining = 0;
                    else {
                        // terminal node of the component: claimed (and recorded) only if nobody holds it yet
                        if (owner[v] == 0) {
                            if (lane == 0) { owner[v] = claim_val; if (plen < b.path_cap) path[plen] = v; }
                            plen++;
                            fnv_mix(h, (uint32_t)v);
                        }
                        remaining -= 1;
                        // the reached pin joins the component with all of its access points
                        for (int i = lane; i < nap; i += 64)
                            if (s_ap_pin[i] == (short)tpin) { s_ap_conn[i] = 1; field[s_ap_f[i]] &= 3u; mask_or(s_defer, (uint32_t)s_ap_f[i]); }
                    }
                }
            }
            if (lane == 0) s_remaining = remaining;
        }
        __syncthreads();
        XR_LAP(3);
    }
    // ---- does the attempt stand (XR-Maze v2)?  Its path uses a node held by another net and attempts are left: rip it up ----
    if (wv == sw) {
        const bool retry = V2 && b.maze_end_iter > 1 && d_held > 0 && attempt + 1 < b.maze_end_iter;
        if (V2 && lane == 0) s_retry = retry ? 1 : 0;
        if (!retry && lane == 0) {
            if (n_isolated > 0) { d_vio += n_isolated; status |= XR_ENV_UNREACHABLE; }
            xr_step_epilogue(b, e, a, d_vio, d_wl, d_via, plen, status, nrounds, h);
        }
    }
    XR_LAP(5);
    if (!V2 || b.maze_end_iter <= 1) break;
    __syncthreads();                                          // the attempt's owner writes and s_retry are visible
    const bool retry = s_retry != 0;
    for (int f = tid; f < N; f += nthr)                       // tentative claims: accepted (-a -> a) or undone (-a -> 0)
        if (owner[f] == (int16_t)-a) owner[f] = retry ? (int16_t)0 : (int16_t)a;
    if (!retry) break;
    attempt++;
    pen5 = ((uint32_t)b.pen_cost << 5) << attempt;
    __syncthreads();                                          // the owner grid is clean again before the field is rebuilt
    build_field();
    if (tid == 0) s_cold_lb = XR_DIAL_INF;
    for (int i = tid; i < nap; i += nthr) s_ap_conn[i] = (unsigned char)(s_ap_conn[i] == 2 ? 2 : (s_ap_pin[i] == (short)first_pin ? 1 : 0));
    __syncthreads();
    }
    XR_TDUMP();
}
